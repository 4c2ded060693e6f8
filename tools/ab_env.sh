#!/bin/bash
# bench frames/s (300 steps and the driver's 20) under environment settings, each run twice:  tools/ab_env.sh "SAS_CU_SPLIT=32" "SAS_CU_SPLIT=48" ""
for e in "$@" "$@"; do
  a=$(env $e timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras --steps 300 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['cold_start']['value']), round(d['roofline']['kernel_ms'],4))")
  b=$(env $e timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['door_a_sync']['value']), round(d['single_view_async']['value']))")
  echo "[$e] steps300: value cold tile_ms = $a | steps20: value doorA single = $b"
done
