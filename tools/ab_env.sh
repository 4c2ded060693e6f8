#!/bin/bash
# Isolated kernel times (stage_probe, config 3) and the pair bench of the in-tree library under values of ONE environment knob,
# on one GPU box, the list twice:   tools/ab_env.sh VAR value...      ("-" = unset)
var=$1; shift
for val in "$@" "$@"; do
  if [ "$val" = "-" ]; then unset $var; else export $var=$val; fi
  st=$(timeout -k 10 200 python tools/stage_probe.py --cfg 3 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['stage_ms']['project'],4), round(d['stage_ms']['blend'],4), round(d['stage_ms']['total'],4))")
  fps=$(timeout -k 10 300 python bench.py --no-cpu-baseline --steps 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['door_a_sync']['value'],1), round(d['door_a_async']['value'],1), round(d['cold_start']['value'],1))")
  echo "$var=$val project_ms,tile_ms,total_ms=$st bench_fps,door_a_sync,door_a_async,cold=$fps"
done
