#!/bin/bash
# A/B of prebuilt library variants on one GPU box: tools/ab.sh base varA varB ...  (variants/lib_<name>.so)
# Each variant: isolated tile-kernel time (stage_probe), bench fps, cfg2/cfg5 fps; the list is run twice.
for v in "$@" "$@"; do
  cp variants/lib_$v.so sim_a_splat_amd/libsas_hip.so || exit 1
  blend=$(timeout -k 10 200 python tools/stage_probe.py --cfg 3 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['stage_ms']['blend'],4))")
  fps=$(timeout -k 10 300 python bench.py --no-cpu-baseline --steps 400 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['value'],1))")
  others=$(timeout -k 10 100 python tools/config_fps.py 2 5 2>/dev/null | cut -d">" -f2 | tr "\n" " ")
  echo "$v blend_ms=$blend bench_fps=$fps $others"
done
