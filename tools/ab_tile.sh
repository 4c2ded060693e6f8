#!/bin/bash
# Isolated projection and tile-kernel time (stage_probe, config 3) and bench frames/s of prebuilt library variants on one
# GPU box: tools/ab_tile.sh [--bench] name...   ("prod" = the in-tree library); the list is run twice.
bench=0; if [ "$1" = "--bench" ]; then bench=1; shift; fi
for v in "$@" "$@"; do
  if [ "$v" = "prod" ]; then unset SAS_LIB_PATH; else export SAS_LIB_PATH=variants/lib_$v.so; fi
  blend=$(timeout -k 10 200 python tools/stage_probe.py --cfg 3 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['stage_ms']['project'],4), round(d['stage_ms']['blend'],4), round(d['stage_ms']['total'],4))")
  fps=""
  if [ $bench = 1 ]; then fps=$(timeout -k 10 300 python bench.py --no-cpu-baseline --steps 300 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['value'],1))"); fi
  echo "$v project_ms,tile_ms,total_ms=$blend bench_fps=$fps"
done
