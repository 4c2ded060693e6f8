#!/bin/bash
# Isolated projection / tile-kernel time (stage_probe, config 3) of library variants x SAS_PROJ_MIX settings on one GPU box:
#   tools/ab_proj.sh [--bench] name[:mix]...     ("prod" = the in-tree library; mix = geometry blocks per 8 leading blocks)
# the list is run twice.
bench=0; if [ "$1" = "--bench" ]; then bench=1; shift; fi
for spec in "$@" "$@"; do
  v=${spec%%:*}; mix=""; if [[ "$spec" == *:* ]]; then mix=${spec#*:}; fi
  if [ "$v" = "prod" ]; then unset SAS_LIB_PATH; else export SAS_LIB_PATH=variants/lib_$v.so; fi
  if [ -n "$mix" ]; then export SAS_PROJ_MIX=$mix; else unset SAS_PROJ_MIX; fi
  st=$(timeout -k 10 200 python tools/stage_probe.py --cfg 3 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['stage_ms']['project'],4), round(d['stage_ms']['blend'],4), round(d['stage_ms']['total'],4))")
  fps=""
  if [ $bench = 1 ]; then fps=$(timeout -k 10 300 python bench.py --no-cpu-baseline --steps 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d.get('door_a_sync',{}).get('value',0),1) if isinstance(d.get('door_a_sync'),dict) else d.get('door_a_sync'))"); fi
  echo "$spec project_ms,tile_ms,total_ms=$st bench_fps,door_a_sync=$fps"
done
