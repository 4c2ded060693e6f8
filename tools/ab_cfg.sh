#!/bin/bash
# Per-config A/B of prebuilt library variants on one GPU box: tools/ab_cfg.sh name...  ("prod" = in-tree)
# prints isolated tile-kernel ms for configs 2 and 3, the Door-B step (tools/door_b_breakdown.py), frames/s of configs 2/3.
for v in "$@" "$@"; do
  if [ "$v" = "prod" ]; then unset SAS_LIB_PATH; else export SAS_LIB_PATH=variants/lib_$v.so; fi
  t2=$(python tools/stage_probe.py --cfg 2 2>/dev/null | tail -n 1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['stage_ms']['blend'],4))")
  t3=$(python tools/stage_probe.py --cfg 3 2>/dev/null | tail -n 1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['stage_ms']['blend'],4))")
  db=$(python tools/door_b_breakdown.py 2>/dev/null | tail -n 2 | tr '\n' ' ' | sed -e 's/isolated frame stage ms://' | cut -c1-200)
  f=$(python tools/config_fps.py 2 3 2>/dev/null | cut -d">" -f2 | tr "\n" " ")
  echo "$v tile_ms cfg2=$t2 cfg3=$t3 | fps $f | doorb $db"
done
