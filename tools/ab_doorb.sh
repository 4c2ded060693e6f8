#!/bin/bash
# Door-B step and the 256x256 config of prebuilt library variants on one GPU box: tools/ab_doorb.sh name...  ("prod" = in-tree); run twice.
for v in "$@" "$@"; do
  if [ "$v" = "prod" ]; then unset SAS_LIB_PATH; else export SAS_LIB_PATH=variants/lib_$v.so; fi
  t1=$(python tools/stage_probe.py --cfg 1 2>/dev/null | tail -n 1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['stage_ms']['blend'],4))")
  db=$(python tools/door_b_breakdown.py 2>/dev/null | tail -n 2 | tr '\n' ' ' | sed -e 's/isolated frame stage ms://' | cut -c1-250)
  echo "$v tile_ms cfg1=$t1 | doorb $db"
done
