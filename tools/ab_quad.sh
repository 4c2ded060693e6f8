#!/bin/bash
# Quad layout of the tile kernel (SAS_QUAD=1: 8-pixel tiles, one workgroup per 8x8 quadrant) against the ordinary one (SAS_QUAD=0) on
# one GPU box: isolated tile-kernel ms of configs 1-3, the Door-B step, frames/s of configs 2/3.  Run twice.
for q in 0 1 0 1; do
  export SAS_QUAD=$q
  t1=$(python tools/stage_probe.py --cfg 1 2>/dev/null | tail -n 1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['stage_ms']['blend'],4))")
  t2=$(python tools/stage_probe.py --cfg 2 2>/dev/null | tail -n 1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['stage_ms']['blend'],4))")
  t3=$(python tools/stage_probe.py --cfg 3 2>/dev/null | tail -n 1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['stage_ms']['blend'],4))")
  db=$(python tools/door_b_breakdown.py 2>/dev/null | tail -n 2 | tr '\n' ' ' | sed -e 's/isolated frame stage ms://' | cut -c1-260)
  f=$(python tools/config_fps.py 1 2 2>/dev/null | cut -d">" -f2 | tr "\n" " ")
  echo "SAS_QUAD=$q tile_ms cfg1=$t1 cfg2=$t2 cfg3=$t3 | fps cfg1,2 $f | doorb $db"
done
