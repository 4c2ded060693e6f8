#!/bin/bash
# Marginal costs inside the single-pass projection (timing only: the frames are wrong): tools/ablate_project.sh
# variants/lib_pabl<mask>.so built with -DSAS_TUNE_PABL=<mask> (1 no count atomics, 2 no key stores, 4 no LDS atomics, 8 no record stores)
for r in 1 2; do
for v in prod pabl1 pabl2 pabl4 pabl8 pabl3 pabl7; do
  if [ "$v" = "prod" ]; then unset SAS_LIB_PATH; else export SAS_LIB_PATH=variants/lib_$v.so; fi
  st=$(timeout -k 10 200 python tools/stage_probe.py --cfg 3 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print(round(s['project'],4))")
  echo "$v project_ms=$st"
done; done
