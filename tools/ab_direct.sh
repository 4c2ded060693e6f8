#!/bin/bash
# Single-pass (SAS_DIRECT=1, the product path) against two-pass binning (SAS_DIRECT=0) at the driver's own command and at 300
# steps, alternating, on one GPU box:  tools/ab_direct.sh [repeats]
n=${1:-3}
for i in $(seq $n); do
  for d in 0 1; do
    v=$(SAS_DIRECT=$d python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['cold_start']['value']), len(d['passes']))")
    w=$(SAS_DIRECT=$d python bench.py --steps 300 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']))")
    echo "SAS_DIRECT=$d driver-cmd value,cold,passes: $v | 300 steps: $w"
  done
done
