#!/bin/bash
# kernel_ms of the timed tile launches against the sampling stride (bench.py --time-every)
mkdir -p gpurun_out
for te in 4 1 3 4; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --time-every $te --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('time_every', $te, 'value', round(d['value']), 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'n', d['roofline']['kernel_launches_timed'], 'passes', len(d['passes']))"
done
python3 bench.py --gpus 1 --steps 300 --warmup 20 --time-every 4 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('300 steps value', round(d['value']), 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'n', d['roofline']['kernel_launches_timed'])"
