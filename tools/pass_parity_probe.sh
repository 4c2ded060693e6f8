#!/bin/bash
# Do the W + K passes of bench.py alternate with the slot pair a pass starts on?  (round-4 verdict, weak 6)
# passes of 25 steps (W 5 + K 20: start slot pair alternates when the ring is not restarted) and of 24 (W 4 + K 20), ring restart on / off
for rr in 0 1; do for w in 5 4; do
  for rep in 1 2; do
    SAS_RING_RESTART=$rr timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup $w --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ring_restart=$rr warmup=$w value', round(d['value']), 'spread', round(d['pass_spread'],4), [round(p['value']) for p in d['passes']])"
  done
done; done
