#!/bin/bash
# Gym-camera loops of prebuilt library variants on one GPU box: tools/ab_gym.sh name...   ("prod" = the in-tree library); run twice.
for v in "$@" "$@"; do
  if [ "$v" = "prod" ]; then unset SAS_LIB_PATH; else export SAS_LIB_PATH=variants/lib_$v.so; fi
  a=$(python3 tools/vec_env_probe.py 1 4 2>/dev/null | sed 's/.*per-env poses: \([0-9]*\) steps.*/\1/' | tr '\n' ' ')
  c=$(python3 examples/demo_synthetic_env.py 2>/dev/null | head -1 | cut -d, -f1)
  d=$(python3 tools/door_b_breakdown.py 2>/dev/null | tail -1 | sed 's/.*blend.: \([0-9.]*\),.*/\1/')
  echo "$v: gym steps/s (1 env, 4 envs) = $a | demo $c | isolated tile ms $d"
done
