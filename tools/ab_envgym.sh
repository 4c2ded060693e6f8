#!/bin/bash
# Gym loops and the blocking config-3 frame under environment settings, each run twice:  tools/ab_envgym.sh "HIP_FORCE_DEV_KERNARG=1" ""
for e in "$@" "$@"; do
  a=$(env $e python3 tools/vec_env_probe.py 1 4 2>/dev/null | sed 's/.*per-env poses: \([0-9]*\) steps.*/\1/' | tr '\n' ' ')
  c=$(env $e python3 examples/demo_synthetic_env.py 2>/dev/null | head -1 | cut -d, -f1)
  b=$(env $e python3 tools/stage_probe.py --cfg 3 --frames 300 --plain 2>/dev/null | tail -1)
  echo "[$e] gym steps/s (1 env, 4 envs) = $a | demo $c | $b"
done
