#!/bin/bash
# kernel timeline of the blocking Gym-camera step (2 x 240x320 uint8 to the host, poses per step) and of the blocking
# config-3 frame (door_a pattern): durations, gaps between dependent kernels, host turnaround between steps
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
rm -rf /tmp/gg && mkdir -p /tmp/gg
rocprofv3 --kernel-trace --output-format csv -d /tmp/gg/a -o t -- python3 $R/tools/vec_env_probe.py 1 > /tmp/gg/a.log 2>&1
f=$(find /tmp/gg/a -name "*kernel_trace.csv" | head -1)
echo "== Gym-camera step (tools/vec_env_probe.py 1)"; python3 $R/tools/step_gaps.py $f
rocprofv3 --kernel-trace --output-format csv -d /tmp/gg/b -o t -- python3 $R/tools/stage_probe.py --cfg 3 --frames 200 --plain > /tmp/gg/b.log 2>&1
f=$(find /tmp/gg/b -name "*kernel_trace.csv" | head -1)
echo "== blocking config-3 frame (tools/stage_probe.py --cfg 3 --plain)"; python3 $R/tools/step_gaps.py $f
tail -3 /tmp/gg/b.log
