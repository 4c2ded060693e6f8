cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/gg && mkdir -p /tmp/gg
rocprofv3 --kernel-trace --output-format csv -d /tmp/gg/a -o t -- python3 $R/examples/demo_synthetic_env.py --steps 800 > /tmp/gg/a.log 2>&1
f=$(find /tmp/gg/a -name "*kernel_trace.csv" | head -1)
python3 $R/tools/step_gaps.py $f
grep "env steps" /tmp/gg/a.log
