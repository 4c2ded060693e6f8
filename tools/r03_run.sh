#!/bin/bash
# gpu tests, then the driver's bench command three times and a long run
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r03; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1; rc=$?
tail -12 $out/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2 3; do
  timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_$i.json 2> $out/bench_driver_$i.err || exit 1
  python3 -c "import json;d=json.load(open('$out/bench_driver_$i.json'));print('driver cmd', round(d['value']),d['ms_per_step'],'doorA',round(d['door_a_sync']['value']),'single',round(d['single_view_async']['value']), d['roofline']['isolated_frame_stage_ms'])"
done
timeout -k 10 300 python3 bench.py --gpus 1 --steps 300 --warmup 20 --no-cpu-baseline > $out/bench_300.json 2> $out/bench_300.err || exit 1
python3 -c "import json;d=json.load(open('$out/bench_300.json'));print('300 steps', round(d['value']),d['ms_per_step'])"
