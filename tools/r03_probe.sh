#!/bin/bash
# Round-3 opening measurements (one gpurun call): the driver's bench command, its kernel timeline, empty-mask share.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r03_probe
rm -rf $out; mkdir -p $out
for i in 1 2 3; do
  timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_$i.json 2> $out/bench_driver_$i.err || exit 1
done
timeout -k 10 300 python3 bench.py --gpus 1 --steps 300 --warmup 20 --no-cpu-baseline > $out/bench_300.json 2> $out/bench_300.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/trace -o t --output-format csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $out/bench_trace.json 2> $out/bench_trace.err || exit 1
python3 tools/timeline.py $(find $out/trace -name "*kernel_trace.csv" | head -n 1) > $out/timeline.txt 2>&1
SAS_LIB_PATH=variants/lib_stats.so timeout -k 10 300 python3 tools/blend_stats.py 3 2 > $out/blend_stats.txt 2>&1
cat $out/blend_stats.txt
for i in 1 2 3; do python3 -c "import json;d=json.load(open('$out/bench_driver_$i.json'));print(d['value'],d['ms_per_step'],d['door_a_sync']['value'],d['single_view_async']['value'])"; done
python3 -c "import json;d=json.load(open('$out/bench_300.json'));print(d['value'],d['ms_per_step'])"
