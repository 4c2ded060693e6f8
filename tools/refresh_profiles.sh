#!/bin/bash
# Re-collect everything under profiles/ on a GPU box (run through gpurun from the repo root):
#   gpurun -- tools/refresh_profiles.sh r01
# then locally:  tools/refresh_profiles.sh --install r01
# (gpurun_out/ is the only directory that comes back from the box.)
set -u
if [ "${1:-}" = "--install" ]; then
  tag=$2; src=gpurun_out/prof_$tag
  cp $src/bench.json profiles/${tag}_bench.json
  cp $src/bench_under_rocprofv3.json profiles/${tag}_bench_under_rocprofv3.json
  cp $(find $src/bench_stats -name "*kernel_stats.csv" | head -n 1) profiles/${tag}_bench_kernel_stats_rocprofv3.csv
  cp $(find $src/single_stats -name "*kernel_stats.csv" | head -n 1) profiles/${tag}_single_frame_kernel_stats_rocprofv3.csv
  cp $(find $src/pmc_fetch -name "*counter_collection.csv" | head -n 1) profiles/${tag}_pmc_fetch_size.csv
  cp $(find $src/pmc_write -name "*counter_collection.csv" | head -n 1) profiles/${tag}_pmc_write_size.csv
  python tools/collect_traffic.py profiles/${tag}_pmc_fetch_size.csv profiles/${tag}_pmc_write_size.csv profiles/hbm_traffic.json > /dev/null
  ls -la profiles
  exit 0
fi
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/prof_$tag
mkdir -p $out
timeout -k 10 500 python bench.py --steps 300 > $out/bench.json 2> $out/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/bench_stats -o b --output-format csv -- python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline > $out/bench_under_rocprofv3.json 2> $out/bench_prof.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/single_stats -o s --output-format csv -- python3 tools/stage_probe.py --cfg 3 --frames 30 > $out/single.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch -o f --output-format csv -- python3 tools/stage_probe.py --cfg 3 --frames 5 > $out/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write -o w --output-format csv -- python3 tools/stage_probe.py --cfg 3 --frames 5 > $out/pmc_write.log 2>&1 || exit 1
tail -c 600 $out/bench.json
