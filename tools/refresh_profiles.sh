#!/bin/bash
# Re-collect everything under profiles/ on a GPU box (run through gpurun from the repo root):
#   gpurun -- tools/refresh_profiles.sh r02
# then locally:  tools/refresh_profiles.sh --install r02
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
  cp $src/tile_sq_counters.txt profiles/${tag}_tile_sq_counters.txt
  cp $src/blend_stats.txt profiles/${tag}_blend_stats.txt
  cp $src/microbench.txt profiles/${tag}_issue_rate_microbench.txt
  python tools/collect_traffic.py profiles/${tag}_pmc_fetch_size.csv profiles/${tag}_pmc_write_size.csv profiles/hbm_traffic.json > /dev/null
  python tools/collect_insts.py $(find $src/pmc_insts_bench -name "*counter_collection.csv" | head -n 1) $src/blend_stats.txt profiles/tile_insts.json
  ls -la profiles
  exit 0
fi
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
timeout -k 10 500 python bench.py --steps 300 > $out/bench.json 2> $out/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/bench_stats -o b --output-format csv -- python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extras > $out/bench_under_rocprofv3.json 2> $out/bench_prof.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/single_stats -o s --output-format csv -- python3 tools/stage_probe.py --cfg 3 --frames 30 > $out/single.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch -o f --output-format csv -- python3 tools/stage_probe.py --cfg 3 --frames 5 > $out/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write -o w --output-format csv -- python3 tools/stage_probe.py --cfg 3 --frames 5 > $out/pmc_write.log 2>&1 || exit 1
# instruction counts of the tile kernel under the bench command itself (both views of a step)
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-include-regex "k_tile_lazy" -d $out/pmc_insts_bench -o p --output-format csv -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras > $out/pmc_insts_bench.log 2>&1 || exit 1
# SQ counter groups of the tile kernel on isolated frames
bash tools/pmc_tile.sh > $out/tile_sq_counters.txt 2>&1
SAS_LIB_PATH=variants/lib_stats.so python tools/blend_stats.py 3 > $out/blend_stats.txt 2>&1
# the issue-rate microbenchmarks are built here from their sources (no binaries in the tree)
for mb in pk_f32_rate clock_probe; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/microbench/$mb.hip -o /tmp/$mb || exit 1
done
{ /tmp/pk_f32_rate | head -9; /tmp/clock_probe; } > $out/microbench.txt 2>&1
tail -c 700 $out/bench.json
