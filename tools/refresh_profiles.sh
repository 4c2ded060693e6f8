#!/bin/bash
# Re-collect everything under profiles/ on a GPU box (run through gpurun from the repo root):
#   gpurun -- tools/refresh_profiles.sh r03
# then locally:  tools/refresh_profiles.sh --install r03
# (gpurun_out/ is the only directory that comes back from the box.)
set -u
if [ "${1:-}" = "--install" ]; then
  tag=$2; src=gpurun_out/prof_$tag
  cp $src/bench.json profiles/${tag}_bench.json
  for i in 1 2 3; do cp $src/bench_driver_cmd_$i.json profiles/${tag}_bench_driver_cmd_$i.json; done
  cp $src/bench_under_rocprofv3.json profiles/${tag}_bench_under_rocprofv3.json
  cp $(find $src/bench_stats -name "*kernel_stats.csv" | head -n 1) profiles/${tag}_bench_kernel_stats_rocprofv3.csv
  cp $(find $src/driver_stats -name "*kernel_stats.csv" | head -n 1) profiles/${tag}_bench_driver_cmd_kernel_stats_rocprofv3.csv
  cp $(find $src/single_stats -name "*kernel_stats.csv" | head -n 1) profiles/${tag}_single_frame_kernel_stats_rocprofv3.csv
  cp $(find $src/pmc_fetch -name "*counter_collection.csv" | head -n 1) profiles/${tag}_pmc_fetch_size.csv
  cp $(find $src/pmc_write -name "*counter_collection.csv" | head -n 1) profiles/${tag}_pmc_write_size.csv
  cp $src/tile_sq_counters.txt profiles/${tag}_tile_sq_counters.txt
  cp $src/blend_stats.txt profiles/${tag}_blend_stats.txt
  cp $src/microbench.txt profiles/${tag}_issue_rate_microbench.txt
  cp $src/config4.json profiles/${tag}_config4.json
  cp $src/config5.json profiles/${tag}_config5.json
  cp $src/config_fps.txt profiles/${tag}_config_fps.txt
  cp $src/ramp.txt profiles/${tag}_clock_ramp.txt
  cp $src/timeline_driver_cmd.txt profiles/${tag}_timeline_driver_cmd.txt
  cp $src/env_steps.txt profiles/${tag}_env_steps.txt
  [ -f $src/api_trace.txt ] && cp $src/api_trace.txt profiles/${tag}_gym_step_api_trace.txt
  [ -f $src/issue_cost.txt ] && cp $src/issue_cost.txt profiles/${tag}_issue_cost_microbench.txt
  [ -f $src/slot_time.txt ] && cp $src/slot_time.txt profiles/${tag}_tile_slot_time_partition.txt
  [ -f $src/proj_laps.txt ] && cp $src/proj_laps.txt profiles/${tag}_projection_workgroup_laps.txt
  [ -f $src/project_sq_counters.txt ] && cp $src/project_sq_counters.txt profiles/${tag}_project_sq_counters.txt
  python tools/kernel_resources.py > profiles/${tag}_kernel_resources.txt 2>&1
  [ -f $src/bounds_quad_tests.log ] && cp $src/bounds_quad_tests.log profiles/${tag}_bounds_build_quad_forced_gpu_tests.log
  [ -f $src/gpu_tests.log ] && cp $src/gpu_tests.log profiles/${tag}_gpu_tests.log
  [ -f $src/bounds_tests.log ] && cp $src/bounds_tests.log profiles/${tag}_bounds_build_gpu_tests.log
  python tools/collect_traffic.py profiles/${tag}_pmc_fetch_size.csv profiles/${tag}_pmc_write_size.csv profiles/hbm_traffic.json > /dev/null
  python tools/collect_insts.py $(find $src/pmc_insts_bench -name "*counter_collection.csv" | head -n 1) $src/blend_stats.txt profiles/tile_insts.json
  ls -la profiles
  exit 0
fi
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/prof_$tag
# PART=a: benches and profiles; PART=b: counters, probes and the GPU suites (two gpurun calls: one call's limit is 20 minutes)
part=${PART:-ab}
if [ "$part" != "b" ]; then
rm -rf $out; mkdir -p $out
# the driver's own command, three times, then a long run
for i in 1 2 3; do
  timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd_$i.json 2> $out/bench_driver_cmd_$i.err || exit 1
done
timeout -k 10 500 python3 bench.py --steps 300 > $out/bench.json 2> $out/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/bench_stats -o b --output-format csv -- python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extras > $out/bench_under_rocprofv3.json 2> $out/bench_prof.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/driver_stats -o d --output-format csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $out/bench_driver_cmd_under_rocprofv3.json 2> $out/driver_prof.err || exit 1
python3 tools/timeline.py $(find $out/driver_stats -name "*kernel_trace.csv" | head -n 1) > $out/timeline_driver_cmd.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/single_stats -o s --output-format csv -- python3 tools/stage_probe.py --cfg 3 --frames 30 > $out/single.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch -o f --output-format csv -- python3 tools/stage_probe.py --cfg 3 --frames 5 > $out/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write -o w --output-format csv -- python3 tools/stage_probe.py --cfg 3 --frames 5 > $out/pmc_write.log 2>&1 || exit 1
# instruction counts of the tile kernel under the bench command itself (both views of a step)
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-include-regex "k_tile_lazy" -d $out/pmc_insts_bench -o p --output-format csv -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras > $out/pmc_insts_bench.log 2>&1 || exit 1
fi
if [ "$part" = "a" ]; then tail -c 400 $out/bench.json; exit 0; fi
mkdir -p $out
# SQ counter groups of the tile kernel on isolated frames
bash tools/pmc_tile.sh > $out/tile_sq_counters.txt 2>&1
SAS_LIB_PATH=variants/lib_stats.so python3 tools/blend_stats.py 3 > $out/blend_stats.txt 2>&1
# the other BASELINE configs on this GPU, the Gym loops, the clock ramp
timeout -k 10 300 python3 bench.py --config 4 --steps 100 > $out/config4.json 2> $out/config4.err || exit 1
timeout -k 10 300 python3 bench.py --config 5 --steps 40 > $out/config5.json 2> $out/config5.err || exit 1
timeout -k 10 300 python3 tools/config_fps.py 1 2 3 5 > $out/config_fps.txt 2>&1
timeout -k 10 200 python3 tools/ramp_probe.py > $out/ramp.txt 2>&1
{ timeout -k 10 200 python3 examples/demo_synthetic_env.py; timeout -k 10 200 python3 tools/door_b_breakdown.py; timeout -k 10 300 python3 tools/vec_env_probe.py 1 4 16; } > $out/env_steps.txt 2>&1
# the issue-rate microbenchmarks are built here from their sources (no binaries in the tree)
for mb in pk_f32_rate clock_probe issue_probe; do
  /opt/rocm/bin/hipcc -w -O3 --offload-arch=gfx950 tools/microbench/$mb.hip -o /tmp/$mb || exit 1
done
{ /tmp/pk_f32_rate | head -9; /tmp/clock_probe; } > $out/microbench.txt 2>&1
/tmp/issue_probe > $out/issue_cost.txt 2>&1
# where a tile workgroup's time goes (exclusive laps of thread 0: the -DSAS_TUNE_WGTIME build under variants/)
SAS_LIB_PATH=variants/lib_wgtime.so timeout -k 10 300 python3 tools/wg_time.py 3 > $out/slot_time.txt 2>&1
# ... and a projection workgroup's (geometry role: exclusive laps; both roles: when they run inside the launch; -DSAS_TUNE_PTIME build)
SAS_LIB_PATH=variants/lib_ptime.so timeout -k 10 300 python3 tools/proj_time.py 3 > $out/proj_laps.txt 2>&1
bash tools/pmc_tile.sh k_project > $out/project_sq_counters.txt 2>&1
# the GPU suite: product library, bounds-checked build, bounds-checked build with the quad layout forced
timeout -k 10 600 python3 -m pytest tests -m gpu -q -s > $out/gpu_tests.log 2>&1
SAS_LIB_PATH=variants/lib_bounds.so timeout -k 10 900 python3 -m pytest tests -m gpu -q > $out/bounds_tests.log 2>&1
SAS_QUAD=1 SAS_LIB_PATH=variants/lib_bounds.so timeout -k 10 900 python3 -m pytest tests -m gpu -q > $out/bounds_quad_tests.log 2>&1
[ -f $out/bench.json ] && tail -c 700 $out/bench.json; true
