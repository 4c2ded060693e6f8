#!/bin/bash
# SQ counters of one kernel on isolated config-3 frames (one rocprofv3 --pmc pass per counter group):
#   tools/pmc_tile.sh [kernel regex, default k_tile_lazy]
# Output: gpurun_out/pmc_tile/<group>/..._counter_collection.csv
kern=${1:-k_tile_lazy}
rm -rf gpurun_out/pmc_tile
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-include-regex "$kern" -d gpurun_out/pmc_tile/g$i -o p --output-format csv -- python3 tools/stage_probe.py --cfg 3 --frames 4 > gpurun_out/pmc_tile_g$i.log 2>&1 || echo "group $i failed"
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_tile/g*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
        print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
