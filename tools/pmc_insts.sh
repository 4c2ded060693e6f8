#!/bin/bash
# Instruction counts of the tile kernel for library variants: tools/pmc_insts.sh name...  ("prod" = in-tree)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for v in "$@"; do
  if [ "$v" = "prod" ]; then unset SAS_LIB_PATH; else export SAS_LIB_PATH=$GRAFT_REPO_ROOT/variants/lib_$v.so; fi
  rm -rf gpurun_out/pmc_insts/$v
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-include-regex "k_tile_lazy" -d gpurun_out/pmc_insts/$v -o p --output-format csv -- python3 tools/stage_probe.py --cfg 3 --frames 4 > gpurun_out/pmc_insts_$v.log 2>&1 || echo "$v failed"
  python3 - $v <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(list)
for f in glob.glob(f'gpurun_out/pmc_insts/{sys.argv[1]}/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
print(sys.argv[1], ' '.join(f"{k}={sum(v)/len(v)/1e6:.2f}M" for k, v in sorted(acc.items())))
PY
done
