#!/bin/bash
# Instruction counts of every kernel of a config-3 frame (view pairs through bench.py): tools/pmc_frame_insts.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/pmc_frame
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d gpurun_out/pmc_frame -o p --output-format csv -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/pmc_frame.log 2>&1 || echo failed
python3 - <<'PY'
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_frame/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.split(r"[(<]", re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]))[0]
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(f"{k:32s} n={len(next(iter(d.values()))):4d} " + ' '.join(f"{c[8:] if c.startswith('SQ_INSTS_') else c}={sum(v)/len(v)/1e6:.2f}M" for c, v in sorted(d.items())))
PY
