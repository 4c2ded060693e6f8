cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*" | sort -u | tr '\n' ' '; echo
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  rm -rf gpurun_out/pmc_ic
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-include-regex "k_tile_lazy|k_project" -d gpurun_out/pmc_ic -o p --output-format csv -- python3 tools/stage_probe.py --cfg 3 --frames 4 > gpurun_out/pmc_ic.log 2>&1 || { echo "group failed: $grp"; tail -3 gpurun_out/pmc_ic.log; continue; }
  python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_ic/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r['Kernel_Name'][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()):
        print(f"{k[0]:42s} {k[1]:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
done
