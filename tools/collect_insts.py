"""profiles/tile_insts.json: instruction counts per launch of k_tile_lazy under the bench command
(rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM) + the -DSAS_TUNE_STATS counters of
tools/blend_stats.py (composited pixel-splat pairs).  bench.py reads it for the `roofline_issue` entry."""
import collections
import csv
import json
import re
import sys
from pathlib import Path

pmc_csv, stats_txt, out = sys.argv[1], sys.argv[2], Path(sys.argv[3])
acc = collections.defaultdict(list)
for r in csv.DictReader(open(pmc_csv)):
    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
mean = {k: sum(v) / len(v) for k, v in acc.items()}
txt = Path(stats_txt).read_text()
m = re.search(r"composited lanes/iter=([0-9.]+)", txt)          # blend_stats prints the total there when iterations with candidates are not counted
comp = re.search(r"composited=(\d+)", txt)
composited = int(comp.group(1)) if comp else (int(float(m.group(1))) if m else 0)
staged = re.search(r"staged=(\d+)", txt)
d = {"workload": "BASELINE config 3, bench.py (two views per step): mean over the k_tile_lazy launches of the profiled run",
     "valu_insts_per_launch": mean.get("SQ_INSTS_VALU", 0.0), "salu_insts_per_launch": mean.get("SQ_INSTS_SALU", 0.0),
     "lds_insts_per_launch": mean.get("SQ_INSTS_LDS", 0.0), "smem_insts_per_launch": mean.get("SQ_INSTS_SMEM", 0.0),
     "launches": len(acc.get("SQ_INSTS_VALU", [])),
     "composited_pixel_splats": composited, "staged_entries": int(staged.group(1)) if staged else 0,
     "valu_per_composited_pixel_splat": 28,
     "valu_per_composited_pixel_splat_note": "sigma 7 + clamp 1 + exp 9 + alpha 2 + skip test / weight 3 + T 1 + half a stop test + accumulate 4, rounded up (the loop's instructions for one pixel-splat pair on a trip where no pixel terminates, at full lane use)"}
out.write_text(json.dumps(d, indent=1))
print(json.dumps(d, indent=1))
