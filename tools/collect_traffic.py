"""Summarise rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE in separate runs) into
profiles/hbm_traffic.json.  FETCH_SIZE is doubled: on gfx950 it reports half the bytes of wide
coalesced reads (MI355X_MICROARCH.md, HBM section; confirmed on k_project: 240 MB read -> 120 MB)."""
import collections
import csv
import json
import re
import sys
from pathlib import Path

fetch_csv, write_csv, out = sys.argv[1], sys.argv[2], Path(sys.argv[3])


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            name = r["Kernel_Name"]
            short = re.split(r"[(<]", re.sub(r"\(anonymous namespace\)::|^void ", "", name))[0]
            acc[short].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fk, wk = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
kernels = {}
for k in sorted(set(fk) | set(wk)):
    f, w = fk.get(k, 0.0) * 1024, wk.get(k, 0.0) * 1024
    kernels[k] = {"fetch_size_bytes_raw": f, "write_size_bytes": w, "hbm_bytes_per_launch": 2 * f + w}
out.write_text(json.dumps({"workload": "BASELINE config 3 (1M Gaussians, 1920x1080), one frame per launch",
                           "correction": "hbm = 2*FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE counts 128-B requests as 64 B)",
                           "kernels": kernels}, indent=1))
print(json.dumps(kernels, indent=1))
