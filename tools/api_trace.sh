#!/bin/bash
# HIP runtime calls of the blocking Gym-camera step (rocprofv3 --hip-runtime-trace --kernel-trace; no counters):
# per call name the count per step and the median duration, and the host-side timeline of one step.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
rm -rf /tmp/at && mkdir -p /tmp/at
rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d /tmp/at -o t -- python3 $R/tools/vec_env_probe.py 1 > /tmp/at/log 2>&1
python3 - <<'P'
import csv, glob, collections
api = glob.glob("/tmp/at/**/*hip_api_trace.csv", recursive=True)[0]
ker = glob.glob("/tmp/at/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]) for r in csv.DictReader(open(api))]
rows.sort()
kr = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-28:]) for r in csv.DictReader(open(ker)))
# steps: from one hipEventSynchronize end to the next
syncs = [i for i, r in enumerate(rows) if r[2] == "hipEventSynchronize"]
mid = syncs[len(syncs) // 2]
a, b = syncs[len(syncs) // 2 - 1], mid
t0 = rows[a][1]
print("one step, host side (us since the previous step's wait returned):")
for s, e, n in rows[a + 1:b + 1]:
    print(f"  {(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f}  {n}")
print("kernels of that step:")
for s, e, n in kr:
    if rows[a][1] <= s <= rows[b][1]:
        print(f"  {(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f}  {n}")
dur = collections.defaultdict(list)
for s, e, n in rows[syncs[50]:syncs[-50]]:
    dur[n].append(e - s)
steps = len(syncs) - 100
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"{n:34s} {len(v) / steps:5.2f} per step, median {v[len(v) // 2] / 1e3:6.2f} us, total {sum(v) / steps / 1e3:6.2f} us per step")
P
