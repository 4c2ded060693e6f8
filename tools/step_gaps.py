"""Kernel timeline of blocking steps from a rocprofv3 --kernel-trace CSV: per step (a step starts with its pose upload
or, for scenes without splat groups, its projection, and ends with its last kernel) the kernel durations and the idle
gaps between them.  python tools/step_gaps.py t_kernel_trace.csv"""
import csv, re, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.split(r"[(<]", name)[0]))
rows.sort()
steps, cur = [], []
for s, e, n in rows:
    first = n.startswith("k_pose_upload") or (n.startswith("k_project") and not any(m.startswith("k_pose_upload") for _, _, m in cur))
    if first and cur:
        steps.append(cur); cur = []
    cur.append((s, e, n))
steps = [st for st in steps if any(n.startswith("k_tile_lazy") for _, _, n in st)][20:-5]
import collections
dur, gap, span, between = collections.defaultdict(list), collections.defaultdict(list), [], []
prev_end = None
for st in steps:
    span.append(st[-1][1] - st[0][0])
    if prev_end: between.append(st[0][0] - prev_end)
    prev_end = st[-1][1]
    for i, (s, e, n) in enumerate(st):
        dur[n].append(e - s)
        if i: gap[f"{st[i-1][2]} -> {n}"].append(s - st[i-1][1])
med = lambda v: sorted(v)[len(v) // 2] / 1e3
print(f"{len(steps)} steps; first kernel start -> last kernel end: {med(span):.1f} us; last kernel end -> next step's first kernel: {med(between):.1f} us")
for n, v in dur.items(): print(f"  {n:28s} {med(v):7.1f} us")
for n, v in gap.items(): print(f"  gap {n:50s} {med(v):6.1f} us")
