"""Per-frame kernel timeline from a rocprofv3 --kernel-trace CSV: overlap and idle time.

    rocprofv3 --kernel-trace -d gpurun_out/trace -o t --output-format csv -- python3 bench.py --steps 60 --warmup 20 --no-cpu-baseline
    python tools/timeline.py gpurun_out/trace/.../t_kernel_trace.csv
"""
import csv
import re
import sys
from collections import defaultdict


def main(path, skip=200):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.split(r"[(<]", name)[0][-40:]))
    rows.sort()
    rows = rows[skip:-40] if len(rows) > skip + 100 else rows
    t0, t1 = rows[0][0], max(e for _, e, _ in rows)
    # busy time (union of intervals) and per-kernel totals
    per = defaultdict(lambda: [0, 0])
    ev = []
    for s, e, n in rows:
        per[n][0] += e - s
        per[n][1] += 1
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, last, busy, two = 0, ev[0][0], 0, 0
    for t, d in ev:
        if depth > 0:
            busy += t - last
        if depth > 1:
            two += t - last
        depth += d
        last = t
    span = t1 - t0
    tiles = [r for r in rows if "k_tile_lazy" in r[2]]
    print(f"span {span/1e3:.1f} us, {len(tiles)} frames -> {span/1e3/max(len(tiles),1):.1f} us/frame")
    print(f"GPU busy (>=1 kernel) {busy/span:.3f}, >=2 kernels {two/span:.3f}")
    for n, (tot, cnt) in sorted(per.items(), key=lambda kv: -kv[1][0]):
        print(f"  {n:42s} n={cnt:5d} mean {tot/cnt/1e3:8.1f} us  sum/span {tot/span:.3f}")
    # gaps between consecutive tile kernels
    gaps = [tiles[i + 1][0] - tiles[i][1] for i in range(len(tiles) - 1)]
    if gaps:
        gaps.sort()
        print(f"tile-kernel gap (next start - prev end): median {gaps[len(gaps)//2]/1e3:.1f} us, min {gaps[0]/1e3:.1f}, max {gaps[-1]/1e3:.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
