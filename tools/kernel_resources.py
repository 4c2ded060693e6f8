"""Registers / spills / LDS / occupancy of every kernel of the library (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kernel_resources.py [filter] [-- extra hipcc flags]"""
import re, subprocess, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sim_a_splat_amd import build
args = sys.argv[1:]
flags = args[args.index("--") + 1:] if "--" in args else []
filt = args[0] if args and args[0] != "--" else ""
cmd = [build.hipcc_path(), build.OPT_LEVEL, "-std=c++17", f"--offload-arch={build.ARCH}", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
       "-fno-slp-vectorize", "-Rpass-analysis=kernel-resource-usage", *flags, "-x", "hip", *map(str, build.SOURCES), "-o", "/tmp/_res.so"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(anonymous namespace\)::|void ", "", cur).split("(")[0]
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z][\w ]*?)\s*(?:\[[\w/]+\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
print(f"{'kernel':58s} VGPR  SGPR  spillV spillS scratch  LDS   occ")
for k, r in rows.items():
    if filt in k:
        print(f"{k[:58]:58s} {r.get('VGPRs', 0):4d} {r.get('TotalSGPRs', r.get('SGPRs', 0)):5d} {r.get('VGPRs Spill', 0):6d} {r.get('SGPRs Spill', 0):6d} "
              f"{r.get('ScratchSize', 0):7d} {r.get('LDS Size', 0):6d} {r.get('Occupancy', r.get('Occupancy [waves/SIMD]', 0)):4d}")
