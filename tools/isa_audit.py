"""Static audit of one kernel's gfx950 assembly: instructions per loop, by issue class.

    python tools/isa_audit.py [kernel-substring] [-- extra hipcc flags]      (default: k_tile_lazyILb0ELb0ELb0E)

Compiles csrc/sas_tile.hip to assembly (-save-temps under /tmp), cuts the kernel out, and attributes every
instruction to the innermost loop the compiler's block comments name ("in Loop: Header=BBn_m Depth=d").
Classes: V = VALU (v_*), S = SALU / branches (s_* except waitcnt / nop / barrier), W = s_waitcnt + s_nop,
L = LDS (ds_*), G = global / buffer / flat, X = scratch (spill traffic), R = v_readlane / v_writelane
(SGPR spills parked in VGPR lanes).  Also lists DPP instructions whose source register was written by a VALU
instruction fewer than two instructions earlier (the gfx9 DPP read hazard the assembler cannot see inside
inline asm).  CPU only: hipcc cross-compiles.
"""
import re
import subprocess
import sys
import tempfile
from collections import OrderedDict, defaultdict
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sim_a_splat_amd import build  # noqa: E402


def classify(op):
    if op.startswith("scratch_"):
        return "X"
    if op in ("v_readlane_b32", "v_writelane_b32"):
        return "R"
    if op.startswith("v_"):
        return "V"
    if op.startswith("ds_"):
        return "L"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "G"
    if op in ("s_waitcnt", "s_nop"):
        return "W"
    if op.startswith("s_"):
        return "S"
    return "?"


def kernel_asm(filt, flags, src="sas_tile.hip"):
    tmp = Path(tempfile.mkdtemp(prefix="sas_isa_"))
    cmd = [build.hipcc_path(), build.OPT_LEVEL, "-std=c++17", f"--offload-arch={build.ARCH}", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
           "-fno-slp-vectorize", *flags, f"-I{build.PKG.parent / 'include'}", "-c", "-x", "hip", str(build.CSRC / src), "-o", "t.o", "-save-temps"]
    subprocess.run(cmd, cwd=tmp, capture_output=True, text=True, check=True)
    text = next(tmp.glob("*gfx950.s")).read_text().split("\n")
    start = next(i for i, l in enumerate(text) if re.match(r"^_Z\w*" + re.escape(filt) + r"\w*:", l))
    end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
    return text[start:end + 1]


def audit(lines):
    loops = OrderedDict()          # header -> (depth, parent)
    per = defaultdict(lambda: defaultdict(int))
    cur = "(outside loops)"
    hazards = []
    recent = []                    # (dst regs) of the last VALU instructions
    for ln, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):\s*;(.*)", l)
        if m or re.match(r"^; %bb\.\d+:", l):
            c = l
            mh = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", c)
            ml = re.search(r"=>\s*This .*Loop Header: Depth=(\d+)|Loop Header: Depth=(\d+)", c)
            if m and ("Loop Header" in c or re.search(r"Parent Loop|=>This", c)) and not mh:
                # a header block: named by its own label
                d = re.findall(r"Depth=(\d+)", c)
                cur = m.group(1)[2:]
                loops.setdefault(cur, int(d[-1]) if d else 0)
            elif mh:
                cur = mh.group(1)
                loops.setdefault(cur, int(mh.group(2)))
            elif "Loop" not in c:
                cur = "(outside loops)"
            continue
        s = l.strip()
        if not s or s.startswith((";", ".")):
            # continuation comment lines of a header ("Parent Loop ... / => This Inner Loop Header: Depth=3")
            mm = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", s)
            if mm:
                loops[cur] = int(mm.group(1))
            continue
        op = s.split()[0]
        k = classify(op)
        per[cur][k] += 1
        if "_dpp" in op or " row_newbcast" in s or " quad_perm" in s or " row_sh" in s:
            regs = re.findall(r"\bv(\d+)\b", s.split(None, 1)[1])
            src0 = regs[1] if len(regs) > 1 else None
            for age, dsts in enumerate(reversed(recent[-2:])):
                if src0 in dsts:
                    hazards.append((ln, s, age))
        if k == "V":
            d = re.findall(r"\bv(\d+)\b", s.split(None, 1)[1].split(",")[0])
            rng = re.findall(r"v\[(\d+):(\d+)\]", s.split(None, 1)[1].split(",")[0])
            dst = set(d)
            for a, b in rng:
                dst |= {str(x) for x in range(int(a), int(b) + 1)}
            recent.append(dst)
        elif k in ("S", "W", "L", "G", "X", "R"):
            recent.append(set() if op != "s_nop" else set())
            if op == "s_nop":   # s_nop N = N + 1 wait states
                n = int(s.split()[1])
                recent.extend([set()] * n)
    return loops, per, hazards


if __name__ == "__main__":
    args = sys.argv[1:]
    flags = args[args.index("--") + 1:] if "--" in args else []
    filt = args[0] if args and args[0] != "--" else "k_tile_lazyILb0ELb0ELb0E"
    lines = kernel_asm(filt, flags)
    loops, per, hazards = audit(lines)
    tot = defaultdict(int)
    print(f"{'loop (header block)':22s} depth     V     S     W     L     G     X     R")
    for name, c in per.items():
        print(f"{name:22s} {loops.get(name, 0):5d} " + " ".join(f"{c.get(k, 0):5d}" for k in "VSWLGXR"))
        for k, v in c.items():
            tot[k] += v
    print(f"{'total':22s}       " + " ".join(f"{tot.get(k, 0):5d}" for k in "VSWLGXR"))
    print("DPP sources written by a VALU instruction < 2 instructions earlier:", len(hazards))
    for ln, s, age in hazards[:20]:
        print("  line", ln, s, "| wait states:", age)
