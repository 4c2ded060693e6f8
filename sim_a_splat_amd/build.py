"""Build libsas_hip.so (the C-ABI shared library) in-tree with hipcc for gfx950.

    python -m sim_a_splat_amd.build [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the arithmetic contract
(DESIGN.md): fused multiply-adds exist only where the source writes fmaf.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libsas_hip.so"
SOURCES = [CSRC / "sas_kernels.hip", CSRC / "sas_tile.hip", CSRC / "sas_api.cpp"]
DEPS = SOURCES + [CSRC / "sas_internal.h", CSRC / "sas_device.h", PKG.parent / "include" / "sim_a_splat_amd.h"]
ARCH = "gfx950"
OPT_LEVEL = "-O2"


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def up_to_date() -> bool:
    return LIB.exists() and all(LIB.stat().st_mtime >= d.stat().st_mtime for d in DEPS)


def build(force: bool = False, verbose: bool = False, out: Path = None, extra_flags=()) -> Path:
    """Default: the product library in-tree.  ``out`` + ``extra_flags`` build a variant elsewhere
    (``python -m sim_a_splat_amd.build --variant stats -DSAS_TUNE_STATS`` -> variants/lib_stats.so)."""
    if out is None and not force and up_to_date():
        return LIB
    LIB_OUT = Path(out) if out is not None else LIB
    # -O2: measured against -O3 on one box, six back-to-back pairs (profiles/r04_ab_compiler_flags.txt): the tile kernel 1 % shorter,
    # the pair bench +1.4 %, two VGPRs fewer spilled; the frames are the same bits either way (no flag here lets the compiler
    # re-associate or contract floating point)
    cmd = [hipcc_path(), OPT_LEVEL, "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-shared",
           "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function",
           # the SLP vectorizer pairs scalar f32 ops into v_pk_*_f32 (1.4x the issue cost of a scalar op on
           # gfx950, tools/microbench/pk_f32_rate.hip) plus the moves that assemble their operands: a net
           # loss in the compositing loop (+4.4 % frames/s at config 3 without it)
           "-fno-slp-vectorize",
           *os.environ.get("SAS_HIPCC_FLAGS", "").split(), *extra_flags,   # experiments only
           "-x", "hip", *map(str, SOURCES), "-o", str(LIB_OUT)]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + res.stdout + res.stderr)
    if verbose:
        print(res.stderr)
    return LIB_OUT


if __name__ == "__main__":
    args = sys.argv[1:]
    if "--variant" in args:
        k = args.index("--variant")
        name = args[k + 1]
        flags = [a for a in args[k + 2:] if a != "-v"]
        vdir = PKG.parent / "variants"
        vdir.mkdir(exist_ok=True)
        print(build(force=True, verbose="-v" in args, out=vdir / f"lib_{name}.so", extra_flags=flags))
    else:
        print(build(force="--force" in args, verbose="-v" in args))
