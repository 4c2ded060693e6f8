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


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def up_to_date() -> bool:
    return LIB.exists() and all(LIB.stat().st_mtime >= d.stat().st_mtime for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and up_to_date():
        return LIB
    cmd = [hipcc_path(), "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-shared",
           "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function",
           # the SLP vectorizer pairs scalar f32 ops into v_pk_*_f32 (1.4x the issue cost of a scalar op on
           # gfx950, tools/microbench/pk_f32_rate.hip) plus the moves that assemble their operands: a net
           # loss in the compositing loop (+4.4 % frames/s at config 3 without it)
           "-fno-slp-vectorize",
           *os.environ.get("SAS_HIPCC_FLAGS", "").split(),   # experiments only
           "-x", "hip", *map(str, SOURCES), "-o", str(LIB)]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + res.stdout + res.stderr)
    if verbose:
        print(res.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
