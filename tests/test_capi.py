"""CPU: the C-ABI library loads and exports every symbol include/sim_a_splat_amd.h declares."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def built_lib():
    from sim_a_splat_amd import build
    return build.build()


def test_header_symbols_are_exported(built_lib):
    from sim_a_splat_amd import _capi
    header = (ROOT / "include" / "sim_a_splat_amd.h").read_text()
    declared = set(re.findall(r"\b(sas_[a-z_]+)\s*\(", header))
    assert declared == set(_capi.EXPORTS), declared ^ set(_capi.EXPORTS)
    L = ctypes.CDLL(str(built_lib))
    for name in declared:
        assert hasattr(L, name), name


def test_stat_and_stage_names_follow_the_header_enums():
    """The Python names of sas_frame_stats / sas_stage_times slots are the header's enums, in order."""
    from sim_a_splat_amd import _capi
    header = (ROOT / "include" / "sim_a_splat_amd.h").read_text()
    stats = [n for n in re.findall(r"\bSAS_S_([A-Z_]+)\b", header[header.index("enum { SAS_S_NVISIBLE"):header.index("SAS_S_COUNT")])]
    assert [n.lower() for n in stats] == [{"n_visible": "nvisible", "n_isect": "nisect", "n_keys": "nkeys"}.get(n, n) for n in _capi.STAT_NAMES]
    stages = re.findall(r"\bSAS_T_([A-Z]+)\b", header[header.index("enum { SAS_T_PROJECT"):header.index("SAS_T_COUNT")])
    assert [n.lower() for n in stages] == list(_capi.STAGE_NAMES)


def test_version_and_no_device_status(built_lib):
    import torch
    from sim_a_splat_amd import _capi
    L = _capi.lib()
    assert b"gfx950" in L.sas_version()
    if not torch.cuda.is_available():
        ctx = ctypes.c_void_p()
        assert L.sas_create(0, ctypes.byref(ctx)) == -5  # SAS_ERR_NO_DEVICE, no crash, no fallback
        assert not ctx.value


def test_product_never_imports_oracle():
    """The product path must not route through oracle/ (or any CPU fallback)."""
    # ... nor do the examples and the measurement tools: what runs the oracle lives under tests/ (tests/tools/),
    # beside bench.py's cpu_baseline leg and __graft_entry__ (build + smoke)
    for d in ("sim_a_splat_amd", "tools", "examples"):
        for py in (ROOT / d).rglob("*.py"):
            src = py.read_text()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), py
    for src in (ROOT / "sim_a_splat_amd" / "csrc").iterdir():
        # the kernels may NAME the oracle's source in comments (the arithmetic contract), never include it
        text = src.read_text()
        assert not re.search(r"#\s*include\s*[<\"][^>\"]*oracle", text), src
        assert "dlopen" not in text, src


def test_rasterizer_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from sim_a_splat_amd.rasterizer import Rasterizer
    from sim_a_splat_amd._capi import SasError
    with pytest.raises(SasError):
        Rasterizer(0)


def test_no_tracked_file_is_a_binary_object():
    """History stays source-only: no ELF / code-object / archive among the tracked files (built .so files travel
    to the GPU box untracked).  .npz fixtures are zip archives of arrays: data, allowed."""
    import subprocess
    res = subprocess.run(["git", "ls-files", "-z"], cwd=ROOT, capture_output=True)
    if res.returncode != 0:
        pytest.skip("not a git checkout (the GPU box receives a snapshot without .git)")
    bad = []
    for rel in filter(None, res.stdout.decode().split("\0")):
        p = ROOT / rel
        if not p.is_file():
            continue
        with open(p, "rb") as f:
            magic = f.read(8)
        if magic[:4] == b"\x7fELF" or magic[:7] == b"!<arch>" or magic[:8] == b"__CLANG_" or ".hipv4-" in rel or ".host-x86" in rel:
            bad.append(rel)
    assert not bad, bad


def test_dpp_operands_of_the_quad_layout_respect_the_read_hazard():
    """The quad layout's compositing rides quad broadcasts on its multiply-adds as DPP operands written in inline asm
    (csrc/sas_tile.hip, SAS_QFMAC), where hipcc's hazard recogniser does not look: on gfx9 a DPP source register must not
    have been written by a VALU instruction within the two preceding wait states.  tools/isa_audit.py compiles the kernel
    to gfx950 assembly (CPU only: hipcc cross-compiles) and checks every DPP instruction of it."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_audit", ROOT / "tools" / "isa_audit.py")
    ia = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ia)
    lines = ia.kernel_asm("k_tile_lazyILb0ELb0ELb1E", [])
    loops, per, hazards = ia.audit(lines)
    n_dpp = sum(1 for l in lines if "_dpp" in l and "quad_perm" in l)
    assert n_dpp >= 16, n_dpp                      # the sixteen broadcast multiply-adds of a trip are there to be checked
    assert not hazards, hazards[:5]
