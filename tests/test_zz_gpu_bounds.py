"""Bounds-checked build of the HIP library (-DSAS_DEBUG_BOUNDS; GPU AddressSanitizer is not available on the
pool).  Runs LAST in the GPU session (file name):

    python -m sim_a_splat_amd.build --variant bounds -DSAS_DEBUG_BOUNDS
    SAS_LIB_PATH=variants/lib_bounds.so python -m pytest tests -m gpu -q

Every index the kernels compute into the key buffers, the LDS queues / chunk / histogram, the projected
records, the tile tables and the output images is tested on the device; a violation is counted and the
access skipped.  After the whole GPU suite (all parity cases: ragged and empty inputs, overflow + regrow,
coplanar fallback, launch groups, camera inside the cloud ...) the count must be zero.  With the product
library these tests skip.
"""
import ctypes

import pytest

pytestmark = pytest.mark.gpu


def _lib_with_bounds():
    from sim_a_splat_amd import _capi
    L = _capi.lib()
    if not hasattr(L, "sas_debug_bounds"):
        pytest.skip("product library: no bounds instrumentation (run with SAS_LIB_PATH=variants/lib_bounds.so)")
    L.sas_debug_bounds.argtypes = [ctypes.c_void_p, ctypes.c_int]
    return L


def test_bounds_build_saw_no_violation_in_the_whole_gpu_suite():
    L = _lib_with_bounds()
    out = (ctypes.c_uint64 * 4)()
    assert L.sas_debug_bounds(out, 0) == 0
    assert out[0] == 0, f"{out[0]} out-of-range accesses; first: code {out[1]}, index {out[2]}, limit {out[3]}"


def test_bounds_checker_counts_a_deliberate_violation():
    L = _lib_with_bounds()
    out = (ctypes.c_uint64 * 4)()
    assert L.sas_debug_bounds(out, 1) == 0
    assert L.sas_debug_bounds_selftest() == 0            # the out-of-range access was skipped ...
    assert L.sas_debug_bounds(out, 1) == 0
    assert (out[0], out[1], out[2], out[3]) == (1, 999, 5, 4)   # ... and reported
    assert L.sas_debug_bounds(out, 0) == 0 and out[0] == 0
