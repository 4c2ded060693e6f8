"""CPU oracle for the render-image hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  Nothing under ``sim_a_splat_amd/`` imports it; the product path fails loudly when
its HIP library is missing instead of falling back to this code.

``oracle.sas_oracle.c``  float32 C restatement (the checker; OpenMP for the CPU baseline timing)
``oracle.np_twin``       float64 NumPy twin used to cross-check the C file
``oracle.ref_math``      NumPy restatement of the in-tree host math (compute_cov, SH2RGB, poses)

Parity status: rows T0-T7 are **parity unpinned** (third-party arithmetic absent from
/root/reference, no reference tests); in-tree rows are pinned by tests/golden/.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path
from typing import Dict, Optional

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "_build" / "libsas_oracle.so"
_lib = None


def build(force: bool = False) -> Path:
    """Compile oracle/sas_oracle.c with gcc (seconds)."""
    src = _HERE / "sas_oracle.c"
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), "-s"], check=True)
    return _LIB_PATH


class _Scene(ctypes.Structure):
    _fields_ = [
        ("n", ctypes.c_int64),
        ("means", ctypes.c_void_p),
        ("quats", ctypes.c_void_p),
        ("scales", ctypes.c_void_p),
        ("cov6", ctypes.c_void_p),
        ("opacities", ctypes.c_void_p),
        ("colors", ctypes.c_void_p),
        ("sh_degree", ctypes.c_int32),
        ("group_id", ctypes.c_void_p),
        ("n_groups", ctypes.c_int32),
        ("group_Rt", ctypes.c_void_p),
    ]


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(str(_LIB_PATH))
        _lib.sas_oracle_expf.restype = ctypes.c_float
        _lib.sas_oracle_expf.argtypes = [ctypes.c_float]
        _lib.sas_oracle_logf.restype = ctypes.c_float
        _lib.sas_oracle_logf.argtypes = [ctypes.c_float]
        _lib.sas_oracle_render.restype = ctypes.c_int
        _lib.sas_oracle_render.argtypes = [ctypes.POINTER(_Scene)] + [ctypes.c_void_p] * 2 + [ctypes.c_int] * 2 + \
            [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 11 + [ctypes.c_int64, ctypes.c_void_p]
        _lib.sas_oracle_unproject.restype = None
        _lib.sas_oracle_unproject.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + \
            [ctypes.c_void_p] * 3
        _lib.sas_oracle_trace_pixel.restype = ctypes.c_int
        _lib.sas_oracle_trace_pixel.argtypes = [ctypes.POINTER(_Scene)] + [ctypes.c_void_p] * 2 + [ctypes.c_int] * 6 + [ctypes.c_void_p] * 2
        _lib.sas_oracle_num_threads.restype = ctypes.c_int
        _lib.sas_oracle_set_variant.argtypes = [ctypes.c_int]
        _lib.sas_oracle_set_num_threads.argtypes = [ctypes.c_int]
    return _lib


def _f32(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    if shape is not None:
        a = a.reshape(shape)
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def expf(x: float) -> float:
    return float(lib().sas_oracle_expf(ctypes.c_float(x)))


def logf(x: float) -> float:
    return float(lib().sas_oracle_logf(ctypes.c_float(x)))


VARIANT_TEXTBOOK_SIGMA, VARIANT_SIGMA_GUARD, VARIANT_T_PRODUCT, VARIANT_LIBM_EXP = 1, 2, 4, 8
VARIANT_POLYNOMIAL_SIGMA = 32   # study: the contract of rounds 1-4 (sigma as a polynomial in the tile-local pixel centre); | 16: about the tile's corner (rounds 1-3)


class variant:
    """``with oracle.variant(mask): ...`` -- switch compositing to textbook float32 forms for the deviation
    study (sas_oracle.c blend_pixel_variant); mask 0 is the contract."""

    def __init__(self, mask: int):
        self.mask = int(mask)

    def __enter__(self):
        lib().sas_oracle_set_variant(self.mask)
        return self

    def __exit__(self, *exc):
        lib().sas_oracle_set_variant(0)
        return False


def num_threads() -> int:
    return int(lib().sas_oracle_num_threads())


def set_num_threads(n: int) -> None:
    lib().sas_oracle_set_num_threads(int(n))


def render(means, opacities, colors, viewmat, K, width: int, height: int, *, quats=None, scales=None,
           cov6=None, sh_degree: int = 3, group_id=None, group_Rt=None,
           background=(0.0, 0.0, 0.0), depth_mode: int = 0, want_rgb8: bool = False,
           dump: bool = False) -> Dict[str, np.ndarray]:
    """Render one frame with the C oracle.  ``sh_degree < 0`` means ``colors`` is final RGB [N,3]."""
    L = lib()
    means = _f32(means, (-1, 3))
    n = means.shape[0]
    quats = _f32(quats, (-1, 4))
    scales = _f32(scales, (-1, 3))
    cov6 = _f32(cov6, (-1, 6))
    if quats is None and cov6 is None:
        raise ValueError("need quats+scales or cov6")
    opacities = _f32(opacities, (-1,))
    kk = (sh_degree + 1) ** 2 if sh_degree >= 0 else 1
    colors = _f32(colors, (n, kk, 3))
    gid = None if group_id is None else np.ascontiguousarray(np.asarray(group_id, dtype=np.uint8))
    gRt = _f32(group_Rt, (-1, 12))
    sc = _Scene(n, _ptr(means), _ptr(quats), _ptr(scales), _ptr(cov6), _ptr(opacities), _ptr(colors),
                int(sh_degree), _ptr(gid), 0 if gRt is None else gRt.shape[0], _ptr(gRt))
    V = _f32(viewmat, (16,))
    Km = _f32(K, (9,))
    bg = _f32(background, (3,))
    W, H = int(width), int(height)
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    rgb = np.empty((H, W, 3), np.float32)
    alpha = np.empty((H, W, 1), np.float32)
    depth = np.empty((H, W, 1), np.float32)
    rgb8 = np.empty((H, W, 3), np.uint8) if want_rgb8 else None
    out: Dict[str, np.ndarray] = {}
    radii = means2d = depths = conics = cols = toff = sids = None
    cap = 0
    if dump:
        radii = np.zeros((n, 2), np.int32)
        means2d = np.zeros((n, 2), np.float32)
        depths = np.zeros((n,), np.float32)
        conics = np.zeros((n, 3), np.float32)
        cols = np.zeros((n, 3), np.float32)
        toff = np.zeros((tiles + 1,), np.int32)
    stats = np.zeros(3, np.int64)
    if dump:
        # first call to learn M, second to fetch ids (cheap at test sizes)
        rc = L.sas_oracle_render(ctypes.byref(sc), _ptr(V), _ptr(Km), W, H, _ptr(bg), depth_mode,
                                 None, None, None, None, None, None, None, None, None, None, None, 0, _ptr(stats))
        if rc != 0:
            raise MemoryError("oracle allocation failed")
        cap = int(stats[1])
        sids = np.zeros((max(cap, 1),), np.int32)
    rc = L.sas_oracle_render(ctypes.byref(sc), _ptr(V), _ptr(Km), W, H, _ptr(bg), int(depth_mode),
                             _ptr(rgb), _ptr(alpha), _ptr(depth), _ptr(rgb8),
                             _ptr(radii), _ptr(means2d), _ptr(depths), _ptr(conics), _ptr(cols),
                             _ptr(toff), _ptr(sids), cap, _ptr(stats))
    if rc != 0:
        raise MemoryError("oracle allocation failed")
    out.update(rgb=rgb, alpha=alpha, depth=depth, n_visible=int(stats[0]), n_isect=int(stats[1]))
    if want_rgb8:
        out["rgb8"] = rgb8
    if dump:
        out.update(radii=radii, means2d=means2d, depths=depths, conics=conics, colors=cols,
                   tile_offsets=toff, sorted_ids=sids[:cap])
    return out


DECISIONS = ("composited", "skipped: alpha < 1/255", "skipped: sigma < 0", "stopped: T' <= 1e-4")


def trace_pixel(means, opacities, colors, viewmat, K, width: int, height: int, px: int, py: int, variant_a: int, variant_b: int, *,
                quats=None, scales=None, cov6=None, sh_degree: int = 3, group_id=None, group_Rt=None) -> Dict[str, object]:
    """Where two variant masks part on pixel (px, py): the first list entry whose DECISION differs (deviation study)."""
    L = lib()
    means = _f32(means, (-1, 3))
    n = means.shape[0]
    quats, scales, cov6 = _f32(quats, (-1, 4)), _f32(scales, (-1, 3)), _f32(cov6, (-1, 6))
    opacities = _f32(opacities, (-1,))
    kk = (sh_degree + 1) ** 2 if sh_degree >= 0 else 1
    colors = _f32(colors, (n, kk, 3))
    gid = None if group_id is None else np.ascontiguousarray(np.asarray(group_id, dtype=np.uint8))
    gRt = _f32(group_Rt, (-1, 12))
    sc = _Scene(n, _ptr(means), _ptr(quats), _ptr(scales), _ptr(cov6), _ptr(opacities), _ptr(colors),
                int(sh_degree), _ptr(gid), 0 if gRt is None else gRt.shape[0], _ptr(gRt))
    V, Km = _f32(viewmat, (16,)), _f32(K, (9,))
    out = np.zeros(4, np.int64)
    vals = np.zeros(6, np.float32)
    if L.sas_oracle_trace_pixel(ctypes.byref(sc), _ptr(V), _ptr(Km), int(width), int(height), int(px), int(py), int(variant_a),
                                int(variant_b), _ptr(out), _ptr(vals)) != 0:
        raise MemoryError("oracle allocation failed")
    return dict(entry=int(out[0]), gaussian=int(out[1]), decision_a=DECISIONS[int(out[2])], decision_b=DECISIONS[int(out[3])],
                alpha_a=float(vals[0]), alpha_b=float(vals[1]), next_T_a=float(vals[2]), next_T_b=float(vals[3]),
                sigma_a=float(vals[4]), sigma_b=float(vals[5]))


def unproject(depth, K, max_depth=1.0):
    """Camera-frame points [H,W,3] and mask [H,W] of a depth image (nerfstudio_utils.py:424-445)."""
    d = _f32(np.asarray(depth).reshape(np.asarray(depth).shape[0], -1))
    H, W = d.shape
    Kc = _f32(K, (3, 3))
    pts = np.zeros((H, W, 3), np.float32)
    mask = np.zeros((H, W), np.uint8)
    md = ctypes.c_float(max_depth) if max_depth is not None else None
    lib().sas_oracle_unproject(_ptr(d), _ptr(Kc), W, H, ctypes.byref(md) if md is not None else None, _ptr(pts), _ptr(mask))
    return pts, mask.astype(bool)


def render_scene(scene, cam, **kw) -> Dict[str, np.ndarray]:
    """Convenience for ``sim_a_splat_amd.synthetic`` dataclasses."""
    return render(scene.means, scene.opacities, scene.sh, cam.viewmat, cam.K, cam.width, cam.height,
                  quats=scene.quats, scales=scene.scales, sh_degree=scene.sh_degree,
                  group_id=kw.pop("group_id", None), **kw)
