"""ctypes binding of libsas_hip.so (include/sim_a_splat_amd.h).

There is no CPU fallback: if the shared library is missing or fails to load, importing this
module's ``lib()`` raises.  The oracle under ``oracle/`` is never imported from here.
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path
from typing import Optional

_PKG = Path(__file__).resolve().parent
# SAS_LIB_PATH points the binding at another build of the same library (the -DSAS_TUNE_* / -DSAS_DEBUG_BOUNDS
# variants under variants/, for A/B measurements and the bounds-checked test run)
LIB_PATH = Path(os.environ.get("SAS_LIB_PATH") or (_PKG / "libsas_hip.so"))

SAS_DEPTH_FILL_MAX = 1
SAS_ASYNC = 2
SAS_FAST_EXP = 4
SAS_TIMING = 8
SAS_FULL_SORT = 16
SAS_TIME_TILES = 32

STAGE_NAMES = ("project", "scan", "scatter", "sort", "blend", "tail", "total")
STAT_NAMES = ("n_visible", "n_isect", "max_tile_len", "capacity", "regrows", "window_misses", "fallback_tiles", "quad_layout", "launch_views", "n_keys")

# every symbol include/sim_a_splat_amd.h declares
EXPORTS = (
    "sas_create", "sas_destroy", "sas_scene_upload", "sas_set_group_poses", "sas_set_link_constants", "sas_set_link_poses", "sas_get_group_poses", "sas_link_attached_frame", "sas_link_group_poses", "sas_attached_frame", "sas_camera_matrices", "sas_render_cameras_host", "sas_render", "sas_render_rgbd", "sas_render_batch", "sas_render_batch_host", "sas_render_batch_posed", "sas_render_batch_host_posed", "sas_wait", "sas_frames_completed",
    "sas_last_error", "sas_stage_times", "sas_stage_time_means", "sas_frame_stats", "sas_read_projection", "sas_read_tile_lists",
    "sas_version",
)

_lib: Optional[ctypes.CDLL] = None


class SasError(RuntimeError):
    """Raised for any non-zero status of the C ABI (the reference raises RuntimeError too,
    sim_a_splat/env/splat/splat_env_wrapper.py:93-94)."""


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise SasError(
            f"{LIB_PATH} is missing: build it with `python -m sim_a_splat_amd.build` "
            "(hipcc, gfx950). There is no CPU fallback for the render path.")
    # ONE HIP runtime per process: the library is handed torch's device pointers and streams, so it has to resolve libamdhip64 to the
    # copy torch has loaded.  Loaded BEFORE torch it binds the system copy instead, and sas_create then finds no device
    # (SAS_ERR_NO_DEVICE) once torch has brought its own -- so torch comes first whenever it is installed.
    try:
        import torch  # noqa: F401
    except ImportError:   # a C / ctypes consumer without torch: the system runtime is the only one
        pass
    L = ctypes.CDLL(str(LIB_PATH))
    vp, ci, cu, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_int64
    L.sas_create.argtypes = [ci, ctypes.POINTER(vp)]
    L.sas_destroy.argtypes = [vp]
    L.sas_scene_upload.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, ci, vp, ci]
    L.sas_set_group_poses.argtypes = [vp, ci, vp]
    L.sas_set_link_constants.argtypes = [vp, ci, ctypes.c_double, vp, vp, vp, vp, vp, vp]
    L.sas_set_link_poses.argtypes = [vp, ci, vp, vp, vp]
    L.sas_get_group_poses.argtypes = [vp, ci, vp]
    cd = ctypes.c_double
    L.sas_link_attached_frame.argtypes = [vp, vp, vp, vp, vp, vp]
    L.sas_link_group_poses.argtypes = [ci, cd, vp, vp, vp, vp, vp, vp, vp, vp]
    L.sas_attached_frame.argtypes = [cd, vp, vp, vp, vp, vp, vp, vp]
    L.sas_camera_matrices.argtypes = [ci, vp, vp, cd, ci, ci, vp, vp]
    L.sas_render_cameras_host.argtypes = [vp, ci, vp, vp, cd, ci, ci, vp, cu, vp, vp]
    L.sas_render.argtypes = [vp, vp, vp, ci, ci, vp, cu, vp, vp, vp, vp, vp]
    L.sas_render_rgbd.argtypes = [vp, vp, vp, ci, ci, vp, cu, vp, vp, vp, vp, vp, vp, vp]
    L.sas_render_batch.argtypes = [vp, ci, vp, vp, ci, ci, vp, cu, vp, vp, vp, vp, vp]
    L.sas_render_batch_host.argtypes = [vp, ci, vp, vp, ci, ci, vp, cu, vp, vp]
    L.sas_render_batch_posed.argtypes = [vp, ci, vp, vp, vp, ci, vp, ci, ci, vp, cu, vp, vp, vp, vp, vp]
    L.sas_render_batch_host_posed.argtypes = [vp, ci, vp, vp, vp, ci, vp, ci, ci, vp, cu, vp, vp]
    L.sas_wait.argtypes = [vp]
    L.sas_frames_completed.argtypes = [vp, vp, vp]
    L.sas_last_error.argtypes = [vp]
    L.sas_last_error.restype = ctypes.c_char_p
    L.sas_stage_times.argtypes = [vp, vp, ci]
    L.sas_stage_time_means.argtypes = [vp, vp, ci, vp, ci]
    L.sas_frame_stats.argtypes = [vp, vp, ci]
    L.sas_read_projection.argtypes = [vp, vp, vp, vp, vp, vp]
    L.sas_read_tile_lists.argtypes = [vp, vp, vp, i64]
    L.sas_version.restype = ctypes.c_char_p
    for name in EXPORTS:
        if name not in ("sas_last_error", "sas_version"):
            getattr(L, name).restype = ci
    _lib = L
    return L


def check(ctx, rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().sas_last_error(ctx).decode() if ctx else ""
        raise SasError(f"{what} failed (status {rc}): {msg}")
