"""Object wrapper over the C ABI: one ``Rasterizer`` = one ``sas_ctx`` on one GPU.

PyTorch is plumbing here (device memory for the outputs, the current HIP stream); all
arithmetic of the frame happens in libsas_hip.so.
"""
from __future__ import annotations

import ctypes
import functools
import threading
from typing import Dict, Iterable, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _capi
from ._capi import SasError

ArrayLike = Union[np.ndarray, torch.Tensor]


def _as_f32(a: ArrayLike, shape: Tuple[int, ...], name: str):
    """Return (keepalive, pointer) of a contiguous float32 array with the given shape."""
    if isinstance(a, torch.Tensor):
        t = a.detach()
        if t.dtype != torch.float32:
            t = t.float()
        t = t.reshape(shape).contiguous()
        return t, ctypes.c_void_p(t.data_ptr())
    arr = np.ascontiguousarray(np.asarray(a, dtype=np.float32)).reshape(shape)
    return arr, arr.ctypes.data_as(ctypes.c_void_p)


def cov3x3_to_cov6(cov: ArrayLike) -> ArrayLike:
    """[n,3,3] symmetric -> [n,6] (xx xy xz yy yz zz), the viser `covariances` argument (Door B)."""
    if isinstance(cov, torch.Tensor):
        c = cov.reshape(-1, 3, 3)
        return torch.stack([c[:, 0, 0], c[:, 0, 1], c[:, 0, 2], c[:, 1, 1], c[:, 1, 2], c[:, 2, 2]], dim=1).contiguous()
    c = np.asarray(cov, dtype=np.float32).reshape(-1, 3, 3)
    return np.stack([c[:, 0, 0], c[:, 0, 1], c[:, 0, 2], c[:, 1, 1], c[:, 1, 2], c[:, 2, 2]], axis=1)


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)   # (private, but what torch.cuda.current_stream itself calls)


def _locked(fn):
    """One caller at a time per context: the C ABI is not re-entrant (include/sim_a_splat_amd.h), and the reference's
    callers are not always single-threaded -- demo_hw_splat.py drives env.step from a ROS2 callback thread
    (examples/demo_hw_splat.py:113-136) while a viewer thread may render through the same scene."""
    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        with self._lock:
            return fn(self, *args, **kwargs)
    return wrapper


class Rasterizer:
    """MI355X Gaussian-splat rasterizer context (HIP, gfx950).  Every method that enters the C ABI holds the
    context's lock (re-entrant: a thread may nest calls)."""

    def __init__(self, device: Union[int, str, torch.device] = 0):
        if not torch.cuda.is_available():
            raise SasError("no HIP device visible: the render path has no CPU fallback")
        dev = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
        if dev.type != "cuda":
            raise SasError(f"Rasterizer needs a cuda (HIP) device, got {dev}")
        self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        self._dev_index = int(self.device.index)
        self._lock = threading.RLock()
        self._L = _capi.lib()
        self._ctx = ctypes.c_void_p()
        rc = self._L.sas_create(self.device.index, ctypes.byref(self._ctx))
        if rc != 0:
            raise SasError(f"sas_create(device={self.device.index}) failed with status {rc}")
        self.n = 0
        self.n_groups = 0
        self._keep = []  # outputs of in-flight async frames (the C ABI keeps up to four)
        self._argcache = {}  # id(argument) -> (argument, float32 array, address): _host_arg

    # -- lifetime ---------------------------------------------------------------------------
    @_locked
    def close(self) -> None:
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._L.sas_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str) -> None:
        _capi.check(self._ctx, rc, what)

    # -- scene ------------------------------------------------------------------------------
    @_locked
    def upload(self, means: ArrayLike, opacities: ArrayLike, colors: ArrayLike, *, quats: Optional[ArrayLike] = None,
               scales: Optional[ArrayLike] = None, covariances: Optional[ArrayLike] = None, sh_degree: int = 3,
               group_id: Optional[ArrayLike] = None, n_groups: int = 0) -> None:
        """Replace the scene.  ``sh_degree < 0``: ``colors`` is final RGB [n,3] (Door B)."""
        n = int(means.shape[0])
        keep = []
        m, pm = _as_f32(means, (n, 3), "means"); keep.append(m)
        o, po = _as_f32(opacities, (n,), "opacities"); keep.append(o)
        kk = (sh_degree + 1) ** 2 if sh_degree >= 0 else 1
        c, pc = _as_f32(colors, (n, kk, 3), "colors"); keep.append(c)
        pq = ps = pcov = None
        if quats is not None and scales is not None:
            q, pq = _as_f32(quats, (n, 4), "quats"); keep.append(q)
            s, ps = _as_f32(scales, (n, 3), "scales"); keep.append(s)
        elif covariances is not None:
            cov = covariances
            if tuple(cov.shape[1:]) == (3, 3):
                cov = cov3x3_to_cov6(cov)
            cv, pcov = _as_f32(cov, (n, 6), "covariances"); keep.append(cv)
        else:
            raise ValueError("need quats+scales or covariances")
        pg = None
        if group_id is not None:
            if isinstance(group_id, torch.Tensor):
                g = group_id.detach().to(torch.uint8).contiguous()
                pg = ctypes.c_void_p(g.data_ptr())
            else:
                g = np.ascontiguousarray(np.asarray(group_id, dtype=np.uint8))
                pg = g.ctypes.data_as(ctypes.c_void_p)
            keep.append(g)
            if n_groups <= 0:
                n_groups = int(g.max()) + 1 if n > 0 else 1
        torch.cuda.synchronize(self.device)  # device-resident inputs must be complete before the copy
        self._check(self._L.sas_scene_upload(self._ctx, n, pm, pq, ps, pcov, po, pc, int(sh_degree), pg, int(n_groups)),
                    "sas_scene_upload")
        self.n = n
        self.n_groups = int(n_groups) if group_id is not None else 0

    @_locked
    def set_group_poses(self, Rt: ArrayLike) -> None:
        """[G,12] (or [G,3,4]) row-major (R|t) per splat group: the poses of the frames submitted from now on
        (frames in flight keep the poses they were submitted with; nothing is waited for)."""
        arr = np.ascontiguousarray(np.asarray(Rt.cpu() if isinstance(Rt, torch.Tensor) else Rt, dtype=np.float32)).reshape(-1, 12)
        self._check(self._L.sas_set_group_poses(self._ctx, arr.shape[0], arr.ctypes.data_as(ctypes.c_void_p)),
                    "sas_set_group_poses")

    @_locked
    def set_link_constants(self, scale: float, Ri, ti, Rfk, tfk, weld=None, groups=None) -> None:
        """Constants of the per-link pose algebra (sas_set_link_constants): ICP similarity ``(scale, Ri [3,3], ti [3])``,
        per-link mask-time forward kinematics ``Rfk [K,3,3]``, ``tfk [K,3]``, weld translation, the group each link drives."""
        f64 = lambda a, shape: np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(shape))
        Rfk = f64(Rfk, (-1, 9))
        K = Rfk.shape[0]
        Ri, ti, tfk = f64(Ri, (9,)), f64(ti, (3,)), f64(tfk, (K, 3))
        w = f64(weld, (3,)) if weld is not None else None
        g = np.ascontiguousarray(np.asarray(groups, dtype=np.int32).reshape(K)) if groups is not None else None
        self._check(self._L.sas_set_link_constants(self._ctx, K, float(scale), Ri.ctypes.data, ti.ctypes.data, Rfk.ctypes.data,
                                                   tfk.ctypes.data, w.ctypes.data if w is not None else None,
                                                   g.ctypes.data if g is not None else None), "sas_set_link_constants")

    @_locked
    def set_link_poses(self, q_msg, p_msg, out: Optional[np.ndarray] = None) -> Optional[np.ndarray]:
        """A draw message's link poses (``q_msg [k,4]`` wxyz, ``p_msg [k,3]``) -> group poses, evaluated inside the
        library (sas_set_link_poses); ``out`` (float32, ``n_groups * 12`` elements) receives all current group poses."""
        q = np.ascontiguousarray(np.asarray(q_msg, dtype=np.float64).reshape(-1, 4))
        p = np.ascontiguousarray(np.asarray(p_msg, dtype=np.float64).reshape(-1, 3))
        if out is not None and not (out.dtype == np.float32 and out.flags.c_contiguous and out.size == 12 * self.n_groups):
            raise ValueError("out must be a contiguous float32 array of n_groups * 12 elements")
        rc = self._L.sas_set_link_poses(self._ctx, q.shape[0], q.ctypes.data, p.ctypes.data, out.ctypes.data if out is not None else None)
        if rc != 0:
            self._check(rc, "sas_set_link_poses")
        return out

    @_locked
    def link_attached_frame(self, q_link, p_link, local_xyz) -> Tuple[np.ndarray, np.ndarray]:
        """(wxyz, xyz) of a camera riding on a link (sas_link_attached_frame; the ICP similarity of set_link_constants)."""
        a = np.empty(17, np.float64)
        a[0:4], a[4:7], a[7:10] = q_link, p_link, local_xyz
        base = a.ctypes.data
        rc = self._L.sas_link_attached_frame(self._ctx, base, base + 32, base + 56, base + 80, base + 112)
        if rc != 0:
            self._check(rc, "sas_link_attached_frame")
        return a[10:14], a[14:17]

    @_locked
    def get_group_poses(self) -> np.ndarray:
        out = np.zeros((self.n_groups, 12), np.float32)
        self._check(self._L.sas_get_group_poses(self._ctx, self.n_groups, out.ctypes.data), "sas_get_group_poses")
        return out

    # -- frames -----------------------------------------------------------------------------
    _SHAPES = {"rgb": (3, torch.float32), "alpha": (1, torch.float32), "depth": (1, torch.float32),
               "rgb8": (3, torch.uint8)}

    @staticmethod
    def _host_f32(a, count: int) -> np.ndarray:
        if isinstance(a, torch.Tensor):
            a = a.detach().cpu().numpy()
        if not (isinstance(a, np.ndarray) and a.dtype == np.float32 and a.flags.c_contiguous and a.size == count):
            a = np.ascontiguousarray(np.asarray(a, dtype=np.float32)).reshape(count)
        return a

    def _stream(self) -> int:
        """The caller's current HIP stream on this context's device (1.9 us through torch.cuda.current_stream, 0.07 us
        through the raw getter it wraps: tools/py_overhead_probe.py)."""
        if _RAW_STREAM is not None:
            return _RAW_STREAM(self._dev_index)
        return torch.cuda.current_stream(self.device).cuda_stream

    def _host_arg(self, a, count: int):
        """(float32 host array, its address) of a call argument.  A call's K, background and often its view matrix are the
        SAME objects as in the call before: the last few (object, address) pairs are remembered -- `.ctypes.data` alone
        costs 1.1 us, three of them a per cent of a blocking 1080p frame.  Only arguments that need no conversion (the
        array IS what the C side reads, so writing into it between calls stays visible) and tuples (immutable) qualify."""
        hit = self._argcache.get(id(a))
        if hit is not None and hit[0] is a:
            return hit[1], hit[2]
        arr = self._host_f32(a, count)
        ptr = arr.ctypes.data
        if arr is a or isinstance(a, tuple):
            if len(self._argcache) >= 16:
                self._argcache.clear()
            self._argcache[id(a)] = (a, arr, ptr)   # (holds `a`: its id cannot be reused while the entry lives)
        return arr, ptr

    @_locked
    def render(self, viewmat: ArrayLike, K: ArrayLike, width: int, height: int,
               background: Sequence[float] = (0.0, 0.0, 0.0), *, want: Iterable[str] = ("rgb", "alpha", "depth"),
               depth_fill_max: bool = False, fast_exp: bool = False, timing: bool = False, block: bool = True,
               full_sort: bool = False, time_tiles: bool = False,
               out: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
        """Render one view; returns device tensors ``rgb [H,W,3]``, ``alpha [H,W,1]``,
        ``depth [H,W,1]`` (float32) and/or ``rgb8 [H,W,3]`` (uint8) as listed in ``want``.

        ``block=False`` only enqueues (SAS_ASYNC): up to four frames are in flight.  Frames complete in
        submission order inside later ``render*`` calls or ``wait()``; a frame's outputs may be consumed
        (and work on the current stream is ordered behind it) only once it is complete --
        ``frames_completed()`` tells how many are.  ``full_sort=True`` orders every tile list
        completely and keeps it for ``read_tile_lists`` (same image, slower)."""
        V, pV = self._host_arg(viewmat, 16)
        Kc, pK = self._host_arg(K, 9)
        bg, pbg = self._host_arg(background, 3)
        W, H = int(width), int(height)
        res: Dict[str, torch.Tensor] = {}
        ptrs = {"rgb": None, "alpha": None, "depth": None, "rgb8": None}
        for k in want:
            ch, dt = self._SHAPES[k]   # KeyError: unknown output
            t = out.get(k) if out is not None else None
            if t is None:
                t = torch.empty((H, W, ch), dtype=dt, device=self.device)
            elif t.shape != (H, W, ch) or t.dtype != dt or not t.is_contiguous() or t.device != self.device:
                raise ValueError(f"out[{k!r}] must be a contiguous {dt} tensor {(H, W, ch)} on {self.device}")
            res[k] = t
            ptrs[k] = t.data_ptr()
        flags = (_capi.SAS_DEPTH_FILL_MAX if depth_fill_max else 0) | (_capi.SAS_FAST_EXP if fast_exp else 0) | \
                (_capi.SAS_TIMING if timing else 0) | (0 if block else _capi.SAS_ASYNC) | \
                (_capi.SAS_FULL_SORT if full_sort else 0) | (_capi.SAS_TIME_TILES if time_tiles else 0)
        stream = self._stream()
        rc = self._L.sas_render(self._ctx, pV, pK, W, H, pbg, flags,
                                ptrs["rgb"], ptrs["alpha"], ptrs["depth"], ptrs["rgb8"], stream)
        if rc != 0:
            self._check(rc, "sas_render")
        self._keep = [] if block else (self._keep + [(res, V, Kc, bg)])[-4:]
        return res

    @_locked
    def render_rgbd(self, viewmat: ArrayLike, K: ArrayLike, width: int, height: int,
                    background: Sequence[float] = (0.0, 0.0, 0.0), *, max_depth: Optional[float] = 1.0,
                    depth_fill_max: bool = True) -> Dict[str, torch.Tensor]:
        """Render with the RGB-D consumer fused into the depth pass (sas_render_rgbd): besides
        ``rgb``/``alpha``/``depth`` returns ``points [H,W,3]`` (camera frame) and ``mask [H,W]``
        (bool, ``depth < max_depth``; all true for ``max_depth=None``) -- nerfstudio_utils.py:424-445."""
        V = self._host_f32(viewmat, 16)
        Kc = self._host_f32(K, 9)
        bg = self._host_f32(background, 3)
        W, H = int(width), int(height)
        res = {k: torch.empty((H, W, ch), dtype=dt, device=self.device)
               for k, (ch, dt) in self._SHAPES.items() if k != "rgb8"}
        res["points"] = torch.empty((H, W, 3), dtype=torch.float32, device=self.device)
        mask8 = torch.empty((H, W), dtype=torch.uint8, device=self.device)
        md = ctypes.c_float(max_depth) if max_depth is not None else None
        flags = _capi.SAS_DEPTH_FILL_MAX if depth_fill_max else 0
        stream = self._stream()
        rc = self._L.sas_render_rgbd(self._ctx, V.ctypes.data, Kc.ctypes.data, W, H, bg.ctypes.data, flags,
                                     ctypes.addressof(md) if md is not None else None,
                                     res["rgb"].data_ptr(), res["alpha"].data_ptr(), res["depth"].data_ptr(),
                                     res["points"].data_ptr(), mask8.data_ptr(), stream)
        if rc != 0:
            self._check(rc, "sas_render_rgbd")
        self._keep = []
        res["mask"] = mask8.view(torch.bool)
        return res

    def _pose_sets(self, pose_sets, pose_set, C: int):
        """(Rt [S,G,12] float32, index [C] int32) of per-view pose sets, validated."""
        Rt = np.ascontiguousarray(np.asarray(pose_sets.cpu() if isinstance(pose_sets, torch.Tensor) else pose_sets, dtype=np.float32))
        if self.n_groups <= 0 or Rt.size % (12 * self.n_groups):
            raise ValueError(f"pose_sets must be [S,{self.n_groups},12] for this scene")
        Rt = Rt.reshape(-1, self.n_groups, 12)
        idx = np.ascontiguousarray(np.asarray(pose_set, dtype=np.int32).reshape(-1))
        if idx.shape[0] != C:
            raise ValueError(f"pose_set must name one pose set per view ({C}), got {idx.shape[0]}")
        return Rt, idx

    @_locked
    def render_batch(self, viewmats: ArrayLike, Ks: ArrayLike, width: int, height: int,
                     background: Sequence[float] = (0.0, 0.0, 0.0), *, want: Iterable[str] = ("rgb",),
                     depth_fill_max: bool = False, block: bool = True, time_tiles: bool = False,
                     out: Optional[Dict[str, torch.Tensor]] = None, pose_sets: Optional[ArrayLike] = None,
                     pose_set: Optional[Sequence[int]] = None) -> Dict[str, torch.Tensor]:
        """Render C same-sized views in one C-ABI call: ``viewmats [C,4,4]``, ``Ks [C,3,3]`` ->
        tensors ``[C,H,W,...]`` (the per-camera loop of the reference, splat_env_wrapper.py:147-158).
        Views are projected two per pass over the scene.  ``block=False`` only enqueues (results valid
        after ``wait()``); ``out`` supplies the ``[C,H,W,...]`` output tensors.  ``pose_sets [S,G,12]`` +
        ``pose_set [C]``: view v is rendered with the group poses ``pose_sets[pose_set[v]]`` (vectorised envs:
        sas_render_batch_posed)."""
        C = int(np.asarray(viewmats).shape[0]) if not isinstance(viewmats, torch.Tensor) else int(viewmats.shape[0])
        V, pV = self._host_arg(viewmats, 16 * C)
        Kc, pK = self._host_arg(Ks, 9 * C)
        bg, pbg = self._host_arg(background, 3)
        W, H = int(width), int(height)
        res: Dict[str, torch.Tensor] = {}
        ptrs = {"rgb": None, "alpha": None, "depth": None, "rgb8": None}
        for k in want:
            ch, dt = self._SHAPES[k]
            t = out.get(k) if out is not None else None
            if t is None:
                t = torch.empty((C, H, W, ch), dtype=dt, device=self.device)
            elif t.shape != (C, H, W, ch) or t.dtype != dt or not t.is_contiguous() or t.device != self.device:
                raise ValueError(f"out[{k!r}] must be a contiguous {dt} tensor {(C, H, W, ch)} on {self.device}")
            res[k] = t
            ptrs[k] = t.data_ptr()
        flags = (_capi.SAS_DEPTH_FILL_MAX if depth_fill_max else 0) | (0 if block else _capi.SAS_ASYNC) | \
                (_capi.SAS_TIME_TILES if time_tiles else 0)
        stream = self._stream()
        if pose_sets is not None:
            Rt, idx = self._pose_sets(pose_sets, pose_set, C)
            rc = self._L.sas_render_batch_posed(self._ctx, C, pV, pK, idx.ctypes.data, Rt.shape[0], Rt.ctypes.data,
                                                W, H, pbg, flags, ptrs["rgb"], ptrs["alpha"], ptrs["depth"], ptrs["rgb8"], stream)
        else:
            rc = self._L.sas_render_batch(self._ctx, C, pV, pK, W, H, pbg, flags,
                                          ptrs["rgb"], ptrs["alpha"], ptrs["depth"], ptrs["rgb8"], stream)
        if rc != 0:
            self._check(rc, "sas_render_batch")
        self._keep = [] if block else (self._keep + [(res, V, Kc, bg)])[-4:]
        return res

    @_locked
    def render_batch_host(self, viewmats: ArrayLike, Ks: ArrayLike, width: int, height: int,
                          background: Sequence[float] = (0.0, 0.0, 0.0), *, out: Optional[torch.Tensor] = None,
                          pose_sets: Optional[ArrayLike] = None, pose_set: Optional[Sequence[int]] = None) -> torch.Tensor:
        """C same-sized views as uint8 frames ON THE HOST (sas_render_batch_host): a pinned ``[C,H,W,3]`` uint8 CPU
        tensor, filled on the frames' own streams right behind the tile kernels -- what Door B's ``get_render``
        hands out (np.uint8 arrays), without a second round trip for the device-to-host copy.  ``out`` supplies the
        tensor (CPU, uint8, contiguous; pinned for speed); otherwise a pinned one comes from torch's caching host
        allocator, so a caller may keep what it gets."""
        C = int(np.asarray(viewmats).shape[0]) if not isinstance(viewmats, torch.Tensor) else int(viewmats.shape[0])
        V, pV = self._host_arg(viewmats, 16 * C)
        Kc, pK = self._host_arg(Ks, 9 * C)
        bg, pbg = self._host_arg(background, 3)
        W, H = int(width), int(height)
        if out is None:
            out = torch.empty((C, H, W, 3), dtype=torch.uint8, pin_memory=True)
        elif out.shape != (C, H, W, 3) or out.dtype != torch.uint8 or not out.is_contiguous() or out.device.type != "cpu":
            raise ValueError(f"out must be a contiguous uint8 CPU tensor {(C, H, W, 3)}")
        stream = self._stream()
        if pose_sets is not None:
            Rt, idx = self._pose_sets(pose_sets, pose_set, C)
            rc = self._L.sas_render_batch_host_posed(self._ctx, C, pV, pK, idx.ctypes.data, Rt.shape[0],
                                                     Rt.ctypes.data, W, H, pbg, 0, out.data_ptr(), stream)
        else:
            rc = self._L.sas_render_batch_host(self._ctx, C, pV, pK, W, H, pbg, 0,
                                               out.data_ptr(), stream)
        if rc != 0:
            self._check(rc, "sas_render_batch_host")
        return out

    @_locked
    def render_cameras_host(self, wxyz, position, fov: float, width: int, height: int,
                            background: Sequence[float] = (0.0, 0.0, 0.0), *, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``render_batch_host`` from camera-to-world POSES (``wxyz [C,4]``, ``position [C,3]``, OpenCV axes) and a vertical
        field of view: the view matrices and intrinsics are computed inside the library (sas_render_cameras_host)."""
        q = np.ascontiguousarray(np.asarray(wxyz, dtype=np.float64).reshape(-1, 4))
        p = np.ascontiguousarray(np.asarray(position, dtype=np.float64).reshape(-1, 3))
        C, W, H = q.shape[0], int(width), int(height)
        bg = self._host_f32(background, 3)
        if out is None:
            out = torch.empty((C, H, W, 3), dtype=torch.uint8, pin_memory=True)
        elif out.shape != (C, H, W, 3) or out.dtype != torch.uint8 or not out.is_contiguous() or out.device.type != "cpu":
            raise ValueError(f"out must be a contiguous uint8 CPU tensor {(C, H, W, 3)}")
        stream = self._stream()
        rc = self._L.sas_render_cameras_host(self._ctx, C, q.ctypes.data, p.ctypes.data, float(fov), W, H, bg.ctypes.data, 0,
                                             out.data_ptr(), stream)
        if rc != 0:
            self._check(rc, "sas_render_cameras_host")
        return out

    @_locked
    def wait(self) -> None:
        self._check(self._L.sas_wait(self._ctx), "sas_wait")
        self._keep = []

    @_locked
    def frames_completed(self) -> Tuple[int, int]:
        """(submitted, completed) frame counts since creation (sas_frames_completed): frames
        ``0 .. completed-1`` are final and the current stream is ordered behind them."""
        sub, com = ctypes.c_int64(0), ctypes.c_int64(0)
        self._check(self._L.sas_frames_completed(self._ctx, ctypes.byref(sub), ctypes.byref(com)), "sas_frames_completed")
        return int(sub.value), int(com.value)

    # -- introspection ------------------------------------------------------------------------
    @_locked
    def stage_times(self) -> Dict[str, float]:
        ms = (ctypes.c_float * len(_capi.STAGE_NAMES))()
        self._check(self._L.sas_stage_times(self._ctx, ms, len(_capi.STAGE_NAMES)), "sas_stage_times")
        return dict(zip(_capi.STAGE_NAMES, [float(x) for x in ms]))

    @_locked
    def stage_time_means(self, reset: bool = True):
        """(mean ms per stage, frames) over the timed frames completed since the last reset."""
        ms = (ctypes.c_float * len(_capi.STAGE_NAMES))()
        nf = ctypes.c_int64(0)
        self._check(self._L.sas_stage_time_means(self._ctx, ms, len(_capi.STAGE_NAMES), ctypes.byref(nf), int(reset)),
                    "sas_stage_time_means")
        return dict(zip(_capi.STAGE_NAMES, [float(x) for x in ms])), int(nf.value)

    @_locked
    def stats(self) -> Dict[str, int]:
        st = (ctypes.c_int64 * len(_capi.STAT_NAMES))()
        self._check(self._L.sas_frame_stats(self._ctx, st, len(_capi.STAT_NAMES)), "sas_frame_stats")
        return dict(zip(_capi.STAT_NAMES, [int(x) for x in st]))

    @_locked
    def read_projection(self) -> Dict[str, np.ndarray]:
        n = self.n
        radii = np.zeros((n, 2), np.int32)
        means2d = np.zeros((n, 2), np.float32)
        depths = np.zeros((n,), np.float32)
        conics = np.zeros((n, 3), np.float32)
        colors = np.zeros((n, 3), np.float32)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        self._check(self._L.sas_read_projection(self._ctx, p(radii), p(means2d), p(depths), p(conics), p(colors)),
                    "sas_read_projection")
        return dict(radii=radii, means2d=means2d, depths=depths, conics=conics, colors=colors)

    @_locked
    def read_tile_lists(self, tiles: int) -> Dict[str, np.ndarray]:
        m = self.stats()["n_isect"]
        off = np.zeros((tiles + 1,), np.int32)
        ids = np.zeros((max(m, 1),), np.int32)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        self._check(self._L.sas_read_tile_lists(self._ctx, p(off), p(ids), m), "sas_read_tile_lists")
        return dict(tile_offsets=off, sorted_ids=ids[:m])
