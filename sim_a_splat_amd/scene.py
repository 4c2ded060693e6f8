"""Door B: the viser surface the Gym wrapper renders through, without viser or a browser.

Mirrors ``server.scene.add_gaussian_splats(...)`` handles with settable ``.wxyz/.position``
(sim_a_splat/splat/splat_handler.py:106-141, :283-288) and ``client.get_render(height, width,
wxyz, position)`` (sim_a_splat/env/splat/splat_env_wrapper.py:148-157).  All groups live in one
HIP scene; their poses go to the GPU as one [G,12] block per frame.
"""
from __future__ import annotations

import threading
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .poses import matrix_to_quat_wxyz, mv3, quat_wxyz_to_matrix, quats_wxyz_to_matrices
from .rasterizer import Rasterizer

DEFAULT_VERTICAL_FOV = float(np.deg2rad(75.0))   # the reference never passes a FOV; viser uses the client's


class GaussianSplatHandle:
    """What ``add_gaussian_splats`` returns: a group whose pose can be reassigned every step.  The scene keeps all
    group poses in one float32 ``[G,3,4]`` block (what goes to the GPU); a handle writes its row when it is assigned and
    reads it back when the poses were last set by the library's own link algebra (``SplatScene.set_link_poses``)."""

    def __init__(self, scene: "SplatScene", name: str, index: int, wxyz, position):
        self._scene, self.name, self.index = scene, name, index
        self._wxyz = np.asarray(wxyz, dtype=np.float64)
        self._position = np.asarray(position, dtype=np.float64)
        self._gen = scene._links_gen

    def _refresh(self) -> None:
        if self._gen != self._scene._links_gen:       # the block was rewritten by set_link_poses since
            row = self._scene._Rt[self.index].astype(np.float64)
            self._wxyz, self._position = matrix_to_quat_wxyz(row[:, :3]), row[:, 3].copy()
            self._gen = self._scene._links_gen

    def _write_row(self) -> None:
        sc = self._scene
        sc._Rt[self.index, :, :3] = quat_wxyz_to_matrix(self._wxyz)
        sc._Rt[self.index, :, 3] = self._position
        sc._poses_dirty = True

    @property
    def wxyz(self) -> np.ndarray:
        self._refresh()
        return self._wxyz

    @wxyz.setter
    def wxyz(self, v) -> None:
        with self._scene.lock:
            self._refresh()
            self._wxyz = np.asarray(v, dtype=np.float64)
            self._write_row()

    @property
    def position(self) -> np.ndarray:
        self._refresh()
        return self._position

    @position.setter
    def position(self, v) -> None:
        with self._scene.lock:
            self._refresh()
            self._position = np.asarray(v, dtype=np.float64)
            self._write_row()


_NO_OWNER_APPLIED = object()   # SplatScene._link_owner_applied: the library's context holds nobody's link constants


class _Camera:
    def __init__(self):
        self.wxyz = np.array([1.0, 0.0, 0.0, 0.0])
        self.position = np.zeros(3)
        self.fov = DEFAULT_VERTICAL_FOV


class SplatScene:
    """Registered splat groups + the renderer (plays both ``server.scene`` and the client).

    Thread safety: ``lock`` (re-entrant) is held from the pose hand-over to the end of the render in every
    ``get_render*`` call, by ``add_gaussian_splats`` and by the handles' setters, so a viewer thread
    (``ViserBridge`` camera callbacks) and ``env.step`` on another thread (examples/demo_hw_splat.py:113-136 steps from
    a ROS2 callback) serialise on one scene instead of entering the C context together; a caller that assigns several
    handles as one update (``SplatHandler.draw_handler``) takes the lock around the whole update."""

    def __init__(self, device=0, background: Sequence[float] = (0.0, 0.0, 0.0)):
        self.lock = threading.RLock()
        self._raster = Rasterizer(device)
        self._groups: List[Dict[str, np.ndarray]] = []
        self._handles: List[GaussianSplatHandle] = []
        self._uploaded = False
        self._poses_dirty = True
        self._Rt = np.zeros((0, 3, 4), np.float32)     # all group poses, the block that goes to the GPU
        self._links_gen = 0                            # bumped when set_link_poses rewrites the block
        # constants of the link-pose algebra PER OWNER (a SplatHandler: its ICP similarity, forward kinematics, weld, groups).
        # The library's context holds one set; the scene applies the caller's before each use, so that two handlers on one
        # scene (two robots, `instance_uid`) never pose their links or cameras with each other's constants.
        self._link_consts: Dict[object, tuple] = {}
        self._link_owner_applied = _NO_OWNER_APPLIED
        self.background = tuple(background)
        self.camera = _Camera()

    def add_gaussian_splats(self, name: str, centers, covariances, rgbs, opacities, wxyz=(1.0, 0.0, 0.0, 0.0),
                            position=(0.0, 0.0, 0.0)) -> GaussianSplatHandle:
        c = np.ascontiguousarray(np.asarray(centers, dtype=np.float32).reshape(-1, 3))
        n = c.shape[0]
        with self.lock:
            if len(self._groups) >= 256:
                raise RuntimeError("at most 256 splat groups")
            self._groups.append(dict(
                centers=c,
                covariances=np.asarray(covariances, dtype=np.float32).reshape(n, 3, 3),
                rgbs=np.asarray(rgbs, dtype=np.float32).reshape(n, 3),
                opacities=np.asarray(opacities, dtype=np.float32).reshape(n)))
            h = GaussianSplatHandle(self, name, len(self._handles), wxyz, position)
            self._handles.append(h)
            row = np.zeros((1, 3, 4), np.float32)
            self._Rt = np.concatenate([self._Rt, row], axis=0)
            h._write_row()
            self._uploaded = False
            self._poses_dirty = True
        return h

    # -- the draw message's pose algebra inside the library (SplatHandler.draw_handler's fast path) ---------------
    def set_link_constants(self, scale: float, Ri, ti, Rfk, tfk, weld=None, groups=None, owner=None) -> None:
        """See ``Rasterizer.set_link_constants``; kept per ``owner`` (the handler they belong to) and applied to the
        library's context whenever that owner's poses or cameras are evaluated next."""
        with self.lock:
            self._link_consts[owner] = (float(scale), np.array(Ri, np.float64), np.array(ti, np.float64), np.array(Rfk, np.float64),
                                        np.array(tfk, np.float64), None if weld is None else np.array(weld, np.float64),
                                        None if groups is None else np.array(groups, np.int32))
            if self._link_owner_applied is owner or self._link_owner_applied == owner:
                self._link_owner_applied = _NO_OWNER_APPLIED   # re-apply on the next use

    def _apply_link_constants(self, owner) -> None:
        """(lock held, scene uploaded)  Make ``owner``'s constants the ones the library's context holds."""
        if owner not in self._link_consts:
            raise RuntimeError("set_link_constants first")
        if self._link_owner_applied is _NO_OWNER_APPLIED or self._link_owner_applied != owner:
            self._raster.set_link_constants(*self._link_consts[owner])
            self._link_owner_applied = owner

    def group_pose_rows(self) -> np.ndarray:
        """A copy of every group's current pose as float32 rows [G,12] (what goes to the GPU): the base of a per-env pose set."""
        with self.lock:
            return self._Rt.reshape(-1, 12).copy()

    def set_link_poses(self, q_msg, p_msg, owner=None) -> None:
        """The first k links' message poses -> their groups' poses (sas_set_link_poses: float64 in C, the arithmetic of
        ``poses.link_splat_poses`` + the quaternion round trip of the handles), with ``owner``'s constants; the other
        groups keep theirs."""
        with self.lock:
            if owner not in self._link_consts:
                raise RuntimeError("set_link_constants first")
            self._sync()                                    # scene + whatever the handles were assigned since
            self._apply_link_constants(owner)
            self._raster.set_link_poses(q_msg, p_msg, out=self._Rt.reshape(-1))
            self._links_gen += 1

    # -- internals ---------------------------------------------------------------------------
    def _sync(self) -> None:
        if not self._uploaded:
            if not self._groups:
                z = np.zeros
                self._raster.upload(z((0, 3), np.float32), z((0,), np.float32), z((0, 3), np.float32),
                                    covariances=z((0, 6), np.float32), sh_degree=-1)
            else:
                cat = lambda k: np.concatenate([g[k] for g in self._groups], axis=0)
                gid = np.concatenate([np.full(g["centers"].shape[0], i, dtype=np.uint8) for i, g in enumerate(self._groups)])
                self._raster.upload(cat("centers"), cat("opacities"), cat("rgbs"), covariances=cat("covariances"),
                                    sh_degree=-1, group_id=gid, n_groups=len(self._groups))
            self._uploaded = True
            self._poses_dirty = True
            self._link_owner_applied = _NO_OWNER_APPLIED    # a fresh upload: the context holds nobody's link constants
        if self._poses_dirty and self._groups:
            self._raster.set_group_poses(self._Rt.reshape(-1, 12))
        self._poses_dirty = False

    @staticmethod
    def _view_and_K(height: int, width: int, wxyz, position, fov: float):
        R = quat_wxyz_to_matrix(wxyz)                      # camera-to-world, OpenCV axes (+z forward, +y down)
        V = np.eye(4)
        V[:3, :3] = R.T
        V[:3, 3] = -mv3(R.T, np.asarray(position, dtype=np.float64))
        f = 0.5 * height / np.tan(0.5 * fov)               # vertical FOV, square pixels
        K = np.array([[f, 0, 0.5 * width], [0, f, 0.5 * height], [0, 0, 1]])
        return V.astype(np.float32), K.astype(np.float32)

    @staticmethod
    def _views_and_Ks(height: int, width: int, wxyz: np.ndarray, position: np.ndarray, fov: float):
        """``_view_and_K`` for C cameras at once: ([C,4,4], [C,3,3]) float32, the same arithmetic per camera."""
        R = quats_wxyz_to_matrices(wxyz)
        C = R.shape[0]
        V = np.zeros((C, 4, 4))
        Rt = np.transpose(R, (0, 2, 1))
        V[:, :3, :3] = Rt
        V[:, :3, 3] = -mv3(Rt, np.asarray(position, dtype=np.float64).reshape(C, 3))
        V[:, 3, 3] = 1.0
        f = 0.5 * height / np.tan(0.5 * fov)
        K = np.array([[f, 0, 0.5 * width], [0, f, 0.5 * height], [0, 0, 1]])
        return V.astype(np.float32), np.broadcast_to(K.astype(np.float32), (C, 3, 3)).copy()

    # -- client side ---------------------------------------------------------------------------
    def get_render(self, height: int, width: int, wxyz=None, position=None, fov: Optional[float] = None) -> np.ndarray:
        """uint8 [H,W,3] frame from a camera pose (camera-to-world, OpenCV axes)."""
        wxyz = self.camera.wxyz if wxyz is None else wxyz
        position = self.camera.position if position is None else position
        with self.lock:
            self._sync()
            # view matrix and intrinsics inside the library (sas_camera_matrices: the arithmetic of _view_and_K)
            return self._raster.render_cameras_host(wxyz, position, self.camera.fov if fov is None else float(fov), int(width), int(height),
                                                    self.background).numpy()[0]

    def get_renders(self, height: int, width: int, cam_poses, fov: Optional[float] = None) -> np.ndarray:
        """uint8 [C,H,W,3] for C same-sized cameras ``[(wxyz, position), ...]`` in one batched call."""
        f = self.camera.fov if fov is None else float(fov)
        C = len(cam_poses)
        qp = np.empty((2, C, 4), np.float64)               # wxyz rows, then xyz rows (padded): one allocation
        for c, (w, p) in enumerate(cam_poses):
            qp[0, c] = w
            qp[1, c, :3] = p
        with self.lock:
            self._sync()
            # view matrices and intrinsics inside the library (sas_camera_matrices: the arithmetic of _views_and_Ks); frames
            # land in pinned host memory on the frames' own streams (sas_render_batch_host): no second round trip
            return self._raster.render_cameras_host(qp[0], qp[1, :, :3], f, int(width), int(height), self.background).numpy()

    def get_renders_posed(self, height: int, width: int, cam_poses, pose_sets, pose_set, fov: Optional[float] = None,
                          out: Optional[torch.Tensor] = None, device_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """uint8 [C,H,W,3] (pinned host tensor) for C same-sized cameras of SEVERAL envs in one call: view v is rendered
        with the group poses ``pose_sets[pose_set[v]]`` ([S,G,12] float32; vectorised envs, sas_render_batch_host_posed)
        instead of the scene's current ones; ``out`` supplies the tensor.  ``device_out`` (a uint8 [C,H,W,3] tensor on the
        scene's GPU) keeps the frames ON THE DEVICE instead (sas_render_batch_posed): what a multi-GPU rollout gathers
        over RCCL without the frames ever visiting the host on the way."""
        f = self.camera.fov if fov is None else float(fov)
        C = len(cam_poses)
        q, p = np.empty((C, 4), np.float64), np.empty((C, 3), np.float64)
        for c, (w, x) in enumerate(cam_poses):
            q[c], p[c] = w, x
        V, K = self._views_and_Ks(int(height), int(width), q, p, f)
        with self.lock:
            self._sync()
            if device_out is not None:
                return self._raster.render_batch(V, K, int(width), int(height), self.background, want=("rgb8",), out={"rgb8": device_out},
                                                 pose_sets=pose_sets, pose_set=pose_set)["rgb8"]
            return self._raster.render_batch_host(V, K, int(width), int(height), self.background, out=out, pose_sets=pose_sets, pose_set=pose_set)

    def attached_frame(self, q_link, p_link, local_xyz, owner=None):
        """(wxyz, xyz) of a camera riding on a link, with the similarity of ``owner``'s ``set_link_constants`` (sas_link_attached_frame)."""
        with self.lock:
            if owner not in self._link_consts:
                raise RuntimeError("set_link_constants first")
            self._sync()
            self._apply_link_constants(owner)
            return self._raster.link_attached_frame(q_link, p_link, local_xyz)

    def get_render_float(self, height: int, width: int, wxyz, position, fov: Optional[float] = None) -> Dict[str, torch.Tensor]:
        V, K = self._view_and_K(int(height), int(width), wxyz, position, self.camera.fov if fov is None else float(fov))
        with self.lock:
            self._sync()
            return self._raster.render(V, K, int(width), int(height), self.background, want=("rgb", "alpha", "depth"))

    def close(self) -> None:
        with self.lock:
            self._raster.close()
