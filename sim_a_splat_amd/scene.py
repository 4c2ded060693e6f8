"""Door B: the viser surface the Gym wrapper renders through, without viser or a browser.

Mirrors ``server.scene.add_gaussian_splats(...)`` handles with settable ``.wxyz/.position``
(sim_a_splat/splat/splat_handler.py:106-141, :283-288) and ``client.get_render(height, width,
wxyz, position)`` (sim_a_splat/env/splat/splat_env_wrapper.py:148-157).  All groups live in one
HIP scene; their poses go to the GPU as one [G,12] block per frame.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .poses import quat_wxyz_to_matrix, quats_wxyz_to_matrices
from .rasterizer import Rasterizer

DEFAULT_VERTICAL_FOV = float(np.deg2rad(75.0))   # the reference never passes a FOV; viser uses the client's


class GaussianSplatHandle:
    """What ``add_gaussian_splats`` returns: a group whose pose can be reassigned every step."""

    def __init__(self, scene: "SplatScene", name: str, index: int, wxyz, position):
        self._scene, self.name, self.index = scene, name, index
        self._wxyz = np.asarray(wxyz, dtype=np.float64)
        self._position = np.asarray(position, dtype=np.float64)

    @property
    def wxyz(self) -> np.ndarray:
        return self._wxyz

    @wxyz.setter
    def wxyz(self, v) -> None:
        self._wxyz = np.asarray(v, dtype=np.float64)
        self._scene._poses_dirty = True

    @property
    def position(self) -> np.ndarray:
        return self._position

    @position.setter
    def position(self, v) -> None:
        self._position = np.asarray(v, dtype=np.float64)
        self._scene._poses_dirty = True


class _Camera:
    def __init__(self):
        self.wxyz = np.array([1.0, 0.0, 0.0, 0.0])
        self.position = np.zeros(3)
        self.fov = DEFAULT_VERTICAL_FOV


class SplatScene:
    """Registered splat groups + the renderer (plays both ``server.scene`` and the client)."""

    def __init__(self, device=0, background: Sequence[float] = (0.0, 0.0, 0.0)):
        self._raster = Rasterizer(device)
        self._groups: List[Dict[str, np.ndarray]] = []
        self._handles: List[GaussianSplatHandle] = []
        self._uploaded = False
        self._poses_dirty = True
        self.background = tuple(background)
        self.camera = _Camera()

    def add_gaussian_splats(self, name: str, centers, covariances, rgbs, opacities, wxyz=(1.0, 0.0, 0.0, 0.0),
                            position=(0.0, 0.0, 0.0)) -> GaussianSplatHandle:
        c = np.ascontiguousarray(np.asarray(centers, dtype=np.float32).reshape(-1, 3))
        n = c.shape[0]
        if len(self._groups) >= 256:
            raise RuntimeError("at most 256 splat groups")
        self._groups.append(dict(
            centers=c,
            covariances=np.asarray(covariances, dtype=np.float32).reshape(n, 3, 3),
            rgbs=np.asarray(rgbs, dtype=np.float32).reshape(n, 3),
            opacities=np.asarray(opacities, dtype=np.float32).reshape(n)))
        h = GaussianSplatHandle(self, name, len(self._handles), wxyz, position)
        self._handles.append(h)
        self._uploaded = False
        self._poses_dirty = True
        return h

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
        if self._poses_dirty and self._groups:
            Rt = np.empty((len(self._handles), 3, 4), dtype=np.float32)
            Rt[:, :, :3] = quats_wxyz_to_matrices(np.stack([h.wxyz for h in self._handles]))
            Rt[:, :, 3] = np.stack([h.position for h in self._handles])
            self._raster.set_group_poses(Rt.reshape(-1, 12))
        self._poses_dirty = False

    @staticmethod
    def _view_and_K(height: int, width: int, wxyz, position, fov: float):
        R = quat_wxyz_to_matrix(wxyz)                      # camera-to-world, OpenCV axes (+z forward, +y down)
        V = np.eye(4)
        V[:3, :3] = R.T
        V[:3, 3] = -R.T @ np.asarray(position, dtype=np.float64)
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
        V[:, :3, 3] = -np.stack([Rt[c] @ np.asarray(position[c], dtype=np.float64) for c in range(C)])
        V[:, 3, 3] = 1.0
        f = 0.5 * height / np.tan(0.5 * fov)
        K = np.array([[f, 0, 0.5 * width], [0, f, 0.5 * height], [0, 0, 1]])
        return V.astype(np.float32), np.broadcast_to(K.astype(np.float32), (C, 3, 3)).copy()

    # -- client side ---------------------------------------------------------------------------
    def get_render(self, height: int, width: int, wxyz=None, position=None, fov: Optional[float] = None) -> np.ndarray:
        """uint8 [H,W,3] frame from a camera pose (camera-to-world, OpenCV axes)."""
        self._sync()
        wxyz = self.camera.wxyz if wxyz is None else wxyz
        position = self.camera.position if position is None else position
        V, K = self._view_and_K(int(height), int(width), wxyz, position, self.camera.fov if fov is None else float(fov))
        return self._raster.render_batch_host(V[None], K[None], int(width), int(height), self.background).numpy()[0]

    def get_renders(self, height: int, width: int, cam_poses, fov: Optional[float] = None) -> np.ndarray:
        """uint8 [C,H,W,3] for C same-sized cameras ``[(wxyz, position), ...]`` in one batched call."""
        self._sync()
        f = self.camera.fov if fov is None else float(fov)
        Vs, Ks = self._views_and_Ks(int(height), int(width), np.stack([np.asarray(w, dtype=np.float64) for w, _ in cam_poses]),
                                    [p for _, p in cam_poses], f)
        # frames land in pinned host memory on the frames' own streams (sas_render_batch_host): no second round trip
        return self._raster.render_batch_host(Vs, Ks, int(width), int(height), self.background).numpy()

    def get_render_float(self, height: int, width: int, wxyz, position, fov: Optional[float] = None) -> Dict[str, torch.Tensor]:
        self._sync()
        V, K = self._view_and_K(int(height), int(width), wxyz, position, self.camera.fov if fov is None else float(fov))
        return self._raster.render(V, K, int(width), int(height), self.background, want=("rgb", "alpha", "depth"))

    def close(self) -> None:
        self._raster.close()
