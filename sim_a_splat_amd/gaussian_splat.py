"""Door A: the ``GaussianSplat.render(pose)`` / ``model.get_outputs_for_camera`` surface of
sim_a_splat/ns_utils/nerfstudio_utils.py:51-177, backed by the HIP rasterizer.

nerfstudio and gsplat are not dependencies: ``SplatModel`` keeps the raw splatfacto parameters
(the ``gauss_params`` names of the checkpoint) and reproduces what ``SplatfactoModel.get_outputs``
does around the rasterizer: activations, OpenGL->OpenCV view matrix,
``rgb = clamp(render + (1 - alpha) * background, 0, 1)``, ``depth = where(alpha > 0, ED, max ED)``.
"""
from __future__ import annotations

import time
from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .rasterizer import Rasterizer
from .synthetic import NERFSTUDIO_EVAL_BACKGROUND


def viewmat_from_c2w_opengl(c2w: Union[np.ndarray, torch.Tensor]) -> np.ndarray:
    """nerfstudio get_viewmat: flip y/z (OpenGL -> OpenCV), invert the rigid transform.  float32."""
    c = np.asarray(c2w.detach().cpu() if isinstance(c2w, torch.Tensor) else c2w, dtype=np.float32)
    R = c[:3, :3] * np.array([1.0, -1.0, -1.0], dtype=np.float32)[None, :]
    t = c[:3, 3]
    V = np.eye(4, dtype=np.float32)
    V[:3, :3] = R.T
    V[:3, 3] = -(R.T @ t)
    return V


@dataclass
class PinholeCamera:
    """The fields of a one-camera nerfstudio ``Cameras`` that the render path reads
    (nerfstudio_utils.py:127-136)."""
    camera_to_worlds: torch.Tensor   # [1,3,4] or [3,4], OpenGL axes
    fx: float
    fy: float
    cx: float
    cy: float
    width: int
    height: int

    def get_intrinsics_matrices(self) -> torch.Tensor:
        return torch.tensor([[self.fx, 0.0, self.cx], [0.0, self.fy, self.cy], [0.0, 0.0, 1.0]], dtype=torch.float32)


class SplatModel:
    """Counterpart of ``pipeline.model`` (a trained SplatfactoModel) for the render path."""

    def __init__(self, means, scales, quats, features_dc, features_rest, opacities, sh_degree: int = 3,
                 background_color: Sequence[float] = NERFSTUDIO_EVAL_BACKGROUND, device: Union[str, torch.device] = "cuda:0"):
        """Raw (pre-activation) parameters, as stored in a splatfacto checkpoint: log ``scales``,
        ``opacities`` logits [N,1], ``features_dc`` [N,3], ``features_rest`` [N,(d+1)^2-1,3]."""
        self.device = torch.device(device)
        t = lambda a: torch.as_tensor(a, dtype=torch.float32)
        self.means, self.scales, self.quats = t(means), t(scales), t(quats)
        self.features_dc, self.features_rest, self.opacities = t(features_dc), t(features_rest), t(opacities)
        self.sh_degree = int(sh_degree)
        self.background_color = torch.tensor(list(background_color), dtype=torch.float32)
        self._raster: Optional[Rasterizer] = None

    @property
    def num_points(self) -> int:
        return int(self.means.shape[0])

    def _rasterizer(self) -> Rasterizer:
        if self._raster is None:
            r = Rasterizer(self.device)
            kk = (self.sh_degree + 1) ** 2
            colors = torch.cat([self.features_dc.reshape(-1, 1, 3), self.features_rest.reshape(-1, kk - 1, 3)], dim=1)
            # activations of get_outputs: exp(scales), sigmoid(opacities); quats stay un-normalised
            r.upload(self.means, torch.sigmoid(self.opacities).reshape(-1), colors, quats=self.quats,
                     scales=torch.exp(self.scales), sh_degree=self.sh_degree)
            self._raster = r
        return self._raster

    @torch.no_grad()
    def get_outputs_for_camera(self, camera: PinholeCamera, obb_box=None, compute_semantics: bool = False) -> Dict[str, torch.Tensor]:
        if obb_box is not None:
            raise NotImplementedError("crop boxes are outside the render-image path (the reference passes None)")
        if compute_semantics:
            raise TypeError("compute_semantics is not supported by a splatfacto model")  # reference retries without it
        c2w = camera.camera_to_worlds
        c2w = c2w[0] if c2w.dim() == 3 else c2w
        V = viewmat_from_c2w_opengl(c2w)
        K = camera.get_intrinsics_matrices().numpy()
        W, H = int(camera.width), int(camera.height)
        bg = self.background_color
        out = self._rasterizer().render(V, K, W, H, bg.tolist(), want=("rgb", "alpha", "depth"), depth_fill_max=True)
        return {"rgb": out["rgb"], "depth": out["depth"], "accumulation": out["alpha"],
                "background": bg.to(self.device).expand(H, W, 3)}


class GaussianSplat:
    """``GaussianSplat`` of the reference with the model injected instead of ``eval_setup``."""

    def __init__(self, model: SplatModel, camera: PinholeCamera, res_factor: Optional[float] = None,
                 device: Union[torch.device, str, None] = None) -> None:
        self.device = torch.device(device) if device is not None else model.device
        self.model = model
        self.res_factor = res_factor
        if res_factor is not None:   # Cameras.rescale_output_resolution
            camera = PinholeCamera(camera.camera_to_worlds, camera.fx * res_factor, camera.fy * res_factor,
                                   camera.cx * res_factor, camera.cy * res_factor,
                                   int(np.floor(camera.width * res_factor + 0.5)), int(np.floor(camera.height * res_factor + 0.5)))
        self.camera0 = camera

    def get_camera_intrinsics(self) -> Tuple[int, int, torch.Tensor]:
        return int(self.camera0.height), int(self.camera0.width), self.camera0.get_intrinsics_matrices()

    def render(self, pose, compute_semantics: Optional[bool] = False, debug_mode: bool = False) -> Dict[str, torch.Tensor]:
        """``pose``: [>=3,4] camera-to-world, OpenGL axes, nerfstudio scene frame (nerfstudio_utils.py:123-177)."""
        c2w = torch.as_tensor(pose, dtype=torch.float32)[None, :3, ...]
        cam = PinholeCamera(c2w, self.camera0.fx, self.camera0.fy, self.camera0.cx, self.camera0.cy,
                            self.camera0.width, self.camera0.height)
        tnow = time.perf_counter()
        try:
            outputs = self.model.get_outputs_for_camera(cam, obb_box=None, compute_semantics=compute_semantics)
        except TypeError:
            outputs = self.model.get_outputs_for_camera(cam, obb_box=None)
        if debug_mode:
            torch.cuda.synchronize(self.device)
            print("Rendering time: ", time.perf_counter() - tnow)
        return outputs

    def generate_RGBD_point_cloud(self, pose, max_depth: Optional[float] = 1.0):
        """RGB-D consumer of nerfstudio_utils.py:375-472 (tensor results only; the open3d cloud is
        ``points[mask]`` / ``rgb[mask]``).  Render and unprojection are one C-ABI call: the depth
        tail kernel writes the camera-frame points and the depth mask."""
        c2w = torch.as_tensor(pose, dtype=torch.float32)[:3, ...]
        V = viewmat_from_c2w_opengl(c2w)
        H, W, K = self.get_camera_intrinsics()
        bg = self.model.background_color
        out = self.model._rasterizer().render_rgbd(V, K.numpy(), W, H, bg.tolist(), max_depth=max_depth)
        outputs = {"rgb": out["rgb"], "depth": out["depth"], "accumulation": out["alpha"],
                   "background": bg.to(self.device).expand(H, W, 3)}
        return out["rgb"], out["points"], None, out["mask"], outputs
