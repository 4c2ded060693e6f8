"""Door A: ``GaussianSplat`` / ``load_model`` of sim_a_splat/ns_utils/nerfstudio_utils.py:51-177,500-516 on the HIP
rasterizer, with the reference's constructor::

    gsplat = GaussianSplat(config_path, res_factor=None, test_mode="inference", dataset_mode="test", device="cuda")
    H, W, K = gsplat.get_camera_intrinsics()
    out = gsplat.render(pose)            # {"rgb", "depth", "accumulation", "background"} device tensors

nerfstudio and gsplat are not dependencies.  ``ns_run.eval_setup`` reads the run (config.yml, the latest
checkpoint, the dataset's transforms.json through a restatement of the Nerfstudio dataparser); ``SplatModel`` keeps
the raw splatfacto parameters under the names the reference reads off ``pipeline.model`` (splat_utils.py:33-45)
and reproduces what ``SplatfactoModel.get_outputs`` does around the rasterizer: activations,
OpenGL->OpenCV view matrix, ``rgb = clamp(render + (1 - alpha) * background, 0, 1)``,
``depth = where(alpha > 0, ED, max ED)``.  ``GaussianSplat.from_model`` is the form with the model injected.
"""
from __future__ import annotations

import time
from dataclasses import dataclass
from pathlib import Path
from types import SimpleNamespace
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


def _scalar(v) -> float:
    return float(v.reshape(-1)[0]) if isinstance(v, (torch.Tensor, np.ndarray)) else float(v)


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
        self._bg_src = None

    @property
    def num_points(self) -> int:
        return int(self.means.shape[0])

    def _rasterizer(self) -> Rasterizer:
        if self._raster is None:
            # eval_setup puts the model on "cuda if available" whatever device the caller names
            # (nerfstudio_utils.py:505, eval_utils.eval_setup); there is no CPU render path here
            r = Rasterizer(self.device if self.device.type == "cuda" else 0)
            kk = (self.sh_degree + 1) ** 2
            # checkpoints carry all 15 higher-order coefficients; the bands in use are the first (d+1)^2 - 1
            rest = self.features_rest.reshape(self.num_points, -1, 3)[:, :kk - 1]
            colors = torch.cat([self.features_dc.reshape(-1, 1, 3), rest], dim=1)
            # activations of get_outputs: exp(scales), sigmoid(opacities); quats stay un-normalised
            r.upload(self.means, torch.sigmoid(self.opacities).reshape(-1), colors, quats=self.quats,
                     scales=torch.exp(self.scales), sh_degree=self.sh_degree)
            self._raster = r
        return self._raster

    @torch.no_grad()
    def get_outputs_for_camera(self, camera, obb_box=None, compute_semantics: bool = False) -> Dict[str, torch.Tensor]:
        """``camera``: a one-camera ``ns_run.Cameras`` or a ``PinholeCamera``."""
        if obb_box is not None:
            raise NotImplementedError("crop boxes are outside the render-image path (the reference passes None)")
        if compute_semantics:
            raise TypeError("compute_semantics is not supported by a splatfacto model")  # reference retries without it
        c2w = camera.camera_to_worlds
        c2w = c2w[0] if c2w.dim() == 3 else c2w
        V = viewmat_from_c2w_opengl(c2w)
        K = np.array([[_scalar(camera.fx), 0.0, _scalar(camera.cx)], [0.0, _scalar(camera.fy), _scalar(camera.cy)],
                      [0.0, 0.0, 1.0]], dtype=np.float32)
        W, H = int(_scalar(camera.width)), int(_scalar(camera.height))
        return self.render_view(V, K, W, H)

    def render_view(self, V: np.ndarray, K: np.ndarray, W: int, H: int) -> Dict[str, torch.Tensor]:
        """The outputs of ``get_outputs_for_camera`` from a world-to-camera matrix and intrinsics (what ``GaussianSplat.render``
        calls: the camera objects of the reference cost more host time per call than a 640x480 frame takes to render)."""
        bg = self.background_color
        r = self._rasterizer()
        if self._bg_src is not bg:                     # (re)assigned background: its list and device copy, once
            self._bg_src, self._bg_list, self._bg_dev = bg, [float(v) for v in bg.reshape(-1)], bg.to(r.device)
        out = r.render(V, K, W, H, self._bg_list, want=("rgb", "alpha", "depth"), depth_fill_max=True)
        return {"rgb": out["rgb"], "depth": out["depth"], "accumulation": out["alpha"], "background": self._bg_dev.expand(H, W, 3)}


class _Dataset:
    """``pipeline.datamanager.{train,eval}_dataset`` as far as the reference reads it: ``.cameras`` and the
    image file names (the images themselves are not shipped with a run)."""

    def __init__(self, dp_out: Dict):
        self.cameras = dp_out["cameras"]
        self._dataparser_outputs = SimpleNamespace(image_filenames=[Path(f) for f in dp_out["image_filenames"]],
                                                   dataparser_scale=dp_out["scale"], dataparser_transform=dp_out["transform"])

    def __len__(self) -> int:
        return len(self.cameras)

    def get_image_float32(self, image_idx: int) -> torch.Tensor:
        """The training / eval image ``image_idx`` as float32 in 0..1, as nerfstudio 1.1.5's ``InputDataset.get_image_float32``
        returns it (restated from public knowledge; the reference's ``get_images`` collects these, nerfstudio_utils.py:101-111):
        the file read as uint8 (resized bilinearly by the dataset's ``scale_factor`` when that is not 1), a grey image repeated
        to three channels, divided by 255; an alpha channel is blended onto the dataparser's ``alpha_color`` ONLY when the
        dataparser sets one -- the ``nerfstudio`` dataparser of the reference's runs does not, so an RGBA file comes back with
        its four channels ``[H,W,4]``.  The images are the capture a run was trained on; the reference's assets do not ship
        them (only ``transforms.json`` names them), so a missing file raises ``FileNotFoundError`` naming the path expected."""
        f = self._dataparser_outputs.image_filenames[image_idx]
        if not Path(f).is_file():
            raise FileNotFoundError(f"{f}: image {image_idx} of the run's dataset is not on disk (the reference's assets ship "
                                    f"transforms.json and the run, not the captured images)")
        from PIL import Image
        pil = Image.open(f)
        scale = float(getattr(self, "scale_factor", 1.0))
        if scale != 1.0:
            w, h = pil.size
            pil = pil.resize((int(w * scale), int(h * scale)), resample=Image.BILINEAR)
        img = np.asarray(pil, dtype=np.uint8)
        if img.ndim == 2:
            img = img[:, :, None].repeat(3, axis=2)
        if img.shape[2] not in (3, 4):
            raise ValueError(f"{f}: image shape {img.shape} is not [H,W,3] or [H,W,4]")
        x = torch.from_numpy(img.astype(np.float32) / 255.0)
        alpha_color = getattr(self._dataparser_outputs, "alpha_color", None)
        if alpha_color is not None and x.shape[-1] == 4:
            ac = torch.as_tensor(alpha_color, dtype=torch.float32)
            x = x[:, :, :3] * x[:, :, -1:] + ac * (1.0 - x[:, :, -1:])
        return x


class GaussianSplat:
    """``GaussianSplat(config_path, res_factor, test_mode, dataset_mode, device)`` of the reference
    (nerfstudio_utils.py:51-121): ``config_path`` is the ``config.yml`` of a splatfacto run."""

    def __init__(self, config_path: Path, res_factor=None, test_mode: str = "inference", dataset_mode: str = "test",
                 device: Union[torch.device, str] = "cpu") -> None:
        self.config_path = config_path
        self.res_factor = res_factor
        self.device = device
        self.init_pipeline(test_mode)
        self.load_dataset(dataset_mode)
        self.get_cameras()

    @classmethod
    def from_model(cls, model: SplatModel, camera, res_factor: Optional[float] = None,
                   device: Union[torch.device, str, None] = None) -> "GaussianSplat":
        """The same object around a model already in memory: ``camera`` (a ``PinholeCamera`` or ``ns_run.Cameras``)
        plays the dataset's cameras (its first camera gives the intrinsics ``render`` uses)."""
        from . import ns_run
        self = cls.__new__(cls)
        self.config_path, self.res_factor = None, res_factor
        self.device = torch.device(device) if device is not None else model.device
        self.config = None
        if isinstance(camera, PinholeCamera):
            c2w = camera.camera_to_worlds.reshape(-1, 3, 4)
            camera = ns_run.Cameras(c2w, camera.fx, camera.fy, camera.cx, camera.cy, camera.width, camera.height)
        ds = _Dataset(dict(cameras=camera, image_filenames=[], scale=1.0, transform=torch.eye(4)[:3]))
        self.pipeline = SimpleNamespace(model=model, datamanager=SimpleNamespace(train_dataset=ds, eval_dataset=ds))
        self.dataset = ds
        self.get_cameras()
        return self

    # -- nerfstudio_utils.py:77-100 ------------------------------------------------------------------------
    def init_pipeline(self, test_mode: str):
        from . import ns_run
        run = ns_run.eval_setup(Path(self.config_path), test_mode=test_mode)
        g = run["gauss_params"]
        model = SplatModel(g["means"], g["scales"], g["quats"], g["features_dc"], g["features_rest"], g["opacities"],
                           sh_degree=run["sh_degree"], background_color=run["background"],
                           device=self.device if torch.device(self.device).type == "cuda" else "cuda:0")
        self.config = run["config"]
        self._run = run
        self.pipeline = SimpleNamespace(model=model, datamanager=SimpleNamespace(train_dataset=_Dataset(run["train"]),
                                                                                 eval_dataset=_Dataset(run["eval"])))

    def load_dataset(self, dataset_mode: str):
        if dataset_mode == "train":
            self.dataset = self.pipeline.datamanager.train_dataset
        elif dataset_mode in ["val", "test"]:
            self.dataset = self.pipeline.datamanager.eval_dataset
        else:
            # the reference builds this ValueError without raising it (:90-93) and then fails on self.dataset
            raise ValueError('Incorrect value for datset_mode. Accepted values include: dataset_mode: Literal["train", "val", "test"].')

    def get_cameras(self):
        self.cameras = self.dataset.cameras
        self._cam0 = None
        if self.res_factor is not None:
            self.cameras.rescale_output_resolution(self.res_factor)

    def get_poses(self):
        return self.cameras.camera_to_worlds

    def get_images(self):
        """nerfstudio_utils.py:101-111: every image of the chosen split as float32 [H,W,3] (see ``_Dataset.get_image_float32``)."""
        return [self.dataset.get_image_float32(image_idx) for image_idx in range(len(self.dataset._dataparser_outputs.image_filenames))]

    def get_camera_intrinsics(self) -> Tuple[int, int, torch.Tensor]:
        K = self.cameras[0].get_intrinsics_matrices().squeeze()
        W = int(self.cameras[0].width.item())
        H = int(self.cameras[0].height.item())
        return H, W, K

    def _camera0(self):
        """(stamp, K float32 [3,3], W, H) of camera 0, cached until the cameras change."""
        k, cams = self._cam0, self.cameras
        stamp = (cams, cams.fx, getattr(cams.fx, "_version", 0), cams.width)   # rescale_output_resolution replaces the tensors
        if k is None or k[0][0] is not cams or k[0][1] is not stamp[1] or k[0][2] != stamp[2] or k[0][3] is not stamp[3]:
            c0 = cams[0]
            K = np.array([[_scalar(c0.fx), 0.0, _scalar(c0.cx)], [0.0, _scalar(c0.fy), _scalar(c0.cy)], [0.0, 0.0, 1.0]], dtype=np.float32)
            k = self._cam0 = (stamp, K, int(_scalar(c0.width)), int(_scalar(c0.height)))
        return k

    # -- :123-177 ------------------------------------------------------------------------------------------
    def render(self, pose, compute_semantics: Optional[bool] = False, debug_mode: bool = False) -> Dict[str, torch.Tensor]:
        """``pose``: [>=3,4] camera-to-world, OpenGL axes, nerfstudio scene frame."""
        from . import ns_run
        model = self.pipeline.model
        tnow = time.perf_counter()
        if isinstance(model, SplatModel) and not compute_semantics:
            # the reference builds a one-camera Cameras object per call from camera 0's intrinsics (:127-136); the same
            # numbers without the dozen small tensor operations (70 of a 200-microsecond call at 640x480)
            k = self._camera0()
            c2w = pose[:3] if isinstance(pose, (np.ndarray, torch.Tensor)) else np.asarray(pose, dtype=np.float32)[:3]
            outputs = model.render_view(viewmat_from_c2w_opengl(c2w), k[1], k[2], k[3])
        else:
            camera_to_world = torch.as_tensor(pose, dtype=torch.float32)[None, :3, ...]
            c0 = self.cameras[0]
            cameras = ns_run.Cameras(camera_to_world, c0.fx, c0.fy, c0.cx, c0.cy, c0.width, c0.height)
            try:
                outputs = model.get_outputs_for_camera(cameras, obb_box=None, compute_semantics=compute_semantics)
            except TypeError:
                outputs = model.get_outputs_for_camera(cameras, obb_box=None)
        if debug_mode:
            torch.cuda.synchronize(outputs["rgb"].device)
            print("Rendering time: ", time.perf_counter() - tnow)
        return outputs

    @property
    def model(self) -> SplatModel:
        return self.pipeline.model

    def generate_RGBD_point_cloud(self, pose, max_depth: Optional[float] = 1.0):
        """RGB-D consumer of nerfstudio_utils.py:375-472 (tensor results only; the open3d cloud is
        ``points[mask]`` / ``rgb[mask]``).  Render and unprojection are one C-ABI call: the depth
        tail kernel writes the camera-frame points and the depth mask."""
        c2w = torch.as_tensor(pose, dtype=torch.float32)[:3, ...]
        V = viewmat_from_c2w_opengl(c2w)
        H, W, K = self.get_camera_intrinsics()
        model = self.pipeline.model
        bg = model.background_color
        r = model._rasterizer()
        out = r.render_rgbd(V, K.numpy(), W, H, bg.tolist(), max_depth=max_depth)
        outputs = {"rgb": out["rgb"], "depth": out["depth"], "accumulation": out["alpha"],
                   "background": bg.to(r.device).expand(H, W, 3)}
        return out["rgb"], out["points"], None, out["mask"], outputs


def load_model(config_path: Path) -> GaussianSplat:
    """``load_model`` of the reference (nerfstudio_utils.py:500-516)."""
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    return GaussianSplat(config_path=config_path, res_factor=None, test_mode="test", dataset_mode="val", device=device)
