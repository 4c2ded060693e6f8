"""Gaussian activation and covariance assembly.

``compute_cov`` serves sim_a_splat/ellipsoids/covariance_utils.py:152-157.  The reference goes
quaternion -> angle-axis -> Rodrigues (with a +1e-6 in the denominator, :91); this is the closed
form of the same rotation, equal to it within ~1e-6 and, unlike it, finite for a quaternion whose
vector part is exactly zero (the reference returns NaN there: 0/0 * 0 in its mask blend, :54-57).
"""
from __future__ import annotations

import torch

C0 = 0.28209479177387814  # degree-0 SH basis (nerfstudio_utils.py:43)


def sh2rgb(sh: torch.Tensor) -> torch.Tensor:
    """Degree-0 colour, no clamp (nerfstudio_utils.py:46-47)."""
    return sh * C0 + 0.5


def quaternion_to_rotation_matrix(quat: torch.Tensor) -> torch.Tensor:
    q = quat / torch.linalg.norm(quat, dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.reshape(q.shape[:-1] + (3, 3))


def compute_cov(quat: torch.Tensor, scaling: torch.Tensor, exp: bool = False) -> torch.Tensor:
    """Sigma = (R diag(s)) (R diag(s))^T  for [N,4] wxyz quaternions and [N,3] scales."""
    s = torch.exp(scaling) if exp else scaling
    M = quaternion_to_rotation_matrix(quat) * s.unsqueeze(-2)
    return M @ M.transpose(-2, -1)


class GSplatLoader:
    """``GSplatLoader(gsplat_location, device)`` of the reference (sim_a_splat/splat/splat_utils.py:13-89): a ``str``
    is the reference's JSON scene file, a ``Path`` the ``config.yml`` of a nerfstudio splatfacto run (loaded through
    ``GaussianSplat(path, test_mode="inference", dataset_mode="train", device)``, kept as ``.splat``); anything else
    raises the reference's ValueError.  Attributes: ``means rots scales covs covs_inv colors opacities``.

    ``GSplatLoader.from_arrays`` takes the raw arrays; ``GSplatLoader.from_path`` also reads a bare checkpoint, a run
    directory without its dataset, or an ``.npz``."""

    def __init__(self, gsplat_location, device="cpu"):
        from pathlib import Path
        self.device = device
        if isinstance(gsplat_location, str):
            self.load_gsplat_from_json(gsplat_location)
        elif isinstance(gsplat_location, Path):
            self.load_gsplat_from_nerfstudio(gsplat_location)
        else:
            raise ValueError("GSplat file must be either a .json or .yml file.")

    def _activate(self, means, quats, log_scales, features_dc, opacity_logits) -> None:
        dev = torch.device(self.device)
        f32 = lambda a: torch.as_tensor(a, dtype=torch.float32).detach().clone().to(dev)
        self.means = f32(means)
        self.rots = f32(quats)
        self.scales = torch.exp(f32(log_scales))                     # :35-36
        self.covs_inv = compute_cov(self.rots, 1.0 / self.scales)    # :38
        self.covs = compute_cov(self.rots, self.scales)              # :39
        self.colors = sh2rgb(f32(features_dc).reshape(-1, 3))        # :41
        self.opacities = torch.sigmoid(f32(opacity_logits)).reshape(-1, 1)   # :43-45

    def load_gsplat_from_nerfstudio(self, gsplat_location) -> None:
        from .gaussian_splat import GaussianSplat
        self.splat = GaussianSplat(gsplat_location, test_mode="inference", dataset_mode="train", device=self.device)
        m = self.splat.pipeline.model
        self._activate(m.means, m.quats, m.scales, m.features_dc, m.opacities)
        print(f"There are {self.means.shape[0]} Gaussians in the GSplat model")

    def load_gsplat_from_json(self, gsplat_location) -> None:
        """:51-89: keys means, rotations, colors, opacities, scalings; ``colors`` are taken as given, opacities get
        a sigmoid, scalings an exp."""
        from . import io
        d = io.load_json(gsplat_location)
        dev = torch.device(self.device)
        f32 = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev)
        self.means, self.rots, self.colors = f32(d["means"]), f32(d["rotations"]), f32(d["colors"])
        self.opacities = torch.sigmoid(f32(d["opacities"]))
        self.scales = torch.exp(f32(d["scalings"]))
        self.covs_inv = compute_cov(self.rots, 1.0 / self.scales)
        self.covs = compute_cov(self.rots, self.scales)

    @classmethod
    def from_arrays(cls, means, quats, log_scales, features_dc, opacity_logits, device="cpu") -> "GSplatLoader":
        """Raw (pre-activation) splatfacto parameters already in memory."""
        self = cls.__new__(cls)
        self.device = device
        self._activate(means, quats, log_scales, features_dc, opacity_logits)
        return self

    @classmethod
    def from_path(cls, path, device="cpu") -> "GSplatLoader":
        """Gaussians from whatever ``path`` names, without needing the run's dataset: the splatfacto ``config.yml``
        or its run directory (``nerfstudio_models/step-*.ckpt``, the latest is read), a checkpoint, or a scene
        ``.json`` / ``.npz``."""
        from pathlib import Path
        from . import io
        p = Path(path)
        if p.suffix == ".json":
            return cls(str(p), device)
        if p.suffix == ".npz":
            d = io.load_npz(p)
            return cls.from_arrays(d["means"], d["quats"], d["scales"], d["features_dc"], d["opacities"], device)
        if p.suffix != ".ckpt":
            run = p.parent if p.is_file() or p.suffix in (".yml", ".yaml") else p
            ckpts = sorted((run / "nerfstudio_models").glob("step-*.ckpt"))
            if not ckpts:
                raise FileNotFoundError(f"no nerfstudio_models/step-*.ckpt next to {path}")
            p = ckpts[-1]
        g = io.load_splatfacto_ckpt(p)
        return cls.from_arrays(g["means"], g["quats"], g["scales"], g["features_dc"], g["opacities"], device)

    @classmethod
    def from_json(cls, path, device="cpu") -> "GSplatLoader":
        return cls(str(path), device)
