"""Deterministic synthetic scenes and cameras for the BASELINE.json configs.

Both trained checkpoints of the reference are Git-LFS pointers
(``assets/*/splatfacto/*/nerfstudio_models/step-000029999.ckpt``, 134 B), so every config runs
on a seeded stand-in whose size, SH degree and background match the real scenes
(``assets/robots-scene-v2/splatfacto/2024-12-06_150850/config.yml:150-193``).  Pure numpy: the
same arrays feed the HIP rasterizer, the CPU oracle (tests only) and the benchmark.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import numpy as np

# nerfstudio's eval-time constant for ``background_color: random`` (config.yml:152)
NERFSTUDIO_EVAL_BACKGROUND = (0.1490, 0.1647, 0.2157)


@dataclass
class SyntheticScene:
    """Activated Gaussian parameters in the layout nerfstudio hands to gsplat."""

    means: np.ndarray        # [N,3] f32
    quats: np.ndarray        # [N,4] f32 wxyz, un-normalised
    scales: np.ndarray       # [N,3] f32, exp() already applied
    opacities: np.ndarray    # [N]   f32, sigmoid() already applied
    sh: np.ndarray           # [N,16,3] f32 (features_dc ++ features_rest)
    sh_degree: int = 3
    group_id: Optional[np.ndarray] = None   # [N] u8
    meta: Dict[str, object] = field(default_factory=dict)

    @property
    def n(self) -> int:
        return int(self.means.shape[0])


@dataclass
class Camera:
    """Pinhole camera: world->camera ``viewmat`` (OpenCV axes, +z forward) and intrinsics."""

    viewmat: np.ndarray  # [4,4] f32
    K: np.ndarray        # [3,3] f32
    width: int
    height: int

    @property
    def tiles(self) -> int:
        return ((self.width + 15) // 16) * ((self.height + 15) // 16)


def make_scene(n: int, seed: int, log_scale_mean: float = float(np.log(0.01)), n_groups: int = 0) -> SyntheticScene:
    """Seeded scene with the distributions of SURVEY.md 8d.

    70 % of the means are uniform in [-1,1]^3, 30 % sit on a noisy shell of radius 0.6;
    log-scales ~ N(log_scale_mean, 0.6^2) clipped to [ln 1e-3, ln 0.2]; quaternions ~ N(0,I);
    opacity logits ~ N(1, 2^2); features_dc ~ N(0,1), features_rest ~ N(0, 0.1^2).
    """
    rng = np.random.default_rng(seed)
    n_shell = int(0.3 * n)
    n_box = n - n_shell
    box = rng.uniform(-1.0, 1.0, size=(n_box, 3))
    d = rng.normal(size=(n_shell, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True) + 1e-12
    shell = d * (0.6 + 0.02 * rng.normal(size=(n_shell, 1)))
    means = np.concatenate([box, shell], axis=0)
    perm = rng.permutation(n)
    means = means[perm]
    log_s = rng.normal(log_scale_mean, 0.6, size=(n, 3))
    log_s = np.clip(log_s, np.log(1e-3), np.log(0.2))
    quats = rng.normal(size=(n, 4))
    logit = rng.normal(1.0, 2.0, size=(n,))
    dc = rng.normal(0.0, 1.0, size=(n, 1, 3))
    rest = rng.normal(0.0, 0.1, size=(n, 15, 3))
    group_id = None
    if n_groups > 0:
        # contiguous spatial-ish groups: robot links are a minority of the scene (a7)
        group_id = np.zeros(n, dtype=np.uint8)
        link = rng.uniform(size=n) < 0.25
        group_id[link] = rng.integers(1, n_groups, size=int(link.sum()), dtype=np.uint8)
    return SyntheticScene(
        means=means.astype(np.float32),
        quats=quats.astype(np.float32),
        scales=np.exp(log_s).astype(np.float32),
        opacities=(1.0 / (1.0 + np.exp(-logit))).astype(np.float32),
        sh=np.concatenate([dc, rest], axis=1).astype(np.float32),
        sh_degree=3,
        group_id=group_id,
        meta={"seed": seed, "log_scale_mean": log_scale_mean},
    )


def look_at_viewmat(eye, target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0)) -> np.ndarray:
    """world->camera, OpenCV convention (x right, y down, z forward)."""
    eye = np.asarray(eye, dtype=np.float64)
    target = np.asarray(target, dtype=np.float64)
    fwd = target - eye
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, np.asarray(up, dtype=np.float64))
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd], axis=0)
    V = np.eye(4)
    V[:3, :3] = R
    V[:3, 3] = -R @ eye
    return V.astype(np.float32)


def c2w_opengl_from_viewmat(viewmat: np.ndarray) -> np.ndarray:
    """Inverse of the T0 conversion: the OpenGL camera-to-world nerfstudio poses use."""
    V = np.asarray(viewmat, dtype=np.float64)
    R = V[:3, :3]
    t = V[:3, 3]
    c2w = np.eye(4)
    c2w[:3, :3] = R.T @ np.diag([1.0, -1.0, -1.0])
    c2w[:3, 3] = -R.T @ t
    return c2w.astype(np.float32)


def intrinsics(fx: float, fy: float, cx: float, cy: float) -> np.ndarray:
    return np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=np.float32)


def ring_camera(width: int, height: int, f: float, yaw_deg: float = 0.0, radius: float = 3.0,
                elev: float = 0.0) -> Camera:
    yaw = np.deg2rad(yaw_deg)
    eye = (radius * np.sin(yaw), elev, radius * np.cos(yaw))
    return Camera(look_at_viewmat(eye), intrinsics(f, f, width / 2.0, height / 2.0), width, height)


# ---- the five BASELINE.json configs -------------------------------------------------------

def config_cameras(cfg: int) -> list:
    """Camera list of BASELINE config ``cfg`` (1..5)."""
    if cfg == 1:
        return [ring_camera(256, 256, 256.0)]
    if cfg == 2:
        return [ring_camera(640, 480, 525.0)]
    if cfg == 3:
        return [ring_camera(1920, 1080, 1000.0)]
    if cfg == 4:
        return [ring_camera(640, 480, 525.0, yaw_deg=45.0 * k) for k in range(8)]
    if cfg == 5:
        return [ring_camera(1920, 1080, 1000.0, yaw_deg=90.0 * k) for k in range(4)]
    raise ValueError(f"unknown BASELINE config {cfg}")


def config_scene_and_cameras(cfg: int, scale: float = 1.0) -> Tuple[SyntheticScene, list]:
    """Scene + camera list of BASELINE config ``cfg`` (1..5).  ``scale`` < 1 shrinks N for tests."""
    cams = config_cameras(cfg)
    if cfg == 1:
        sc = make_scene(max(1, int(10_000 * scale)), seed=1)
    elif cfg in (2, 4):
        sc = make_scene(max(1, int(292_247 * scale)), seed=2, n_groups=7)
    elif cfg == 3:
        sc = make_scene(max(1, int(1_000_000 * scale)), seed=3, log_scale_mean=float(np.log(0.006)))
    else:
        sc = make_scene(max(1, int(5_000_000 * scale)), seed=5, log_scale_mean=float(np.log(0.006)))
    sc.meta["config"] = cfg
    return sc, cams


def random_group_poses(n_groups: int, seed: int, max_angle: float = 0.3, max_shift: float = 0.1) -> np.ndarray:
    """[G,12] row-major (R|t); group 0 (static scene) stays identity like /scene_ohne_robot."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n_groups, 3, 4), dtype=np.float64)
    for g in range(n_groups):
        if g == 0:
            R, t = np.eye(3), np.zeros(3)
        else:
            axis = rng.normal(size=3)
            axis /= np.linalg.norm(axis)
            ang = rng.uniform(-max_angle, max_angle)
            Kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
            R = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * (Kx @ Kx)
            t = rng.uniform(-max_shift, max_shift, size=3)
        out[g, :, :3] = R
        out[g, :, 3] = t
    return out.reshape(n_groups, 12).astype(np.float32)
