#!/usr/bin/env python3
"""Door A on the MI355X rasterizer: `GaussianSplat.render(pose)` and `generate_RGBD_point_cloud(pose)`
as the reference calls them (sim_a_splat/ns_utils/nerfstudio_utils.py:123-177, :375-472), on a
synthetic splatfacto-shaped model (raw log-scales, opacity logits, SH features) instead of a
checkpoint loaded through nerfstudio's eval_setup (both checkpoints upstream are Git-LFS pointers).

    python examples/demo_render_pose.py [--n 300000] [--frames 100] [--save rgb.npy]
"""
from __future__ import annotations

import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from sim_a_splat_amd.gaussian_splat import GaussianSplat, PinholeCamera, SplatModel  # noqa: E402
from sim_a_splat_amd.synthetic import c2w_opengl_from_viewmat, make_scene, ring_camera  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=300_000)
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--save", type=str, default="")
    a = ap.parse_args()

    sc = make_scene(a.n, seed=2)
    model = SplatModel(sc.means, np.log(sc.scales), sc.quats, sc.sh[:, 0], sc.sh[:, 1:],
                       np.log(sc.opacities / (1 - sc.opacities)).reshape(-1, 1), sh_degree=3, device="cuda:0")
    cam = ring_camera(640, 480, 525.0, yaw_deg=0.0)
    K = cam.K
    gs = GaussianSplat.from_model(model, PinholeCamera(torch.eye(4)[None, :3], float(K[0, 0]), float(K[1, 1]), float(K[0, 2]),
                                            float(K[1, 2]), cam.width, cam.height))
    H, W, _ = gs.get_camera_intrinsics()
    poses = [torch.from_numpy(c2w_opengl_from_viewmat(ring_camera(W, H, 525.0, yaw_deg=3.6 * i).viewmat)) for i in range(a.frames)]
    out = gs.render(poses[0])                                       # builds the scene on the GPU
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for p in poses:
        out = gs.render(p)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rgb, pts, _, mask, _ = gs.generate_RGBD_point_cloud(poses[0], max_depth=2.8)
    print(f"{a.frames / dt:.0f} blocking renders/s at {W}x{H}, {a.n} Gaussians; outputs {sorted(out)}; "
          f"rgb {tuple(out['rgb'].shape)} in [{float(out['rgb'].min()):.3f}, {float(out['rgb'].max()):.3f}]; "
          f"RGB-D cloud: {int(mask.sum())} of {mask.numel()} pixels closer than 2.8, points {tuple(pts.shape)}")
    if a.save:
        np.save(a.save, out["rgb"].cpu().numpy())


if __name__ == "__main__":
    main()
