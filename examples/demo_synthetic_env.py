#!/usr/bin/env python3
"""Gym-style observation loop on the MI355X rasterizer, shaped like the reference's
examples/demo_pusht_splat.py:54-78,167-169 -- camera dictionary (one viewport camera, one camera
riding on a link), a Drake-style draw message per step, `camera_i` uint8 CHW observations --
without Drake, viser or a browser.  The scene is the 113,831-Gaussian synthetic stand-in for
`robots-scene-v2` with 7 link groups (the trained checkpoint is a Git-LFS pointer upstream).

    python examples/demo_synthetic_env.py [--steps 200] [--save frame.npy]
"""
from __future__ import annotations

import argparse
import sys
import time
import types
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from sim_a_splat_amd.covariance import GSplatLoader  # noqa: E402
from sim_a_splat_amd.handler import CameraRig, SplatHandler  # noqa: E402
from sim_a_splat_amd.poses import matrix_to_quat_wxyz  # noqa: E402
from sim_a_splat_amd.synthetic import make_scene  # noqa: E402


def rot_z(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--save", type=str, default="")
    a = ap.parse_args()

    n_links = 7
    sc = make_scene(113_831, seed=2, n_groups=n_links + 1)            # group 0 = static scene
    # what GSplatLoader hands to the handler (splat_utils.py:24-49): activated, DC colour, covariances
    L = GSplatLoader(sc.means, sc.quats, np.log(sc.scales), sc.sh[:, 0], np.log(sc.opacities / (1 - sc.opacities)))
    masks = {f"link{i}": sc.group_id == i + 1 for i in range(n_links)}
    icp = np.eye(4)                                                    # masks were made in the splat frame itself
    fk = [np.eye(4) for _ in range(n_links)]                           # link frames at mask time
    handler = SplatHandler(L.means.numpy(), L.covs.numpy(), np.clip(L.colors.numpy(), 0, 1), L.opacities.numpy(),
                           masks, icp, fk, device=0)
    rig = CameraRig({
        0: {"link_name": "world", "local_frame": ((0.0, 1.0, 0.0, 0.0), (0.0, 0.0, 3.0)), "type": "viewport",
            "render_size": [240, 320]},
        1: {"link_name": "link6", "local_frame": ((0.0, 1.0, 0.0, 0.0), (0.0, 0.2, 2.5)), "type": "moving",
            "render_size": [240, 320]},
    })

    def draw_msg(t):
        """lcmt_viewer_draw-shaped message: the links swing about z, like joint motion would move them."""
        q, p = [], []
        for i in range(n_links):
            R = rot_z(0.3 * np.sin(0.05 * t + i))
            q.append(matrix_to_quat_wxyz(R).tolist())
            p.append([0.02 * np.sin(0.03 * t + i), 0.0, 0.0])
        return types.SimpleNamespace(num_links=n_links, robot_num=[3] * n_links, quaternion=q, position=p,
                                     link_name=[f"plant::link{i}" for i in range(n_links)])

    obs = None
    for t in range(10):                                                # warm-up
        msg = draw_msg(t)
        handler.draw_handler(msg)
        obs = rig.get_obs(handler, msg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(a.steps):
        msg = draw_msg(t)
        handler.draw_handler(msg)                                      # env.step -> link poses
        obs = rig.get_obs(handler, msg)                                # {"camera_0": u8[3,240,320], "camera_1": ...}
    dt = time.perf_counter() - t0
    print(f"{a.steps / dt:.0f} env steps/s, {2 * a.steps / dt:.0f} frames/s; obs keys {list(obs)}, "
          f"shape {obs['camera_0'].shape} {obs['camera_0'].dtype}, mean pixel {obs['camera_1'].mean():.1f}")
    if a.save:
        np.save(a.save, obs["camera_1"])
    handler.scene.close()


if __name__ == "__main__":
    main()
