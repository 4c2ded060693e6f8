#!/usr/bin/env python3
"""Gym-style observation loop on the MI355X rasterizer, shaped like the reference's
examples/demo_pusht_splat.py:54-78,167-169 -- `SplatEnvWrapper` over an inner env, the camera dictionary
(one viewport camera, one camera riding on a link, SE3 local frames), a Drake-style draw message per step,
`camera_i` uint8 CHW observations -- without Drake, viser or a browser.  The scene is the 113,831-Gaussian
synthetic stand-in for `robots-scene-v2` with 7 link groups (the trained checkpoint is a Git-LFS pointer
upstream); the inner env is a stand-in that only produces the draw messages joint motion would.

    python examples/demo_synthetic_env.py [--steps 200] [--save frame.npy]
"""
from __future__ import annotations

import argparse
import gc
import sys
import time
import types
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from sim_a_splat_amd.covariance import GSplatLoader  # noqa: E402
from sim_a_splat_amd.env_wrapper import SplatEnvWrapper  # noqa: E402
from sim_a_splat_amd.handler import SplatHandler  # noqa: E402
from sim_a_splat_amd.poses import SE3  # noqa: E402
from sim_a_splat_amd.synthetic import make_scene  # noqa: E402

N_LINKS = 7


class SwingingArmEnv:
    """The members of ManipulatorSimEnv that SplatEnvWrapper touches; the links swing about z."""
    visualize_robot_flag = False

    def __init__(self):
        self.t = 0
        self._names = [f"plant::link{i}" for i in range(N_LINKS)]
        self._cycle = {}     # draw messages by time step (the stand-in is not what this demo measures)

    def reset(self, seed=None, reset_to_state=None):
        self.t = 0

    def step(self, action):
        self.t += 1
        return {}, 0.0, False, False, {}

    def render(self):
        pass

    def _get_obs(self):
        return {"robot_pos": np.zeros(N_LINKS - 1)}

    def _generate_draw_msg(self):
        """lcmt_viewer_draw-shaped message (num_links, robot_num, link_name, quaternion wxyz, position)."""
        key = self.t % 512                                            # a 512-step cycle of the swing
        msg = self._cycle.get(key)
        if msg is None:
            msg = self._cycle[key] = self._make_msg(key)
        return msg

    def _make_msg(self, t):
        i = np.arange(N_LINKS)
        half = 0.15 * np.sin(0.05 * t + i)                           # rotation by 2 * half about z: (cos, 0, 0, sin)
        q = np.stack([np.cos(half), np.zeros(N_LINKS), np.zeros(N_LINKS), np.sin(half)], axis=1)
        p = np.stack([0.02 * np.sin(0.03 * t + i), np.zeros(N_LINKS), np.zeros(N_LINKS)], axis=1)
        return types.SimpleNamespace(num_links=N_LINKS, robot_num=[3] * N_LINKS, quaternion=q.tolist(), position=p.tolist(),
                                     link_name=self._names)

    def close(self):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--save", type=str, default="")
    ap.add_argument("--no-gc-freeze", action="store_true", help="leave Python's collector as it is (one ~40 ms pause inside the loop)")
    a = ap.parse_args()

    sc = make_scene(113_831, seed=2, n_groups=N_LINKS + 1)            # group 0 = static scene
    # what GSplatLoader hands to the handler (splat_utils.py:24-49): activated, DC colour, covariances
    L = GSplatLoader.from_arrays(sc.means, sc.quats, np.log(sc.scales), sc.sh[:, 0], np.log(sc.opacities / (1 - sc.opacities)))
    masks = {f"link{i}": sc.group_id == i + 1 for i in range(N_LINKS)}
    handler = SplatHandler.from_arrays(L.means.numpy(), L.covs.numpy(), np.clip(L.colors.numpy(), 0, 1), L.opacities.numpy(),
                                       masks, np.eye(4), [np.eye(4)] * N_LINKS, device=0)   # masks made in the splat frame itself
    env = SplatEnvWrapper(SwingingArmEnv(), splat_handler=handler)
    env._configure_cameras({
        0: {"link_name": "world", "local_frame": SE3(wxyz_xyz=np.array([0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 3.0])), "type": "viewport",
            "render_size": [240, 320]},
        1: {"link_name": "link6", "local_frame": SE3(wxyz_xyz=np.array([0.0, 1.0, 0.0, 0.0, 0.0, 0.2, 2.5])), "type": "moving",
            "render_size": [240, 320]},
    })
    env.reset()
    obs = None
    for _ in range(600):                                               # warm-up (and one cycle of the stand-in's messages)
        obs, *_ = env.step(None)
    torch.cuda.synchronize()
    if not a.no_gc_freeze:
        # One full collection over the objects torch's import leaves behind is a ~40 ms pause that falls, once, somewhere
        # around the 1000th step of a process (tools/slow_state_probe.py): 10 % of a 2000-step measurement, nothing of a
        # training run.  Freezing what exists after set-up keeps it out of the loop.
        gc.collect()
        gc.freeze()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        obs, reward, terminated, truncated, info = env.step(None)      # {"robot_pos", "camera_0": u8[3,240,320], "camera_1"}
    dt = time.perf_counter() - t0
    print(f"{a.steps / dt:.0f} env steps/s, {2 * a.steps / dt:.0f} frames/s; obs keys {list(obs)}, "
          f"shape {obs['camera_0'].shape} {obs['camera_0'].dtype}, mean pixel {obs['camera_1'].mean():.1f}")
    if a.save:
        np.save(a.save, obs["camera_1"])
    env.close()


if __name__ == "__main__":
    main()
