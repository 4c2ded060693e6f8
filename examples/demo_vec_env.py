#!/usr/bin/env python3
"""Vectorised rollouts on the MI355X rasterizer: E copies of the reference's Gym loop
(sim_a_splat/env/splat/splat_env_wrapper.py:121-159, cameras as in examples/demo_pusht_splat.py:54-78) stepped as ONE
batched render per rank through ``SplatVecEnv``; with several ranks (one per GPU) the envs are sharded over them and
rank 0 receives every env's uint8 ``camera_i`` observations (RCCL gather on a multi-GPU node).

    python examples/demo_vec_env.py [--envs 8] [--steps 500]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 examples/demo_vec_env.py --envs 8

The scene is the 113,831-Gaussian stand-in for `robots-scene-v2` with 7 link groups; the inner envs are stand-ins that
only produce the draw messages joint motion would (each env swings its arm with its own phase).
"""
from __future__ import annotations

import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))

from demo_synthetic_env import N_LINKS, SwingingArmEnv  # noqa: E402
from sim_a_splat_amd import distributed as D  # noqa: E402
from sim_a_splat_amd.covariance import GSplatLoader  # noqa: E402
from sim_a_splat_amd.handler import SplatHandler  # noqa: E402
from sim_a_splat_amd.poses import SE3  # noqa: E402
from sim_a_splat_amd.synthetic import make_scene  # noqa: E402
from sim_a_splat_amd.vec_env import SplatVecEnv  # noqa: E402


class PhasedArmEnv(SwingingArmEnv):
    """Env e's arm runs e * 37 steps ahead: every env shows a different pose each step."""

    def __init__(self, e):
        super().__init__()
        self.phase = 37 * e

    def _generate_draw_msg(self):
        self.t += self.phase
        try:
            return super()._generate_draw_msg()
        finally:
            self.t -= self.phase


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=8)
    ap.add_argument("--steps", type=int, default=500)
    a = ap.parse_args()
    rank, world, local = D.init_from_env()                                # one process per GPU under a launcher; else a single rank
    dev = int(local) if torch.cuda.device_count() > local else 0
    sc = make_scene(113_831, seed=2, n_groups=N_LINKS + 1)
    L = GSplatLoader.from_arrays(sc.means, sc.quats, np.log(sc.scales), sc.sh[:, 0], np.log(sc.opacities / (1 - sc.opacities)))
    masks = {f"link{i}": sc.group_id == i + 1 for i in range(N_LINKS)}
    handler = SplatHandler.from_arrays(L.means.numpy(), L.covs.numpy(), np.clip(L.colors.numpy(), 0, 1), L.opacities.numpy(),
                                       masks, np.eye(4), [np.eye(4)] * N_LINKS, device=dev)   # the scene is replicated per rank
    cameras = {
        0: {"link_name": "world", "local_frame": SE3(wxyz_xyz=np.array([0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 3.0])), "type": "viewport", "render_size": [240, 320]},
        1: {"link_name": "link6", "local_frame": SE3(wxyz_xyz=np.array([0.0, 1.0, 0.0, 0.0, 0.0, 0.2, 2.5])), "type": "moving", "render_size": [240, 320]},
    }
    envs = [PhasedArmEnv(e) if e % world == rank else None for e in range(a.envs)]
    venv = SplatVecEnv(envs, handler, cameras, rank=rank, world=world)
    obs = venv.reset()
    for _ in range(50):
        obs, *_ = venv.step([None] * a.envs)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        obs, reward, terminated, truncated, info = venv.step([None] * a.envs)   # rank 0: every env's obs
    dt = time.perf_counter() - t0
    if rank == 0:
        got = [o for o in obs if o is not None]
        print(f"{world} rank(s), {a.envs} envs x 2 cameras of 240x320: {a.steps / dt:.0f} vectorised steps/s = {2 * a.envs * a.steps / dt:.0f} frames/s; "
              f"rank 0 holds {len(got)} observations, camera_0 {got[0]['camera_0'].shape} {got[0]['camera_0'].dtype}, "
              f"mean pixel of env 0 / env {a.envs - 1}: {got[0]['camera_1'].mean():.1f} / {got[-1]['camera_1'].mean():.1f}")
    venv.close()


if __name__ == "__main__":
    main()
