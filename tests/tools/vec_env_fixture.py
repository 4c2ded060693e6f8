"""Shared by the SplatVecEnv tests (CPU over a stand-in renderer, GPU over the HIP rasterizer): E deterministic
stand-in envs (the members of ManipulatorSimEnv that SplatEnvWrapper touches, splat_env_wrapper.py:81-135), a
handler on a synthetic grouped scene, and the reference's camera dictionary with one moving and one viewport camera."""
import types

import numpy as np

from sim_a_splat_amd import poses

N, K_LINKS, H, W = 600, 3, 48, 64


class FakeEnv:
    """Link poses follow (env id, step, action) deterministically; reward / info echo them."""

    def __init__(self, e):
        self.e, self.t, self.a, self.closed = e, 0, 0.0, False

    def reset(self, seed=None, reset_to_state=None):
        self.t, self.a, self.seed = 0, 0.0, seed

    def step(self, action):
        self.t += 1
        self.a = float(action)
        return {"inner": self.t}, 10.0 * self.e + self.a, False, self.t >= 1000, {"e": self.e, "t": self.t}

    def _get_obs(self):
        return {"robot_pos": np.array([self.e, self.t, self.a], np.float64)}

    def _generate_draw_msg(self):
        r = np.random.default_rng(1000 * self.e + 17 * self.t + int(round(100 * self.a)))
        q = r.normal(size=(K_LINKS, 4))
        p = 0.2 * r.normal(size=(K_LINKS, 3))
        return types.SimpleNamespace(num_links=K_LINKS, robot_num=[3] * K_LINKS, link_name=[f"plant::link{j}" for j in range(K_LINKS)],
                                     quaternion=q.tolist(), position=p.tolist())

    def close(self):
        self.closed = True


def scene_arrays(seed=3):
    rng = np.random.default_rng(seed)
    means = rng.uniform(-0.6, 0.6, size=(N, 3)).astype(np.float32)
    A = rng.normal(size=(N, 3, 3)) * 0.05
    covs = (A @ A.transpose(0, 2, 1) + 1e-4 * np.eye(3)).astype(np.float32)
    colors = rng.uniform(size=(N, 3)).astype(np.float32)
    opac = rng.uniform(0.2, 0.95, size=N).astype(np.float32)
    masks = {f"link{j}": (np.arange(N) % 5) == j for j in range(K_LINKS)}
    Ricp = poses.quat_wxyz_to_matrix(np.array([0.9, 0.1, -0.2, 0.3]))
    icp = np.eye(4)
    icp[:3, :3], icp[:3, 3] = 0.8 * Ricp, [0.05, -0.02, 0.1]
    fk = []
    for j in range(K_LINKS):
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = poses.quat_wxyz_to_matrix(rng.normal(size=4)), 0.1 * rng.normal(size=3)
        fk.append(T)
    return means, covs, colors, opac, masks, icp, fk


def camera_info():
    return {0: {"link_name": "world", "local_frame": poses.SE3(wxyz_xyz=np.array([0.0, 1.0, 0, 0, 0.0, 0.0, 2.2])), "type": "viewport",
                "render_size": [H, W]},
            1: {"link_name": "link1", "local_frame": poses.SE3(wxyz_xyz=np.array([0.0, 1.0, 0, 0, 0.02, 0.0, 2.5])), "type": "moving",
                "render_size": [H, W]}}
