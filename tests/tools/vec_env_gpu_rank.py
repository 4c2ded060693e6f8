"""One rank of tests/test_gpu_b_vec_env.py: SplatVecEnv on the HIP rasterizer, ranks sharing the one card of the GPU box,
torch.distributed over gloo (RCCL needs a GPU per rank; its arm runs in tests/test_gpu_a_nccl_world1.py).  Rank 0 checks
EVERY env's gathered camera observations of every step bit for bit against the oracle rendered with that env's link
poses (oracle/ref_math's NumPy restatement of splat_handler.py:265-288) and camera poses (:316-332).  Prints one JSON line."""
import json
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import oracle  # noqa: E402
from oracle import ref_math  # noqa: E402
import vec_env_fixture as fx  # noqa: E402
from sim_a_splat_amd import distributed as D, poses  # noqa: E402
from sim_a_splat_amd.handler import SplatHandler  # noqa: E402
from sim_a_splat_amd.vec_env import SplatVecEnv  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rank, world, _ = D.init_from_env(backend="gloo")
means, covs, colors, opac, masks, icp, fk = fx.scene_arrays()
h = SplatHandler.from_arrays(means, covs, colors, opac, masks, icp, fk, device=0)
envs = [fx.FakeEnv(e) if e % world == rank else None for e in range(E)]
venv = SplatVecEnv(envs, h, fx.camera_info(), rank=rank, world=world)

# the oracle's view of the scene: Gaussians in group order (link0.., rest), 3x3 covariances as cov6, final RGB
gid = np.full(fx.N, fx.K_LINKS)
for j in range(fx.K_LINKS):
    gid[masks[f"link{j}"]] = j
order = np.concatenate([np.nonzero(gid == g)[0] for g in range(fx.K_LINKS + 1)])
group_of = np.concatenate([np.full(int((gid == g).sum()), g, np.uint8) for g in range(fx.K_LINKS + 1)])
cov6 = np.stack([covs[:, 0, 0], covs[:, 0, 1], covs[:, 0, 2], covs[:, 1, 1], covs[:, 1, 2], covs[:, 2, 2]], 1)[order]
s_icp, Ri, ti = poses.decompose_icp(icp)


def oracle_obs(e, t, a):
    env = fx.FakeEnv(e)
    env.t, env.a = t, a
    msg = env._generate_draw_msg()
    Rt = []
    for j in range(fx.K_LINKS):
        R, tt = ref_math.link_splat_pose(Ri, ti, s_icp, fk[j][:3, :3], fk[j][:3, 3], msg.quaternion[j], msg.position[j])
        Rt.append(poses.rt_to_row12(poses.quat_wxyz_to_matrix(poses.matrix_to_quat_wxyz(R)), tt))
    Rt.append(poses.rt_to_row12(np.eye(3), np.zeros(3)))
    out = []
    info = fx.camera_info()
    cams = []
    Rm, tm = poses.attached_frame(s_icp, Ri, ti, msg.quaternion[1], msg.position[1], poses.pose_wxyz_xyz(info[1]["local_frame"])[1])
    cams.append((poses.matrix_to_quat_wxyz(Rm), tm))                    # moving cameras first
    cams.append(poses.pose_wxyz_xyz(info[0]["local_frame"]))
    for wxyz, pos in cams:
        V, K = h.scene._view_and_K(fx.H, fx.W, wxyz, pos, h.scene.camera.fov)
        ref = oracle.render(means[order], opac[order], colors[order], V, K, fx.W, fx.H, cov6=cov6, sh_degree=-1, group_id=group_of,
                            group_Rt=np.stack(Rt), background=(0, 0, 0), want_rgb8=True)
        out.append((np.moveaxis(ref["rgb8"], -1, 0), ref["n_visible"]))
    return out


checked, visible, ok = 0, 0, True
owned = set(range(E)) if rank == 0 else set(D.shard_views(E, rank, world))


def check(obs, t, acts):
    global checked, visible, ok
    for e in range(E):
        if e not in owned:
            ok &= obs[e] is None
            continue
        want = oracle_obs(e, t, acts[e] if acts else 0.0)
        for c in range(2):
            ok &= bool(np.array_equal(obs[e][f"camera_{c}"], want[c][0]))
            visible = max(visible, want[c][1])
            checked += 1


check(venv.reset(seed=3), 0, None)
for t in range(1, 3):
    acts = [0.25 * t + 0.1 * e for e in range(E)]
    obs, rew, term, trunc, info = venv.step(acts)
    check(obs, t, acts)
tk = venv.step_async([1.5] * E)
check(venv.collect(tk)[0], 3, [1.5] * E)
print(json.dumps({"rank": rank, "world": world, "envs": E, "frames_checked": checked, "bit_equal_to_oracle": bool(ok), "max_visible": int(visible)}), flush=True)
import torch.distributed as dist  # noqa: E402
if dist.is_initialized():
    dist.barrier()
    dist.destroy_process_group()
venv.close()
