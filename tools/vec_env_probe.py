"""Vectorised Gym envs: E envs x 2 cameras of 240x320 per step in ONE sas_render_batch_host_posed call -- every env
has its OWN link poses each step (E pose sets; splat_env_wrapper.py:121-159 poses the scene per env), uint8 frames
to the host.  Beside it: the same frames with one pose set for all envs (what round 2 measured), and a check of one
step against per-env blocking renders.  Usage: python tools/vec_env_probe.py [E ...]"""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sim_a_splat_amd.rasterizer import Rasterizer
from sim_a_splat_amd.synthetic import NERFSTUDIO_EVAL_BACKGROUND as BG, make_scene, random_group_poses, ring_camera

G = 8
sc = make_scene(113_831, seed=2, n_groups=G)
r = Rasterizer(0)
r.upload(sc.means, sc.opacities, sc.sh, quats=sc.quats, scales=sc.scales, sh_degree=3, group_id=sc.group_id, n_groups=G)
for E in [int(a) for a in sys.argv[1:]] or [1, 4, 16]:
    cams = [ring_camera(320, 240, 262.0, yaw_deg=(360.0 * i) / (2 * E), elev=0.5 * (i & 1)) for i in range(2 * E)]
    V = np.stack([c.viewmat for c in cams]); K = np.stack([c.K for c in cams])
    idx = [v // 2 for v in range(2 * E)]
    rollouts = [np.stack([random_group_poses(G, seed=1000 * s + e) for e in range(E)]) for s in range(16)]   # [E,G,12] per step
    # one step checked: the batch's frames of env e == blocking renders with env e's poses
    out = r.render_batch_host(V, K, 320, 240, BG, pose_sets=rollouts[0], pose_set=idx).clone()
    for e in range(E):
        r.set_group_poses(rollouts[0][e])
        for v in (2 * e, 2 * e + 1):
            one = r.render_batch_host(V[v:v + 1], K[v:v + 1], 320, 240, BG)
            assert torch.equal(one[0], out[v]), (e, v)
    res = {}
    for mode in ("per-env poses", "one pose set"):
        for it in range(2):
            steps = 10 if it == 0 else max(40, 400 // E)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for s in range(steps):
                if mode == "per-env poses":
                    out = r.render_batch_host(V, K, 320, 240, BG, pose_sets=rollouts[s % 16], pose_set=idx)
                else:
                    r.set_group_poses(rollouts[s % 16][0])
                    out = r.render_batch_host(V, K, 320, 240, BG)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res[mode] = steps / dt
    print(f"{E} envs x 2 cameras of 240x320, uint8 to host: " + "; ".join(f"{k}: {v:.0f} steps/s = {2 * E * v:.0f} frames/s" for k, v in res.items()), flush=True)
