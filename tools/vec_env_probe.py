"""Vectorised Gym envs: E envs x 2 cameras of 240x320 per step in ONE sas_render_batch call (same scene, one
pose set per step), uint8 frames to the host.  Usage: python tools/vec_env_probe.py [E ...]"""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sim_a_splat_amd.rasterizer import Rasterizer
from sim_a_splat_amd.synthetic import NERFSTUDIO_EVAL_BACKGROUND as BG, make_scene, random_group_poses, ring_camera

sc = make_scene(113_831, seed=2, n_groups=8)
r = Rasterizer(0)
r.upload(sc.means, sc.opacities, sc.sh, quats=sc.quats, scales=sc.scales, sh_degree=3, group_id=sc.group_id, n_groups=8)
for E in [int(a) for a in sys.argv[1:]] or [1, 4, 16]:
    cams = [ring_camera(320, 240, 262.0, yaw_deg=(360.0 * i) / (2 * E), elev=0.5 * (i & 1)) for i in range(2 * E)]
    V = np.stack([c.viewmat for c in cams]); K = np.stack([c.K for c in cams])
    for it in range(2):
        steps = 10 if it == 0 else max(40, 400 // E)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for s in range(steps):
            r.set_group_poses(random_group_poses(8, seed=s))
            out = r.render_batch_host(V, K, 320, 240, BG)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{E} envs: {steps/dt:.0f} steps/s = {2*E*steps/dt:.0f} frames/s ({2*E} cameras of 240x320 per step, uint8 to host)")
