"""Throughput of the Gym-observation path (Door B shape): 113,831-Gaussian stand-in scene with 7 link
groups, two 240x320 cameras per env step (examples/demo_pusht_splat.py:54-78), uint8 frames."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sim_a_splat_amd.rasterizer import Rasterizer
from sim_a_splat_amd.synthetic import NERFSTUDIO_EVAL_BACKGROUND as BG, make_scene, random_group_poses, ring_camera

sc = make_scene(113_831, seed=2, n_groups=8)
r = Rasterizer(0)
r.upload(sc.means, sc.opacities, sc.sh, quats=sc.quats, scales=sc.scales, sh_degree=3, group_id=sc.group_id, n_groups=8)
cams = [ring_camera(320, 240, 262.0, yaw_deg=0.0), ring_camera(320, 240, 262.0, yaw_deg=60.0, elev=0.5)]
V = np.stack([c.viewmat for c in cams]); K = np.stack([c.K for c in cams])
for mode in ("batch", "loop"):
    for it in range(2):
        steps = 20 if it == 0 else 300
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for s in range(steps):
            r.set_group_poses(random_group_poses(8, seed=s))     # per-step link poses (draw_handler)
            if mode == "batch":
                out = r.render_batch_host(V, K, 320, 240, BG)      # pinned host tensor, filled inside the call
            else:
                out = [r.render(c.viewmat, c.K, 320, 240, BG, want=("rgb8",))["rgb8"].cpu() for c in cams]
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{mode}: {steps/dt:.0f} env steps/s = {2*steps/dt:.0f} frames/s (2 cameras 240x320, uint8 to host, pose update per step)")
