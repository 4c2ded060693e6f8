"""Where an env step of the Door-B loop spends its host time (113,831 Gaussians, 8 groups, 2 cameras 240x320)."""
import gc, sys, time
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
poses = [random_group_poses(8, seed=s) for s in range(300)]
host = torch.empty((2, 240, 320, 3), dtype=torch.uint8).pin_memory()
acc = {"poses": 0.0, "render": 0.0, "copy": 0.0}
gc.collect(); gc.freeze()   # a full collection over torch's objects pauses ~40 ms, once, around the 1000th step (tools/slow_state_probe.py)
for it in range(2):
    acc = {k: 0.0 for k in acc}
    t_all = time.perf_counter()
    for s in range(300):
        t0 = time.perf_counter(); r.set_group_poses(poses[s]); t1 = time.perf_counter()
        out = r.render_batch(V, K, 320, 240, BG, want=("rgb8",))["rgb8"]; t2 = time.perf_counter()
        host.copy_(out); t3 = time.perf_counter()
        acc["poses"] += t1 - t0; acc["render"] += t2 - t1; acc["copy"] += t3 - t2
    total = time.perf_counter() - t_all
print({k: round(v / 300 * 1e6, 1) for k, v in acc.items()}, "us per step; total", round(total / 300 * 1e6, 1), "us ->", round(300 / total), "env steps/s")
# the same step with the frames delivered to pinned host memory by the library itself (sas_render_batch_host)
for it in range(2):
    t_all = time.perf_counter()
    for s in range(300):
        r.set_group_poses(poses[s])
        r.render_batch_host(V, K, 320, 240, BG, out=host)
    total_h = time.perf_counter() - t_all
print("frames to the host inside the call:", round(total_h / 300 * 1e6, 1), "us per step ->", round(300 / total_h), "env steps/s")
for k in range(3):
    out = r.render_batch(V, K, 320, 240, BG, want=("rgb8",), block=True)
    r.render(cams[0].viewmat, cams[0].K, 320, 240, BG, want=("rgb8",), timing=True)
print("isolated frame stage ms:", {k: round(v, 4) for k, v in r.stage_times().items()})
