"""Blocking Gym-camera steps for a few seconds: per 100 steps the wall time per step and the tile kernel's own duration
(HIP events on its launch).  Separates a slow GPU (clock governor) from a slow host turn-around."""
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
poses = [random_group_poses(8, seed=s) for s in range(64)]
dev = torch.zeros((2, 240, 320, 3), dtype=torch.uint8, device="cuda:0")
pause = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0      # host work between steps, us (a physics step)
if "--freeze" in sys.argv:   # a full collection over torch's objects is a ~40 ms pause, once, around the 1000th step
    gc.collect(); gc.freeze()
rows = []
for blk in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    r.stage_time_means(reset=True)
    t0 = time.perf_counter()
    for s in range(100):
        r.set_group_poses(poses[s % 64])
        r.render_batch(V, K, 320, 240, BG, want=("rgb8",), out={"rgb8": dev}, time_tiles=True)
        if pause:
            t1 = time.perf_counter()
            while (time.perf_counter() - t1) * 1e6 < pause:
                pass
    dt = (time.perf_counter() - t0) / 100 * 1e6 - pause
    m, n = r.stage_time_means(reset=True)
    rows.append((dt, m["blend"] * 1e3))
print(f"pause {pause:.0f} us between steps; per block of 100 steps: step us (without the pause) / tile kernel us")
print(" ".join(f"{a:.0f}/{b:.0f}" for a, b in rows))
