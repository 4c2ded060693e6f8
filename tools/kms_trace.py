#!/usr/bin/env python3
"""Per-frame tile-kernel durations (SAS_TIME_TILES events) through back-to-back passes of the bench's step:
which frames of a pass carry the long durations."""
import sys, time
import numpy as np, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sim_a_splat_amd.rasterizer import Rasterizer
from sim_a_splat_amd.synthetic import NERFSTUDIO_EVAL_BACKGROUND as BG, make_scene, ring_camera
sc = make_scene(1_000_000, seed=3, log_scale_mean=float(np.log(0.006)))
cams = [ring_camera(1920, 1080, 1000.0, yaw_deg=180.0 * v) for v in range(2)]
r = Rasterizer(0)
r.upload(sc.means, sc.opacities, sc.sh, quats=sc.quats, scales=sc.scales, sh_degree=3)
Vs, Ks = np.stack([c.viewmat for c in cams]), np.stack([c.K for c in cams])
W, H = 1920, 1080
dev = torch.device("cuda:0")
bufs = [{"rgb": torch.zeros((2, H, W, 3), device=dev), "rgb8": torch.zeros((2, H, W, 3), dtype=torch.uint8, device=dev)} for _ in range(4)]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for p in range(5):
    seq = []
    done0 = r.frames_completed()[1]
    t0 = time.perf_counter()
    for i in range(steps):
        r.render_batch(Vs, Ks, W, H, BG, want=("rgb", "rgb8"), out=bufs[i % 4], block=False, time_tiles=True)
        d = r.frames_completed()[1]
        if d != done0:
            seq.append(round(r.stage_times()["blend"], 3)); done0 = d
    r.wait(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    m, n = r.stage_time_means(reset=True)
    print(f"pass {p}: {2 * steps / dt:.0f} frames/s, mean tile ms {m['blend']:.3f} over {n}; seen at completion: {seq}")
