"""Host-side cost of one async sas_render call (Python wrapper + C ABI enqueue). GPU box only."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sim_a_splat_amd.rasterizer import Rasterizer
from sim_a_splat_amd.synthetic import NERFSTUDIO_EVAL_BACKGROUND as BG, make_scene, ring_camera
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
sc = make_scene(n, seed=3, log_scale_mean=float(np.log(0.006)))
cam = ring_camera(1920, 1080, 1000.0)
r = Rasterizer(0)
r.upload(sc.means, sc.opacities, sc.sh, quats=sc.quats, scales=sc.scales, sh_degree=3)
bufs = [{"rgb": torch.empty((1080, 1920, 3), dtype=torch.float32, device="cuda:0")} for _ in range(3)]
for i in range(50):
    r.render(cam.viewmat, cam.K, 1920, 1080, BG, want=("rgb",), out=bufs[i % 3], block=False)
r.wait()
t0 = time.perf_counter()
K = 500
for i in range(K):
    r.render(cam.viewmat, cam.K, 1920, 1080, BG, want=("rgb",), out=bufs[i % 3], block=False)
t1 = time.perf_counter()
r.wait()
t2 = time.perf_counter()
print(f"n={n}: host enqueue {1e6*(t1-t0)/K:.1f} us/frame, end-to-end {1e6*(t2-t0)/K:.1f} us/frame")
