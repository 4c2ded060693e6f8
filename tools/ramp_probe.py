"""Does the first 10 ms after an idle period run slower?  Config-3 view pairs in chunks of 20 steps
(submit 20 async pair-steps, wait), chunk after chunk, from a cold (idle) GPU; then again after a 2 s pause."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sim_a_splat_amd.rasterizer import Rasterizer
from sim_a_splat_amd.synthetic import NERFSTUDIO_EVAL_BACKGROUND as BG, make_scene, ring_camera
sc = make_scene(1_000_000, seed=3, log_scale_mean=float(np.log(0.006)))
cams = [ring_camera(1920, 1080, 1000.0, yaw_deg=180.0 * v) for v in range(2)]
r = Rasterizer(0)
r.upload(sc.means, sc.opacities, sc.sh, quats=sc.quats, scales=sc.scales, sh_degree=3)
Vs, Ks = np.stack([c.viewmat for c in cams]), np.stack([c.K for c in cams])
bufs = [{"rgb": torch.zeros((2, 1080, 1920, 3), device="cuda:0"), "rgb8": torch.zeros((2, 1080, 1920, 3), dtype=torch.uint8, device="cuda:0")} for _ in range(4)]
def chunk(k=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(k):
        r.render_batch(Vs, Ks, 1920, 1080, BG, want=("rgb", "rgb8"), out=bufs[i % 4], block=False)
    r.wait(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3
for rep in range(2):
    time.sleep(2.0)
    print(f"after a 2 s pause: ms/step per chunk of 20 steps:", " ".join(f"{chunk():.3f}" for _ in range(12)), flush=True)
time.sleep(2.0)
print("chunks of 5 after a pause:", " ".join(f"{chunk(5):.3f}" for _ in range(12)), flush=True)
