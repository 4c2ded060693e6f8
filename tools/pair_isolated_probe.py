"""k_project<3,2> ALONE on the GPU: blocking view pairs (one sas_render_batch call of two views per step, nothing else in flight),
for a rocprofv3 kernel trace:  rocprofv3 --kernel-trace --stats -d gpurun_out/pair_iso -o p --output-format csv -- python3 tools/pair_isolated_probe.py"""
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sim_a_splat_amd.rasterizer import Rasterizer
from sim_a_splat_amd.synthetic import NERFSTUDIO_EVAL_BACKGROUND as BG, make_scene, ring_camera
sc = make_scene(1_000_000, seed=3, log_scale_mean=float(np.log(0.006)))
cams = [ring_camera(1920, 1080, 1000.0, yaw_deg=180.0 * v) for v in range(2)]
Vs, Ks = np.stack([c.viewmat for c in cams]), np.stack([c.K for c in cams])
r = Rasterizer(0)
r.upload(sc.means, sc.opacities, sc.sh, quats=sc.quats, scales=sc.scales, sh_degree=3)
out = {"rgb": torch.empty((2, 1080, 1920, 3), device="cuda"), "rgb8": torch.empty((2, 1080, 1920, 3), dtype=torch.uint8, device="cuda")}
for i in range(34):
    r.render_batch(Vs, Ks, 1920, 1080, BG, want=("rgb", "rgb8"), out=out)   # blocking
print("done", r.stats())
