"""Where the quad layout (8-pixel tiles) stops paying: blocking and pipelined frames/s of one view at growing image sizes,
both layouts forced (SAS_QUAD is read per context), on the 292 247-Gaussian stand-in scene."""
import os, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sim_a_splat_amd.rasterizer import Rasterizer
from sim_a_splat_amd.synthetic import NERFSTUDIO_EVAL_BACKGROUND as BG, make_scene, ring_camera
n = int(sys.argv[1]) if len(sys.argv) > 1 else 292_247
sc = make_scene(n, seed=2)
sizes = [(320, 240), (400, 304), (480, 368), (560, 416), (640, 480), (800, 608)]
for quad in ("0", "1"):
    os.environ["SAS_QUAD"] = quad
    r = Rasterizer(0)
    r.upload(sc.means, sc.opacities, sc.sh, quats=sc.quats, scales=sc.scales, sh_degree=3)
    row = []
    for W, H in sizes:
        cam = ring_camera(W, H, 0.82 * W)
        outs = [{"rgb8": torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0")} for _ in range(4)]
        for i in range(30): r.render(cam.viewmat, cam.K, W, H, BG, want=("rgb8",), out=outs[0])
        t0 = time.perf_counter()
        for i in range(200): r.render(cam.viewmat, cam.K, W, H, BG, want=("rgb8",), out=outs[0])
        blocking = 200 / (time.perf_counter() - t0)
        for i in range(30): r.render(cam.viewmat, cam.K, W, H, BG, want=("rgb8",), out=outs[i % 4], block=False)
        r.wait(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(400): r.render(cam.viewmat, cam.K, W, H, BG, want=("rgb8",), out=outs[i % 4], block=False)
        r.wait(); torch.cuda.synchronize()
        row.append((W, H, ((W + 15) // 16) * ((H + 15) // 16), blocking, 400 / (time.perf_counter() - t0)))
    print(f"SAS_QUAD={quad}: " + "  ".join(f"{W}x{H} ({t} tiles): {b:.0f} blocking / {a:.0f} pipelined" for W, H, t, b, a in row), flush=True)
    r.close()
