"""Frames/s of view pairs (render_batch, two views per projection pass) against single views."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sim_a_splat_amd.rasterizer import Rasterizer
from sim_a_splat_amd.synthetic import NERFSTUDIO_EVAL_BACKGROUND as BG, config_scene_and_cameras, ring_camera
for cfg in [int(a) for a in sys.argv[1:]] or [3]:
    sc, cams = config_scene_and_cameras(cfg)
    cam = cams[0]
    cam2 = ring_camera(cam.width, cam.height, float(cam.K[0, 0]), yaw_deg=20.0)
    r = Rasterizer(0)
    r.upload(sc.means, sc.opacities, sc.sh, quats=sc.quats, scales=sc.scales, sh_degree=3)
    H, W = cam.height, cam.width
    Vs = np.stack([cam.viewmat, cam2.viewmat]); Ks = np.stack([cam.K, cam2.K])
    # correctness: the pair equals the two single renders, bit for bit
    a = r.render(cam.viewmat, cam.K, W, H, BG, want=("rgb",))["rgb"].clone()
    b = r.render(cam2.viewmat, cam2.K, W, H, BG, want=("rgb",))["rgb"].clone()
    pr = r.render_batch(Vs, Ks, W, H, BG, want=("rgb",))["rgb"]
    same = bool(torch.equal(pr[0], a) and torch.equal(pr[1], b))
    bufs = [{"rgb": torch.empty((2, H, W, 3), dtype=torch.float32, device="cuda:0")} for _ in range(3)]
    res = {}
    for mode in ("single", "pair", "single", "pair"):
        for K in (20, 200):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(K):
                if mode == "pair":
                    r.render_batch(Vs, Ks, W, H, BG, want=("rgb",), out=bufs[i % 3], block=False)
                else:
                    for v in range(2):
                        r.render(Vs[v], Ks[v], W, H, BG, want=("rgb",), out={"rgb": bufs[i % 3]["rgb"][v]}, block=False)
            r.wait(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res.setdefault(mode, []).append(2 * K / dt)
    print(f"cfg{cfg}: single {np.mean(res['single']):.0f} frames/s, paired {np.mean(res['pair']):.0f} frames/s; pair == singles: {same}", flush=True)
    r.close()
