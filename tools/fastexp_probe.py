"""Frames/s and max abs difference of the v_exp_f32 path (SAS_FAST_EXP) against the contract path."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sim_a_splat_amd.rasterizer import Rasterizer
from sim_a_splat_amd.synthetic import NERFSTUDIO_EVAL_BACKGROUND as BG, config_scene_and_cameras
for cfg in [int(a) for a in sys.argv[1:]] or [3]:
    sc, cams = config_scene_and_cameras(cfg)
    cam = cams[0]
    r = Rasterizer(0)
    r.upload(sc.means, sc.opacities, sc.sh, quats=sc.quats, scales=sc.scales, sh_degree=3)
    ref = r.render(cam.viewmat, cam.K, cam.width, cam.height, BG, want=("rgb", "alpha", "depth", "rgb8"))
    fst = r.render(cam.viewmat, cam.K, cam.width, cam.height, BG, want=("rgb", "alpha", "depth", "rgb8"), fast_exp=True)
    d = {k: float((ref[k].float() - fst[k].float()).abs().max()) for k in ref}
    bufs = [{"rgb": torch.empty((cam.height, cam.width, 3), dtype=torch.float32, device="cuda:0")} for _ in range(3)]
    res = {}
    for fast in (False, True, False, True):
        for K in (30, 300):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(K):
                r.render(cam.viewmat, cam.K, cam.width, cam.height, BG, want=("rgb",), out=bufs[i % 3], block=False, fast_exp=fast)
            r.wait(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res.setdefault(fast, []).append(K / dt)
    print(f"cfg{cfg}: contract {np.mean(res[False]):.0f} fps, v_exp {np.mean(res[True]):.0f} fps; max abs diff {d}", flush=True)
    r.close()
