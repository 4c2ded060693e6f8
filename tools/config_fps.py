"""Frames/s of one view of a BASELINE config with two frames in flight (same loop as bench.py)."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sim_a_splat_amd.rasterizer import Rasterizer
from sim_a_splat_amd.synthetic import NERFSTUDIO_EVAL_BACKGROUND as BG, config_scene_and_cameras
for cfg in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 5]:
    sc, cams = config_scene_and_cameras(cfg)
    cam = cams[0]
    r = Rasterizer(0)
    r.upload(sc.means, sc.opacities, sc.sh, quats=sc.quats, scales=sc.scales, sh_degree=3)
    bufs = [{"rgb": torch.empty((cam.height, cam.width, 3), dtype=torch.float32, device="cuda:0")} for _ in range(3)]
    for K in (30, 300):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(K):
            r.render(cam.viewmat, cam.K, cam.width, cam.height, BG, want=("rgb",), out=bufs[i % 3], block=False)
        r.wait(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = r.stats()
    print(f"cfg{cfg}: N={sc.n} {cam.width}x{cam.height} M={st['n_isect']} max_list={st['max_tile_len']} -> {K/dt:.0f} frames/s ({1e3*dt/K:.3f} ms)", flush=True)
    r.close()
