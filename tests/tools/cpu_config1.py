"""BASELINE config 1 (10 000 Gaussians, 256x256: the reference's CPU-runnable case) on the CPU oracle -- one thread
and all threads (SURVEY.md 8d asks for both) -- and, when a HIP device is present, on the GPU.
This tool is measurement infrastructure like bench.py's cpu_baseline leg: the oracle is the thing timed, not shipped."""
import os, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import oracle
from sim_a_splat_amd.synthetic import NERFSTUDIO_EVAL_BACKGROUND as BG, config_scene_and_cameras

sc, cams = config_scene_and_cameras(1)
cam = cams[0]

def cpu_fps(threads, frames=20):
    oracle.set_num_threads(threads)
    oracle.render_scene(sc, cam, background=BG)
    ts = []
    for _ in range(frames):
        t0 = time.perf_counter(); oracle.render_scene(sc, cam, background=BG); ts.append(time.perf_counter() - t0)
    return 1.0 / float(np.median(ts))

sweep = [t for t in (1, 4, 8, 16, 32, 64, 128) if t <= (os.cpu_count() or 1)]
res = {t: cpu_fps(t) for t in sweep}
best = max(res, key=res.get)
print(f"config 1 (N={sc.n}, {cam.width}x{cam.height}): CPU oracle 1 thread {res[1]:.1f} frames/s; best of the sweep "
      f"{res[best]:.1f} frames/s with {best} threads ({os.cpu_count()} host CPUs; " + ", ".join(f"{t}: {v:.0f}" for t, v in res.items()) + ")")
try:
    import torch
    if torch.cuda.is_available():
        from sim_a_splat_amd.rasterizer import Rasterizer
        r = Rasterizer(0)
        r.upload(sc.means, sc.opacities, sc.sh, quats=sc.quats, scales=sc.scales, sh_degree=3)
        for _ in range(20): r.render(cam.viewmat, cam.K, cam.width, cam.height, BG, want=("rgb",))
        t0 = time.perf_counter()
        for _ in range(500): r.render(cam.viewmat, cam.K, cam.width, cam.height, BG, want=("rgb",))
        dt = time.perf_counter() - t0
        print(f"  MI355X, blocking sas_render: {500 / dt:.0f} frames/s ({dt / 500 * 1e6:.0f} us per frame)")
        r.close()
except ImportError:
    pass
