"""Tile-kernel time on long translucent lists (nothing saturates: every tile walks all its rounds) --
the case where re-scanning the whole key list per round used to cost n^2 / 512 key reads."""
import sys, json
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sim_a_splat_amd.rasterizer import Rasterizer
from sim_a_splat_amd.synthetic import NERFSTUDIO_EVAL_BACKGROUND as BG, make_scene, ring_camera
for n in (40_000, 160_000, 400_000):
    rng = np.random.default_rng(22)
    sc = make_scene(n, seed=98, log_scale_mean=float(np.log(0.01)))
    sc.means[:] = rng.uniform(-0.4, 0.4, size=sc.means.shape).astype(np.float32)
    sc.opacities[:] = np.clip(sc.opacities, 0.0040, 0.0042)        # ~1/250: thousands of splats before a pixel saturates
    cam = ring_camera(80, 64, 110.0)
    r = Rasterizer(0)
    r.upload(sc.means, sc.opacities, sc.sh, quats=sc.quats, scales=sc.scales, sh_degree=3)
    acc = []
    for i in range(6):
        r.render(cam.viewmat, cam.K, 80, 64, BG, want=("rgb",), timing=True)
        if i >= 2:
            acc.append(r.stage_times()["blend"])
    st = r.stats()
    print(f"N={n}: max list {st['max_tile_len']}, M={st['n_isect']}, tile kernel {np.median(acc):.3f} ms")
    r.close()
