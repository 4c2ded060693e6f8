"""Which absurd input values make the HIP path and the oracle part ways: ONE value written over 1 % of ONE array at a time.
    python tests/tools/poison_probe.py"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import oracle  # noqa: E402
from sim_a_splat_amd.rasterizer import Rasterizer  # noqa: E402
from sim_a_splat_amd.synthetic import NERFSTUDIO_EVAL_BACKGROUND as BG, make_scene, ring_camera  # noqa: E402

r = Rasterizer(0)
cam = ring_camera(160, 120, 140.0, yaw_deg=30.0, elev=0.2)
vals = [np.nan, np.inf, -np.inf, 1e30, -1e30, 1e-30, 0.0, -1.0, 2.0]
for name in ("means", "scales", "quats", "opacities", "sh"):
    for v in vals:
        sc = make_scene(3000, seed=7, log_scale_mean=float(np.log(0.05)))
        rng = np.random.default_rng(3)
        flat = getattr(sc, name).reshape(-1)
        flat[rng.integers(0, flat.size, size=max(1, flat.size // 100))] = np.float32(v)
        r.upload(sc.means, sc.opacities, sc.sh, quats=sc.quats, scales=sc.scales, sh_degree=3)
        o = r.render(cam.viewmat, cam.K, cam.width, cam.height, BG, want=("rgb", "alpha", "depth", "rgb8"))
        ref = oracle.render(sc.means, sc.opacities, sc.sh, cam.viewmat, cam.K, cam.width, cam.height, quats=sc.quats, scales=sc.scales,
                            sh_degree=3, background=BG, want_rgb8=True)
        st = r.stats()
        out = []
        for k in ("rgb", "alpha", "depth", "rgb8"):
            g = o[k].cpu().numpy()
            if not np.array_equal(g, ref[k], equal_nan=g.dtype != np.uint8):
                out.append(f"{k}:{int((~np.isclose(g.astype(np.float64), ref[k].astype(np.float64), rtol=0, atol=0, equal_nan=True)).sum())}")
        cnt = "" if (st["n_visible"], st["n_isect"]) == (ref["n_visible"], ref["n_isect"]) else \
            f" counts vis {st['n_visible']}/{ref['n_visible']} isect {st['n_isect']}/{ref['n_isect']}"
        print(f"{name:10s} <- {v!s:6s}: {'bit-equal' if not out else 'DIFFERENT ' + ' '.join(out)}{cnt}", flush=True)
r.close()
