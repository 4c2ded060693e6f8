"""Per-stage timing probe (hipEvents inside the C ABI) for a BASELINE config.  GPU box only."""
import argparse
import json
import time

import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sim_a_splat_amd.rasterizer import Rasterizer  # noqa: E402
from sim_a_splat_amd.synthetic import NERFSTUDIO_EVAL_BACKGROUND, config_scene_and_cameras

ap = argparse.ArgumentParser()
ap.add_argument("--cfg", type=int, default=3)
ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--frames", type=int, default=10)
ap.add_argument("--fast-exp", action="store_true")
ap.add_argument("--plain", action="store_true", help="no stage events: blocking frames as a caller issues them (wall time per frame)")
a = ap.parse_args()
sc, cams = config_scene_and_cameras(a.cfg, a.scale)
cam = cams[0]
r = Rasterizer(0)
t0 = time.time()
r.upload(sc.means, sc.opacities, sc.sh, quats=sc.quats, scales=sc.scales, sh_degree=3)
print("upload s", time.time() - t0, flush=True)
out = None
acc = {}
if a.plain:
    for i in range(a.frames + 20):
        if i == 20:
            t0 = time.perf_counter()
        out = r.render(cam.viewmat, cam.K, cam.width, cam.height, NERFSTUDIO_EVAL_BACKGROUND, want=("rgb",), out=out)
    print(json.dumps({"cfg": a.cfg, "us_per_blocking_frame": (time.perf_counter() - t0) / a.frames * 1e6}))
    sys.exit(0)
for i in range(a.frames + 2):
    out = r.render(cam.viewmat, cam.K, cam.width, cam.height, NERFSTUDIO_EVAL_BACKGROUND, want=("rgb",),
                   timing=True, fast_exp=a.fast_exp, out=out)
    if i >= 2:
        for k, v in r.stage_times().items():
            acc.setdefault(k, []).append(v)
med = {k: float(np.median(v)) for k, v in acc.items()}
print(json.dumps({"cfg": a.cfg, "n": sc.n, "wh": [cam.width, cam.height], "stats": r.stats(), "stage_ms": med}))
