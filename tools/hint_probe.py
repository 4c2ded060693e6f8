"""EXPERIMENT (upper bound of depth-hinted tile lists, -DSAS_TUNE_HINT build, SAS_TUNE_HINT=1): the same view rendered repeatedly; from the
second frame on the projection leaves out the keys beyond the depth at which each tile saturated in the previous frame.  Prints per frame:
keys binned, projection / tile-kernel time, and whether the frame equals the first one bit for bit."""
import json, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sim_a_splat_amd.rasterizer import Rasterizer
from sim_a_splat_amd.synthetic import NERFSTUDIO_EVAL_BACKGROUND as BG, config_scene_and_cameras
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sc, cams = config_scene_and_cameras(cfg)
cam = cams[0]
r = Rasterizer(0)
r.upload(sc.means, sc.opacities, sc.sh, quats=sc.quats, scales=sc.scales, sh_degree=3)
first = None
for i in range(24):   # (four slots: a slot meets its own hints every fourth blocking frame... blocking frames restart at slot 0)
    out = r.render(cam.viewmat, cam.K, cam.width, cam.height, BG, want=("rgb", "alpha", "depth"), timing=True)
    img = torch.cat([out["rgb"].flatten(), out["alpha"].flatten(), out["depth"].flatten()]).cpu()
    if first is None:
        first = img
    st, tm = r.stats(), r.stage_times()
    if i < 3 or i % 6 == 5:
        print(json.dumps({"frame": i, "n_keys": st["n_keys"], "project_ms": round(tm["project"], 4), "tile_ms": round(tm["blend"], 4), "equal_first": bool(torch.equal(img, first))}))
