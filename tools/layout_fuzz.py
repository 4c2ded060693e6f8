"""Both layouts of the tile kernel on random scenes: every output of SAS_QUAD=1 must equal SAS_QUAD=0 bit for bit.
Scene size, splat scale, opacity range, image size (ragged), depth fill and group counts are drawn per seed.
    python tools/layout_fuzz.py [n_seeds]"""
import os, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sim_a_splat_amd.synthetic import NERFSTUDIO_EVAL_BACKGROUND as BG, make_scene, ring_camera

def contexts():
    from sim_a_splat_amd.rasterizer import Rasterizer
    rs = []
    for q in ("0", "1"):
        os.environ["SAS_QUAD"] = q      # read at sas_create
        rs.append(Rasterizer(0))
    del os.environ["SAS_QUAD"]
    return rs

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
r0, r1 = contexts()
bad = 0
for seed in range(n_seeds):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([50, 800, 6000, 30000, 120000]))
    ls = float(rng.uniform(np.log(0.004), np.log(0.15)))
    sc = make_scene(n, seed=2000 + seed, log_scale_mean=ls)
    lo = float(rng.choice([0.004, 0.05, 0.5]))
    sc.opacities[:] = np.clip(sc.opacities, lo, min(1.0, lo * 20 + 0.01)).astype(np.float32)
    if rng.random() < 0.25:
        sc.means[:, 2] = np.round(sc.means[:, 2] * 4) / 4          # depth planes: crowded buckets / ties
    W, H = int(rng.integers(17, 420)), int(rng.integers(17, 300))
    cams = [ring_camera(W, H, float(rng.uniform(0.4, 1.2)) * W, yaw_deg=float(rng.uniform(0, 360)), elev=float(rng.uniform(-0.5, 0.5)))
            for _ in range(int(rng.integers(1, 4)))]
    Vs, Ks = np.stack([c.viewmat for c in cams]), np.stack([c.K for c in cams])
    fill = bool(rng.random() < 0.5)
    outs = []
    for r in (r0, r1):
        r.upload(sc.means, sc.opacities, sc.sh, quats=sc.quats, scales=sc.scales, sh_degree=3)
        o = r.render_batch(Vs, Ks, W, H, BG, want=("rgb", "alpha", "depth", "rgb8"), depth_fill_max=fill)
        outs.append({k: v.clone() for k, v in o.items()})
        st = r.stats()
    same = all(torch.equal(outs[0][k].view(torch.int32) if outs[0][k].dtype == torch.float32 else outs[0][k],
                           outs[1][k].view(torch.int32) if outs[1][k].dtype == torch.float32 else outs[1][k]) for k in outs[0])
    bad += not same
    print(f"seed {seed}: n={n} {W}x{H} views={len(cams)} fill={fill} max_list={st['max_tile_len']} fallback={st['fallback_tiles']} -> {'same' if same else 'DIFFERENT'}")
print("layouts agree on every frame" if bad == 0 else f"{bad} frames differ")
sys.exit(1 if bad else 0)
