"""Exact tile culling on random scenes: every output with SAS_CULL=1 must equal SAS_CULL=0 bit for bit, in both tile layouts.
Scene size, splat scales (per axis, ratios up to 1 : 5000: needles and discs in every orientation), opacity range, image size
(ragged), focal length, camera distance (also inside the cloud) and depth fill are drawn per seed.
    python tools/cull_fuzz.py [n_seeds]"""
import os, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sim_a_splat_amd.synthetic import NERFSTUDIO_EVAL_BACKGROUND as BG, make_scene, ring_camera

def contexts():
    from sim_a_splat_amd.rasterizer import Rasterizer
    rs = {}
    for quad in ("0", "1"):
        for cull in ("1", "0"):
            os.environ["SAS_QUAD"], os.environ["SAS_CULL"] = quad, cull      # read at sas_create
            rs[(quad, cull)] = Rasterizer(0)
    del os.environ["SAS_QUAD"], os.environ["SAS_CULL"]
    return rs

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rs = contexts()
bad = 0
keys_on = keys_off = 0
for seed in range(n_seeds):
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([50, 800, 6000, 30000, 100000]))
    sc = make_scene(n, seed=8000 + seed, log_scale_mean=float(rng.uniform(np.log(0.004), np.log(0.1))))
    kind = int(rng.integers(0, 4))
    if kind == 1:      # needles / discs: independent log-uniform scales per axis
        sc.scales[:] = np.exp(rng.uniform(np.log(6e-5), np.log(0.3), size=sc.scales.shape)).astype(np.float32)
    elif kind == 2:    # large splats
        sc.scales[:] = (sc.scales * float(rng.uniform(3, 30))).astype(np.float32)
    lo = float(rng.choice([0.004, 0.0045, 0.05, 0.5, 0.99]))
    sc.opacities[:] = np.clip(sc.opacities, lo, min(0.9999, lo * 20 + 0.01)).astype(np.float32)
    W, H = int(rng.integers(17, 700)), int(rng.integers(17, 500))
    radius = float(rng.choice([0.15, 0.6, 1.5, 3.0, 8.0]))
    cam = ring_camera(W, H, float(rng.uniform(0.3, 1.5)) * W, yaw_deg=float(rng.uniform(0, 360)), radius=radius, elev=float(rng.uniform(-0.5, 0.5)) * radius)
    fill = bool(rng.random() < 0.5)
    outs, st = {}, {}
    for k, r in rs.items():
        r.upload(sc.means, sc.opacities, sc.sh, quats=sc.quats, scales=sc.scales, sh_degree=3)
        o = r.render(cam.viewmat, cam.K, W, H, BG, want=("rgb", "alpha", "depth", "rgb8"), depth_fill_max=fill)
        outs[k] = {name: v.clone() for name, v in o.items()}
        st[k] = r.stats()
    def same(a, b):
        return all(torch.equal(a[k].view(torch.int32) if a[k].dtype == torch.float32 else a[k], b[k].view(torch.int32) if b[k].dtype == torch.float32 else b[k]) for k in a)
    ok = all(same(outs[("0", "0")], outs[k]) for k in outs) and all(st[k]["n_isect"] == st[("0", "0")]["n_isect"] for k in st)
    bad += not ok
    keys_on += st[("0", "1")]["n_keys"]; keys_off += st[("0", "0")]["n_keys"]
    print(f"seed {seed}: n={n} kind={kind} {W}x{H} radius={radius} fill={fill} n_isect={st[('0','0')]['n_isect']} keys culled/whole: 16 px {st[('0','1')]['n_keys']}/{st[('0','0')]['n_keys']}, 8 px {st[('1','1')]['n_keys']}/{st[('1','0')]['n_keys']} -> {'same' if ok else 'DIFFERENT'}")
print(f"keys binned with the culling: {keys_on / max(keys_off, 1):.3f} of the rectangles' ({keys_on} of {keys_off})")
print("culled and whole-rectangle lists give the same frames everywhere" if bad == 0 else f"{bad} frames differ")
sys.exit(1 if bad else 0)
