"""Diagnostic: render_batch_host per step with different host buffers, after the device-output loop of door_b_breakdown."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sim_a_splat_amd.rasterizer import Rasterizer
from sim_a_splat_amd.synthetic import NERFSTUDIO_EVAL_BACKGROUND as BG, make_scene, random_group_poses, ring_camera
sc = make_scene(113_831, seed=2, n_groups=8)
r = Rasterizer(0)
r.upload(sc.means, sc.opacities, sc.sh, quats=sc.quats, scales=sc.scales, sh_degree=3, group_id=sc.group_id, n_groups=8)
cams = [ring_camera(320, 240, 262.0, yaw_deg=0.0), ring_camera(320, 240, 262.0, yaw_deg=60.0, elev=0.5)]
V = np.stack([c.viewmat for c in cams]); K = np.stack([c.K for c in cams])
poses = [random_group_poses(8, seed=s) for s in range(300)]
host = torch.empty((2, 240, 320, 3), dtype=torch.uint8).pin_memory()
fresh = torch.empty((2, 240, 320, 3), dtype=torch.uint8, pin_memory=True)
def loop(fn, n=300):
    for s in range(30): fn(s)
    t0 = time.perf_counter()
    for s in range(n): fn(s)
    return (time.perf_counter() - t0) / n * 1e6
def a(s): r.set_group_poses(poses[s]); r.render_batch_host(V, K, 320, 240, BG, out=host)
def b(s): r.set_group_poses(poses[s]); r.render_batch_host(V, K, 320, 240, BG, out=fresh)
def c(s): r.set_group_poses(poses[s]); r.render_batch_host(V, K, 320, 240, BG)
def d(s): r.set_group_poses(poses[s % 64]); r.render_batch_host(V, K, 320, 240, BG, out=host)
def e(s): r.set_group_poses(poses[s]); o = r.render_batch(V, K, 320, 240, BG, want=("rgb8",))["rgb8"]; host.copy_(o)
print("before any copy_: out=host %.1f  out=fresh %.1f  out=None %.1f  64 poses %.1f" % (loop(a), loop(b), loop(c), loop(d)))
print("device output + host.copy_: %.1f" % loop(e))
print("after copy_:      out=host %.1f  out=fresh %.1f  out=None %.1f  64 poses %.1f" % (loop(a), loop(b), loop(c), loop(d)))
worst = max(range(300), key=lambda s: (r.set_group_poses(poses[s]), r.render_batch_host(V, K, 320, 240, BG, out=fresh), r.stats()["max_tile_len"])[2])
print("stats of the heaviest pose set", worst, r.stats())
ts = []
for s in range(300):
    r.set_group_poses(poses[s]); t0 = time.perf_counter(); r.render_batch_host(V, K, 320, 240, BG, out=fresh); ts.append((time.perf_counter() - t0) * 1e6)
ts = np.array(ts); print("per-pose-set step us: median %.1f  p90 %.1f  max %.1f  mean %.1f" % (np.median(ts), np.percentile(ts, 90), ts.max(), ts.mean()))
