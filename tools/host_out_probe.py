"""Why is render_batch_host slow with some `out=` tensors, and per-env pose sets at E=16?  (diagnostic)"""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sim_a_splat_amd.rasterizer import Rasterizer
from sim_a_splat_amd.synthetic import NERFSTUDIO_EVAL_BACKGROUND as BG, make_scene, random_group_poses, ring_camera
G = 8
sc = make_scene(113_831, seed=2, n_groups=G)
r = Rasterizer(0)
r.upload(sc.means, sc.opacities, sc.sh, quats=sc.quats, scales=sc.scales, sh_degree=3, group_id=sc.group_id, n_groups=G)
cams = [ring_camera(320, 240, 262.0, yaw_deg=0.0), ring_camera(320, 240, 262.0, yaw_deg=60.0, elev=0.5)]
V = np.stack([c.viewmat for c in cams]); K = np.stack([c.K for c in cams])
def t(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
a = torch.empty((2, 240, 320, 3), dtype=torch.uint8, pin_memory=True)
b = torch.empty((2, 240, 320, 3), dtype=torch.uint8).pin_memory()
c = torch.empty((2, 240, 320, 3), dtype=torch.uint8)
print("out=None              %.1f us" % t(lambda: r.render_batch_host(V, K, 320, 240, BG)))
print("out=empty(pin=True)   %.1f us" % t(lambda: r.render_batch_host(V, K, 320, 240, BG, out=a)))
print("out=empty().pin_memory() %.1f us" % t(lambda: r.render_batch_host(V, K, 320, 240, BG, out=b)))
print("out=pageable          %.1f us" % t(lambda: r.render_batch_host(V, K, 320, 240, BG, out=c)))
print("a is_pinned", a.is_pinned(), "b", b.is_pinned(), "ptr mod 4096:", a.data_ptr() % 4096, b.data_ptr() % 4096)
poses = [random_group_poses(G, seed=s) for s in range(64)]
k = [0]
def step_set():
    k[0] += 1; r.set_group_poses(poses[k[0] % 64]); r.render_batch_host(V, K, 320, 240, BG, out=a)
print("set_group_poses + out=a  %.1f us" % t(step_set))
# E = 16: per-env poses vs one pose set, same content (every env gets the SAME poses through its own set)
E = 16
cams16 = [ring_camera(320, 240, 262.0, yaw_deg=(360.0 * i) / (2 * E), elev=0.5 * (i & 1)) for i in range(2 * E)]
V16 = np.stack([c_.viewmat for c_ in cams16]); K16 = np.stack([c_.K for c_ in cams16]); idx = [v // 2 for v in range(2 * E)]
same = np.stack([poses[3]] * E); diff = np.stack([random_group_poses(G, seed=100 + e) for e in range(E)])
r.set_group_poses(poses[3])
print("E=16 one pose set            %.1f us" % t(lambda: r.render_batch_host(V16, K16, 320, 240, BG), 60))
print("E=16 sets, identical poses   %.1f us" % t(lambda: r.render_batch_host(V16, K16, 320, 240, BG, pose_sets=same, pose_set=idx), 60))
print("E=16 sets, different poses   %.1f us" % t(lambda: r.render_batch_host(V16, K16, 320, 240, BG, pose_sets=diff, pose_set=idx), 60))
print("stats", r.stats())
