import sys, time, cProfile, pstats
sys.path.insert(0, ".")
import numpy as np, torch
from sim_a_splat_amd.gaussian_splat import GaussianSplat, PinholeCamera, SplatModel
from sim_a_splat_amd.synthetic import c2w_opengl_from_viewmat, make_scene, ring_camera, NERFSTUDIO_EVAL_BACKGROUND as BG
sc = make_scene(292_247, seed=2)
model = SplatModel(sc.means, np.log(sc.scales), sc.quats, sc.sh[:, 0], sc.sh[:, 1:], np.log(sc.opacities / (1 - sc.opacities)).reshape(-1, 1), sh_degree=3, device="cuda:0")
cam = ring_camera(640, 480, 525.0)
K = cam.K
gs = GaussianSplat.from_model(model, PinholeCamera(torch.eye(4)[None, :3], float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), 640, 480))
poses = [torch.from_numpy(c2w_opengl_from_viewmat(ring_camera(640, 480, 525.0, yaw_deg=3.6 * i).viewmat)) for i in range(100)]
gs.render(poses[0]); torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    for i in range(400): out = gs.render(poses[i % 100])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("Door A GaussianSplat.render: %.1f us per call" % (dt / 400 * 1e6))
r = model._rasterizer()
cams = [ring_camera(640, 480, 525.0, yaw_deg=3.6 * i) for i in range(100)]
outs = None
for rep in range(2):
    t0 = time.perf_counter()
    for i in range(400):
        c = cams[i % 100]
        o = r.render(c.viewmat, c.K, 640, 480, BG, want=("rgb", "alpha", "depth"), depth_fill_max=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("Rasterizer.render same outputs: %.1f us per call" % (dt / 400 * 1e6))
pr = cProfile.Profile(); pr.enable()
for i in range(400): out = gs.render(poses[i % 100])
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(14)
