"""Python-side cost of one Rasterizer.render call, piece by piece (us): python tools/py_overhead_probe.py"""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sim_a_splat_amd.rasterizer import Rasterizer
from sim_a_splat_amd.synthetic import NERFSTUDIO_EVAL_BACKGROUND as BG, make_scene, ring_camera

def t(fn, n=100000):
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e6

dev = torch.device("cuda:0")
x = torch.empty((64, 64, 3), device=dev)
print(f"torch.cuda.current_stream(dev).cuda_stream  {t(lambda: torch.cuda.current_stream(dev).cuda_stream):.2f}")
raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
if raw is not None:
    print(f"torch._C._cuda_getCurrentRawStream(0)        {t(lambda: raw(0)):.2f}")
print(f"tensor.data_ptr()                            {t(lambda: x.data_ptr()):.2f}")
print(f"shape / dtype / contiguous / device checks   {t(lambda: (x.shape != (64, 64, 3) or x.dtype != torch.float32 or not x.is_contiguous() or x.device != dev)):.2f}")
r = Rasterizer(0)
sc = make_scene(50, seed=1)
r.upload(sc.means, sc.opacities, sc.sh, quats=sc.quats, scales=sc.scales, sh_degree=3)
cam = ring_camera(64, 64, 60.0)
out = {"rgb": x}
r.render(cam.viewmat, cam.K, 64, 64, BG, want=("rgb",), out=out)
n = 20000
t0 = time.perf_counter()
for i in range(n):
    r.render(cam.viewmat, cam.K, 64, 64, BG, want=("rgb",), out=out, block=False)
r.wait()
print(f"Rasterizer.render, 50 Gaussians at 64x64, not blocking: {(time.perf_counter() - t0) / n * 1e6:.2f} per call (Python + C ABI + three launches)")
L, ctx = r._L, r._ctx
V, K, bgv = cam.viewmat, cam.K, np.asarray(BG, np.float32)
pV, pK, pbg, px = V.ctypes.data, K.ctypes.data, bgv.ctypes.data, x.data_ptr()
from sim_a_splat_amd import _capi
st = torch.cuda.current_stream(dev).cuda_stream
t0 = time.perf_counter()
for i in range(n):
    L.sas_render(ctx, pV, pK, 64, 64, pbg, _capi.SAS_ASYNC, px, None, None, None, st)
r.wait()
print(f"the bare C-ABI call in the same loop:                    {(time.perf_counter() - t0) / n * 1e6:.2f} per call")
