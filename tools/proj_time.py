"""Per-workgroup laps of the projection's two roles (needs a -DSAS_TUNE_PTIME build, SAS_LIB_PATH=variants/lib_ptime.so):
where a geometry workgroup's residency goes, when geometry and colour workgroups run inside the launch."""
import ctypes, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sim_a_splat_amd import _capi
from sim_a_splat_amd.rasterizer import Rasterizer
from sim_a_splat_amd.synthetic import NERFSTUDIO_EVAL_BACKGROUND as BG, config_scene_and_cameras
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sc, cams = config_scene_and_cameras(cfg)
cam = cams[0]
r = Rasterizer(0)
r.upload(sc.means, sc.opacities, sc.sh, quats=sc.quats, scales=sc.scales, sh_degree=3)
L = _capi.lib()
L.sas_debug_proj_laps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
for _ in range(4):
    r.render(cam.viewmat, cam.K, cam.width, cam.height, BG, want=("rgb",))
r.render(cam.viewmat, cam.K, cam.width, cam.height, BG, want=("rgb",), timing=True)
n = min((sc.n + 255) // 256, 8192)
geo = (ctypes.c_uint64 * (16 * n))()
col = (ctypes.c_uint64 * (2 * n))()
assert L.sas_debug_proj_laps(geo, col, n) == 0
g = np.array(geo, dtype=np.uint64).reshape(n, 16).astype(np.int64)
c = np.array(col, dtype=np.uint64).reshape(n, 2).astype(np.int64)
k0 = min(g[:, 14].min(), c[:, 0][c[:, 0] > 0].min())
gs, ge = (g[:, 14] - k0) * 0.01, (g[:, 15] - k0) * 0.01
cs, ce = (c[:, 0] - k0) * 0.01, (c[:, 1] - k0) * 0.01
print(f"cfg{cfg}: projection {r.stage_times()['project'] * 1e3:.1f} us by events; geometry workgroups span {gs.min():.1f}..{ge.max():.1f} us, colour {cs.min():.1f}..{ce.max():.1f} us")
d = ge - gs
okc = c[:, 1] > 0
print(f"  geometry: {n} workgroups, mean residency {d.mean():.2f} us (sum {d.sum() / 1e3:.2f} ms -> {d.sum() / ge.max():.0f} resident on average); colour: mean {(ce - cs)[okc].mean():.2f} us ({(ce - cs)[okc].sum() / max(ce.max(), 1e-9):.0f} resident)")
names = ["loads + T1 + record stores", "window", "zero + cull/count pass", "count atomics (returning)", "emit pass", "big rectangles", "visible counts",
         "store acks + barrier", "ticket", "tail: rest", "tail: fence + visible counts", "tail: pass A", "tail: class starts", "tail: pass B + stats"]
lap = g[:, :14] * 0.01
for k, nm in enumerate(names):
    print(f"    {nm:30s} mean {lap[:, k].mean():6.2f} us   p90 {np.quantile(lap[:, k], 0.9):6.2f}   share {lap[:, k].sum() / d.sum():5.3f}")
for q in (0.25, 0.5, 0.75, 0.9, 1.0):
    print(f"  {int(q * 100):3d} % of geometry ended by {np.quantile(ge, q):6.1f} us, of colour by {np.quantile(ce, q):6.1f} us")
print("  resident workgroups over time (geometry + colour):  " + "  ".join(
    f"{t:.0f}us:{int(((gs <= t) & (ge > t)).sum())}+{int(((cs <= t) & (ce > t) & okc).sum())}" for t in np.arange(2.0, max(ge.max(), ce.max()), 6.0)))
for q in (0.95, 0.99, 0.999):
    print(f"  geometry: {q * 100:.1f} % ended by {np.quantile(ge, q):6.1f} us; residency quantile {np.quantile(d, q):6.1f} us")
print("  slowest geometry workgroups (index, begin, end, laps):")
for i in np.argsort(-d)[:8]:
    print(f"    {i:5d} {gs[i]:6.1f} {ge[i]:6.1f}  " + " ".join(f"{lap[i, k]:5.1f}" for k in range(14)))
print("  last to end:")
for i in np.argsort(-ge)[:6]:
    print(f"    {i:5d} {gs[i]:6.1f} {ge[i]:6.1f}  " + " ".join(f"{lap[i, k]:5.1f}" for k in range(14)))
last = int(np.argmax(ge))
print(f"  tail workgroup {last}: began {gs[last]:.1f}, tail laps {lap[last, 9:14]} us, ended {ge[last]:.1f}")
r.close()
