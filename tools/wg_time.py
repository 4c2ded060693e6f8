"""Per-workgroup timing of the tile kernel (needs a -DSAS_TUNE_WGTIME build): which tiles run longest,
when they start and end inside the kernel, and how full the chip is over the kernel's duration."""
import ctypes, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sim_a_splat_amd import _capi
from sim_a_splat_amd.rasterizer import Rasterizer
from sim_a_splat_amd.synthetic import NERFSTUDIO_EVAL_BACKGROUND as BG, config_scene_and_cameras
for cfg in [a if a == "gym" else int(a) for a in sys.argv[1:]] or [3]:
    r = Rasterizer(0)
    if cfg == "gym":   # one 240x320 Gym camera on the 113 831-Gaussian stand-in scene (quad layout: 8-pixel tiles)
        from sim_a_splat_amd.synthetic import make_scene, ring_camera
        sc = make_scene(113_831, seed=2, n_groups=8)
        cam = ring_camera(320, 240, 262.0, yaw_deg=0.0)
        r.upload(sc.means, sc.opacities, sc.sh, quats=sc.quats, scales=sc.scales, sh_degree=3, group_id=sc.group_id, n_groups=8)
    else:
        sc, cams = config_scene_and_cameras(cfg)
        cam = cams[0]
        r.upload(sc.means, sc.opacities, sc.sh, quats=sc.quats, scales=sc.scales, sh_degree=3)
    L = _capi.lib()
    r.render(cam.viewmat, cam.K, cam.width, cam.height, BG, want=("rgb",))
    px = 8 if r.stats()["quad_layout"] else 16
    tiles = ((cam.width + px - 1) // px) * ((cam.height + px - 1) // px)
    n = min(tiles, 16384)
    out = (ctypes.c_uint64 * (3 * n))()
    for _ in range(3):
        r.render(cam.viewmat, cam.K, cam.width, cam.height, BG, want=("rgb",))
    r.render(cam.viewmat, cam.K, cam.width, cam.height, BG, want=("rgb",), timing=True)
    L.sas_debug_wg(out, n)
    a = np.array(out, dtype=np.uint64).reshape(n, 3).astype(np.int64)
    t0, t1, ln = a[:, 0], a[:, 1], a[:, 2]
    k0 = t0.min()
    s, e = (t0 - k0) * 0.01, (t1 - k0) * 0.01            # us
    d = e - s
    print(f"cfg{cfg}: tile kernel {r.stage_times()['blend'] * 1e3:.1f} us by events; workgroups span {e.max():.1f} us; {n} workgroups, "
          f"sum of durations {d.sum() / 1e3:.1f} ms -> mean residency {d.sum() / e.max():.0f} workgroups")
    ph = None
    if hasattr(L, "sas_debug_wg_phases"):
        o4 = (ctypes.c_uint64 * (8 * n))()
        L.sas_debug_wg_phases(o4, n)
        ph = np.array(o4, dtype=np.uint64).reshape(n, 8).astype(np.int64)
    for i in np.argsort(-d)[:5]:
        extra = ""
        if ph is not None:   # thread 0's clock: ordering, compositing (of which the trips), batches
            trips = f" (wave 0's trips {ph[i, 2] * 0.01:5.1f}) us in {ph[i, 3]} batches" if r.stats()["quad_layout"] else " us"   # (the quad layout's loop is instrumented)
            if r.stats()["quad_layout"]:
                trips += f"; per batch: first barrier {ph[i, 4] * 0.01:4.1f}, mask + staging (waits for the gathered records) {ph[i, 5] * 0.01:4.1f}, second barrier {ph[i, 6] * 0.01:4.1f}, queue {ph[i, 7] * 0.01:4.1f} us in all"
            extra = f"; ordering {ph[i, 0] * 0.01:5.1f}, compositing {ph[i, 1] * 0.01:5.1f}{trips}"
        print(f"  long: launch index {i:5d} list {ln[i]:6d} start {s[i]:7.1f} end {e[i]:7.1f} ran {d[i]:6.1f} us{extra}")
    if ph is not None:
        tot = d.sum()
        if not r.stats()["quad_layout"]:
            print(f"  long lists (ordinary layout): passes over the keys {ph[:, 2].sum() * 0.01 / tot:.2f} of the slot time, of which the rounds' collect passes {ph[:, 3].sum() * 0.01 / tot:.2f}")
        print(f"  all workgroups: ordering {ph[:, 0].sum() * 0.01 / tot:.2f}, compositing (staging, masks, queues, trips, waits for the slowest wave) "
              f"{ph[:, 1].sum() * 0.01 / tot:.2f} of the slot time" + (f", wave 0's trips {ph[:, 2].sum() * 0.01 / tot:.2f}" if r.stats()["quad_layout"] else ""))
    if hasattr(L, "sas_debug_wg_laps") and not r.stats()["quad_layout"]:
        o16 = (ctypes.c_uint64 * (16 * n))()
        L.sas_debug_wg_laps(o16, n)
        lap = np.array(o16, dtype=np.uint64).reshape(n, 16).astype(np.int64) * 0.01   # us
        names = ["tile order + offsets", "min/max pass", "histogram pass", "bucket selection", "partition", "collect", "chunk ordering",
                 "first batch: records + barrier", "staging + masks + barrier", "queue building", "trips", "batch barrier (other waves)",
                 "epilogue", "short list: loads + ordering"]
        tot = d.sum()
        print("  exclusive partition of thread 0's time, share of the slot time (all workgroups | lists > 512 | 1..512 | empty):")
        cls = [np.ones(n, bool), ln > 512, (ln > 0) & (ln <= 512), ln == 0]
        for k, nm in enumerate(names):
            print(f"    {nm:32s} " + " | ".join(f"{lap[m, k].sum() / max(d[m].sum(), 1e-9):5.3f}" for m in cls))
        print(f"    {'(sum of the laps)':32s} " + " | ".join(f"{lap[m].sum() / max(d[m].sum(), 1e-9):5.3f}" for m in cls)
              + "   class share of slot time: " + " | ".join(f"{d[m].sum() / tot:4.2f}" for m in cls))
    for i in np.argsort(-e)[:3]:
        print(f"  last: launch index {i:5d} list {ln[i]:6d} start {s[i]:7.1f} end {e[i]:7.1f} ran {d[i]:6.1f} us")
    for q in (0.25, 0.5, 0.75, 0.9):
        print(f"  {int(q * 100)} % of the workgroups have ended by {np.quantile(e, q):6.1f} us", end=";")
    print()
    # slot time by list-length class: where the workgroup-microseconds go
    edges = [0, 1, 65, 257, 513, 1025, 2049, 4097, 1 << 30]
    names = ["empty", "1-64", "65-256", "257-512", "513-1024", "1025-2048", "2049-4096", "> 4096"]
    for lo, hi, nm in zip(edges[:-1], edges[1:], names):
        m = (ln >= lo) & (ln < hi)
        if m.any():
            print(f"  lists {nm:10s}: {int(m.sum()):5d} workgroups, mean {d[m].mean():6.1f} us, {100 * d[m].sum() / d.sum():5.1f} % of the slot time, "
                  f"{d[m].sum() / max(ln[m].sum(), 1) * 1e3:7.1f} ns per key")
    r.close()
