"""Lane utilisation of the compositing loop (needs a -DSAS_TUNE_STATS build: SAS_LIB_PATH=variants/lib_stats.so)."""
import ctypes, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sim_a_splat_amd import _capi
from sim_a_splat_amd.rasterizer import Rasterizer
from sim_a_splat_amd.synthetic import NERFSTUDIO_EVAL_BACKGROUND as BG, config_scene_and_cameras
for cfg in [int(a) for a in sys.argv[1:]] or [3]:
    sc, cams = config_scene_and_cameras(cfg)
    cam = cams[0]
    r = Rasterizer(0)
    r.upload(sc.means, sc.opacities, sc.sh, quats=sc.quats, scales=sc.scales, sh_degree=3)
    L = _capi.lib()
    out = (ctypes.c_uint64 * 16)()
    r.render(cam.viewmat, cam.K, cam.width, cam.height, BG, want=("rgb",))
    L.sas_debug_counters(out, 1)
    r.render(cam.viewmat, cam.K, cam.width, cam.height, BG, want=("rgb",))
    L.sas_debug_counters(out, 1)
    it, stop_trips, stops, upd, staged, queued = [int(x) for x in out[:6]]
    trips = it // 2
    st = r.stats()
    print(f"cfg{cfg}: M={st['n_isect']} staged={staged} ({staged/st['n_isect']:.2f} of M) queued (entry, 4x4 block) pairs={queued} ({queued/max(staged,1):.2f} per staged entry)")
    print(f"  trips={trips} (two queue entries per 16-lane group each); composited={upd} pixel-splat pairs = {upd/max(trips*128,1):.2f} of the {trips*128} lane-slots issued")
    print(f"  trips on which a pixel terminated: {stop_trips} ({stop_trips/max(trips,1):.2f} of the trips), {stops} pixels")
    used, sentinel = int(out[8]), int(out[9])
    print(f"  (entry, 16-lane group) slots: {trips*8} issued, {sentinel} sentinels ({sentinel/max(trips*8,1):.2f}), {used} with a compositing lane ({used/max(trips*8,1):.2f})")
    gdead, nopass = int(out[10]), int(out[11])
    print(f"  slots of a group that had entirely terminated: {gdead} ({gdead/max(trips*8,1):.2f}); slots of a live group without a lane passing the alpha test, sentinels included: {nopass} ({nopass/max(trips*8,1):.2f})")
    wait_c, loop_c = int(out[6]), int(out[7])
    print(f"  wave cycles inside the trips {loop_c/1e6:.1f} M, waiting at the batch barrier for the slowest wave {wait_c/1e6:.1f} M ({wait_c/max(loop_c,1):.2f} of the trip time)")
    print(f"  staged entries with an empty block mask (reach no 4x4 block of their tile): {int(out[12])} of {int(out[13])} ({int(out[12])/max(int(out[13]),1):.3f})")
    print(f"  queued (entry, 4x4 block) pairs the splat's own bounding box (mean +- radii) would drop: {int(out[14]) - int(out[15])} of {int(out[14])} ({1 - int(out[15]) / max(int(out[14]), 1):.3f})")
