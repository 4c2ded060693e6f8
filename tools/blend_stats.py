"""Lane utilisation of the compositing loop (needs a -DSAS_TUNE_STATS build of the library)."""
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
    out = (ctypes.c_uint64 * 8)()
    r.render(cam.viewmat, cam.K, cam.width, cam.height, BG, want=("rgb",))
    L.sas_debug_counters(out, 1)
    r.render(cam.viewmat, cam.K, cam.width, cam.height, BG, want=("rgb",))
    L.sas_debug_counters(out, 1)
    it, itc, cand, upd, staged, queued = [int(x) for x in out[:6]]
    wait_cyc, loop_cyc = int(out[6]), int(out[7])
    st = r.stats()
    print(f"cfg{cfg}: M={st['n_isect']} staged={staged} ({staged/st['n_isect']:.2f} of M) queued pairs={queued} ({queued/max(staged,1):.2f} per staged entry)")
    print(f"  wave cycles in the compositing loop {loop_cyc:.3g}, waiting at the batch barrier {wait_cyc:.3g} ({wait_cyc/max(loop_cyc,1):.2f} of loop)")
    print(f"  wave-iterations={it} with candidates={itc} ({itc/max(it,1):.2f}); candidate lanes/iter={cand/max(itc,1):.1f} composited lanes/iter={upd/max(itc,1):.1f}")
