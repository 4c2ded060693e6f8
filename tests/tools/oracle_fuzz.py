"""Differential fuzzer: the HIP path (through the C ABI) against the C oracle on randomly drawn scenes, cameras and entry points.
Every output -- rgb, accumulation, expected depth, uint8 frame, visible and intersection counts -- must equal the oracle's
bit for bit (the arithmetic contract); the seeded pytest cases fix a few dozen inputs, this draws the rest:

  scene      n in {1 .. 120 000}, splat scale over two decades, opacity bands, depth planes (ties), SH degree 0..3,
             final RGB + 3x3 covariances (Door B's input, degree -1), 0 - 200 link groups with random rigid poses; one case in
             sixteen is large (up to 1M Gaussians, up to 1920x1080); one in twelve is POISONED (NaN, +-Inf, 1e+-30, 0 written over
             1 % of the means / scales / quaternions / opacities / colours)
  camera     ragged image sizes from 17x17, strips of one tile row / column thousands of pixels long, focal length (now and then
             fish-eye-short or telescope-long), radius (a camera INSIDE the cloud crosses the near plane), principal points off
             centre or outside the image, a view matrix that is not quite a rotation
  entry      one blocking frame, a batch of 2-3 views (the pair projection), a batch with one pose set per view, host-delivered
             uint8 frames, a blocking frame with the complete sorted lists kept, and PIPELINED steps (3-6 steps enqueued without waiting -- single frames or batches, new group poses
             before every step, four frames in flight over the slot ring -- then one wait); depth fill on or off; nerfstudio's eval background or a drawn one

    python tests/tools/oracle_fuzz.py [n_seeds] [first_seed] [poison-all]         (exit code 1 on the first difference; prints each case)

Test infrastructure: lives under tests/ because it calls the oracle (the checker)."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import oracle  # noqa: E402
from sim_a_splat_amd.synthetic import (NERFSTUDIO_EVAL_BACKGROUND as BG, Camera, intrinsics, look_at_viewmat, make_scene,  # noqa: E402
                                       random_group_poses)

KEYS = ("rgb", "alpha", "depth", "rgb8")


def draw_case(seed: int, poison_all: bool = False) -> dict:
    rng = np.random.default_rng(77_000 + seed)
    n = int(rng.choice([1, 7, 50, 800, 6000, 30000, 120000], p=[0.04, 0.06, 0.1, 0.25, 0.25, 0.2, 0.1]))
    ls = float(rng.uniform(np.log(0.003), np.log(0.3)))
    n_groups = int(rng.choice([0, 0, 0, 3, 3, 7, 7, 40, 200]))      # (40, 200: beyond the pose rows a launch carries in its arguments)
    sc = make_scene(n, seed=88_000 + seed, log_scale_mean=ls, n_groups=n_groups)
    lo = float(rng.choice([0.004, 0.05, 0.5]))
    sc.opacities[:] = np.clip(sc.opacities, lo, min(1.0, lo * 20 + 0.01)).astype(np.float32)
    if rng.random() < 0.25:
        sc.means[:, 2] = np.round(sc.means[:, 2] * 4) / 4                    # depth planes: crowded buckets, ties
    if rng.random() < 0.15:
        sc.means *= np.float32(0.05)                                         # everything in a few tiles: long lists
    deg = int(rng.choice([-1, 0, 1, 2, 3, 3, 3]))
    W, H = int(rng.integers(17, 420)), int(rng.integers(17, 300))
    if rng.random() < 0.05:                                                  # a strip: one row or one column of tiles, thousands of pixels long
        W, H = (int(rng.integers(1000, 4000)), int(rng.integers(1, 17))) if rng.random() < 0.5 else (int(rng.integers(1, 17)), int(rng.integers(1000, 3000)))
    poisoned = bool(rng.random() < 0.08) or poison_all
    if rng.random() < 0.06:                                                  # now and then a large frame and a large scene
        W, H = int(rng.integers(640, 1921)), int(rng.integers(480, 1081))
        n = int(rng.choice([120000, 500000, 1000000]))
        sc = make_scene(n, seed=88_000 + seed, log_scale_mean=float(rng.uniform(np.log(0.004), np.log(0.03))), n_groups=n_groups)
    if poisoned:     # non-finite and absurd values in ~1 % of the Gaussians: both sides must cull or clamp them the same way, and the
        # device must not leave its buffers (the bounds-checked build counts)
        bad_vals = np.array([np.nan, np.inf, -np.inf, 1e30, -1e30, 1e-30, 0.0, 3e6, 1e12, -1.0], np.float32)
        for arr in (sc.means, sc.scales, sc.quats, sc.opacities, sc.sh):
            flat = arr.reshape(-1)
            k = max(1, flat.size // 100)
            flat[rng.integers(0, flat.size, size=k)] = bad_vals[rng.integers(0, bad_vals.size, size=k)]
    n_views = int(rng.choice([1, 1, 2, 3]))
    cams = []
    for _ in range(n_views):
        radius = float(rng.choice([0.3, 1.0, 3.0, 3.0, 6.0]))               # 0.3 / 1.0: inside the cloud
        yaw, elev = float(rng.uniform(0, 2 * np.pi)), float(rng.uniform(-0.8, 0.8)) * radius
        eye = (radius * np.sin(yaw), elev, radius * np.cos(yaw))
        f = float(rng.uniform(0.4, 1.5)) * W
        cx, cy = W / 2.0 + float(rng.uniform(-0.2, 0.2)) * W, H / 2.0 + float(rng.uniform(-0.2, 0.2)) * H
        V = look_at_viewmat(eye)
        if rng.random() < 0.1:                                               # odd cameras: fish-eye-short or telescope-long focal lengths, the
            f = float(rng.choice([0.03, 0.1, 8.0, 40.0])) * max(W, H)        # principal point outside the image, a view matrix that is not quite a rotation
            cx, cy = float(rng.uniform(-1.0, 2.0)) * W, float(rng.uniform(-1.0, 2.0)) * H
            V = V.copy(); V[:3, :3] *= np.float32(rng.uniform(0.97, 1.03))
        cams.append(Camera(V, intrinsics(f, f * float(rng.uniform(0.8, 1.25)), cx, cy), W, H))
    entry = "single" if n_views == 1 else str(rng.choice(["batch", "batch", "posed", "host"]))
    if n_groups == 0 and entry == "posed":
        entry = "batch"
    bg = BG if rng.random() < 0.5 else tuple(float(v) for v in rng.uniform(0, 1, size=3).astype(np.float32))
    full_sort = bool(entry == "single" and rng.random() < 0.15)       # complete sorted lists kept (the T4/T5 arrays' path)
    steps = 1
    if rng.random() < 0.2:
        entry, steps = "pipelined", int(rng.integers(3, 7))
    poses = [random_group_poses(n_groups, seed=99_000 + 7 * seed + v, max_angle=0.6, max_shift=0.3) for v in range(max(n_views, steps))] if n_groups else None
    if entry == "pipelined":     # every step looks from its own place
        yaw0 = float(rng.uniform(0, 2 * np.pi))
        step_cams = []
        for s_ in range(steps):
            row = []
            for v in range(n_views):
                a = yaw0 + 0.7 * s_ + 2.1 * v
                row.append(Camera(look_at_viewmat((3.0 * np.sin(a), 0.3 * s_ - 0.5, 3.0 * np.cos(a))), cams[v].K, W, H))
            step_cams.append(row)
        return dict(seed=seed, scene=sc, deg=deg, n_groups=n_groups, cams=cams, entry=entry, poses=poses, fill=bool(rng.random() < 0.5), W=W, H=H,
                    steps=steps, step_cams=step_cams, bg=bg, full_sort=False, poisoned=poisoned)
    return dict(seed=seed, scene=sc, deg=deg, n_groups=n_groups, cams=cams, entry=entry, poses=poses, fill=bool(rng.random() < 0.5), W=W, H=H, bg=bg,
                full_sort=full_sort, poisoned=poisoned)


def scene_inputs(c: dict) -> dict:
    """What both sides are handed: SH coefficients cut to the degree, or final RGB + covariances (degree -1)."""
    sc, deg = c["scene"], c["deg"]
    if deg >= 0:
        return dict(colors=np.ascontiguousarray(sc.sh[:, :(deg + 1) ** 2]), quats=sc.quats, scales=sc.scales, cov=None, cov6=None)
    rng = np.random.default_rng(55_000 + c["seed"])
    q = sc.quats / np.linalg.norm(sc.quats, axis=1, keepdims=True)
    w, x, y, z = q.T
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                  2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    M = R * sc.scales[:, None, :]
    cov = (M @ M.transpose(0, 2, 1)).astype(np.float32)
    cov = ((cov + cov.transpose(0, 2, 1)) * np.float32(0.5)).astype(np.float32)
    cov6 = np.ascontiguousarray(np.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], 1))
    colors = rng.uniform(0, 1, size=(sc.means.shape[0], 3)).astype(np.float32)
    if c["poisoned"]:
        flat = colors.reshape(-1)
        k = max(1, flat.size // 100)
        flat[rng.integers(0, flat.size, size=k)] = np.array([np.nan, np.inf, -np.inf, 1e30, -1e30], np.float32)[rng.integers(0, 5, size=k)]
    return dict(colors=colors, quats=None, scales=None, cov=cov, cov6=cov6)


def run_case(r, c: dict) -> list:
    """Renders the case on the GPU through its entry point and with the oracle view by view; returns the differences found."""
    sc, cams, W, H, fill, BG = c["scene"], c["cams"], c["W"], c["H"], c["fill"], c["bg"]
    inp = scene_inputs(c)
    gid = sc.group_id if c["n_groups"] else None
    r.upload(sc.means, sc.opacities, inp["colors"], quats=inp["quats"], scales=inp["scales"], covariances=inp["cov"], sh_degree=c["deg"],
             group_id=gid, n_groups=c["n_groups"])
    Vs, Ks = np.stack([cm.viewmat for cm in cams]), np.stack([cm.K for cm in cams])
    view_pose = [None] * len(cams)
    got = []
    if c["entry"] == "pipelined":
        return run_pipelined(r, c, inp, gid)
    if c["entry"] == "single":
        if c["poses"]:
            r.set_group_poses(c["poses"][0]); view_pose[0] = c["poses"][0]
        o = r.render(Vs[0], Ks[0], W, H, BG, want=KEYS, depth_fill_max=fill, full_sort=c["full_sort"])
        got.append({k: v.cpu().numpy() for k, v in o.items()})
    elif c["entry"] == "posed":
        o = r.render_batch(Vs, Ks, W, H, BG, want=KEYS, depth_fill_max=fill, pose_sets=np.stack(c["poses"]), pose_set=list(range(len(cams))))
        got = [{k: v[i].cpu().numpy() for k, v in o.items()} for i in range(len(cams))]
        view_pose = list(c["poses"])
    else:
        if c["poses"]:
            r.set_group_poses(c["poses"][0]); view_pose = [c["poses"][0]] * len(cams)
        if c["entry"] == "host":
            frames = r.render_batch_host(Vs, Ks, W, H, BG).numpy()
            got = [{"rgb8": frames[i]} for i in range(len(cams))]
        else:
            o = r.render_batch(Vs, Ks, W, H, BG, want=KEYS, depth_fill_max=fill)
            got = [{k: v[i].cpu().numpy() for k, v in o.items()} for i in range(len(cams))]
    diffs = []
    for i, cm in enumerate(cams):
        ref = oracle.render(sc.means, sc.opacities, inp["colors"], cm.viewmat, cm.K, W, H, quats=inp["quats"], scales=inp["scales"], cov6=inp["cov6"],
                            sh_degree=c["deg"], group_id=gid, group_Rt=view_pose[i], background=BG, depth_mode=1 if fill else 0, want_rgb8=True)
        for k, g in got[i].items():
            if not np.array_equal(g, ref[k], equal_nan=g.dtype != np.uint8):
                d = np.abs(g.astype(np.float64) - ref[k].astype(np.float64))
                diffs.append(f"view {i} {k}: {int((d > 0).sum())} values differ, max {d.max():.3e}")
        if c["entry"] == "single":
            st = r.stats()
            if st["n_visible"] != ref["n_visible"] or st["n_isect"] != ref["n_isect"]:
                diffs.append(f"counts: visible {st['n_visible']} / {ref['n_visible']}, intersections {st['n_isect']} / {ref['n_isect']}")
    return diffs


def run_pipelined(r, c: dict, inp: dict, gid) -> list:
    """c['steps'] steps enqueued back to back (block=False), each into its own output tensors and after its own set_group_poses; one
    wait at the end; then every frame of every step against the oracle."""
    import torch
    W, H, fill, sc, BG = c["W"], c["H"], c["fill"], c["scene"], c["bg"]
    outs = []
    for s_ in range(c["steps"]):
        row = c["step_cams"][s_]
        if c["poses"]:
            r.set_group_poses(c["poses"][s_])
        if len(row) == 1:
            o = r.render(row[0].viewmat, row[0].K, W, H, BG, want=KEYS, depth_fill_max=fill, block=False)
            outs.append({k: v[None] for k, v in o.items()})
        else:
            outs.append(r.render_batch(np.stack([cm.viewmat for cm in row]), np.stack([cm.K for cm in row]), W, H, BG, want=KEYS, depth_fill_max=fill,
                                       block=False))
    r.wait()
    torch.cuda.synchronize()
    diffs = []
    for s_ in range(c["steps"]):
        for i, cm in enumerate(c["step_cams"][s_]):
            ref = oracle.render(sc.means, sc.opacities, inp["colors"], cm.viewmat, cm.K, W, H, quats=inp["quats"], scales=inp["scales"], cov6=inp["cov6"],
                                sh_degree=c["deg"], group_id=gid, group_Rt=c["poses"][s_] if c["poses"] else None, background=BG,
                                depth_mode=1 if fill else 0, want_rgb8=True)
            for k in KEYS:
                g = outs[s_][k][i].cpu().numpy()
                if not np.array_equal(g, ref[k], equal_nan=g.dtype != np.uint8):
                    d = np.abs(g.astype(np.float64) - ref[k].astype(np.float64))
                    diffs.append(f"step {s_} view {i} {k}: {int((d > 0).sum())} values differ, max {d.max():.3e}")
    return diffs


def describe(c: dict) -> str:
    return (f"seed {c['seed']}: n={c['scene'].means.shape[0]} degree={c['deg']} groups={c['n_groups']} {c['W']}x{c['H']} views={len(c['cams'])} "
            f"entry={c['entry']}{'x%d' % c['steps'] if c['entry'] == 'pipelined' else ''} fill={c['fill']}{' full_sort' if c['full_sort'] else ''}{'' if c['bg'] is BG else ' bg=drawn'}{' POISONED' if c['poisoned'] else ''}")


def main(argv) -> int:
    from sim_a_splat_amd.rasterizer import Rasterizer
    n_seeds = int(argv[1]) if len(argv) > 1 else 60
    first = int(argv[2]) if len(argv) > 2 else 0
    poison_all = len(argv) > 3 and argv[3] == "poison-all"
    r = Rasterizer(0)
    bad = 0
    for seed in range(first, first + n_seeds):
        c = draw_case(seed, poison_all)
        diffs = run_case(r, c)
        st = r.stats()
        print(describe(c), f"max_list={st['max_tile_len']} ->", "bit-equal" if not diffs else "DIFFERENT: " + "; ".join(diffs), flush=True)
        bad += bool(diffs)
    from sim_a_splat_amd import _capi
    L = _capi.lib()
    if hasattr(L, "sas_debug_bounds"):      # the bounds-checked build (SAS_LIB_PATH=variants/lib_bounds.so): every computed index was range-checked
        import ctypes
        out = (ctypes.c_uint64 * 4)()
        L.sas_debug_bounds.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.sas_debug_bounds(out, 0)
        print(f"bounds-checked build: {out[0]} out-of-range accesses" + (f" (first: code {out[1]}, index {out[2]}, limit {out[3]})" if out[0] else ""))
        bad += int(out[0] != 0)
    r.close()
    print(f"{n_seeds} cases from seed {first}: " + ("every output bit-equal to the oracle" if bad == 0 else f"{bad} cases differ"))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
