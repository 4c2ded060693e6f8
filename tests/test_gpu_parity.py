"""GPU: the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs.

The arithmetic contract makes the two bit-identical; the asserted bar is the north-star one
(max-abs <= 1e-4 per channel), and bit-equality is asserted separately so that a contract
slip shows up as its own failure.
"""
import numpy as np
import pytest

import oracle
from conftest import TWIN_CASES, load_twin_fixture, twin_scene_kwargs
from sim_a_splat_amd.synthetic import (NERFSTUDIO_EVAL_BACKGROUND, config_scene_and_cameras, make_scene,
                                       random_group_poses, ring_camera)

pytestmark = pytest.mark.gpu
BG = NERFSTUDIO_EVAL_BACKGROUND
TOL = 1e-4


def _upload(r, sc, **kw):
    r.upload(sc.means, sc.opacities, sc.sh, quats=sc.quats, scales=sc.scales, sh_degree=sc.sh_degree,
             group_id=kw.get("group_id"), n_groups=kw.get("n_groups", 0))


def _compare(r, sc, cam, group_id=None, group_Rt=None, exact=True, depth_fill=False, full_sort=False):
    out = r.render(cam.viewmat, cam.K, cam.width, cam.height, BG, want=("rgb", "alpha", "depth", "rgb8"),
                   depth_fill_max=depth_fill, full_sort=full_sort)
    ref = oracle.render(sc.means, sc.opacities, sc.sh, cam.viewmat, cam.K, cam.width, cam.height, quats=sc.quats,
                        scales=sc.scales, sh_degree=sc.sh_degree, group_id=group_id, group_Rt=group_Rt,
                        background=BG, depth_mode=1 if depth_fill else 0, want_rgb8=True, dump=True)
    got = {k: v.cpu().numpy() for k, v in out.items()}
    st = r.stats()
    assert st["n_visible"] == ref["n_visible"]
    assert st["n_isect"] == ref["n_isect"]
    assert np.abs(got["rgb"] - ref["rgb"]).max() <= TOL
    assert np.abs(got["alpha"] - ref["alpha"]).max() <= TOL
    m = ref["alpha"] > 0.5
    if m.any():
        assert (np.abs(got["depth"] - ref["depth"])[m] / ref["depth"][m]).max() <= 1e-3
    assert np.abs(got["rgb8"].astype(int) - ref["rgb8"].astype(int)).max() <= 1
    if exact:
        for k in ("rgb", "alpha", "depth", "rgb8"):
            assert np.array_equal(got[k], ref[k]), f"{k} not bit-identical: max diff {np.abs(got[k].astype(np.float64) - ref[k]).max()}"
    return got, ref


def _check_crop_against_twin(got, cfg, keys=("rgb", "alpha")):
    """The committed float64-twin window of a BASELINE config (oracle/make_golden.py gen_cfg_crop) against the HIP frame."""
    from conftest import GOLDEN
    g = np.load(GOLDEN / f"render_twin_cfg{cfg}_crop.npz")
    tx0, ty0, tx1, ty1 = [int(v) for v in g["crop_tiles"]]
    win = (slice(16 * ty0, 16 * ty1), slice(16 * tx0, 16 * tx1))
    for k in keys:
        d = np.abs(got[k][win] - g[k]).max(axis=2)
        assert int((d > TOL).sum()) <= 3, (k, int((d > TOL).sum()), float(d.max()))
        assert float(d[d <= TOL].max()) <= 5e-5
        print(f"config {cfg} crop, {k}: {int((d > TOL).sum())} of {d.size} pixels beyond {TOL} against the float64 twin (max {d.max():.1e})")


def test_projection_arrays_bit_exact(rasterizer):
    sc = make_scene(5000, seed=21, log_scale_mean=float(np.log(0.03)))
    cam = ring_camera(200, 120, 150.0, yaw_deg=15.0, elev=0.4)
    _upload(rasterizer, sc)
    _compare(rasterizer, sc, cam)                              # production path: lazy per-tile ordering
    _, ref = _compare(rasterizer, sc, cam, full_sort=True)     # full path: complete lists kept (T4/T5)
    p = rasterizer.read_projection()
    vis = (ref["radii"] > 0).all(axis=1)
    assert np.array_equal(p["radii"], ref["radii"])
    for k in ("means2d", "depths", "conics", "colors"):
        assert np.array_equal(p[k][vis], ref[k][vis]), k
    tl = rasterizer.read_tile_lists(cam.tiles)
    assert np.array_equal(tl["tile_offsets"], ref["tile_offsets"])
    assert np.array_equal(tl["sorted_ids"], ref["sorted_ids"])


@pytest.mark.parametrize("n,w,h,f,ls", [(1, 32, 32, 40.0, 0.15), (2, 48, 32, 40.0, 0.2), (64, 64, 64, 60.0, 0.08),
                                        (2000, 128, 96, 120.0, 0.03), (3000, 250, 130, 200.0, 0.05)])
def test_small_scenes(rasterizer, n, w, h, f, ls):
    sc = make_scene(n, seed=100 + n, log_scale_mean=float(np.log(ls)))
    cam = ring_camera(w, h, f, yaw_deg=20.0, elev=0.3)
    _upload(rasterizer, sc)
    _compare(rasterizer, sc, cam, depth_fill=(n == 64))


def test_config1_10k_256(rasterizer):
    sc, cams = config_scene_and_cameras(1)
    _upload(rasterizer, sc)
    _compare(rasterizer, sc, cams[0])


def test_group_poses_dynamic_scene(rasterizer):
    sc = make_scene(4000, seed=31, log_scale_mean=float(np.log(0.03)), n_groups=7)
    cam = ring_camera(160, 120, 130.0)
    _upload(rasterizer, sc, group_id=sc.group_id, n_groups=7)
    ident = random_group_poses(7, seed=0, max_angle=0.0, max_shift=0.0)
    _compare(rasterizer, sc, cam, group_id=sc.group_id, group_Rt=ident)
    for step in range(2):
        Rt = random_group_poses(7, seed=50 + step)
        rasterizer.set_group_poses(Rt)
        _compare(rasterizer, sc, cam, group_id=sc.group_id, group_Rt=Rt)


def test_empty_and_fully_culled_scenes(rasterizer):
    sc = make_scene(16, seed=5)
    cam = ring_camera(40, 24, 50.0)
    # everything behind the camera
    sc.means[:, 2] += 100.0
    _upload(rasterizer, sc)
    got, _ = _compare(rasterizer, sc, cam)
    assert np.allclose(got["rgb"], np.array(BG, np.float32)) and np.all(got["alpha"] == 0)
    # N = 0
    rasterizer.upload(np.zeros((0, 3), np.float32), np.zeros((0,), np.float32), np.zeros((0, 16, 3), np.float32),
                      quats=np.zeros((0, 4), np.float32), scales=np.zeros((0, 3), np.float32))
    out = rasterizer.render(cam.viewmat, cam.K, cam.width, cam.height, BG, want=("rgb", "alpha"))
    assert np.allclose(out["rgb"].cpu().numpy(), np.array(BG, np.float32))


def test_screen_filling_and_dense_tile(rasterizer):
    """A few huge Gaussians (wave-cooperative tile walk) plus >4096 splats in one tile (global sort path)."""
    rng = np.random.default_rng(9)
    sc = make_scene(6000, seed=41, log_scale_mean=float(np.log(0.004)))
    sc.means[:] = rng.normal(0, 0.01, size=sc.means.shape).astype(np.float32)   # all in the centre tile
    sc.scales[:8] = 0.8                                                          # screen-filling
    sc.opacities[:] = np.clip(sc.opacities, 0.02, 0.2)                           # keep transmittance alive
    cam = ring_camera(96, 80, 100.0)
    _upload(rasterizer, sc)
    _compare(rasterizer, sc, cam)
    assert rasterizer.stats()["max_tile_len"] > 4096


def test_sh_degrees_and_direct_rgb(rasterizer):
    base = make_scene(1500, seed=51, log_scale_mean=float(np.log(0.04)))
    cam = ring_camera(120, 90, 100.0)
    for deg in (0, 1, 2):
        kk = (deg + 1) ** 2
        sh = np.ascontiguousarray(base.sh[:, :kk])
        rasterizer.upload(base.means, base.opacities, sh, quats=base.quats, scales=base.scales, sh_degree=deg)
        out = rasterizer.render(cam.viewmat, cam.K, cam.width, cam.height, BG, want=("rgb",))
        ref = oracle.render(base.means, base.opacities, sh, cam.viewmat, cam.K, cam.width, cam.height,
                            quats=base.quats, scales=base.scales, sh_degree=deg, background=BG)
        assert np.array_equal(out["rgb"].cpu().numpy(), ref["rgb"]), deg
    # Door-B form: covariances + final RGB
    from oracle import ref_math
    q = base.quats / np.linalg.norm(base.quats, axis=1, keepdims=True)
    R = np.stack([ref_math.quat_wxyz_to_R(x) for x in q]).astype(np.float32)
    M = R * base.scales[:, None, :]
    cov = (M @ M.transpose(0, 2, 1)).astype(np.float32)
    rgb = np.clip(ref_math.sh2rgb(base.sh[:, 0]), 0, 1).astype(np.float32)
    rasterizer.upload(base.means, base.opacities, rgb, covariances=cov, sh_degree=-1)
    out = rasterizer.render(cam.viewmat, cam.K, cam.width, cam.height, BG, want=("rgb", "rgb8"))
    cov6 = np.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], 1)
    ref = oracle.render(base.means, base.opacities, rgb, cam.viewmat, cam.K, cam.width, cam.height, cov6=cov6,
                        sh_degree=-1, background=BG, want_rgb8=True)
    assert np.array_equal(out["rgb"].cpu().numpy(), ref["rgb"])
    assert np.array_equal(out["rgb8"].cpu().numpy(), ref["rgb8"])


def test_fast_exp_delta_is_small_but_not_the_contract(rasterizer):
    sc = make_scene(3000, seed=61, log_scale_mean=float(np.log(0.04)))
    cam = ring_camera(160, 120, 130.0)
    _upload(rasterizer, sc)
    a = rasterizer.render(cam.viewmat, cam.K, cam.width, cam.height, BG, want=("rgb",))["rgb"].cpu().numpy()
    b = rasterizer.render(cam.viewmat, cam.K, cam.width, cam.height, BG, want=("rgb",), fast_exp=True)["rgb"].cpu().numpy()
    d = np.abs(a - b)
    # v_exp_f32 differs in the last bits; away from threshold flips the image moves by < 1e-5
    assert np.quantile(d, 0.999) < 1e-5


def test_depth_ties_order_by_caller_index(rasterizer):
    """Thousands of splats on a few constant-depth planes: the sort sees long runs of identical depth
    bits and must order them by the caller's index (stable radix order of the reference)."""
    rng = np.random.default_rng(3)
    sc = make_scene(6000, seed=71, log_scale_mean=float(np.log(0.05)))
    sc.means[:, 2] = rng.choice(np.array([0.0, 0.25, 0.5], np.float32), size=sc.n)   # camera looks down -z
    sc.means[:, :2] *= 0.4
    sc.opacities[:] = np.clip(sc.opacities, 0.05, 0.5)
    cam = ring_camera(128, 96, 110.0)          # yaw 0: depth = 3 - z exactly
    _upload(rasterizer, sc)
    _compare(rasterizer, sc, cam)
    _, ref = _compare(rasterizer, sc, cam, full_sort=True)
    assert len(np.unique(ref["depths"][(ref["radii"] > 0).all(1)])) <= 3
    tl = rasterizer.read_tile_lists(cam.tiles)
    assert np.array_equal(tl["sorted_ids"], ref["sorted_ids"])


def test_mid_size_lists_use_the_large_lds_class(rasterizer):
    """Per-tile lists between 2048 and 8192 entries (second sort class)."""
    rng = np.random.default_rng(10)
    sc = make_scene(5000, seed=81, log_scale_mean=float(np.log(0.004)))
    sc.means[:] = rng.normal(0, 0.012, size=sc.means.shape).astype(np.float32)
    sc.opacities[:] = np.clip(sc.opacities, 0.02, 0.15)
    cam = ring_camera(96, 80, 100.0)
    _upload(rasterizer, sc)
    _compare(rasterizer, sc, cam)
    assert 2048 < rasterizer.stats()["max_tile_len"] <= 8192


def test_more_than_8192_tiles(rasterizer):
    """2100x1300 = 132x82 tiles: the tile scan needs a second round; ragged right/bottom tiles."""
    sc = make_scene(3000, seed=91, log_scale_mean=float(np.log(0.02)))
    cam = ring_camera(2100, 1300, 1100.0, yaw_deg=5.0)
    _upload(rasterizer, sc)
    _compare(rasterizer, sc, cam)


def test_list_longer_than_every_lds_class(rasterizer):
    """> 16384 splats in one tile: the in-place global fallback of the large sort class."""
    rng = np.random.default_rng(12)
    sc = make_scene(20000, seed=95, log_scale_mean=float(np.log(0.003)))
    sc.means[:] = rng.normal(0, 0.004, size=sc.means.shape).astype(np.float32)
    sc.opacities[:] = np.clip(sc.opacities, 0.01, 0.03)
    cam = ring_camera(64, 48, 100.0)
    _upload(rasterizer, sc)
    _compare(rasterizer, sc, cam)
    assert rasterizer.stats()["max_tile_len"] > 16384


def test_every_sort_class_in_one_frame(rasterizer):
    """Dense centre falling off outwards: tiles of < 1024, 1024..4095 and >= 4096 entries together."""
    rng = np.random.default_rng(13)
    sc = make_scene(60000, seed=96, log_scale_mean=float(np.log(0.004)))
    sc.means[:] = rng.normal(0, 0.3, size=sc.means.shape).astype(np.float32)
    sc.opacities[:] = np.clip(sc.opacities, 0.01, 0.1)
    cam = ring_camera(160, 128, 260.0)
    _upload(rasterizer, sc)
    _, ref = _compare(rasterizer, sc, cam)
    lens = np.diff(ref["tile_offsets"])
    assert (lens >= 4096).any() and ((lens >= 1024) & (lens < 4096)).any() and ((lens > 1) & (lens < 1024)).any()


def test_lazy_and_full_paths_agree_and_fallback_is_exercised(rasterizer):
    """20000 coplanar splats: every tile has far more than 1024 entries in ONE depth bucket, so the lazy
    kernel hands the tiles to the full path; the image must not change, with or without SAS_FULL_SORT."""
    rng = np.random.default_rng(21)
    sc = make_scene(20000, seed=97, log_scale_mean=float(np.log(0.02)))
    sc.means[:, 2] = 0.0
    sc.means[:, :2] = rng.uniform(-0.5, 0.5, size=(sc.n, 2)).astype(np.float32)
    sc.opacities[:] = np.clip(sc.opacities, 0.01, 0.05)
    cam = ring_camera(96, 64, 90.0)
    _upload(rasterizer, sc)
    a, _ = _compare(rasterizer, sc, cam)
    assert rasterizer.stats()["fallback_tiles"] > 0
    b, _ = _compare(rasterizer, sc, cam, full_sort=True)
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_lazy_multi_round_tiles(rasterizer):
    """Long lists of nearly transparent splats: nothing saturates, so the lazy kernel has to walk
    every depth bucket range of every tile (many rounds), still bit-exact."""
    rng = np.random.default_rng(22)
    sc = make_scene(52000, seed=98, log_scale_mean=float(np.log(0.01)))   # (lists beyond 4096 after the exact tile culling too)
    sc.means[:] = rng.uniform(-0.4, 0.4, size=sc.means.shape).astype(np.float32)
    sc.opacities[:] = np.clip(sc.opacities, 0.004, 0.01)
    cam = ring_camera(80, 64, 110.0)
    _upload(rasterizer, sc)
    _compare(rasterizer, sc, cam)
    st = rasterizer.stats()
    assert st["max_tile_len"] > 4096 and st["fallback_tiles"] == 0


# ---- BASELINE.json full-size configurations -----------------------------------------------------
@pytest.mark.parametrize("cfg", [2, 3])
def test_full_size_config_against_oracle(rasterizer, cfg):
    """config 2 (292,247 Gaussians, 640x480) and config 3 (1M Gaussians, 1920x1080), full size."""
    sc, cams = config_scene_and_cameras(cfg)
    _upload(rasterizer, sc)
    got, ref = _compare(rasterizer, sc, cams[0], depth_fill=True)
    assert ref["n_isect"] > 1_000_000
    # size-independent properties: the two ordering strategies give the same frame, and the frame
    # is a convex blend of splat colours and background (alpha in [0,1], empty pixels = background)
    full = rasterizer.render(cams[0].viewmat, cams[0].K, cams[0].width, cams[0].height, BG, want=("rgb", "alpha"),
                             full_sort=True)
    assert np.array_equal(full["rgb"].cpu().numpy(), got["rgb"])
    a = got["alpha"]
    assert a.min() >= 0.0 and a.max() <= 1.0
    empty = a[..., 0] == 0
    assert empty.any() and np.array_equal(got["rgb"][empty], np.broadcast_to(np.array(BG, np.float32), got["rgb"][empty].shape))
    # the float64 twin on this config itself (tests/golden/render_twin_cfg{2,3}_crop.npz: 8 x 6 tiles on the densest part of
    # this view -- config 3: lists of 2 565 .. 5 675 entries): the HIP frame's window within the north_star tolerance of it
    # on every pixel but the threshold flips (at most 3 of 12 288; tests/test_oracle.py shows what a flip is)
    _check_crop_against_twin(got, cfg)
    if cfg == 3:
        # the bench's step at full size: a view pair through sas_render_batch (paired by default at this
        # scene size); its first view is the frame just checked against the oracle
        import torch
        cam2 = ring_camera(cams[0].width, cams[0].height, float(cams[0].K[0, 0]), yaw_deg=180.0)
        pair = rasterizer.render_batch(np.stack([cams[0].viewmat, cam2.viewmat]), np.stack([cams[0].K, cam2.K]),
                                       cams[0].width, cams[0].height, BG, want=("rgb",))["rgb"]
        alone = rasterizer.render(cam2.viewmat, cam2.K, cam2.width, cam2.height, BG, want=("rgb",))["rgb"]
        assert np.array_equal(pair[0].cpu().numpy(), got["rgb"]) and torch.equal(pair[1], alone)


def test_background_enters_linearly_at_full_size(rasterizer):
    """rgb(bg) - rgb(0) == (1 - alpha) * bg wherever nothing clamps; alpha does not depend on bg."""
    sc, cams = config_scene_and_cameras(3)
    cam = cams[0]
    _upload(rasterizer, sc)
    r0 = rasterizer.render(cam.viewmat, cam.K, cam.width, cam.height, (0.0, 0.0, 0.0), want=("rgb", "alpha"))
    r0 = {k: v.cpu().numpy() for k, v in r0.items()}
    r1 = rasterizer.render(cam.viewmat, cam.K, cam.width, cam.height, (0.25, 0.5, 0.125), want=("rgb", "alpha"))
    r1 = {k: v.cpu().numpy() for k, v in r1.items()}
    assert np.array_equal(r0["alpha"], r1["alpha"])
    w = (np.float32(1.0) - r0["alpha"]) * np.array([0.25, 0.5, 0.125], np.float32)
    free = (r1["rgb"] < 1.0) & (r0["rgb"] > 0.0)
    assert free.mean() > 0.5
    assert np.abs((r1["rgb"] - r0["rgb"]) - w)[free].max() <= 2e-7 * 4


def test_config5_all_four_views_5m_gaussians(rasterizer):
    """Config 5 (5M Gaussians, four 1080p views): long lists, every sort class, HBM-heavy projection.  One view with
    the full set of outputs, then all four as one batch (two view pairs) against the oracle."""
    sc, cams = config_scene_and_cameras(5)
    _upload(rasterizer, sc)
    _compare(rasterizer, sc, cams[1])
    st = rasterizer.stats()
    assert st["n_isect"] > 10_000_000 and st["max_tile_len"] > 8192   # (23 500 in the ordinary layout, 13 100 in culled 8-pixel tiles)
    batch = rasterizer.render_batch(np.stack([c.viewmat for c in cams]), np.stack([c.K for c in cams]), 1920, 1080, BG, want=("rgb",))
    for v, cam in enumerate(cams):
        ref = oracle.render_scene(sc, cam, background=BG)
        assert np.array_equal(batch["rgb"][v].cpu().numpy(), ref["rgb"]), v
    # ... and view 0 against the float64 twin on its densest 6 x 4 tiles (lists of 19 845 .. 28 317 entries)
    _check_crop_against_twin({"rgb": batch["rgb"][0].cpu().numpy()}, 5, keys=("rgb",))


@pytest.mark.parametrize("name", TWIN_CASES)
def test_golden_twin_fixtures_through_the_c_abi(rasterizer, name):
    """The committed float64-twin fixtures (tests/golden/render_twin_*.npz: textbook formulas, libm exp, an
    independent implementation) rendered by the HIP path through the C ABI and compared with the stored
    frames DIRECTLY, at the north_star tolerance: rgb / alpha <= 1e-4 max abs, depth <= 1e-3 relative.
    Covers lists of thousands with several lazy chunks and early termination ("dense"), Door B's cov6 +
    final RGB + group poses ("doorb") and a camera inside the cloud ("inside").  The HIP frame must also
    equal the C oracle bit for bit, so the two pins cannot drift apart."""
    g = load_twin_fixture(name)
    means, op, colors, kw = twin_scene_kwargs(g)
    W, H = [int(v) for v in g["wh"]]
    up = dict(kw)
    cov6 = up.pop("cov6", None)
    gid = up.pop("group_id", None)
    rasterizer.upload(means, op, colors, quats=up.get("quats"), scales=up.get("scales"), covariances=cov6,
                      sh_degree=up["sh_degree"], group_id=gid, n_groups=int(g["group_Rt"].shape[0]) if gid is not None else 0)
    if gid is not None:
        rasterizer.set_group_poses(g["group_Rt"])
    bg = tuple(float(v) for v in g["background"])
    out = rasterizer.render(g["viewmat"], g["K"], W, H, bg, want=("rgb", "alpha", "depth"))
    rgb, alpha, depth = (out[k].cpu().numpy() for k in ("rgb", "alpha", "depth"))
    assert np.abs(rgb - g["rgb"]).max() <= TOL
    assert np.abs(alpha - g["alpha"]).max() <= TOL
    solid = g["alpha"] > 0.5
    if solid.any():
        assert (np.abs(depth - g["depth"])[solid] / g["depth"][solid]).max() <= 1e-3
    st = rasterizer.stats()
    ref = oracle.render(means, op, colors, g["viewmat"], g["K"], W, H, group_Rt=g["group_Rt"] if gid is not None else None,
                        background=bg, **kw)
    assert ref["n_isect"] == int(g["n_isect"]) == st["n_isect"]
    assert st["n_visible"] == int(g["valid"].sum())
    rasterizer.render(g["viewmat"], g["K"], W, H, bg, want=("rgb",), full_sort=True)
    assert rasterizer.stats()["n_isect"] == int(g["n_isect"])
    assert np.array_equal(rgb, ref["rgb"]) and np.array_equal(alpha, ref["alpha"]) and np.array_equal(depth, ref["depth"])


def test_async_frames_and_stream_ordering(rasterizer):
    """SAS_ASYNC: up to four frames in flight on internal streams.  Work put on the caller's stream
    must see every COMPLETED frame (sas_frames_completed) without a host wait; wait() completes all."""
    import torch
    sc = make_scene(20000, seed=111, log_scale_mean=float(np.log(0.02)))
    _upload(rasterizer, sc)
    cams = [ring_camera(320, 240, 260.0, yaw_deg=20.0 * k, elev=0.1 * k) for k in range(6)]
    sync = [rasterizer.render(c.viewmat, c.K, c.width, c.height, BG, want=("rgb",))["rgb"].clone() for c in cams]
    bufs = [{"rgb": torch.empty((240, 320, 3), dtype=torch.float32, device="cuda:0")} for _ in cams]
    snaps = []
    base = rasterizer.frames_completed()[1]
    for i, c in enumerate(cams):
        rasterizer.render(c.viewmat, c.K, c.width, c.height, BG, want=("rgb",), out=bufs[i], block=False)
        while len(snaps) < rasterizer.frames_completed()[1] - base:
            snaps.append(bufs[len(snaps)]["rgb"].clone())  # stream-ordered consumer of a completed frame
    assert 1 <= len(snaps) < len(cams)                     # frames complete while later ones are in flight
    rasterizer.wait()
    while len(snaps) < len(cams):
        snaps.append(bufs[len(snaps)]["rgb"].clone())
    torch.cuda.synchronize()
    for i in range(len(cams)):
        assert torch.equal(snaps[i], sync[i]), i
        assert torch.equal(bufs[i]["rgb"], sync[i]), i


def test_config4_eight_poses_uint8_batched(rasterizer):
    """config 4 shape: the ~300k-Gaussian stand-in scene with 7 link groups, 8 ring poses at 640x480,
    uint8 frames (Door B form), rendered as one batch; every view against the oracle."""
    sc, cams = config_scene_and_cameras(4)
    _upload(rasterizer, sc, group_id=sc.group_id, n_groups=7)
    Rt = random_group_poses(7, seed=404)
    rasterizer.set_group_poses(Rt)
    out = rasterizer.render_batch(np.stack([c.viewmat for c in cams]), np.stack([c.K for c in cams]), 640, 480, BG,
                                  want=("rgb8", "rgb"))
    assert out["rgb8"].shape == (8, 480, 640, 3)
    got8, got = out["rgb8"].cpu().numpy(), out["rgb"].cpu().numpy()
    for v in (0, 3, 7):
        ref = oracle.render(sc.means, sc.opacities, sc.sh, cams[v].viewmat, cams[v].K, 640, 480, quats=sc.quats,
                            scales=sc.scales, sh_degree=3, group_id=sc.group_id, group_Rt=Rt, background=BG, want_rgb8=True)
        assert np.abs(got[v] - ref["rgb"]).max() <= TOL
        assert np.array_equal(got8[v], ref["rgb8"]) and np.array_equal(got[v], ref["rgb"])


def test_view_pairs_equal_single_views(monkeypatch):
    """sas_render_batch projects two views per pass over the scene (by default for scenes of 0.5 M
    Gaussians and more; forced here): every output of every view of an odd-sized batch (pairs + one
    single), one of them looking away from the scene, equals the one-view-at-a-time render; an
    asynchronous batch keeps the stream-ordering contract."""
    import torch
    from sim_a_splat_amd.rasterizer import Rasterizer
    monkeypatch.setenv("SAS_PAIR", "1")
    rasterizer = Rasterizer("cuda:0")
    sc = make_scene(30000, seed=222, log_scale_mean=float(np.log(0.02)))
    _upload(rasterizer, sc)
    cams = [ring_camera(300, 200, 240.0, yaw_deg=y, elev=e) for y, e in ((0.0, 0.0), (75.0, 0.3), (200.0, -0.2))]
    away = cams[1].viewmat.copy()
    away[:3, :3] = np.diag([-1.0, 1.0, -1.0]).astype(np.float32) @ away[:3, :3]      # turn the camera around
    away[:3, 3] = np.diag([-1.0, 1.0, -1.0]).astype(np.float32) @ away[:3, 3]
    Vs = np.stack([cams[0].viewmat, away, cams[2].viewmat])
    Ks = np.stack([c.K for c in cams])
    singles = [{k: v.clone() for k, v in rasterizer.render(Vs[i], Ks[i], 300, 200, BG, want=("rgb", "alpha", "depth", "rgb8"),
                                                            depth_fill_max=True).items()} for i in range(3)]
    assert float(singles[1]["alpha"].max()) == 0.0 and float(singles[0]["alpha"].max()) > 0.5
    batch = rasterizer.render_batch(Vs, Ks, 300, 200, BG, want=("rgb", "alpha", "depth", "rgb8"), depth_fill_max=True)
    for i in range(3):
        for k in ("rgb", "alpha", "depth", "rgb8"):
            assert torch.equal(batch[k][i], singles[i][k]), (i, k)
    # asynchronous pairs: work on the caller's stream is ordered behind every COMPLETED frame
    outs = [{"rgb": torch.empty((2, 200, 300, 3), device="cuda:0")} for _ in range(5)]
    snaps, taken = [], 0
    base = rasterizer.frames_completed()[1]
    for j in range(5):
        rasterizer.render_batch(Vs[[0, 2]], Ks[[0, 2]], 300, 200, BG, want=("rgb",), out=outs[j], block=False)
        done_batches = (rasterizer.frames_completed()[1] - base) // 2
        while taken < done_batches:
            snaps.append(outs[taken]["rgb"].clone())   # stream-ordered consumer, no host synchronisation
            taken += 1
    assert 1 <= taken < 5                              # four slots: batches complete while later ones are in flight
    rasterizer.wait()
    sub, com = rasterizer.frames_completed()
    assert sub == com == base + 10
    for sn in snaps + [o["rgb"] for o in outs]:
        assert torch.equal(sn[0], singles[0]["rgb"]) and torch.equal(sn[1], singles[2]["rgb"])
    rasterizer.close()


@pytest.mark.parametrize("group", [2, 4])
def test_launch_groups_equal_single_views(monkeypatch, group):
    """sas_render_batch on a small scene renders the views in launch groups (one projection / scan / scatter /
    tile launch with grid.y = view, enqueue_group): every output of every view of a 5-view batch (full groups +
    a remainder), one view looking away from the scene, equals the one-view-at-a-time render; an overflowing
    group is rendered again as a group; asynchronous groups complete as a whole."""
    import torch
    from sim_a_splat_amd.rasterizer import Rasterizer
    monkeypatch.setenv("SAS_GROUP", str(group))
    r = Rasterizer("cuda:0")
    sc = make_scene(20000, seed=555, log_scale_mean=float(np.log(0.025)), n_groups=4)
    _upload(r, sc, group_id=sc.group_id, n_groups=4)
    r.set_group_poses(random_group_poses(4, seed=5))
    cams = [ring_camera(200, 136, 170.0, yaw_deg=y, elev=e) for y, e in ((0.0, 0.0), (70.0, 0.3), (140.0, -0.2), (250.0, 0.1), (300.0, 0.4))]
    Vs, Ks = np.stack([c.viewmat for c in cams]), np.stack([c.K for c in cams])
    Vs[3, :3, :3] = np.diag([-1.0, 1.0, -1.0]).astype(np.float32) @ Vs[3, :3, :3]      # turn camera 3 around: empty frame
    Vs[3, :3, 3] = np.diag([-1.0, 1.0, -1.0]).astype(np.float32) @ Vs[3, :3, 3]
    want = ("rgb", "alpha", "depth", "rgb8")
    singles = [{k: v.clone() for k, v in r.render(Vs[i], Ks[i], 200, 136, BG, want=want, depth_fill_max=True).items()} for i in range(5)]
    assert float(singles[3]["alpha"].max()) == 0.0 and float(singles[0]["alpha"].max()) > 0.5
    batch = r.render_batch(Vs, Ks, 200, 136, BG, want=want, depth_fill_max=True)
    for i in range(5):
        for k in want:
            assert torch.equal(batch[k][i], singles[i][k]), (i, k)
    # asynchronous groups: frames complete group by group
    base = r.frames_completed()[1]
    outs = [{"rgb8": torch.empty((4, 136, 200, 3), dtype=torch.uint8, device="cuda:0")} for _ in range(3)]
    for j in range(3):
        r.render_batch(Vs[:4], Ks[:4], 200, 136, BG, want=("rgb8",), out=outs[j], block=False)
        assert (r.frames_completed()[1] - base) % min(group, 4) == 0
    r.wait()
    assert r.frames_completed()[1] - base == 12
    for o in outs:
        for i in range(4):
            assert torch.equal(o["rgb8"][i], singles[i]["rgb8"]), i
    r.close()
    # overflow inside a group: fresh buffers (2^20 keys), every view needs more
    r = Rasterizer("cuda:0")
    big = make_scene(30000, seed=444, log_scale_mean=float(np.log(0.12)))
    _upload(r, big)
    bc = [ring_camera(640, 480, 500.0, yaw_deg=90.0 * k) for k in range(2)]
    refs = [oracle.render_scene(big, c_, background=BG) for c_ in bc]
    assert min(ref["n_isect"] for ref in refs) > (1 << 20)
    out = r.render_batch(np.stack([c_.viewmat for c_ in bc]), np.stack([c_.K for c_ in bc]), 640, 480, BG, want=("rgb",))
    assert r.stats()["regrows"] >= 1
    for v in range(2):
        assert np.array_equal(out["rgb"][v].cpu().numpy(), refs[v]["rgb"]), v
    r.close()


def test_group_pose_updates_between_frames(rasterizer):
    """Per-step set_group_poses between frames (the Gym loop: poses, then one blocking render per camera):
    the poses travel through a pinned staging block on each frame's own stream, and every frame slot is
    reused several times.  Frames follow the poses of their step."""
    sc = make_scene(20000, seed=333, log_scale_mean=float(np.log(0.03)), n_groups=5)
    _upload(rasterizer, sc, group_id=sc.group_id, n_groups=5)
    cams = [ring_camera(160, 120, 130.0, yaw_deg=0.0), ring_camera(160, 120, 130.0, yaw_deg=70.0, elev=0.4)]
    last = None
    for step in range(12):                       # every frame slot is used several times
        Rt = random_group_poses(5, seed=step)
        rasterizer.set_group_poses(Rt)
        last = [rasterizer.render(c.viewmat, c.K, 160, 120, BG, want=("rgb8", "rgb"))["rgb"].cpu().numpy() for c in cams]
    for c, got in zip(cams, last):
        ref = oracle.render(sc.means, sc.opacities, sc.sh, c.viewmat, c.K, 160, 120, quats=sc.quats, scales=sc.scales,
                            sh_degree=3, group_id=sc.group_id, group_Rt=random_group_poses(5, seed=11), background=BG)
        assert np.array_equal(got, ref["rgb"])


@pytest.mark.parametrize("direct", ["1", "0"])
def test_intersection_buffer_regrows_for_single_views_and_pairs(monkeypatch, direct):
    """More intersections than the initial buffer holds -- a tile longer than its segment (single-pass binning, the
    product path: 2 048 keys per tile to start with here) or more keys than max(8 N, 2^20) in all (two-pass binning,
    SAS_DIRECT=0): the frame is detected as overflowed from its stats, the buffer grows to the measured need and the
    frame is rendered again -- for a blocking single view, and for both views of an asynchronous pair."""
    import torch
    from sim_a_splat_amd.rasterizer import Rasterizer
    monkeypatch.setenv("SAS_PAIR", "1")
    monkeypatch.setenv("SAS_DIRECT", direct)
    r = Rasterizer("cuda:0")
    sc = make_scene(30000, seed=444, log_scale_mean=float(np.log(0.12)))
    _upload(r, sc)
    cams = [ring_camera(640, 480, 500.0, yaw_deg=0.0), ring_camera(640, 480, 500.0, yaw_deg=90.0)]
    refs = [oracle.render_scene(sc, c, background=BG) for c in cams]
    assert min(ref["n_isect"] for ref in refs) > (1 << 20)
    got = r.render(cams[0].viewmat, cams[0].K, 640, 480, BG, want=("rgb",))["rgb"].cpu().numpy()
    st = r.stats()
    assert st["regrows"] >= 1 and st["capacity"] >= st["n_isect"] == refs[0]["n_isect"]
    assert np.array_equal(got, refs[0]["rgb"])
    r.close()
    r = Rasterizer("cuda:0")                     # fresh buffers: now the pair overflows, both slots
    _upload(r, sc)
    out = r.render_batch(np.stack([c.viewmat for c in cams]), np.stack([c.K for c in cams]), 640, 480, BG, want=("rgb",),
                         block=False)
    r.wait()
    assert r.stats()["regrows"] >= 2
    for v in range(2):
        assert np.array_equal(out["rgb"][v].cpu().numpy(), refs[v]["rgb"]), v
    r.close()


def test_stream_ordered_consumer_of_overflowing_async_frames():
    """An asynchronous sequence whose every frame slot overflows its intersection buffer on first use:
    a consumer on the caller's stream (a device-side copy, no host synchronisation) that takes each
    frame as soon as sas_frames_completed reports it must see the final frame, never the truncated
    first attempt, and the re-render must not race with it."""
    import torch
    from sim_a_splat_amd.rasterizer import Rasterizer
    r = Rasterizer("cuda:0")
    sc = make_scene(30000, seed=444, log_scale_mean=float(np.log(0.12)))
    _upload(r, sc)
    cams = [ring_camera(640, 480, 500.0, yaw_deg=90.0 * (k % 2)) for k in range(7)]
    refs = [oracle.render_scene(sc, c, background=BG) for c in cams[:2]]
    assert min(ref["n_isect"] for ref in refs) > (1 << 20)
    outs = [torch.empty((480, 640, 3), device="cuda:0") for _ in cams]
    copies, taken = [], 0
    for k, c in enumerate(cams):
        r.render(c.viewmat, c.K, 640, 480, BG, want=("rgb",), out={"rgb": outs[k]}, block=False)
        while taken < r.frames_completed()[1]:
            copies.append(outs[taken].clone())
            outs[taken].zero_()                        # the consumer owns the buffer from here on
            taken += 1
    assert 1 <= taken < len(cams)
    r.wait()
    while taken < len(cams):
        copies.append(outs[taken].clone())
        taken += 1
    assert r.stats()["regrows"] >= 4                   # every slot overflowed once
    for k, cp in enumerate(copies):
        assert np.array_equal(cp.cpu().numpy(), refs[k % 2]["rgb"]), k
    r.close()


def test_c_abi_error_paths_and_timing_means():
    """Status codes instead of exceptions across the ABI; the wrapper raises RuntimeError (SasError)."""
    import ctypes
    import torch
    from sim_a_splat_amd import _capi
    from sim_a_splat_amd.rasterizer import Rasterizer
    L = _capi.lib()
    ctx = ctypes.c_void_p()
    assert L.sas_create(9999, ctypes.byref(ctx)) == -1 and not ctx.value            # SAS_ERR_INVALID
    r = Rasterizer(0)
    cam = ring_camera(64, 48, 60.0)
    with pytest.raises(_capi.SasError, match="sas_scene_upload"):
        r.render(cam.viewmat, cam.K, 64, 48)                                         # SAS_ERR_NO_SCENE
    sc = make_scene(500, seed=7, log_scale_mean=float(np.log(0.05)))
    with pytest.raises(_capi.SasError, match="sh_degree"):
        r.upload(sc.means, sc.opacities, np.zeros((500, 25, 3), np.float32), quats=sc.quats, scales=sc.scales, sh_degree=4)
    with pytest.raises(_capi.SasError, match="group_id"):
        r.upload(sc.means, sc.opacities, sc.sh, quats=sc.quats, scales=sc.scales, group_id=np.full(500, 9, np.uint8), n_groups=3)
    _upload(r, sc)
    with pytest.raises(_capi.SasError, match="image size"):
        r.render(cam.viewmat, cam.K, 0, 48)
    bad_K = cam.K.copy()
    bad_K[0, 0] = -1.0
    with pytest.raises(_capi.SasError, match="focal"):
        r.render(cam.viewmat, bad_K, 64, 48)
    with pytest.raises(_capi.SasError, match="groups"):
        r.set_group_poses(np.zeros((2, 12), np.float32))                             # scene has no groups
    with pytest.raises(ValueError):
        r.render(cam.viewmat, cam.K, 64, 48, out={"rgb": torch.empty((48, 64, 3), device="cuda:0", dtype=torch.float16)})
    # round-3 entry points: pose sets / link poses on a scene without splat groups, link poses before their constants
    V1, K1, bgf = np.ascontiguousarray(cam.viewmat[None], np.float32), np.ascontiguousarray(cam.K[None], np.float32), np.asarray(BG, np.float32)
    idx, rt = np.zeros(1, np.int32), np.zeros((1, 12), np.float32)
    out8 = torch.empty((1, 48, 64, 3), dtype=torch.uint8, device="cuda:0")
    rc = L.sas_render_batch_posed(r._ctx, 1, V1.ctypes.data, K1.ctypes.data, idx.ctypes.data, 1, rt.ctypes.data, 64, 48, bgf.ctypes.data, 0,
                                  None, None, None, out8.data_ptr(), None)
    assert rc == -1 and b"no splat groups" in L.sas_last_error(r._ctx)
    q1, p1 = np.array([[1.0, 0, 0, 0]]), np.zeros((1, 3))
    assert L.sas_set_link_poses(r._ctx, 1, q1.ctypes.data, p1.ctypes.data, None) == -1 and b"sas_set_link_constants" in L.sas_last_error(r._ctx)
    w4, x3 = np.zeros(4), np.zeros(3)
    assert L.sas_link_attached_frame(r._ctx, q1.ctypes.data, p1.ctypes.data, p1.ctypes.data, w4.ctypes.data, x3.ctypes.data) == -1
    eye = np.eye(3)
    assert L.sas_set_link_constants(r._ctx, 1, 1.0, eye.ctypes.data, x3.ctypes.data, eye.ctypes.data, x3.ctypes.data, None, None) == -1   # more links than groups
    gs = make_scene(400, seed=8, log_scale_mean=float(np.log(0.05)), n_groups=3)
    _upload(r, gs, group_id=gs.group_id, n_groups=3)
    bad = np.array([5], np.int32)
    rc = L.sas_render_batch_posed(r._ctx, 1, V1.ctypes.data, K1.ctypes.data, bad.ctypes.data, 1, np.zeros((1, 3, 12), np.float32).ctypes.data, 64, 48,
                                  bgf.ctypes.data, 0, None, None, None, out8.data_ptr(), None)
    assert rc == -1 and b"pose_set[0]=5" in L.sas_last_error(r._ctx)
    grp = np.array([7], np.int32)
    assert L.sas_set_link_constants(r._ctx, 1, 1.0, eye.ctypes.data, x3.ctypes.data, eye.ctypes.data, x3.ctypes.data, None, grp.ctypes.data) == -1
    assert L.sas_render_cameras_host(r._ctx, 1, None, None, 1.0, 64, 48, bgf.ctypes.data, 0, out8.data_ptr(), None) == -1
    _upload(r, sc)
    # the context is still usable after every failure
    a = r.render(cam.viewmat, cam.K, 64, 48, BG, want=("rgb",))["rgb"].cpu().numpy()
    ref = oracle.render(sc.means, sc.opacities, sc.sh, cam.viewmat, cam.K, 64, 48, quats=sc.quats, scales=sc.scales, background=BG)
    assert np.array_equal(a, ref["rgb"])
    # timing: per-stage events of an isolated frame, and the pipelined tile-kernel timing means
    r.render(cam.viewmat, cam.K, 64, 48, BG, want=("rgb",), timing=True)
    st = r.stage_times()
    assert st["total"] > 0 and st["project"] > 0 and st["blend"] > 0
    r.stage_time_means(reset=True)
    for _ in range(5):
        r.render(cam.viewmat, cam.K, 64, 48, BG, want=("rgb",), block=False, time_tiles=True)
    r.wait()
    means, frames = r.stage_time_means(reset=True)
    assert frames == 5 and 0 < means["blend"] < 5.0
    r.close()


def _rand_rot(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


@pytest.mark.parametrize("group", ["1", "2"])
def test_frames_delivered_to_host_memory(monkeypatch, group):
    """sas_render_batch_host: uint8 frames copied to host memory on the frames' own streams (what get_render hands
    out).  Equal, byte for byte, to the device frames of sas_render_batch: one view, a launch group plus a
    remainder (and, with SAS_GROUP=1, views one at a time), pinned and pageable destinations, a frame that
    overflows the intersection buffer on the way (rendered again, copied again); SAS_ASYNC is refused."""
    import torch
    from sim_a_splat_amd import _capi
    from sim_a_splat_amd.rasterizer import Rasterizer
    monkeypatch.setenv("SAS_GROUP", group)
    r = Rasterizer("cuda:0")
    try:
        sc = make_scene(20000, seed=555, log_scale_mean=float(np.log(0.025)))
        _upload(r, sc)
        cams = [ring_camera(200, 136, 170.0, yaw_deg=y, elev=e) for y, e in ((0.0, 0.0), (70.0, 0.3), (140.0, -0.2))]
        Vs, Ks = np.stack([c.viewmat for c in cams]), np.stack([c.K for c in cams])
        dev = r.render_batch(Vs, Ks, 200, 136, BG, want=("rgb8",))["rgb8"].cpu()
        host = r.render_batch_host(Vs, Ks, 200, 136, BG)
        assert host.is_pinned() and host.device.type == "cpu" and torch.equal(host, dev)
        pageable = torch.zeros((3, 136, 200, 3), dtype=torch.uint8)
        assert r.render_batch_host(Vs, Ks, 200, 136, BG, out=pageable) is pageable and torch.equal(pageable, dev)
        one = r.render_batch_host(Vs[1:2], Ks[1:2], 200, 136, BG)
        assert torch.equal(one[0], dev[1])
        L = _capi.lib()
        bg = np.asarray(BG, dtype=np.float32)
        rc = L.sas_render_batch_host(r._ctx, 3, np.ascontiguousarray(Vs, np.float32).ctypes.data, np.ascontiguousarray(Ks, np.float32).ctypes.data,
                                     200, 136, bg.ctypes.data, _capi.SAS_ASYNC, host.data_ptr(), None)
        assert rc != 0
    finally:
        r.close()
    # overflow on the way: fresh buffers (2^20 keys), both views need more
    r = Rasterizer("cuda:0")
    try:
        big = make_scene(30000, seed=444, log_scale_mean=float(np.log(0.12)))
        _upload(r, big)
        bc = [ring_camera(640, 480, 500.0, yaw_deg=90.0 * k) for k in range(2)]
        out = r.render_batch_host(np.stack([c_.viewmat for c_ in bc]), np.stack([c_.K for c_ in bc]), 640, 480, BG)
        assert r.stats()["regrows"] >= 1
        for v in range(2):
            ref = oracle.render_scene(big, bc[v], background=BG, want_rgb8=True)
            assert np.array_equal(out[v].numpy(), ref["rgb8"]), v
    finally:
        r.close()


@pytest.mark.parametrize("quad", ["0", "1"])
def test_both_tile_kernel_layouts(monkeypatch, quad):
    """A frame has one of two layouts: 16-pixel tiles with one workgroup per tile (16-lane groups walk their 4x4
    block's queue, two entries per trip) or, for frames of a few hundred tiles, SAS_QUAD: the view is BINNED in 8-pixel
    tiles, one workgroup per 8x8 quadrant with its own list, one wave per 4x4 block, the four lanes of a DPP quad
    evaluating four consecutive entries of one pixel (the sigma polynomials still refer to the 16-pixel tile's origin).
    Left to itself the library picks by tile count, so here each layout is forced onto every kind of frame: a twin
    fixture with thousands of entries per tile, an image whose size is a multiple of neither 8 nor 16, coplanar splats
    (a crowded depth bucket: the complete ordering), long translucent lists (many rounds), the per-tile depth maxima
    behind SAS_DEPTH_FILL_MAX, a launch group, and a frame far past the automatic threshold.  All bit-identical to
    the oracle; n_isect stays the count of 16-pixel intersections in both, n_keys is what the frame binned."""
    import torch
    from sim_a_splat_amd.rasterizer import Rasterizer
    monkeypatch.setenv("SAS_QUAD", quad)
    r = Rasterizer("cuda:0")
    try:
        g = load_twin_fixture("dense")
        means, op, colors, kw = twin_scene_kwargs(g)
        W, H = [int(v) for v in g["wh"]]
        r.upload(means, op, colors, quats=kw.get("quats"), scales=kw.get("scales"), sh_degree=kw["sh_degree"])
        bg = tuple(float(v) for v in g["background"])
        out = r.render(g["viewmat"], g["K"], W, H, bg, want=("rgb", "alpha", "depth"))
        ref = oracle.render(means, op, colors, g["viewmat"], g["K"], W, H, background=bg, **kw)
        for k in ("rgb", "alpha", "depth"):
            assert np.array_equal(out[k].cpu().numpy(), ref[k]), k
        # ragged image, depth fill (per-tile maxima)
        sc = make_scene(3000, seed=31, log_scale_mean=float(np.log(0.05)))
        _upload(r, sc)
        _compare(r, sc, ring_camera(75, 53, 70.0), depth_fill=True)
        # crowded depth bucket -> complete ordering
        rng = np.random.default_rng(21)
        sc = make_scene(20000, seed=97, log_scale_mean=float(np.log(0.02)))
        sc.means[:, 2] = 0.0
        sc.means[:, :2] = rng.uniform(-0.5, 0.5, size=(sc.n, 2)).astype(np.float32)
        sc.opacities[:] = np.clip(sc.opacities, 0.01, 0.05)
        _upload(r, sc)
        _compare(r, sc, ring_camera(96, 64, 90.0), depth_fill=True)
        assert r.stats()["fallback_tiles"] > 0
        # many rounds
        rng = np.random.default_rng(22)
        sc = make_scene(52000, seed=98, log_scale_mean=float(np.log(0.01)))
        sc.means[:] = rng.uniform(-0.4, 0.4, size=sc.means.shape).astype(np.float32)
        sc.opacities[:] = np.clip(sc.opacities, 0.004, 0.01)
        _upload(r, sc)
        _compare(r, sc, ring_camera(80, 64, 110.0))
        assert r.stats()["max_tile_len"] > 4096
        # a launch group of three views and a 1200-tile frame
        sc = make_scene(20000, seed=555, log_scale_mean=float(np.log(0.025)))
        _upload(r, sc)
        cams = [ring_camera(200, 136, 170.0, yaw_deg=y) for y in (0.0, 100.0, 220.0)]
        batch = r.render_batch(np.stack([c.viewmat for c in cams]), np.stack([c.K for c in cams]), 200, 136, BG,
                               want=("rgb", "depth"), depth_fill_max=True)
        for i, cam in enumerate(cams):
            ref = oracle.render_scene(sc, cam, background=BG, depth_mode=1)
            assert np.array_equal(batch["rgb"][i].cpu().numpy(), ref["rgb"]), i
            assert np.array_equal(batch["depth"][i].cpu().numpy(), ref["depth"]), i
        _compare(r, sc, ring_camera(640, 480, 500.0))
        st = r.stats()
        assert st["quad_layout"] == int(quad)
        # (ordinary layout: the lists hold T3's rectangle intersections minus the tiles a Gaussian cannot reach)
        assert st["n_keys"] > st["n_isect"] if quad == "1" else 0 < st["n_keys"] <= st["n_isect"]
    finally:
        r.close()


def test_tile_kernel_layout_follows_the_view_size(rasterizer):
    """Left to itself the library renders views of at most 640 tiles (SAS_QUAD_TILES) in the quad layout and larger
    ones in the ordinary one -- up to 960 tiles when the call is a blocking one for a single view with nothing else in
    flight (the caller waits for that frame's heaviest tile; frames that share the chip lose with four times the
    workgroups: tools/quad_threshold.py); sas_frame_stats reports which."""
    import os
    if os.environ.get("SAS_QUAD") is not None or os.environ.get("SAS_QUAD_TILES") is not None:
        pytest.skip("the layout is forced through the environment in this run")
    sc = make_scene(3000, seed=77, log_scale_mean=float(np.log(0.04)))
    _upload(rasterizer, sc)
    _compare(rasterizer, sc, ring_camera(320, 240, 260.0))      # 300 tiles
    assert rasterizer.stats()["quad_layout"] == 1
    _compare(rasterizer, sc, ring_camera(640, 480, 520.0))      # 1200 tiles
    assert rasterizer.stats()["quad_layout"] == 0
    cam = ring_camera(480, 368, 400.0)                          # 690 tiles
    _compare(rasterizer, sc, cam)                               # ... blocking and alone
    assert rasterizer.stats()["quad_layout"] == 1
    ref = oracle.render_scene(sc, cam, background=BG)
    outs = [rasterizer.render(cam.viewmat, cam.K, 480, 368, BG, want=("rgb",), block=False)["rgb"] for _ in range(3)]   # ... pipelined
    rasterizer.wait()
    assert rasterizer.stats()["quad_layout"] == 0
    for o in outs:
        assert np.array_equal(o.cpu().numpy(), ref["rgb"])


@pytest.mark.parametrize("seed", range(12))
def test_randomised_edge_cases(rasterizer, seed):
    """Random cameras (also INSIDE the scene: near-plane culls, clamped Jacobian limits), odd image
    sizes, off-centre principal points, non-square pixels, extreme anisotropy, screen-filling
    splats, SH degree 0..3 or covariance + RGB input with group poses."""
    from sim_a_splat_amd.synthetic import Camera
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 6000))
    sc = make_scene(n, seed=2000 + seed, log_scale_mean=float(np.log(rng.choice([0.005, 0.03, 0.15]))))
    sc.scales[:] = (sc.scales * np.exp(rng.normal(0, 1.0, size=sc.scales.shape))).astype(np.float32)   # anisotropy
    if n > 10:
        sc.scales[: max(1, n // 200)] *= 20.0                                                           # huge splats
    W, H = [(1, 1), (15, 17), (16, 16), (33, 7), (100, 100), (257, 129), (640, 360)][seed % 7]
    fx = float(rng.uniform(0.3, 2.0) * W + 5)
    fy = float(fx * rng.uniform(0.8, 1.25))
    K = np.array([[fx, 0, W * rng.uniform(0.2, 0.8)], [0, fy, H * rng.uniform(0.2, 0.8)], [0, 0, 1]], np.float32)
    R = _rand_rot(rng)
    eye = rng.uniform(-1, 1, size=3) * (0.5 if seed % 2 else 3.0)        # odd seeds: camera inside the cloud
    V = np.eye(4)
    V[:3, :3] = R
    V[:3, 3] = -R @ eye
    cam = Camera(V.astype(np.float32), K, W, H)
    deg = seed % 5 - 1
    gid = Rt = None
    if deg < 0:                                                          # Door-B form
        from oracle import ref_math
        q = sc.quats / np.linalg.norm(sc.quats, axis=1, keepdims=True)
        Rm = np.stack([ref_math.quat_wxyz_to_R(x) for x in q]).astype(np.float32)
        M = Rm * sc.scales[:, None, :]
        cov = (M @ M.transpose(0, 2, 1)).astype(np.float32)
        cov6 = np.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], 1)
        rgb = rng.uniform(0, 1, size=(n, 3)).astype(np.float32)
        gid = rng.integers(0, 5, size=n).astype(np.uint8)
        Rt = random_group_poses(5, seed=seed)
        rasterizer.upload(sc.means, sc.opacities, rgb, covariances=cov, sh_degree=-1, group_id=gid, n_groups=5)
        rasterizer.set_group_poses(Rt)
        ref = oracle.render(sc.means, sc.opacities, rgb, cam.viewmat, K, W, H, cov6=cov6, sh_degree=-1, group_id=gid,
                            group_Rt=Rt, background=BG, want_rgb8=True, depth_mode=1)
    else:
        sh = np.ascontiguousarray(sc.sh[:, : (deg + 1) ** 2])
        rasterizer.upload(sc.means, sc.opacities, sh, quats=sc.quats, scales=sc.scales, sh_degree=deg)
        ref = oracle.render(sc.means, sc.opacities, sh, cam.viewmat, K, W, H, quats=sc.quats, scales=sc.scales,
                            sh_degree=deg, background=BG, want_rgb8=True, depth_mode=1)
    out = rasterizer.render(cam.viewmat, K, W, H, BG, want=("rgb", "alpha", "depth", "rgb8"), depth_fill_max=True)
    st = rasterizer.stats()
    assert st["n_visible"] == ref["n_visible"] and st["n_isect"] == ref["n_isect"]
    for k in ("rgb", "alpha", "depth", "rgb8"):
        got = out[k].cpu().numpy()
        assert np.abs(got.astype(np.float64) - ref[k]).max() <= (1 if k == "rgb8" else 1e-4), k
        assert np.array_equal(got, ref[k]), k


def test_quad_layout_crowded_bucket_with_a_wide_depth_range(monkeypatch):
    """The quad layout's complete-ordering path on tiles whose keys span a WIDE depth range (so that their workgroups
    really do bucket and scan minima / maxima) while one depth bucket is crowded (3 000 coplanar splats).  Round 2's
    quad layout shared a 16-pixel tile's list between four workgroups and raced here; since round 3 every 8x8 quadrant
    is a tile of its own with its own key and id segments, ordered by its one workgroup.  Several frames in flight,
    twice."""
    import torch
    from sim_a_splat_amd.rasterizer import Rasterizer
    from sim_a_splat_amd.synthetic import SyntheticScene
    monkeypatch.setenv("SAS_QUAD", "1")
    rng = np.random.default_rng(77)
    n_plane, n_rest = 3000, 3000
    sc = make_scene(n_plane + n_rest, seed=612, log_scale_mean=float(np.log(0.03)))
    sc.means[:n_plane, 2] = 0.25                                              # one crowded depth
    sc.means[:n_plane, :2] = rng.uniform(-0.35, 0.35, size=(n_plane, 2)).astype(np.float32)
    sc.means[n_plane:] = np.stack([rng.uniform(-0.5, 0.5, n_rest), rng.uniform(-0.5, 0.5, n_rest), rng.uniform(-2.5, 2.0, n_rest)], 1).astype(np.float32)
    sc.opacities[:] = np.clip(sc.opacities, 0.004, 0.03)                      # nothing saturates: every list is walked to its end
    r = Rasterizer("cuda:0")
    try:
        _upload(r, sc)
        cam = ring_camera(64, 48, 60.0)
        _compare(r, sc, cam, depth_fill=True)
        st = r.stats()
        assert st["fallback_tiles"] > 0 and st["quad_layout"] == 1 and st["max_tile_len"] > 1024
        ref = oracle.render_scene(sc, cam, background=BG)
        outs = [torch.empty((48, 64, 3), device="cuda:0") for _ in range(4)]
        for rep in range(2):
            for i in range(4):
                r.render(cam.viewmat, cam.K, 64, 48, BG, want=("rgb",), out={"rgb": outs[i]}, block=False)
            r.wait()
            for i in range(4):
                assert np.array_equal(outs[i].cpu().numpy(), ref["rgb"]), (rep, i)
    finally:
        r.close()


def test_radius_beyond_16_bits_reads_back_whole(rasterizer):
    """A camera inside a large, close Gaussian: projected radii beyond 65 535 pixels (z = 0.02, fx = 2000).  The
    parity hook reports them whole (they used to be packed in 16 bits each); the image is unaffected either way."""
    from sim_a_splat_amd.synthetic import SyntheticScene, Camera, intrinsics
    means = np.array([[0.0, 0.0, 0.02], [0.3, 0.1, 1.0]], np.float32)
    sc = SyntheticScene(means=means, quats=np.array([[1, 0, 0, 0], [1, 0, 0, 0]], np.float32),
                        scales=np.array([[0.5, 0.5, 0.001], [0.05, 0.05, 0.05]], np.float32), opacities=np.array([0.9, 0.8], np.float32),
                        sh=np.zeros((2, 16, 3), np.float32), sh_degree=3)
    sc.sh[:, 0] = 1.0
    cam = Camera(np.eye(4, dtype=np.float32), intrinsics(2000.0, 2000.0, 32.0, 24.0), 64, 48)
    _upload(rasterizer, sc)
    _, ref = _compare(rasterizer, sc, cam)
    proj = rasterizer.read_projection()
    assert ref["radii"].max() > 65535
    assert np.array_equal(proj["radii"], ref["radii"])


def test_per_view_pose_sets_of_vectorised_envs(rasterizer):
    """sas_render_batch_posed: four envs, each with its OWN link poses, two cameras per env, one call; every frame
    equals the oracle's frame of that env's poses (splat_env_wrapper.py:121-159 poses the scene per env).  Also through
    the host-delivery form, and with sas_set_group_poses in between without disturbing frames in flight."""
    G, E = 6, 4
    sc = make_scene(30000, seed=808, log_scale_mean=float(np.log(0.03)), n_groups=G)
    _upload(rasterizer, sc, group_id=sc.group_id, n_groups=G)
    sets = np.stack([random_group_poses(G, seed=300 + e, max_angle=0.6, max_shift=0.25) for e in range(E)])
    cams = [ring_camera(160, 120, 130.0, yaw_deg=20.0), ring_camera(160, 120, 130.0, yaw_deg=140.0, elev=0.5)]
    Vs = np.stack([cams[v % 2].viewmat for v in range(2 * E)])
    Ks = np.stack([cams[v % 2].K for v in range(2 * E)])
    idx = [v // 2 for v in range(2 * E)]
    refs = [oracle.render(sc.means, sc.opacities, sc.sh, cams[v % 2].viewmat, cams[v % 2].K, 160, 120, quats=sc.quats, scales=sc.scales,
                          sh_degree=3, group_id=sc.group_id, group_Rt=sets[v // 2], background=BG, want_rgb8=True) for v in range(2 * E)]
    assert not np.array_equal(refs[0]["rgb"], refs[2]["rgb"])                  # the envs really differ
    rasterizer.set_group_poses(random_group_poses(G, seed=999))               # the context's own poses: not used by posed views
    out = rasterizer.render_batch(Vs, Ks, 160, 120, BG, want=("rgb", "rgb8"), pose_sets=sets, pose_set=idx)
    for v in range(2 * E):
        assert np.array_equal(out["rgb"][v].cpu().numpy(), refs[v]["rgb"]), v
        assert np.array_equal(out["rgb8"][v].cpu().numpy(), refs[v]["rgb8"]), v
    host = rasterizer.render_batch_host(Vs, Ks, 160, 120, BG, pose_sets=sets, pose_set=idx)
    for v in range(2 * E):
        assert np.array_equal(host[v].numpy(), refs[v]["rgb8"]), v
    # a pose update while frames are in flight: the frames keep the poses they were submitted with
    outs = []
    for e in range(E):
        rasterizer.set_group_poses(sets[e])
        outs.append(rasterizer.render(cams[0].viewmat, cams[0].K, 160, 120, BG, want=("rgb",), block=False))
    rasterizer.set_group_poses(random_group_poses(G, seed=5))
    rasterizer.wait()
    for e in range(E):
        assert np.array_equal(outs[e]["rgb"].cpu().numpy(), refs[2 * e]["rgb"]), e
    with pytest.raises(Exception):
        rasterizer.render_batch(Vs, Ks, 160, 120, BG, pose_sets=sets, pose_set=[9] * (2 * E))



@pytest.mark.parametrize("quad", ["0", "1"])
def test_tile_kernel_delivers_complete_tile_frames_to_the_host(monkeypatch, quad):
    """Frames wanted in pinned host memory whose tiles are all complete (W, H multiples of 16) are stored there by the
    tile kernel itself -- a tile's rows packed in LDS, 16 bytes per lane (8 per lane for a quadrant in the quad
    layout) -- instead of through a device frame and a copy kernel.  Both layouts forced, against the oracle: the Gym
    camera size (240x320), 640x480, a launch group with per-view pose sets, the complete-ordering path, an overflowing
    frame (rendered again), and the
    sizes that must NOT take this path (ragged image, pageable destination) beside it."""
    import torch
    from sim_a_splat_amd.rasterizer import Rasterizer
    monkeypatch.setenv("SAS_QUAD", quad)
    r = Rasterizer("cuda:0")
    try:
        G = 5
        sc = make_scene(25000, seed=4242, log_scale_mean=float(np.log(0.03)), n_groups=G)
        _upload(r, sc, group_id=sc.group_id, n_groups=G)
        sets = np.stack([random_group_poses(G, seed=70 + e) for e in range(2)])
        for (W, H, f) in ((320, 240, 262.0), (640, 480, 500.0), (200, 136, 170.0)):
            cams = [ring_camera(W, H, f, yaw_deg=y, elev=e) for y, e in ((0.0, 0.0), (70.0, 0.3), (140.0, -0.2))]
            Vs, Ks = np.stack([c.viewmat for c in cams]), np.stack([c.K for c in cams])
            idx = [0, 1, 1]
            host = r.render_batch_host(Vs, Ks, W, H, BG, pose_sets=sets, pose_set=idx)
            pageable = torch.zeros((3, H, W, 3), dtype=torch.uint8)
            r.render_batch_host(Vs, Ks, W, H, BG, out=pageable, pose_sets=sets, pose_set=idx)
            for v, c in enumerate(cams):
                ref = oracle.render(sc.means, sc.opacities, sc.sh, c.viewmat, c.K, W, H, quats=sc.quats, scales=sc.scales, sh_degree=3,
                                    group_id=sc.group_id, group_Rt=sets[idx[v]], background=BG, want_rgb8=True)
                assert np.array_equal(host[v].numpy(), ref["rgb8"]), (W, H, v)
                assert np.array_equal(pageable[v].numpy(), ref["rgb8"]), (W, H, v, "pageable")
            one = r.render_batch_host(Vs[2:3], Ks[2:3], W, H, BG, pose_sets=sets, pose_set=[1])      # a single frame, not a group
            assert torch.equal(one[0], host[2])
        # a crowded depth bucket on top of a wide depth range: the complete ordering
        rng = np.random.default_rng(5)
        flat = make_scene(6000, seed=613, log_scale_mean=float(np.log(0.03)))
        flat.means[:3000, 2] = 0.25
        flat.means[:3000, :2] = rng.uniform(-0.35, 0.35, size=(3000, 2)).astype(np.float32)
        flat.means[3000:] = np.stack([rng.uniform(-0.5, 0.5, 3000), rng.uniform(-0.5, 0.5, 3000), rng.uniform(-2.5, 2.0, 3000)], 1).astype(np.float32)
        flat.opacities[:] = np.clip(flat.opacities, 0.004, 0.03)
        _upload(r, flat)
        cam = ring_camera(64, 48, 60.0)
        out = r.render_batch_host(cam.viewmat[None], cam.K[None], 64, 48, BG)
        assert r.stats()["fallback_tiles"] > 0
        assert np.array_equal(out[0].numpy(), oracle.render_scene(flat, cam, background=BG, want_rgb8=True)["rgb8"])
    finally:
        r.close()
    # an overflowing frame is rendered again and delivered again
    r = Rasterizer("cuda:0")
    try:
        big = make_scene(30000, seed=444, log_scale_mean=float(np.log(0.12)))
        _upload(r, big)
        bc = ring_camera(640, 480, 500.0, yaw_deg=90.0)
        out = r.render_batch_host(bc.viewmat[None], bc.K[None], 640, 480, BG)
        assert r.stats()["regrows"] >= 1
        assert np.array_equal(out[0].numpy(), oracle.render_scene(big, bc, background=BG, want_rgb8=True)["rgb8"])
    finally:
        r.close()


def test_single_pass_and_two_pass_binning_render_the_same_frames(monkeypatch):
    """The product path bins in ONE pass (fixed-stride tile segments, the projection emits the keys, the tile order comes
    from its tail); SAS_DIRECT=0 keeps the two-pass path of rounds 1-3 (count, scan, scatter), which SAS_FULL_SORT frames and
    frames whose segments would not fit the memory budget still take.  Same frames, bit for bit, equal to the oracle: a
    1080p view of 200 k Gaussians, a view pair, a 5-view launch group with link poses, and a frame whose segments exceed a
    1 MB budget (falls back to two passes by itself)."""
    import torch
    from sim_a_splat_amd.rasterizer import Rasterizer
    sc = make_scene(200_000, seed=77, log_scale_mean=float(np.log(0.012)))
    cams = [ring_camera(1920, 1080, 1000.0, yaw_deg=30.0 * k) for k in range(2)]
    small = make_scene(20_000, seed=78, log_scale_mean=float(np.log(0.03)), n_groups=3)
    scams = [ring_camera(320, 240, 260.0, yaw_deg=72.0 * k) for k in range(5)]
    sV, sK = np.stack([c.viewmat for c in scams]), np.stack([c.K for c in scams])
    ref = oracle.render_scene(sc, cams[0], background=BG)
    frames = {}
    for mode, env in (("single-pass", {"SAS_DIRECT": "1"}), ("two-pass", {"SAS_DIRECT": "0"}), ("budget", {"SAS_DIRECT": "1", "SAS_DIRECT_BUDGET_MB": "1"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        r = Rasterizer("cuda:0")
        _upload(r, sc)
        one = r.render(cams[0].viewmat, cams[0].K, 1920, 1080, BG, want=("rgb", "alpha", "depth"), depth_fill_max=True)
        st = r.stats()
        assert st["n_isect"] == ref["n_isect"] and st["capacity"] >= st["n_isect"]
        pair = r.render_batch(np.stack([c.viewmat for c in cams]), np.stack([c.K for c in cams]), 1920, 1080, BG, want=("rgb",))["rgb"]
        out = [one["rgb"].cpu().numpy(), one["alpha"].cpu().numpy(), one["depth"].cpu().numpy(), pair.cpu().numpy()]
        _upload(r, small, group_id=small.group_id, n_groups=3)
        r.set_group_poses(random_group_poses(3, seed=9))
        out.append(r.render_batch(sV, sK, 320, 240, BG, want=("rgb8",))["rgb8"].cpu().numpy())
        frames[mode] = out
        r.close()
        monkeypatch.delenv("SAS_DIRECT_BUDGET_MB", raising=False)
    assert np.array_equal(frames["single-pass"][0], ref["rgb"]) and np.array_equal(frames["single-pass"][3][0], ref["rgb"])
    for mode in ("two-pass", "budget"):
        for a, b in zip(frames["single-pass"], frames[mode]):
            assert np.array_equal(a, b), mode


@pytest.mark.parametrize("quad", ["0", "1"])
def test_exact_tile_culling_changes_no_pixel(monkeypatch, quad):
    """Single-pass frames leave a Gaussian out of the lists of those tiles of its rectangle in which no pixel centre can
    get alpha >= 1/255 from it (the minimum of sigma over the tile's pixel centres against ln(255 o), with margins:
    sas_kernels.hip, tile_reached).  Such an entry composites nothing wherever it stands in its list, so the frame is the
    same bit for bit -- here against the oracle, whose lists are T3's whole rectangles, and against SAS_CULL=0: config-2-like
    content, needles (scale ratios up to 1 : 3000, every orientation: the conic's cross term cancels its square terms to
    a part in 10^6 and more), splats larger than the image, opacities from just above 1/255 to 0.999, a camera inside the
    cloud; both tile layouts.  n_isect stays T3's count; n_keys is what was binned."""
    import os, torch
    from sim_a_splat_amd.rasterizer import Rasterizer
    if os.environ.get("SAS_DIRECT") == "0":
        pytest.skip("two-pass binning is forced through the environment in this run: its lists are T3's")
    monkeypatch.setenv("SAS_QUAD", quad)
    rng = np.random.default_rng(2024)
    scenes = []
    sc = make_scene(60_000, seed=41, log_scale_mean=float(np.log(0.012)))
    scenes.append((sc, ring_camera(640, 480, 525.0)))
    needles = make_scene(30_000, seed=42, log_scale_mean=float(np.log(0.01)))
    needles.scales[:] = np.exp(rng.uniform(np.log(1e-4), np.log(0.3), size=needles.scales.shape)).astype(np.float32)
    needles.opacities[:] = rng.choice(np.array([0.004, 0.0045, 0.02, 0.3, 0.9, 0.999], np.float32), size=needles.opacities.shape)
    scenes.append((needles, ring_camera(333, 251, 300.0, yaw_deg=40.0)))
    big = make_scene(4_000, seed=43, log_scale_mean=float(np.log(0.4)))
    big.opacities[:] = np.clip(big.opacities, 0.004, 0.2)
    scenes.append((big, ring_camera(320, 240, 200.0, yaw_deg=10.0)))
    inside = make_scene(50_000, seed=44, log_scale_mean=float(np.log(0.02)))
    cam_in = ring_camera(400, 304, 250.0, radius=0.2)
    scenes.append((inside, cam_in))
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent / "tools"))
    from tile_cull_model import kept_pairs
    import oracle
    keys, model = {}, {}
    for cull in ("1", "0"):
        monkeypatch.setenv("SAS_CULL", cull)
        r = Rasterizer("cuda:0")
        try:
            for i, (sc, cam) in enumerate(scenes):
                _upload(r, sc)
                _, ref = _compare(r, sc, cam, depth_fill=True)
                st = r.stats()
                keys[(cull, i)] = (st["n_keys"], st["n_isect"])
                if cull == "1" and quad == "0":
                    # the EXACT number of keys the kernel may bin: the NumPy restatement of its test (tests/tools/tile_cull_model.py,
                    # operation for operation) on the oracle's projection of the same view
                    o = oracle.render(sc.means, sc.opacities, sc.sh, cam.viewmat, cam.K, cam.width, cam.height, quats=sc.quats,
                                      scales=sc.scales, sh_degree=3, dump=True)
                    n_rect, keep, *_ = kept_pairs(o, sc.opacities, cam.width, cam.height, oracle.logf)
                    model[i] = (n_rect, int(keep.sum()))
        finally:
            r.close()
    for i in range(len(scenes)):
        on, off = keys[("1", i)], keys[("0", i)]
        assert on[1] == off[1]                      # T3's count either way
        assert on[0] < off[0], (i, on, off)         # and shorter lists
        if quad == "0":
            assert off[0] == off[1]
            # an over-cull that happened to leave these images unchanged, or a drift between the kernel and its model, shows here
            assert model[i] == (on[1], on[0]), (i, model[i], on)


def test_segment_sizing_follows_the_frame(monkeypatch):
    """Single-pass binning gives every tile a fixed-stride segment; a frame whose segments would exceed the memory budget
    (SAS_DIRECT_BUDGET_MB) takes the two-pass path instead.  That verdict belongs to the frame size and the scene it was
    reached for: the same context goes back to single-pass binning (seen here by the culled lists: n_keys < n_isect) when
    a smaller scene is uploaded, and both frames equal the oracle."""
    import os
    from sim_a_splat_amd.rasterizer import Rasterizer
    if os.environ.get("SAS_DIRECT") == "0" or os.environ.get("SAS_CULL") == "0":
        pytest.skip("the binning is forced through the environment in this run")
    monkeypatch.setenv("SAS_DIRECT_BUDGET_MB", "128")
    monkeypatch.setenv("SAS_QUAD", "0")
    r = Rasterizer("cuda:0")
    try:
        big = make_scene(200_000, seed=77, log_scale_mean=float(np.log(0.012)))
        _upload(r, big)
        _compare(r, big, ring_camera(1920, 1080, 1000.0))     # 8 160 tiles x 2 048 keys x 12 B = 200 MB: two passes
        st = r.stats()
        assert st["n_keys"] == st["n_isect"]
        small = make_scene(20_000, seed=78, log_scale_mean=float(np.log(0.03)))
        _upload(r, small)
        _compare(r, small, ring_camera(640, 480, 500.0))      # 1 200 tiles x 2 048 keys x 12 B = 29 MB: one pass again
        st = r.stats()
        assert 0 < st["n_keys"] < st["n_isect"]
    finally:
        r.close()
