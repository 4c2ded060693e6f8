"""Generate tests/golden/*.npz.  Run from the repo root in the build container:

    python oracle/make_golden.py

(1) compute_cov.npz  -- inputs and outputs of the REFERENCE's own compute_cov, obtained by
    importing /root/reference/sim_a_splat/ellipsoids/covariance_utils.py by file path (a
    torch-only module).  Only data is stored, no reference source.
(2) render_twin_*.npz -- seeded scenes rendered by the float64 NumPy twin (textbook formulas, libm exp);
    they pin the C oracle (tests/test_oracle.py) AND the HIP kernels directly
    (tests/test_gpu_parity.py::test_golden_twin_fixtures_through_the_c_abi): small cases, a dense
    early-terminating slab (config 3's regime), a Door-B cov6 + RGB + group-pose case, a camera inside
    the cloud.
(3) scene_assets_xarm6_1.npz -- the small data files the reference ships for its real scene
    (robots-scene-v2 / xarm6-1): ICP similarity, mask-time joint configuration, dataparser
    transform and the seven per-link masks, bit-packed.  Data only; the masks' pickled .npy is
    read with the whitelisting unpickler of sim_a_splat_amd.io, never np.load(allow_pickle=True).
(4) scene_assets_divar113vhw.npz -- the same for the ~300k scene (divar113vhw: six link masks over
    292,247 Gaussians, ICP similarity; that scene ships no joint_config.npy).
(5) ns_run_<scene>.npz -- the DATA FILES of both trained runs as the reference ships them, byte for byte:
    the nerfstudio config.yml (a YAML dump), dataparser_transforms.json and the dataset's transforms.json
    (camera intrinsics + 293 / 307 camera poses).  Tests materialise them into the reference's directory
    layout, add a fabricated checkpoint (both real ones are Git-LFS pointers) and construct
    GaussianSplat(config_path, ...) on it; dataparser_transforms.json pins the dataparser restatement.
The reference tree does not exist on the GPU box; tests read only the committed fixtures.
"""
from __future__ import annotations

import importlib.util
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
GOLD = ROOT / "tests" / "golden"
REF = Path("/root/reference/sim_a_splat/ellipsoids/covariance_utils.py")


def gen_compute_cov():
    import torch
    spec = importlib.util.spec_from_file_location("ref_covariance_utils", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    torch.manual_seed(0)
    quats = torch.randn(256, 4)
    scales = torch.rand(256, 3)
    # edge cases: identity (Taylor branch), tiny rotation, near-180 deg (cos_theta < eps branch), negative w
    edge_q = torch.tensor([[1.0, 0, 0, 0], [1.0, 1e-4, 0, 0], [1e-7, 1.0, 0, 0], [0.0, 0, 1.0, 0],
                           [-1.0, 0.2, 0.1, 0.3], [2.0, 0, 0, 2.0], [-0.5, -0.5, -0.5, -0.5], [1e-3, 0.7, 0.7, 0.1]])
    edge_s = torch.tensor([[0.1, 0.2, 0.3]]).repeat(edge_q.shape[0], 1)
    quats = torch.cat([quats, edge_q])
    scales = torch.cat([scales, edge_s])
    covs = mod.compute_cov(quats, scales)
    covs_inv = mod.compute_cov(quats, 1.0 / scales)   # splat_utils.py:38
    rots = mod.quaternion_to_rotation_matrix(quats)
    np.savez_compressed(GOLD / "compute_cov.npz", quats=quats.numpy(), scales=scales.numpy(),
                        covs=covs.numpy(), covs_inv=covs_inv.numpy(), rots=rots.numpy())
    print("compute_cov.npz", covs.shape)


def _dense_scene(n, seed):
    """Config-3-like regime on a small image: a thick slab of small, fairly opaque Gaussians in front of
    the camera, so that every tile list runs to thousands of entries, is composited through several
    512-entry chunks and almost every pixel ends on the T <= 1e-4 stop after hundreds of splats."""
    from sim_a_splat_amd.synthetic import SyntheticScene
    rng = np.random.default_rng(seed)
    means = np.stack([rng.uniform(-1.3, 1.3, n), rng.uniform(-0.9, 0.9, n), rng.uniform(-0.8, 0.8, n)], 1)
    log_s = np.clip(rng.normal(np.log(0.06), 0.5, (n, 3)), np.log(2e-3), np.log(0.2))
    logit = rng.normal(-1.2, 1.2, n)
    dc = rng.normal(0.0, 1.0, (n, 1, 3))
    rest = rng.normal(0.0, 0.15, (n, 3, 3))
    return SyntheticScene(means=means.astype(np.float32), quats=rng.normal(size=(n, 4)).astype(np.float32),
                          scales=np.exp(log_s).astype(np.float32),
                          opacities=(1.0 / (1.0 + np.exp(-logit))).astype(np.float32),
                          sh=np.concatenate([dc, rest], 1).astype(np.float32), sh_degree=1)


def gen_twin_renders():
    from oracle import np_twin, ref_math
    from sim_a_splat_amd.synthetic import (make_scene, ring_camera, random_group_poses, look_at_viewmat, intrinsics, Camera,
                                           NERFSTUDIO_EVAL_BACKGROUND)
    cases = {
        "one": dict(n=1, seed=11, w=32, h=32, f=40.0, ls=np.log(0.15)),
        "two": dict(n=2, seed=12, w=48, h=32, f=40.0, ls=np.log(0.2)),
        "n64": dict(n=64, seed=13, w=64, h=64, f=60.0, ls=np.log(0.08)),
        "n2k": dict(n=2000, seed=14, w=128, h=96, f=120.0, ls=np.log(0.03)),
        "n2k_groups": dict(n=2000, seed=15, w=100, h=70, f=90.0, ls=np.log(0.03), groups=5),
        # the regime config 3 lives in: lists of thousands, several lazy chunks per tile, early termination
        "dense": dict(dense=20000, seed=16, w=96, h=64, f=75.0),
        # Door B: 3x3 covariances (reference compute_cov semantics) + final RGB + group poses
        "doorb": dict(n=3000, seed=17, w=112, h=80, f=100.0, ls=np.log(0.04), groups=4, door_b=True),
        # camera inside the cloud: near-plane cull, clamped Jacobian (0.3 tan_fov limits), huge footprints
        "inside": dict(n=1500, seed=18, w=96, h=80, f=70.0, ls=np.log(0.05), inside=True),
    }
    for name, c in cases.items():
        if "dense" in c:
            sc = _dense_scene(c["dense"], c["seed"])
        else:
            sc = make_scene(c["n"], seed=c["seed"], log_scale_mean=float(c["ls"]), n_groups=c.get("groups", 0))
        if c.get("inside"):
            cam = Camera(look_at_viewmat((0.15, -0.1, 0.2), target=(0.4, 0.1, -1.0)), intrinsics(c["f"], c["f"], c["w"] / 2.0, c["h"] / 2.0),
                         c["w"], c["h"])
        else:
            cam = ring_camera(c["w"], c["h"], c["f"], yaw_deg=20.0, elev=0.3)
        gRt = random_group_poses(c["groups"], seed=c["seed"]) if c.get("groups") else None
        kw = dict(quats=sc.quats, scales=sc.scales, sh_degree=sc.sh_degree)
        colors, cov6 = sc.sh, np.zeros((0, 6), np.float32)
        if c.get("door_b"):
            cov = ref_math.compute_cov(sc.quats, sc.scales).astype(np.float32)          # a5: what the loader hands to Door B
            cov6 = np.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], 1)
            colors = ref_math.sh2rgb(sc.sh[:, 0, :]).astype(np.float32)                  # a3: SH2RGB(features_dc), no clamp
            kw = dict(cov6=cov6, sh_degree=-1)
        out = np_twin.render(sc.means, sc.opacities, colors, cam.viewmat, cam.K, cam.width, cam.height,
                             group_id=sc.group_id, group_Rt=gRt, background=NERFSTUDIO_EVAL_BACKGROUND, depth_mode=0, **kw)
        P = out["proj"]
        np.savez_compressed(
            GOLD / f"render_twin_{name}.npz",
            means=sc.means, quats=sc.quats if not c.get("door_b") else np.zeros((0, 4), np.float32),
            scales=sc.scales if not c.get("door_b") else np.zeros((0, 3), np.float32), cov6=cov6,
            opacities=sc.opacities, colors=colors, sh_degree=np.int32(kw["sh_degree"]),
            group_id=sc.group_id if sc.group_id is not None else np.zeros(0, np.uint8),
            group_Rt=gRt if gRt is not None else np.zeros((0, 12), np.float32),
            viewmat=cam.viewmat, K=cam.K, wh=np.array([cam.width, cam.height]),
            background=np.array(NERFSTUDIO_EVAL_BACKGROUND, np.float32),
            rgb=out["rgb"].astype(np.float32), alpha=out["alpha"].astype(np.float32),
            depth=out["depth"].astype(np.float32), radii=P["radii"], valid=P["valid"],
            means2d=P["means2d"].astype(np.float32), conics=P["conics"].astype(np.float32),
            rgb_gauss=P["colors"].astype(np.float32), n_isect=np.array(out["n_isect"]))
        term = int(((1.0 - out["alpha"]) < 1e-3).sum())   # a stopped pixel keeps the T it had BEFORE the stopping splat
        print(f"{name}: visible {out['n_visible']}, intersections {out['n_isect']}, saturated pixels (T < 1e-3) {term} of {cam.width * cam.height}")


def gen_cfg_crop(cfg=3, view=0, tiles_x=8, tiles_y=6):
    """The float64 twin on a BASELINE config ITSELF (config 3: 1 M Gaussians, 1920x1080, view 0; also config 2, view 0 and
    config 5, view 0): a window of tiles_x x tiles_y tiles placed on the densest part of the frame (largest sum of list
    lengths, from the C oracle's tile offsets), rendered from the Gaussians whose tile rectangle touches it.  The scene is
    seeded (config_scene_and_cameras(cfg)), so the fixture holds the window and the twin's frames only."""
    import oracle
    from oracle import np_twin
    from sim_a_splat_amd.synthetic import NERFSTUDIO_EVAL_BACKGROUND, config_scene_and_cameras
    sc, cams = config_scene_and_cameras(cfg)
    cam = cams[view]
    ref = oracle.render(sc.means, sc.opacities, sc.sh, cam.viewmat, cam.K, cam.width, cam.height, quats=sc.quats, scales=sc.scales,
                        sh_degree=sc.sh_degree, background=NERFSTUDIO_EVAL_BACKGROUND, dump=True)
    tw, th = (cam.width + 15) // 16, (cam.height + 15) // 16
    cnt = np.diff(ref["tile_offsets"].astype(np.int64)).reshape(th, tw)
    S = np.zeros((th + 1, tw + 1), np.int64)
    S[1:, 1:] = cnt.cumsum(0).cumsum(1)
    win = S[tiles_y:, tiles_x:] - S[:-tiles_y, tiles_x:] - S[tiles_y:, :-tiles_x] + S[:-tiles_y, :-tiles_x]
    ty0, tx0 = np.unravel_index(int(win.argmax()), win.shape)
    crop = (int(tx0), int(ty0), int(tx0) + tiles_x, int(ty0) + tiles_y)
    out = np_twin.render(sc.means, sc.opacities, sc.sh, cam.viewmat, cam.K, cam.width, cam.height, quats=sc.quats, scales=sc.scales,
                         sh_degree=sc.sh_degree, background=NERFSTUDIO_EVAL_BACKGROUND, crop=crop)
    # (float64 and float32 rectangles may differ where a radius lands within rounding of a tile border -- two of 572 104 at config 5;
    # such a tile gets nothing from the Gaussian either way)
    n_oracle = int(cnt[crop[1]:crop[3], crop[0]:crop[2]].sum())
    assert abs(out["n_isect"] - n_oracle) <= max(2, n_oracle // 100_000), "twin and C oracle bin the window differently"
    np.savez_compressed(GOLD / f"render_twin_cfg{cfg}_crop.npz", config=np.int32(cfg), view=np.int32(view), crop_tiles=np.array(crop, np.int32),
                        wh=np.array([cam.width, cam.height]), background=np.array(NERFSTUDIO_EVAL_BACKGROUND, np.float32),
                        rgb=out["rgb"].astype(np.float32), alpha=out["alpha"].astype(np.float32), depth=out["depth"].astype(np.float32),
                        n_isect=np.int64(n_oracle), n_isect_twin=np.int64(out["n_isect"]), tile_lengths=cnt[crop[1]:crop[3], crop[0]:crop[2]].astype(np.int32))
    sat = int(((1.0 - out["alpha"]) < 1e-3).sum())
    print(f"cfg{cfg} view {view} crop tiles {crop}: {out['n_isect']} intersections (lists {cnt[crop[1]:crop[3], crop[0]:crop[2]].min()}..{cnt[crop[1]:crop[3], crop[0]:crop[2]].max()}), "
          f"saturated pixels {sat} of {out['alpha'].size}")


ASSETS = Path("/root/reference/assets/robots-scene-v2")


def gen_scene_assets():
    from sim_a_splat_amd import io
    d = ASSETS / "masks" / "xarm6-1"
    masks = io.load_link_masks(d / "link_masks_global_dict.npy")
    icp = io.load_icp_transformation(d / "icp_transformation.npy")
    jc = io.load_joint_config(d / "joint_config.npy")
    T, scale = io.load_dataparser_transforms(next((ASSETS / "splatfacto").glob("*/dataparser_transforms.json")))
    names = sorted(masks, key=lambda k: int(k[4:]))
    n = len(masks[names[0]])
    np.savez_compressed(GOLD / "scene_assets_xarm6_1.npz", icp_transformation=icp, joint_config=jc,
                        dataparser_transform=T, dataparser_scale=np.float64(scale), n=np.int64(n),
                        link_names=np.array(names), mask_bits=np.stack([np.packbits(masks[k]) for k in names]),
                        mask_counts=np.array([int(masks[k].sum()) for k in names], np.int64))
    print("scene_assets_xarm6_1.npz", n, [int(masks[k].sum()) for k in names])


def gen_scene_assets_divar():
    from sim_a_splat_amd import io
    base = Path("/root/reference/assets/divar113vhw")
    d = base / "masks" / "divar113vhw"
    masks = io.load_link_masks(d / "link_masks_global_dict.npy")
    icp = io.load_icp_transformation(d / "icp_transformation.npy")
    T, scale = io.load_dataparser_transforms(next((base / "splatfacto").glob("*/dataparser_transforms.json")))
    names = sorted(masks, key=lambda k: int(k[4:]))
    n = len(masks[names[0]])
    np.savez_compressed(GOLD / "scene_assets_divar113vhw.npz", icp_transformation=icp,
                        polygon_bounds=np.load(d / "polygon_bounds.npy", allow_pickle=False),
                        trans_init=np.load(d / "trans_init.npy", allow_pickle=False),
                        dataparser_transform=T, dataparser_scale=np.float64(scale), n=np.int64(n),
                        link_names=np.array(names), mask_bits=np.stack([np.packbits(masks[k]) for k in names]),
                        mask_counts=np.array([int(masks[k].sum()) for k in names], np.int64))
    print("scene_assets_divar113vhw.npz", n, [int(masks[k].sum()) for k in names])


def gen_run_files():
    for scene in ("divar113vhw", "robots-scene-v2"):
        base = Path("/root/reference/assets") / scene
        run = next((base / "splatfacto").glob("*/config.yml")).parent
        raw = lambda p: np.frombuffer(Path(p).read_bytes(), dtype=np.uint8)
        np.savez_compressed(GOLD / f"ns_run_{scene}.npz", scene=np.array(scene), timestamp=np.array(run.name),
                            config_yml=raw(run / "config.yml"), dataparser_transforms_json=raw(run / "dataparser_transforms.json"),
                            transforms_json=raw(base / "transforms.json"))
        print(f"ns_run_{scene}.npz", run.name)


if __name__ == "__main__":
    GOLD.mkdir(parents=True, exist_ok=True)
    if REF.exists():
        gen_compute_cov()
    else:
        print("reference tree absent: compute_cov.npz not regenerated")
    if ASSETS.exists():
        gen_scene_assets()
        gen_scene_assets_divar()
        gen_run_files()
    CROPS = {"--cfg3-crop": (3, 0, 8, 6), "--cfg2-crop": (2, 0, 8, 6), "--cfg5-crop": (5, 0, 6, 4)}
    asked = [a for a in sys.argv if a in CROPS]
    if asked:
        for a in asked:
            gen_cfg_crop(*CROPS[a])
    elif "--assets-only" not in sys.argv:
        gen_twin_renders()
        for a in CROPS:
            gen_cfg_crop(*CROPS[a])
