"""Generate tests/golden/*.npz.  Run from the repo root in the build container:

    python oracle/make_golden.py

(1) compute_cov.npz  -- inputs and outputs of the REFERENCE's own compute_cov, obtained by
    importing /root/reference/sim_a_splat/ellipsoids/covariance_utils.py by file path (a
    torch-only module).  Only data is stored, no reference source.
(2) render_twin_*.npz -- tiny seeded scenes rendered by the float64 NumPy twin; they pin the
    C oracle (tests/test_oracle.py) and, through it, the HIP kernels.
(3) scene_assets_xarm6_1.npz -- the small data files the reference ships for its real scene
    (robots-scene-v2 / xarm6-1): ICP similarity, mask-time joint configuration, dataparser
    transform and the seven per-link masks, bit-packed.  Data only; the masks' pickled .npy is
    read with the whitelisting unpickler of sim_a_splat_amd.io, never np.load(allow_pickle=True).
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


def gen_twin_renders():
    from oracle import np_twin
    from sim_a_splat_amd.synthetic import make_scene, ring_camera, random_group_poses, NERFSTUDIO_EVAL_BACKGROUND
    cases = {
        "one": dict(n=1, seed=11, w=32, h=32, f=40.0, ls=np.log(0.15)),
        "two": dict(n=2, seed=12, w=48, h=32, f=40.0, ls=np.log(0.2)),
        "n64": dict(n=64, seed=13, w=64, h=64, f=60.0, ls=np.log(0.08)),
        "n2k": dict(n=2000, seed=14, w=128, h=96, f=120.0, ls=np.log(0.03)),
        "n2k_groups": dict(n=2000, seed=15, w=100, h=70, f=90.0, ls=np.log(0.03), groups=5),
    }
    for name, c in cases.items():
        sc = make_scene(c["n"], seed=c["seed"], log_scale_mean=float(c["ls"]), n_groups=c.get("groups", 0))
        cam = ring_camera(c["w"], c["h"], c["f"], yaw_deg=20.0, elev=0.3)
        gRt = random_group_poses(c["groups"], seed=c["seed"]) if c.get("groups") else None
        out = np_twin.render(sc.means, sc.opacities, sc.sh, cam.viewmat, cam.K, cam.width, cam.height,
                             quats=sc.quats, scales=sc.scales, sh_degree=3, group_id=sc.group_id, group_Rt=gRt,
                             background=NERFSTUDIO_EVAL_BACKGROUND, depth_mode=0)
        P = out["proj"]
        np.savez_compressed(
            GOLD / f"render_twin_{name}.npz",
            means=sc.means, quats=sc.quats, scales=sc.scales, opacities=sc.opacities, sh=sc.sh,
            group_id=sc.group_id if sc.group_id is not None else np.zeros(0, np.uint8),
            group_Rt=gRt if gRt is not None else np.zeros((0, 12), np.float32),
            viewmat=cam.viewmat, K=cam.K, wh=np.array([cam.width, cam.height]),
            background=np.array(NERFSTUDIO_EVAL_BACKGROUND, np.float32),
            rgb=out["rgb"].astype(np.float32), alpha=out["alpha"].astype(np.float32),
            depth=out["depth"].astype(np.float32), radii=P["radii"], valid=P["valid"],
            means2d=P["means2d"].astype(np.float32), conics=P["conics"].astype(np.float32),
            colors=P["colors"].astype(np.float32), n_isect=np.array(out["n_isect"]))
        print(name, out["n_visible"], out["n_isect"])


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


if __name__ == "__main__":
    GOLD.mkdir(parents=True, exist_ok=True)
    if REF.exists():
        gen_compute_cov()
    else:
        print("reference tree absent: compute_cov.npz not regenerated")
    if ASSETS.exists():
        gen_scene_assets()
    gen_twin_renders()
