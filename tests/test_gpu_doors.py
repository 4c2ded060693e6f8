"""GPU: the two reference doors (host-side mirror -> C ABI -> HIP) against the oracle."""
import types

import numpy as np
import pytest
import torch

import oracle
from oracle import ref_math
from sim_a_splat_amd import poses
from sim_a_splat_amd.poses import SE3
from sim_a_splat_amd.synthetic import NERFSTUDIO_EVAL_BACKGROUND as BG
from sim_a_splat_amd.synthetic import c2w_opengl_from_viewmat, make_scene, ring_camera

pytestmark = pytest.mark.gpu


def test_door_a_gaussian_splat_render():
    """GaussianSplat.render(pose): raw splatfacto params in, nerfstudio output dict out (row a1/T0)."""
    from sim_a_splat_amd.gaussian_splat import GaussianSplat, PinholeCamera, SplatModel, viewmat_from_c2w_opengl
    sc = make_scene(4000, seed=201, log_scale_mean=float(np.log(0.03)))
    cam = ring_camera(200, 150, 170.0, yaw_deg=30.0, elev=0.5)
    raw_scales = np.log(sc.scales)
    raw_opac = np.log(sc.opacities / (1 - sc.opacities)).reshape(-1, 1)
    model = SplatModel(sc.means, raw_scales, sc.quats, sc.sh[:, 0], sc.sh[:, 1:], raw_opac, sh_degree=3, device="cuda:0")
    K = cam.K
    pc = PinholeCamera(torch.eye(4)[None, :3], float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), cam.width, cam.height)
    gs = GaussianSplat.from_model(model, pc)
    H, W, Kt = gs.get_camera_intrinsics()
    assert (H, W) == (150, 200) and np.allclose(Kt.numpy(), K)
    pose = torch.from_numpy(c2w_opengl_from_viewmat(cam.viewmat))
    out = gs.render(pose)
    assert set(out) == {"rgb", "depth", "accumulation", "background"}
    assert out["rgb"].shape == (150, 200, 3) and out["depth"].shape == (150, 200, 1) and out["background"].shape == (150, 200, 3)
    # oracle on the same activated parameters and the same T0 view matrix
    act_scales = torch.exp(torch.from_numpy(raw_scales)).numpy()
    act_op = torch.sigmoid(torch.from_numpy(raw_opac)).reshape(-1).numpy()
    V = viewmat_from_c2w_opengl(pose)
    ref = oracle.render(sc.means, act_op, sc.sh, V, K, 200, 150, quats=sc.quats, scales=act_scales, sh_degree=3,
                        background=BG, depth_mode=1)
    assert np.abs(out["rgb"].cpu().numpy() - ref["rgb"]).max() <= 1e-4
    assert np.abs(out["accumulation"].cpu().numpy() - ref["alpha"]).max() <= 1e-4
    assert np.array_equal(out["rgb"].cpu().numpy(), ref["rgb"]) and np.array_equal(out["depth"].cpu().numpy(), ref["depth"])
    rgb, pts, _, mask, _ = gs.generate_RGBD_point_cloud(pose, max_depth=3.0)
    d = out["depth"].squeeze()
    assert pts.shape == (150, 200, 3) and torch.equal(pts[..., 2], d) and torch.equal(mask, d < 3.0)
    # fused depth tail against the oracle's unprojection of the oracle's depth: bit-exact
    opts, omask = oracle.unproject(ref["depth"], K, max_depth=3.0)
    assert np.array_equal(pts.cpu().numpy(), opts) and np.array_equal(mask.cpu().numpy(), omask)
    assert 0 < omask.sum() < omask.size
    assert torch.equal(rgb, out["rgb"])
    _, pts2, _, mask2, _ = gs.generate_RGBD_point_cloud(pose, max_depth=None)
    assert mask2.all() and torch.equal(pts2, pts)


def test_render_rgbd_without_fill_and_with_partial_outputs():
    """sas_render_rgbd: raw expected depth (no fill) unprojected; points / mask individually optional."""
    import ctypes
    from sim_a_splat_amd import _capi
    from sim_a_splat_amd.rasterizer import Rasterizer
    sc = make_scene(3000, seed=77, log_scale_mean=float(np.log(0.03)))
    cam = ring_camera(97, 61, 80.0, yaw_deg=10.0)
    r = Rasterizer("cuda:0")
    r.upload(sc.means, sc.opacities, sc.sh, quats=sc.quats, scales=sc.scales, sh_degree=3)
    out = r.render_rgbd(cam.viewmat, cam.K, 97, 61, BG, max_depth=2.9, depth_fill_max=False)
    ref = oracle.render_scene(sc, cam, background=BG, depth_mode=0)
    assert np.array_equal(out["depth"].cpu().numpy(), ref["depth"]) and np.array_equal(out["rgb"].cpu().numpy(), ref["rgb"])
    opts, omask = oracle.unproject(ref["depth"], cam.K, max_depth=2.9)
    assert np.array_equal(out["points"].cpu().numpy(), opts) and np.array_equal(out["mask"].cpu().numpy(), omask)
    # mask only / points only, and the error path: points without depth
    L = _capi.lib()
    V = np.ascontiguousarray(cam.viewmat, np.float32)
    Kc = np.ascontiguousarray(cam.K, np.float32)
    depth = torch.empty((61, 97, 1), device="cuda:0")
    mask8 = torch.zeros((61, 97), dtype=torch.uint8, device="cuda:0")
    md = ctypes.c_float(2.9)
    rc = L.sas_render_rgbd(r._ctx, V.ctypes.data, Kc.ctypes.data, 97, 61, None, 0, ctypes.addressof(md), None, None,
                           depth.data_ptr(), None, mask8.data_ptr(), None)
    assert rc == 0 and np.array_equal(mask8.cpu().numpy().astype(bool), omask)
    pts = torch.empty((61, 97, 3), device="cuda:0")
    rc = L.sas_render_rgbd(r._ctx, V.ctypes.data, Kc.ctypes.data, 97, 61, None, 0, None, None, None, None,
                           pts.data_ptr(), None, None)
    assert rc == _capi.SAS_ERR_INVALID if hasattr(_capi, "SAS_ERR_INVALID") else rc == -1
    assert b"depth" in L.sas_last_error(r._ctx)


def _fake_msg(rng, n_links, robot_num=3):
    q = rng.normal(size=(n_links, 4))
    return types.SimpleNamespace(num_links=n_links, robot_num=[robot_num] * n_links, position=rng.normal(0, 0.05, size=(n_links, 3)).tolist(),
                                 quaternion=(q / np.linalg.norm(q, axis=1, keepdims=True)).tolist(),
                                 link_name=[f"plant::link{i}" for i in range(n_links)])


def test_door_b_handler_groups_and_get_render():
    """add_gaussian_splats groups + draw_handler poses + get_render uint8 frames (rows a7-a11, T7)."""
    from sim_a_splat_amd.covariance import compute_cov, sh2rgb
    from sim_a_splat_amd.handler import CameraRig, SplatHandler
    rng = np.random.default_rng(5)
    sc = make_scene(6000, seed=202, log_scale_mean=float(np.log(0.03)))
    covs = compute_cov(torch.from_numpy(sc.quats), torch.from_numpy(sc.scales)).numpy()
    colors = np.clip(sh2rgb(torch.from_numpy(sc.sh[:, 0])).numpy(), 0, 1)
    K_links = 4
    gid = rng.integers(0, K_links + 1, size=sc.n)                      # K_links = "no link"
    masks = {f"link{i}": gid == i for i in range(K_links)}
    icp = np.eye(4)
    icp[:3, :3] = 0.9 * ref_math.quat_wxyz_to_R(rng.normal(size=4))
    icp[:3, 3] = [0.02, -0.01, 0.03]
    fk = []
    for _ in range(K_links):
        T = np.eye(4)
        T[:3, :3] = ref_math.quat_wxyz_to_R(rng.normal(size=4))
        T[:3, 3] = rng.normal(0, 0.05, size=3)
        fk.append(T)
    h = SplatHandler.from_arrays(sc.means, covs, colors, sc.opacities, masks, icp, fk, device=0)
    msg = _fake_msg(rng, K_links)
    h.draw_handler(msg)
    rig = CameraRig({0: {"link_name": "world", "local_frame": ((0.0, 1.0, 0.0, 0.0), (0.0, 0.0, 3.0)), "type": "viewport", "render_size": [60, 80]},
                     1: {"link_name": "link2", "local_frame": ((1.0, 0, 0, 0), (0.0, 0.0, -3.0)), "type": "moving", "render_size": [48, 64]}})
    obs = rig.get_obs(h, msg)
    assert obs["camera_0"].shape == (3, 48, 64) and obs["camera_1"].shape == (3, 60, 80) and obs["camera_0"].dtype == np.uint8
    # oracle: same groups in the handler's registration order (links, then the static rest), same poses
    order = np.concatenate([np.nonzero(masks[f"link{i}"])[0] for i in range(K_links)] + [np.nonzero(gid == K_links)[0]])
    group_of = np.concatenate([np.full((gid == i).sum(), i, np.uint8) for i in range(K_links)] + [np.full((gid == K_links).sum(), K_links, np.uint8)])
    s, Ri, ti = poses.decompose_icp(icp)
    Rt = []
    for i in range(K_links):
        R, t = ref_math.link_splat_pose(Ri, ti, s, fk[i][:3, :3], fk[i][:3, 3], msg.quaternion[i], msg.position[i])
        Rq = poses.quat_wxyz_to_matrix(poses.matrix_to_quat_wxyz(R))          # the handle stores a quaternion
        Rt.append(poses.rt_to_row12(Rq, t))
    Rt.append(poses.rt_to_row12(np.eye(3), np.zeros(3)))
    cov6 = np.stack([covs[:, 0, 0], covs[:, 0, 1], covs[:, 0, 2], covs[:, 1, 1], covs[:, 1, 2], covs[:, 2, 2]], 1)[order]
    cams = rig.poses(h, msg)
    seen = []
    for key, (wxyz, pos), (H, W) in zip(("camera_0", "camera_1"), cams, rig.sizes()):
        V, K = h.scene._view_and_K(H, W, wxyz, pos, h.scene.camera.fov)
        ref = oracle.render(sc.means[order], sc.opacities[order], colors[order], V, K, W, H, cov6=cov6, sh_degree=-1,
                            group_id=group_of, group_Rt=np.stack(Rt), background=(0, 0, 0), want_rgb8=True)
        got = np.moveaxis(obs[key], 0, -1)
        assert np.abs(got.astype(int) - ref["rgb8"].astype(int)).max() <= 1
        assert np.array_equal(got, ref["rgb8"])
        seen.append(ref["n_visible"])
    assert max(seen) > 100          # the viewport camera looks at the scene (the link camera may not)
    h.scene.close()


def test_door_b_with_the_shipped_masks_and_icp(golden_dir):
    """cfg2 variant: N = 113,831 synthetic Gaussians (the count of robots-scene-v2) partitioned by
    the reference's own xarm6-1 link masks, its ICP similarity, URDF-style FK, 640x480 frames."""
    from sim_a_splat_amd import urdf_fk
    from sim_a_splat_amd.covariance import compute_cov, sh2rgb
    from sim_a_splat_amd.handler import SplatHandler
    with np.load(golden_dir / "scene_assets_xarm6_1.npz") as z:
        a = {k: z[k] for k in z.files}
    n = int(a["n"])
    masks = {str(k): np.unpackbits(b, count=n).astype(bool) for k, b in zip(a["link_names"], a["mask_bits"])}
    rng = np.random.default_rng(11)
    sc = make_scene(n, seed=2, log_scale_mean=float(np.log(0.012)))
    covs = compute_cov(torch.from_numpy(sc.quats), torch.from_numpy(sc.scales)).numpy()
    colors = np.clip(sh2rgb(torch.from_numpy(sc.sh[:, 0])).numpy(), 0, 1)
    chain = "".join(f'<link name="l{i}"><visual><geometry><mesh filename="l{i}.obj"/></geometry></visual></link>' for i in range(7))
    chain += "".join(f'<joint name="j{i}" type="revolute"><parent link="l{i - 1}"/><child link="l{i}"/>'
                     f'<origin xyz="0.05 0 0.1" rpy="{0.3 * i} 0 0"/><axis xyz="0 0 1"/></joint>' for i in range(1, 7))
    fk = urdf_fk.visual_mesh_fk(urdf_fk.load(f"<robot>{chain}</robot>"), a["joint_config"])
    assert len(fk) == 7
    h = SplatHandler.from_arrays(sc.means, covs, colors, sc.opacities, masks, a["icp_transformation"], fk, device=0)
    msg = _fake_msg(rng, 7)
    h.draw_handler(msg)
    cam_q, cam_p = (0.0, 1.0, 0.0, 0.0), (0.0, 0.0, 3.0)
    frame = h.render(h.scene, [(np.array(cam_q), np.array(cam_p))], [[480, 640]])[0]
    assert frame.shape == (480, 640, 3) and frame.dtype == np.uint8
    # oracle: registration order = links (a Gaussian in two masks is registered twice), then the rest
    idx = [np.nonzero(masks[f"link{i}"])[0] for i in range(7)]
    rest = np.nonzero(~np.logical_or.reduce(list(masks.values())))[0]
    order = np.concatenate(idx + [rest])
    group_of = np.concatenate([np.full(len(ix), i, np.uint8) for i, ix in enumerate(idx)] + [np.full(len(rest), 7, np.uint8)])
    s, Ri, ti = poses.decompose_icp(a["icp_transformation"])
    Rt = []
    for i in range(7):
        R, t = ref_math.link_splat_pose(Ri, ti, s, fk[i][:3, :3], fk[i][:3, 3], msg.quaternion[i], msg.position[i])
        Rt.append(poses.rt_to_row12(poses.quat_wxyz_to_matrix(poses.matrix_to_quat_wxyz(R)), t))
    Rt.append(poses.rt_to_row12(np.eye(3), np.zeros(3)))
    cov6 = np.stack([covs[:, 0, 0], covs[:, 0, 1], covs[:, 0, 2], covs[:, 1, 1], covs[:, 1, 2], covs[:, 2, 2]], 1)[order]
    V, K = h.scene._view_and_K(480, 640, cam_q, cam_p, h.scene.camera.fov)
    ref = oracle.render(sc.means[order], sc.opacities[order], colors[order], V, K, 640, 480, cov6=cov6, sh_degree=-1,
                        group_id=group_of, group_Rt=np.stack(Rt), background=(0, 0, 0), want_rgb8=True)
    assert ref["n_visible"] > 50_000
    assert np.array_equal(frame, ref["rgb8"])
    h.scene.close()


def test_handler_from_assets_files(golden_dir, tmp_path):
    """SplatHandler.from_assets: the reference's constructor flow from files -- JSON scene through
    GSplatLoader, pickled link masks, ICP and joint-config .npy, a URDF -- with an AABB crop; one frame
    against the oracle on the cropped, regrouped arrays."""
    import json
    from sim_a_splat_amd import urdf_fk
    from sim_a_splat_amd.covariance import GSplatLoader
    from sim_a_splat_amd.handler import SplatHandler, aabb_mask
    with np.load(golden_dir / "scene_assets_xarm6_1.npz") as z:
        a = {k: z[k] for k in z.files}
    n = 6000
    rng = np.random.default_rng(21)
    sc = make_scene(n, seed=21, log_scale_mean=float(np.log(0.03)))
    (tmp_path / "scene.json").write_text(json.dumps({
        "means": sc.means.tolist(), "rotations": sc.quats.tolist(), "colors": rng.uniform(0, 1, size=(n, 3)).tolist(),
        "opacities": np.log(sc.opacities / (1 - sc.opacities)).reshape(-1, 1).tolist(), "scalings": np.log(sc.scales).tolist()}))
    gid = rng.integers(0, 8, size=n)                                     # 7 = no link
    np.save(tmp_path / "link_masks_global_dict.npy", {f"link{i}": gid == i for i in range(7)}, allow_pickle=True)
    np.save(tmp_path / "icp_transformation.npy", a["icp_transformation"])
    np.save(tmp_path / "joint_config.npy", a["joint_config"])
    chain = "".join(f'<link name="l{i}"><visual><geometry><mesh filename="l{i}.obj"/></geometry></visual></link>' for i in range(7))
    chain += "".join(f'<joint name="j{i}" type="revolute"><parent link="l{i - 1}"/><child link="l{i}"/>'
                     f'<origin xyz="0.04 0 0.08" rpy="0 {0.2 * i} 0"/><axis xyz="0 0 1"/></joint>' for i in range(1, 7))
    (tmp_path / "robot.urdf").write_text(f"<robot>{chain}</robot>")
    loader = GSplatLoader.from_json(tmp_path / "scene.json")
    bounds = np.array([[-0.8, 0.8], [-0.8, 0.8], [-0.7, 1.0]], np.float32)
    h = SplatHandler.from_assets(loader, tmp_path, tmp_path / "robot.urdf", bounds=bounds, device=0)
    msg = _fake_msg(rng, 7)
    h.draw_handler(msg)
    cam_q, cam_p = (0.0, 1.0, 0.0, 0.0), (0.0, 0.0, 3.0)
    frame = h.render(h.scene, [SE3(np.concatenate((cam_q, cam_p)))], [[120, 160]])[0]
    # oracle on the same crop and registration order
    keep = aabb_mask(sc.means, bounds)
    assert 0.3 * n < keep.sum() < n
    means, covs = loader.means.numpy()[keep], loader.covs.numpy()[keep]
    cols, ops, g = loader.colors.numpy()[keep], loader.opacities.numpy()[keep].reshape(-1), gid[keep]
    order = np.concatenate([np.nonzero(g == i)[0] for i in range(8)])
    group_of = np.concatenate([np.full((g == i).sum(), i, np.uint8) for i in range(8)])
    fk = urdf_fk.visual_mesh_fk(urdf_fk.load(tmp_path / "robot.urdf"), a["joint_config"])
    s, Ri, ti = poses.decompose_icp(a["icp_transformation"])
    Rt = []
    for i in range(7):
        R, t = ref_math.link_splat_pose(Ri, ti, s, fk[i][:3, :3], fk[i][:3, 3], msg.quaternion[i], msg.position[i])
        Rt.append(poses.rt_to_row12(poses.quat_wxyz_to_matrix(poses.matrix_to_quat_wxyz(R)), t))
    Rt.append(poses.rt_to_row12(np.eye(3), np.zeros(3)))
    cov6 = np.stack([covs[:, 0, 0], covs[:, 0, 1], covs[:, 0, 2], covs[:, 1, 1], covs[:, 1, 2], covs[:, 2, 2]], 1)[order]
    V, K = h.scene._view_and_K(120, 160, cam_q, cam_p, h.scene.camera.fov)
    ref = oracle.render(means[order], ops[order], cols[order], V, K, 160, 120, cov6=cov6, sh_degree=-1, group_id=group_of,
                        group_Rt=np.stack(Rt), background=(0, 0, 0), want_rgb8=True)
    assert ref["n_visible"] > 500 and np.array_equal(frame, ref["rgb8"])
    h.scene.close()


def _door_b_setup(n, seed, k_links):
    """A synthetic Door-B scene: Gaussians with 3x3 covariances + RGB, `k_links` link masks, an ICP similarity
    and FK poses; returns what SplatHandler.from_arrays takes and what the oracle needs to follow it."""
    from sim_a_splat_amd.covariance import compute_cov, sh2rgb
    rng = np.random.default_rng(seed)
    sc = make_scene(n, seed=seed, log_scale_mean=float(np.log(0.03)))
    covs = compute_cov(torch.from_numpy(sc.quats), torch.from_numpy(sc.scales)).numpy()
    colors = np.clip(sh2rgb(torch.from_numpy(sc.sh[:, 0])).numpy(), 0, 1)
    gid = rng.integers(0, k_links + 1, size=sc.n)                      # k_links = "no link"
    masks = {f"link{i}": gid == i for i in range(k_links)}
    icp = np.eye(4)
    icp[:3, :3] = 0.9 * ref_math.quat_wxyz_to_R(rng.normal(size=4))
    icp[:3, 3] = [0.02, -0.01, 0.03]
    fk = []
    for _ in range(k_links):
        T = np.eye(4)
        T[:3, :3] = ref_math.quat_wxyz_to_R(rng.normal(size=4))
        T[:3, 3] = rng.normal(0, 0.05, size=3)
        fk.append(T)
    order = np.concatenate([np.nonzero(gid == i)[0] for i in range(k_links + 1)])
    group_of = np.concatenate([np.full((gid == i).sum(), i, np.uint8) for i in range(k_links + 1)])
    cov6 = np.stack([covs[:, 0, 0], covs[:, 0, 1], covs[:, 0, 2], covs[:, 1, 1], covs[:, 1, 2], covs[:, 2, 2]], 1)
    return dict(sc=sc, covs=covs, colors=colors, masks=masks, icp=icp, fk=fk, order=order, group_of=group_of, cov6=cov6, rng=rng)


def _oracle_door_b(d, msg, wxyz, pos, H, W, fov, scene):
    s, Ri, ti = poses.decompose_icp(d["icp"])
    Rt = []
    for i in range(len(d["fk"])):
        R, t = ref_math.link_splat_pose(Ri, ti, s, d["fk"][i][:3, :3], d["fk"][i][:3, 3], msg.quaternion[i], msg.position[i])
        Rt.append(poses.rt_to_row12(poses.quat_wxyz_to_matrix(poses.matrix_to_quat_wxyz(R)), t))   # the handle stores a quaternion
    Rt.append(poses.rt_to_row12(np.eye(3), np.zeros(3)))
    V, K = scene._view_and_K(H, W, wxyz, pos, fov)
    o, sc = d["order"], d["sc"]
    return oracle.render(sc.means[o], sc.opacities[o], d["colors"][o], V, K, W, H, cov6=d["cov6"][o], sh_degree=-1,
                         group_id=d["group_of"], group_Rt=np.stack(Rt), background=(0, 0, 0), want_rgb8=True)


def test_splat_env_wrapper_step_against_the_oracle():
    """SplatEnvWrapper (row a11) end to end on the GPU: reset, two steps with the reference's camera dictionary
    (SE3 local frames, a viewport and an eye-in-hand camera of one size -> one batched render): every
    camera_i of every step equals the oracle's uint8 frame for that step's link poses and camera pose."""
    from sim_a_splat_amd.env_wrapper import SplatEnvWrapper
    from sim_a_splat_amd.handler import SplatHandler
    K_links = 3
    d = _door_b_setup(5000, 303, K_links)
    h = SplatHandler.from_arrays(d["sc"].means, d["covs"], d["colors"], d["sc"].opacities, d["masks"], d["icp"], d["fk"], device=0)

    class Inner:
        visualize_robot_flag = False

        def __init__(self):
            self.msgs = [_fake_msg(np.random.default_rng(50 + k), K_links) for k in range(3)]
            self.t = 0

        def reset(self, seed=None, reset_to_state=None):
            self.t = 0

        def step(self, action):
            self.t += 1
            return {}, 0.0, False, False, {}

        def render(self):
            pass

        def _get_obs(self):
            return {"robot_pos": np.zeros(2)}

        def _generate_draw_msg(self):
            return self.msgs[self.t]

        def close(self):
            pass

    inner = Inner()
    env = SplatEnvWrapper(inner, splat_handler=h)
    info = {0: {"link_name": "world", "local_frame": SE3(wxyz_xyz=np.array([0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 3.0])), "type": "viewport",
                "render_size": [60, 80]},
            1: {"link_name": "link2", "local_frame": SE3(wxyz_xyz=np.array([0.0, 1.0, 0.0, 0.0, 0.0, 0.1, 2.8])), "type": "moving",
                "render_size": [60, 80]}}
    env._configure_cameras(info)
    env.reset()
    for t in (1, 2):
        obs, *_ = env.step(None)
        msg = inner.msgs[t]
        cam_poses = [h.get_attached_frame("link2", info[1]["local_frame"], msg), poses.pose_wxyz_xyz(info[0]["local_frame"])]
        for i, (wxyz, pos) in enumerate(cam_poses):          # camera_0 = the moving camera, camera_1 = the viewport
            ref = _oracle_door_b(d, msg, wxyz, pos, 60, 80, h.scene.camera.fov, h.scene)
            assert np.array_equal(np.moveaxis(obs[f"camera_{i}"], 0, -1), ref["rgb8"]), (t, i)
        assert obs["camera_1"].max() > 30                    # the viewport sees the scene
    env.close()


def test_viser_bridge_pushes_hip_frames_of_the_client_camera():
    """Row f4 on the GPU: ViserBridge over a real SplatScene and the stand-in server -- the background image
    pushed to a client is the HIP frame of THAT client's camera (pose, vertical fov, aspect -> width / K) and
    equals the oracle's uint8 frame; a camera update and a group-pose change + refresh() push new frames."""
    from conftest import FakeViserClient, FakeViserServer
    from sim_a_splat_amd.handler import SplatHandler
    from sim_a_splat_amd.viser_bridge import ViserBridge
    K_links = 2
    d = _door_b_setup(4000, 404, K_links)
    h = SplatHandler.from_arrays(d["sc"].means, d["covs"], d["colors"], d["sc"].opacities, d["masks"], d["icp"], d["fk"], device=0)
    msg0 = _fake_msg(np.random.default_rng(1), K_links)
    h.draw_handler(msg0)
    server = FakeViserServer()
    bridge = ViserBridge(server, h.scene, height=90, max_width=400)
    client = FakeViserClient(7)
    client.camera.fov, client.camera.aspect = 0.9, 1.6                # -> 144 x 90
    server.connect[0](client)                                          # a browser connects
    img, kw = client.images[-1]
    assert img.shape == (90, 144, 3) and img.dtype == np.uint8 and kw["format"] == "jpeg"
    ref = _oracle_door_b(d, msg0, client.camera.wxyz, client.camera.position, 90, 144, 0.9, h.scene)
    assert ref["n_visible"] > 100 and np.array_equal(img, ref["rgb8"])
    client.camera.move([0.3, -0.2, 2.5])                               # the user orbits: on_update -> a new frame
    ref = _oracle_door_b(d, msg0, client.camera.wxyz, client.camera.position, 90, 144, 0.9, h.scene)
    assert len(client.images) == 2 and np.array_equal(client.images[-1][0], ref["rgb8"])
    msg1 = _fake_msg(np.random.default_rng(2), K_links)
    h.draw_handler(msg1)                                               # the robot moves
    assert bridge.refresh() == 1
    ref = _oracle_door_b(d, msg1, client.camera.wxyz, client.camera.position, 90, 144, 0.9, h.scene)
    assert np.array_equal(client.images[-1][0], ref["rgb8"]) and not np.array_equal(client.images[-1][0], client.images[-2][0])
    server.disconnect[0](client)
    assert bridge.refresh() == 0
    h.scene.close()


# ---- the reference's own run files and segmentation (tests/golden/ns_run_*, scene_assets_*) ------------------------
def test_door_a_from_the_reference_run_directory(tmp_path, monkeypatch):
    """GaussianSplat(config_path, ...) constructed as the reference constructs it -- on its shipped divar113vhw
    config.yml / transforms.json (only the checkpoint is fabricated: the real one is a Git-LFS pointer) -- and
    rendered at the reference's real intrinsics (1080x1920, fx 1787.17) from one of its real camera poses,
    against the oracle."""
    from conftest import fabricated_gauss_params, materialize_run
    from sim_a_splat_amd.gaussian_splat import GaussianSplat, viewmat_from_c2w_opengl
    n = 60_000
    g = fabricated_gauss_params(n, 31, spread=0.3)
    cfg_path = materialize_run("divar113vhw", tmp_path, g)
    monkeypatch.chdir(tmp_path)
    gs = GaussianSplat(cfg_path.relative_to(tmp_path), res_factor=None, test_mode="inference", dataset_mode="test", device="cuda:0")
    H, W, K = gs.get_camera_intrinsics()
    assert (H, W) == (1920, 1080)
    # the eval pose that sees most of the stand-in cloud
    poses_c2w = gs.get_poses()
    best, seen = 0, -1
    for k in range(poses_c2w.shape[0]):
        V = viewmat_from_c2w_opengl(poses_c2w[k])
        pc = g["means"] @ V[:3, :3].T + V[:3, 3]
        u, v = K[0, 0].item() * pc[:, 0] / pc[:, 2] + K[0, 2].item(), K[1, 1].item() * pc[:, 1] / pc[:, 2] + K[1, 2].item()
        cnt = int(((pc[:, 2] > 0.05) & (u > 0) & (u < W) & (v > 0) & (v < H)).sum())
        best, seen = (k, cnt) if cnt > seen else (best, seen)
    assert seen > n // 10
    pose = poses_c2w[best]
    out = gs.render(pose)
    assert out["rgb"].shape == (1920, 1080, 3) and out["depth"].shape == (1920, 1080, 1)
    V = viewmat_from_c2w_opengl(pose)
    sh = np.concatenate([g["features_dc"][:, None, :], g["features_rest"]], 1)
    ref = oracle.render(g["means"], torch.sigmoid(torch.from_numpy(g["opacities"])).reshape(-1).numpy(),
                        sh, V, K.numpy(), W, H, quats=g["quats"], scales=torch.exp(torch.from_numpy(g["scales"])).numpy(), sh_degree=3,
                        background=BG, depth_mode=1)
    assert ref["n_visible"] > n // 10
    assert np.abs(out["rgb"].cpu().numpy() - ref["rgb"]).max() <= 1e-4
    assert np.array_equal(out["rgb"].cpu().numpy(), ref["rgb"]) and np.array_equal(out["accumulation"].cpu().numpy(), ref["alpha"])
    assert np.array_equal(out["depth"].cpu().numpy(), ref["depth"])
    gs.pipeline.model._rasterizer().close()


def _real_partition_handler(seed):
    """Door B on the reference's divar113vhw segmentation: 292,247 Gaussians split by its six shipped link masks
    (+ the static rest) and posed through its shipped ICP similarity; Gaussians, FK poses and the draw message are
    synthetic (checkpoint: LFS pointer; the scene ships no joint_config.npy)."""
    from conftest import real_link_masks
    from sim_a_splat_amd.covariance import compute_cov, sh2rgb
    from sim_a_splat_amd.handler import SplatHandler
    masks, icp, n = real_link_masks("divar113vhw")
    assert n == 292_247 and len(masks) == 6
    rng = np.random.default_rng(seed)
    sc = make_scene(n, seed=2, log_scale_mean=float(np.log(0.012)))
    covs = compute_cov(torch.from_numpy(sc.quats), torch.from_numpy(sc.scales)).numpy()
    colors = np.clip(sh2rgb(torch.from_numpy(sc.sh[:, 0])).numpy(), 0, 1)
    fk = []
    for _ in range(6):
        T = np.eye(4)
        T[:3, :3] = ref_math.quat_wxyz_to_R(rng.normal(size=4))
        T[:3, 3] = rng.normal(0, 0.05, size=3)
        fk.append(T)
    h = SplatHandler.from_arrays(sc.means, covs, colors, sc.opacities, masks, icp, fk, device=0)
    msg = _fake_msg(rng, 6)
    h.draw_handler(msg)
    idx = [np.nonzero(masks[f"link{i}"])[0] for i in range(6)]
    rest = np.nonzero(~np.logical_or.reduce(list(masks.values())))[0]
    order = np.concatenate(idx + [rest])
    group_of = np.concatenate([np.full(len(ix), i, np.uint8) for i, ix in enumerate(idx)] + [np.full(len(rest), 6, np.uint8)])
    s, Ri, ti = poses.decompose_icp(icp)
    Rt = []
    for i in range(6):
        R, t = ref_math.link_splat_pose(Ri, ti, s, fk[i][:3, :3], fk[i][:3, 3], msg.quaternion[i], msg.position[i])
        Rt.append(poses.rt_to_row12(poses.quat_wxyz_to_matrix(poses.matrix_to_quat_wxyz(R)), t))
    Rt.append(poses.rt_to_row12(np.eye(3), np.zeros(3)))
    cov6 = np.stack([covs[:, 0, 0], covs[:, 0, 1], covs[:, 0, 2], covs[:, 1, 1], covs[:, 1, 2], covs[:, 2, 2]], 1)[order]

    def ref_frame(cam_q, cam_p, H, W):
        V, K = h.scene._view_and_K(H, W, cam_q, cam_p, h.scene.camera.fov)
        return oracle.render(sc.means[order], sc.opacities[order], colors[order], V, K, W, H, cov6=cov6, sh_degree=-1,
                             group_id=group_of, group_Rt=np.stack(Rt), background=(0, 0, 0), want_rgb8=True)
    return h, ref_frame, len(order)


def test_config2_on_the_reference_6_mask_partition():
    """BASELINE config 2 (pushT scene, ~300k Gaussians, 640x480) on the reference's REAL partition."""
    h, ref_frame, n_reg = _real_partition_handler(41)
    assert n_reg >= 292_247                                            # a Gaussian in two masks is registered twice
    cam_q, cam_p = (0.0, 1.0, 0.0, 0.0), (0.0, 0.0, 3.0)
    frame = h.render(h.scene, [(np.array(cam_q), np.array(cam_p))], [[480, 640]])[0]
    ref = ref_frame(cam_q, cam_p, 480, 640)
    assert ref["n_visible"] > 150_000 and frame.shape == (480, 640, 3)
    assert np.array_equal(frame, ref["rgb8"])
    h.scene.close()


def test_config4_eight_poses_on_the_reference_6_mask_partition():
    """BASELINE config 4: eight Gym-camera poses at 640x480 of the same real partition, one batched call;
    every frame against the oracle."""
    h, ref_frame, _ = _real_partition_handler(42)
    cams = []
    for k in range(8):
        yaw = np.deg2rad(45.0 * k)
        # camera-to-world, OpenCV axes, on a ring of radius 3 looking at the origin
        eye = np.array([3.0 * np.sin(yaw), 0.3, 3.0 * np.cos(yaw)])
        fwd = -eye / np.linalg.norm(eye)
        right = np.cross(fwd, [0.0, 1.0, 0.0]); right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        cams.append((poses.matrix_to_quat_wxyz(np.stack([right, down, fwd], 1)), eye))
    frames = h.render(h.scene, cams, [[480, 640]] * 8)
    assert len(frames) == 8
    for k, (q, p) in enumerate(cams):
        ref = ref_frame(q, p, 480, 640)
        assert ref["n_visible"] > 100_000, k
        assert np.array_equal(frames[k], ref["rgb8"]), k
    h.scene.close()


def test_link_pose_algebra_in_the_library_equals_the_numpy_form():
    """sas_set_link_constants / sas_set_link_poses: the draw message's pose algebra (splat_handler.py:265-288) in C,
    float64, against poses.link_splat_poses + the handles' quaternion round trip -- the float32 pose block that reaches
    the GPU must be the same, message after message; handles read their rows back; other groups keep their poses."""
    from sim_a_splat_amd.scene import SplatScene
    d = _door_b_setup(3000, 61, 7)
    sc = d["sc"]
    from sim_a_splat_amd.handler import SplatHandler
    h = SplatHandler.from_arrays(sc.means, d["covs"], d["colors"], sc.opacities, d["masks"], d["icp"], d["fk"], device=0,
                                 weld_translation=(0.01, -0.02, 0.1))
    assert h._fast
    rng = np.random.default_rng(3)
    h.scene_handle.position = (0.1, 0.2, 0.3)                      # a group the links do not drive
    worst = 0
    for step in range(200):
        msg = _fake_msg(rng, 7)
        if step % 50 == 0:                                         # a half-turn: the trace <= 0 branch of matrix -> quaternion
            msg.quaternion[2] = [0.0, 1.0, 0.0, 0.0]
        h.draw_handler(msg)
        got = h.scene._raster.get_group_poses().reshape(-1, 3, 4)
        q = np.asarray(msg.quaternion, np.float64)
        p = np.asarray(msg.position, np.float64)
        R, t = poses.link_splat_poses(h.scale_factor, h.Ri, h.ti, h._fkR, h._fkt, q, p, h.weld_translation)
        want = np.zeros((8, 3, 4), np.float32)
        want[:7, :, :3] = poses.quats_wxyz_to_matrices(poses.matrices_to_quats_wxyz(R))
        want[:7, :, 3] = t
        want[7, :, :3] = np.eye(3)
        want[7, :, 3] = (0.1, 0.2, 0.3)
        worst = max(worst, int((got != want).sum()))
        assert np.array_equal(h.scene._Rt, got)
    assert worst == 0
    # handles read their rows back
    hq = h.splat_links_handler[3]
    assert np.allclose(poses.quat_wxyz_to_matrix(hq.wxyz), got[3, :, :3], atol=1e-6) and np.allclose(hq.position, got[3, :, 3], atol=1e-7)
    # a frame after the fast path equals the oracle's frame of the NumPy poses (weld translation and moved static group included)
    V, K = h.scene._view_and_K(96, 128, (0.0, 1.0, 0.0, 0.0), (0.0, 0.0, 3.0), h.scene.camera.fov)
    o = d["order"]
    ref = oracle.render(sc.means[o], sc.opacities[o], d["colors"][o], V, K, 128, 96, cov6=d["cov6"][o], sh_degree=-1,
                        group_id=d["group_of"], group_Rt=want.reshape(8, 12), background=(0, 0, 0), want_rgb8=True)
    frame = h.render(h.scene, [((0.0, 1.0, 0.0, 0.0), (0.0, 0.0, 3.0))], [[96, 128]])[0]
    assert ref["n_visible"] > 500 and np.array_equal(frame, ref["rgb8"])
    h.scene.close()
