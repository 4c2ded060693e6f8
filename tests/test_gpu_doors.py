"""GPU: the two reference doors (host-side mirror -> C ABI -> HIP) against the oracle."""
import types

import numpy as np
import pytest
import torch

import oracle
from oracle import ref_math
from sim_a_splat_amd import poses
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
    gs = GaussianSplat(model, pc)
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
    u = torch.arange(200, device=d.device)[None, :].expand(150, 200)
    assert torch.allclose(pts[..., 0], (u - K[0, 2]) * d / K[0, 0])


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
    h = SplatHandler(sc.means, covs, colors, sc.opacities, masks, icp, fk, device=0)
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
