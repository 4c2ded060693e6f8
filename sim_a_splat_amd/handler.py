"""The dynamic-scene front end of the Gym wrapper on the HIP rasterizer.

``SplatHandler`` partitions the Gaussians into per-link groups plus the static rest
(sim_a_splat/splat/splat_handler.py:104-143), turns Drake draw messages into group poses
(:227-314) and renders camera lists (:334-346).  ``CameraRig`` holds the camera dictionary logic
of ``SplatEnvWrapper._configure_cameras/render/_get_obs`` (splat_env_wrapper.py:33-65, :105-159).
Neither imports viser, pydrake or gymnasium: messages and poses are duck-typed / plain arrays.
"""
from __future__ import annotations

import logging
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import poses
from .scene import SplatScene


def aabb_mask(means, bounds) -> np.ndarray:
    """Axis-aligned bounding-box mask of ``_load_saved_splats`` (splat_handler.py:91-97): ``bounds`` is
    [3,2] (min, max per axis) or None for "keep everything" (the reference passes None)."""
    means = np.asarray(means)
    if bounds is None:
        return np.ones(means.shape[0], dtype=bool)
    b = np.asarray(bounds, dtype=means.dtype)
    return np.all((means - b[:, 0] >= 0) & (b[:, 1] - means >= 0), axis=-1)


class SplatHandler:
    def __init__(self, means, covs, colors, opacities, link_masks: Dict[str, np.ndarray], icp_transformation: np.ndarray,
                 fk_transforms: Sequence[np.ndarray], instance_uid: str = "robot", robot_num: int = 3,
                 weld_translation=(0.0, 0.0, 0.0), scene: Optional[SplatScene] = None, device=0):
        self.scene = scene if scene is not None else SplatScene(device)
        self.instance_uid = instance_uid
        self.rbt_idx = robot_num                               # splat_handler.py:58
        self.weld_translation = np.asarray(weld_translation, dtype=np.float64)
        self.scale_factor, self.Ri, self.ti = poses.decompose_icp(icp_transformation)
        self.fk = [(np.asarray(T, np.float64)[:3, :3], np.asarray(T, np.float64)[:3, 3]) for T in fk_transforms]
        means, covs = np.asarray(means, np.float32), np.asarray(covs, np.float32)
        colors, opacities = np.asarray(colors, np.float32), np.asarray(opacities, np.float32).reshape(-1)
        n = means.shape[0]
        self.robot_splat_idxs = np.zeros(n, dtype=bool)
        self.splat_links_handler = []
        for ii in range(len(link_masks)):                      # link0..linkK-1 in order (:124-143)
            idxs = np.asarray(link_masks[f"link{ii}"], dtype=bool)
            self.splat_links_handler.append(self.scene.add_gaussian_splats(
                f"{instance_uid}/splat_robot/link{ii}", means[idxs], covs[idxs], colors[idxs], opacities[idxs]))
            self.robot_splat_idxs |= idxs
        rest = ~self.robot_splat_idxs                          # "/scene_ohne_robot" (:112-119)
        self.scene_handle = self.scene.add_gaussian_splats("/scene_ohne_robot", means[rest], covs[rest], colors[rest],
                                                           opacities[rest])

    @classmethod
    def from_assets(cls, loader, masks_dir, urdf_path, bounds=None, **kw) -> "SplatHandler":
        """The constructor flow of the reference (splat_handler.py:44-55): Gaussians from a
        ``GSplatLoader`` (optionally AABB-masked), ``link_masks_global_dict`` (.npy or the pickle-free
        .npz), ``icp_transformation.npy`` and ``joint_config.npy`` from ``masks_dir``, and the URDF's
        visual-mesh forward kinematics at that joint configuration."""
        from pathlib import Path
        from . import io, urdf_fk
        d = Path(masks_dir)
        mfile = d / "link_masks_global_dict.npz"
        masks = io.load_link_masks(mfile if mfile.exists() else d / "link_masks_global_dict.npy")
        icp = io.load_icp_transformation(d / "icp_transformation.npy")
        fk = urdf_fk.visual_mesh_fk(urdf_fk.load(urdf_path), io.load_joint_config(d / "joint_config.npy"))
        keep = aabb_mask(loader.means.cpu().numpy(), bounds)
        arr = lambda t: t.cpu().numpy()[keep]
        masks = {k: np.asarray(v, dtype=bool)[keep] for k, v in masks.items()}
        return cls(arr(loader.means), arr(loader.covs), arr(loader.colors), arr(loader.opacities), masks, icp, fk, **kw)

    def draw_handler(self, msg) -> None:
        """``msg``: lcmt_viewer_draw-shaped (num_links, robot_num[], position[][3], quaternion[][4] wxyz)."""
        local_idx = 0
        for idx in range(msg.num_links):
            if msg.robot_num[idx] != self.rbt_idx:
                continue
            try:
                Rfk, tfk = self.fk[local_idx]
                R, t = poses.link_splat_pose(self.scale_factor, self.Ri, self.ti, Rfk, tfk, msg.quaternion[idx],
                                             msg.position[idx], self.weld_translation)
                if local_idx < 7 and local_idx < len(self.splat_links_handler):   # :282
                    h = self.splat_links_handler[local_idx]
                    h.wxyz = poses.matrix_to_quat_wxyz(R)
                    h.position = t
                local_idx += 1
            except IndexError:
                logging.warning(f"Warning: Received draw command for non-existent Link index {idx}.")

    def get_attached_frame(self, body_name: str, local_xyz, msg) -> Tuple[np.ndarray, np.ndarray]:
        idx = list(msg.link_name).index("plant::" + body_name)
        R, t = poses.attached_frame(self.scale_factor, self.Ri, self.ti, msg.quaternion[idx], msg.position[idx], local_xyz)
        return poses.matrix_to_quat_wxyz(R), t

    def render(self, cam_poses: Sequence[Tuple[np.ndarray, np.ndarray]], render_size: Sequence[Sequence[int]]) -> List[np.ndarray]:
        """``cam_poses``: (wxyz, position) per camera; ``render_size``: [H, W] per camera.  Cameras of
        equal size go to the GPU as one batch (the reference renders them one by one)."""
        sizes = [(int(s[0]), int(s[1])) for s in render_size]
        out: List[Optional[np.ndarray]] = [None] * len(cam_poses)
        for hw in dict.fromkeys(sizes):
            idx = [i for i, s in enumerate(sizes) if s == hw]
            imgs = self.scene.get_renders(hw[0], hw[1], [cam_poses[i] for i in idx])
            for j, i in enumerate(idx):
                out[i] = imgs[j]
        return out


class CameraRig:
    """Camera dictionary ``{id: {link_name, local_frame, type, render_size}}`` of the reference,
    with ``local_frame`` given as (wxyz, xyz).  Order: moving cameras, then viewport + static."""

    def __init__(self, camera_setup_info: Dict):
        self.camera_setup_info = camera_setup_info
        self.moving = {k: v for k, v in camera_setup_info.items() if v.get("type") == "moving"}
        fixed = {k: v for k, v in camera_setup_info.items() if v.get("type") in ("viewport", "static")}
        self.fixed_cam_poses = [tuple(np.asarray(p, dtype=np.float64) for p in v["local_frame"]) for v in fixed.values()]
        self.render_cam_keys = list(self.moving.keys()) + list(fixed.keys())

    def poses(self, handler: SplatHandler, msg) -> List[Tuple[np.ndarray, np.ndarray]]:
        moving = [handler.get_attached_frame(v["link_name"], v["local_frame"][1], msg) for v in self.moving.values()]
        return moving + self.fixed_cam_poses

    def sizes(self) -> List[Sequence[int]]:
        return [self.camera_setup_info[k]["render_size"] for k in self.render_cam_keys]

    def get_obs(self, handler: SplatHandler, msg) -> Dict[str, np.ndarray]:
        """``camera_i`` -> uint8 [3,H,W]  (splat_env_wrapper.py:132-138).  Uses the CURRENT message:
        the reference reads the one stored at reset (its moving cameras lag)."""
        imgs = handler.render(self.poses(handler, msg), self.sizes())
        return {f"camera_{i}": np.moveaxis(img, -1, 0) for i, img in enumerate(imgs)}
