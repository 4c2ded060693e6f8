"""The dynamic-scene front end of the Gym wrapper on the HIP rasterizer.

``SplatHandler`` (same constructor arguments as sim_a_splat/splat/splat_handler.py:22-38) partitions the
Gaussians into per-link groups plus the static rest (:104-143), turns Drake draw messages into group
poses (:227-314) and renders camera lists (:334-346).  ``CameraRig`` holds the camera dictionary logic
of ``SplatEnvWrapper._configure_cameras/render/_get_obs`` (splat_env_wrapper.py:33-65, :105-159); the
wrapper itself is ``sim_a_splat_amd.env_wrapper.SplatEnvWrapper``.  Nothing here imports viser, pydrake
or gymnasium: messages, poses and the weld frame are duck-typed.
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


def _translation_of(x) -> np.ndarray:
    """Translation of the robot weld frame: a 3-vector, or an object with ``.translation()`` (pydrake
    RigidTransform in the reference, splat_handler.py:229; only identity rotations are supported there)."""
    if x is None:
        return np.zeros(3)
    if hasattr(x, "translation"):
        t = x.translation() if callable(x.translation) else x.translation
        return np.asarray(t, dtype=np.float64).reshape(3)
    return np.asarray(x, dtype=np.float64).reshape(3)


class SplatHandler:
    """``SplatHandler(splat_assets_path, match_object_name, splat_config_name, package_path, package_name,
    urdf_name, task_assets_path=None, task_assets_name=None, sim_robot_weld_frame_transform=..., server=None)``
    -- the reference's constructor (splat_handler.py:22-60) with the same argument meaning:

    * masks, ICP similarity and mask-time joint configuration from ``{splat_assets_path}/masks/{match_object_name}/``;
    * Gaussians from ``{splat_assets_path}/splatfacto/{splat_config_name}`` (``GSplatLoader.from_path``);
    * URDF ``{package_path}/{package_name}urdf/{urdf_name}`` (the reference concatenates exactly so, :52,148)
      for the visual-mesh forward kinematics;
    * ``server``: where the reference takes a viser server, this takes the object that plays ``server.scene`` +
      the client, a ``SplatScene`` (created on ``device`` when None).  Task / robot meshes are not displayed
      (viewer-only, out of scope); ``task_assets_*`` are accepted and kept for the Drake namespaces.

    ``SplatHandler.from_arrays`` builds the same object from arrays already in memory."""

    def __init__(self, splat_assets_path: str, match_object_name: str, splat_config_name: str, package_path: str,
                 package_name: str, urdf_name: str, task_assets_path: Optional[str] = None,
                 task_assets_name: Optional[str] = None, sim_robot_weld_frame_transform=None, server: Optional[SplatScene] = None,
                 *, device=0, bounds=None):
        from pathlib import Path
        from . import io, urdf_fk
        from .covariance import GSplatLoader
        masks_dir = Path(f"{splat_assets_path}/masks/{match_object_name}/").resolve()
        mfile = masks_dir / "link_masks_global_dict.npz"
        masks = io.load_link_masks(mfile if mfile.exists() else masks_dir / "link_masks_global_dict.npy")
        icp = io.load_icp_transformation(masks_dir / "icp_transformation.npy")
        loader = GSplatLoader.from_path(Path(f"{splat_assets_path}/splatfacto/{splat_config_name}").resolve())
        robot_description_dir = package_path + "/" + package_name
        fk = urdf_fk.visual_mesh_fk(urdf_fk.load(Path(robot_description_dir + f"urdf/{urdf_name}")),
                                    io.load_joint_config(masks_dir / "joint_config.npy"))
        keep = aabb_mask(loader.means.cpu().numpy(), bounds)
        arr = lambda t: t.cpu().numpy()[keep]
        masks = {k: np.asarray(v, dtype=bool)[keep] for k, v in masks.items()}
        self._setup(arr(loader.means), arr(loader.covs), arr(loader.colors), arr(loader.opacities), masks, icp, fk,
                    instance_uid=match_object_name, weld=sim_robot_weld_frame_transform, scene=server, device=device)
        self.masks_dir = str(masks_dir)
        self.robot_description_dir = robot_description_dir
        self.rbt_drake_namespace = f"plant::{urdf_name.split('.')[0]}::"                       # :58-60
        self.blk_drake_namespace = f"plant::{task_assets_name.split('.')[0]}::" if task_assets_name else None

    @classmethod
    def from_arrays(cls, means, covs, colors, opacities, link_masks: Dict[str, np.ndarray], icp_transformation: np.ndarray,
                    fk_transforms: Sequence[np.ndarray], instance_uid: str = "robot", robot_num: int = 3,
                    weld_translation=(0.0, 0.0, 0.0), scene: Optional[SplatScene] = None, device=0) -> "SplatHandler":
        """The same handler from arrays: Gaussians [N,...], the per-link boolean masks ``link0..``, the 4x4 ICP
        similarity and one 4x4 forward-kinematics pose per visual mesh at the mask-time joint configuration."""
        self = cls.__new__(cls)
        self._setup(means, covs, colors, opacities, link_masks, icp_transformation, fk_transforms, instance_uid=instance_uid,
                    weld=weld_translation, scene=scene, device=device, robot_num=robot_num)
        return self

    def _setup(self, means, covs, colors, opacities, link_masks, icp_transformation, fk_transforms, *, instance_uid, weld,
               scene, device, robot_num: int = 3) -> None:
        self.scene = scene if scene is not None else SplatScene(device)
        self.server = self.scene                               # the reference's name for it (close(), clients)
        self.instance_uid = instance_uid
        self.rbt_idx, self.blk_idx = robot_num, 2              # splat_handler.py:58
        self.weld_translation = _translation_of(weld)
        self.scale_factor, self.Ri, self.ti = poses.decompose_icp(icp_transformation)
        self.fk = [(np.asarray(T, np.float64)[:3, :3], np.asarray(T, np.float64)[:3, 3]) for T in fk_transforms]
        self._fkR = np.stack([R for R, _ in self.fk]) if self.fk else np.zeros((0, 3, 3))
        self._fkt = np.stack([t for _, t in self.fk]) if self.fk else np.zeros((0, 3))
        means, covs = np.asarray(means, np.float32), np.asarray(covs, np.float32)
        colors, opacities = np.asarray(colors, np.float32), np.asarray(opacities, np.float32).reshape(-1)
        self.means, self.covs, self.colors, self.opacities = means, covs, colors, opacities   # :99-102
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
        # the draw message's pose algebra runs inside the library when the scene offers it (sas_set_link_poses)
        self._k_fast = min(len(self.fk), 7, len(self.splat_links_handler))
        self._fast = hasattr(self.scene, "set_link_poses") and self._k_fast > 0
        if self._fast:
            # kept by the scene under THIS handler's key: a second handler on the same scene has constants of its own
            self.scene.set_link_constants(self.scale_factor, self.Ri, self.ti, self._fkR[:self._k_fast], self._fkt[:self._k_fast],
                                          self.weld_translation, [h.index for h in self.splat_links_handler[:self._k_fast]], owner=id(self))

    @classmethod
    def from_assets(cls, loader, masks_dir, urdf_path, bounds=None, **kw) -> "SplatHandler":
        """Like the path constructor, with the Gaussians already loaded (a ``GSplatLoader``) and the masks
        directory / URDF file given directly; ``bounds`` is the AABB crop of ``_load_saved_splats``."""
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
        return cls.from_arrays(arr(loader.means), arr(loader.covs), arr(loader.colors), arr(loader.opacities), masks, icp, fk, **kw)

    def draw_handler(self, msg) -> None:
        """``msg``: lcmt_viewer_draw-shaped (num_links, robot_num[], position[][3], quaternion[][4] wxyz).  The
        k-th link of the robot (``robot_num == rbt_idx``, message order) drives splat group k, as in the reference
        (:227-314); all links are posed in one batch of small matrix products."""
        rn, rbt = msg.robot_num, self.rbt_idx
        idxs = [idx for idx in range(msg.num_links) if rn[idx] == rbt]
        if len(idxs) > len(self.fk):
            for idx in idxs[len(self.fk):]:
                logging.warning(f"Warning: Received draw command for non-existent Link index {idx}.")
            idxs = idxs[:len(self.fk)]
        k = min(len(idxs), 7, len(self.splat_links_handler))               # :282: at most seven link groups
        if k == 0:
            return
        if self._fast and idxs[k - 1] == k - 1:          # the robot's links lead the message (Drake's order): no gather
            self.scene.set_link_poses(msg.quaternion[:k], msg.position[:k], owner=id(self))
            return
        q = np.asarray([msg.quaternion[i] for i in idxs[:k]], dtype=np.float64)
        p = np.asarray([msg.position[i] for i in idxs[:k]], dtype=np.float64)
        if self._fast:
            self.scene.set_link_poses(q, p, owner=id(self))      # float64 in C, the arithmetic below; handles read their rows back on demand
            return
        R, t = poses.link_splat_poses(self.scale_factor, self.Ri, self.ti, self._fkR[:k], self._fkt[:k], q, p, self.weld_translation)
        wxyz = poses.matrices_to_quats_wxyz(R)
        lock = getattr(self.scene, "lock", None)
        if lock is not None:
            lock.acquire()      # the k assignments are ONE update: a render on another thread sees all or none of it
        try:
            for j in range(k):
                h = self.splat_links_handler[j]
                h.wxyz = wxyz[j]
                h.position = t[j]
        finally:
            if lock is not None:
                lock.release()

    def link_pose_rows(self, msg) -> Tuple[np.ndarray, np.ndarray]:
        """The pose rows ``draw_handler(msg)`` would give this handler's link groups, WITHOUT touching the scene:
        ``(group indices [k], rows [k,12] float32)``.  Vectorised envs share one scene and keep a pose set per env
        (``SplatVecEnv``); the rows are the library's context-free ``sas_link_group_poses`` (float64 in C, the
        arithmetic of splat_handler.py:265-288), the same bits ``sas_set_link_poses`` writes."""
        from . import _capi
        rn, rbt = msg.robot_num, self.rbt_idx
        idxs = [idx for idx in range(msg.num_links) if rn[idx] == rbt][:len(self.fk)]
        k = min(len(idxs), 7, len(self.splat_links_handler))
        groups = np.array([h.index for h in self.splat_links_handler[:k]], dtype=np.int64)
        rows = np.zeros((k, 12), np.float32)
        if k == 0:
            return groups, rows
        c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        q, p = c([msg.quaternion[i] for i in idxs[:k]]), c([msg.position[i] for i in idxs[:k]])
        Ri, ti, Rfk, tfk, weld = c(self.Ri), c(self.ti), c(self._fkR[:k]), c(self._fkt[:k]), c(self.weld_translation)
        rc = _capi.lib().sas_link_group_poses(k, float(self.scale_factor), Ri.ctypes.data, ti.ctypes.data, Rfk.ctypes.data, tfk.ctypes.data,
                                              weld.ctypes.data, q.ctypes.data, p.ctypes.data, rows.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"sas_link_group_poses failed ({rc})")
        return groups, rows

    def get_attached_frame(self, body_name: str, local_frame_pos, msg) -> Tuple[np.ndarray, np.ndarray]:
        """``local_frame_pos``: the camera's ``local_frame`` (SE3-like, as the reference passes it, :316-319) or
        just its translation; only the translation is used -- added in world axes, the reference's behaviour."""
        if type(local_frame_pos) is poses.SE3:
            local_xyz = local_frame_pos.wxyz_xyz[4:]
        elif hasattr(local_frame_pos, "translation") or (isinstance(local_frame_pos, (tuple, list)) and len(local_frame_pos) == 2):
            local_xyz = poses.pose_wxyz_xyz(local_frame_pos)[1]
        else:
            local_xyz = np.asarray(local_frame_pos, dtype=np.float64).reshape(3)
        idx = msg.link_name.index("plant::" + body_name) if isinstance(msg.link_name, list) else list(msg.link_name).index("plant::" + body_name)
        if self._fast and hasattr(self.scene, "attached_frame"):
            return self.scene.attached_frame(msg.quaternion[idx], msg.position[idx], local_xyz, owner=id(self))
        R, t = poses.attached_frame(self.scale_factor, self.Ri, self.ti, msg.quaternion[idx], msg.position[idx], local_xyz)
        return poses.matrix_to_quat_wxyz(R), t

    def render(self, chs, cam_poses, render_size) -> List[np.ndarray]:
        """``render(chs, cam_poses: List[SE3], render_size)`` of the reference (:334-346): ``chs`` is the client
        handle (here the ``SplatScene``: ``get_render`` + a ``camera``), ``cam_poses`` SE3-like objects or
        ``(wxyz, xyz)`` pairs (None: the client's own camera), ``render_size[i] = [H, W]``.  Returns uint8
        [H,W,3] frames in camera order.  Cameras of equal size go to the GPU as one batch when the client
        offers ``get_renders`` (the reference renders them one by one)."""
        if cam_poses is None:
            cam_poses = [(chs.camera.wxyz, chs.camera.position)]
        cam = [poses.pose_wxyz_xyz(p) for p in cam_poses]
        n = len(cam)
        s0 = render_size[0] if n else None
        if n and hasattr(chs, "get_renders") and all(s[0] == s0[0] and s[1] == s0[1] for s in render_size[1:n]):
            return list(chs.get_renders(int(s0[0]), int(s0[1]), cam))      # the usual rig: every camera the same size, one batch
        sizes = [(int(s[0]), int(s[1])) for s in render_size]
        out: List[Optional[np.ndarray]] = [None] * len(cam)
        for hw in dict.fromkeys(sizes[:len(cam)]):
            idx = [i for i, s in enumerate(sizes[:len(cam)]) if s == hw]
            if hasattr(chs, "get_renders"):
                imgs = chs.get_renders(hw[0], hw[1], [cam[i] for i in idx])
            else:
                imgs = [chs.get_render(height=hw[0], width=hw[1], wxyz=cam[i][0], position=cam[i][1]) for i in idx]
            for j, i in enumerate(idx):
                out[i] = np.asarray(imgs[j])
        return out


class CameraRig:
    """Camera dictionary ``{id: {link_name, local_frame, type, render_size}}`` of the reference
    (examples/demo_pusht_splat.py:54-78; ``local_frame`` SE3-like or ``(wxyz, xyz)``), resolved to render
    poses.  Order: moving cameras, then viewport + static (splat_env_wrapper.py:33-55,146)."""

    def __init__(self, camera_setup_info: Dict):
        self.camera_setup_info = camera_setup_info
        self.moving = {k: v for k, v in camera_setup_info.items() if v.get("type") == "moving"}
        self.viewport = {k: v for k, v in camera_setup_info.items() if v.get("type") == "viewport"}
        fixed = {k: v for k, v in camera_setup_info.items() if v.get("type") in ("viewport", "static")}
        self.fixed_cam_poses = [poses.pose_wxyz_xyz(v["local_frame"]) for v in fixed.values()]
        self.render_cam_keys = list(self.moving.keys()) + list(fixed.keys())

    def poses(self, handler: SplatHandler, msg) -> List[Tuple[np.ndarray, np.ndarray]]:
        moving = [handler.get_attached_frame(v["link_name"], v["local_frame"], msg) for v in self.moving.values()]
        return moving + self.fixed_cam_poses

    def sizes(self) -> List[Sequence[int]]:
        return [self.camera_setup_info[k]["render_size"] for k in self.render_cam_keys]

    def get_obs(self, handler: SplatHandler, msg) -> Dict[str, np.ndarray]:
        """``camera_i`` -> uint8 [3,H,W]  (splat_env_wrapper.py:132-138).  Uses the CURRENT message:
        the reference reads the one stored at reset (its moving cameras lag)."""
        imgs = handler.render(handler.scene, self.poses(handler, msg), self.sizes())
        return {f"camera_{i}": np.moveaxis(img, -1, 0) for i, img in enumerate(imgs)}
