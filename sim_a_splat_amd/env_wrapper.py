"""``SplatEnvWrapper``: the Gym wrapper of the reference on the HIP rasterizer
(sim_a_splat/env/splat/splat_env_wrapper.py:14-163), same constructor, methods and argument meaning:

    env = SplatEnvWrapper(inner_env, splat_assets_path, match_object_name, splat_config_name,
                          task_assets_path=None, task_assets_name=None)
    env._configure_cameras(camera_setup_info)      # {id: {link_name, local_frame: SE3, type, render_size}}
    obs = env.reset(reset_to_state=...)
    obs, reward, terminated, truncated, info = env.step(action, noobs=False)
    frames = env.render()                           # list of uint8 [H,W,3], moving cameras first
    env.close()

What differs, on purpose:

* No gymnasium / pydrake / viser import.  The inner env is duck-typed: ``step``, ``render`` and -- on
  ``env.unwrapped`` (or the env itself) -- ``reset(seed=, reset_to_state=)``, ``_generate_draw_msg()``,
  ``_get_obs()``, ``visualize_robot_flag``, ``package_path``, ``package_name``, ``urdf_name``,
  ``weld_frame_transform``, ``close()``: exactly the members the reference's wrapper touches.
* There is no browser client to wait for (:87-95): ``self.ch``, the "client handle" the reference renders
  through, is the ``SplatScene`` of the handler; frames come from ``sas_render_batch`` instead of a websocket
  round trip + WebGL draw + JPEG per camera.  All cameras of a step that share a size are one batched call.
* Moving cameras follow the CURRENT draw message.  The reference stores the message only in ``reset`` (:100)
  and reuses it in every later ``render`` (:146), so its eye-in-hand camera stays at the reset pose
  (SURVEY.md 3.1); here ``step`` stores the message it has just generated.
"""
from __future__ import annotations

import logging
from typing import Dict, List, Optional

import numpy as np

from . import poses
from .handler import CameraRig, SplatHandler


class SplatEnvWrapper:
    def __init__(self, env, splat_assets_path: Optional[str] = None, match_object_name: Optional[str] = None,
                 splat_config_name: Optional[str] = None, task_assets_path: Optional[str] = None,
                 task_assets_name: Optional[str] = None, *, splat_handler: Optional[SplatHandler] = None, device=0):
        """``splat_handler`` (keyword-only, not in the reference): a handler built elsewhere -- from arrays, or
        shared between vectorised envs -- instead of loading the assets named by the path arguments."""
        self.env = env
        self._device = device
        self.draw_msg = None
        self._rig: Optional[CameraRig] = None
        if splat_handler is not None:
            self.splat_handler = splat_handler
            self.ch = splat_handler.scene
        else:
            self.ch = self._setup_splats(splat_assets_path=splat_assets_path, match_object_name=match_object_name,
                                         splat_config_name=splat_config_name, task_assets_path=task_assets_path,
                                         task_assets_name=task_assets_name)

    # gym.Wrapper surface the reference relies on
    @property
    def unwrapped(self):
        return getattr(self.env, "unwrapped", self.env)

    def __getattr__(self, name):           # attribute pass-through to the wrapped env, as gym.Wrapper does
        if name in ("env", "splat_handler", "ch"):
            raise AttributeError(name)
        return getattr(self.env, name)

    # -- splat_env_wrapper.py:33-65 ---------------------------------------------------------------------
    def _configure_cameras(self, camera_setup_info: dict):
        rig = CameraRig(camera_setup_info)
        self.moving_cameras_info = rig.moving
        self.fixed_cam_poses = [poses.SE3(np.concatenate(p)) for p in rig.fixed_cam_poses]
        self.render_cam_keys = rig.render_cam_keys
        if not rig.viewport:
            raise KeyError("camera_setup_info needs one camera of type 'viewport' (the reference indexes the first one)")
        wxyz, xyz = poses.pose_wxyz_xyz(next(iter(rig.viewport.values()))["local_frame"])
        self.ch.camera.position = xyz
        self.ch.camera.wxyz = wxyz
        self.camera_setup_info = camera_setup_info
        self._rig = rig

    # -- :67-96 --------------------------------------------------------------------------------------------
    def _setup_splats(self, splat_assets_path, match_object_name, splat_config_name, task_assets_path, task_assets_name,
                      wait_steps=50):
        u = self.unwrapped
        self.splat_handler = SplatHandler(splat_assets_path, match_object_name, splat_config_name, u.package_path,
                                          u.package_name, u.urdf_name, task_assets_path=task_assets_path,
                                          task_assets_name=task_assets_name,
                                          sim_robot_weld_frame_transform=getattr(u, "weld_frame_transform", None),
                                          device=self._device)
        logging.info("splat scene ready: %d Gaussians in %d groups", self.splat_handler.means.shape[0],
                     len(self.splat_handler.splat_links_handler) + 1)
        return self.splat_handler.scene     # the renderer is in-process: no client to wait for (wait_steps unused)

    # -- :98-104 -------------------------------------------------------------------------------------------
    def reset(self, seed: Optional[int] = None, reset_to_state=None):
        self.unwrapped.reset(seed=seed, reset_to_state=reset_to_state)
        self.draw_msg = self.unwrapped._generate_draw_msg()
        self.splat_handler.draw_handler(self.draw_msg)
        if getattr(self.unwrapped, "visualize_robot_flag", False):
            self.env.render()
        return self.unwrapped._get_obs()

    # -- :105-119 ------------------------------------------------------------------------------------------
    def get_moving_camera_poses(self, msg) -> List[poses.SE3]:
        moving_camera_poses = []
        try:
            for info in self.moving_cameras_info.values():
                wxyz, xyz = self.splat_handler.get_attached_frame(info["link_name"], info["local_frame"], msg)
                v = np.empty(7)
                v[:4], v[4:] = wxyz, xyz
                moving_camera_poses.append(poses.SE3(wxyz_xyz=v))
        except AttributeError as e:
            logging.error(f"Error getting moving camera poses: {e}. Have you configured the cameras with _configure_cameras?")
        return moving_camera_poses

    # -- :121-130 ------------------------------------------------------------------------------------------
    def step(self, action, noobs=False):
        obs_in, reward, terminated, truncated, info_in = self.env.step(action)
        self.draw_msg = self.unwrapped._generate_draw_msg()     # kept: moving cameras follow the current step
        self.splat_handler.draw_handler(self.draw_msg)
        observation = None
        if not noobs:
            if getattr(self.unwrapped, "visualize_robot_flag", False):
                self.env.render()
            observation = self._get_obs()
        return observation, reward, terminated, truncated, info_in

    # -- :132-138 ------------------------------------------------------------------------------------------
    def _get_obs(self) -> Dict[str, np.ndarray]:
        obs = self.unwrapped._get_obs()
        img_out = self.render()
        for ii in range(len(img_out)):
            img_out[ii] = img_out[ii].transpose(2, 0, 1)       # np.moveaxis(img, -1, 0): the same view
        obs.update({f"camera_{ii}": img_out[ii] for ii in range(len(img_out))})
        return obs

    # -- :140-159 ------------------------------------------------------------------------------------------
    def render(self, mode="rgb_array") -> List[np.ndarray]:
        self.env.render()
        if self._rig is None:
            raise AttributeError("cameras are not configured: call _configure_cameras(camera_setup_info) first")
        render_cam_poses = self.get_moving_camera_poses(self.draw_msg) + self.fixed_cam_poses
        sizes = [self.camera_setup_info[k]["render_size"] for k in self.render_cam_keys]
        return self.splat_handler.render(self.ch, render_cam_poses, sizes)

    # -- :161-163 ------------------------------------------------------------------------------------------
    def close(self):
        self.splat_handler.scene.close()
        self.unwrapped.close()
