"""``SplatVecEnv``: E Gym environments rendered as ONE batch per step, sharded over the ranks of a node.

The reference steps one env and renders its cameras one by one through a browser
(sim_a_splat/env/splat/splat_env_wrapper.py:121-159); a rollout collector runs E such envs.  Their
camera views are independent renders of one read-only scene that differ only in the link poses of
their env -- north_star's "independent camera views from the Gym env's vectorised rollouts shard
one-view-per-GPU ... with a RCCL gather of finished frames".  This module is that front end as a
library object (round 3 had it only as a loop inside bench.py):

* every rank holds ONE ``SplatHandler`` / ``SplatScene`` (the scene replicated per GPU) and the envs
  ``e`` with ``e % world == rank`` (``distributed.shard_views``): physics, pose algebra and rendering
  of an env happen on its rank;
* per step and env: ``env.step(action)`` -> draw message -> the handler's link-pose algebra
  (splat_handler.py:265-288) as a row block ``[G,12]`` of the env's own (``SplatHandler.link_pose_rows``:
  the library's context-free ``sas_link_group_poses``, so E envs never fight over the scene's one pose
  block) -> moving-camera poses (:316-332) -> all E_local x C cameras of the rank in ONE
  ``sas_render_batch_host_posed`` call, view v rendered with the pose set of its env;
* the uint8 frames (``camera_i`` observations, splat_env_wrapper.py:135-137) of all ranks are gathered to
  rank 0 through ``distributed.StepPipeline`` (backend ``nccl`` = RCCL on a multi-GPU node, ``gloo`` in
  the CPU tests); the small non-image observations, rewards and flags travel as objects beside them.
  On the ``nccl`` backend the frames stay DEVICE-RESIDENT until they are on rank 0: rendered into a device
  buffer (``sas_render_batch_posed``), gathered over xGMI as they are, copied to pinned host memory once, on the
  root (round 4 rendered to the host and uploaded the frames again as the gather's payload: host -> device ->
  xGMI -> device -> host; ``h2d_frame_copies`` counts such uploads and stays 0 now).  A single rank, the other
  ranks' own frames and ``gloo`` keep the host path (the tile kernel stores the frames to pinned memory itself).

SPMD calling convention: every rank constructs the same ``SplatVecEnv`` and calls ``reset`` / ``step``
with the actions of ALL envs; a rank applies those of its own envs.  Rank 0 gets the observations of
every env; the other ranks get ``None`` for the envs they do not own.

``step`` is synchronous (a policy needs step t's images to act at t + 1).  ``step_async`` /
``collect`` keep one step in flight: the frames of step t are rendered and gathered while the caller
steps the physics of t + 1 (rollouts whose actions do not depend on the images: teleoperation logs,
scripted policies, dataset rendering).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import distributed as D
from . import poses
from .handler import CameraRig, SplatHandler


class SplatVecEnv:
    def __init__(self, envs: Sequence, splat_handler: SplatHandler, camera_setup_info: Dict, *, rank: int = 0, world: int = 1,
                 fov: Optional[float] = None, collective: Optional[bool] = None):
        """``envs``: the E inner envs, duck-typed like ``SplatEnvWrapper``'s (``step``, and on ``unwrapped``:
        ``reset(seed=, reset_to_state=)``, ``_generate_draw_msg()``, ``_get_obs()``); a rank only touches
        ``envs[e]`` for its own ``e`` (the others may be ``None``).  ``splat_handler``: this rank's handler (its
        scene holds the replicated Gaussians).  ``camera_setup_info``: the reference's camera dictionary
        (examples/demo_pusht_splat.py:54-78), the same for every env; all cameras must share one ``render_size``.
        ``collective``: as in ``distributed.FrameGather`` (True forces the gather through the backend for a world of one)."""
        self.envs = list(envs)
        self.E = len(self.envs)
        self.rank, self.world = int(rank), int(world)
        self.mine = D.shard_views(self.E, self.rank, self.world)
        self.handler, self.scene = splat_handler, splat_handler.scene
        self.rig = CameraRig(camera_setup_info)
        sizes = {tuple(int(x) for x in s) for s in self.rig.sizes()}
        if len(sizes) != 1:
            raise ValueError("SplatVecEnv renders the cameras of all envs as one batch: they must share one render_size")
        (self.H, self.W), self.C = next(iter(sizes)), len(self.rig.render_cam_keys)
        self.fov = fov
        self.per_rank = (self.E + self.world - 1) // self.world          # envs per rank, padded: one gather shape for all
        self._collective = (self.world > 1) if collective is None else bool(collective)
        if self._collective and not dist.is_initialized():
            raise RuntimeError("SplatVecEnv(collective=True) gathers through torch.distributed: initialise the process group first "
                               "(sim_a_splat_amd.distributed.init_from_env)")
        self._msgs: List = [None] * self.E
        self._pin: Optional[bool] = None
        # RCCL gathers device tensors: there the frames are rendered into DEVICE buffers and travel as they are
        self._device_payload = self._collective and dist.get_backend() == "nccl"
        self._device = getattr(getattr(self.scene, "_raster", None), "device", None) if self._device_payload else None   # the scene's GPU
        if self._device_payload and self._device is None:
            self._device = torch.device("cuda", torch.cuda.current_device())
        self.h2d_frame_copies = 0                                         # uploads of finished frames (the nccl arm must not need any)
        self.d2h_frame_copies = 0                                         # downloads of gathered / own device frames (one per step and rank that consumes them)
        self._bufs = [None, None, None]                                   # a ring of three steps: every step brings a buffer of its own (_submit)
        self._pending: Dict[int, Tuple] = {}                              # step index -> what travels beside its frames
        self._gathered: Dict[int, Optional[List[torch.Tensor]]] = {}
        self._step_idx = 0
        self._done_steps = 0
        self._pipe = D.StepPipeline(self.world, self.rank, self._bufs, self._submit, lambda: self._done_steps, lambda: None,
                                    payload=self._payload, on_gathered=self._on_gathered, collective=self._collective)
        self._pipe.begin()

    # -- buffers -----------------------------------------------------------------------------------------------
    def _new_buffer(self) -> torch.Tensor:
        """A frame buffer of this rank's envs: pinned host memory (the tile kernel stores the frames into it), or device
        memory on the nccl arm.  torch's caching allocators hand recently freed blocks back, so a buffer per step costs no
        allocation in steady state.  Rows beyond this rank's envs (E % world != 0, a rank without envs) are padding of the
        gather's common shape: zeroed, not whatever the block held."""
        shape = (self.per_rank * self.C, self.H, self.W, 3)
        n_mine = len(self.mine) * self.C
        if self._device_payload:
            buf = torch.empty(shape, dtype=torch.uint8, device=self._device)
            if n_mine < shape[0]:
                buf[n_mine:].zero_()
            return buf
        buf = self._new_host_buffer(shape)
        if n_mine < shape[0]:
            buf[n_mine:].zero_()
        return buf

    def _new_host_buffer(self, shape) -> torch.Tensor:
        if self._pin is None:                                             # asked once: torch.cuda.is_available() costs 0.4 ms a call
            self._pin = bool(torch.cuda.is_available())
        try:
            return torch.empty(shape, dtype=torch.uint8, pin_memory=self._pin)
        except RuntimeError:                                              # no HIP runtime (CPU tests)
            self._pin = False
            return torch.empty(shape, dtype=torch.uint8)

    def _payload(self, buf: torch.Tensor) -> torch.Tensor:
        # what the gather moves IS the buffer the frames were rendered into: device memory under RCCL, host memory under gloo
        if self._device_payload and not buf.is_cuda:                      # (cannot happen with _new_buffer's buffers; counted if it ever does)
            self.h2d_frame_copies += 1
            return buf.cuda(non_blocking=True)
        return buf

    def _to_host(self, frames: torch.Tensor) -> torch.Tensor:
        """Finished frames for the caller (np.uint8 observations live on the host): device frames come down once, into pinned memory."""
        if not frames.is_cuda:
            return frames
        host = self._new_host_buffer(tuple(frames.shape))
        host.copy_(frames, non_blocking=True)
        torch.cuda.current_stream(frames.device).synchronize()
        self.d2h_frame_copies += 1
        return host

    # -- the per-step work of a rank ------------------------------------------------------------------------------
    def _pose_set_of(self, msg) -> np.ndarray:
        """[G,12] float32: every group's pose for this env's draw message (static groups keep the scene's row)."""
        rows = self.scene.group_pose_rows()
        idx, link_rows = self.handler.link_pose_rows(msg)
        rows[idx] = link_rows
        return rows

    def _submit(self, i: int, buf: torch.Tensor) -> None:
        """Render the rank's envs for step ``i`` (blocking: the frames are on the host when it returns) into a buffer of the
        step's own -- it takes the ring's place for this step, and the observations handed out are views of it, so nothing
        is copied and nothing a caller still holds is ever overwritten."""
        buf = self._new_buffer()
        self._pipe.set_buffer(i, buf)
        msgs = [self._msgs[e] for e in self.mine]
        if msgs:
            pose_sets = np.stack([self._pose_set_of(m) for m in msgs])                     # [E_local, G, 12]
            cams = [self.rig.poses(self.handler, m) for m in msgs]                         # E_local x C (wxyz, xyz)
            flat = [c for env_c in cams for c in env_c]
            idx = [k for k in range(len(msgs)) for _ in range(self.C)]
            n = len(flat)
            if self._device_payload:
                self.scene.get_renders_posed(self.H, self.W, flat, pose_sets, idx, fov=self.fov, device_out=buf[:n])
            else:
                self.scene.get_renders_posed(self.H, self.W, flat, pose_sets, idx, fov=self.fov, out=buf[:n])
        self._done_steps += 1

    def _on_gathered(self, step: int, got) -> None:
        # (the gather's receive buffers are reused by the next step: the observations are views of copies.)  Device frames --
        # the RCCL arm -- come down here, on the root, once per step: all ranks' blocks through one pinned buffer
        if got is None:
            self._gathered[step] = None
        elif got[0].is_cuda:
            host = self._to_host(torch.stack(list(got)) if len(got) > 1 else got[0].unsqueeze(0))
            self._gathered[step] = [host[r] for r in range(len(got))]
        else:
            self._gathered[step] = [g.clone() for g in got]

    # -- Gym surface ---------------------------------------------------------------------------------------------
    def _u(self, e: int):
        env = self.envs[e]
        return getattr(env, "unwrapped", env)

    def reset(self, seed: Optional[int] = None, reset_to_state=None) -> List[Optional[Dict[str, np.ndarray]]]:
        """Reset every env of this rank (``seed + e``; ``reset_to_state`` a list per env or one value for all) and
        return the first observations like ``step``."""
        states = reset_to_state if isinstance(reset_to_state, (list, tuple)) and len(reset_to_state) == self.E else [reset_to_state] * self.E
        inner = {}
        for e in self.mine:
            u = self._u(e)
            u.reset(seed=None if seed is None else seed + e, reset_to_state=states[e])
            self._msgs[e] = u._generate_draw_msg()
            inner[e] = u._get_obs()
        return self._observe(inner, None)[0]

    def step(self, actions: Sequence, noobs: bool = False):
        """``actions[e]`` for every env (each rank uses its own).  Returns ``(obs, reward, terminated, truncated, info)``,
        each a list over ALL envs on rank 0 (entries of foreign envs are ``None`` elsewhere); ``obs[e]`` = the inner
        env's observation + ``camera_i`` uint8 [3,H,W] (splat_env_wrapper.py:132-138).  ``noobs``: step the physics only."""
        inner, extra = self._step_physics(actions)
        if noobs:
            return [None] * self.E, *self._exchange_extra(extra)
        obs, ex = self._observe(inner, extra)
        return (obs, *ex)

    def step_async(self, actions: Sequence) -> int:
        """Step the physics and ENQUEUE the rendering + gather of this step; returns its ticket for ``collect``.  At most
        ``len(buffers) - 1`` steps may be outstanding."""
        if len(self._pending) >= len(self._bufs) - 1:
            raise RuntimeError(f"at most {len(self._bufs) - 1} steps may be outstanding: collect() one first")
        inner, extra = self._step_physics(actions)
        t = self._step_idx
        self._pending[t] = (inner, extra)
        self._pipe.step()             # renders this rank's envs and starts the gather; returns while the frames travel
        self._step_idx += 1
        return t

    def collect(self, ticket: int):
        """The result of ``step_async(...)`` with that ticket, as ``step`` returns it."""
        if ticket not in self._pending:
            raise KeyError(f"no outstanding step {ticket}")
        if self._collective and ticket not in self._gathered:
            self._pipe.drain()
        inner, extra = self._pending.pop(ticket)
        obs = self._assemble(ticket, inner)
        return (obs, *self._exchange_extra(extra))

    def close(self, close_scene: bool = True) -> None:
        """Drain the pipeline and close this rank's envs.  The scene belongs to the caller's handler; it is closed with the
        vectorised env unless ``close_scene=False`` (a caller that goes on using the handler)."""
        self._pipe.drain()
        for e in self.mine:
            c = getattr(self._u(e), "close", None)
            if c:
                c()
        if close_scene:
            self.scene.close()

    # -- internals -----------------------------------------------------------------------------------------------
    def _step_physics(self, actions):
        inner, extra = {}, {}
        for e in self.mine:
            o, rew, term, trunc, info = self.envs[e].step(actions[e])
            u = self._u(e)
            self._msgs[e] = u._generate_draw_msg()          # the CURRENT message poses links and moving cameras
            inner[e] = u._get_obs()
            extra[e] = (rew, term, trunc, info)
        return inner, extra

    def _observe(self, inner, extra):
        t = self._step_idx
        self._pending[t] = (inner, extra)
        self._pipe.step()
        self._step_idx += 1
        self._pipe.drain()
        inner, extra = self._pending.pop(t)
        return self._assemble(t, inner), (self._exchange_extra(extra) if extra is not None else None)

    def _assemble(self, t: int, inner) -> List[Optional[Dict[str, np.ndarray]]]:
        """camera_i per env from the gathered frames (rank 0: all envs) or from this rank's own buffer."""
        frames_of = {}
        got = self._gathered.pop(t, None) if self._collective else None
        if got is not None:                                                # rank 0: rank r's block holds its envs in order
            for r in range(self.world):
                for k, e in enumerate(D.shard_views(self.E, r, self.world)):
                    frames_of[e] = got[r][k * self.C:(k + 1) * self.C]
        else:                                                              # a single rank, or a rank that is not the root: its own envs
            buf = self._to_host(self._pipe.buffer_of(t))
            for k, e in enumerate(self.mine):
                frames_of[e] = buf[k * self.C:(k + 1) * self.C]
        all_inner = self._exchange_objects(inner)
        obs: List[Optional[Dict[str, np.ndarray]]] = [None] * self.E
        for e, fr in frames_of.items():
            o = dict(all_inner.get(e) or {})
            a = fr.numpy() if isinstance(fr, torch.Tensor) else np.asarray(fr)
            for i in range(self.C):
                o[f"camera_{i}"] = a[i].transpose(2, 0, 1)      # np.moveaxis(img, -1, 0), as the reference does: a view
            obs[e] = o
        return obs

    def _exchange_objects(self, mine: Dict[int, object]) -> Dict[int, object]:
        """Small per-env Python objects of every rank, merged on rank 0 (this rank's own elsewhere)."""
        if not self._collective or not dist.is_initialized():
            return dict(mine)
        box = [None] * self.world if self.rank == 0 else None
        dist.gather_object(mine, box, dst=0)
        if self.rank != 0:
            return dict(mine)
        out: Dict[int, object] = {}
        for d in box:
            out.update(d)
        return out

    def _exchange_extra(self, extra):
        """(reward, terminated, truncated, info) lists over all envs (rank 0) / this rank's envs."""
        merged = self._exchange_objects(extra)
        cols = [[None] * self.E for _ in range(4)]
        for e, tup in merged.items():
            for j in range(4):
                cols[j][e] = tup[j]
        return tuple(cols)
