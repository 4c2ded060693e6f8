"""One process per GPU: shard independent camera views over ranks, gather finished frames.

The reference renders its cameras one after the other in a Python loop
(sim_a_splat/env/splat/splat_env_wrapper.py:147-158) and has no multi-GPU code; views are
independent renders of one read-only scene, so they shard with no data-path collective.  The
only exchange is the gather of finished frames to rank 0 (RCCL over xGMI: backend "nccl";
"gloo" for the CPU tests).  Each peer owns a direct xGMI link to the root, so the gather is a
direct 7-way send/recv, not a ring.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None, force: bool = False) -> tuple:
    """(rank, world, local_rank); initialises torch.distributed when WORLD_SIZE > 1 -- or, with ``force``, also for a
    single rank (the one-GPU rehearsal of the RCCL path: a world of one still runs every collective through RCCL)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # SAS_DIST_BACKEND=gloo is for rehearsing the multi-rank path on a box with one GPU
            backend = os.environ.get("SAS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("SAS_FORCE_DEVICE", local_rank)))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def shard_views(n_views: int, rank: int, world: int) -> List[int]:
    """Round-robin view ownership: rank r renders views r, r+world, ...."""
    return list(range(rank, n_views, world))


def broadcast_scene(scene, rank: int, world: int, src: int = 0, device: Optional[torch.device] = None,
                    fields: Sequence[str] = ("means", "quats", "scales", "opacities", "sh", "group_id"), collective: Optional[bool] = None):
    """The scene, replicated: rank ``src`` passes an object with the array attributes ``fields`` (NumPy arrays or tensors; a
    ``None`` attribute is skipped) plus ``sh_degree``; every other rank passes ``None`` and RECEIVES it -- one
    ``dist.broadcast`` per array over the backend (RCCL: device tensors over xGMI, ``ncclBroadcast``; gloo: host tensors),
    instead of every rank reading or generating its own copy ("Scene broadcast at load: ncclBroadcast once";
    1 M Gaussians = 236 MB).  Returns a ``types.SimpleNamespace`` with the same attributes as tensors on ``device`` (RCCL) or
    on the host (gloo) -- ``Rasterizer.upload`` takes either -- and ``n``, ``sh_degree``.  A world of one returns ``scene``
    (``collective=True``: it runs the broadcasts through the backend all the same: the one-GPU rehearsal of the RCCL path)."""
    import types
    import numpy as np
    if (world <= 1 and not collective) or not dist.is_initialized():
        return scene
    on_device = dist.get_backend() == "nccl"
    dev = (device or torch.device("cuda", torch.cuda.current_device())) if on_device else torch.device("cpu")
    meta = [None]
    if rank == src:
        if scene is None:
            raise ValueError("the source rank must pass the scene")
        present = {}
        for k in fields:
            a = getattr(scene, k, None)
            if a is not None:
                t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
                present[k] = (tuple(t.shape), str(t.dtype).replace("torch.", ""))
        meta = [{"fields": present, "sh_degree": int(getattr(scene, "sh_degree", 3))}]
    dist.broadcast_object_list(meta, src=src)
    out = types.SimpleNamespace(sh_degree=meta[0]["sh_degree"])
    for k in fields:
        if k not in meta[0]["fields"]:
            setattr(out, k, None)
            continue
        shape, dt = meta[0]["fields"][k]
        if rank == src:
            a = getattr(scene, k)
            t = (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))).to(dev).contiguous()
        else:
            t = torch.empty(shape, dtype=getattr(torch, dt), device=dev)
        dist.broadcast(t, src=src)
        setattr(out, k, t)
    out.n = int(out.means.shape[0])
    return out


class FrameGather:
    """Gather equally-shaped frames to rank 0 with one collective per step.

    ``start(frame)`` enqueues the gather behind the producer's stream and returns at once
    (``async_op``); ``finish()`` waits.  Keeping one gather in flight overlaps the xGMI transfer of
    frame i with the rendering of frame i+1.
    """

    def __init__(self, world: int, rank: int, dst: int = 0, collective: Optional[bool] = None):
        self.world, self.rank, self.dst = world, rank, dst
        # a single rank keeps its frame (no process group needed) -- unless `collective` asks for the real gather, which a
        # world of one still runs through the backend (RCCL): the one-GPU rehearsal of the multi-GPU path
        self.collective = world > 1 if collective is None else bool(collective)
        self._work = None
        self._bufs: Optional[List[torch.Tensor]] = None
        if self.collective and not dist.is_initialized():
            raise RuntimeError("FrameGather(collective=True) needs an initialised torch.distributed process group (init_from_env)")
        self._via_host = self.collective and dist.get_backend() == "gloo"   # gloo cannot gather device tensors

    def start(self, frame: torch.Tensor):
        if not self.collective:
            self._bufs = [frame]
            return
        self.finish()
        if self._via_host and frame.is_cuda:
            frame = frame.cpu()
        gl = None
        if self.rank == self.dst:
            if self._bufs is None or self._bufs[0].shape != frame.shape or self._bufs[0].dtype != frame.dtype:
                self._bufs = [torch.empty_like(frame) for _ in range(self.world)]
            gl = self._bufs
        self._work = dist.gather(frame, gather_list=gl, dst=self.dst, async_op=True)

    def finish(self) -> Optional[List[torch.Tensor]]:
        if self._work is not None:
            self._work.wait()
            self._work = None
        return self._bufs if self.rank == self.dst else None


class StepPipeline:
    """Render steps asynchronously into a ring of output buffers and gather each step's frames to
    rank 0 as soon as the renderer reports the step complete (bench.py's loop; the CPU tests drive it
    with a stand-in renderer over gloo).

    ``submit(i, buf)`` enqueues step ``i`` into ``buf``; ``steps_completed()`` is the number of leading
    steps whose frames are final (``Rasterizer.frames_completed`` // views per step: only completed
    frames may be consumed, include/sim_a_splat_amd.h SAS_ASYNC); ``wait()`` completes every submitted
    step.  ``payload(buf)`` picks the tensor that travels (the uint8 frames).  One gather is in flight
    at a time, so the xGMI transfer of step g overlaps the rendering of the steps after it.  A buffer
    is handed to a new step only after the gather that reads it has finished.
    """

    def __init__(self, world: int, rank: int, buffers: Sequence, submit, steps_completed, wait, payload=lambda b: b,
                 on_gathered=None, collective: Optional[bool] = None):
        if len(buffers) < 2:
            raise ValueError("need at least two buffers (entries may be None when every step brings its own: set_buffer)")
        self.world, self.rank = world, rank
        self.bufs = list(buffers)
        self._submit, self._completed, self._wait, self._payload = submit, steps_completed, wait, payload
        self._on_gathered = on_gathered
        self.gather = FrameGather(world, rank, collective=collective)
        self.base = 0          # steps_completed() at the start of this run
        self.submitted = 0     # steps submitted in this run
        self.gathered = 0      # steps whose gather has been started
        self._reading: Optional[int] = None   # step whose gather may still be reading its buffer

    def set_buffer(self, step: int, buf) -> None:
        """Give step ``step`` a buffer of its own in place of the ring's (called from ``submit``: a caller that hands the
        frames out without copying them allocates one per step)."""
        self.bufs[step % len(self.bufs)] = buf

    def buffer_of(self, step: int):
        return self.bufs[step % len(self.bufs)]

    def begin(self) -> None:
        self.base = self._completed()
        self.submitted = self.gathered = 0
        self._reading = None

    def _finish_gather(self) -> None:
        got = self.gather.finish()
        if self._reading is not None and self._on_gathered is not None and self.gather.collective:
            self._on_gathered(self._reading, got)
        self._reading = None

    def _start_gathers(self, upto: int) -> None:
        while self.gathered < upto:
            if self.gather.collective:
                self._finish_gather()
                self.gather.start(self._payload(self.bufs[self.gathered % len(self.bufs)]))
                self._reading = self.gathered
            self.gathered += 1

    def step(self) -> int:
        i, R = self.submitted, len(self.bufs)
        if i - self.gathered >= R:            # the ring is full of ungathered steps: complete them first
            self._wait()
            self._start_gathers(i)
        if self._reading is not None and self._reading % R == i % R:
            self._finish_gather()             # that buffer is about to be overwritten
        self._submit(i, self.bufs[i % R])
        self.submitted += 1
        self._start_gathers(min(self._completed() - self.base, self.submitted))
        return i

    def drain(self) -> None:
        """Complete every submitted step and gather what is left."""
        self._wait()
        self._start_gathers(self.submitted)
        if self.gather.collective:
            self._finish_gather()


def frame_meta(frames: Sequence[torch.Tensor], rank: int, world: int, src: int = 0, collective: Optional[bool] = None):
    """(shape, dtype) of the frames being gathered, agreed over all ranks: a rank that owns no view
    (n_views < world) has no frame to read them from, so rank ``src`` (which always owns view 0)
    broadcasts them."""
    meta = [None]
    if rank == src:
        if not frames:
            raise ValueError("the source rank owns view 0 and must pass its frame")
        meta = [(tuple(frames[0].shape), frames[0].dtype)]
    if world > 1 or collective:
        dist.broadcast_object_list(meta, src=src)
    return meta[0]


def gather_frames(frames: Sequence[torch.Tensor], n_views: int, rank: int, world: int,
                  collective: Optional[bool] = None) -> Optional[List[torch.Tensor]]:
    """Synchronous helper: every rank passes the frames of its ``shard_views`` views (same shape; a
    rank without views passes an empty list); rank 0 gets the list of all ``n_views`` frames in view
    order.  ``collective=True`` runs the gathers through the backend even for a world of one."""
    if world <= 1 and not collective:
        return list(frames)
    shape, dtype = frame_meta(frames, rank, world, collective=collective)
    if frames:
        device = frames[0].device
    else:   # RCCL gathers device tensors, gloo host tensors
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    rounds = (n_views + world - 1) // world
    out: List[Optional[torch.Tensor]] = [None] * n_views
    g = FrameGather(world, rank, collective=collective)
    for r in range(rounds):
        mine = frames[r] if r < len(frames) else torch.zeros(shape, dtype=dtype, device=device)
        g.start(mine)
        got = g.finish()
        if rank == 0:
            for src in range(world):
                v = r * world + src
                if v < n_views:
                    out[v] = got[src].clone()
    return out if rank == 0 else None
