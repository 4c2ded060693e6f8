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


def init_from_env(backend: Optional[str] = None) -> tuple:
    """(rank, world, local_rank); initialises torch.distributed when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
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


class FrameGather:
    """Gather equally-shaped frames to rank 0 with one collective per step.

    ``start(frame)`` enqueues the gather behind the producer's stream and returns at once
    (``async_op``); ``finish()`` waits.  Keeping one gather in flight overlaps the xGMI transfer of
    frame i with the rendering of frame i+1.
    """

    def __init__(self, world: int, rank: int, dst: int = 0):
        self.world, self.rank, self.dst = world, rank, dst
        self._work = None
        self._bufs: Optional[List[torch.Tensor]] = None
        self._via_host = world > 1 and dist.get_backend() == "gloo"   # gloo cannot gather device tensors

    def start(self, frame: torch.Tensor):
        if self.world <= 1:
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


def gather_frames(frames: Sequence[torch.Tensor], n_views: int, rank: int, world: int) -> Optional[List[torch.Tensor]]:
    """Synchronous helper: every rank passes the frames of its ``shard_views`` views (same shape);
    rank 0 gets the list of all ``n_views`` frames in view order."""
    if world <= 1:
        return list(frames)
    rounds = (n_views + world - 1) // world
    out: List[Optional[torch.Tensor]] = [None] * n_views
    g = FrameGather(world, rank)
    for r in range(rounds):
        mine = frames[r] if r < len(frames) else torch.zeros_like(frames[0])
        g.start(mine)
        got = g.finish()
        if rank == 0:
            for src in range(world):
                v = r * world + src
                if v < n_views:
                    out[v] = got[src].clone()
    return out if rank == 0 else None
