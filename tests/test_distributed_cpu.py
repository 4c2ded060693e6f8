"""CPU, world_size 2 over gloo: view sharding and the gather of finished frames."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sim_a_splat_amd import distributed as sdist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_views, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = sdist.init_from_env(backend="gloo")
    mine = sdist.shard_views(n_views, r, w)
    # a "frame" that encodes its view id, as if rendered by that rank
    frames = [torch.full((4, 6, 3), float(v), dtype=torch.float32) for v in mine]
    got = sdist.gather_frames(frames, n_views, r, w)
    # async one-behind gather, as bench.py uses it
    g = sdist.FrameGather(w, r)
    last = None
    for step in range(3):
        g.start(torch.full((2, 2), float(10 * step + r)))
        last = g.finish()
    if r == 0:
        q.put(([int(f[0, 0, 0].item()) for f in got], [int(t[0, 0].item()) for t in last]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_views", [2, 5])
def test_shard_and_gather_two_ranks(n_views):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_views, q)) for r in range(2)]
    for p in procs:
        p.start()
    views, last = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert views == list(range(n_views))          # rank 0 holds every view, in view order
    assert last == [20, 21]                       # last async gather: step 2 from ranks 0 and 1


def test_shard_views_partition():
    for n in (1, 4, 8, 9):
        for w in (1, 2, 4, 8):
            parts = [sdist.shard_views(n, r, w) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
