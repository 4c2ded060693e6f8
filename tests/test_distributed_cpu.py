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


@pytest.mark.parametrize("n_views", [1, 2, 5])
def test_shard_and_gather_two_ranks(n_views):
    """n_views = 1: rank 1 owns no view (the reference's 2 cameras over 8 GPUs, in small) and must still
    take part in the gather instead of raising and leaving rank 0 blocked."""
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


def _scene_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sdist.init_from_env(backend="gloo")
    from sim_a_splat_amd.synthetic import make_scene
    mine = make_scene(300, seed=9, n_groups=3) if rank == 0 else None       # only the source rank has (loads, generates) the scene
    got = sdist.broadcast_scene(mine, rank, world)
    ref = make_scene(300, seed=9, n_groups=3)
    ok = got.n == 300 and got.sh_degree == 3 and all(
        isinstance(getattr(got, k), torch.Tensor) and np.array_equal(getattr(got, k).numpy(), getattr(ref, k))
        for k in ("means", "quats", "scales", "opacities", "sh", "group_id"))
    plain = sdist.broadcast_scene(make_scene(10, seed=1) if rank == 0 else None, rank, world)   # no group ids: the field stays None
    ok = ok and plain.group_id is None and plain.n == 10
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_scene_broadcast_from_rank0_two_ranks():
    """the scene is replicated with ONE broadcast at load (ncclBroadcast on a multi-GPU node; gloo here): rank 0
    holds the arrays, every other rank receives them bit for bit (and None for a field the scene does not have)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_scene_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == {0: True, 1: True}
    assert sdist.broadcast_scene("anything", 0, 1) == "anything"          # a world of one keeps what it has


def test_shard_views_partition():
    for n in (1, 4, 8, 9):
        for w in (1, 2, 4, 8):
            parts = [sdist.shard_views(n, r, w) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


class _FakeRenderer:
    """Stand-in for the asynchronous rasterizer: a step's frames become final `lag` submissions later
    (or at wait()), exactly as frames complete inside later sas_render* calls.  Until then the buffer
    holds garbage; it also records any write into a buffer a gather might still be reading."""

    def __init__(self, rank, lag):
        self.rank, self.lag = rank, lag
        self.pending = []          # (step, buf)
        self.completed = 0
        self.reading = {}          # id(buf) -> step being gathered
        self.violations = []

    def value(self, step):
        return float(100 * step + self.rank)

    def submit(self, i, buf):
        if id(buf) in self.reading:
            self.violations.append(("overwrite while gathering", i, self.reading[id(buf)]))
        buf.fill_(-1.0)            # a truncated / in-flight frame
        self.pending.append((i, buf))
        while len(self.pending) > self.lag:
            self._complete_one()

    def _complete_one(self):
        i, buf = self.pending.pop(0)
        buf.fill_(self.value(i))
        self.completed += 1

    def steps_completed(self):
        return self.completed

    def wait(self):
        while self.pending:
            self._complete_one()


def _pipeline_worker(rank, world, port, n_steps, lag, n_bufs, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = sdist.init_from_env(backend="gloo")
    fake = _FakeRenderer(r, lag)
    bufs = [torch.zeros((3, 5)) for _ in range(n_bufs)]
    seen = []

    def on_gathered(step, got):
        fake.reading = {}
        if r == 0:
            seen.append((step, [float(t[0, 0].item()) for t in got]))

    pipe = sdist.StepPipeline(w, r, bufs, fake.submit, fake.steps_completed, fake.wait, on_gathered=on_gathered)
    orig_start = pipe.gather.start

    def start(frame):
        orig_start(frame)
        fake.reading = {id(b): pipe.gathered for b in bufs if b is frame}
    pipe.gather.start = start
    for rnd in range(2):           # warm-up run, then the timed run: bench.py calls begin() twice
        pipe.begin()
        for _ in range(n_steps):
            pipe.step()
        pipe.drain()
    if r == 0:
        q.put((seen, fake.violations))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("lag,n_bufs", [(2, 4), (1, 2), (3, 3)])
def test_step_pipeline_gathers_only_completed_steps(lag, n_bufs):
    """bench.py's buffer rotation and lagging gather, driven over gloo by a stand-in renderer whose
    frames become final `lag` steps after submission: rank 0 must receive every step exactly once, in
    order, with the FINAL values of both ranks, and no buffer may be handed to a new step while the
    gather that reads it is in flight -- also when the ring is smaller than the completion lag."""
    n_steps = 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, n_steps, lag, n_bufs, q)) for r in range(2)]
    for p in procs:
        p.start()
    seen, violations = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert violations == []
    want = [(s, [100.0 * s, 100.0 * s + 1.0]) for s in range(n_steps)] * 2
    assert seen == want


# ---- bench.py --gpus N launches its own ranks ----------------------------------------------------------------------
def _bench(args, env_extra=None, timeout=240):
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, str(root / "bench.py"), *args], capture_output=True, text=True, env=env, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr


def test_bench_gpus_n_without_a_launcher_starts_n_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset must run TWO ranks (it used to print n_gpus: 1): the parent
    starts torch.distributed.run itself; the stand-in renderer (--dry-run, gloo) exercises sharding and the gather."""
    rc, line, err = _bench(["--gpus", "2", "--dry-run", "--steps", "4", "--warmup", "1"])
    assert rc == 0, err[-2000:]
    assert line["n_gpus"] == 2 and line["dry_run"] and line["gather_ok"] and line["gathered_steps"] == 5
    assert line["config"]["views_per_step"] == 4          # weak scaling: two views per rank


def test_bench_world_size_must_match_gpus():
    rc, line, err = _bench(["--gpus", "8", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert rc != 0 and line is None and "WORLD_SIZE=1" in err


def test_bench_gather_with_a_world_that_does_not_divide_the_views():
    """Config 4 (8 views) on 3 ranks: every rank's gather payload is ceil(8 / 3) = 3 frames (dist.gather needs one
    shape on all ranks), ranks with fewer views pad."""
    rc, line, err = _bench(["--gpus", "3", "--dry-run", "--config", "4", "--steps", "3", "--warmup", "1"])
    assert rc == 0, err[-2000:]
    assert line["n_gpus"] == 3 and line["gather_ok"] and line["config"]["views_per_step"] == 8
