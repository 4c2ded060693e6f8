"""CPU: SplatVecEnv -- E envs, per-env link poses, one batched render per rank, the uint8 observations of all ranks
gathered to rank 0 (world 2 over gloo) -- over a stand-in renderer that paints what it was asked to render.
Reference anchors: splat_env_wrapper.py:121-159 (step / _get_obs / render), examples/demo_pusht_splat.py:54-78 (cameras)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, str(Path(__file__).resolve().parent / "tools"))
import vec_env_fixture as fx  # noqa: E402


class _PaintRaster:
    """Stand-in for Rasterizer: a frame's bytes are a digest of (the view matrix, the pose set it was rendered with), so a
    test can tell WHICH env's poses and WHICH camera a gathered frame came from.  One set of link constants, like sas_ctx."""

    def __init__(self, device):
        self.n_groups, self.consts = 0, None

    def upload(self, means, *a, n_groups=0, **k):
        self.n_groups = n_groups

    def set_group_poses(self, Rt):
        pass

    def set_link_constants(self, *c):
        self.consts = c

    def link_attached_frame(self, q, p, local):
        import ctypes
        from sim_a_splat_amd import _capi
        c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        s, Ri, ti = self.consts[0], c(self.consts[1]), c(self.consts[2])
        q, p, local, w, x = c(q), c(p), c(local), np.zeros(4), np.zeros(3)
        assert _capi.lib().sas_attached_frame(s, Ri.ctypes.data, ti.ctypes.data, q.ctypes.data, p.ctypes.data, local.ctypes.data, w.ctypes.data, x.ctypes.data) == 0
        return w, x

    @staticmethod
    def paint(view, rows, H, W):
        d = np.frombuffer(np.ascontiguousarray(view, np.float32).tobytes() + np.ascontiguousarray(rows, np.float32).tobytes(), np.uint8)
        img = np.resize(d, H * W * 3).reshape(H, W, 3).copy()
        img[0, 0, 0] = int(d.astype(np.uint64).sum() % 251)
        return img

    def render_batch_host(self, V, K, W, H, bg, out=None, pose_sets=None, pose_set=None):
        V = np.asarray(V, np.float32).reshape(-1, 16)
        if out is None:
            out = torch.empty((V.shape[0], H, W, 3), dtype=torch.uint8)
        for v in range(V.shape[0]):
            out[v] = torch.from_numpy(self.paint(V[v], np.asarray(pose_sets)[pose_set[v]], H, W))
        return out

    def close(self):
        pass


def _build(rank, world, E, collective=None):
    from sim_a_splat_amd import scene as scene_mod
    from sim_a_splat_amd.handler import SplatHandler
    from sim_a_splat_amd.vec_env import SplatVecEnv
    scene_mod.Rasterizer = _PaintRaster
    sc = scene_mod.SplatScene(0)
    means, covs, colors, opac, masks, icp, fk = fx.scene_arrays()
    h = SplatHandler.from_arrays(means, covs, colors, opac, masks, icp, fk, scene=sc)
    envs = [fx.FakeEnv(e) if e % world == rank else None for e in range(E)]
    return SplatVecEnv(envs, h, fx.camera_info(), rank=rank, world=world, collective=collective), h


def _expected_obs(h, e, t, a):
    """What env e's cameras must show at (t, a): painted from ITS draw message's pose set and camera poses."""
    from sim_a_splat_amd.handler import CameraRig
    env = fx.FakeEnv(e)
    env.t, env.a = t, a
    msg = env._generate_draw_msg()
    rows = h.scene._Rt.reshape(-1, 12).copy()
    idx, link_rows = h.link_pose_rows(msg)
    rows[idx] = link_rows
    cams = CameraRig(fx.camera_info()).poses(h, msg)
    q, p = np.array([c[0] for c in cams]), np.array([c[1] for c in cams])
    V, _ = h.scene._views_and_Ks(fx.H, fx.W, q, p, h.scene.camera.fov)
    return [np.moveaxis(_PaintRaster.paint(V[c], rows, fx.H, fx.W), -1, 0) for c in range(len(cams))]


def _check(venv, h, obs, E, t, acts, owned):
    for e in range(E):
        if e not in owned:
            assert obs[e] is None
            continue
        want = _expected_obs(h, e, t, acts[e] if acts is not None else 0.0)
        assert set(obs[e]) == {"robot_pos", "camera_0", "camera_1"}
        assert obs[e]["robot_pos"].tolist() == [e, t, acts[e] if acts is not None else 0.0]
        for c in range(2):
            assert obs[e][f"camera_{c}"].shape == (3, fx.H, fx.W) and obs[e][f"camera_{c}"].dtype == np.uint8
            assert np.array_equal(obs[e][f"camera_{c}"], want[c]), (e, c)


def test_single_rank_steps_and_async_steps():
    """World of one: moving cameras first (camera_0 rides on link1), every env rendered with ITS OWN link poses; the
    asynchronous form returns the same observations one step later; link poses never touch the shared scene's block."""
    E = 5
    venv, h = _build(0, 1, E)
    assert venv.rig.render_cam_keys == [1, 0] and venv.mine == list(range(E))
    base = h.scene._Rt.copy()
    obs = venv.reset(seed=7)
    _check(venv, h, obs, E, 0, None, set(range(E)))
    acts = [0.25 * e for e in range(E)]
    obs, rew, term, trunc, info = venv.step(acts)
    _check(venv, h, obs, E, 1, acts, set(range(E)))
    assert rew == [10.0 * e + acts[e] for e in range(E)] and term == [False] * E and [i["t"] for i in info] == [1] * E
    assert np.array_equal(h.scene._Rt, base)                  # E envs, one scene: its own pose block is not the envs' business
    t1 = venv.step_async([1.0] * E)
    t2 = venv.step_async([2.0] * E)
    with pytest.raises(RuntimeError):
        venv.step_async([3.0] * E)                            # the ring holds two outstanding steps
    _check(venv, h, venv.collect(t1)[0], E, 2, [1.0] * E, set(range(E)))
    _check(venv, h, venv.collect(t2)[0], E, 3, [2.0] * E, set(range(E)))
    o, r_, *_ = venv.step([0.5] * E, noobs=True)
    assert o == [None] * E and r_[2] == 20.5
    venv.close()
    assert all(env.closed for env in venv.envs)


def test_collective_without_a_process_group_is_a_clear_error_and_padding_rows_are_zero():
    """ADVICE (round 4): ``collective=True`` with no torch.distributed process group used to fail inside ``dist.get_backend()``;
    rows of a rank's buffer beyond its own envs (the gather's common shape) were whatever the allocator handed out; ``close``
    closed the caller's scene unconditionally."""
    from sim_a_splat_amd import distributed as D
    assert not dist.is_initialized()
    with pytest.raises(RuntimeError, match="process group"):
        _build(0, 1, 3, collective=True)
    with pytest.raises(RuntimeError, match="process group"):
        D.FrameGather(1, 0, collective=True)
    venv, h = _build(1, 2, 5, collective=False)          # rank 1 of 2 owns envs 1 and 3: two of its three rows
    assert venv.mine == [1, 3] and venv.per_rank == 3
    venv.reset(seed=3)
    buf = venv._pipe.buffer_of(0)
    assert buf.shape[0] == 3 * venv.C and int(buf[2 * venv.C:].max()) == 0 and int(buf[:2 * venv.C].max()) > 0
    assert venv.h2d_frame_copies == 0 and venv.d2h_frame_copies == 0      # host frames: nothing to move
    closed = []
    h.scene.close = lambda: closed.append(True)
    venv.close(close_scene=False)
    assert not closed
    venv.close()
    assert closed == [True]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank(rank, world, port, E, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from sim_a_splat_amd import distributed as D
    D.init_from_env(backend="gloo")
    venv, h = _build(rank, world, E)
    try:
        owned = set(range(E)) if rank == 0 else set(D.shard_views(E, rank, world))
        obs = venv.reset(seed=1)
        _check(venv, h, obs, E, 0, None, owned)
        for t in range(1, 4):
            acts = [0.5 * t + 0.01 * e for e in range(E)]
            obs, rew, term, trunc, info = venv.step(acts)
            _check(venv, h, obs, E, t, acts, owned)
            if rank == 0:
                assert rew == [10.0 * e + acts[e] for e in range(E)] and [i["e"] for i in info] == list(range(E))
        tk = venv.step_async([9.0] * E)
        _check(venv, h, venv.collect(tk)[0], E, 4, [9.0] * E, owned)
        q.put((rank, "ok"))
    except Exception as ex:   # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("E", [4, 5])
def test_two_ranks_gather_every_envs_observations_to_rank0(E):
    """World 2 over gloo: env e lives on rank e mod 2 (physics, poses, rendering); rank 0 receives the camera
    observations of ALL envs, each painted from its own env's pose set and cameras; rank 1 sees its own.  E = 5: the
    ranks own 3 and 2 envs and still gather one shape."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, E, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == {0: "ok", 1: "ok"}, res
