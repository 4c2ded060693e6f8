"""GPU: two library contexts driven from two host threads AT THE SAME TIME (a viewer thread and an env thread with a scene each,
examples/demo_hw_splat.py:113-136 in the reference).  A context is not re-entrant, but contexts are independent: each thread's
frames -- blocking and pipelined, new group poses per step -- must equal the oracle's, whatever the other thread is doing."""
import threading

import numpy as np
import pytest
import torch

import oracle
from sim_a_splat_amd.rasterizer import Rasterizer
from sim_a_splat_amd.synthetic import NERFSTUDIO_EVAL_BACKGROUND as BG, make_scene, random_group_poses, ring_camera

pytestmark = pytest.mark.gpu
STEPS = 24


def _worker(idx, errors, done, start):
    try:
        W, H = (176, 120) if idx == 0 else (331, 207)        # (quad layout / 16-pixel tiles)
        sc = make_scene(4000 + 9000 * idx, seed=300 + idx, log_scale_mean=float(np.log(0.03)), n_groups=4)
        cams = [ring_camera(W, H, 0.9 * W, yaw_deg=25.0 * s + 100.0 * idx, elev=0.05 * s - 0.4) for s in range(STEPS)]
        poses = [random_group_poses(4, seed=500 + 31 * idx + s) for s in range(STEPS)]
        want = [oracle.render(sc.means, sc.opacities, sc.sh, cams[s].viewmat, cams[s].K, W, H, quats=sc.quats, scales=sc.scales, sh_degree=3,
                              group_id=sc.group_id, group_Rt=poses[s], background=BG, want_rgb8=True) for s in range(STEPS)]
        start.wait(120)                                       # both threads enter the library together
        r = Rasterizer(0)
        r.upload(sc.means, sc.opacities, sc.sh, quats=sc.quats, scales=sc.scales, sh_degree=3, group_id=sc.group_id, n_groups=4)
        for rep in range(8):
            outs = []
            for s in range(STEPS):
                r.set_group_poses(poses[s])
                blocking = (s + rep + idx) % 3 == 0
                o = r.render(cams[s].viewmat, cams[s].K, W, H, BG, want=("rgb", "alpha", "depth", "rgb8"), block=blocking)
                outs.append(o)
            r.wait()
            torch.cuda.synchronize()
            for s in range(STEPS):
                for k in ("rgb", "alpha", "depth", "rgb8"):
                    if not np.array_equal(outs[s][k].cpu().numpy(), want[s][k]):
                        errors.append(f"thread {idx} rep {rep} step {s}: {k} differs from the oracle")
        r.close()
        done[idx] = True
    except Exception as e:                                     # pragma: no cover
        errors.append(f"thread {idx}: {e!r}")


def test_two_contexts_from_two_threads_render_their_own_scenes_bit_equal_to_the_oracle():
    assert torch.cuda.is_available()
    errors, done = [], [False, False]
    start = threading.Barrier(2)
    ts = [threading.Thread(target=_worker, args=(i, errors, done, start)) for i in range(2)]
    [t.start() for t in ts]
    [t.join(300) for t in ts]
    assert not any(t.is_alive() for t in ts), "a thread did not finish"
    assert not errors, errors[:5]
    assert all(done)
