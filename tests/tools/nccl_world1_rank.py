"""The ONE rank of tests/test_gpu_a_nccl_world1.py: torch.distributed on backend "nccl" (= RCCL on ROCm) with a world of one,
on the one card of the GPU box.  Everything sim_a_splat_amd.distributed does for a multi-GPU run goes through RCCL here with
DEVICE tensors -- init_from_env's nccl arm, FrameGather.start/finish, StepPipeline with asynchronous steps, gather_frames --
and what rank 0 gathers is compared bit for bit with blocking renders of the same views.  Prints one JSON line."""
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from sim_a_splat_amd import distributed as D  # noqa: E402
from sim_a_splat_amd.rasterizer import Rasterizer  # noqa: E402
from sim_a_splat_amd.synthetic import NERFSTUDIO_EVAL_BACKGROUND as BG, make_scene, random_group_poses, ring_camera  # noqa: E402

os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
rank, world, local = D.init_from_env(backend="nccl", force=True)
assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
dev = torch.device("cuda", torch.cuda.current_device())

W, H, V, STEPS = 320, 240, 4, 6
sc = make_scene(60_000, seed=41, log_scale_mean=float(np.log(0.02)), n_groups=4)
r = Rasterizer(dev.index)
r.upload(sc.means, sc.opacities, sc.sh, quats=sc.quats, scales=sc.scales, sh_degree=sc.sh_degree, group_id=sc.group_id, n_groups=4)
cams = [ring_camera(W, H, 280.0, yaw_deg=90.0 * k, elev=0.1 * k) for k in range(V)]
Vs, Ks = np.stack([c.viewmat for c in cams]), np.stack([c.K for c in cams])
poses = [random_group_poses(4, seed=100 + s) for s in range(STEPS)]

# what every step must deliver: blocking renders, one view at a time
want = []
for s in range(STEPS):
    r.set_group_poses(poses[s])
    want.append(torch.stack([r.render(Vs[v], Ks[v], W, H, BG, want=("rgb8",))["rgb8"].clone() for v in range(V)]))

# ---- StepPipeline: asynchronous steps, each step's uint8 frames gathered through RCCL while later steps render ----------
bufs = [{"rgb8": torch.empty((V, H, W, 3), dtype=torch.uint8, device=dev)} for _ in range(3)]
gathered = {}


def submit(i, buf):
    r.set_group_poses(poses[i])
    r.render_batch(Vs, Ks, W, H, BG, want=("rgb8",), out=buf, block=False)


def on_gathered(step, got):
    assert len(got) == 1 and got[0].is_cuda
    gathered[step] = got[0].clone()


base = r.frames_completed()[1]
pipe = D.StepPipeline(world, rank, bufs, submit, lambda: (r.frames_completed()[1] - base) // V, r.wait, payload=lambda b: b["rgb8"],
                      on_gathered=on_gathered, collective=True)
pipe.begin()
for i in range(STEPS):
    pipe.step()
pipe.drain()
pipeline_ok = sorted(gathered) == list(range(STEPS)) and all(torch.equal(gathered[s], want[s]) for s in range(STEPS))

# ---- gather_frames: the synchronous helper, device tensors, view order ----------------------------------------------------
mine = [want[0][v] for v in D.shard_views(V, rank, world)]
got = D.gather_frames(mine, V, rank, world, collective=True)
helper_ok = got is not None and len(got) == V and all(g.is_cuda and torch.equal(g, want[0][v]) for v, g in enumerate(got))

# ---- FrameGather alone: one gather in flight, a float32 frame (24.9 MB at 1080p; here the same path at 320x240) ---------
fg = D.FrameGather(world, rank, collective=True)
f32 = r.render(Vs[1], Ks[1], W, H, BG, want=("rgb",))["rgb"].clone()
fg.start(f32)
res = fg.finish()
float_ok = res is not None and torch.equal(res[0], f32)

# ---- SplatVecEnv with its gather forced through RCCL (device payload up, gathered frames back down) ------------------------
sys.path.insert(0, str(Path(__file__).resolve().parent))
import vec_env_fixture as fx  # noqa: E402
from sim_a_splat_amd.handler import SplatHandler  # noqa: E402
from sim_a_splat_amd.vec_env import SplatVecEnv  # noqa: E402
means, covs, colors, opac, masks, icp, fk = fx.scene_arrays()
h = SplatHandler.from_arrays(means, covs, colors, opac, masks, icp, fk, device=dev.index)
E = 3
plain = SplatVecEnv([fx.FakeEnv(e) for e in range(E)], h, fx.camera_info(), rank=0, world=1)                       # no collective: the reference
via_rccl = SplatVecEnv([fx.FakeEnv(e) for e in range(E)], h, fx.camera_info(), rank=0, world=1, collective=True)   # gather through nccl
vec_ok, seen = via_rccl._device_payload, 0
for step in range(3):
    acts = [0.1 * step + 0.01 * e for e in range(E)]
    a = plain.reset(seed=5) if step == 0 else plain.step(acts)[0]
    b = via_rccl.reset(seed=5) if step == 0 else via_rccl.step(acts)[0]
    for e in range(E):
        for c in range(2):
            vec_ok = vec_ok and bool(np.array_equal(a[e][f"camera_{c}"], b[e][f"camera_{c}"]))
            seen = max(seen, int(a[e][f"camera_{c}"].max()))
vec_ok = vec_ok and seen > 0          # (the viewport camera looks at the scene; the link camera may not)
# the RCCL arm keeps the frames on the device until they are on the root: rendered into device buffers, gathered as they
# are, ONE download per step on rank 0 -- no finished frame is ever uploaded (round 4: host -> device -> xGMI -> device -> host)
vec_device_resident = via_rccl._device_payload and via_rccl.h2d_frame_copies == 0 and via_rccl.d2h_frame_copies == 3 \
    and plain.h2d_frame_copies == 0 and plain.d2h_frame_copies == 0

# ---- the scene broadcast (ncclBroadcast of device tensors; SURVEY.md 8e "Scene broadcast at load") ---------------------------------
bs = D.broadcast_scene(sc, rank, world, device=dev, collective=True)
scene_ok = bs.n == sc.n and all(getattr(bs, k).is_cuda and np.array_equal(getattr(bs, k).cpu().numpy(), getattr(sc, k))
                                for k in ("means", "quats", "scales", "opacities", "sh", "group_id"))
r2 = Rasterizer(dev.index)
r2.upload(bs.means, bs.opacities, bs.sh, quats=bs.quats, scales=bs.scales, sh_degree=bs.sh_degree, group_id=bs.group_id, n_groups=4)
r.set_group_poses(poses[0]); r2.set_group_poses(poses[0])
scene_ok = scene_ok and torch.equal(r2.render(Vs[0], Ks[0], W, H, BG, want=("rgb8",))["rgb8"], r.render(Vs[0], Ks[0], W, H, BG, want=("rgb8",))["rgb8"])
r2.close()

ver = ".".join(str(x) for x in torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else "?"
print(json.dumps({"backend": dist.get_backend(), "world": world, "rccl_version": ver, "steps": STEPS, "views_per_step": V,
                  "pipeline_bit_equal": bool(pipeline_ok), "gather_frames_bit_equal": bool(helper_ok), "float_frame_bit_equal": bool(float_ok),
                  "scene_broadcast_bit_equal": bool(scene_ok), "vec_env_bit_equal": bool(vec_ok), "vec_env_device_resident": bool(vec_device_resident),
                  "vec_env_copies": {"h2d": via_rccl.h2d_frame_copies, "d2h": via_rccl.d2h_frame_copies}, "device_tensors": True}))
r.close()
dist.destroy_process_group()
