#!/usr/bin/env python3
"""Headline benchmark: rendered frames/s at 1M Gaussians, 1920x1080 (BASELINE.json config 3).

    python bench.py --gpus N --steps K --warmup W [--config 3|4|5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path (project + SH, bin, per-tile depth order, composite) over one batch
of independent views per GPU of a synthetic scene resident in HBM:

  config 3 (default, the metric's config)  1M Gaussians, SH degree 3, 1920x1080, two views per GPU per
            step (one sas_render_batch call; the pair shares one projection pass), float32 RGB + uint8 RGB
            out per view.  Weak scaling: every rank renders its own views of the replicated scene.
  config 4  292,247-Gaussian grouped scene (7 link groups, poses updated every step), 8 ring poses at
            640x480, uint8 frames, the 8 views sharded over the ranks (strong scaling, north_star
            "one per GPU on 8 GPUs").
  config 5  5M Gaussians, four 1920x1080 views sharded over (up to 4) ranks (strong scaling).

With N > 1 the finished uint8 frames are gathered to rank 0 over RCCL as soon as the renderer reports
them complete (sim_a_splat_amd.distributed.StepPipeline).  Rank 0 prints ONE JSON line; value = frames/s
over all ranks.  The oracle is used only for the `cpu_baseline` leg (rank 0, N = 1, bounded sample).

`--gpus N` without a launcher (WORLD_SIZE unset) starts the N ranks ITSELF (torch.distributed.run as a child
process, before anything in this process touches a GPU) and passes their output and exit code on; under a
launcher WORLD_SIZE must equal N.  `--dry-run` replaces the renderer by a stand-in that needs no GPU (gloo):
the launch, sharding and gather path on its own, for the CPU tests.

Clock ramp.  After an idle period (the seconds of scene generation in front of the first frame) the MI355X takes
~30 ms of sustained load to reach its steady clocks (tools/ramp_probe.py, profiles/r03_clock_ramp.txt: 0.38 -> 0.32 ms
per step over the first four chunks of 20 steps).  W = 5 warm-up steps are 2 ms.  The bench therefore REPEATS the
W + K protocol back to back -- W untimed steps, a synchronisation, exactly K timed steps, a synchronisation -- until
a pass is within 1 % of the one before it and three passes lie behind the ramp (at most 12 passes; a K-step pass of more
than 0.25 s is long enough by itself and is not repeated).  `value` is the MEDIAN of the passes behind the ramp (those that start once ~30 ms of load
have gone by; protocol version 3 -- version 2 reported the LAST pass, which depended on which of two alternating
interleavings of the frames in flight the pass cap fell on); `converged` says whether the last two passes agreed within
1 %, `pass_spread` is (max - min) / median over the same passes; every pass is listed in `passes`, the first one -- from
the idle GPU -- also as `cold_start`.  `--single-pass` runs one pass and reports that.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from sim_a_splat_amd import distributed as sdist  # noqa: E402
from sim_a_splat_amd.rasterizer import Rasterizer  # noqa: E402
from sim_a_splat_amd.synthetic import (NERFSTUDIO_EVAL_BACKGROUND as BG, config_scene_and_cameras, make_scene,  # noqa: E402
                                       random_group_poses, ring_camera)

HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# Vector-instruction issue peak: 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction
# (MI355X_MICROARCH.md: v_fma_f32 2 cycles per SIMD = the 157 TFLOP/s fp32 vector peak).  Measured with
# tools/microbench/clock_probe.hip: 1.61 shader cycles per instruction per SIMD with 8 waves resident at the
# 1.9-1.95 GHz the chip holds under that load = 1.21e12/s, the same number; 1.98 cycles with 4 waves, and one
# wave alone issues at most once per 6.75 cycles (profiles/r02_issue_rate_microbench.txt).
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2.0


def scene_pass_bytes(n):
    """the N*236 B of Gaussian attributes one projection pass reads."""
    return n * 236


def view_bytes(n_vis, m, w, h, out_bytes_per_pixel):
    """SURVEY.md 8d, the per-view terms: N_vis*44 + M*60 + W*H*out."""
    return n_vis * 44 + m * 60 + w * h * out_bytes_per_pixel


def tile_kernel_bytes(m, w, h, out_bytes_per_pixel):
    """Dominant kernel (k_tile_lazy = per-tile ordering + compositing), SURVEY.md 8d terms:
    8 B key read per intersection + 44 B gather per intersection (id 4 + record 40) + the frame."""
    return m * (8 + 44) + w * h * out_bytes_per_pixel


def committed(name):
    p = ROOT / "profiles" / name
    try:
        return json.loads(p.read_text())
    except (OSError, ValueError):
        return None


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc summary (FETCH_SIZE is
    doubled as MI355X_MICROARCH.md prescribes for gfx950); None when no summary is committed."""
    d = committed("hbm_traffic.json")
    try:
        return d["kernels"][kernel]["hbm_bytes_per_launch"]
    except (KeyError, TypeError):
        return None


def cpu_baseline(scene, cam, budget_s=20.0):
    import oracle
    threads = oracle.num_threads()
    kw = dict(quats=scene.quats, scales=scene.scales, sh_degree=3, background=BG)
    t0 = time.perf_counter()
    oracle.render(scene.means, scene.opacities, scene.sh, cam.viewmat, cam.K, cam.width, cam.height, **kw)
    first = time.perf_counter() - t0
    times = []
    while len(times) < 10 and (sum(times) + first) < budget_s or len(times) < 2:
        t0 = time.perf_counter()
        oracle.render(scene.means, scene.opacities, scene.sh, cam.viewmat, cam.K, cam.width, cam.height, **kw)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return {"value": 1.0 / med, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{len(times)} full frames of the same workload after 1 warm-up (median), C oracle with "
                      f"OpenMP on {threads} threads of {os.cpu_count()} host CPUs"}


def workload(cfg, rank, world, views_per_step, n_gaussians, with_scene=True):
    """(scene, cameras of THIS rank, want, per-step group poses or None, description, scaling, all-rank views per step).
    ``with_scene=False`` (ranks that receive the scene from rank 0: distributed.broadcast_scene): scene is None."""
    if cfg == 3:
        scene = make_scene(n_gaussians, seed=3, log_scale_mean=float(np.log(0.006))) if with_scene else None
        cams = [ring_camera(1920, 1080, 1000.0, yaw_deg=45.0 * rank + 180.0 * v) for v in range(views_per_step)]
        desc = (f"BASELINE config 3: {n_gaussians / 1e6:g}M synthetic Gaussians (seed 3, SH degree 3), 1920x1080, fx=fy=1000, "
                f"{views_per_step} independent view(s) per GPU per step"
                + (" (a view pair shares one projection pass)" if views_per_step == 2 else "") + ", float32 RGB + uint8 RGB out per view "
                "(no accumulation / depth in `value`: the pipelined figure with every output GaussianSplat.render returns is `door_a_async`)")
        return scene, cams, ("rgb", "rgb8"), None, desc, "weak", world * views_per_step
    if with_scene:
        scene, all_cams = config_scene_and_cameras(cfg)
    else:
        from sim_a_splat_amd.synthetic import config_cameras
        scene, all_cams = None, config_cameras(cfg)
    mine = sdist.shard_views(len(all_cams), rank, world)
    cams = [all_cams[v] for v in mine]
    if cfg == 4:
        poses = [random_group_poses(7, seed=1000 + s) for s in range(16)]   # a rollout's link poses, cycled
        desc = ("BASELINE config 4: 292,247-Gaussian stand-in of the pushT scene (seed 2, SH degree 3, 7 link groups, poses updated every "
                f"step), 8 ring poses at 640x480 sharded over {world} GPU(s), uint8 RGB out (Door B)")
        return scene, cams, ("rgb8",), poses, desc, "strong", len(all_cams)
    desc = (f"BASELINE config 5: 5M synthetic Gaussians (seed 5, SH degree 3), four 1920x1080 views (yaw 0/90/180/270) sharded over "
            f"{min(world, 4)} GPU(s), float32 RGB + uint8 RGB out per view")
    return scene, cams, ("rgb", "rgb8"), None, desc, "strong", len(all_cams)


def cameras_of(cfg, rank, world, views_per_step):
    """The cameras rank `rank` renders per step (workload()'s second result, without the scene)."""
    if cfg == 3:
        return [ring_camera(1920, 1080, 1000.0, yaw_deg=45.0 * rank + 180.0 * v) for v in range(views_per_step)]
    from sim_a_splat_amd.synthetic import config_cameras
    all_cams = config_cameras(cfg)
    return [all_cams[v] for v in sdist.shard_views(len(all_cams), rank, world)]


def self_launch(n, argv, dry_run):
    """`bench.py --gpus N` without a launcher: N ranks through torch.distributed.run, as the driver starts them.
    Returns the child's exit code; its stdout (rank 0's JSON line) is passed through."""
    import socket
    import subprocess
    rehearsal = "SAS_FORCE_DEVICE" in os.environ   # all ranks on one GPU (gloo): the multi-rank path on a 1-GPU box
    if not dry_run and not rehearsal and torch.cuda.device_count() < n:   # (device_count does not initialise the GPU)
        print(f"bench.py: --gpus {n} but only {torch.cuda.device_count()} GPU(s) are visible", file=sys.stderr)
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *argv]
    return subprocess.run(cmd).returncode


def dry_run(a, rank, world):
    """The N-rank path without a GPU: views sharded, a stand-in renderer whose frames are complete at once, the
    uint8 frames gathered to rank 0 over gloo by the same StepPipeline; rank 0 checks what arrived."""
    n_views = {3: world * a.views_per_step, 4: 8, 5: 4}[a.config]
    mine = list(range(rank * a.views_per_step, (rank + 1) * a.views_per_step)) if a.config == 3 else sdist.shard_views(n_views, rank, world)
    VB = max(1, a.views_per_step if a.config == 3 else -(-n_views // world))
    bufs = [torch.zeros((VB, 4, 6, 3), dtype=torch.uint8) for _ in range(4)]
    done = [0]
    got = []

    def submit(i, buf):
        for k, v in enumerate(mine):
            buf[k] = (7 * i + v) % 251
        done[0] += 1

    def on_gathered(step, frames):
        if rank == 0:
            got.append((step, [int(f[0, 0, 0, 0]) for f in frames]))

    pipe = sdist.StepPipeline(world, rank, bufs, submit, lambda: done[0], lambda: None, on_gathered=on_gathered)
    t0 = time.perf_counter()
    pipe.begin()
    for _ in range(a.warmup + a.steps):
        pipe.step()
    pipe.drain()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ok = True
    if rank == 0 and world > 1:
        for step, firsts in got:
            for src, val in enumerate(firsts):
                own = list(range(src * a.views_per_step, (src + 1) * a.views_per_step)) if a.config == 3 else sdist.shard_views(n_views, src, world)
                ok = ok and (not own or val == (7 * step + own[0]) % 251)
        ok = ok and len(got) == a.warmup + a.steps
    if rank == 0:
        print(json.dumps({"metric": "dry run (no GPU): launch / shard / gather path", "value": n_views * a.steps / elapsed, "unit": "stand-in frames/s",
                          "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "dry_run": True, "gathered_steps": len(got) if world > 1 else a.warmup + a.steps,
                          "gather_ok": bool(ok), "config": {"workload": f"stand-in renderer, config {a.config} sharding", "views_per_step": n_views}}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=3, choices=(3, 4, 5), help="BASELINE.json config (3 = the metric's)")
    ap.add_argument("--gaussians", type=int, default=1_000_000, help="config 3 only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the single-view / Door-A / per-stage measurements after the timed region")
    ap.add_argument("--time-every", type=int, default=4, help="steps between tile-kernel timing samples")
    ap.add_argument("--views-per-step", type=int, default=2, choices=(1, 2),
                    help="config 3: independent views each GPU renders per step; 2 = a view pair (one projection pass for both)")
    ap.add_argument("--single-pass", action="store_true", help="one W + K pass from the idle GPU (no second, clock-ramped pass)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: a stand-in renderer over gloo (launch / shard / gather path only)")
    a = ap.parse_args()

    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # no launcher: start the ranks ourselves, BEFORE this process touches a GPU (a child process, not an exec)
        raise SystemExit(self_launch(a.gpus, sys.argv[1:], a.dry_run))
    rank, world, local_rank = sdist.init_from_env("gloo" if a.dry_run else None)
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus must agree")
    if a.dry_run:
        return dry_run(a, rank, world)
    dev_index = int(os.environ.get("SAS_FORCE_DEVICE", local_rank))   # rehearsal on a 1-GPU box only
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    # the scene is made once, on rank 0, and broadcast (RCCL: ncclBroadcast of device tensors over xGMI; gloo in the rehearsal)
    scene, cams, want, step_poses, desc, scaling, views_all = workload(a.config, rank, world, a.views_per_step, a.gaussians, with_scene=rank == 0)
    scene = sdist.broadcast_scene(scene, rank, world, device=dev)
    V = len(cams)                                  # views this rank renders per step (0 for a rank beyond the view count)
    cam = cams[0] if cams else ring_camera(1920, 1080, 1000.0)
    W, H = (cam.width, cam.height)
    r = Rasterizer(dev)
    r.upload(scene.means, scene.opacities, scene.sh, quats=scene.quats, scales=scene.scales, sh_degree=3,
             group_id=scene.group_id if step_poses else None, n_groups=7 if step_poses else 0)
    Vs = np.stack([c.viewmat for c in cams]) if cams else np.zeros((0, 4, 4), np.float32)
    Ks = np.stack([c.K for c in cams]) if cams else np.zeros((0, 3, 3), np.float32)
    # Every rank writes the outputs the config names; the gather moves the uint8 frames, the format Gym
    # observations are exchanged in (splat_env_wrapper.py:135-137): 6.2 MB instead of 24.9 MB per 1080p frame
    # keeps the xGMI transfer shorter than a frame.  Four buffers: frames complete up to two steps behind
    # their submission and one gather may still be reading (distributed.StepPipeline).
    # every rank's gather payload has the same shape: ceil(views / world) frames (ranks with fewer views pad)
    VB = max(V, 1) if a.config == 3 else max(1, -(-views_all // world))
    shapes = {"rgb": ((VB, H, W, 3), torch.float32), "rgb8": ((VB, H, W, 3), torch.uint8)}
    bufs = [{k: torch.zeros(shapes[k][0], dtype=shapes[k][1], device=dev) for k in want} for _ in range(4)]
    timing_on = [False]
    idle_steps = [0]   # a rank beyond the view count (config 5 on 8 GPUs) renders nothing: its steps are complete at once

    def submit(i, out):
        if step_poses is not None:
            r.set_group_poses(step_poses[i % len(step_poses)])   # the Gym step's new link poses (frames in flight keep theirs)
        if V:
            r.render_batch(Vs, Ks, W, H, BG, want=want, out={k: v[:V] for k, v in out.items()}, block=False, time_tiles=timing_on[0])
        else:
            idle_steps[0] += 1

    # what rank 0 received for the LAST step of a timed run, checked against its own renders of every rank's views after
    # the timing (the scene is replicated: rank 0 can render any view)
    keep = {"on": False, "step": None, "frames": None}

    def on_gathered(step, frames):
        if rank == 0 and keep["on"] and frames is not None and step == a.steps - 1:
            keep["step"], keep["frames"] = step, [f.clone() for f in frames]

    pipe = sdist.StepPipeline(world, rank, bufs, submit, lambda: (r.frames_completed()[1] // V) if V else idle_steps[0], r.wait,
                              payload=lambda b: b["rgb8"], on_gathered=on_gathered)

    def run(steps, time_every):
        pipe.begin()
        for i in range(steps):
            # SAS_TIME_TILES on every time_every-th step: HIP events around the dominant kernel of that
            # step's frames (on the kernel's own stream); the frames keep pipelining
            timing_on[0] = time_every > 0 and i % time_every == 0
            pipe.step()
        pipe.drain()               # completes every frame, gathers what is left
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    def timed_pass():
        run(a.warmup, 1)
        r.stage_time_means(reset=True)
        keep["on"] = True
        t0 = time.perf_counter()
        run(a.steps, a.time_every)
        dt = time.perf_counter() - t0
        keep["on"] = False
        means, frames = r.stage_time_means(reset=True)
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, means["blend"], frames

    # first pass from the idle GPU (its clocks ramp for ~30 ms: module docstring); repeated until two passes agree
    cold_elapsed, tile_ms, timed_frames = timed_pass()
    elapsed = cold_elapsed
    pass_times = [cold_elapsed]
    # (two passes agreeing is accepted only once the passes so far cover the ~30 ms the clocks take to ramp: a chance
    # agreement of the second and third pass, 16 ms in, ended runs 5 % low)
    converged = a.single_pass or pass_times[-1] >= 0.25
    MAX_PASSES, MIN_BEHIND_RAMP = 12, 3
    while not a.single_pass and len(pass_times) < MAX_PASSES and pass_times[-1] < 0.25:
        elapsed, tile_ms, timed_frames = timed_pass()
        pass_times.append(elapsed)
        behind = sum(1 for k in range(len(pass_times)) if sum(pass_times[:k]) >= 0.030)
        converged = sum(pass_times[:-1]) >= 0.030 and abs(pass_times[-1] - pass_times[-2]) <= 0.01 * pass_times[-2]
        if converged and behind >= MIN_BEHIND_RAMP:   # (a median wants more than one pass behind the ramp)
            break
    # the reported pass time: the median of the passes that STARTED behind the clock ramp (>= 30 ms of load in front of
    # them); a run too short to have one reports its last pass
    post_ramp = [t for k, t in enumerate(pass_times) if sum(pass_times[:k]) >= 0.030] or [pass_times[-1]]
    elapsed = float(np.median(post_ramp))
    pass_spread = (max(post_ramp) - min(post_ramp)) / elapsed

    gather_check = None
    if rank == 0 and world > 1 and keep["frames"] is not None:
        # every rank's frames of that step, as they arrived, against rank 0's own blocking renders of the same views
        if step_poses is not None:
            r.set_group_poses(step_poses[keep["step"] % len(step_poses)])
        n_cmp, equal = 0, True
        for src in range(world):
            for k, c_ in enumerate(cameras_of(a.config, src, world, a.views_per_step)):
                own = r.render(c_.viewmat, c_.K, W, H, BG, want=("rgb8",))["rgb8"]
                equal = equal and bool(torch.equal(own.cpu(), keep["frames"][src][k].cpu()))
                n_cmp += 1
        gather_check = {"step": keep["step"], "frames_compared": n_cmp, "bit_equal_to_rank0_renders": equal}
        keep["frames"] = None

    line = None
    if rank == 0:
        # intersections / visible Gaussians of each view of the step (their mean enters the byte counts)
        if step_poses is not None:
            r.set_group_poses(step_poses[0])
        per_view = []
        for c_ in cams:
            r.render(c_.viewmat, c_.K, W, H, BG, want=("rgb8",), out={"rgb8": bufs[0]["rgb8"][0]})
            per_view.append(r.stats())
        st = {k: int(round(np.mean([pv[k] for pv in per_view]))) for k in ("n_visible", "n_isect")}
        out_bpp = (12 if "rgb" in want else 0) + 3
        ms_per_step = elapsed / a.steps * 1e3
        fps = views_all * a.steps / elapsed
        blend_s = max(tile_ms, 1e-9) * 1e-3   # mean duration of k_tile_lazy over the timed region (HIP events on its stream)
        # scenes below 0.5 M Gaussians render a batch in launch groups: one tile launch covers `per_launch` views
        per_launch = 1
        if V >= 2:   # a batch's last frame tells how many views shared its launches
            r.render_batch(Vs, Ks, W, H, BG, want=("rgb8",), out={"rgb8": bufs[0]["rgb8"][:V]})
            per_launch = max(1, r.stats()["launch_views"])
        achieved_co = per_launch * tile_kernel_bytes(st["n_isect"], W, H, out_bpp) / blend_s / 1e9
        # the dominant kernel ALONE on the GPU: ten blocking frames of this rank's first view behind the timed region,
        # HIP events around the kernel on its own stream (SAS_TIMING).  This is the roofline's divisor: under the bench
        # several launches of the kernel and the next frames' binning share the chip, and a launch then lasts longer
        # than a whole step (round-3 verdict: "a divisor that exceeds ms_per_step is not a roofline").
        stage = {}
        if cams:
            c0_ = cams[0]
            for i in range(12):
                r.render(c0_.viewmat, c0_.K, W, H, BG, want=want, out={k: v[0] for k, v in bufs[0].items()}, timing=True)
                if i >= 2:
                    for k, v in r.stage_times().items():
                        stage.setdefault(k, []).append(v)
        iso_ms = float(np.mean(stage["blend"])) if stage else blend_s * 1e3
        achieved = tile_kernel_bytes(st["n_isect"], W, H, out_bpp) / (iso_ms * 1e-3) / 1e9
        # bytes one step of THIS rank moves: one pass over the scene per view pair (config 3 with two views per
        # step: ONE pass for both), plus the per-view terms
        pair = a.config == 3 and a.views_per_step == 2 or (a.config != 3 and scene.n >= 500_000 and V >= 2)
        passes = (V + 1) // 2 if pair else V
        step_bytes = passes * scene_pass_bytes(scene.n) + V * view_bytes(st["n_visible"], st["n_isect"], W, H, out_bpp)
        step_gbps = step_bytes / (elapsed / a.steps) / 1e9
        metric = "rendered frames/sec at 1M Gaussians 1920x1080" if a.config == 3 else \
            f"rendered frames/sec, BASELINE config {a.config} ({scene.n} Gaussians, {W}x{H})"
        line = {
            "metric": metric, "value": fps, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            # 1 (rounds 1-2): value = the first W + K pass from the idle GPU, today's cold_start; 2 (rounds 3-4): the last pass
            "value_protocol_version": 3,
            "converged": bool(converged), "pass_spread": pass_spread, "passes_behind_ramp": len(post_ramp),
            "effective_warmup_steps": a.warmup + (len(pass_times) - len(post_ramp)) * (a.warmup + a.steps),   # untimed + discarded timed steps in front of the first pass that counts
            "protocol": ("one W + K pass from the idle GPU" if a.single_pass else
                         "W warm-up + K timed steps repeated back to back until a pass is within 1 % of the one before and the passes in "
                         "front of it cover the ~30 ms the GPU's clocks ramp for after idle, and three passes lie behind that ramp (<= 12 passes); value = the MEDIAN of the passes "
                         "that start behind the ramp, converged = the last two agreed, cold_start = the first pass"),
            "passes": [{"value": views_all * a.steps / t, "ms_per_step": t / a.steps * 1e3} for t in pass_times],
            "cold_start": {"value": views_all * a.steps / cold_elapsed, "unit": "frames/s", "ms_per_step": cold_elapsed / a.steps * 1e3,
                           "what": "the same W warm-up + K timed steps started on the idle GPU (first ~25 ms: clock ramp)"},
            "config": {"workload": desc, "n_gaussians": scene.n, "n_visible": st["n_visible"], "n_intersections": st["n_isect"],
                       "views_per_step": views_all, "parallelism": f"views{world}x{V}",
                       "gather": "uint8 frames to rank 0 (RCCL), each step as soon as its frames are complete" if world > 1 else "none",
                       "gather_check": gather_check},
            "roofline": {"bound": "hbm", "kernel": "k_tile_lazy", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": pmc_traffic("k_tile_lazy") if a.config == 3 else None,
                         "kernel_ms": iso_ms,
                         "frac_is": "the kernel ALONE on the GPU: algorithmic bytes of one launch / its mean duration over ten blocking "
                                    "frames measured in this run behind the timed region (HIP events on the kernel's stream); the committed "
                                    "rocprofv3 single-frame summary holds the same duration",
                         "isolated_frame_stage_ms": {k: float(np.mean(v)) for k, v in stage.items()},
                         # inside the timed region the kernel shares the chip: not a roofline figure, kept for the record
                         "co_resident": {"kernel_ms": blend_s * 1e3, "achieved": achieved_co, "frac": achieved_co / HBM_PEAK_GBPS,
                                         "kernel_launches_timed": timed_frames, "views_per_kernel_launch": per_launch,
                                         "tile_kernels_in_flight": (V / per_launch) * blend_s * 1e3 / ms_per_step if V else 0.0,
                                         "what": "launch duration INSIDE the timed region (HIP events attached to the launches, every "
                                                 "time_every-th step), where `tile_kernels_in_flight` launches of this kernel plus the next "
                                                 "frames' binning share the chip; the same frames/s occurs with 0.24 ms and with 0.41 ms launches "
                                                 "(DESIGN.md s6)"},
                         "step_algorithmic_bytes": step_bytes, "frame_algorithmic_GBps": step_gbps,
                         "frame_frac": step_gbps / HBM_PEAK_GBPS},
        }
        # The projection (k_project, round 5: a geometry role and a colour role in one launch) against the same roof, isolated, from
        # the same ten blocking frames: SURVEY.md 8d's terms N*236 (scene) + N_vis*44 (record + colour) + keys*8 (what was binned)
        if stage and cams:
            nk = int(round(np.mean([pv.get("n_keys", pv["n_isect"]) for pv in per_view])))
            pb = scene_pass_bytes(scene.n) + st["n_visible"] * 44 + nk * 8
            pms = float(np.mean(stage["project"]))
            line["roofline_projection"] = {"bound": "hbm", "kernel": "k_project", "achieved": pb / (pms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                           "frac": pb / (pms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "kernel_ms": pms, "algorithmic_bytes": pb, "n_keys": nk,
                                           "traffic": pmc_traffic("k_project") if a.config == 3 else None,
                                           "what": "one view's projection alone on the GPU (tail included): N*236 + N_vis*44 + keys*8 bytes / its mean duration"}
        # The tile kernel moves 211 MB in ~135 us: HBM is not what bounds it.  Its roof is instruction issue;
        # instruction counts per launch come from the committed rocprofv3 --pmc summary of this command.
        insts = committed("tile_insts.json") if a.config == 3 else None
        if insts:
            valu, salu = insts["valu_insts_per_launch"], insts["salu_insts_per_launch"]
            line["roofline_issue"] = {
                "bound": "vector-instruction issue", "kernel": "k_tile_lazy", "unit": "G wave-instructions/s",
                "achieved": valu / (iso_ms * 1e-3) / 1e9, "peak": VALU_ISSUE_PEAK / 1e9, "frac": valu / (iso_ms * 1e-3) / VALU_ISSUE_PEAK,
                "valu_insts_per_launch": valu, "salu_insts_per_launch": salu,
                "frac_with_scalar": (valu + salu) / (iso_ms * 1e-3) / VALU_ISSUE_PEAK,   # scalar instructions take issue slots too (one scalar unit per CU)
                "useful_frac": insts.get("composited_pixel_splats", 0) * insts.get("valu_per_composited_pixel_splat", 24) / 64 / (iso_ms * 1e-3) / VALU_ISSUE_PEAK,
                "kernel_ms": iso_ms,
                "source": "profiles/tile_insts.json (SQ_INSTS_VALU / SQ_INSTS_SALU per launch, -DSAS_TUNE_STATS counters)"}

    if rank == 0 and world == 1 and a.config == 3 and not a.no_extras:
        # ---- what the reference's own callers do (not part of `value`) --------------------------------------------
        c0 = cams[0]
        K2 = max(50, a.steps)
        # (1) one view per call, asynchronous
        for rep in range(2):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(K2):
                r.render(c0.viewmat, c0.K, W, H, BG, want=("rgb", "rgb8"), out={k: v[0] for k, v in bufs[i % 4].items()}, block=False)
            r.wait()
            torch.cuda.synchronize(dev)
            dt1 = time.perf_counter() - t0
        line["single_view_async"] = {"value": K2 / dt1, "unit": "frames/s", "what": "one view per sas_render call, SAS_ASYNC, rgb + rgb8"}
        # (2) Door A exactly as GaussianSplat.render times it (nerfstudio_utils.py:163-175): blocking, one view,
        #     rgb + accumulation + depth with the max-depth fill
        o3 = {"rgb": bufs[0]["rgb"][0], "alpha": torch.empty((H, W, 1), device=dev), "depth": torch.empty((H, W, 1), device=dev)}
        for rep in range(2):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(K2):
                r.render(c0.viewmat, c0.K, W, H, BG, want=("rgb", "alpha", "depth"), depth_fill_max=True, out=o3)
            torch.cuda.synchronize(dev)
            dt2 = time.perf_counter() - t0
        line["door_a_sync"] = {"value": K2 / dt2, "unit": "frames/s", "ms_per_frame": dt2 / K2 * 1e3,
                               "what": "blocking single view, rgb + alpha + depth, SAS_DEPTH_FILL_MAX (the reference's call pattern)"}
        # (3) the same output set -- everything GaussianSplat.render returns (nerfstudio_utils.py:123-177: rgb, accumulation,
        #     depth with the max-depth fill; background is a constant) -- PIPELINED: SAS_ASYNC, four output sets in rotation
        o4 = [{"rgb": bufs[k]["rgb"][0], "alpha": torch.empty((H, W, 1), device=dev), "depth": torch.empty((H, W, 1), device=dev)} for k in range(4)]
        for rep in range(2):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(K2):
                r.render(c0.viewmat, c0.K, W, H, BG, want=("rgb", "alpha", "depth"), depth_fill_max=True, out=o4[i % 4], block=False)
            r.wait()
            torch.cuda.synchronize(dev)
            dt3 = time.perf_counter() - t0
        line["door_a_async"] = {"value": K2 / dt3, "unit": "frames/s", "ms_per_frame": dt3 / K2 * 1e3, "frames": K2,
                                "what": "one view per call, SAS_ASYNC, rgb + alpha + depth with SAS_DEPTH_FILL_MAX: every output of Door A, pipelined"}
    if rank == 0 and world == 1 and a.config == 3 and not a.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(scene, cams[0])
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
