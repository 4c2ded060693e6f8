#!/usr/bin/env python3
"""Headline benchmark: rendered frames/s at 1M Gaussians, 1920x1080 (BASELINE.json config 3).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path (project + SH, bin, per-tile sort, blend) over one batch of two
independent 1080p views per GPU (one sas_render_batch call; the pair shares one projection pass) of
one synthetic 1M-Gaussian scene resident in HBM, float32 RGB + uint8 RGB out per view.  With N > 1
every rank renders its own views of the replicated scene (weak scaling) and the finished
uint8 frames are gathered to rank 0 over RCCL, one step behind the renderer.  Rank 0 prints ONE JSON
line; value = frames/s over all ranks.  The oracle is used only for the `cpu_baseline` leg (rank 0, N = 1, bounded sample).
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
from sim_a_splat_amd.synthetic import NERFSTUDIO_EVAL_BACKGROUND, make_scene, ring_camera  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(n, n_vis, m, w, h, out_bytes_per_pixel=15):
    """N*236 + N_vis*44 + M*60 + W*H*out."""
    return n * 236 + n_vis * 44 + m * 60 + w * h * out_bytes_per_pixel


def tile_kernel_bytes(m, w, h, out_bytes_per_pixel=15):
    """Dominant kernel (k_tile_lazy = per-tile ordering + compositing), SURVEY.md 8d terms:
    8 B key read per intersection + 44 B gather per intersection (id 4 + record 40) + the frame."""
    return m * (8 + 44) + w * h * out_bytes_per_pixel


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc summary (FETCH_SIZE is
    doubled as MI355X_MICROARCH.md prescribes for gfx950); None when no summary is committed."""
    p = ROOT / "profiles" / "hbm_traffic.json"
    if not p.exists():
        return None
    try:
        return json.loads(p.read_text())["kernels"][kernel]["hbm_bytes_per_launch"]
    except (KeyError, ValueError):
        return None


def cpu_baseline(scene, cam, budget_s=20.0):
    import oracle
    threads = oracle.num_threads()
    kw = dict(quats=scene.quats, scales=scene.scales, sh_degree=3, background=NERFSTUDIO_EVAL_BACKGROUND)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--time-every", type=int, default=4, help="steps between tile-kernel timing samples")
    ap.add_argument("--views-per-step", type=int, default=2, choices=(1, 2),
                    help="independent views each GPU renders per step; 2 = a view pair (one projection pass for both)")
    a = ap.parse_args()

    rank, world, local_rank = sdist.init_from_env()
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    dev_index = int(os.environ.get("SAS_FORCE_DEVICE", local_rank))   # rehearsal on a 1-GPU box only
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    W, H = 1920, 1080
    scene = make_scene(a.gaussians, seed=3, log_scale_mean=float(np.log(0.006)))
    # independent views: GPU g renders the ring cameras at yaw 45 g (and 45 g + 180 for the second view of a pair)
    VPS = a.views_per_step
    cams = [ring_camera(W, H, 1000.0, yaw_deg=45.0 * rank + 180.0 * v) for v in range(VPS)]
    cam = cams[0]
    Vs, Ks = np.stack([c.viewmat for c in cams]), np.stack([c.K for c in cams])
    r = Rasterizer(dev)
    r.upload(scene.means, scene.opacities, scene.sh, quats=scene.quats, scales=scene.scales, sh_degree=3)
    # every rank writes the float32 frames (the metric's output) and their uint8 twins; the gather moves
    # the uint8 frames, the format Gym observations are exchanged in (splat_env_wrapper.py:135-137):
    # 6.2 MB instead of 24.9 MB per frame keeps the xGMI transfer shorter than a frame
    bufs = [{"rgb": torch.empty((VPS, H, W, 3), dtype=torch.float32, device=dev),
             "rgb8": torch.empty((VPS, H, W, 3), dtype=torch.uint8, device=dev)} for _ in range(3)]
    gather = sdist.FrameGather(world, rank)

    def step(i, timing):
        out = bufs[i % 3]
        # one C-ABI call per step: the step's views go through the frame slots back to back, a pair of
        # views shares one pass over the scene (sas_render_batch)
        r.render_batch(Vs, Ks, W, H, NERFSTUDIO_EVAL_BACKGROUND, want=("rgb", "rgb8"), out=out, block=False,
                       time_tiles=timing)
        # After the call returns, the current stream is ordered behind every view of step i-1 (C ABI
        # contract): gather those, so the xGMI transfer of step i-1 overlaps the rendering of step i.
        if world > 1 and i > 0:
            gather.start(bufs[(i - 1) % 3]["rgb8"])

    def sync(last_i):
        r.wait()                       # orders the current stream behind every frame
        if world > 1:
            if last_i >= 0:
                gather.start(bufs[last_i % 3]["rgb8"])
            gather.finish()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    for i in range(a.warmup):
        step(i, True)
    sync(a.warmup - 1)
    r.stage_time_means(reset=True)
    t0 = time.perf_counter()
    for i in range(a.steps):
        # SAS_TIME_TILES on every TIME_EVERY-th step: HIP events around the dominant kernel of that
        # step's frames (on the kernel's own stream), the frames keep pipelining
        step(i, i % a.time_every == 0)
    sync(a.steps - 1)
    elapsed = time.perf_counter() - t0
    means, timed_frames = r.stage_time_means(reset=True)
    tile_ms = means["blend"]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # intersections / visible Gaussians of each view of the step (the mean enters the byte counts)
    per_view = []
    for c_ in cams:
        r.render(c_.viewmat, c_.K, W, H, NERFSTUDIO_EVAL_BACKGROUND, want=("rgb",), out={"rgb": bufs[0]["rgb"][0]})
        per_view.append(r.stats())
    st = {k: int(round(np.mean([pv[k] for pv in per_view]))) for k in ("n_visible", "n_isect")}

    # per-stage breakdown of an isolated frame (nothing else on the GPU): 10 frames with events at
    # every stage boundary, after the timed region
    stage = {}
    for i in range(10):
        r.render(cam.viewmat, cam.K, W, H, NERFSTUDIO_EVAL_BACKGROUND, want=("rgb", "rgb8"),
                 out={k: v[0] for k, v in bufs[0].items()}, timing=True)
        for k, v in r.stage_times().items():
            stage.setdefault(k, []).append(v)
    blend_s = tile_ms * 1e-3   # mean duration of k_tile_lazy over the timed region (HIP events on its stream)

    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        fps = world * VPS * a.steps / elapsed
        achieved = tile_kernel_bytes(st["n_isect"], W, H) / blend_s / 1e9
        frame_bytes = algorithmic_bytes(scene.n, st["n_visible"], st["n_isect"], W, H)
        line = {
            "metric": "rendered frames/sec at 1M Gaussians 1920x1080",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE config 3: 1M synthetic Gaussians (seed 3, SH degree 3), 1920x1080, "
                                   f"fx=fy=1000, {VPS} independent view(s) per GPU per step"
                                   + (" (a view pair shares one projection pass)" if VPS == 2 else "")
                                   + ", float32 RGB + uint8 RGB out per view",
                       "n_gaussians": scene.n, "n_visible": st["n_visible"], "n_intersections": st["n_isect"],
                       "views_per_step": world * VPS, "parallelism": f"views{world}x{VPS}",
                       "gather": "uint8 frames to rank 0 (RCCL), one step behind the renderer" if world > 1 else "none"},
            "roofline": {"bound": "hbm", "kernel": "k_tile_lazy", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": pmc_traffic("k_tile_lazy"),
                         "kernel_ms": blend_s * 1e3, "kernel_launches_timed": timed_frames,
                         "kernel_ms_isolated_frame": float(np.mean(stage["blend"])),
                         "frame_algorithmic_GBps": frame_bytes / (elapsed / a.steps / VPS) / 1e9,
                         "frame_frac": frame_bytes / (elapsed / a.steps / VPS) / 1e9 / HBM_PEAK_GBPS,
                         "isolated_frame_stage_ms": {k: float(np.mean(v)) for k, v in stage.items()}},
        }
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(scene, cam)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
