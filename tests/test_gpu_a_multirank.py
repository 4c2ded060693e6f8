"""GPU: bench.py's multi-rank path with REAL frames on a one-GPU box -- two ranks (child processes started by bench.py
itself) share the card, torch.distributed over gloo (RCCL needs one GPU per rank), the uint8 frames of every step gathered
to rank 0 by the StepPipeline; rank 0 then compares what arrived for the last timed step with its own renders of every
rank's views.  SURVEY.md 8e / BASELINE config 4: eight Gym poses sharded over the ranks, link poses updated every step.
(The file sorts in front of the other GPU tests on purpose: the ranks are started before this process has touched the GPU;
it also passes behind them.)"""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.gpu
def test_bench_two_ranks_gather_real_frames_over_gloo():
    env = dict(os.environ, SAS_DIST_BACKEND="gloo", SAS_FORCE_DEVICE="0", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--config", "4", "--steps", "6", "--warmup", "2",
                        "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True, timeout=420)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    chk = line["config"]["gather_check"]
    assert chk == {"step": 5, "frames_compared": 8, "bit_equal_to_rank0_renders": True}
