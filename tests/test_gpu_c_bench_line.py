"""GPU: the ONE JSON line of `python bench.py --gpus 1 --steps K --warmup W` carries what the driver's contract and the
hot-path tier ask for: the contract keys, `roofline` (the dominant kernel alone on the GPU against HBM, with the PMC traffic
of the committed profile) and `cpu_baseline` (the C oracle timed on the box's host cores in the same run), plus this
repository's own disclosures (cold_start, passes, effective_warmup_steps, the co-resident kernel duration)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.gpu
def test_bench_line_schema_at_the_drivers_command_shape():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "SAS_DIST_BACKEND", "SAS_FORCE_DEVICE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                              # ONE line
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic" and d["unit"] == "frames/s"
    assert "1M Gaussians 1920x1080" in d["metric"] and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 1000 and abs(d["value"] - 2 * 6 / (d["ms_per_step"] * 6e-3)) < 1e-6 * d["value"]      # whole-job frames over the timed steps
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["kernel"] == "k_tile_lazy"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.05 < r["frac"] < 1.0
    assert r["traffic"] is None or r["traffic"] > 1e7                   # HBM bytes per launch from the committed PMC passes
    assert r["kernel_ms"] < d["ms_per_step"]                            # the roofline's divisor is the kernel alone: shorter than a step
    assert r["co_resident"]["kernel_ms"] >= r["kernel_ms"] * 0.9        # ... the launch inside the timed region shares the chip
    rp = d["roofline_projection"]                                       # the second kernel of a frame, an HBM stream, against the same roof
    assert rp["kernel"] == "k_project" and rp["bound"] == "hbm" and abs(rp["frac"] - rp["achieved"] / rp["peak"]) < 1e-9 and 0.05 < rp["frac"] < 1.0   # (the bounds-checked build runs this test too: slower kernels)
    assert rp["n_keys"] > 0 and 0.0 < rp["kernel_ms"] < 2.0 * d["ms_per_step"]   # (n_keys: what was binned -- fewer than the rectangles' intersections with exact tile culling, more when the frame is binned in 8-pixel tiles)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "frames/s" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert d["value"] / c["value"] > 100                                # reported beside, not the target
    assert d["value_protocol_version"] == 3 and d["effective_warmup_steps"] >= 2 and len(d["passes"]) >= 1
    assert abs(d["cold_start"]["value"] - d["passes"][0]["value"]) < 1e-6 * d["value"]
    # value = the median of the passes behind the clock ramp (not whichever pass the cap fell on), with its own disclosures
    assert isinstance(d["converged"], bool) and d["pass_spread"] >= 0.0 and 1 <= d["passes_behind_ramp"] <= len(d["passes"])
    behind = sorted(p_["value"] for p_ in d["passes"][len(d["passes"]) - d["passes_behind_ramp"]:])
    assert behind[0] * (1 - 1e-9) <= d["value"] <= behind[-1] * (1 + 1e-9)
    assert d["door_a_sync"]["value"] > 1000 and d["single_view_async"]["value"] > 1000
    # every output of Door A, pipelined (rgb + accumulation + depth with the fill): not slower than the blocking form
    assert d["door_a_async"]["value"] >= 0.95 * d["door_a_sync"]["value"] and d["door_a_async"]["frames"] >= 50
    assert "door_a_async" in d["config"]["workload"]                    # the line says what `value` does not write
