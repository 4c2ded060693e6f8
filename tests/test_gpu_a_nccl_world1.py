"""GPU: the `nccl` (= RCCL) arm of sim_a_splat_amd.distributed, run for real.  The pool offers one GPU per call, so no
multi-GPU curve can be measured here (say so wherever 1/2/4/8 GPUs are mentioned); what CAN run is a world of ONE rank on
backend "nccl": RCCL is loaded and initialised, and FrameGather / StepPipeline / gather_frames move DEVICE tensors through
its gather -- the calls an 8-GPU run makes -- with the gathered frames bit-equal to direct renders.
SURVEY.md 8e; the frames are the uint8 observations of splat_env_wrapper.py:147-158.
(The rank is a child process, started before this process touches the GPU, like tests/test_gpu_a_multirank.py.)"""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.gpu
def test_rccl_gathers_device_frames_at_world_size_one():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "SAS_DIST_BACKEND"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, str(ROOT / "tests" / "tools" / "nccl_world1_rank.py")], env=env, capture_output=True, text=True, timeout=420)
    assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-3000:])
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["backend"] == "nccl" and line["world"] == 1 and line["device_tensors"]
    assert line["pipeline_bit_equal"] and line["gather_frames_bit_equal"] and line["float_frame_bit_equal"], line
    assert line["scene_broadcast_bit_equal"], line      # the scene replicated by RCCL's broadcast (device tensors) renders the same frame
    assert line["vec_env_bit_equal"], line      # SplatVecEnv's observations gathered through RCCL == its single-process ones
    # ... and its frames never bounce through the host on the way: no upload of a finished frame, one download per step on the root
    assert line["vec_env_device_resident"], line
    print("RCCL", line["rccl_version"], line)
