"""GPU: the vectorised, rank-sharded rollout front end (sim_a_splat_amd/vec_env.py) on the HIP rasterizer -- two ranks on
the one card over gloo, five envs with their own link poses per step, two cameras each (one riding on a link), the uint8
observations gathered to rank 0 through the StepPipeline: every gathered observation equals the oracle's frame for
that env's poses, bit for bit.  north_star: "independent camera views from the Gym env's vectorised rollouts shard ...
with a gather of finished frames"; splat_env_wrapper.py:121-159, examples/demo_pusht_splat.py:54-78."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2])
def test_vec_env_observations_of_every_env_equal_the_oracle(world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "tests" / "tools" / "vec_env_gpu_rank.py"), "5"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    lines = {}
    for p in procs:
        out, err = p.communicate(timeout=420)
        assert p.returncode == 0, err[-3000:]
        d = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
        lines[d["rank"]] = d
    assert lines[0]["frames_checked"] == 5 * 2 * 4 and lines[0]["bit_equal_to_oracle"] and lines[0]["max_visible"] > 300, lines
    if world == 2:
        assert lines[1]["frames_checked"] == 2 * 2 * 4 and lines[1]["bit_equal_to_oracle"], lines
