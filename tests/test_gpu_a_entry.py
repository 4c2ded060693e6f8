"""GPU: the driver's two entry points in ONE process, in the order build() -> smoke().  build() loads the C-ABI library to bind its
symbols; loaded before torch it would bind the system's HIP runtime, and sas_create would find no device once torch had brought its
own (round 5: SAS_ERR_NO_DEVICE in exactly this sequence) -- the binding imports torch first."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.gpu
def test_build_then_smoke_in_one_process():
    p = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke()"], cwd=str(ROOT), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    assert "smoke ok" in p.stdout
