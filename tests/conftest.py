import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def rasterizer():
    """One HIP context for the whole GPU session (fails loudly when the library is missing)."""
    import torch
    from sim_a_splat_amd.rasterizer import Rasterizer
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    r = Rasterizer(0)
    yield r
    r.close()


TWIN_CASES = ["one", "two", "n64", "n2k", "n2k_groups", "dense", "doorb", "inside"]


def load_twin_fixture(name):
    """tests/golden/render_twin_<name>.npz (oracle/make_golden.py): inputs + float64-twin frames."""
    import numpy as np
    return np.load(GOLDEN / f"render_twin_{name}.npz")


def twin_scene_kwargs(g):
    """Scene arguments of a twin fixture in the form both oracle.render and Rasterizer.upload take:
    (means, opacities, colors, kwargs) with quats+scales or cov6, sh_degree (< 0: final RGB), groups."""
    kw = dict(sh_degree=int(g["sh_degree"]))
    if g["cov6"].size:
        kw["cov6"] = g["cov6"]
    else:
        kw.update(quats=g["quats"], scales=g["scales"])
    if g["group_id"].size:
        kw["group_id"] = g["group_id"]
    return g["means"], g["opacities"], g["colors"], kw


# ---- stand-ins for the handful of viser attributes ViserBridge uses (viser is absent from this image) ----
class FakeViserCamera:
    def __init__(self):
        import numpy as np
        self.wxyz, self.position, self.fov, self.aspect = np.array([0.0, 1.0, 0, 0]), np.array([0.0, 0, 3.0]), 1.2, 4 / 3
        self._cbs = []

    def on_update(self, cb):
        self._cbs.append(cb)

    def move(self, position):
        import numpy as np
        self.position = np.asarray(position, float)
        for cb in self._cbs:
            cb(self)


class FakeViserClient:
    def __init__(self, cid):
        import types
        self.client_id, self.camera, self.images = cid, FakeViserCamera(), []
        self.scene = types.SimpleNamespace(set_background_image=lambda img, **kw: self.images.append((img, kw)))


class FakeViserServer:
    def __init__(self):
        self.connect, self.disconnect = [], []

    def on_client_connect(self, cb):
        self.connect.append(cb)

    def on_client_disconnect(self, cb):
        self.disconnect.append(cb)
