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


# ---- the reference's trained runs, materialised from their shipped data files -----------------------------------
RUN_SCENES = {"divar113vhw": "divar113vhw", "robots-scene-v2": "xarm6-1"}   # scene -> its masks directory


def fabricated_gauss_params(n, seed, spread=0.35, sh_rest=15):
    """Raw (pre-activation) splatfacto parameters of a stand-in scene around the origin of the scene frame."""
    import numpy as np
    rng = np.random.default_rng(seed)
    return dict(means=rng.normal(0.0, spread, size=(n, 3)).astype(np.float32),
                scales=np.clip(rng.normal(np.log(0.02), 0.5, size=(n, 3)), np.log(2e-3), np.log(0.15)).astype(np.float32),
                quats=rng.normal(size=(n, 4)).astype(np.float32),
                features_dc=rng.normal(0.0, 1.0, size=(n, 3)).astype(np.float32),
                features_rest=rng.normal(0.0, 0.1, size=(n, sh_rest, 3)).astype(np.float32),
                opacities=rng.normal(0.5, 2.0, size=(n, 1)).astype(np.float32))


def materialize_run(scene, root, gauss, step=29999):
    """The reference's directory layout under ``root`` (= its repository root): ``assets/<scene>/transforms.json``,
    ``assets/<scene>/splatfacto/<timestamp>/{config.yml, dataparser_transforms.json, nerfstudio_models/step-*.ckpt}``
    from tests/golden/ns_run_<scene>.npz (the reference's own files, byte for byte) plus a FABRICATED checkpoint in
    nerfstudio's state-dict layout (the real ones are Git-LFS pointers).  Returns the config.yml path."""
    import numpy as np
    import torch
    from pathlib import Path
    with np.load(GOLDEN / f"ns_run_{scene}.npz") as z:
        files = {k: z[k].tobytes() for k in ("config_yml", "dataparser_transforms_json", "transforms_json")}
        ts = str(z["timestamp"])
    base = Path(root) / "assets" / scene
    run = base / "splatfacto" / ts
    (run / "nerfstudio_models").mkdir(parents=True, exist_ok=True)
    (base / "transforms.json").write_bytes(files["transforms_json"])
    (run / "config.yml").write_bytes(files["config_yml"])
    (run / "dataparser_transforms.json").write_bytes(files["dataparser_transforms_json"])
    sd = {"step": step, "pipeline": {f"_model.gauss_params.{k}": torch.from_numpy(np.asarray(v, np.float32)) for k, v in gauss.items()}}
    torch.save(sd, run / "nerfstudio_models" / f"step-{step:09d}.ckpt")
    return run / "config.yml"


def real_link_masks(scene):
    """(masks dict link0.., icp 4x4, n) of the reference's shipped segmentation for ``scene``."""
    import numpy as np
    name = {"divar113vhw": "scene_assets_divar113vhw.npz", "robots-scene-v2": "scene_assets_xarm6_1.npz"}[scene]
    with np.load(GOLDEN / name) as z:
        n = int(z["n"])
        masks = {str(k): np.unpackbits(b, count=n).astype(bool) for k, b in zip(z["link_names"], z["mask_bits"])}
        return masks, z["icp_transformation"].copy(), n

