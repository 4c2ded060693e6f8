"""Reading a trained nerfstudio splatfacto RUN without nerfstudio: what ``eval_setup(config_path)`` hands to
``GaussianSplat`` in the reference (sim_a_splat/ns_utils/nerfstudio_utils.py:77-121) -- the model parameters, the
dataset's cameras in the scene frame, the dataparser transform.

nerfstudio 1.1.5 (fork e70d6eee, pixi.lock:4259-4261) is absent from the reference tree and from this image, so
the steps below restate its published behaviour; each is anchored on the reference's call site or on a data file
the reference ships:

* ``config.yml`` is a YAML dump of python objects (``!!python/object:nerfstudio...TrainerConfig``).  It is read with
  a SAFE loader that turns every python tag into plain dicts / tuples / paths: nothing in the file is imported or
  executed (``yaml.Loader``, which nerfstudio uses, would import what the file names).
* checkpoint directory = ``output_dir/experiment_name/method_name/timestamp/relative_model_dir`` relative to the
  working directory (``TrainerConfig.get_checkpoint_dir``), latest ``step-*.ckpt`` (``eval_load_checkpoint``);
  when that does not exist from here, the run directory is the one holding ``config.yml``.
* dataset = ``pipeline.datamanager.data`` (or ``.dataparser.data`` when set): ``transforms.json``.  The
  ``Nerfstudio`` dataparser sorts frames by file name, splits them (``eval_mode: fraction``: equally spaced train
  images, the rest eval), orients ("up") and centres ("poses") the camera-to-world matrices, scales them by
  1 / max |translation| and rescales the intrinsics by 1 / downscale_factor.  The resulting 3x4 transform (times the
  file's ``applied_transform``) and scale are what nerfstudio saved next to the config as
  ``dataparser_transforms.json`` -- reference-held data this module is pinned against
  (tests/test_host_logic.py::test_dataparser_restatement_reproduces_the_shipped_transforms, both scenes).
"""
from __future__ import annotations

import json
import math
from pathlib import Path, PurePosixPath
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

MAX_AUTO_RESOLUTION = 1600   # nerfstudio_dataparser.py: images larger than this are downscaled when images_<k>/ exist


# ---- config.yml -----------------------------------------------------------------------------------------
def _safe_loader():
    import yaml

    class _Loader(yaml.SafeLoader):
        pass

    def construct(loader, suffix, node):
        kind = suffix.split(":", 1)[0]
        name = suffix.split(":", 1)[1] if ":" in suffix else ""
        if isinstance(node, yaml.MappingNode):
            d = loader.construct_mapping(node, deep=True)
            d["__class__"] = name
            return d
        if isinstance(node, yaml.SequenceNode):
            seq = loader.construct_sequence(node, deep=True)
            if kind == "tuple":
                return tuple(seq)
            if name.endswith("Path"):
                return Path(*[str(s) for s in seq]) if seq else Path()
            return seq[0] if len(seq) == 1 and kind == "object/apply" else tuple(seq)   # enums: EventName('...')
        return name or loader.construct_scalar(node)                                    # python/name:pkg.attr ''

    _Loader.add_multi_constructor("tag:yaml.org,2002:python/", construct)
    return _Loader


def load_config(config_path) -> Dict:
    """``config.yml`` of a nerfstudio run as plain data (dicts carry the dumped class in ``__class__``)."""
    import yaml
    with open(config_path, "r") as f:
        cfg = yaml.load(f, Loader=_safe_loader())
    if not isinstance(cfg, dict) or "pipeline" not in cfg:
        raise ValueError(f"{config_path}: not a nerfstudio TrainerConfig dump")
    return cfg


def checkpoint_dir(config_path, cfg: Dict) -> Path:
    """``TrainerConfig.get_checkpoint_dir()`` (relative to the working directory, as the reference's scripts run
    from the repository root); falls back to the directory of ``config.yml``."""
    rel = Path(str(cfg.get("relative_model_dir", "nerfstudio_models")))
    base = Path(f"{cfg.get('output_dir', 'outputs')}/{cfg.get('experiment_name')}/{cfg.get('method_name')}/{cfg.get('timestamp')}")
    for cand in (base / rel, Path(config_path).resolve().parent / rel):
        if cand.is_dir():
            return cand
    raise FileNotFoundError(f"no checkpoint directory {base / rel} (from {Path.cwd()}) nor next to {config_path}")


def latest_checkpoint(ckpt_dir: Path) -> Tuple[Path, int]:
    """``eval_load_checkpoint``: the largest step among ``step-<n>.ckpt``."""
    steps = sorted(int(p.name[p.name.find("-") + 1: p.name.find(".")]) for p in Path(ckpt_dir).glob("step-*.ckpt"))
    if not steps:
        raise FileNotFoundError(f"no step-*.ckpt in {ckpt_dir}")
    return Path(ckpt_dir) / f"step-{steps[-1]:09d}.ckpt", steps[-1]


def data_path(config_path, cfg: Dict) -> Path:
    """Dataset location of the run: the dataparser's ``data`` when set, else the datamanager's, else the trainer's
    (``VanillaDataManager.__init__``); relative paths are tried from the working directory, then from every parent
    of ``config.yml`` (the run usually lives inside the dataset directory: assets/<scene>/splatfacto/<timestamp>/)."""
    dm = cfg["pipeline"]["datamanager"]
    cands = [dm.get("dataparser", {}).get("data"), dm.get("data"), cfg.get("data")]
    data = next((Path(str(c)) for c in cands if c is not None and str(c) not in ("", ".")), None)
    if data is None:
        raise ValueError("the run's config names no dataset path")
    tries = [data] + [par / data for par in Path(config_path).resolve().parents] + \
            [par for par in Path(config_path).resolve().parents if par.name == data.name]
    for t in tries:
        if (t / "transforms.json").exists() or (t.is_file() and t.suffix == ".json"):
            return t
    raise FileNotFoundError(f"dataset {data} (transforms.json) not found from {Path.cwd()} or above {config_path}")


# ---- cameras ---------------------------------------------------------------------------------------------
class Cameras:
    """The part of ``nerfstudio.cameras.cameras.Cameras`` the reference's render path reads
    (nerfstudio_utils.py:95-136): per-camera ``fx fy cx cy`` [C,1] float32, ``width height`` [C,1] int64,
    ``camera_to_worlds`` [C,3,4]; indexing gives a one-camera view with 0-d ``.item()``-able fields."""

    def __init__(self, camera_to_worlds, fx, fy, cx, cy, width, height):
        self.camera_to_worlds = torch.as_tensor(camera_to_worlds, dtype=torch.float32).reshape(-1, 3, 4)
        C = self.camera_to_worlds.shape[0]
        col = lambda v, dt: torch.as_tensor(v, dtype=dt).reshape(-1, 1).expand(C, 1).clone()
        self.fx, self.fy, self.cx, self.cy = (col(v, torch.float32) for v in (fx, fy, cx, cy))
        self.width, self.height = col(width, torch.int64), col(height, torch.int64)

    def __len__(self) -> int:
        return int(self.camera_to_worlds.shape[0])

    @property
    def size(self) -> int:
        return len(self)

    def __getitem__(self, i) -> "Cameras":
        if isinstance(i, int):
            i = slice(i, i + 1) if i != -1 else slice(i, None)
        sub = Cameras.__new__(Cameras)
        for k in ("camera_to_worlds", "fx", "fy", "cx", "cy", "width", "height"):
            setattr(sub, k, getattr(self, k)[i])
        return sub

    def to(self, device) -> "Cameras":
        return self

    def get_intrinsics_matrices(self) -> torch.Tensor:
        K = torch.zeros((len(self), 3, 3), dtype=torch.float32)
        K[:, 0, 0], K[:, 1, 1] = self.fx.squeeze(-1), self.fy.squeeze(-1)
        K[:, 0, 2], K[:, 1, 2] = self.cx.squeeze(-1), self.cy.squeeze(-1)
        K[:, 2, 2] = 1.0
        return K

    def rescale_output_resolution(self, scaling_factor: float, scale_rounding_mode: str = "floor") -> None:
        """nerfstudio's in-place rescale: focal lengths and principal point scaled, sizes floored (its default)."""
        s = float(scaling_factor)
        self.fx, self.fy, self.cx, self.cy = self.fx * s, self.fy * s, self.cx * s, self.cy * s
        if scale_rounding_mode == "floor":
            self.height, self.width = (self.height * s).to(torch.int64), (self.width * s).to(torch.int64)
        elif scale_rounding_mode == "round":
            self.height, self.width = torch.floor(0.5 + self.height * s).to(torch.int64), torch.floor(0.5 + self.width * s).to(torch.int64)
        elif scale_rounding_mode == "ceil":
            self.height, self.width = torch.ceil(self.height * s).to(torch.int64), torch.ceil(self.width * s).to(torch.int64)
        else:
            raise ValueError("Scale rounding mode must be 'floor', 'round' or 'ceil'.")


# ---- Nerfstudio dataparser (pose / intrinsics path only) ---------------------------------------------------
def rotation_matrix(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """nerfstudio camera_utils.rotation_matrix: the rotation taking unit vector a to unit vector b (Rodrigues)."""
    a = a / torch.linalg.norm(a)
    b = b / torch.linalg.norm(b)
    v = torch.linalg.cross(a, b)
    eps = 1e-6
    if torch.sum(torch.abs(v)) < eps:
        x = torch.tensor([1.0, 0, 0]) if abs(a[0]) < eps else torch.tensor([0, 1.0, 0])
        v = torch.linalg.cross(a, x)
    v = v / torch.linalg.norm(v)
    skew = torch.tensor([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=torch.float32)
    theta = torch.acos(torch.clip(torch.dot(a, b), -1, 1))
    return torch.eye(3) + torch.sin(theta) * skew + (1 - torch.cos(theta)) * (skew @ skew)


def auto_orient_and_center_poses(poses: torch.Tensor, method: str = "up", center_method: str = "poses"):
    """camera_utils.auto_orient_and_center_poses for the methods the shipped runs use ("up" / "none" orientation,
    "poses" / "none" centring); returns (oriented poses [N,3,4], transform [3,4])."""
    origins = poses[..., :3, 3]
    mean_origin = torch.mean(origins, dim=0)
    if center_method == "poses":
        translation = mean_origin
    elif center_method == "none":
        translation = torch.zeros_like(mean_origin)
    else:
        raise NotImplementedError(f"center_method {center_method!r} (the shipped runs use 'poses')")
    if method == "up":
        up = torch.mean(poses[:, :3, 1], dim=0)
        up = up / torch.linalg.norm(up)
        rotation = rotation_matrix(up, torch.tensor([0.0, 0.0, 1.0]))
        transform = torch.cat([rotation, rotation @ -translation[..., None]], dim=-1)
        oriented = transform @ poses
    elif method == "none":
        transform = torch.eye(4)[:3]
        transform[:3, 3] = -translation
        oriented = transform @ poses
    else:
        raise NotImplementedError(f"orientation_method {method!r} (the shipped runs use 'up')")
    return oriented, transform


def split_indices(filenames: Sequence[str], split: str, eval_mode: str, train_split_fraction: float, eval_interval: int):
    """Train / eval image indices (nerfstudio data/utils/dataparsers_utils.py)."""
    n = len(filenames)
    i_all = np.arange(n)
    if eval_mode == "fraction":
        n_train = math.ceil(n * train_split_fraction)
        i_train = np.linspace(0, n - 1, n_train, dtype=int)
        i_eval = np.setdiff1d(i_all, i_train)
    elif eval_mode == "interval":
        i_train = i_all[i_all % eval_interval != 0]
        i_eval = i_all[i_all % eval_interval == 0]
    elif eval_mode == "filename":
        base = [PurePosixPath(f).name for f in filenames]
        i_train = np.array([i for i, b in enumerate(base) if "train" in b], dtype=int)
        i_eval = np.array([i for i, b in enumerate(base) if "eval" in b], dtype=int)
    elif eval_mode == "all":
        i_train = i_eval = i_all
    else:
        raise ValueError(f"Unknown eval mode {eval_mode}")
    if split == "train":
        return i_train
    if split in ("val", "test"):
        return i_eval
    raise ValueError(f"Unknown dataparser split {split}")


def _downscale_factor(data_dir: Path, first_file: str, configured) -> int:
    if configured is not None:
        return int(configured)
    img = data_dir / first_file
    if not img.exists():
        return 1            # images are not shipped with the run: intrinsics as stored
    try:
        from PIL import Image
        with Image.open(img) as im:
            max_res = max(im.size)
    except Exception:
        return 1
    df = 0
    while True:
        if (max_res / 2 ** df) <= MAX_AUTO_RESOLUTION:
            break
        if not (data_dir / f"images_{2 ** (df + 1)}" / PurePosixPath(first_file).name).exists():
            break
        df += 1
    return 2 ** df


def dataparser_outputs(data, dp_cfg: Optional[Dict] = None, split: str = "train") -> Dict:
    """The ``Nerfstudio`` dataparser's cameras for ``split`` plus its transform and scale.

    ``data``: dataset directory (with ``transforms.json``) or the json itself, or an already-loaded dict of its
    content.  ``dp_cfg``: the ``dataparser`` mapping of config.yml (defaults = nerfstudio's)."""
    dp = dict(orientation_method="up", center_method="poses", auto_scale_poses=True, scale_factor=1.0, downscale_factor=None,
              eval_mode="fraction", train_split_fraction=0.9, eval_interval=8)
    dp.update({k: v for k, v in (dp_cfg or {}).items() if k in dp})
    if isinstance(data, dict):
        meta, data_dir = data, Path(".")
    else:
        data = Path(data)
        meta_file = data if data.suffix == ".json" else data / "transforms.json"
        data_dir = meta_file.parent
        with open(meta_file, "r") as f:
            meta = json.load(f)
    frames = meta["frames"]
    order = sorted(range(len(frames)), key=lambda i: PurePosixPath(frames[i]["file_path"]).parts)   # np.argsort of the Paths
    frames = [frames[i] for i in order]
    names = [fr["file_path"] for fr in frames]

    def per_frame(key_meta, key_frame=None):
        if key_meta in meta:
            return [float(meta[key_meta])] * len(frames)
        return [float(fr[key_frame or key_meta]) for fr in frames]
    fx, fy, cx, cy = per_frame("fl_x"), per_frame("fl_y"), per_frame("cx"), per_frame("cy")
    height = [int(v) for v in per_frame("h")]
    width = [int(v) for v in per_frame("w")]
    poses = torch.from_numpy(np.array([fr["transform_matrix"] for fr in frames]).astype(np.float32))

    explicit = [f"{s}_filenames" in meta for s in ("train", "val", "test")]
    if any(explicit):
        want = set(meta.get(f"{split}_filenames", []))
        idx = np.array([i for i, nme in enumerate(names) if nme in want], dtype=int)
    else:
        idx = split_indices(names, split, dp["eval_mode"], dp["train_split_fraction"], dp["eval_interval"])

    poses, transform = auto_orient_and_center_poses(poses, method=dp["orientation_method"], center_method=dp["center_method"])
    scale = 1.0
    if dp["auto_scale_poses"]:
        scale /= float(torch.max(torch.abs(poses[:, :3, 3])))
    scale *= float(dp["scale_factor"])
    poses[:, :3, 3] *= scale

    it = torch.as_tensor(idx, dtype=torch.long)
    sel = lambda v, dt: torch.tensor(v, dtype=dt)[it]
    cams = Cameras(poses[it, :3, :4], sel(fx, torch.float32), sel(fy, torch.float32), sel(cx, torch.float32), sel(cy, torch.float32),
                   sel(width, torch.int64), sel(height, torch.int64))
    down = _downscale_factor(data_dir, names[0], dp["downscale_factor"])
    cams.rescale_output_resolution(1.0 / down)

    # what nerfstudio saves as dataparser_transforms.json: the transform above composed with the file's applied_transform
    saved = transform
    if "applied_transform" in meta:
        A = torch.tensor(meta["applied_transform"], dtype=transform.dtype)
        saved = transform @ torch.cat([A, torch.tensor([[0, 0, 0, 1]], dtype=transform.dtype)], 0)
    return dict(cameras=cams, image_filenames=[names[i] for i in idx], transform=transform, scale=float(scale),
                dataparser_transform=saved, downscale_factor=down, indices=idx)


# ---- the run ---------------------------------------------------------------------------------------------
BACKGROUNDS = {"random": (0.1490, 0.1647, 0.2157), "black": (0.0, 0.0, 0.0), "white": (1.0, 1.0, 1.0)}   # SplatfactoModel eval


def eval_setup(config_path, test_mode: str = "inference") -> Dict:
    """Counterpart of ``nerfstudio.utils.eval_utils.eval_setup`` for a splatfacto run, as plain data:
    ``config`` (dict), ``gauss_params`` (numpy, raw), ``step``, ``sh_degree`` in use, ``background``,
    ``train`` / ``eval`` dataparser outputs (the eval split is "test" for test_mode "test"/"inference", else "val":
    the same frames either way for the nerfstudio dataparser)."""
    from . import io
    config_path = Path(config_path)
    cfg = load_config(config_path)
    model_cfg = cfg["pipeline"]["model"]
    if "splatfacto" not in str(model_cfg.get("__class__", "")).lower() and "splatfacto" not in str(cfg.get("method_name", "")):
        raise NotImplementedError(f"{config_path}: method {cfg.get('method_name')!r}; only splatfacto runs have a render path here")
    ckpt, step = latest_checkpoint(checkpoint_dir(config_path, cfg))
    sd_step, gauss = io.load_splatfacto_ckpt(ckpt, with_step=True)
    step = sd_step if sd_step is not None else step
    dp_cfg = cfg["pipeline"]["datamanager"].get("dataparser", {})
    data = data_path(config_path, cfg)
    eval_split = "test" if test_mode in ("test", "inference") else "val"
    sh_max = int(model_cfg.get("sh_degree", 3))
    interval = int(model_cfg.get("sh_degree_interval", 1000))
    bg = model_cfg.get("background_color", "random")
    return dict(config=cfg, config_path=config_path, checkpoint=ckpt, step=int(step), gauss_params=gauss,
                # SplatfactoModel.load_state_dict sets self.step = 30000, whatever the trainer's step was: get_outputs then
                # evaluates min(30000 // sh_degree_interval, sh_degree) bands
                sh_degree=min(30000 // interval, sh_max) if sh_max > 0 else 0,
                background=BACKGROUNDS.get(bg, BACKGROUNDS["random"]) if isinstance(bg, str) else tuple(bg),
                train=dataparser_outputs(data, dp_cfg, "train"), eval=dataparser_outputs(data, dp_cfg, eval_split),
                data=data)
