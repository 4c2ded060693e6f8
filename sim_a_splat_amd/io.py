"""Scene files: the reference's JSON format, a compact .npz, the state-dict
layout of a nerfstudio splatfacto checkpoint when one is supplied (both checkpoints in the
reference tree are Git-LFS pointers), the dataparser transform that maps world to the
nerfstudio-normalised scene frame, and the segmentation products of ``match_splat.py``
(per-link masks, ICP transform, mask-time joint configuration)."""
from __future__ import annotations

import io as _io
import json
import pickle
from pathlib import Path
from typing import Dict, Tuple

import numpy as np

JSON_KEYS = ("means", "rotations", "colors", "opacities", "scalings")   # splat_utils.py:56


def load_json(path) -> Dict[str, np.ndarray]:
    """Pre-activation arrays of ``GSplatLoader.load_gsplat_from_json`` (splat_utils.py:51-89)."""
    with open(path, "r") as f:
        data = json.load(f)
    missing = [k for k in JSON_KEYS if k not in data]
    if missing:
        raise KeyError(f"{path}: missing keys {missing}")
    return {k: np.asarray(data[k], dtype=np.float32) for k in JSON_KEYS}


def save_npz(path, **arrays) -> None:
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})


def load_npz(path) -> Dict[str, np.ndarray]:
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


CKPT_PREFIX = "_model.gauss_params."
CKPT_FIELDS = ("means", "scales", "quats", "features_dc", "features_rest", "opacities")


def load_splatfacto_ckpt(path, trust_pickle: bool = False, with_step: bool = False):
    """``gauss_params`` of a nerfstudio 1.1.x splatfacto checkpoint (``pipeline`` state dict; a ``module.`` prefix
    left by DistributedDataParallel is dropped, as ``Pipeline.load_pipeline`` does).

    Loaded with ``weights_only=True`` (tensors and primitives only: nothing in the file is executed).
    ``trust_pickle=True`` is the explicit opt-in for a checkpoint that needs the full unpickler.
    ``with_step=True`` returns ``(step or None, params)``: the trainer stores its step beside the pipeline."""
    import torch
    p = Path(path)
    if p.stat().st_size < 1024 and p.read_bytes().startswith(b"version https://git-lfs"):
        raise FileNotFoundError(f"{path} is a Git-LFS pointer, not a checkpoint")
    loaded = torch.load(p, map_location="cpu", weights_only=not trust_pickle)
    sd = loaded.get("pipeline", loaded)
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
    out = {}
    for f in CKPT_FIELDS:
        key = CKPT_PREFIX + f
        if key not in sd:
            raise KeyError(f"{path}: {key} not in checkpoint")
        out[f] = sd[key].detach().float().numpy()
    if with_step:
        step = loaded.get("step") if isinstance(loaded, dict) else None
        return (int(step) if step is not None else None), out
    return out


# ---- dataparser transform ---------------------------------------------------------------------------
def load_dataparser_transforms(path) -> Tuple[np.ndarray, float]:
    """``dataparser_transforms.json`` next to a splatfacto config: (transform [3,4] f32, scale).
    A world point p maps to the scene frame as ``scale * (transform[:, :3] @ p + transform[:, 3])``
    (nerfstudio convention; read by ``eval_setup`` for the pose handed to
    ``GaussianSplat.render``, nerfstudio_utils.py:123-136)."""
    with open(path, "r") as f:
        d = json.load(f)
    T = np.asarray(d["transform"], dtype=np.float32)
    if T.shape != (3, 4):
        raise ValueError(f"{path}: transform must be 3x4, got {T.shape}")
    return T, float(d["scale"])


def world_to_scene_pose(c2w_world, transform, scale: float) -> np.ndarray:
    """Camera-to-world pose in metric world coordinates -> pose in the scene frame ([3,4])."""
    c2w = np.asarray(c2w_world, dtype=np.float64)[:3, :4]
    T = np.asarray(transform, dtype=np.float64)
    out = T[:, :3] @ c2w
    out[:, 3] = (out[:, 3] + T[:, 3]) * scale
    return out.astype(np.float32)


# ---- segmentation products (splat_handler.py:62-83) -------------------------------------------------
class _NumpyOnlyUnpickler(pickle.Unpickler):
    """``link_masks_global_dict.npy`` is a pickled dict of bool arrays (an object-dtype .npy).
    ``np.load(allow_pickle=True)`` would run any pickle; this reader only resolves the three
    globals numpy's own array pickles use and refuses everything else."""
    _ALLOWED = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
                ("numpy", "ndarray"), ("numpy", "dtype")}

    def find_class(self, module, name):
        if (module, name) not in self._ALLOWED:
            raise pickle.UnpicklingError(f"refusing to load {module}.{name} from a mask file")
        return super().find_class(module, name)


def load_link_masks(path) -> Dict[str, np.ndarray]:
    """Per-link boolean masks ``{"link0": bool[N], ...}`` (splat_handler.py:63-66,121-143).
    Accepts the reference's pickled ``.npy`` (restricted unpickler) or the pickle-free ``.npz``
    written by ``save_link_masks``."""
    p = Path(path)
    if p.suffix == ".npz":
        with np.load(p) as z:
            n = int(z["n"])
            return {k[5:]: np.unpackbits(z[k], count=n).astype(bool) for k in z.files if k.startswith("bits_")}
    with open(p, "rb") as f:
        version = np.lib.format.read_magic(f)
        header = np.lib.format.read_array_header_1_0 if version == (1, 0) else np.lib.format.read_array_header_2_0
        shape, _, dtype = header(f)
        if dtype != np.dtype(object) or shape != ():
            raise ValueError(f"{path}: expected a 0-d object array holding a dict")
        obj = _NumpyOnlyUnpickler(_io.BytesIO(f.read())).load()
    if isinstance(obj, np.ndarray) and obj.shape == ():   # numpy pickles the 0-d object array itself
        obj = obj.item()
    if not isinstance(obj, dict):
        raise ValueError(f"{path}: payload is {type(obj).__name__}, not a dict")
    masks = {str(k): np.asarray(v, dtype=bool) for k, v in obj.items()}
    lens = {m.shape for m in masks.values()}
    if len(lens) > 1 or any(m.ndim != 1 for m in masks.values()):
        raise ValueError(f"{path}: masks must be 1-d and of one length, got {sorted(lens)}")
    return masks


def save_link_masks(path, masks: Dict[str, np.ndarray]) -> None:
    """Pickle-free, bit-packed rewrite of the mask dict (8x smaller, loadable with allow_pickle=False)."""
    n = len(next(iter(masks.values()))) if masks else 0
    np.savez_compressed(path, n=np.int64(n), **{f"bits_{k}": np.packbits(np.asarray(v, dtype=bool)) for k, v in masks.items()})


def load_icp_transformation(path) -> np.ndarray:
    """``icp_transformation.npy``: 4x4 similarity (scale * R | t) from splat to robot frame."""
    T = np.load(path, allow_pickle=False)
    if T.shape != (4, 4):
        raise ValueError(f"{path}: expected 4x4, got {T.shape}")
    return np.asarray(T, dtype=np.float64)


def load_joint_config(path) -> np.ndarray:
    """``joint_config.npy``: joint positions at which the masks were segmented (splat_handler.py:162)."""
    return np.asarray(np.load(path, allow_pickle=False), dtype=np.float64).reshape(-1)
