"""Scene files: the reference's JSON format, a compact .npz, and the
state-dict layout of a nerfstudio splatfacto checkpoint when one is supplied (both checkpoints in
the reference tree are Git-LFS pointers)."""
from __future__ import annotations

import json
from pathlib import Path
from typing import Dict

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


def load_splatfacto_ckpt(path) -> Dict[str, np.ndarray]:
    """``gauss_params`` of a nerfstudio 1.1.x splatfacto checkpoint (``pipeline`` state dict)."""
    import torch
    p = Path(path)
    if p.stat().st_size < 1024 and p.read_bytes().startswith(b"version https://git-lfs"):
        raise FileNotFoundError(f"{path} is a Git-LFS pointer, not a checkpoint")
    sd = torch.load(p, map_location="cpu", weights_only=False)
    sd = sd.get("pipeline", sd)
    out = {}
    for f in CKPT_FIELDS:
        key = CKPT_PREFIX + f
        if key not in sd:
            raise KeyError(f"{path}: {key} not in checkpoint")
        out[f] = sd[key].detach().float().numpy()
    return out
