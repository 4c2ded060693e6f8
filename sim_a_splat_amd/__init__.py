"""sim_a_splat_amd -- MI355X-native Gaussian-splat rasterizer behind sim_a_splat's render-image calls.

Only what the hot path needs lives here: ``csrc/`` (HIP kernels + C ABI), the ctypes binding,
and the host-side mirror of the reference's two render doors.  Heavy imports are lazy so that
``import sim_a_splat_amd`` works on a machine without a GPU (build / CPU tests).
"""
from __future__ import annotations

__version__ = "0.1.0"

_LAZY = {
    "Rasterizer": ("rasterizer", "Rasterizer"),
    "SasError": ("_capi", "SasError"),
    "GaussianSplat": ("gaussian_splat", "GaussianSplat"),
    "SplatModel": ("gaussian_splat", "SplatModel"),
    "SplatScene": ("scene", "SplatScene"),
    "SplatHandler": ("handler", "SplatHandler"),
    "SplatEnvWrapper": ("env_wrapper", "SplatEnvWrapper"),
    "SplatVecEnv": ("vec_env", "SplatVecEnv"),
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        mod, attr = _LAZY[name]
        return getattr(importlib.import_module(f"{__name__}.{mod}"), attr)
    raise AttributeError(name)
