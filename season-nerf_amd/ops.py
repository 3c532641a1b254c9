"""torch.ops.season_nerf.* - the PyTorch-ROCm custom-op layer (csrc/ops.cpp, TORCH_LIBRARY(season_nerf)) over the C ABI.

`load()` registers the ops (idempotent).  There is no fallback: a missing library raises."""
import ctypes
import os

import torch

from . import _lib

OPS_PATH = os.path.join(_lib.HERE, "libseason_nerf_ops.so")
_loaded = False


def load():
    global _loaded
    if _loaded:
        return torch.ops.season_nerf
    if not os.path.exists(OPS_PATH):
        raise RuntimeError(f"season_nerf_amd: custom-op library not built ({OPS_PATH} missing). Run `python season-nerf_amd/build.py`.")
    ctypes.CDLL(_lib.LIB_PATH, mode=ctypes.RTLD_GLOBAL)      # the C ABI the op layer links against
    torch.ops.load_library(OPS_PATH)
    _loaded = True
    return torch.ops.season_nerf


def model_view(handle):
    """torch.classes.season_nerf.Model viewing a C-ABI model the ctypes binding owns (network.T_NeRF.device_model())."""
    return load().model_from_handle(int(handle))
