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
        raise RuntimeError(f"season_nerf_amd: custom-op library not built ({OPS_PATH} missing). Run `python season_nerf_amd/build.py`.")
    ctypes.CDLL(_lib.LIB_PATH, mode=ctypes.RTLD_GLOBAL)      # the C ABI the op layer links against
    torch.ops.load_library(OPS_PATH)
    _register_fakes()
    _loaded = True
    return torch.ops.season_nerf


def _register_fakes():
    """Shape / dtype propagation ("fake" kernels) for the ops whose arguments are plain tensors, so that FakeTensor tracing
    (torch.compile, torch.library.opcheck) sees through them.  The ops that take the packed model (a torchbind object) are
    dispatcher-visible but opaque to tracing."""
    reg = torch.library.register_fake

    @reg("season_nerf::composite")
    def _(top, bot, tvals, rho, col, solar_vis, sky, flags, rho_prior, trust):
        R, S = top.shape[0], tvals.numel()
        e = top.new_empty
        return [e(R, 3), e(R, 3), e(R, S, 1), e(R, S, 1), e(R, S, 1), e(R, S, 1), e(R), e(R), e(R, 3), e(R)]

    @reg("season_nerf::composite_sweep")
    def _(top, bot, tvals, rho, col_raw, adjust, solar_vis, sky, class_vecs, flags, classic):
        R, T = top.shape[0], class_vecs.shape[0]
        e = top.new_empty
        return [e(T, R, 3), e(T, R, 3), e(R, 3), e(R, 3), e(R), e(T if classic else 0, R, 3)]

    @reg("season_nerf::fused_adam_")
    def _(param, grad, m, v, lr, beta1, beta2, eps, step):
        return None

    @reg("season_nerf::trainer_adam_step_")
    def _(trainer, params, grads, lr, beta1, beta2, eps, step):
        return None

    @reg("season_nerf::trainer_adam_step_dev_")
    def _(trainer, params, grads, hyper):
        return None

    @reg("season_nerf::trainer_zero_grad_")
    def _(trainer, grads):
        return None

    @reg("season_nerf::prior_density")
    def _(pts, delta, height_map, outside):
        return pts.new_empty(pts.shape[0], 1)

    @reg("season_nerf::loss_scratch_prepare")
    def _(like):
        return None

    @reg("season_nerf::loss_terms")
    def _(rgb, gt, albedo, sky, solar_vis, pv_exact, pe, albedo_min_global, world):
        return rgb.new_empty(5), rgb.new_empty(6)

    @reg("season_nerf::loss_terms_bwd")
    def _(g_vals, rgb, gt, albedo, sky, solar_vis, pv_exact, min, world):
        return torch.empty_like(rgb), torch.empty_like(albedo), torch.empty_like(sky), torch.empty_like(solar_vis)

    @reg("season_nerf::train_fwd_image")
    def _(trainer, top, bot, tvals, sun, time, train_bn, classic, n_classes, height_map, trust, trust_dev, params):
        R, S, C = top.shape[0], tvals.numel(), n_classes
        e = top.new_empty
        m3 = (R, 3) if height_map is not None else (0,)
        r = [e(R, 3), e(R, 3), e(R, 3), e(R, S, 1), e(*m3), e(*m3), e(R, S, 1), e(R, S, 1), e(R, S, 1), e(R, C), e(R, S, 1), e(R, S, 1), e(R, S, 3),
             e(R, S, 3), e(R, S, 3)]
        return r + ([e(R, S, 1) for _ in range(8)] if height_map is not None else [])

    @reg("season_nerf::train_fwd_points")
    def _(trainer, x, sun, time, train_bn, n_classes, params):
        N, C, e = x.shape[0], n_classes, x.new_empty
        return [e(N, 1), e(N, 3), e(N, 1), e(N, 3), e(N, C), e(N, 3), e(N, 3), e(N, C, 3)]

    @reg("season_nerf::train_fwd_solar")
    def _(trainer, top, bot, tvals, sun, train_bn, params):
        R, S, e = top.shape[0], tvals.numel(), top.new_empty
        return [e(R, S, 1), e(R, S, 1), e(R, S, 1), e(R, 3), e(R, S, 1), e(R, S, 3), e(R, S, 1)]


def model_view(handle):
    """torch.classes.season_nerf.Model viewing (not owning) a C-ABI model created elsewhere, e.g. by a C host."""
    return load().model_from_handle(int(handle))
