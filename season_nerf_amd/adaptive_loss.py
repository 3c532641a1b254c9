"""Barron's adaptive robust loss as the reference consumes it (`robust_loss_pytorch.adaptive.AdaptiveLossFunction`,
imported at T_NeRF_Full_2/Net_Tool_2.py:8, built at :69-82, called at Eval_Tools_2.py:426-442 through `.lossfun(x)`,
`.alpha()`, `.scale()` and `.parameters()`).

PARITY UNPINNED.  robust_loss_pytorch is an un-vendored, un-pinned third-party dependency of the reference (README.md:26)
that is not installable in this image; this module restates the published definition (Barron, "A General and Adaptive
Robust Loss Function", CVPR 2019, eqs. 1, 16, 17) and is pinned only by closed-form known answers
(tests/test_adaptive_loss.py):   alpha = 2: 1/2 (x/c)^2 + log c + 1/2 log 2 pi;    alpha = 0: log(1/2 (x/c)^2 + 1) + log c + log(pi sqrt 2).
The log-partition function log Z(alpha) = log int exp(-rho(x, alpha, 1)) dx is tabulated once by quadrature and interpolated
with a cubic Hermite spline (the package ships a pre-fitted spline of the same integral); the gradient of the loss with
respect to the network does not depend on Z at all.

The loss sees R x 3 residuals per step - it is host-side plumbing in torch ops on the device, not a hot kernel.
"""
from __future__ import annotations

import math

import numpy as np
import torch
from torch import nn

_EPS = float(np.finfo(np.float32).eps)            # the package clamps |alpha-2| and alpha away from 0 by float32 eps


def general_loss(x, alpha, scale):
    """rho(x, alpha, c) (eq. 1) in the numerically safe form: b = |alpha-2|+eps, d = alpha +- eps."""
    z = x / scale
    z = z * z                                  # not `** 2`: its backward copies through the runtime's memcpy (training._sq)
    b = torch.abs(alpha - 2) + _EPS
    d = torch.where(alpha >= 0, alpha + _EPS, alpha - _EPS)
    return (b / d) * (torch.pow(z / b + 1.0, 0.5 * d) - 1.0)


def _log_partition_table(a_max=4.0, n=513):
    """log Z(alpha) on a uniform alpha grid by composite Simpson quadrature in t with x = sinh(t) (heavy tails at alpha -> 0)."""
    alphas = np.linspace(0.0, a_max, n)
    t = np.linspace(0.0, 40.0, 20001)                      # x up to sinh(40) ~ 1e17
    x = np.sinh(t)
    dx = np.cosh(t)
    out = np.empty(n)
    w = np.ones_like(t)
    w[1:-1:2], w[2:-1:2] = 4.0, 2.0
    w *= (t[1] - t[0]) / 3.0
    z = x * x
    for i, a in enumerate(alphas):
        if abs(a - 2.0) < 1e-9:
            rho = 0.5 * z
        elif a < 1e-9:
            rho = np.log1p(0.5 * z)
        else:
            b = abs(a - 2.0)
            rho = (b / a) * (np.power(z / b + 1.0, 0.5 * a) - 1.0)
        with np.errstate(over="ignore", invalid="ignore"):
            f = np.exp(-rho + np.log(dx))
        f[~np.isfinite(f)] = 0.0
        out[i] = np.log(2.0 * np.sum(w * f))
    return alphas, out


_TABLE = None
_TABLE_DEV = {}


def _table():
    global _TABLE
    if _TABLE is None:
        a, v = _log_partition_table()
        v[0] = math.log(math.pi * math.sqrt(2.0))           # closed forms at the two analytic points
        v[np.argmin(np.abs(a - 2.0))] = 0.5 * math.log(2.0 * math.pi)
        slope = np.gradient(v, a)
        _TABLE = (a, v, slope)
    return _TABLE


def log_base_partition_function(alpha):
    """log Z(alpha) for alpha in [0, 4], differentiable in alpha (cubic Hermite interpolation of the table)."""
    a, v, s = _table()
    h = float(a[1] - a[0])
    key = (alpha.dtype, alpha.device)
    if key not in _TABLE_DEV:            # uploaded once per device: a per-call host->device copy would stall the stream
        _TABLE_DEV[key] = (torch.as_tensor(v, dtype=alpha.dtype, device=alpha.device), torch.as_tensor(s, dtype=alpha.dtype, device=alpha.device))
    va, sa = _TABLE_DEV[key]
    u = torch.clamp(alpha, 0.0, float(a[-1])) / h
    i = torch.clamp(u.detach().floor().long(), 0, len(a) - 2)
    f = u - i
    g = 1 - f
    g2 = g * g                                 # (not `** 2`: training._sq)
    h00 = (1 + 2 * f) * g2
    h10 = f * g2
    h01 = f * f * (3 - 2 * f)
    h11 = f * f * (f - 1)
    return h00 * va[i] + h10 * h * sa[i] + h01 * va[i + 1] + h11 * h * sa[i + 1]


def _inv_softplus(y):
    return math.log(math.expm1(y))


class AdaptiveLossFunction(nn.Module):
    """Same constructor and methods as the class the reference builds (Net_Tool_2.py:69,78,82):
    `AdaptiveLossFunction(num_dims, float_dtype, device, alpha_lo, alpha_hi, alpha_init, scale_lo, scale_init)`;
    `lossfun(x[M, num_dims]) -> nll[M, num_dims]`, `alpha() -> [1, num_dims]`, `scale() -> [1, num_dims]`."""

    def __init__(self, num_dims, float_dtype=torch.float32, device="cpu", alpha_lo=0.001, alpha_hi=1.999, alpha_init=None,
                 scale_lo=1e-5, scale_init=1.0):
        super().__init__()
        if not (0 <= alpha_lo <= alpha_hi <= 4.0):
            raise ValueError("alpha range must satisfy 0 <= alpha_lo <= alpha_hi <= 4")
        if scale_lo <= 0 or scale_init < scale_lo:
            raise ValueError("scale_lo must be > 0 and scale_init >= scale_lo")
        self.num_dims, self.float_dtype = int(num_dims), float_dtype
        self.alpha_lo, self.alpha_hi, self.scale_lo, self.scale_init = float(alpha_lo), float(alpha_hi), float(scale_lo), float(scale_init)
        if alpha_lo == alpha_hi:
            self.register_buffer("fixed_alpha", torch.full((1, num_dims), float(alpha_lo), dtype=float_dtype, device=device))
            self.latent_alpha = None
        else:
            a0 = 0.5 * (alpha_lo + alpha_hi) if alpha_init is None else float(alpha_init)
            if not (alpha_lo < a0 < alpha_hi):
                raise ValueError("alpha_init must lie strictly inside (alpha_lo, alpha_hi)")
            p = (a0 - alpha_lo) / (alpha_hi - alpha_lo)
            self.latent_alpha = nn.Parameter(torch.full((1, num_dims), math.log(p / (1 - p)), dtype=float_dtype, device=device))
        if scale_lo == scale_init:
            self.register_buffer("fixed_scale", torch.full((1, num_dims), float(scale_init), dtype=float_dtype, device=device))
            self.latent_scale = None
        else:
            self.latent_scale = nn.Parameter(torch.zeros((1, num_dims), dtype=float_dtype, device=device))

    def alpha(self):
        if self.latent_alpha is None:
            return self.fixed_alpha
        return torch.sigmoid(self.latent_alpha) * (self.alpha_hi - self.alpha_lo) + self.alpha_lo

    def scale(self):
        if self.latent_scale is None:
            return self.fixed_scale
        # affine softplus: latent 0 -> scale_init, latent -> -inf -> scale_lo
        return (self.scale_init - self.scale_lo) * nn.functional.softplus(self.latent_scale + _inv_softplus(1.0)) + self.scale_lo

    def lossfun(self, x):
        if x.dim() != 2 or x.shape[1] != self.num_dims:
            raise ValueError(f"x must be [M, {self.num_dims}], got {tuple(x.shape)}")
        alpha, scale = self.alpha(), self.scale()
        return general_loss(x, alpha, scale) + torch.log(scale) + log_base_partition_function(alpha)
