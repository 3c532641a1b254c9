"""Known-answer pins of the Barron adaptive loss restatement (season_nerf_amd/adaptive_loss.py; the third-party package the
reference imports at Net_Tool_2.py:8 is not installable here, SURVEY 8c: PARITY UNPINNED beyond these closed forms)."""
import math

import numpy as np
import pytest
import torch

import season_nerf_amd as sn
from oracle import season_nerf_oracle as orc


def make(alpha, scale, dims=3):
    # alpha_lo == alpha_hi / scale_lo == scale_init pin the shape parameters exactly
    return sn.AdaptiveLossFunction(dims, torch.float64, "cpu", alpha_lo=alpha, alpha_hi=alpha, scale_lo=scale, scale_init=scale)


def test_closed_forms():
    x = torch.linspace(-3, 3, 41, dtype=torch.float64).reshape(-1, 1).repeat(1, 3)
    c = 0.37
    got = make(2.0, c).lossfun(x)
    exp = 0.5 * (x / c) ** 2 + math.log(c) + 0.5 * math.log(2 * math.pi)
    np.testing.assert_allclose(got.numpy(), exp.numpy(), rtol=2e-6, atol=2e-5)
    got = make(0.0, c).lossfun(x)
    exp = torch.log(0.5 * (x / c) ** 2 + 1) + math.log(c) + math.log(math.pi * math.sqrt(2))
    np.testing.assert_allclose(got.numpy(), exp.numpy(), rtol=2e-6, atol=2e-5)
    # alpha = 1 (pseudo-Huber): Z = 2 e K_1(1)
    from scipy.special import k1
    got = make(1.0, c).lossfun(x)
    exp = torch.sqrt((x / c) ** 2 + 1) - 1 + math.log(c) + math.log(2 * math.e * k1(1.0))
    np.testing.assert_allclose(got.numpy(), exp.numpy(), rtol=2e-6, atol=2e-5)


def test_is_a_normalised_density():
    """exp(-nll) integrates to 1 for alphas between the table nodes."""
    x = torch.linspace(-400, 400, 800001, dtype=torch.float64).reshape(-1, 1)
    for a in (1.337, 1.9931, 2.5077, 2.99):
        p = torch.exp(-make(a, 0.5, 1).lossfun(x))[:, 0]
        assert abs(float(torch.trapz(p, x[:, 0])) - 1.0) < 2e-4, a


def test_matches_oracle_rho_and_reference_construction():
    ada = sn.AdaptiveLossFunction(3, torch.float32, "cpu", alpha_hi=2.99, alpha_init=2.0, scale_init=0.03, scale_lo=0.01)   # Net_Tool_2.py:69
    assert torch.allclose(ada.alpha(), torch.full((1, 3), 2.0), atol=1e-6)
    assert torch.allclose(ada.scale(), torch.full((1, 3), 0.03), atol=1e-7)
    assert len(list(ada.parameters())) == 2
    x = torch.randn(50, 3) * 0.1
    nll = ada.lossfun(x)
    rho = orc.barron_rho(x, ada.alpha().detach(), ada.scale().detach())
    const = torch.log(ada.scale()) + 0.5 * math.log(2 * math.pi)
    assert torch.allclose(nll, rho + const, atol=2e-4, rtol=1e-5)
    torch.mean(nll).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in ada.parameters())
    with pytest.raises(ValueError):
        ada.lossfun(torch.zeros(4, 2))
