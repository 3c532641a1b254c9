"""Odd class counts, ray counts and sample counts through every fused arithmetic mode against the oracle in float64 (on the CPU): the goldens and the benchmark
all sit at C = 4 and friendly sizes; the kernels' tile tails (a 64-point tile that ends inside a ray, one ray, seven samples), the per-ray networks and the class
mixing for C = 1 .. 5 (program.h kMaxClasses) are exercised here.  Tolerances: the modes' bars on init-law weights (bf16x3 1e-5, int8 digits 5e-5 on RGB)."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import season_nerf_oracle as orc      # noqa: E402  (checker)

T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)
LIMIT = {"bf16x3": 1e-5, "i8x3": 5e-5, "bf16": 5e-3}


@pytest.mark.gpu
@pytest.mark.parametrize("W", [64, 256, 512])
@pytest.mark.parametrize("C", [1, 2, 3, 5])
def test_eval_at_odd_class_and_batch_shapes(W, C):
    import season_nerf_amd as sn
    sd = orc.init_weights(W, C, 11 + C)
    sd64 = orc.cast_weights(sd, torch.float64)
    for R, S in ((37, 50), (1, 96), (130, 33), (513, 7)):
        rng = np.random.Generator(np.random.PCG64(R + S))
        data = {"Top": T(np.concatenate([rng.uniform(-1, 1, (R, 2)), np.ones((R, 1))], 1)), "Bot": T(np.concatenate([rng.uniform(-1, 1, (R, 2)), -np.ones((R, 1))], 1)),
                "Sun_Angle": torch.nn.functional.normalize(T(rng.uniform(0.1, 1, (R, 3))), dim=1), "Time_Encoded": T(rng.uniform(-1, 1, (R, 4))),
                "GT_Color": T(rng.uniform(0, 1, (R, 3)))}
        ref = orc.eval_rays(sd64, {k: v.double() for k, v in data.items()}, S, False)
        for prec in ("bf16x3", "i8x3") + (("bf16",) if W != 512 else ()):
            net = sn.T_NeRF(W, C)
            net.load_state_dict(sd)
            net.precision = prec
            net = net.cuda().eval()
            args = SimpleNamespace(n_samples=S, Use_Reg=True, Solar_Type_2=False, Use_MSE_loss=True, Use_Solar=True, sc_lambda=0.03, number_low_frequency_cases=C)
            ev = sn.All_in_One_Eval(args, torch.device("cuda"), 10, False, None, np.eye(4), np.zeros(3))
            out = ev.eval(data, net, 0, False)
            assert net.fused and net.resolved_precision == prec
            e_rgb = float((out["Rendered_Col"].detach().cpu().double() - ref["Rendered_Col"]).abs().max())
            assert e_rgb < LIMIT[prec], (W, C, R, S, prec, e_rgb)
            for k in ("Rho", "Solar_Vis", "PS"):
                a, b = out[k].detach().cpu().double().reshape(R, S), ref[k].reshape(R, S)
                assert bool(torch.isfinite(a).all()) and float(((a - b).abs() / b.abs().clamp_min(1e-2)).max()) < 200 * LIMIT[prec], (W, C, R, S, prec, k)
