"""Soak of the captured training step in the reference's DEFAULT configuration - Barron's adaptive loss, DSM-prior jump start (phase 1 = 20 % of the run), then the
free phase - through T_NeRF_Net_Tool(..., use_graph=True): every phase captured after two eager steps, validation renders at the save points in between.  Checks:
finite losses at every step, a falling colour error, the same learning-rate trajectory and Adam step count as the eager driver, final colour error within 10 % of it.
    python3 tools/soak_driver.py [steps=400]"""
import math
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import season_nerf_amd as sn  # noqa: E402

n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
R, S, W = 1024, 64, 64
rng = np.random.Generator(np.random.PCG64(3))
hm = rng.uniform(-0.6, 0.4, (32, 32))
t = lambda a: torch.tensor(a, dtype=torch.float32)
top = np.concatenate([rng.uniform(-1, 1, (R, 2)), np.ones((R, 1))], 1)
data = {"Top": t(top), "Bot": t(np.concatenate([top[:, :2] + rng.uniform(-0.1, 0.1, (R, 2)), -np.ones((R, 1))], 1).clip(-1, 1)),
        "Sun_Angle": torch.nn.functional.normalize(t(rng.uniform(0.1, 1, (R, 3))), dim=1), "Time_Encoded": t(rng.uniform(-1, 1, (R, 4)))}
data["GT_Color"] = torch.stack([0.5 + 0.4 * torch.sin(3 * data["Top"][:, 0]), 0.5 + 0.4 * torch.cos(2 * data["Top"][:, 1]), 0.5 + 0.3 * torch.sin(data["Top"][:, 0] + data["Top"][:, 1])], 1)
WC, H4 = np.array([41.29, -95.9, 300.0]), np.array([[310.0, 12.0, 0.0, -11650.0], [-9.0, 240.0, 0.0, 23390.0], [0.0, 0.0, 0.01, -3.0], [0, 0, 0, 1.0]])
out = {}
for use_graph in (False, True):
    args = SimpleNamespace(max_train_steps=n_steps, n_saves=8, fc_units=W, number_low_frequency_cases=4, lr=3e-4, lr_alpha_scale=30, jump_start=True, Use_MSE_loss=False,
                           batch_size=R, n_samples=S, Use_Reg=True, Solar_Type_2=False, Use_Solar=True, sc_lambda=0.03)
    tool = sn.T_NeRF_Net_Tool(args, hm, hm, "cuda", H4, WC, get_data=lambda eval_mode: data, use_graph=use_graph)
    tool.network.load_state_dict(sn.synthetic_state_dict(tool.network, 1, bn_stats="identity"))
    np.random.seed(3); torch.manual_seed(3)
    col, lrs, graphed = [], [], 0
    for k in range(n_steps):
        tool.step()
        c = float(tool.last_loss["Color"][0])
        assert math.isfinite(c) and all(math.isfinite(float(v[0])) for v in tool.last_loss.values()), (use_graph, k, {n: float(v[0]) for n, v in tool.last_loss.items()})
        col.append(c); lrs.append(tool.sched.get_last_lr()[0])
        graphed += tool._graphed is not None and tool._graphed.graph is not None
    out[use_graph] = (col, lrs, tool.network._param_store.adam_steps, graphed)
    print(f"use_graph={use_graph}: colour error {col[0]:.4f} -> {col[n_steps // 5 - 1]:.4f} (end of the prior phase) -> {col[-1]:.4f}, {graphed} of {n_steps} steps replayed", flush=True)
assert out[True][1] == out[False][1] and out[True][2] == out[False][2]
assert out[True][3] >= n_steps - 4 and out[False][3] == 0
assert out[True][0][-1] < 0.5 * out[True][0][0] and abs(out[True][0][-1] - out[False][0][-1]) < 0.1 * out[False][0][-1], (out[True][0][-1], out[False][0][-1])
print("SOAK OK")
