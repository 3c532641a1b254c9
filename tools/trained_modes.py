"""GPU: the really-trained weight fixtures (tests/golden/trained_W*.npz) in every arithmetic mode against the reference's eval, beside the
pack-time error model's prediction - the table `estimate_i8`'s head weights are fitted on (csrc/pack.cpp).   python tools/trained_modes.py"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

REPO = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import season_nerf_amd as sn  # noqa: E402

rel = lambda a, b: float(((a.detach().cpu().double().reshape(b.shape) - b).abs() / b.abs().clamp_min(1e-3)).max())
for W in (64, 256, 512):
    path = os.path.join(REPO, "tests", "golden", f"trained_W{W}.npz")
    if not os.path.exists(path):
        continue
    g = dict(np.load(path, allow_pickle=False))
    sd = {k[3:]: torch.tensor(v) for k, v in g.items() if k.startswith("sd_")}
    S = int(g["S"])
    args = SimpleNamespace(n_samples=S, Use_Reg=True, Solar_Type_2=False, Use_MSE_loss=True, Use_Solar=True, sc_lambda=0.03, number_low_frequency_cases=4)
    data = {k: torch.tensor(g["in_" + k]) for k in ("Top", "Bot", "Sun_Angle", "Time_Encoded")}
    for mode in ("auto", "i8x3", "bf16x3"):
        net = sn.T_NeRF(W, 4)
        net.load_state_dict(sd)
        net.precision = mode
        net = net.to("cuda").eval()
        ev = sn.All_in_One_Eval(args, torch.device("cuda"), 10, False, None, np.eye(4), np.zeros(3))
        try:
            with torch.no_grad():
                out = ev.eval(data, net, 0, False)
        except RuntimeError as ex:
            print(f"W={W} {mode}: {ex}")
            continue
        e = net.i8_estimate()
        ps, dl = out["PS"], out["deltas"]
        dist = torch.sum(torch.cumsum(dl, 1) * ps, 1) / torch.sum(ps, 1)
        mx = float(ps.max(1)[0].mean())
        print(f"W={W} {mode:7s} -> {str(net.resolved_precision):7s} rgb_pred {e['rgb_pred']:.2e} heads " + " ".join(f"{v:.2e}" for v in e["head_rms"]) +
              f" | RGB {rel(out['Rendered_Col'], torch.tensor(g['eval_Rendered_Col']).double()):.2e} albedo {rel(out['Albedo_Color'], torch.tensor(g['eval_Albedo_Color']).double()):.2e}"
              f" depth {rel(dist, torch.tensor(g['eval_surf_dist']).double()):.2e} Rho {rel(out['Rho'], torch.tensor(g['eval_Rho']).double()):.2e}"
              f" Col {rel(out['Col'], torch.tensor(g['eval_Col']).double()):.2e}   mean max-PS per ray {mx:.2f}", flush=True)
