#!/usr/bin/env python3
"""GPU: every fused arithmetic mode on the density-gain ladder of tests/golden/sharp_sweep_W*.npz (trained weights, density head x g, g = 1 ... 256: fog to hard
surfaces; references = the reference's own eval per g, tools/make_sharp_golden.py sweep) - observed RGB / depth error of forced int8 digits and of bf16x3 against the
reference, the pack-time prediction `rgb_pred`, and what `auto` resolves to.  The table this prints is what csrc/pack.cpp estimate_i8 is validated against
(profiles/r5/sharp_modes.txt).

    python tools/sharp_modes.py [64 256 512]
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import season_nerf_amd as sn  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")
HEAD = ("G_NeRF_net.fc10Sigma.weight", "G_NeRF_net.fc10Sigma.bias")


def rel(a, b):
    a = a.detach().cpu().double().numpy().reshape(np.asarray(b).shape)
    b = np.asarray(b, dtype=np.float64)
    return float((np.abs(a - b) / np.maximum(np.abs(b), 1e-3)).max())


def render(sd, W, S, data, precision):
    net = sn.T_NeRF(W, 4)
    net.load_state_dict(sd)
    net.precision = precision
    net = net.to("cuda").eval()
    if precision != "auto" and net.resolved_precision is None:
        return net, None
    args = SimpleNamespace(n_samples=S, Use_Reg=True, Solar_Type_2=False, Use_MSE_loss=True, Use_Solar=True, sc_lambda=0.03, number_low_frequency_cases=4)
    ev = sn.All_in_One_Eval(args, torch.device("cuda"), 10, False, None, np.eye(4), np.zeros(3))
    with torch.no_grad():
        out = ev.eval(data, net, 0, False)
    out["surf_dist"] = torch.sum(torch.cumsum(out["deltas"], 1) * out["PS"], 1) / torch.sum(out["PS"], 1)
    return net, out


def main():
    widths = [int(a) for a in sys.argv[1:]] or [64, 256, 512]
    print(f"{'W':>4} {'g':>4} {'max-PS':>7} | {'i8x3 vs ref':>11} {'vs f64':>9} {'rgb_pred':>9} {'pred/obs':>8} | {'bf16x3 vs ref':>13} {'vs f64':>9} | {'ref vs f64':>10} | auto")
    for W in widths:
        g = dict(np.load(os.path.join(GOLD, f"sharp_sweep_W{W}.npz"), allow_pickle=False))
        t = dict(np.load(os.path.join(GOLD, str(g["source"])), allow_pickle=False))
        S = int(g["S"])
        data = {k: torch.tensor(g["in_" + k]) for k in ("Top", "Bot", "Sun_Angle", "Time_Encoded")}
        for gain in [int(v) for v in g["gains"]]:
            sd = {k[3:]: torch.tensor(v) * (float(gain) if k[3:] in HEAD else 1.0) for k, v in t.items() if k.startswith("sd_")}
            keys = ("Rendered_Col", "Albedo_Color", "surf_dist")
            worst = lambda o: max(rel(o[k], g[f"g{gain}_{k}"]) for k in keys)
            net8, o8 = render(sd, W, S, data, "i8x3")
            est = net8.i8_estimate()
            r8, x8 = worst(o8), rel(o8["Rendered_Col"], g[f"g{gain}_Rendered_Col64"])
            if W != 512:
                _, o3 = render(sd, W, S, data, "bf16x3")
                r3, x3 = worst(o3), rel(o3["Rendered_Col"], g[f"g{gain}_Rendered_Col64"])
                b3 = f"{r3:13.2e} {x3:9.2e}"
            else:
                b3 = f"{'-':>13} {'-':>9}"
            neta, _ = render(sd, W, S, data, "auto")
            fl = float((np.abs(g[f"g{gain}_Rendered_Col"].astype(np.float64) - g[f"g{gain}_Rendered_Col64"]) / np.maximum(np.abs(g[f"g{gain}_Rendered_Col64"]), 1e-3)).max())
            print(f"{W:4d} {gain:4d} {float(g[f'g{gain}_max_ps'].mean()):7.3f} | {r8:11.2e} {x8:9.2e} {est['rgb_pred']:9.2e} {est['rgb_pred'] / r8:8.2f} | {b3} | {fl:10.2e} | "
                  f"{neta.resolved_precision}" + (f" (probe: rgb {pr['rgb_dev']:.1e} depth {pr['depth_dev']:.1e})" if (pr := neta.i8_probe()) and pr.get("ran") else ""), flush=True)


if __name__ == "__main__":
    main()
