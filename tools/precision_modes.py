"""GPU: the three arithmetic modes of the fused field kernel side by side - error against the reference goldens
(tests/golden/eval_W256_R64_S96.npz, eval_W64_R48_S64.npz) and time of the 4096 x 96 field launch.

    python tools/precision_modes.py [reps]
"""
import ctypes as C
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import season_nerf_amd as sn  # noqa: E402
from oracle import season_nerf_oracle as orc  # noqa: E402  (weights generator only)


def errs(out, g, keys):
    r = {}
    for k in keys:
        a = out[k].detach().cpu().double().numpy().reshape(g["eval_" + k].shape)
        b = g["eval_" + k].astype(np.float64)
        r[k] = (np.abs(a - b).max(), (np.abs(a - b) / np.maximum(np.abs(b), 1e-3)).max())
    return r


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    dev = torch.device("cuda")
    for name in ["eval_W256_R64_S96.npz", "eval_W64_R48_S64.npz"]:
        g = dict(np.load(os.path.join(REPO, "tests", "golden", name), allow_pickle=False))
        Wd, Cc, seed, S = int(g["W"]), int(g["C"]), int(g["seed"]), int(g["S"])
        sd = orc.init_weights(Wd, Cc, seed)
        data = {k: torch.tensor(g["in_" + k], dtype=torch.float32) for k in ["Top", "Bot", "Sun_Angle", "Time_Encoded"]}
        args = SimpleNamespace(n_samples=S, Use_Reg=True, Solar_Type_2=False, Use_MSE_loss=True, Use_Solar=True, sc_lambda=0.03,
                               number_low_frequency_cases=Cc)
        ev = sn.All_in_One_Eval(args, dev, 10, False, None, np.eye(4), np.zeros(3))
        for prec in ["bf16x3", "i8x3", "bf16"]:
            net = sn.T_NeRF(Wd, Cc)
            net.load_state_dict(sd)
            net.precision = prec
            net = net.to(dev).eval()
            out = ev.eval(data, net, 0, False)
            e = errs(out, g, ["Rendered_Col", "Albedo_Color", "Rho", "Col", "Solar_Vis", "Adjust", "PS"])
            print(f"{name} {prec:7s} " + " | ".join(f"{k} abs {v[0]:.2e} rel {v[1]:.2e}" for k, v in e.items()), flush=True)
    # timing at the benchmark size
    R, S, Wd, Cc = 4096, 96, 256, 4
    rng = np.random.Generator(np.random.PCG64(0))
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    top = t(np.concatenate([rng.uniform(-1, 1, (R, 2)), np.ones((R, 1))], 1))
    bot = t(np.concatenate([rng.uniform(-1, 1, (R, 2)), -np.ones((R, 1))], 1))
    sun = rng.uniform(0, 1, (R, 3)); sun = t(sun / np.linalg.norm(sun, axis=1, keepdims=True))
    cls = torch.softmax(torch.randn(R, Cc, device=dev), 1)
    tv = sn.sample_parameters(S, eval_mode=True).to(dev)
    rho, sv, col = (torch.empty(R * S, device=dev), torch.empty(R * S, device=dev), torch.empty(R * S, 3, device=dev))
    fo = sn._lib.FieldOut(d_rho=rho.data_ptr(), d_solar_vis=sv.data_ptr(), d_col=col.data_ptr())
    L = sn._lib.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ref = None
    for prec in ["bf16x3", "i8x3", "bf16"]:
        net = sn.T_NeRF(Wd, Cc)
        net.load_state_dict(sn.synthetic_state_dict(net, 0))
        net.precision = prec
        net = net.to(dev).eval()
        model = net.device_model()
        run = lambda: sn._lib.check(L.snerf_field_forward_rays(model, 0, R, S, top.data_ptr(), bot.data_ptr(), tv.data_ptr(), 1,
                                                              sun.data_ptr(), cls.data_ptr(), C.byref(fo), st), "field")
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        snap = (rho.clone(), col.clone())
        if ref is None:
            ref = snap
        d_rho = ((snap[0] - ref[0]).abs() / ref[0].abs().clamp_min(1e-3)).max().item()
        d_col = (snap[1] - ref[1]).abs().max().item()
        print(f"4096x96 W=256 {prec:7s} field kernel {ms:.3f} ms  ({R * S / ms / 1e3:.3e} ray-samples/s; algorithmic frac of 2.5 PF: "
              f"{1.489e6 * R * S / (ms * 1e-3) / 2.5e15:.3f})  vs bf16x3: rho rel {d_rho:.2e}, col abs {d_col:.2e}", flush=True)


if __name__ == "__main__":
    main()
