"""GPU: the K-split bf16x3 field kernel of width 512 (csrc/kernels_ks.hip) against the reference's goldens, against the int8-digit kernel over many
tiles, and timed at the benchmark size.  python3 tools/ks_check.py [--time-only]"""
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import season_nerf_amd as sn                                   # noqa: E402
from oracle import season_nerf_oracle as orc                   # noqa: E402  (checker only)

G = os.path.join(REPO, "tests", "golden")
T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)


def err(a, b):
    a = a.detach().cpu().double().numpy().reshape(np.asarray(b).shape)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max(), (np.abs(a - b) / np.maximum(np.abs(b), 1e-3)).max()


def make_net(W, C, seed, precision):
    net = sn.T_NeRF(W, C)
    net.load_state_dict(orc.init_weights(W, C, seed))
    net.precision = precision
    return net.to("cuda").eval()


def goldens():
    g = dict(np.load(os.path.join(G, "net_W512_s3.npz"), allow_pickle=False))
    net = make_net(512, int(g["C"]), int(g["seed"]), "bf16x3")
    assert net.fused and net.resolved_precision == "bf16x3"
    X, sun, tim = T(g["X"]).cuda(), T(g["sun"]).cuda(), T(g["time"]).cuda()
    worst = 0.0
    for k, v in zip(["Rho", "Col", "Solar_Vis", "Sky_Col", "Class", "Adjust"], net.forward(X, sun, tim)):
        a, r = err(v, g["fwd_" + k])
        print(f"  net_W512 fwd_{k:10s} max abs {a:.2e} max rel {r:.2e}")
        worst = max(worst, a)
    r = net.forward_Solar(X, sun, tim)
    print("  forward_Solar rho rel %.2e sv rel %.2e" % (err(r[0], g["solar_Rho"])[1], err(r[1], g["solar_Solar_Vis"])[1]))
    print("  sigma only rel %.2e" % err(net.forward_Classic_Sigma_Only(X), g["sigma_only"])[1])
    o = net.forward_seperate(X, sun, tim)
    print("  seperate col_raw abs %.2e adjust abs %.2e" % (err(o[1], g["sep_Col"])[0] if "sep_Col" in g else -1, err(o[5], g["sep_Adjust"])[0] if "sep_Adjust" in g else -1))
    g = dict(np.load(os.path.join(G, "eval_W512_R64_S96.npz"), allow_pickle=False))
    net = make_net(512, int(g["C"]), int(g["seed"]), "bf16x3")
    data = {k: T(g["in_" + k]) for k in ["Top", "Bot", "Sun_Angle", "Time_Encoded", "GT_Color"]}
    args = SimpleNamespace(n_samples=int(g["S"]), Use_Reg=True, Solar_Type_2=False, Use_MSE_loss=True, Use_Solar=True, sc_lambda=0.03, number_low_frequency_cases=4)
    ev = sn.All_in_One_Eval(args, torch.device("cuda"), 10, False, None, np.eye(4), np.zeros(3))
    out = ev.eval(data, net, 0, False)
    for k in ["Rendered_Col", "Albedo_Color", "Rho", "Col", "Solar_Vis", "PS"]:
        a, r = err(out[k], g["eval_" + k])
        print(f"  eval_W512 {k:14s} max abs {a:.2e} max rel {r:.2e}")
    return worst


def many_tiles():
    """4096 x 96 points: every workgroup walks its ring many times; compare point by point with the int8-digit kernel (a ring / exchange mistake is O(1))."""
    W, R, S = 512, 4096, 96
    rng = np.random.Generator(np.random.PCG64(5))
    top = T(np.concatenate([rng.uniform(-1, 1, (R, 2)), np.ones((R, 1))], 1)).cuda()
    bot = T(np.concatenate([rng.uniform(-1, 1, (R, 2)), -np.ones((R, 1))], 1)).cuda()
    sun = rng.uniform(0.1, 1, (R, 3)); sun /= np.linalg.norm(sun, axis=1, keepdims=True)
    sun = T(sun).cuda()
    a = rng.uniform(0, 2 * np.pi, R)
    tim = T(np.stack([np.cos(a), np.sin(a), np.ones(R), np.zeros(R)], 1)).cuda()
    tv = sn.sample_parameters(S, eval_mode=True).cuda()
    res = {}
    for prec in ("bf16x3", "i8x3"):
        net = make_net(W, 4, 12, prec)
        ops = torch.ops.season_nerf
        rgb, depth, _, _ = ops.render_fwd(net.op_model(), top, bot, sun, tim, tv, 0, False)
        pts = (top[:, None, :] * (1 - tv[None, :, None]) + bot[:, None, :] * tv[None, :, None]).reshape(-1, 3).contiguous()
        f = net.forward(pts, sun.repeat_interleave(S, 0), tim.repeat_interleave(S, 0))
        torch.cuda.synchronize()
        res[prec] = (rgb, depth, f)
        for _ in range(3):
            ops.render_fwd(net.op_model(), top, bot, sun, tim, tv, 0, False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            ops.render_fwd(net.op_model(), top, bot, sun, tim, tv, 0, False)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print(f"  {prec}: {ms:.3f} ms per 4096x96 render step ({R * S / ms * 1e3:.3e} ray-samples/s)")
    (rgb3, d3, f3), (rgb8, d8, f8) = res["bf16x3"], res["i8x3"]
    rel = lambda x, y: float(((x - y).abs() / y.abs().clamp_min(1e-3)).max())
    print(f"  bf16x3 (K-split) vs i8x3 over {R * S} points: rgb rel {rel(rgb3, rgb8):.2e} depth rel {rel(d3[:, 0], d8[:, 0]):.2e} "
          f"rho rel {rel(f3[0], f8[0]):.2e} col abs {float((f3[1] - f8[1]).abs().max()):.2e} sv abs {float((f3[2] - f8[2]).abs().max()):.2e}")
    assert bool(torch.isfinite(rgb3).all())
    return rel(rgb3, rgb8)


def visibility():
    """variant 3 against its composition (density-only pass + transmittance) on a few hundred rays, S = 96 and 33"""
    net = make_net(512, 4, 12, "bf16x3")
    ops = torch.ops.season_nerf
    rng = np.random.Generator(np.random.PCG64(7))
    for S in (96, 33):
        R = 301
        top = T(np.concatenate([rng.uniform(-1, 1, (R, 2)), np.ones((R, 1))], 1)).cuda()
        bot = T(np.concatenate([rng.uniform(-1, 1, (R, 2)), -np.ones((R, 1))], 1)).cuda()
        tv = sn.sample_parameters(S, eval_mode=True).cuda()
        vis = ops.ray_visibility(net.op_model(), top, bot, tv, 0)
        pts = (top[:, None, :] * (1 - tv[None, :, None]) + bot[:, None, :] * tv[None, :, None]).reshape(-1, 3).contiguous()
        rho = net.forward_Classic_Sigma_Only(pts).reshape(R, S)
        delta = (top - bot).norm(dim=1, keepdim=True) / S
        ref = torch.exp(-(rho[:, :-1] * delta).sum(1))
        print(f"  ray_visibility S={S}: max abs dev from composition {float((vis - ref).abs().max()):.2e}")


if __name__ == "__main__":
    if "--time-only" not in sys.argv:
        goldens()
        visibility()
    many_tiles()
