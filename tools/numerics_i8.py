"""Numerics experiment (CPU, oracle `mm=` hook): which matmul number formats keep the rendered colour within the
1e-4 parity bar?  Emulates, in fp64 arithmetic on exactly-representable operands:

  bf16x3   w = w_hi + w_lo, h = h_hi + h_lo (bf16 each),  w_hi h_hi + w_lo h_hi + w_hi h_lo       (round-1 kernel)
  i8x3     16-bit fixed point in two signed int8 digits (x = 256 d1 + d2, d2 in [-128,127]); per-row weight scale,
           activations in [-1,1] on a fixed scale; products d1 d1, d1 d2 + d2 d1 in exact integers, d2 d2 dropped
           (3 int8 MFMAs per 32 features = half the matrix-pipe time of bf16x3)
  bf16     plain bf16 operands (the "fast" mode)

Usage: python tools/numerics_i8.py [W] [R] [S]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import season_nerf_oracle as orc  # noqa: E402


def bf16_round(x):
    return x.to(torch.float32).to(torch.bfloat16).to(torch.float64)


def mm_bf16x3(x, w):
    x, w = x.double(), w.double()
    xh, wh = bf16_round(x), bf16_round(w)
    xl, wl = bf16_round(x - xh), bf16_round(w - wh)
    return (xh @ wh.t() + xl @ wh.t() + xh @ wl.t()).to(torch.float32)


def mm_bf16(x, w):
    return (bf16_round(x.double()) @ bf16_round(w.double()).t()).to(torch.float32)


def digits(v):
    d1 = torch.floor((v + 128) / 256)
    return d1, v - 256 * d1


def make_mm_i8(act_scale=32512.0, w_levels=32512.0, drop_low=True, x_range=1.0, stats=None):
    def mm(x, w):
        x, w = x.double(), w.double()
        xs = act_scale / x_range
        xf = torch.round(torch.clamp(x, -x_range, x_range) * xs)
        sw = w.abs().amax(1, keepdim=True).clamp_min(1e-30) / w_levels       # per output row
        wf = torch.round(w / sw)
        x1, x2 = digits(xf)
        w1, w2 = digits(wf)
        top = x1 @ w1.t()
        mid = x1 @ w2.t() + x2 @ w1.t()
        acc = 65536.0 * top + 256.0 * mid
        if not drop_low:
            acc = acc + x2 @ w2.t()
        if stats is not None:
            stats["top_max"] = max(stats.get("top_max", 0), float(top.abs().max()))
            stats["mid_max"] = max(stats.get("mid_max", 0), float(mid.abs().max()))
        return (acc * sw.t() / xs).to(torch.float32)
    return mm


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 96
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 96
    torch.set_num_threads(8)
    for seed in (0, 1):
        sd = orc.init_weights(W, 4, seed=seed)
        rng = np.random.Generator(np.random.PCG64(seed + 5))
        t = lambda a: torch.tensor(a, dtype=torch.float32)
        top = np.concatenate([rng.uniform(-1, 1, (R, 2)), np.ones((R, 1))], 1)
        bot = np.concatenate([rng.uniform(-1, 1, (R, 2)), -np.ones((R, 1))], 1)
        sun = rng.uniform(0.1, 1, (R, 3)); sun /= np.linalg.norm(sun, axis=1, keepdims=True)
        tau, d = rng.uniform(0, 1, R), rng.uniform(0, 1, R)
        tim = np.stack([np.cos(2 * np.pi * tau), np.sin(2 * np.pi * tau), np.cos(2 * np.pi * d), np.sin(2 * np.pi * d)], 1)
        data = {"Top": t(top), "Bot": t(bot), "Sun_Angle": t(sun), "Time_Encoded": t(tim)}
        with torch.no_grad():
            ref64 = orc.eval_rays(orc.cast_weights(sd, torch.float64), {k: v.double() for k, v in data.items()}, S, False)
            ref32 = orc.eval_rays(sd, data, S, False)
            st = {}
            modes = {"fp32": None, "bf16x3": mm_bf16x3, "i8x3": make_mm_i8(stats=st), "i8x4 (low kept)": make_mm_i8(drop_low=False),
                     "i8x3 15b w": make_mm_i8(w_levels=16256.0), "bf16": mm_bf16}
            for name, mm in modes.items():
                out = ref32 if mm is None else orc.eval_rays(sd, data, S, False, mm=mm)
                def rel(k):
                    a, b = out[k].double(), ref64[k]
                    return float(((a - b).abs() / b.abs().clamp_min(1e-30)).max())
                def mabs(k):
                    return float((out[k].double() - ref64[k]).abs().max())
                print(f"seed {seed} {name:18s} RGB rel {rel('Rendered_Col'):.2e} abs {mabs('Rendered_Col'):.2e} | Rho rel {rel('Rho'):.2e} "
                      f"| Col abs {mabs('Col'):.2e} | SolarVis abs {mabs('Solar_Vis'):.2e} | Adjust abs {mabs('Adjust'):.2e}", flush=True)
            print("   int accumulators: max |top| %.3g (x256 must stay < 2^31 = 2.1e9 -> %.3g), max |mid| %.3g" %
                  (st["top_max"], st["top_max"] * 256, st["mid_max"]))


if __name__ == "__main__":
    main()
