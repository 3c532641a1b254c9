#!/usr/bin/env python3
"""Calibration of the int8 error model (csrc/pack.cpp estimate_i8, include/season_nerf_hip.h snerf_i8_estimate) - CPU only.

For weight sets shaped like trained checkpoints (oracle.stress_weights) prints, side by side,
  * the model's prediction: RMS error of the raw head outputs, from the packed integers alone (C ABI, no GPU);
  * what the digit arithmetic really does: the oracle with its matmul replaced by an exact emulation of the i8x3 products
    (tools/numerics_i8.py), against the same network in fp64 - max relative error of the rendered colour, the surface depth and
    the density, RMS error of the raw heads.
The budget SNERF_I8_BUDGET (csrc/api.cpp) is set from this table: every set whose prediction is under it must render within
5e-5 (half the 1e-4 bar).

    python tools/calibrate_i8_bound.py [W] [R] [S]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import season_nerf_oracle as orc  # noqa: E402
from season_nerf_amd import _lib  # noqa: E402
from tools.numerics_i8 import make_mm_i8, mm_bf16x3  # noqa: E402


def estimate(sd, W, Cn):
    L = _lib.lib()
    m = L.snerf_model_create(W, Cn)
    assert m
    try:
        for k, v in sd.items():
            if v.is_floating_point():
                a = np.ascontiguousarray(v.numpy(), dtype=np.float32)
                _lib.check(L.snerf_model_set_tensor(m, k.encode(), a.ctypes.data_as(C.c_void_p), a.size), "set_tensor")
        e = _lib.I8Estimate()
        _lib.check(L.snerf_model_i8_estimate(m, C.byref(e)), "i8_estimate")
        return e
    finally:
        L.snerf_model_destroy(m)


def field_only(sd, mm):
    """The digit arithmetic for the per-point (field) layers only: the per-ray networks (time -> classes, sun -> sky) never run in
    int8 digits (bf16x3 kernel at W = 64 / 256, exact fp32 at 512)."""
    per_ray = {sd[k].data_ptr() for k in sd if k.startswith(("time_layer", "get_class_layer", "G_NeRF_net.fc_sky_color"))}

    def f(x, w):
        return x @ w.t() if w.data_ptr() in per_ray else mm(x, w)
    return f


def rays(R, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    t = lambda a: torch.tensor(a, dtype=torch.float32)
    top = np.concatenate([rng.uniform(-1, 1, (R, 2)), np.ones((R, 1))], 1)
    bot = np.concatenate([rng.uniform(-1, 1, (R, 2)), -np.ones((R, 1))], 1)
    sun = rng.uniform(0.1, 1, (R, 3)); sun /= np.linalg.norm(sun, axis=1, keepdims=True)
    tau, d = rng.uniform(0, 1, R), rng.uniform(0, 1, R)
    tim = np.stack([np.cos(2 * np.pi * tau), np.sin(2 * np.pi * tau), np.cos(2 * np.pi * d), np.sin(2 * np.pi * d)], 1)
    return {"Top": t(top), "Bot": t(bot), "Sun_Angle": t(sun), "Time_Encoded": t(tim)}


def depth(r):
    return torch.sum(torch.cumsum(r["deltas"], 1) * r["PS"], 1) / torch.sum(r["PS"], 1)


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 96
    torch.set_num_threads(8)
    print(f"W={W} R={R} S={S}   pred = predicted RMS of the worst raw head; emu = emulated i8x3 vs fp64")
    print(f"{'weights':14s} {'rgb_pred':>10s} {'pred rho':>9s} {'hidden':>9s} {'acc bound':>10s} | {'RGB rel':>9s} {'depth rel':>9s} {'Rho rel':>9s} {'bf16x3 RGB':>10s} {'fp32 RGB':>9s}")
    for kind in ("init",) + orc.STRESS_KINDS:
        for seed in (0,):
            sd = orc.stress_weights(W, 4, seed, kind) if kind != "init" else orc.init_weights(W, 4, seed)
            e = estimate(sd, W, 4)
            data = rays(R, 77 + seed)
            with torch.no_grad():
                ref = orc.eval_rays(orc.cast_weights(sd, torch.float64), {k: v.double() for k, v in data.items()}, S, False)
                out = orc.eval_rays(sd, data, S, False, mm=field_only(sd, make_mm_i8()))
                o3 = orc.eval_rays(sd, data, S, False, mm=mm_bf16x3)
                o32 = orc.eval_rays(sd, data, S, False)
            rel = lambda a, b: float(((a.double() - b).abs() / b.abs().clamp_min(1e-30)).max())
            print("   heads rho %.2e col %.2e sv %.2e adj %.2e" % tuple(e.head_rms))
            print(f"{kind:14s} {e.rgb_pred:10.2e} {e.head_rms[0]:9.2e} {e.hidden_rms:9.2e} {e.acc_bound:10.3g} | "
                  f"{rel(out['Rendered_Col'], ref['Rendered_Col']):9.2e} {rel(depth(out), depth(ref)):9.2e} {rel(out['Rho'], ref['Rho']):9.2e} "
                  f"{rel(o3['Rendered_Col'], ref['Rendered_Col']):10.2e} {rel(o32['Rendered_Col'], ref['Rendered_Col']):9.2e}", flush=True)


if __name__ == "__main__":
    main()
