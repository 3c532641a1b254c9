#!/usr/bin/env python3
"""Weight sets that were REALLY trained - by the reference's own loop - and the reference's renderings of them.

Build container only (needs /root/reference; nothing of it is copied: it is imported by path as in tools/make_golden.py).
The reference's training step (mg_run_NeRF.py:288-326: `get_loss` -> weighted total -> `backward` -> `Adam.step` ->
`OneCycleLR.step`, the optimiser / schedule of Net_Tool_2.py:111-130 with main_lite.py's defaults: lr 3 * 10^-4.86, batch 512,
sc_lambda 0.03, MSE colour loss, solar rays on) runs for N steps on a synthetic scene made here (a terrain with buildings, a
seasonal albedo, cast shadows; 12 "images" with their own view direction, sun direction and day of the year).  Stored under
tests/golden/trained_W{W}.npz: the resulting `state_dict` (every array, fp32 / int64 as the reference holds them), the training
loss trajectory, and `All_in_One_Eval.eval` of the reference (eval mode, Eval_Tools_2.py:165-252) on held-out rays of the scene.

    python tools/make_trained_golden.py 64 300        # width, steps   (W = 256: ~3 s per step on 8 cores)
"""
import os
import sys
import time

import numpy as np

sys.argv, ARGV = sys.argv[:1], sys.argv[1:]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg                                   # noqa: E402  (imports the reference with the App. B stubs)
import torch                                               # noqa: E402

orc, T_NeRF, All_in_One_Eval, WC, H4, f32 = mg.orc, mg.T_NeRF, mg.All_in_One_Eval, mg.WC, mg.H4, mg.f32


# ------------------------------------------------------------------ the synthetic scene (ours; the reference never sees this code)
BOXES = [(-0.55, -0.15, -0.6, -0.1, 0.35), (0.1, 0.5, 0.2, 0.7, 0.5), (0.3, 0.75, -0.7, -0.35, 0.15), (-0.8, -0.5, 0.3, 0.8, 0.25)]


def height(x, y):
    h = -0.35 + 0.12 * np.sin(2.1 * x + 0.3) * np.cos(2.7 * y - 0.2) + 0.05 * x
    for x0, x1, y0, y1, top in BOXES:
        h = np.where((x >= x0) & (x <= x1) & (y >= y0) & (y <= y1), top, h)
    return h


def albedo(x, y, tau):
    roof = np.zeros_like(x, dtype=bool)
    for x0, x1, y0, y1, _ in BOXES:
        roof |= (x >= x0) & (x <= x1) & (y >= y0) & (y <= y1)
    veg = (np.sin(5 * x) * np.sin(4 * y + 1) > 0.1) & ~roof
    season = 0.5 + 0.5 * np.cos(2 * np.pi * (tau - 0.55))           # 1 in summer, 0 in winter
    base = np.stack([0.45 + 0.15 * np.sin(7 * x), 0.40 + 0.10 * np.cos(6 * y), 0.35 + 0.1 * np.sin(3 * x + 2 * y)], -1)
    green = np.stack([0.15 + 0.25 * (1 - season), 0.30 + 0.35 * season, 0.10 + 0.10 * (1 - season)], -1)
    roofc = np.stack([0.65 + 0 * x, 0.62 + 0.05 * np.sin(20 * x), 0.60 + 0 * x], -1)
    col = np.where(veg[..., None], green, base)
    col = np.where(roof[..., None], roofc, col)
    return np.clip(col, 0.02, 0.98)


def first_hit(top, bot, n=384):
    ts = np.linspace(0, 1, n)[None, :, None]
    p = top[:, None, :] * (1 - ts) + bot[:, None, :] * ts
    below = p[..., 2] <= height(p[..., 0], p[..., 1])
    idx = np.where(below.any(1), below.argmax(1), n - 1)
    return p[np.arange(len(top)), idx]


def lit(pts, sun, n=192):
    s = np.linspace(0.01, 2.5, n)[None, :, None]
    q = pts[:, None, :] + sun[:, None, :] * s
    inside = (np.abs(q[..., 0]) <= 1) & (np.abs(q[..., 1]) <= 1)
    blocked = inside & (q[..., 2] < height(q[..., 0], q[..., 1]) - 0.01)
    return ~blocked.any(1)


def make_scene(n_img, rays_per_img, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    rows = {k: [] for k in ("Top", "Bot", "Sun_Angle", "Time_Encoded", "GT_Color")}
    for _ in range(n_img):
        az, off = rng.uniform(0, 2 * np.pi), rng.uniform(0.0, 0.45)
        v = np.array([np.sin(az) * np.sin(off), np.cos(az) * np.sin(off), np.cos(off)])
        saz, sel = rng.uniform(0, 2 * np.pi), np.deg2rad(rng.uniform(30, 75))
        sun = np.array([np.sin(saz) * np.cos(sel), np.cos(saz) * np.cos(sel), np.sin(sel)])
        tau, day = rng.uniform(0, 1), rng.uniform(0, 1)
        mid = np.concatenate([rng.uniform(-0.75, 0.75, (rays_per_img, 2)), np.zeros((rays_per_img, 1))], 1)
        top, bot = mid + v / v[2], mid - v / v[2]
        hit = first_hit(top, bot)
        sunr = np.tile(sun, (rays_per_img, 1))
        shade = lit(hit, sunr)[:, None]
        sky = np.array([0.30, 0.36, 0.52])
        col = albedo(hit[:, 0], hit[:, 1], tau) * np.where(shade, 1.0, sky)
        rows["Top"].append(top), rows["Bot"].append(bot), rows["Sun_Angle"].append(sunr), rows["GT_Color"].append(col)
        rows["Time_Encoded"].append(np.tile([np.cos(2 * np.pi * tau), np.sin(2 * np.pi * tau), np.cos(2 * np.pi * day), np.sin(2 * np.pi * day)],
                                            (rays_per_img, 1)))
    return {k: torch.tensor(np.concatenate(v), dtype=torch.float32) for k, v in rows.items()}


# ------------------------------------------------------------------ the reference's training loop
def train(W, n_steps, batch=512, S=96, seed=0):
    torch.manual_seed(1000 + W)
    np.random.seed(1000 + W)
    net, _ = mg.make_net(W, 4, 40 + seed, train=True)
    args = mg.args_ns(S)
    ev = All_in_One_Eval(args, torch.device("cpu"), n_steps, False, None, H4, WC)
    lr = 10 ** (-4.86) * 3                                                       # main_lite.py:75
    opt = torch.optim.Adam(net.parameters(), lr=lr)                              # Net_Tool_2.py:111-112
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=lr, total_steps=n_steps, base_momentum=0.85, max_momentum=0.95,
                                                cycle_momentum=False)            # Net_Tool_2.py:123-125
    pool = make_scene(12, 4096, 7)
    n = pool["Top"].shape[0]
    rng = np.random.Generator(np.random.PCG64(3))
    traj = []
    t0 = time.time()
    for step in range(n_steps):                                                  # mg_run_NeRF.py:288-326
        sel = torch.tensor(rng.choice(n, batch, replace=False))
        data = {k: v[sel] for k, v in pool.items()}
        opt.zero_grad()
        loss = ev.get_loss(data, net, step, True)
        total = 0
        for k in loss:
            total = total + loss[k][0] * loss[k][1]
        total.backward()
        opt.step()
        sched.step()
        traj.append([float(total)] + [float(loss[k][0]) for k in ("Color", "Solar_Correction", "Albedo_Color", "Sky_Color_Var") if k in loss])
        if step % 10 == 0 or step == n_steps - 1:
            print(f"W{W} step {step:4d} total {float(total):.5f} colour {float(loss['Color'][0]):.5f}  ({time.time() - t0:.0f} s)", flush=True)
    return net, np.asarray(traj)


def main():
    W = int(ARGV[0]) if ARGV else 64
    n_steps = int(ARGV[1]) if len(ARGV) > 1 else 300
    torch.set_num_threads(int(os.environ.get("SNERF_GOLDEN_THREADS", "4")))
    net, traj = train(W, n_steps)
    net.train(False)
    R, S = 64, 96
    held = make_scene(4, R // 4, 99)                                            # held-out views / suns / days of the same scene
    out = {"W": W, "C": 4, "S": S, "n_steps": n_steps, "loss_trajectory": traj}
    for k, v in net.state_dict().items():
        out["sd_" + k] = v.detach().cpu().numpy()
    for k, v in held.items():
        out["in_" + k] = f32(v)
    with torch.no_grad():
        ev = All_in_One_Eval(mg.args_ns(S), torch.device("cpu"), 10, False, None, H4, WC)
        r = ev.eval(held, net, 0, False)
        for k in ["Rendered_Col", "Albedo_Color", "Rho", "Solar_Vis", "Col", "Sky_Col", "Classes"]:
            out["eval_" + k] = f32(r[k][:, 0] if k in ("Sky_Col", "Classes") else r[k])
        loc = torch.sum(r["PS"] * r["sample_pts"], 1) / (torch.sum(r["PS"], 1) + 1e-8)       # mg_run_NeRF.py:188
        dist = torch.sum(torch.cumsum(r["deltas"], 1) * r["PS"], 1) / torch.sum(r["PS"], 1)   # mg_run_NeRF.py:189
        out["eval_surf_loc"], out["eval_surf_dist"] = f32(loc), f32(dist)
    path = os.path.join(mg.OUT, f"trained_W{W}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; final colour loss", traj[-1][1], "first", traj[0][1])


if __name__ == "__main__":
    main()
