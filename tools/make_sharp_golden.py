#!/usr/bin/env python3
"""Weight sets WITH SURFACES IN THEM, rendered by the reference.

The "really trained" fixtures of round 4 (tests/golden/trained_W*.npz: 300-600 steps of the reference's loop) are fog: no sample owns more
than 5 % of a ray.  A converged Season-NeRF has opaque surfaces - the density head `G_NeRF_net.fc10Sigma` (G_NeRF.py:52,96; softplus at
T_NeRF_net_v2.py:91) grows until one or two samples own a ray, and the reference's first 20 % of training force exactly that with the
DSM prior (main_lite.py:44-47, Net_Tool_2.py:23-33, Eval_Tools_2.py:218-248).  Two families, both through the REFERENCE (imported by path
as in tools/make_golden.py; nothing of it is copied, only arrays are stored):

  scaled   tests/golden/sharp_W{W}.npz     the trained fixture's weights with fc10Sigma (weight and bias) scaled by g, g the smallest power
                                           of two for which the reference's own `eval` gives MEAN MAX-PS PER RAY >= 0.5 on the fixture's rays;
                                           stored: g, the rays, the reference's eval (fp32) and - for W = 256 - both renderers
                                           (component_render_by_dir incl. exact solar, the image assembly, the 12-step sweep, Quick_Run_Net).
                                           The weights themselves are NOT stored again: tests rebuild them from trained_W{W}.npz and g.
  prior    tests/golden/prior_trained_W{W}.npz   the reference's own DSM-prior phase (`use_prior=True`: supervised density, merged
                                           renderings, Alpha_Adjust; Eval_Tools_2.py:218-248, :413-420) run for N steps on the synthetic scene of
                                           tools/make_trained_golden.py with its true height map, then the free phase for M steps; stored: the
                                           resulting state_dict, the loss trajectory and the reference's eval of held-out rays.

Every fixture also carries `eval64_*`: OUR oracle in float64 on the same inputs (not a pin - the yardstick that says how far the reference's
own fp32 arithmetic is from exact on these weights; tests print it next to their tolerance).

    python tools/make_sharp_golden.py scaled 64 256 512
    python tools/make_sharp_golden.py prior 64 400 200
    python tools/make_sharp_golden.py converged 64 12000 6          (74 CPU-minutes -> trained12k_W64.npz;  `converged 64 40000 8`: 3.9 CPU-hours -> trained40k_W64.npz)
"""
import os
import sys
import time

import numpy as np

sys.argv, ARGV = sys.argv[:1], sys.argv[1:]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg                                     # noqa: E402  (imports the reference with the App. B stubs)
import torch                                                 # noqa: E402

orc, T_NeRF, All_in_One_Eval, WC, H4, f32 = mg.orc, mg.T_NeRF, mg.All_in_One_Eval, mg.WC, mg.H4, mg.f32
HEAD = ("G_NeRF_net.fc10Sigma.weight", "G_NeRF_net.fc10Sigma.bias")


def scaled_state(sd, g):
    return {k: (v * g if k in HEAD else v) for k, v in sd.items()}


def ref_net(W, sd, hm=None):
    net = T_NeRF(W, 4) if hm is None else T_NeRF(W, 4, HM=hm)
    r = net.load_state_dict(sd, strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    net.train(False)
    return net


def ref_eval(net, data, S):
    with torch.no_grad():
        ev = All_in_One_Eval(mg.args_ns(S), torch.device("cpu"), 10, False, None, H4, WC)
        return ev.eval(data, net, 0, False)


def eval_record(out, r):
    for k in ["Rendered_Col", "Albedo_Color", "Rho", "Solar_Vis", "Col", "Sky_Col", "Classes"]:
        out["eval_" + k] = f32(r[k][:, 0] if k in ("Sky_Col", "Classes") else r[k])
    loc = torch.sum(r["PS"] * r["sample_pts"], 1) / (torch.sum(r["PS"], 1) + 1e-8)       # mg_run_NeRF.py:188
    dist = torch.sum(torch.cumsum(r["deltas"], 1) * r["PS"], 1) / torch.sum(r["PS"], 1)   # mg_run_NeRF.py:189
    out["eval_surf_loc"], out["eval_surf_dist"] = f32(loc), f32(dist)
    out["max_ps"] = f32(r["PS"].max(1).values.reshape(-1))


def oracle64(out, sd, data, S):
    sd64 = orc.cast_weights(sd, torch.float64)
    d64 = {k: v.double() for k, v in data.items()}
    with torch.no_grad():
        r = orc.eval_rays(sd64, d64, S, train_mode=False)
    out["eval64_Rendered_Col"] = r["Rendered_Col"].numpy()
    out["eval64_Albedo_Color"] = r["Albedo_Color"].numpy()
    dist = torch.sum(torch.cumsum(r["deltas"], 1) * r["PS"], 1) / torch.sum(r["PS"], 1)
    out["eval64_surf_dist"] = dist.numpy()


def rays_for(g, extra=64):
    """The trained fixture's held-out rays (which leave the cube near its rim) + `extra` in-cube rays of the benchmark's law."""
    held = {k[3:]: torch.tensor(v) for k, v in g.items() if k.startswith("in_")}
    more = mg.synth_rays(extra, 901)
    return {k: torch.cat([held[k], more[k]], 0) for k in held}


def gen_scaled(W, S=96):
    g = dict(np.load(os.path.join(mg.OUT, f"trained_W{W}.npz"), allow_pickle=False))
    sd = {k[3:]: torch.tensor(v) for k, v in g.items() if k.startswith("sd_")}
    data = rays_for(g)
    chosen = None
    for gain in (8, 16, 32, 64, 128, 256, 512):
        r = ref_eval(ref_net(W, scaled_state(sd, gain)), data, S)
        mps = float(r["PS"].max(1).values.mean())
        print(f"W{W} g {gain:4d}: mean max-PS per ray {mps:.3f}", flush=True)
        if mps >= 0.5:
            chosen = gain
            break
    assert chosen is not None
    sds = scaled_state(sd, chosen)
    net = ref_net(W, sds)
    out = {"W": W, "C": 4, "S": S, "g": chosen, "source": np.array(f"trained_W{W}.npz")}
    for k, v in data.items():
        out["in_" + k] = f32(v)
    eval_record(out, r)
    oracle64(out, sds, data, S)
    if W == 256:
        render_record(out, net)
    path = os.path.join(mg.OUT, f"sharp_W{W}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; g", chosen, "mean max-PS", float(out["max_ps"].mean()))


def render_record(out, net):
    """Both renderers of the reference on the sharp weights (as tools/make_trained_render_golden.py), plus the exact-solar pass
    (mg_Img_Eval.py:57-70, the default of both renderers) at a size the CPU finishes in a minute."""
    size = (20, 18, 96)
    view, sun, tf = (75, 40), (40, 120), 0.55
    out.update({"WC": WC, "H": H4, "size": np.array(size), "view": np.array(view), "sun": np.array(sun), "time_frac": tf})
    d = mg.component_render_by_dir(net, view, sun, tf, size, WC, H4, torch.device("cpu"), include_exact_solar=False)
    im = mg.get_imgs_from_Img_Dict(d, size, False)
    for k in ["Base_Img", "Season_Adj_Img", "Shadow_Adjust", "Shadow_Mask", "Raw_Shadow_Mask"]:
        out["img_" + k] = im[k]
    taus = np.arange(12) / 12.0
    with torch.no_grad():
        cls = net.get_class_only(torch.tensor(np.stack([mg.encode_time(t) for t in taus]), dtype=torch.float32)).numpy()
    out["sweep_classes"] = cls
    out["sweep_imgs"] = mg.get_imgs_from_Img_Dict_t_step(d, size, cls.astype(np.float64))
    xs = (24, 20, 96)
    t0 = time.time()
    dx = mg.component_render_by_dir(net, view, sun, tf, xs, WC, H4, torch.device("cpu"), include_exact_solar=True)
    out["xs_size"] = np.array(xs)
    out["xs_Exact_Solar"] = np.asarray(dx["Exact_Solar"])
    imx = mg.get_imgs_from_Img_Dict(dx, xs, True)
    for k in imx:
        out["xs_img_" + k] = imx[k]
    print(f"exact solar {xs}: {time.time() - t0:.0f} s; keys {sorted(imx)}", flush=True)
    qr = mg.Quick_Run_Net(net, mg.args_ns(96), WC, H4, torch.device("cpu"), use_full_solar=False)
    imgs, mask = qr.render_img((65, 20), (50, 100), 0.3, 22)
    out["qr_Col_Img"], out["qr_Shadow_Mask"], out["qr_mask"] = imgs["Col_Img"], imgs["Shadow_Mask"], mask
    out["qr_DSM"] = qr.get_DSM((14, 14))
    qx = mg.Quick_Run_Net(net, mg.args_ns(96), WC, H4, torch.device("cpu"), use_full_solar=True)      # path A's exact solar (Eval_Tools_2.py:255-295)
    imgs, mask = qx.render_img((70, 200), (50, 100), 0.6, 9)
    out["qrx_Col_Img"], out["qrx_Shadow_Mask"], out["qrx_Est_Shadow_Mask"], out["qrx_mask"] = imgs["Col_Img"], imgs["Shadow_Mask"], imgs["Estimated_Shadow_Mask"], mask


def gen_prior(W, n_prior, n_free, batch=512, S=96):
    """The reference's two learning phases (Net_Tool_2.py:23-54) at a reduced length: `n_prior` steps with the DSM prior (learning mode 1 with
    `jump_start`), then `n_free` steps without (mode 4), each with a fresh Adam + OneCycleLR over the phase (Net_Tool_2.py:111-130)."""
    import make_trained_golden as mt
    torch.manual_seed(2000 + W)
    np.random.seed(2000 + W)
    n = 96
    gx, gy = np.meshgrid(np.linspace(-1, 1, n), np.linspace(-1, 1, n), indexing="ij")
    hm = mt.height(gx, gy)                                   # the scene's true surface as the training DSM (cube units)
    sd0 = orc.init_weights(W, 4, 40)
    net = T_NeRF(W, 4, HM=hm)
    r = net.load_state_dict(sd0, strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    net.train(True)
    in_cube = lambda d: ((d["Top"].abs() <= 1).all(1) & (d["Bot"].abs() <= 1).all(1))
    cull = lambda d: {k: v[in_cube(d)] for k, v in d.items()}    # the reference's training rays never leave the cube (its Supervised_Sample indexes the DSM unguarded)
    pool = cull(mt.make_scene(12, 4096, 7))
    npool = pool["Top"].shape[0]
    rng = np.random.Generator(np.random.PCG64(4))
    lr = 10 ** (-4.86) * 3                                   # main_lite.py:75
    traj, t0, step_all = [], time.time(), 0
    for phase, n_steps, prior in ((1, n_prior, True), (4, n_free, False)):
        if n_steps <= 0:
            continue
        ev = All_in_One_Eval(mg.args_ns(S), torch.device("cpu"), n_steps, prior, None, H4, WC)
        opt = torch.optim.Adam(net.parameters(), lr=lr)
        sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=lr, total_steps=n_steps, base_momentum=0.85, max_momentum=0.95, cycle_momentum=False)
        for step in range(n_steps):                          # mg_run_NeRF.py:288-326
            sel = torch.tensor(rng.choice(npool, batch, replace=False))
            data = {k: v[sel] for k, v in pool.items()}
            opt.zero_grad()
            loss = ev.get_loss(data, net, step, True)
            total = 0
            for k in loss:
                total = total + loss[k][0] * loss[k][1]
            total.backward()
            opt.step()
            sched.step()
            traj.append([phase, float(total), float(loss["Color"][0]), float(loss["Alpha_Adjust"][0]) if "Alpha_Adjust" in loss else np.nan])
            if step % 20 == 0 or step == n_steps - 1:
                print(f"W{W} phase {phase} step {step:4d} total {float(total):.5f} colour {float(loss['Color'][0]):.5f}  ({time.time() - t0:.0f} s)", flush=True)
            step_all += 1
    net.train(False)
    held = cull(mt.make_scene(4, 32, 99))
    held = {k: v[:48] for k, v in held.items()}
    more = mg.synth_rays(64, 902)
    data = {k: torch.cat([held[k], more[k]], 0) for k in held}
    out = {"W": W, "C": 4, "S": S, "n_prior": n_prior, "n_free": n_free, "loss_trajectory": np.asarray(traj), "hm": hm}
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    for k, v in sd.items():
        out["sd_" + k] = v.cpu().numpy()
    for k, v in data.items():
        out["in_" + k] = f32(v)
    r = ref_eval(net, data, S)
    eval_record(out, r)
    oracle64(out, sd, data, S)
    path = os.path.join(mg.OUT, f"prior_trained_W{W}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; mean max-PS per ray", float(out["max_ps"].mean()))


def gen_converged(W, n_total, hours, target=0.3, batch=512, S=96):
    """tests/golden/trained{n_total / 1000}k_W{W}.npz: the reference's own schedule (Net_Tool_2.py:23-33, main_lite.py:44-47: the first 20 % of the steps under
    the DSM prior, learning mode 1, the rest free, mode 4; a fresh Adam + OneCycleLR per phase, Net_Tool_2.py:111-130) on the synthetic scene, checked every 250
    steps with the reference's own eval on held-out rays and STOPPED in the free phase once the mean max-PS per ray reaches `target` (or at `hours` of wall
    clock): surfaces that training produced, not a scaled density head.  Stored as prior_trained: state_dict, trajectory, the reference's eval, the oracle in float64."""
    import make_trained_golden as mt
    torch.manual_seed(3000 + W)
    np.random.seed(3000 + W)
    n = 96
    gx, gy = np.meshgrid(np.linspace(-1, 1, n), np.linspace(-1, 1, n), indexing="ij")
    hm = mt.height(gx, gy)
    net = T_NeRF(W, 4, HM=hm)
    r = net.load_state_dict(orc.init_weights(W, 4, 41), strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    net.train(True)
    in_cube = lambda d: ((d["Top"].abs() <= 1).all(1) & (d["Bot"].abs() <= 1).all(1))
    cull = lambda d: {k: v[in_cube(d)] for k, v in d.items()}
    pool = cull(mt.make_scene(12, 4096, 7))
    npool = pool["Top"].shape[0]
    held = cull(mt.make_scene(4, 32, 99))
    held = {k: v[:48] for k, v in held.items()}
    more = mg.synth_rays(64, 902)
    data_h = {k: torch.cat([held[k], more[k]], 0) for k in held}
    rng = np.random.Generator(np.random.PCG64(5))
    lr = 10 ** (-4.86) * 3                                   # main_lite.py:75
    n_prior = int(round(0.2 * n_total))
    traj, checks, t0, done = [], [], time.time(), False
    for phase, n_steps, prior in ((1, n_prior, True), (4, n_total - n_prior, False)):
        ev = All_in_One_Eval(mg.args_ns(S), torch.device("cpu"), n_steps, prior, None, H4, WC)
        opt = torch.optim.Adam(net.parameters(), lr=lr)
        sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=lr, total_steps=n_steps, base_momentum=0.85, max_momentum=0.95, cycle_momentum=False)
        for step in range(n_steps):                          # mg_run_NeRF.py:288-326
            sel = torch.tensor(rng.choice(npool, batch, replace=False))
            data = {k: v[sel] for k, v in pool.items()}
            opt.zero_grad()
            loss = ev.get_loss(data, net, step, True)
            total = 0
            for k in loss:
                total = total + loss[k][0] * loss[k][1]
            total.backward()
            opt.step()
            sched.step()
            traj.append([phase, float(total), float(loss["Color"][0])])
            if step % 250 == 249 or step == n_steps - 1:
                net.train(False)
                mps = float(ref_eval(net, data_h, S)["PS"].max(1).values.mean())
                net.train(True)
                checks.append([phase, step, mps, time.time() - t0])
                print(f"W{W} phase {phase} step {step:5d} total {float(total):.5f} colour {float(loss['Color'][0]):.5f} held-out mean max-PS {mps:.3f} ({time.time() - t0:.0f} s)", flush=True)
                if phase == 4 and (mps >= target or time.time() - t0 > hours * 3600):
                    done = True
                    break
        if done:
            break
    net.train(False)
    out = {"W": W, "C": 4, "S": S, "n_total": n_total, "loss_trajectory": np.asarray(traj), "checks": np.asarray(checks), "hm": hm}
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    for k, v in sd.items():
        out["sd_" + k] = v.cpu().numpy()
    for k, v in data_h.items():
        out["in_" + k] = f32(v)
    r = ref_eval(net, data_h, S)
    eval_record(out, r)
    oracle64(out, sd, data_h, S)
    path = os.path.join(mg.OUT, f"trained{n_total // 1000}k_W{W}.npz")          # trained12k_W64.npz, trained40k_W64.npz: the committed fixtures
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; mean max-PS per ray", float(out["max_ps"].mean()), "steps", len(traj))


def gen_sweep(W, S=96, gains=(1, 2, 4, 8, 16, 32, 64, 128, 256)):
    """tests/golden/sharp_sweep_W{W}.npz: the reference's eval (RGB, albedo, depth, max-PS) of the trained fixture's weights for a LADDER of density-head
    gains, fog (g = 1) to hard surfaces (g = 256): what the pack-time error model of the int8 digits is validated against (tools/sharp_modes.py)."""
    g = dict(np.load(os.path.join(mg.OUT, f"trained_W{W}.npz"), allow_pickle=False))
    sd = {k[3:]: torch.tensor(v) for k, v in g.items() if k.startswith("sd_")}
    data = rays_for(g)
    out = {"W": W, "C": 4, "S": S, "gains": np.array(gains), "source": np.array(f"trained_W{W}.npz")}
    for k, v in data.items():
        out["in_" + k] = f32(v)
    for gain in gains:
        sds = scaled_state(sd, gain)
        r = ref_eval(ref_net(W, sds), data, S)
        out[f"g{gain}_Rendered_Col"], out[f"g{gain}_Albedo_Color"] = f32(r["Rendered_Col"]), f32(r["Albedo_Color"])
        out[f"g{gain}_surf_dist"] = f32(torch.sum(torch.cumsum(r["deltas"], 1) * r["PS"], 1) / torch.sum(r["PS"], 1))
        out[f"g{gain}_max_ps"] = f32(r["PS"].max(1).values.reshape(-1))
        tmp = {}
        oracle64(tmp, sds, data, S)
        out[f"g{gain}_Rendered_Col64"] = tmp["eval64_Rendered_Col"]
        print(f"W{W} g {gain:4d}: mean max-PS {float(out[f'g{gain}_max_ps'].mean()):.3f}", flush=True)
    path = os.path.join(mg.OUT, f"sharp_sweep_W{W}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def main():
    torch.set_num_threads(int(os.environ.get("SNERF_GOLDEN_THREADS", "4")))
    mode = ARGV[0] if ARGV else "scaled"
    if mode == "scaled":
        for W in [int(a) for a in ARGV[1:]] or [64, 256, 512]:
            gen_scaled(W)
    elif mode == "sweep":
        for W in [int(a) for a in ARGV[1:]] or [64, 256, 512]:
            gen_sweep(W)
    elif mode == "prior":
        gen_prior(int(ARGV[1]), int(ARGV[2]), int(ARGV[3]) if len(ARGV) > 3 else 0)
    elif mode == "converged":          # width, total steps of the schedule, wall-clock cap in hours
        gen_converged(int(ARGV[1]), int(ARGV[2]), float(ARGV[3]) if len(ARGV) > 3 else 6.0)
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
