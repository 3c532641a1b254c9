#!/usr/bin/env python3
"""A training TRAJECTORY of the reference: N consecutive steps of its own loop (mg_run_NeRF.py:288-326 - get_loss, weighted total, backward, Adam.step,
OneCycleLR.step; optimiser / schedule as Net_Tool_2.py:111-130 builds them) on fixed batches of the synthetic scene of tools/make_trained_golden.py, with the host
RNGs (numpy: sun-ray angles; torch: jitter vectors, sun-ray positions and times) seeded once at the start.  Stored: every step's batch, every step's loss dict,
the learning rates, the final state_dict.  Build container only (imports /root/reference by path; nothing of it is copied).

tests/test_gpu_train.py::test_training_follows_the_reference_trajectory replays it on the GPU through season_nerf_amd.Net_tool with the same seeds: the whole
driver - RNG draw order, sun-ray generator, both passes, loss terms, backward, Adam, schedule, BatchNorm running statistics - against the reference, step by step.

    python tools/make_trajectory_golden.py [W] [steps]
"""
import os
import sys

import numpy as np

sys.argv, ARGV = sys.argv[:1], sys.argv[1:]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_trained_golden as mt                             # noqa: E402  (scene generator; imports the reference through make_golden)
import torch                                                 # noqa: E402

mg = mt.mg


def main():
    W = int(ARGV[0]) if ARGV else 64
    n_steps = int(ARGV[1]) if len(ARGV) > 1 else 40
    classic = "--classic" in ARGV                                 # Solar_Type_2: per-sample shading, the solar branch trained from the image pass (Eval_Tools_2.py:211-212,366-370)
    prior = "--prior" in ARGV                                     # the DSM-prior ("jump start") phase: use_prior, trust = step / n_steps (Eval_Tools_2.py:218-248)
    batch, S, lr = 192, 48, (5e-4 if W <= 64 else 1e-4)         # (a wide network's hidden weights span +-0.005: 5e-4 per Adam step would move them by 10 % a step)
    torch.set_num_threads(4)
    hm = np.random.Generator(np.random.PCG64(9)).uniform(-0.8, 0.6, (48, 48)) if prior else None
    net, _ = mg.make_net(W, 4, 41, hm=hm, train=True)
    ev = mg.All_in_One_Eval(mg.args_ns(S, classic), torch.device("cpu"), n_steps, prior, None, mg.H4, mg.WC)
    opt = torch.optim.Adam(net.parameters(), lr=lr)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=lr, total_steps=n_steps, base_momentum=0.85, max_momentum=0.95, cycle_momentum=False)
    pool = mt.make_scene(6, 1024, 21)
    if prior:      # Supervised_Sample indexes the height map with every sample point of an image ray (T_NeRF_net_v2.py:175-181): keep the rays that stay inside the cube
        inside = ((pool["Top"][:, :2].abs() <= 1) & (pool["Bot"][:, :2].abs() <= 1)).all(1)
        pool = {k: v[inside] for k, v in pool.items()}
    rng = np.random.Generator(np.random.PCG64(5))
    np.random.seed(2024)
    torch.manual_seed(2024)
    out = {"W": W, "C": 4, "S": S, "n_steps": n_steps, "batch": batch, "lr": lr, "seed": 2024, "init_seed": 41, "prior": int(prior), "classic": int(classic)}
    if prior:
        out["hm"] = hm
    names, vals, lrs, snaps = None, [], [], []
    for step in range(n_steps):
        sel = torch.tensor(rng.choice(pool["Top"].shape[0], batch, replace=False))
        data = {k: v[sel] for k, v in pool.items()}
        for k, v in data.items():
            out[f"step{step}_{k}"] = mg.f32(v)
        opt.zero_grad()
        loss = ev.get_loss(data, net, step, True)
        total = 0
        for k in loss:
            total = total + loss[k][0] * loss[k][1]
        total.backward()
        opt.step()
        sched.step()
        snaps.append(np.concatenate([net.state_dict()[f"G_NeRF_net.fc{i}.norm.running_mean"].numpy() for i in (2, 9)] +
                                    [net.state_dict()[f"G_NeRF_net.fc{i}.norm.running_var"].numpy() for i in (2, 9)]))
        names = list(loss.keys())
        vals.append([float(loss[k][0]) for k in names] + [float(total)])
        lrs.append(sched.get_last_lr()[0])
        print(step, vals[-1][-1], flush=True)
    out["loss_names"] = np.array(names + ["total"])
    out["loss_weights"] = np.array([float(loss[k][1]) for k in names])
    out["loss_values"] = np.asarray(vals)
    out["lrs"] = np.asarray(lrs)
    out["running_snapshots"] = np.asarray(snaps, dtype=np.float32)      # per step: [fc2 running_mean | fc9 running_mean | fc2 running_var | fc9 running_var]
    for k, v in net.state_dict().items():
        if W <= 64 or v.numel() <= 4096:                          # wider networks: the small tensors (biases, BatchNorm, heads) + the norm of every large one
            out["sd_" + k] = v.detach().cpu().numpy()
        else:
            out["sdnorm_" + k] = np.float64(v.double().norm())
    path = os.path.join(mg.OUT, f"trajectory{'_prior' if prior else ''}{'_classic' if classic else ''}_W{W}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
