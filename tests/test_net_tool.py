"""`T_NeRF_Net_Tool` (Net_Tool_2.py:11-145): the learning-phase schedule and save points on the CPU (known answers produced by
the reference's own misc.get_output_loc_lin_first, tools/make_golden.py::gen_schedule), the phase switch / optimiser reset /
learning-rate trajectory on the GPU."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch


def _args(n_steps, n_saves=40, use_mse=True, jump_start=True, W=64, lr=10 ** -4.86 * 3):
    return SimpleNamespace(max_train_steps=n_steps, n_saves=n_saves, fc_units=W, number_low_frequency_cases=4, lr=lr, lr_alpha_scale=1000,
                           jump_start=jump_start, Use_MSE_loss=use_mse, batch_size=32, n_samples=32, Use_Reg=True, Solar_Type_2=False,
                           Use_Solar=True, sc_lambda=0.03)


def test_save_points_match_reference(golden_dir):
    from season_nerf_amd.trainer import get_output_loc_lin_first
    g = np.load(os.path.join(golden_dir, "micro.npz"), allow_pickle=False)
    keys = [k for k in g.files if k.startswith("outloc_")]
    assert len(keys) >= 6
    for k in keys:
        n_steps, n_out, gap = (int(x) for x in k.split("_")[1:])
        np.testing.assert_array_equal(get_output_loc_lin_first(n_steps, n_out, gap), g[k], err_msg=k)


def test_phase_schedule_arithmetic():
    """ps = [0.2, 0, 0, 0.8] (Net_Tool_2.py:23-41): section starts / ends / lengths and which mode a step falls into (:135)."""
    import season_nerf_amd as sn
    tool = sn.T_NeRF_Net_Tool.__new__(sn.T_NeRF_Net_Tool)      # schedule only: no device needed
    n = 5000
    ps = [0.2, 0.0, 0.0, 0.8]
    p = [int(ps[0] * n), 0, 0]
    p.append(n - sum(p))
    starts = np.array([0, p[0], p[0], p[0]])
    # the same arithmetic through the class, without touching the GPU: build with a stub network
    import season_nerf_amd.network as nw
    real = nw.T_NeRF.to
    try:
        nw.T_NeRF.to = lambda self, *a, **k: self
        tool.__init__(_args(n), np.zeros((4, 4)), np.zeros((4, 4)), "cpu", np.eye(4), np.zeros(3))
    finally:
        nw.T_NeRF.to = real
    np.testing.assert_array_equal(tool.section_starts, starts)
    np.testing.assert_array_equal(tool.section_Ends, [1000, 1000, 1000, 5000])
    assert tool.Section_Steps == [1000, 0, 0, 4000]
    modes = [int(np.sum(s >= tool.section_starts)) for s in (0, 999, 1000, 4999)]
    assert modes == [1, 1, 4, 4]
    np.testing.assert_array_equal(tool.sub_section_outputs[0], np.linspace(1, 1000, 9, dtype=int)[1:])       # 8 saves x 1000 >= 1000
    assert tool.sub_section_outputs[3][-1] == n and len(tool.sub_section_outputs[1]) == 0
    assert tool.network.layer_width == 64 and tuple(tool.network.hm.shape) == (4, 4)


@pytest.mark.gpu
def test_phase_switch_and_lr_trajectory(golden_dir):
    """10 steps: phase 1 (DSM prior on, steps 0-1) -> phase 4 (prior off): a new evaluator, fresh Adam moments and a new
    OneCycleLR over the phase length at the switch; the learning rate follows torch's OneCycleLR exactly as the reference
    configures it (pinned for a 1000-step phase against the reference-side trajectory in micro.npz)."""
    import season_nerf_amd as sn
    from oracle import season_nerf_oracle as orc
    rng = np.random.Generator(np.random.PCG64(3))
    hm = rng.uniform(-0.8, 0.6, (24, 24))
    R = 32
    t = lambda a: torch.tensor(a, dtype=torch.float32)
    data = {"Top": t(np.concatenate([rng.uniform(-1, 1, (R, 2)), np.ones((R, 1))], 1)),
            "Bot": t(np.concatenate([rng.uniform(-1, 1, (R, 2)), -np.ones((R, 1))], 1)),
            "Sun_Angle": torch.nn.functional.normalize(t(rng.uniform(0.1, 1, (R, 3))), dim=1),
            "Time_Encoded": t(rng.uniform(-1, 1, (R, 4))), "GT_Color": t(rng.uniform(0, 1, (R, 3)))}
    calls = []
    for use_mse in (True, False):
        args = _args(10, n_saves=5, use_mse=use_mse)
        WC, H4 = np.array([41.29, -95.9, 300.0]), np.array([[310.0, 12.0, 0.0, -11650.0], [-9.0, 240.0, 0.0, 23390.0], [0.0, 0.0, 0.01, -3.0], [0, 0, 0, 1.0]])
        tool = sn.T_NeRF_Net_Tool(args, hm, hm, "cuda", H4, WC, get_data=lambda eval_mode: (calls.append(eval_mode), data)[1])
        tool.network.load_state_dict(orc.init_weights(64, 4, 1))
        lrs, priors, evs, adam_steps = [], [], [], []
        for s in range(10):
            tool.step()
            lrs.append(tool.sched.get_last_lr()[0])
            priors.append(tool.eval_tool.use_prior)
            evs.append(id(tool.eval_tool))
            adam_steps.append(tool.network._param_store.adam_steps)
            if s == 1:
                ada1 = tool.eval_tool.ada_loss
        assert priors == [True] * 2 + [False] * 8
        assert len(set(evs[:2])) == 1 and len(set(evs[2:])) == 1 and evs[0] != evs[2]
        assert adam_steps == [1, 2, 1, 2, 3, 4, 5, 6, 7, 8]            # a new Adam at the phase entry starts from step 0
        assert tool.eval_tool.n_steps == 10 and tool._step_count == 10

        def ref_lrs(total):
            p = torch.nn.Parameter(torch.zeros(1))
            opt = torch.optim.Adam([p], lr=args.lr)
            sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=args.lr, total_steps=total, base_momentum=0.85, max_momentum=0.95, cycle_momentum=False)
            out = []
            for _ in range(total):                   # the reference steps the scheduler once per training step (mg_run_NeRF.py:322)
                opt.step()
                sch.step()
                out.append(sch.get_last_lr()[0])
            return out
        np.testing.assert_allclose(lrs[:2], ref_lrs(2)[:2], rtol=1e-12)
        np.testing.assert_allclose(lrs[2:], ref_lrs(8), rtol=1e-12)
        if not use_mse:                                  # Barron mode: [colour loss, alpha loss] in phase 1, one inherited loss object after
            assert isinstance(ada1, list) and len(ada1) == 2 and tool.optim2 is not None
            assert not isinstance(tool.eval_tool.ada_loss, (list, tuple))
        else:
            assert tool.optim2 is None and tool.eval_tool.ada_loss is None
    assert calls.count(True) >= 2                        # eval_step at the save points pulled validation batches
    g = np.load(os.path.join(golden_dir, "micro.npz"), allow_pickle=False)
    # the reference's trajectory for a 1000-step phase (main_lite settings) through OUR driver's scheduler construction
    args = _args(5000, lr=float(g["onecycle_max_lr"]))
    tool = sn.T_NeRF_Net_Tool(args, hm, hm, "cuda", np.eye(4), np.zeros(3))
    tool.learning_mode = 1
    tool.reset_eval()
    got = []
    for _ in range(int(g["onecycle_total"]) - 1):
        tool.sched.step()                                # the LR law only: no gradient step needed (torch warns about the order)
        got.append(tool.sched.get_last_lr()[0])
    np.testing.assert_allclose(got, g["onecycle_lr"], rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("use_mse,prior", [(True, False), (False, False), (True, True), (False, True)])
def test_graphed_train_step_follows_the_eager_step(use_mse, prior):
    """trainer.GraphedTrainStep: the training step (mg_run_NeRF.py:288-326) captured once and replayed as one hipGraph launch per step must BE the
    eager step: same host draws (jitter, random sun rays: seeded alike), same loss values, same parameters after eight steps under the OneCycleLR
    schedule (the learning rate and Adam's bias corrections reach the captured Adam kernel through device memory) - to the noise of atomically
    reduced gradients.
    All four configurations of the reference's run: MSE / Barron's adaptive loss (its alpha and scale have their own Adam and schedule, captured in
    torch's capturable form: they must end where the eager ones end) x free phase / DSM-prior phase (trust = step / n_steps changes every step:
    the captured composite kernels read it from one device float)."""
    import season_nerf_amd as sn
    from oracle import season_nerf_oracle as orc
    from season_nerf_amd.adaptive_loss import AdaptiveLossFunction
    W, R, S, n_steps = 64, 96, 32, 8
    hm = np.random.Generator(np.random.PCG64(3)).uniform(-0.8, 0.6, (24, 24))
    WC, H4 = np.array([41.29, -95.9, 300.0]), np.array([[310.0, 12.0, 0.0, -11650.0], [-9.0, 240.0, 0.0, 23390.0], [0.0, 0.0, 0.01, -3.0], [0, 0, 0, 1.0]])
    rng = np.random.Generator(np.random.PCG64(5))
    t = lambda a: torch.tensor(a, dtype=torch.float32, device="cuda")
    sun = rng.uniform(0.1, 1, (R, 3)); sun /= np.linalg.norm(sun, axis=1, keepdims=True)
    tau = rng.uniform(0, 1, (R, 2))
    batches = []
    for k in range(n_steps):
        batches.append({"Top": t(np.concatenate([rng.uniform(-1, 1, (R, 2)), np.ones((R, 1))], 1)), "Bot": t(np.concatenate([rng.uniform(-1, 1, (R, 2)), -np.ones((R, 1))], 1)),
                        "Sun_Angle": t(sun), "Time_Encoded": t(np.stack([np.cos(6.28 * tau[:, 0]), np.sin(6.28 * tau[:, 0]), np.cos(6.28 * tau[:, 1]), np.sin(6.28 * tau[:, 1])], 1)),
                        "GT_Color": t(rng.uniform(0, 1, (R, 3)))})

    def run(graphed):
        net = sn.T_NeRF(W, 4, HM=hm) if prior else sn.T_NeRF(W, 4)
        net.load_state_dict(orc.init_weights(W, 4, 7, bn_stats="identity"))
        net = net.cuda().train()
        args = SimpleNamespace(n_samples=S, Use_Reg=True, Solar_Type_2=False, Use_MSE_loss=use_mse, Use_Solar=True, sc_lambda=0.03, number_low_frequency_cases=4)
        mk = lambda dims, s0, slo: AdaptiveLossFunction(dims, torch.float32, "cuda", alpha_hi=2.99, alpha_init=2.0, scale_init=s0, scale_lo=slo)
        ada = None if use_mse else ([mk(3, .03, .01), mk(1, .5, .05)] if prior else mk(3, .03, .01))       # as Net_Tool_2.py:69-82 builds them
        ev = sn.All_in_One_Eval(args, torch.device("cuda"), 10, prior, ada, H4, WC)
        tool = sn.Net_tool(net, ev, 3e-4, total_steps=n_steps + 1, lr_alpha_scale=30.0, writer=None)   # (a large rate lets Adam amplify the rounding noise of the atomics into the losses)
        assert (tool.optim2 is None) == use_mse
        step = sn.GraphedTrainStep(tool, batches[0], warmup=2) if graphed else None
        ada0 = [p.detach().clone() for p in tool._ada_params]
        np.random.seed(11); torch.manual_seed(11)
        losses = []
        for k in range(n_steps):
            loss = step(batches[k], k) if graphed else tool.train_step(batches[k], k)
            losses.append({n: float(v[0]) for n, v in loss.items()})
        if graphed:
            assert step.graph is not None and step.calls == n_steps
        lr2 = None if use_mse else float(tool.optim2.param_groups[0]["lr"])
        return losses, {k: v.detach().clone() for k, v in net.state_dict().items()}, (tool.sched.get_last_lr()[0], lr2), [(p.detach().clone(), p0) for p, p0 in zip(tool._ada_params, ada0)]

    le, pe, lre, ae = run(False)
    lg, pg, lrg, ag = run(True)
    assert lre[0] == lrg[0] and (use_mse or lrg[1] == pytest.approx(lre[1], rel=1e-6))       # the second schedule's rate: a device float in the captured run
    assert len(ae) == (0 if use_mse else (4 if prior else 2))
    for (a, a0), (b, _) in zip(ae, ag):       # alpha / scale latents: moved by Adam steps of ~lr2 each, the same way in both runs
        moved = float((a - a0).abs().max())
        assert moved > 1e-3 and float((a - b).abs().max()) <= 0.03 * moved, (a0, a, b)
    for k in range(n_steps):
        for n, v in le[k].items():
            assert abs(lg[k][n] - v) <= 2e-4 * max(abs(v), 1e-3), (k, n, lg[k][n], v)
    # Parameters: Adam divides by sqrt(v), so the rounding noise of atomically reduced gradients moves weights whose gradient is ~0 by up to lr per
    # step in either run - element-wise equality is not a property of the eager step either.  What must hold: the two runs made the SAME update
    # (schedule, bias corrections, moments), i.e. their difference is small against the distance travelled from the initial weights.
    p0 = orc.init_weights(W, 4, 7, bn_stats="identity")
    moved = diff = 0.0
    for k, v in pe.items():
        if v.is_floating_point() and "running" not in k:
            moved += float((v.cpu() - p0[k]).abs().sum())
            diff += float((pg[k] - v).abs().sum())
        elif not v.is_floating_point():
            assert torch.equal(pg[k], v), k                      # num_batches_tracked: the captured step counts its forwards too
    print(f"  graphed vs eager after {n_steps} steps: sum |difference| / sum |update| = {diff / moved:.2e}")
    assert moved > 0 and diff < 0.02 * moved, (diff, moved)
    for k, v in pe.items():
        if "running" in k:
            # (the running means follow the gradient-free biases in front of BatchNorm, which Adam moves by rounding noise: 4e-3 of the largest mean with
            # the adaptive loss, whose gradients are ~500x those of the MSE loss at scale 0.03)
            assert float((pg[k] - v).abs().max()) <= (5e-3 if use_mse else 1.5e-2) * max(float(v.abs().max()), 1e-2), k


@pytest.mark.gpu
def test_graphed_steps_without_readback_keep_their_own_adam_scalars():
    """`FusedAdam.set_hyper` used to rewrite ONE pinned buffer per step and copy it asynchronously - with no per-step sync the
    host runs several steps ahead and the Adam kernel of step k could read the learning rate / bias corrections of step k+1..k+5 (just after the
    capture 1 - beta1^t goes 0.1 -> 0.47: early step sizes change by large factors).  24 captured steps with NO host readback between them (the
    host queues all of them before the GPU has finished the first few) must land on the eager run's parameters; `keep=True` losses, read only at
    the end, must be the per-step values (not 24 copies of the last one)."""
    import season_nerf_amd as sn
    from oracle import season_nerf_oracle as orc
    W, R, S, n_steps = 64, 512, 64, 24          # ~a millisecond of GPU work per step: the host queues far ahead
    WC, H4 = np.array([41.29, -95.9, 300.0]), np.array([[310.0, 12.0, 0.0, -11650.0], [-9.0, 240.0, 0.0, 23390.0], [0.0, 0.0, 0.01, -3.0], [0, 0, 0, 1.0]])
    rng = np.random.Generator(np.random.PCG64(9))
    t = lambda a: torch.tensor(a, dtype=torch.float32, device="cuda")
    sun = rng.uniform(0.1, 1, (R, 3)); sun /= np.linalg.norm(sun, axis=1, keepdims=True)
    batch = {"Top": t(np.concatenate([rng.uniform(-1, 1, (R, 2)), np.ones((R, 1))], 1)), "Bot": t(np.concatenate([rng.uniform(-1, 1, (R, 2)), -np.ones((R, 1))], 1)),
             "Sun_Angle": t(sun), "Time_Encoded": t(rng.uniform(-1, 1, (R, 4))), "GT_Color": t(rng.uniform(0, 1, (R, 3)))}

    def run(graphed):
        net = sn.T_NeRF(W, 4)
        net.load_state_dict(orc.init_weights(W, 4, 7, bn_stats="identity"))
        net = net.cuda().train()
        args = SimpleNamespace(n_samples=S, Use_Reg=True, Solar_Type_2=False, Use_MSE_loss=True, Use_Solar=True, sc_lambda=0.03, number_low_frequency_cases=4)
        ev = sn.All_in_One_Eval(args, torch.device("cuda"), 10, False, None, H4, WC)
        tool = sn.Net_tool(net, ev, 3e-4, total_steps=n_steps + 1, writer=None)
        step = sn.GraphedTrainStep(tool, batch, warmup=2, keep=True) if graphed else None
        np.random.seed(5); torch.manual_seed(5)
        kept = []
        for k in range(n_steps):                      # no float(), no .cpu(), no synchronize inside the loop
            kept.append((step(batch, k) if graphed else tool.train_step(batch, k))["Color"][0])
        torch.cuda.synchronize()
        return [float(v) for v in kept], {k: v.detach().clone() for k, v in net.state_dict().items()}

    le, pe = run(False)
    lg, pg = run(True)
    assert len(set(lg[2:])) > n_steps // 2, lg        # per-step values survived the later replays
    for k in range(n_steps):
        assert abs(lg[k] - le[k]) <= 5e-4 * max(abs(le[k]), 1e-3), (k, lg[k], le[k])
    p0 = orc.init_weights(W, 4, 7, bn_stats="identity")
    moved = diff = 0.0
    for k, v in pe.items():
        if v.is_floating_point() and "running" not in k:
            moved += float((v.cpu() - p0[k]).abs().sum())
            diff += float((pg[k] - v).abs().sum())
    print(f"  24 captured steps without readback vs eager: sum |difference| / sum |update| = {diff / moved:.2e}")
    assert moved > 0 and diff < 0.03 * moved, (diff, moved)


@pytest.mark.gpu
def test_driver_with_use_graph_switches_to_the_captured_step():
    """T_NeRF_Net_Tool(..., use_graph=True): 20 steps - phase 1 (DSM prior, 4 steps) then phase 4; in each phase the first two steps run eagerly and
    the rest as one hipGraph launch each; a new phase (new evaluator / optimisers) drops the captured step.  The learning rate follows the same
    OneCycleLR trajectory as the eager driver, Adam's step count too, and the loss keeps falling through the switch."""
    import season_nerf_amd as sn
    from oracle import season_nerf_oracle as orc
    rng = np.random.Generator(np.random.PCG64(3))
    hm = rng.uniform(-0.8, 0.6, (24, 24))
    R = 48
    t = lambda a: torch.tensor(a, dtype=torch.float32)
    data = {"Top": t(np.concatenate([rng.uniform(-1, 1, (R, 2)), np.ones((R, 1))], 1)), "Bot": t(np.concatenate([rng.uniform(-1, 1, (R, 2)), -np.ones((R, 1))], 1)),
            "Sun_Angle": torch.nn.functional.normalize(t(rng.uniform(0.1, 1, (R, 3))), dim=1), "Time_Encoded": t(rng.uniform(-1, 1, (R, 4))),
            "GT_Color": t(rng.uniform(0, 1, (R, 3)))}
    WC, H4 = np.array([41.29, -95.9, 300.0]), np.array([[310.0, 12.0, 0.0, -11650.0], [-9.0, 240.0, 0.0, 23390.0], [0.0, 0.0, 0.01, -3.0], [0, 0, 0, 1.0]])
    out = {}
    for use_graph in (False, True):
        args = _args(20, n_saves=5, use_mse=True)
        tool = sn.T_NeRF_Net_Tool(args, hm, hm, "cuda", H4, WC, get_data=lambda eval_mode: data, use_graph=use_graph)
        tool.network.load_state_dict(orc.init_weights(64, 4, 1))
        np.random.seed(3); torch.manual_seed(3)
        lrs, steps, colour, graphed = [], [], [], []
        for s_ in range(20):
            tool.step()
            lrs.append(tool.sched.get_last_lr()[0])
            steps.append(tool.network._param_store.adam_steps)
            colour.append(float(tool.last_loss["Color"][0]))
            graphed.append(tool._graphed is not None and tool._graphed.graph is not None)
        out[use_graph] = (lrs, steps, colour, graphed)
    assert out[False][3] == [False] * 20
    assert out[True][3] == [False] * 2 + [True] * 2 + [False] * 2 + [True] * 14      # steps 0-3: prior phase (2 eager + 2 replays); 4-5: eager warm-up of the new phase; then replays
    assert out[True][0] == out[False][0] and out[True][1] == out[False][1]
    np.testing.assert_allclose(out[True][2], out[False][2], rtol=2e-3)


@pytest.mark.gpu
def test_graphed_step_checks_its_state_and_falls_back_to_eager():
    """the captured step tests the parameters, Adam's moments and the adaptive loss object's state for non-finite values on the device every
    `check_every` replays (asynchronous, read one period later) and, on a hit, warns and runs eagerly from then on.  A healthy Barron run never trips it;
    a moment poisoned by hand does, two periods later at most."""
    import warnings
    import season_nerf_amd as sn
    from oracle import season_nerf_oracle as orc
    from season_nerf_amd.adaptive_loss import AdaptiveLossFunction
    W, R, S = 64, 64, 24
    WC, H4 = np.array([41.29, -95.9, 300.0]), np.array([[310.0, 12.0, 0.0, -11650.0], [-9.0, 240.0, 0.0, 23390.0], [0.0, 0.0, 0.01, -3.0], [0, 0, 0, 1.0]])
    rng = np.random.Generator(np.random.PCG64(15))
    t = lambda a: torch.tensor(a, dtype=torch.float32, device="cuda")
    sun = rng.uniform(0.1, 1, (R, 3)); sun /= np.linalg.norm(sun, axis=1, keepdims=True)
    batch = {"Top": t(np.concatenate([rng.uniform(-1, 1, (R, 2)), np.ones((R, 1))], 1)), "Bot": t(np.concatenate([rng.uniform(-1, 1, (R, 2)), -np.ones((R, 1))], 1)),
             "Sun_Angle": t(sun), "Time_Encoded": t(np.tile([1.0, 0.0, 0.0, 1.0], (R, 1))), "GT_Color": t(rng.uniform(0, 1, (R, 3)))}
    net = sn.T_NeRF(W, 4)
    net.load_state_dict(orc.init_weights(W, 4, 7, bn_stats="identity"))
    net = net.cuda().train()
    args = SimpleNamespace(n_samples=S, Use_Reg=True, Solar_Type_2=False, Use_MSE_loss=False, Use_Solar=True, sc_lambda=0.03, number_low_frequency_cases=4)
    ada = AdaptiveLossFunction(3, torch.float32, "cuda", alpha_hi=2.99, alpha_init=2.0, scale_init=.03, scale_lo=.01)
    ev = sn.All_in_One_Eval(args, torch.device("cuda"), 10, False, ada, H4, WC)
    tool = sn.Net_tool(net, ev, 1e-4, total_steps=64, lr_alpha_scale=30.0, writer=None)
    step = sn.GraphedTrainStep(tool, batch, warmup=2, check_every=3)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                      # a healthy run: no warning in 2 eager + 12 replayed steps (4 checks)
        for k in range(14):
            step(batch, k)
        torch.cuda.synchronize()
    assert step.graph is not None and not step.disabled
    net._param_store.adam_v[5] = float("nan")               # poison one second moment: Adam turns it into a NaN parameter at the next replay
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        for k in range(14, 24):
            step(batch, k)
            torch.cuda.synchronize()
    assert step.disabled and any("non-finite" in str(x.message) for x in w), (step.disabled, [str(x.message) for x in w])
    n = step.calls
    step(batch, 24)                                          # eager from here on: the call count of the captured path no longer moves
    assert step.calls == n
