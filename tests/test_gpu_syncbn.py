"""Data-parallel BatchNorm over the global batch (TrainEngine.sync_batchnorm -> snerf_trainer_set_allreduce).

One GPU, no second process: a run on the WHOLE batch records the statistics buffers the engine hands to the collective
(their global sums G_k); then each half of the rays runs as a "rank" of a 2-rank job whose all-reduce returns G_k.
Checks: (1) the shards' own local sums add up to G_k - so a real sum-all-reduce would have produced exactly these values;
(2) with them, every shard reproduces the full-batch run, which is what the single-process reference computes: same rendered
colours per ray, same BatchNorm running statistics, averaged shard gradients equal to the full-batch gradients."""
import ctypes as C
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(sn, orc, W, S):
    net = sn.T_NeRF(W, 4)
    net.load_state_dict(orc.init_weights(W, 4, 5))
    net = net.to("cuda").train()
    args = SimpleNamespace(n_samples=S, Use_Reg=True, Solar_Type_2=False, Use_MSE_loss=True, Use_Solar=False, sc_lambda=0.0,
                           number_low_frequency_cases=4)
    ev = sn.All_in_One_Eval(args, torch.device("cuda"), 10, False, None, np.eye(4), np.zeros(3))
    return net, ev


def _rays(R, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    t = lambda a: torch.tensor(a, dtype=torch.float32)
    sun = rng.uniform(0.1, 1, (R, 3))
    tau = rng.uniform(0, 1, (R, 2))
    return {"Top": t(np.concatenate([rng.uniform(-1, 1, (R, 2)), np.ones((R, 1))], 1)),
            "Bot": t(np.concatenate([rng.uniform(-1, 1, (R, 2)), -np.ones((R, 1))], 1)),
            "Sun_Angle": t(sun / np.linalg.norm(sun, axis=1, keepdims=True)),
            "Time_Encoded": t(np.stack([np.cos(6.28 * tau[:, 0]), np.sin(6.28 * tau[:, 0]), np.cos(6.28 * tau[:, 1]), np.sin(6.28 * tau[:, 1])], 1)),
            "GT_Color": t(rng.uniform(0, 1, (R, 3)))}


def _step(net, ev, data, after_forward=lambda: None):
    out = ev.eval(data, net, 0, False)                    # eval-mode sampling (no RNG), train-mode BatchNorm (net.training)
    after_forward()
    rgb = out["Rendered_Col"]
    loss = torch.mean((rgb - data["GT_Color"].cuda()) ** 2)
    loss.backward()
    return rgb.detach().clone(), {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("gemm", ["bf16x3", "fp32"])
def test_two_shards_equal_full_batch(monkeypatch, gemm):
    import season_nerf_amd as sn
    from season_nerf_amd import _lib, training
    from oracle import season_nerf_oracle as orc
    monkeypatch.setenv("SNERF_TRAIN_GEMM", gemm)
    W, S, R = 64, 64, 16                                   # 1024 points per shard: the bf16x3 row kernels are in play
    full = _rays(2 * R, 3)
    shards = [{k: v[i * R:(i + 1) * R] for k, v in full.items()} for i in range(2)]

    net_f, ev_f = _setup(sn, orc, W, S)

    def view_of(eng, ptr, count, is_double):
        off, size = ptr - eng.ws.data_ptr(), count * (8 if is_double else 4)
        assert 0 <= off and off + size <= eng.ws.numel()
        return eng.ws[off:off + size].view(torch.float64 if is_double else torch.float32)

    # full batch, one "rank": record what the engine would all-reduce
    G = []
    eng_f = training._engine_for(net_f, 2 * R, 0, S)
    rec = _lib.ALLREDUCE_FN(lambda user, ptr, count, dbl, stream: (G.append(view_of(eng_f, ptr, count, dbl).clone()), 0)[1])
    _lib.check(_lib.lib().snerf_trainer_set_allreduce(eng_f.h, C.cast(rec, C.c_void_p), None, 1), "set_allreduce")
    n_fwd = []
    rgb_f, grads_f = _step(net_f, ev_f, full, lambda: n_fwd.append(len(G)))
    n_fwd = n_fwd[0]
    bn_f = {k: v.clone() for k, v in net_f.state_dict().items() if "running_" in k}
    assert n_fwd >= 8 and len(G) - n_fwd == 8             # 8 BatchNorm layers: forward statistics, one backward collective each
    # each rank's loss is a mean over ITS rays, so its output gradients are world x those of the global-mean loss: the backward
    # sums a real all-reduce returns are world x the full-batch ones (the gradient average over ranks undoes the factor)
    scale = lambda k: 2.0 if k >= n_fwd else 1.0

    nets = [_setup(sn, orc, W, S) for _ in range(2)]
    results, local = [], []
    for r in range(2):
        eng = training._engine_for(nets[r][0], R, 0, S)
        mine = []

        def cb(user, ptr, count, dbl, stream, eng=eng, mine=mine):
            v = view_of(eng, ptr, count, dbl)
            mine.append(v.clone())
            v.copy_(G[len(mine) - 1] * scale(len(mine) - 1))      # what the 2-rank sum-all-reduce returns
            return 0
        fn = _lib.ALLREDUCE_FN(cb)
        _lib.check(_lib.lib().snerf_trainer_set_allreduce(eng.h, C.cast(fn, C.c_void_p), None, 2), "set_allreduce")
        results.append(_step(nets[r][0], nets[r][1], shards[r]))
        _lib.check(_lib.lib().snerf_trainer_set_allreduce(eng.h, None, None, 1), "set_allreduce")
        local.append(mine)
    assert len(local[0]) == len(local[1]) == len(G)
    for k, g in enumerate(G):                              # (1) local sums of the shards add up to the global ones
        s_ = (local[0][k] + local[1][k]).double().cpu().numpy()
        np.testing.assert_allclose(s_, scale(k) * g.double().cpu().numpy(), rtol=5e-4, atol=1e-3 * float(g.abs().max()), err_msg=f"collective {k}")
    torch.cuda.synchronize()
    rgb = torch.cat([results[0][0], results[1][0]])
    np.testing.assert_allclose(rgb.cpu().numpy(), rgb_f.cpu().numpy(), rtol=2e-4, atol=2e-6)
    gmax = max(float(v.abs().max()) for v in grads_f.values())
    worst = 0.0
    for n, gf in grads_f.items():
        gs = 0.5 * (results[0][1][n] + results[1][1][n])  # data-parallel average of the shard gradients
        scale = max(float(gf.abs().max()), 1e-3 * gmax)
        worst = max(worst, float((gs - gf).abs().max()) / scale)
    assert worst < 2e-3, worst
    for r in range(2):
        sd = nets[r][0].state_dict()
        for k, v in bn_f.items():
            np.testing.assert_allclose(sd[k].cpu().numpy(), v.cpu().numpy(), rtol=1e-4, atol=1e-6, err_msg=k)
    # without the collective the shard statistics differ from the global ones (the test above is not vacuous)
    net_l, ev_l = _setup(sn, orc, W, S)
    rgb_l, _ = _step(net_l, ev_l, shards[0])
    assert float((rgb_l - rgb_f[:R]).abs().max()) > 1e-4


def test_sync_batchnorm_needs_process_group():
    import season_nerf_amd as sn
    from season_nerf_amd import training
    from oracle import season_nerf_oracle as orc
    net, _ = _setup(sn, orc, 64, 32)
    eng = training._engine_for(net, 8, 0, 32)
    if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        with pytest.raises(RuntimeError):
            eng.sync_batchnorm(True)
    eng.sync_batchnorm(False)                              # always allowed: back to per-rank statistics


def test_two_shards_reproduce_full_batch_loss_and_gradients(monkeypatch):
    """The whole `get_loss` (Eval_Tools_2.py:340-459) under data parallelism with global-batch BatchNorm: two half-batch "ranks"
    reproduce the LOSS DICT and the GRADIENTS of the full batch - including `Albedo_Color`, whose minimum the reference takes over
    the whole batch (:374-378): one `ReduceOp.MIN` all-reduce (parallel.global_min, replayed here as the BatchNorm sums are) and the
    global ray count in its denominator.  The colour head is biased dark so that the term is active, and it is weighted 1.0 so that
    a per-rank minimum would be visible in the gradients."""
    import season_nerf_amd as sn
    from season_nerf_amd import _lib, training, parallel
    from oracle import season_nerf_oracle as orc
    W, S, R = 64, 64, 16
    full = _rays(2 * R, 3)
    sol = _rays(2 * R, 4)
    sol["Bot"] = sol["Top"] - 2 * sol["Sun_Angle"] / sol["Sun_Angle"][:, 2:]
    cut = lambda d, i: {k: v[i * R:(i + 1) * R] for k, v in d.items()}

    def setup(solar):
        net = sn.T_NeRF(W, 4)
        sd = orc.init_weights(W, 4, 5)
        sd["G_NeRF_net.fc10Col.bias"] = sd["G_NeRF_net.fc10Col.bias"] - 1.5
        net.load_state_dict(sd)
        net = net.to("cuda").train()
        args = SimpleNamespace(n_samples=S, Use_Reg=True, Solar_Type_2=False, Use_MSE_loss=True, Use_Solar=True, sc_lambda=1.0, number_low_frequency_cases=4)
        ev = sn.All_in_One_Eval(args, torch.device("cuda"), 10, False, None, np.eye(4), np.zeros(3))
        ev.solar_creation_tool = lambda n, include_times=True: (solar["Top"], solar["Bot"], solar["Sun_Angle"], solar["Time_Encoded"], None)
        return net, ev

    def view_of(eng, ptr, count, is_double):
        off, size = ptr - eng.ws.data_ptr(), count * (8 if is_double else 4)
        assert 0 <= off and off + size <= eng.ws.numel()
        return eng.ws[off:off + size].view(torch.float64 if is_double else torch.float32)

    def run(net, ev, data, cb, world, gmin):
        eng = training._engine_for(net, data["Top"].shape[0], data["Top"].shape[0], S)
        fn = _lib.ALLREDUCE_FN(lambda user, ptr, count, dbl, stream: cb(view_of(eng, ptr, count, dbl)))
        _lib.check(_lib.lib().snerf_trainer_set_allreduce(eng.h, C.cast(fn, C.c_void_p), None, world), "set_allreduce")
        monkeypatch.setattr(parallel, "data_parallel", lambda group=None: True)
        monkeypatch.setattr(parallel, "global_min", gmin)
        torch.manual_seed(11)                                   # the jitter vectors of both passes: the same draw in every run
        loss = ev.get_loss(data, net, 0, True)
        n_fwd = cb.count()
        total = sum(v[0] * v[1] for v in loss.values())
        total.backward()
        torch.cuda.synchronize()
        _lib.check(_lib.lib().snerf_trainer_set_allreduce(eng.h, None, None, 1), "set_allreduce")
        return ({k: float(v[0]) for k, v in loss.items()}, {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}, n_fwd)

    class Recorder:
        def __init__(self):
            self.G = []

        def __call__(self, v):
            self.G.append(v.clone())
            return 0

        def count(self):
            return len(self.G)

    rec, mins = Recorder(), []
    net_f, ev_f = setup(sol)
    loss_f, grads_f, n_fwd = run(net_f, ev_f, full, rec, 1, lambda m, group=None: (mins.append(m.detach().clone()), (m.detach().clone(), 1))[1])
    G = rec.G
    assert n_fwd == 16 and len(G) == 24, (n_fwd, len(G))       # 8 BatchNorm layers: image forward, sun-ray forward, image backward
    assert len(mins) == 1 and float(mins[0].min()) < 0.2         # the albedo term is active
    assert loss_f["Albedo_Color"] > 0

    class Replay:
        def __init__(self):
            self.k = 0

        def __call__(self, v):
            v.copy_(G[self.k] * (2.0 if self.k >= n_fwd else 1.0))      # what the 2-rank sum-all-reduce returns (see the test above)
            self.k += 1
            return 0

        def count(self):
            return self.k

    results, owners = [], 0
    for r in range(2):
        net, ev = setup(cut(sol, r))

        def gmin(m, group=None):
            assert bool((m.detach() >= mins[0] - 1e-5).all())                  # a shard's minimum is not below the global one
            return torch.minimum(m.detach(), mins[0]), 2

        results.append(run(net, ev, cut(full, r), Replay(), 2, gmin))
    for k, v in loss_f.items():                                                 # every rank reports the global-batch value of a min term;
        mean = 0.5 * (results[0][0][k] + results[1][0][k])                      # mean-type terms average to it
        assert abs(mean - v) <= 2e-4 * max(abs(v), 1e-3), (k, mean, v, results[0][0][k], results[1][0][k])
    assert abs(results[0][0]["Albedo_Color"] - loss_f["Albedo_Color"]) <= 2e-4 * loss_f["Albedo_Color"]
    assert abs(results[1][0]["Albedo_Color"] - loss_f["Albedo_Color"]) <= 2e-4 * loss_f["Albedo_Color"]
    gmax = max(float(v.abs().max()) for v in grads_f.values())
    worst = 0.0
    for n, gf in grads_f.items():
        gs = 0.5 * (results[0][1][n] + results[1][1][n])
        worst = max(worst, float((gs - gf).abs().max()) / max(float(gf.abs().max()), 1e-3 * gmax))
    assert worst < 2e-3, worst
    # not vacuous: with a per-rank minimum (no exchange) the averaged gradients leave the full-batch ones
    per_rank = []
    for r in range(2):
        net, ev = setup(cut(sol, r))
        per_rank.append(run(net, ev, cut(full, r), Replay(), 2, lambda m, group=None: (m.detach().clone(), 1)))
    off = max(float((0.5 * (per_rank[0][1][n] + per_rank[1][1][n]) - gf).abs().max()) / max(float(gf.abs().max()), 1e-3 * gmax) for n, gf in grads_f.items())
    assert off > 5 * worst and off > 5e-3, (off, worst)
