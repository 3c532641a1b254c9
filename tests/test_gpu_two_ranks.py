"""Data-parallel training and rendering with REAL RANKS (two, four) on the one GPU of a test box.

RCCL cannot put two ranks on one device, gloo can: two processes share cuda:0, their collectives - the same `torch.distributed` calls the RCCL path
issues (flat gradient-arena all-reduce in `FusedAdam.step`, the BatchNorm-statistics all-reduces of `sync_batchnorm`, the `Albedo_Color` MIN all-reduce
of `get_loss`) - run over gloo on device tensors.  Each rank trains on half of the rays; checked against the SAME step on the whole batch in one process
(what the single-process reference computes: Eval_Tools_2.py:340-459, mg_run_NeRF.py:288-326): the loss dict, the averaged gradients, the BatchNorm
running statistics, and the Adam update.  This is the N > 1 path end to end - processes, process group, collectives, averaging - without a second GPU;
what it cannot show is RCCL's transport and any scaling number."""
import os
import tempfile
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

W, S, R, LR = 64, 48, 24, 1e-4            # R rays per rank: 1152 points per rank, the bf16x3 row kernels are in play
WC = np.array([41.29, -95.9, 300.0])
H4 = np.array([[310.0, 12.0, 0.0, -11650.0], [-9.0, 240.0, 0.0, 23390.0], [0.0, 0.0, 0.01, -3.0], [0, 0, 0, 1.0]])


def _rays(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    t = lambda a: torch.tensor(a, dtype=torch.float32)
    sun = rng.uniform(0.1, 1, (n, 3))
    sun /= np.linalg.norm(sun, axis=1, keepdims=True)
    tau = rng.uniform(0, 1, (n, 2))
    top = np.concatenate([rng.uniform(-1, 1, (n, 2)), np.ones((n, 1))], 1)
    return {"Top": t(top), "Bot": t(np.concatenate([rng.uniform(-1, 1, (n, 2)), -np.ones((n, 1))], 1)), "Sun_Angle": t(sun),
            "Time_Encoded": t(np.stack([np.cos(6.28 * tau[:, 0]), np.sin(6.28 * tau[:, 0]), np.cos(6.28 * tau[:, 1]), np.sin(6.28 * tau[:, 1])], 1)),
            "GT_Color": t(rng.uniform(0, 1, (n, 3)))}


def _problem():
    full, sol = _rays(2 * R, 3), _rays(2 * R, 4)
    sol["Bot"] = sol["Top"] - 2 * sol["Sun_Angle"] / sol["Sun_Angle"][:, 2:]
    return full, sol


def _one_step(rows, group_sync):
    """One `Net_tool.train_step` on rays `rows` of the problem; returns what the comparison needs (CPU tensors / floats)."""
    import season_nerf_amd as sn
    from season_nerf_amd import training
    from oracle import season_nerf_oracle as orc
    full, sol = _problem()
    data = {k: v[rows] for k, v in full.items()}
    sun = {k: v[rows] for k, v in sol.items()}
    sd = orc.init_weights(W, 4, 5, bn_stats="identity")
    sd["G_NeRF_net.fc10Col.bias"] = sd["G_NeRF_net.fc10Col.bias"] - 1.5          # dark albedo: the Albedo_Color hinge is active
    net = sn.T_NeRF(W, 4)
    net.load_state_dict(sd)
    net = net.to("cuda").train()
    args = SimpleNamespace(n_samples=S, Use_Reg=True, Solar_Type_2=False, Use_MSE_loss=True, Use_Solar=True, sc_lambda=0.5, number_low_frequency_cases=4)
    ev = sn.All_in_One_Eval(args, torch.device("cuda"), 10, False, None, H4, WC)
    ev.solar_creation_tool = lambda n, include_times=True: (sun["Top"], sun["Bot"], sun["Sun_Angle"], sun["Time_Encoded"], None)
    n = data["Top"].shape[0]
    if group_sync:
        training._engine_for(net, n, n, S).sync_batchnorm(True)
    tool = sn.Net_tool(net, ev, LR, total_steps=3, writer=None)
    p0 = {k: v.detach().clone() for k, v in net.named_parameters()}
    torch.manual_seed(11)                                                         # the jitter vectors of both passes: the same draw everywhere
    loss = tool.train_step(data, 0)
    torch.cuda.synchronize()
    return {"loss": {k: float(v[0]) for k, v in loss.items()},
            "grads": {k: p.grad.detach().cpu().clone() for k, p in net.named_parameters() if p.grad is not None},      # after the arena all-reduce + averaging
            "update": {k: (p.detach() - p0[k]).cpu() for k, p in net.named_parameters()},
            "bn": {k: v.detach().cpu().clone() for k, v in net.state_dict().items() if "running_" in k},
            "collectives": dict(sn.parallel.COLLECTIVES)}


def _rank(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    res = _one_step(slice(rank * R, (rank + 1) * R), group_sync=True)
    torch.save(res, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_real_ranks_reproduce_the_full_batch_step():
    import torch.multiprocessing as mp
    ref = _one_step(slice(0, 2 * R), group_sync=False)                            # the whole batch in this process: no process group
    assert ref["loss"]["Albedo_Color"] > 0 and not ref["collectives"]
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_rank, args=(2, 29500 + (os.getpid() + 4321) % 2000, d), nprocs=2, join=True)
        ranks = [torch.load(os.path.join(d, f"rank{r}.pt")) for r in range(2)]
    for r in ranks:                                                               # every exchange of the design ran, once per step / layer
        c = r["collectives"]
        assert c["grad_arena_all_reduce"] == 1 and c["albedo_min_all_reduce"] == 1 and c["bn_stats_all_reduce"] == 24, c
    # loss: a min term is reported as its global-batch value by every rank; mean-type terms average to the full-batch value
    for k, v in ref["loss"].items():
        mean = 0.5 * (ranks[0]["loss"][k] + ranks[1]["loss"][k])
        assert abs(mean - v) <= 3e-4 * max(abs(v), 1e-3), (k, mean, v)
    for r in ranks:
        assert abs(r["loss"]["Albedo_Color"] - ref["loss"]["Albedo_Color"]) <= 3e-4 * ref["loss"]["Albedo_Color"]
    # gradients: after the all-reduce both ranks hold the SAME arena, and it is the full-batch gradient
    gmax = max(float(v.abs().max()) for v in ref["grads"].values())
    worst = 0.0
    for k, g in ref["grads"].items():
        assert torch.equal(ranks[0]["grads"][k], ranks[1]["grads"][k]), k
        worst = max(worst, float((ranks[0]["grads"][k] - g).abs().max()) / max(float(g.abs().max()), 1e-3 * gmax))
    print(f"  two gloo ranks on one GPU vs the full batch: worst relative gradient error {worst:.2e}")
    assert worst < 2e-3, worst
    for k, v in ref["bn"].items():                                                # global-batch BatchNorm: the running statistics of the whole batch
        np.testing.assert_allclose(ranks[0]["bn"][k].numpy(), v.numpy(), rtol=2e-4, atol=2e-6, err_msg=k)
        assert torch.equal(ranks[0]["bn"][k], ranks[1]["bn"][k]), k
    # Adam: identical gradients -> identical updates on both ranks; against the full batch where the gradient is not rounding noise
    moved = diff = 0.0
    for k, u in ref["update"].items():
        assert torch.equal(ranks[0]["update"][k], ranks[1]["update"][k]), k
        big = ref["grads"][k].abs() > 1e-3 * gmax if k in ref["grads"] else torch.zeros_like(u, dtype=torch.bool)
        moved += float(u[big].abs().sum())
        diff += float((ranks[0]["update"][k] - u)[big].abs().sum())
    assert moved > 0 and diff < 0.02 * moved, (diff, moved)


def _render_rank(rank, world, port, out_dir):
    import torch.distributed as dist
    import season_nerf_amd as sn
    from oracle import season_nerf_oracle as orc
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    net = sn.T_NeRF(W, 4)
    net.load_state_dict(orc.init_weights(W, 4, 2))
    net = net.to("cuda").eval()
    img = sn.render_season_sweep(net, (80, 0), (30, 90), [k / 12.0 for k in range(12)], (23, 19, 48), WC, H4, torch.device("cuda"), sharded=True)
    # the per-step tile exchange of the benchmark's render loop: every rank's [R, 3] tiles of 5 steps, 2 steps per asynchronous collective
    tg = sn.parallel.TileGroupGather((7, 3), group=2, device=torch.device("cuda"))
    for step in range(5):
        tg.slot().copy_(torch.full((7, 3), float(10 * rank + step), device="cuda"))
        tg.commit()
    tg.flush()
    last = torch.stack([tg.gathered(s_) for s_ in (3, 4)]).cpu()                    # [2 steps, world, 7, 3]
    torch.save({"img": img.cpu(), "tiles": last, "collectives": dict(sn.parallel.COLLECTIVES)}, os.path.join(out_dir, f"render{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_real_ranks_render_tiles_and_gather():
    """BASELINE configs[4]'s multi-GPU leg with two real ranks (gloo, one GPU): `render_season_sweep(..., sharded=True)` - every rank renders its block of
    the ray grid, the [rays, T, 3] tiles are all-gathered - gives every rank the image a single process renders, bit for bit; and the grouped asynchronous
    tile gather of the benchmark loop (`parallel.TileGroupGather`) delivers every rank's tiles."""
    import torch.multiprocessing as mp
    import season_nerf_amd as sn
    from oracle import season_nerf_oracle as orc
    net = sn.T_NeRF(W, 4)
    net.load_state_dict(orc.init_weights(W, 4, 2))
    net = net.to("cuda").eval()
    ref = sn.render_season_sweep(net, (80, 0), (30, 90), [k / 12.0 for k in range(12)], (23, 19, 48), WC, H4, torch.device("cuda")).cpu()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_render_rank, args=(2, 29500 + (os.getpid() + 5432) % 2000, d), nprocs=2, join=True)
        ranks = [torch.load(os.path.join(d, f"render{r}.pt")) for r in range(2)]
    for r in ranks:
        assert tuple(r["img"].shape) == (12, 23, 19, 3) and torch.equal(r["img"], ref)          # 437 rays: shards of 219 and 218
        assert r["collectives"]["rows_all_gather"] == 1 and r["collectives"]["tile_group_all_gather"] == 3
        for i, step in enumerate((3, 4)):
            for src in range(2):
                assert bool((r["tiles"][i, src] == float(10 * src + step)).all()), (step, src)


def _bench_two_ranks(extra, world=2):
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...` - the command the driver runs for its scaling curve - with both ranks on
    device 0 and gloo instead of RCCL (SNERF_BENCH_DEVICE / SNERF_BENCH_BACKEND): the N > 1 code of bench.py itself, executed with two real processes."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SNERF_BENCH_DEVICE="0", SNERF_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port",
           str(29500 + (os.getpid() + 6543 + world) % 2000), os.path.join(repo, "bench.py"), "--gpus", str(world)] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("world", [2, 4])      # (4 ranks + this process: within the 6 processes a test box lets onto its GPU)
def test_bench_render_with_real_ranks(world):
    d = _bench_two_ranks(["--steps", "16", "--warmup", "8", "--prewarm", "0", "--no-cpu-baseline", "--no-sweep", "--no-train"], world)
    assert d["n_gpus"] == world and d["scaling"] == "weak" and d["value"] > 1e7
    assert d["collectives"]["tile_group_all_gather"] == 3                       # 8 warm-up steps = one tile group, 16 timed = two
    print(f"  bench.py --gpus {world}, all ranks on one GPU over gloo: {d['value']:.3e} ray-samples/s (the ranks SHARE the GPU: not a scaling number)")


@pytest.mark.parametrize("world,loss", [(2, "mse"), (4, "mse"), (2, "barron")])
def test_bench_training_with_real_ranks(world, loss):
    d = _bench_two_ranks(["--workload", "train", "--steps", "3", "--warmup", "2", "--bn_sync", "global", "--loss", loss, "--no-cpu-baseline"], world)
    steps = 3 + 2 + 6
    c = d["collectives"]
    if loss == "barron":                     # the loss object's alpha / scale gradients travel as one small all-reduce per step (trainer._allreduce_mean_grads)
        assert c["ada_loss_all_reduce"] == steps, c
    assert d["n_gpus"] == world and c["grad_arena_all_reduce"] == steps and c["albedo_min_all_reduce"] == steps and c["bn_stats_all_reduce"] == 24 * (steps - 1), c
    # (the Barron total carries the logged-only entries with weight 1 and solar weights divided by scale^2, Eval_Tools_2.py:428-444: hundreds by construction)
    assert np.isfinite(d["final_loss"]) and 0 < d["final_loss"] < (10 if loss == "mse" else 1e4)
