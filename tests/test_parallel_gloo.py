"""world_size-2 gloo test (CPU) of the N>1 path: ray sharding, unequal-block all-gather, ShardedEval plumbing.
The per-shard compute is a stand-in callable here (the HIP path needs a GPU); what is tested is that sharded
evaluation + gather equals the unsharded evaluation for ragged sizes."""
import importlib.util
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_parallel():
    spec = importlib.util.spec_from_file_location("snerf_parallel", os.path.join(REPO, "season_nerf_amd", "parallel.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class FakeEval:
    def eval(self, d, net, step, train_mode):
        rgb = torch.sigmoid(d["Top"] * 2 - d["Bot"] + d["Sun_Angle"] * d["Time_Encoded"][:, :3])
        return {"Rendered_Col": rgb, "Albedo_Color": rgb * 0.5}


def _worker(rank, world, port, n):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    par = _load_parallel()
    g = torch.Generator().manual_seed(0)
    data = {"Top": torch.rand(n, 3, generator=g), "Bot": torch.rand(n, 3, generator=g),
            "Sun_Angle": torch.rand(n, 3, generator=g), "Time_Encoded": torch.rand(n, 4, generator=g), "S": 7}
    full = FakeEval().eval(data, None, 0, False)
    sh = par.ShardedEval(FakeEval(), None, keys=("Rendered_Col", "Albedo_Color")).eval(data, 0, False)
    assert torch.equal(sh["Rendered_Col"], full["Rendered_Col"]) and torch.equal(sh["Albedo_Color"], full["Albedo_Color"])
    lo, hi = par.shard_bounds(n, world)[rank]
    assert par.shard_dict(data, rank, world)["Top"].shape[0] == hi - lo and par.shard_dict(data, rank, world)["S"] == 7
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [1, 5, 64, 4097])
def test_sharded_eval_matches_unsharded(n):
    port = 29500 + (os.getpid() + n) % 2000
    mp.spawn(_worker, args=(2, port, n), nprocs=2, join=True)


def test_shard_bounds():
    par = _load_parallel()
    for n in (0, 1, 7, 8, 4096):
        for w in (1, 2, 3, 8):
            b = par.shard_bounds(n, w)
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def _grad_worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    par = _load_parallel()
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    net[1].bias.requires_grad_(True)
    x = torch.arange(20, dtype=torch.float32).reshape(4, 5) * (rank + 1)
    net(x).sum().backward()
    if rank == 1:
        net[1].bias.grad = None                      # a parameter without gradient on one rank
    local = [None if p.grad is None else p.grad.clone() for p in net.parameters()]
    par.allreduce_gradients(net)
    # expected: average of the two ranks' gradients (rank r input is (r+1) * x0 -> gradients known analytically)
    gathered = [torch.zeros_like(p) if g is None else g for p, g in zip(net.parameters(), local)]
    for p, g in zip(net.parameters(), gathered):
        both = [torch.zeros_like(g) for _ in range(world)]
        dist.all_gather(both, g)
        assert torch.allclose(p.grad, sum(both) / world, rtol=1e-6, atol=1e-6)
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_gradients_two_ranks():
    mp.spawn(_grad_worker, args=(2, 29500 + (os.getpid() + 777) % 2000), nprocs=2, join=True)


def _tile_worker(rank, world, port, steps, group):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    par = _load_parallel()
    tg = par.TileGroupGather((5, 3), group=group, device="cpu")
    for i in range(steps):
        tg.slot().fill_(100.0 * rank + i)          # "the kernel of step i writes its tile"
        tg.commit()
    tg.flush()
    first_valid = max(0, (steps - 1) // group - 1) * group      # the last two tile groups are still held
    for k in range(first_valid, steps):
        got = tg.gathered(k)
        assert got.shape == (world, 5, 3)
        for r in range(world):
            assert torch.all(got[r] == 100.0 * r + k), (k, r, got[r][0])
    tg.reset()                                       # reusable after a reset (warm-up -> timed region in bench.py)
    tg.slot().fill_(7.0 + rank)
    tg.commit()
    tg.flush()
    assert torch.all(tg.gathered(0)[1 - rank] == 7.0 + (1 - rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("steps,group", [(8, 3), (6, 3), (1, 8), (17, 8)])
def test_tile_group_gather_two_ranks(steps, group):
    """The grouped, double-buffered asynchronous tile all-gather that bench.py's N > 1 render path uses."""
    port = 31500 + (os.getpid() + 7 * steps + group) % 2000
    mp.spawn(_tile_worker, args=(2, port, steps, group), nprocs=2, join=True)


def test_bench_self_launches_ranks():
    """`python bench.py --gpus 2` with no rank environment starts its own ranks (a child torch.distributed.run, before any GPU
    call) - VERDICT r1 item 2.  Exercised here with the gloo self-test backend: no GPU, no kernels, the real launch path."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                      # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rank_sum"] == 3.0 and out["self_launched"] is True


def _ada_worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = importlib.util.spec_from_file_location("season_nerf_amd", os.path.join(REPO, "season_nerf_amd", "__init__.py"),
                                                  submodule_search_locations=[os.path.join(REPO, "season_nerf_amd")])
    import sys
    pkg = importlib.util.module_from_spec(spec)
    sys.modules["season_nerf_amd"] = pkg
    spec.loader.exec_module(pkg)
    from season_nerf_amd.trainer import _allreduce_mean_grads
    ada = pkg.AdaptiveLossFunction(3, torch.float32, "cpu", alpha_hi=2.99, alpha_init=2.0, scale_init=0.03, scale_lo=0.01)
    opt = torch.optim.Adam(ada.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(100 + rank)                     # every rank sees different residuals (its own ray shard)
    for _ in range(3):
        opt.zero_grad()
        torch.mean(ada.lossfun(torch.randn(64, 3, generator=g) * 0.1)).backward()
        _allreduce_mean_grads(list(ada.parameters()))
        opt.step()
    flat = torch.cat([p.detach().reshape(-1) for p in ada.parameters()])
    both = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    assert torch.equal(both[0], both[1]), "adaptive-loss parameters diverged across ranks"
    dist.barrier()
    dist.destroy_process_group()


def test_adaptive_loss_parameters_stay_identical_across_ranks():
    """the second Adam (loss alpha / scale) must see all-reduced gradients under data parallelism, or the replicas
    optimise different objectives.  Net_tool.train_step calls `_allreduce_mean_grads` before `optim2.step()`."""
    port = 29500 + (os.getpid() + 77) % 2000
    mp.spawn(_ada_worker, args=(2, port), nprocs=2, join=True)


def _albedo_worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = importlib.util.spec_from_file_location("season_nerf_amd", os.path.join(REPO, "season_nerf_amd", "__init__.py"),
                                                  submodule_search_locations=[os.path.join(REPO, "season_nerf_amd")])
    import sys
    pkg = importlib.util.module_from_spec(spec)
    sys.modules["season_nerf_amd"] = pkg
    spec.loader.exec_module(pkg)
    from season_nerf_amd.training import albedo_min_loss
    from season_nerf_amd import parallel
    g = torch.Generator().manual_seed(5)
    full = 0.25 + torch.rand(16, 3, generator=g) * 0.5
    full[3, 0], full[12, 1] = 0.05, 0.02                              # minima on rank 0 and on rank 1; the third channel stays above the hinge
    n = full.shape[0] // world
    before = parallel.COLLECTIVES["albedo_min_all_reduce"]
    # the single-process reference: minimum over the whole batch, divided by the whole batch's ray count (Eval_Tools_2.py:374-379)
    ref_in = full.clone().requires_grad_(True)
    a = ref_in.min(0)[0]
    ref = torch.sum(torch.where(a < .2, (1 - a / .2) ** 2, torch.zeros_like(a))) / full.shape[0]
    ref.backward()
    mine = full[rank * n:(rank + 1) * n].clone().requires_grad_(True)
    loss = albedo_min_loss(mine)
    loss.backward()
    assert parallel.COLLECTIVES["albedo_min_all_reduce"] == before + 1
    assert torch.allclose(loss.detach(), ref.detach(), rtol=1e-6), (loss, ref)          # every rank reports the global-batch value
    grads = [torch.zeros_like(mine.grad) for _ in range(world)]
    dist.all_gather(grads, mine.grad)
    avg = torch.cat(grads) / world                                                       # what the gradient all-reduce (mean) leaves
    assert torch.allclose(avg, ref_in.grad, rtol=1e-6, atol=1e-9), (avg, ref_in.grad)
    assert float(ref_in.grad.abs().sum()) > 0
    dist.barrier()
    dist.destroy_process_group()


def test_albedo_minimum_is_taken_over_the_global_batch():
    """`Albedo_Color` (Eval_Tools_2.py:374-379) takes its minimum over the whole batch; under data parallelism that is
    one MIN all-reduce of 3 floats, the global ray count in the denominator, and the gradient on the rank that owns the minimum."""
    port = 29500 + (os.getpid() + 1277) % 2000
    mp.spawn(_albedo_worker, args=(2, port), nprocs=2, join=True)
