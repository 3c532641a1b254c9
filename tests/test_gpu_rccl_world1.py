"""GPU: the RCCL code path on the ONE GPU a test box has.  `bench.py` under `python -m torch.distributed.run --nproc-per-node=1`
initialises the "nccl" (= RCCL) process group and issues every collective of the data-parallel design with a world of one rank:
the grouped RGB-tile all-gather of the render workload, and - training with `--bn_sync global --loss barron` - the flat
gradient-arena all-reduce, the adaptive-loss-parameter all-reduce and the BatchNorm-statistics all-reduces the engine calls back
for.  It proves NOTHING about scaling (no second rank, no xGMI traffic); it turns "RCCL never initialised anywhere" into
"initialised, stream-ordered, counted and timed at world 1", and gives the latency the 24+ small collectives of global-batch
BatchNorm add to a step.  Each run is a fresh child process (never a re-launch of this one, which owns the GPU)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, timeout=900):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(REPO, "bench.py"), "--gpus", "1"] + extra
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_render_tile_gather_over_rccl_world1():
    d = _run(["--steps", "16", "--warmup", "8", "--prewarm", "0", "--no-cpu-baseline", "--no-sweep", "--no-train"])
    assert d["n_gpus"] == 1 and d["value"] > 1e7
    c = d["collectives"]
    # 8 warm-up steps = one full tile group, 16 timed steps = two: three asynchronous all-gathers
    assert c["tile_group_all_gather"] == 3, c
    print(f"  render under torch.distributed.run, world 1: {d['ms_per_step']:.3f} ms per step, collectives {c}")


def test_training_collectives_over_rccl_world1():
    local = _run(["--workload", "train", "--steps", "4", "--warmup", "2", "--loss", "barron", "--no-cpu-baseline"])
    glob = _run(["--workload", "train", "--steps", "4", "--warmup", "2", "--loss", "barron", "--bn_sync", "global", "--no-cpu-baseline"])
    cl, cg = local["collectives"], glob["collectives"]
    steps = 4 + 2 + 6          # timed + warm-up + the six steps bench.py enqueues onto an idle GPU afterwards (host_enqueue_ms_per_step)
    for c in (cl, cg):
        assert c["grad_arena_all_reduce"] == steps and c["ada_loss_all_reduce"] == steps, c
        assert c["albedo_min_all_reduce"] == steps, c          # the batch-wide minimum of get_loss's Albedo_Color term (Eval_Tools_2.py:374)
    assert "bn_stats_all_reduce" not in cl
    # global-batch BatchNorm: 8 BatchNorm layers x (2 passes forward + backward sums) - enabled after the engine's first step
    per_step = cg["bn_stats_all_reduce"] / (steps - 1)
    assert per_step >= 24 and per_step == int(per_step), cg
    extra = glob["ms_per_step"] - local["ms_per_step"]
    print(f"  training under torch.distributed.run, world 1: {local['ms_per_step']:.2f} ms per step with per-rank BatchNorm, "
          f"{glob['ms_per_step']:.2f} ms with {per_step:.0f} BatchNorm-statistics all-reduces per step "
          f"({extra * 1e3 / per_step:.0f} us per collective, latency only: world of one rank)")
    # one rank: the global batch IS the local batch (six Barron-loss steps with atomically reduced gradients: a few 1e-3 apart)
    assert glob["final_loss"] == pytest.approx(local["final_loss"], rel=2e-2)
