"""The captured training step (trainer.GraphedTrainStep) against the failure family of rounds 4-6.

Cause, found in round 6: a hipMemcpyAsync / hipMemsetAsync inside the captured step becomes a hipGraph MEMCPY / MEMSET node, and on ROCm 7.2 replays of
a graph that holds such nodes compute garbage once eager work of the same process has run between them (the runtime's AQL packet capture of graph nodes;
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 hides it).  The engine's copies and fills are kernels since (csrc/train.cpp snerf_copy_async / snerf_zero_async), and
the loss terms avoid the one torch op whose backward copies through the runtime (`x ** 2`, training._sq).  Two guards:
  * the captured graphs of every phase hold KERNEL nodes only (the runtime's own DOT print, in a child process: the switch is read at HIP start-up);
  * the state of a captured driver run - gradient arena, parameters, Adam's moments - follows the eager run step by step through save points and the
    phase switch (NaN-aware: the losses the older test compares stay plausible for a while when a layer's gradients are garbage, Adam normalises them)."""
import glob
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from test_net_tool import _args

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WC, H4 = np.array([41.29, -95.9, 300.0]), np.array([[310.0, 12.0, 0.0, -11650.0], [-9.0, 240.0, 0.0, 23390.0], [0.0, 0.0, 0.01, -3.0], [0, 0, 0, 1.0]])


def _data(R=48):
    rng = np.random.Generator(np.random.PCG64(3))
    hm = rng.uniform(-0.8, 0.6, (24, 24))
    t = lambda a: torch.tensor(a, dtype=torch.float32)
    data = {"Top": t(np.concatenate([rng.uniform(-1, 1, (R, 2)), np.ones((R, 1))], 1)), "Bot": t(np.concatenate([rng.uniform(-1, 1, (R, 2)), -np.ones((R, 1))], 1)),
            "Sun_Angle": torch.nn.functional.normalize(t(rng.uniform(0.1, 1, (R, 3))), dim=1), "Time_Encoded": t(rng.uniform(-1, 1, (R, 4))),
            "GT_Color": t(rng.uniform(0, 1, (R, 3)))}
    return hm, data


CHILD = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch
import season_nerf_amd as sn
from oracle import season_nerf_oracle as orc            # initial weights only
from test_gpu_graph_nodes import _data, WC, H4
from test_net_tool import _args
hm, data = _data()
for use_mse in (%s,):
    tool = sn.T_NeRF_Net_Tool(_args(20, n_saves=3, use_mse=use_mse), hm, hm, "cuda", H4, WC, get_data=lambda eval_mode: data, use_graph=True)
    tool.network.load_state_dict(orc.init_weights(64, 4, 1))
    n = 0
    for s_ in range(20):
        tool.step()
        n += int(tool._graphed is not None and tool._graphed.graph is not None)
    torch.cuda.synchronize()
    print("replayed", n, flush=True)
"""


@pytest.mark.gpu
@pytest.mark.parametrize("use_mse", [True, False])
def test_captured_steps_hold_kernel_nodes_only(tmp_path, use_mse):
    """Two captures per run - the DSM-prior phase and the free phase - with the MSE colour loss and with Barron's adaptive loss, printed by the runtime
    (DEBUG_HIP_GRAPH_DOT_PRINT): hundreds of kernel nodes, not one memory-operation node."""
    env = dict(os.environ, DEBUG_HIP_GRAPH_DOT_PRINT="1")
    env.pop("SNERF_TRAIN_MEMOPS", None)
    r = subprocess.run([sys.executable, "-c", CHILD % (REPO, REPO, use_mse)], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert re.findall(r"replayed (\d+)", r.stdout) == ["16"], r.stdout          # 20 steps, 2 phases x 2 eager warm-up steps, the other 16 replayed
    dots = sorted(glob.glob(os.path.join(str(tmp_path), "graph_*dot_print*")))
    assert len(dots) == 2, (dots, r.stderr[-500:])
    for d in dots:
        s = open(d).read()
        labels = re.findall(r'label="\d+\n([^\n"]*)', s)
        assert len(labels) > 100, (d, len(labels))
        memops = [l for l in labels if re.search(r"memcpy|memset", l, re.I)]
        assert not memops, f"{os.path.basename(d)}: {len(memops)} memory-operation nodes among {len(labels)}: {sorted(set(memops))}"
        assert any("snerf" in l for l in labels)                                      # (the engine's kernels are in there: this is the step, not a stub)


@pytest.mark.gpu
def test_captured_driver_state_follows_the_eager_run_step_by_step():
    """20 steps (prior phase, phase switch, 5 save points with their eager validation between replays), eager and captured from the same seeds: after EVERY step
    the gradient arena, the parameters and both Adam moments agree to the noise of the float atomics (1e-3 of the tensor's largest element); a non-finite
    value anywhere is a failure.  With SNERF_TRAIN_MEMOPS=1 this fails at the first replay after a save point, 9 runs of 9 (tools/graph_wait_probe4.py)."""
    import season_nerf_amd as sn
    from oracle import season_nerf_oracle as orc
    hm, data = _data()

    def run(use_graph):
        tool = sn.T_NeRF_Net_Tool(_args(20, n_saves=5, use_mse=True), hm, hm, "cuda", H4, WC, get_data=lambda eval_mode: data, use_graph=use_graph)
        tool.network.load_state_dict(orc.init_weights(64, 4, 1))
        np.random.seed(3); torch.manual_seed(3)
        snaps = []
        for _ in range(20):
            tool.step()
            torch.cuda.synchronize()
            st = tool.network._param_store
            snaps.append({"grads": st.grads.detach().cpu().clone(), "params": st.params.detach().cpu().clone(), "m": st.adam_m.cpu().clone(), "v": st.adam_v.cpu().clone()})
        return snaps

    a, b = run(False), run(True)
    for s_ in range(20):
        for key in ("grads", "params", "m", "v"):
            x, y = a[s_][key], b[s_][key]
            assert bool(torch.isfinite(y).all()), f"step {s_}: non-finite values in the captured run's {key}"
            scale = float(x.abs().max()) + 1e-30
            d = float((x - y).abs().max())
            assert d <= 1e-3 * scale, f"step {s_}: {key} differs by {d:.3e} (largest element {scale:.3e})"
