"""Does hipGraphLaunch of a graph that is still running block the host?  Host time of each GraphedTrainStep call (no synchronisation in between) and of
graph.replay() alone, at the benchmark's size.   python3 tools/graph_replay_probe.py"""
import os, sys, time
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import season_nerf_amd as sn

dev = torch.device("cuda")
net = sn.T_NeRF(256, 4)
net.load_state_dict(sn.synthetic_state_dict(net, 0, bn_stats="identity"))
net = net.to(dev).train()
args = SimpleNamespace(n_samples=bench.S, Use_Reg=True, Solar_Type_2=False, Use_MSE_loss=True, Use_Solar=True, sc_lambda=0.03, number_low_frequency_cases=4)
WC, H4 = np.array([41.29, -95.9, 300.0]), np.array([[310.0, 12.0, 0.0, -11650.0], [-9.0, 240.0, 0.0, 23390.0], [0.0, 0.0, 0.01, -3.0], [0, 0, 0, 1.0]])
ev = sn.All_in_One_Eval(args, dev, 10, False, None, H4, WC)
d = bench.synth(0, dev)
d["GT_Color"] = torch.rand(bench.R, 3, device=dev)
tool = sn.Net_tool(net, ev, 1e-4, total_steps=200, writer=None)
step = sn.GraphedTrainStep(tool, d, warmup=2)
for k in range(6):
    step(d, k)
torch.cuda.synchronize()
ts = []
t_all = time.perf_counter()
for k in range(12):
    t0 = time.perf_counter()
    step(d, 6 + k)
    ts.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
print("host ms per GraphedTrainStep call, 12 calls back to back:", " ".join(f"{t:.2f}" for t in ts), f"| wall {1e3 * (time.perf_counter() - t_all) / 12:.2f} ms per step", flush=True)
ts = []
t_all = time.perf_counter()
for k in range(12):
    t0 = time.perf_counter()
    step.graph.replay()
    ts.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
print("host ms per bare graph.replay(), 12 back to back:        ", " ".join(f"{t:.2f}" for t in ts), f"| wall {1e3 * (time.perf_counter() - t_all) / 12:.2f} ms per step", flush=True)
