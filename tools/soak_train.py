"""Soak: 1000 captured + 300 eager training steps at configs[2]'s size on the synthetic scene of tools/make_trained_golden.py-like random data: finite losses,
a falling colour loss, no growth of device memory.   python tools/soak_train.py"""
import os, sys, time
from types import SimpleNamespace
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(v, "1")
import numpy as np
import torch
import bench
import season_nerf_amd as sn

dev = torch.device("cuda")
R, S = bench.R, bench.S
net = sn.T_NeRF(int(os.environ.get("SOAK_WIDTH", "256")), 4)      # SOAK_WIDTH=512: the reference's default width (tools/power_under.py runs both)
net.load_state_dict(sn.synthetic_state_dict(net, 0, bn_stats="identity"))
net = net.to(dev).train()
args = SimpleNamespace(n_samples=S, Use_Reg=True, Solar_Type_2=False, Use_MSE_loss=True, Use_Solar=True, sc_lambda=0.03, number_low_frequency_cases=4)
WC, H4 = np.array([41.29, -95.9, 300.0]), np.array([[310.0, 12.0, 0.0, -11650.0], [-9.0, 240.0, 0.0, 23390.0], [0.0, 0.0, 0.01, -3.0], [0, 0, 0, 1.0]])
ev = sn.All_in_One_Eval(args, dev, 10, False, None, H4, WC)
d = bench.synth(0, dev)
# a learnable target: colour = a smooth function of where the ray hits z = 0
mid = 0.5 * (d["Top"] + d["Bot"])
d["GT_Color"] = torch.stack([0.5 + 0.4 * torch.sin(3 * mid[:, 0]), 0.5 + 0.4 * torch.cos(2 * mid[:, 1]), 0.5 + 0.3 * torch.sin(mid[:, 0] + mid[:, 1])], 1)
n_g, n_e = int(os.environ.get("SOAK_GRAPHED", "1000")), int(os.environ.get("SOAK_EAGER", "300"))
tool = sn.Net_tool(net, ev, 3e-4, total_steps=n_g + n_e + 8, writer=None)
step = sn.GraphedTrainStep(tool, d, warmup=3)
mem, col = [], []
t0 = time.perf_counter()
for k in range(n_g):
    loss = step(d, k)
    if k % 100 == 0 or k == n_g - 1:
        col.append(float(loss["Color"][0])); mem.append(torch.cuda.memory_allocated() / 2**20)
        print(f"graph step {k:4d}  colour {col[-1]:.5f}  total {float(loss.total()):.5f}  mem {mem[-1]:.0f} MiB  {time.perf_counter() - t0:.1f} s", flush=True)
for k in range(n_e):
    loss = tool.train_step(d, n_g + k)
    if k % 100 == 0 or k == n_e - 1:
        col.append(float(loss["Color"][0])); mem.append(torch.cuda.memory_allocated() / 2**20)
        print(f"eager step {k:4d}  colour {col[-1]:.5f}  mem {mem[-1]:.0f} MiB", flush=True)
torch.cuda.synchronize()
assert all(np.isfinite(col)), col
assert col[-1] < 0.5 * col[0], (col[0], col[-1])
assert max(mem[2:]) - min(mem[2:]) < 64, mem
net.eval()
with torch.no_grad():
    out = ev.eval(d, net, 0, False)
print("eval after training: RGB finite", bool(torch.isfinite(out["Rendered_Col"]).all()), " precision ->", net.resolved_precision, " rgb_pred", net.i8_estimate()["rgb_pred"],
      " MSE vs target", float(((out["Rendered_Col"] - d["GT_Color"]) ** 2).mean()))
print("SOAK OK")
