"""Host time of a training step by section, each step enqueued onto an IDLE GPU (configs[2], MSE):  python tools/host_sections.py"""
import os, sys, time
from types import SimpleNamespace
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(v, "1")
import numpy as np
import torch
import bench
import season_nerf_amd as sn
from season_nerf_amd import training, evaluator

dev = torch.device("cuda")
R, S = bench.R, bench.S
net = sn.T_NeRF(256, 4)
net.load_state_dict(sn.synthetic_state_dict(net, 0, bn_stats="identity"))
net = net.to(dev).train()
args = SimpleNamespace(n_samples=S, Use_Reg=True, Solar_Type_2=False, Use_MSE_loss=True, Use_Solar=True, sc_lambda=0.03, number_low_frequency_cases=4)
WC, H4 = np.array([41.29, -95.9, 300.0]), np.array([[310.0, 12.0, 0.0, -11650.0], [-9.0, 240.0, 0.0, 23390.0], [0.0, 0.0, 0.01, -3.0], [0, 0, 0, 1.0]])
ev = sn.All_in_One_Eval(args, dev, 10, False, None, H4, WC)
d = bench.synth(0, dev)
d["GT_Color"] = torch.rand(R, 3, device=dev)
tool = sn.Net_tool(net, ev, 10 ** -4.86, total_steps=100, lr_alpha_scale=1000.0, writer=None)
T = {}
def sec(name, t0):
    T.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
    return time.perf_counter()

orig_eval, orig_rho, orig_sun = ev.eval, ev.eval_Rho_Only, ev.solar_creation_tool
def w_eval(*a, **k):
    t0 = time.perf_counter(); r = orig_eval(*a, **k); sec("  get_loss: image pass (ev.eval)", t0); return r
def w_rho(*a, **k):
    t0 = time.perf_counter(); r = orig_rho(*a, **k); sec("  get_loss: sun-ray pass (eval_Rho_Only)", t0); return r
def w_sun(*a, **k):
    t0 = time.perf_counter(); r = orig_sun(*a, **k); sec("  get_loss: sun-ray generator (host numpy)", t0); return r
ev.eval, ev.eval_Rho_Only, ev.solar_creation_tool = w_eval, w_rho, w_sun
for i in range(25):
    torch.cuda.synchronize()
    t00 = t0 = time.perf_counter()
    tool.optim.zero_grad(); t0 = sec("zero_grad", t0)
    loss = ev.get_loss(d, net, 0, train_mode=True); t0 = sec("get_loss (all)", t0)
    total = loss.total() if getattr(loss, "vec", None) is not None else sum(v * w for v, w in loss.values()); t0 = sec("total", t0)
    total.backward(); t0 = sec("backward", t0)
    tool.optim.step(); t0 = sec("optim.step", t0)
    tool.sched.step(); t0 = sec("sched.step", t0)
    sec("STEP", t00)
torch.cuda.synchronize()
print("fused loss path:", getattr(loss, "vec", None) is not None)
for k, v in T.items():
    v = sorted(v[5:])
    print(f"{k:45s} median {v[len(v)//2]:.3f} ms   min {v[0]:.3f}")
