"""GPU: WHAT between two replays of the captured step breaks the next replay?  The driver runs with no save point before step 19 (N_SAVES=2: clean, see
graph_wait_probe4.py) and INJECT names the eager work placed between the steps:
  none | torch (3000 small torch kernels) | malloc (hipMalloc + hipMemcpy + hipFree through a C-ABI model build) | ours (eval-mode fused forward: re-pack + our kernels)
  | ours_nopack (our kernels on a second, fixed network: no hipMalloc/hipFree) | sync (device synchronize only)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("SNERF_GRAPH_PREPARE", "0")
import numpy as np, torch
import season_nerf_amd as sn
from oracle import season_nerf_oracle as orc
from tests.test_net_tool import _args

INJECT = os.environ.get("INJECT", "none")
rng = np.random.Generator(np.random.PCG64(3))
hm = rng.uniform(-0.8, 0.6, (24, 24))
R = 48
t = lambda a: torch.tensor(a, dtype=torch.float32)
data = {"Top": t(np.concatenate([rng.uniform(-1, 1, (R, 2)), np.ones((R, 1))], 1)), "Bot": t(np.concatenate([rng.uniform(-1, 1, (R, 2)), -np.ones((R, 1))], 1)),
        "Sun_Angle": torch.nn.functional.normalize(t(rng.uniform(0.1, 1, (R, 3))), dim=1), "Time_Encoded": t(rng.uniform(-1, 1, (R, 4))), "GT_Color": t(rng.uniform(0, 1, (R, 3)))}
WC, H4 = np.array([41.29, -95.9, 300.0]), np.array([[310.0, 12.0, 0.0, -11650.0], [-9.0, 240.0, 0.0, 23390.0], [0.0, 0.0, 0.01, -3.0], [0, 0, 0, 1.0]])
other = sn.T_NeRF(64, 4)
other.load_state_dict(orc.init_weights(64, 4, 5))
other = other.cuda().eval()
X = torch.rand(4096, 3, device="cuda") * 2 - 1
sun = torch.nn.functional.normalize(torch.rand(4096, 3, device="cuda"), dim=1)
tim = torch.rand(4096, 4, device="cuda")
other.forward(X, sun, tim)                       # packed once, before anything is captured


KEEP = []


def inject(tool):
    if INJECT == "torch":
        z = torch.zeros(1024, device="cuda")
        for _ in range(3000):
            z = z + 1.0
    elif INJECT == "malloc":
        m = sn.T_NeRF(64, 4)
        m.load_state_dict(orc.init_weights(64, 4, 7))
        m = m.cuda().eval()
        m.device_model()
        del m
    elif INJECT == "ours":
        net = tool.network
        was = net.training
        net.eval()
        with torch.no_grad():
            net.forward(X, sun, tim)
        net.train(was)
    elif INJECT == "ours_nopack":
        with torch.no_grad():
            for _ in range(int(os.environ.get("INJECT_N", "20"))):
                other.forward(X, sun, tim)
    elif INJECT == "free":                        # hipMalloc + hipFree of 1 MiB straight through the runtime, nothing else
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")
        p_ = C.c_void_p()
        assert hip.hipMalloc(C.byref(p_), C.c_size_t(1 << 20)) == 0
        assert hip.hipFree(p_) == 0
    elif INJECT == "malloc_only":
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")
        p_ = C.c_void_p()
        assert hip.hipMalloc(C.byref(p_), C.c_size_t(1 << 20)) == 0
    elif INJECT == "empty_cache":
        torch.cuda.empty_cache()
    elif INJECT == "eval_step":
        tool.eval_step(data, 0)
    elif INJECT in ("ev_eval", "ev_rho", "modeflip", "repack_only", "repack_nofree", "statedict_only", "ev_eval_nofree"):
        net, ev = tool.network, tool.eval_tool
        with torch.no_grad():
            net.eval()
            if INJECT == "ev_eval":
                ev.eval(data, net, 0, False)
            elif INJECT == "ev_eval_nofree":
                KEEP.append(net.__dict__.get("_packed")); net.__dict__["_packed"] = None; net.__dict__["_op_model"] = None; net.__dict__["_handle"] = None
                ev.eval(data, net, 0, False)
            elif INJECT == "ev_rho":
                st_, en_, vec_, stime_, _ = ev.solar_creation_tool(R, include_times=True)
                ev.eval_Rho_Only({"Top": st_, "Bot": en_, "Sun_Angle": vec_, "Time_Encoded": stime_}, net, False, 0)
            elif INJECT == "repack_only":
                net.invalidate_packed(); net.device_model()
            elif INJECT == "repack_nofree":
                KEEP.append(net.__dict__.get("_packed")); net.__dict__["_packed"] = None; net.__dict__["_op_model"] = None; net.__dict__["_handle"] = None
                net.invalidate_packed(); net.device_model()
            elif INJECT == "statedict_only":
                torch.cat([v.detach().reshape(-1).float() for v in net.state_dict().values() if v.is_floating_point()]).cpu()
            net.train()
    elif INJECT == "sync":
        torch.cuda.synchronize()


def run(use_graph):
    tool = sn.T_NeRF_Net_Tool(_args(20, n_saves=2, use_mse=True), hm, hm, "cuda", H4, WC, get_data=lambda eval_mode: data, use_graph=use_graph)
    tool.network.load_state_dict(orc.init_weights(64, 4, 1))
    np.random.seed(3); torch.manual_seed(3)
    snaps = []
    for s_ in range(19):
        tool.step()
        torch.cuda.synchronize()
        snaps.append(tool.network._train_engine.grads.detach().cpu().clone())
        inject(tool)
        torch.cuda.synchronize()
    return snaps


a, b = run(False), run(True)
for s_, (x, y) in enumerate(zip(a, b)):
    scale = float(x.abs().max()) + 1e-30
    d = torch.nan_to_num((x - y).abs(), nan=float("inf"), posinf=float("inf"))
    if not (float(d.max()) <= 1e-3 * scale):
        print(f"INJECT={INJECT}: step {s_}: gradients differ, max abs {float(d.max()):.3e} of scale {scale:.3e}, {int((d > 1e-3 * scale).sum())} of {d.numel()} elements")
        sys.exit(0)
print(f"INJECT={INJECT}: no gradient difference above 1e-3 in 19 steps")
