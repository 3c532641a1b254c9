"""GPU: WHICH STATE of a captured driver run leaves the eager run first?  After every step of an eager and of a captured run (same seeds, 20 steps: DSM-prior phase,
free phase, N_SAVES save points) the gradient arena, the parameters and Adam's moments are compared tensor by tensor (NaN-aware); prints the first step at which
anything differs by more than the atomics' noise, which tensors, and what the values look like.
  default                   : 0 of 12 runs differ (the engine's copies and fills are kernels: no hipGraph memory-operation node)
  SNERF_TRAIN_MEMOPS=1      : 9 of 9 differ (hipMemcpyAsync / hipMemsetAsync inside the step -> MEMCPY / MEMSET nodes): garbage gradients in whole layers from the
                              first replay after a save point's validation on
  ... DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 as well: 0 of 6 (the runtime's AQL packet capture of graph nodes off)
SNERF_GRAPH_PREPARE (default here: other) adds the two event waits in front of the capture that round 6 first found the failure with."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("SNERF_GRAPH_PREPARE", "other")
import numpy as np, torch
import season_nerf_amd as sn
from oracle import season_nerf_oracle as orc
from tests.test_net_tool import _args

rng = np.random.Generator(np.random.PCG64(3))
hm = rng.uniform(-0.8, 0.6, (24, 24))
R = 48
t = lambda a: torch.tensor(a, dtype=torch.float32)
data = {"Top": t(np.concatenate([rng.uniform(-1, 1, (R, 2)), np.ones((R, 1))], 1)), "Bot": t(np.concatenate([rng.uniform(-1, 1, (R, 2)), -np.ones((R, 1))], 1)),
        "Sun_Angle": torch.nn.functional.normalize(t(rng.uniform(0.1, 1, (R, 3))), dim=1), "Time_Encoded": t(rng.uniform(-1, 1, (R, 4))), "GT_Color": t(rng.uniform(0, 1, (R, 3)))}
WC, H4 = np.array([41.29, -95.9, 300.0]), np.array([[310.0, 12.0, 0.0, -11650.0], [-9.0, 240.0, 0.0, 23390.0], [0.0, 0.0, 0.01, -3.0], [0, 0, 0, 1.0]])


def run(use_graph):
    tool = sn.T_NeRF_Net_Tool(_args(20, n_saves=int(os.environ.get("N_SAVES", "5")), use_mse=True), hm, hm, "cuda", H4, WC, get_data=lambda eval_mode: data, use_graph=use_graph)
    tool.network.load_state_dict(orc.init_weights(64, 4, 1))
    np.random.seed(3); torch.manual_seed(3)
    snaps = []
    for s_ in range(20):
        tool.step()
        torch.cuda.synchronize()
        st = tool.network._param_store
        eng = tool.network._train_engine
        names = {n: (p.data_ptr() - st.params.data_ptr()) // 4 for n, p in tool.network.named_parameters() if st.params.data_ptr() <= p.data_ptr() < st.params.data_ptr() + st.params.numel() * 4}
        snaps.append({"grads": eng.grads.detach().cpu().clone(), "params": st.params.detach().cpu().clone(), "m": st.adam_m.cpu().clone(), "v": st.adam_v.cpu().clone(),
                      "loss": {k: float(v[0]) for k, v in tool.last_loss.items()}, "names": names, "sizes": {n: p.numel() for n, p in tool.network.named_parameters()}})
    return snaps


a, b = run(False), run(True)
for s_ in range(20):
    for key in (("params", "m", "v") if os.environ.get("PROBE_SKIP_GRADS") else ("grads", "params", "m", "v")):
        x, y = a[s_][key], b[s_][key]
        scale = float(x.abs().max()) + 1e-30
        d = torch.nan_to_num((x - y).abs(), nan=float("inf"), posinf=float("inf"))
        if not (float(d.max()) <= 1e-3 * scale):
            idx = int(d.argmax())
            owner = [n for n, o in a[s_]["names"].items() if o <= idx < o + a[s_]["sizes"][n]]
            worst = sorted(((float(d[o:o + a[s_]["sizes"][n]].max()) / (float(x[o:o + a[s_]["sizes"][n]].abs().max()) + 1e-30), n) for n, o in a[s_]["names"].items()), reverse=True)[:6]
            print(f"   ({int((d > 1e-3 * scale).sum())} of {d.numel()} elements differ; first at {int(torch.nonzero(d > 1e-3 * scale)[0])}, last at {int(torch.nonzero(d > 1e-3 * scale)[-1])})")
            hit = [(n, int((d[o:o + a[s_]["sizes"][n]] > 1e-3 * scale).sum()), a[s_]["sizes"][n]) for n, o in sorted(a[s_]["names"].items(), key=lambda kv: kv[1])]
            print("   tensors hit (name, elements off, size):", [h for h in hit if h[1]])
            print("   tensors clean:", [h[0] for h in hit if not h[1]])
            bad = torch.nonzero(d > 1e-3 * scale).reshape(-1)
            j = int(bad[len(bad) // 2])
            win = y[j - 4:j + 8]
            print("   window (float):", [f"{float(w):.4g}" for w in win])
            print("   window (hex):  ", [f"{int(w):08x}" for w in win.view(torch.int32) & 0xffffffff] if False else [hex(int(w) & 0xffffffff) for w in win.view(torch.int32)])
            print("   gaps between bad indices:", torch.unique(bad[1:] - bad[:-1])[:8].tolist(), " first bad", int(bad[0]))
            print(f"step {s_}: {key} differs: max abs {float(d.max()):.3e} of scale {scale:.3e} at {idx} ({owner}); worst tensors (rel to own max): {[(round(w, 4), n) for w, n in worst]}")
            print("   losses eager", {k: round(v, 6) for k, v in a[s_]["loss"].items()}); print("   losses graph", {k: round(v, 6) for k, v in b[s_]["loss"].items()})
            sys.exit(0)
print("no state difference above 1e-3 in 20 steps")
