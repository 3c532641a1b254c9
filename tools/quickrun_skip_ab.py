"""GPU: renderer A (Quick_Run_Net.render_img, exact solar on - its default) with and without the weight-based skip of secondary rays
(`skip_weightless`, All_in_One_Eval.eval_exact_solar): ms per 256 x 256 x 96 image, secondary rays walked, largest change of any pixel of the three images.
Sets: converged (sharp) weights at W = 256 / 512, the reference's 12 000-step DSM-prior run with its density head x 32 (opaque ground), init-law fog."""
import os, sys, time
from types import SimpleNamespace
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
import bench, season_nerf_amd as sn
dev = torch.device("cuda", 0)
WC, H4 = np.array([41.29, -95.9, 300.0]), np.array([[310.0, 12.0, 0.0, -11650.0], [-9.0, 240.0, 0.0, 23390.0], [0.0, 0.0, 0.01, -3.0], [0, 0, 0, 1.0]])
def ground_sd():
    t = np.load(os.path.join(REPO, "tests", "golden", "trained12k_W64.npz"), allow_pickle=False)
    head = ("G_NeRF_net.fc10Sigma.weight", "G_NeRF_net.fc10Sigma.bias")
    return {k[3:]: torch.tensor(t[k]) * (32.0 if k[3:] in head else 1.0) for k in t.files if k.startswith("sd_")}
args = SimpleNamespace(n_samples=96, Use_Reg=True, Solar_Type_2=False, Use_MSE_loss=True, Use_Solar=True, sc_lambda=0.03, number_low_frequency_cases=4)
for name, W, sd in (("init_W256", 256, None), ("sharp_W256", 256, bench.sharp_state_dict(256)[0]), ("sharp_W512", 512, bench.sharp_state_dict(512)[0]), ("ground_W64", 64, ground_sd())):
    net = sn.T_NeRF(W, 4)
    net.load_state_dict(sd if sd is not None else sn.synthetic_state_dict(net, 0))
    net = net.to(dev).eval()
    res = {}
    for skip in (None, 1e-9):
        qr = sn.Quick_Run_Net(net, args, WC, H4, dev, use_full_solar=True, skip_weightless=skip)
        qr.render_img((70, 20), (40, 110), 0.3, 64)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        imgs, mask = qr.render_img((70, 20), (40, 110), 0.3, 256)
        torch.cuda.synchronize()
        res[skip] = (imgs, (time.perf_counter() - t0) * 1e3, getattr(qr.eval_tool, "last_exact_solar_rays", None))
    a, b = res[None], res[1e-9]
    d = max(float(np.abs(a[0][k] - b[0][k]).max()) for k in a[0])
    print(f"{name:11s} [{net.resolved_precision}] every sample {a[1]:8.1f} ms   skip PS < 1e-9 {b[1]:8.1f} ms ({a[1] / b[1]:.2f}x), secondary rays walked {b[2][0]} of {b[2][1]} "
          f"({100.0 * b[2][0] / b[2][1]:.0f} %)   largest change of a pixel {d:.2e}", flush=True)
