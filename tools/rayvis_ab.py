"""GPU: the exact-solar pass (season_nerf::ray_visibility) with and without the early-out of saturated rays, and with its passes walking the secondary ray from
the point outwards (shipped) or from the sun side inwards (SNERF_RAYVIS_SUN_FIRST=1, the order before the reversal): on converged (sharp) weights at W = 256 and 512,
on init-law weights, and on the reference's own 12 000-step run under the DSM prior (tests/golden/trained12k_W64.npz, density head x 32: opaque GROUND where the
height map put it - the geometry a real checkpoint has).  ms for a 256 x 256 x 96 image's secondary rays and the largest change of any visibility.  Each arm in a
child process (the switches are read once)."""
import json, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import json, os, sys
sys.path.insert(0, %r)
import numpy as np, torch
import bench
import season_nerf_amd as sn
dev = torch.device("cuda", 0)
out = {}
def ground_sd():
    t = np.load(os.path.join(%r, "tests", "golden", "trained12k_W64.npz"), allow_pickle=False)
    head = ("G_NeRF_net.fc10Sigma.weight", "G_NeRF_net.fc10Sigma.bias")
    return {k[3:]: torch.tensor(t[k]) * (32.0 if k[3:] in head else 1.0) for k in t.files if k.startswith("sd_")}
for name, W, sharp in (("init_W256", 256, False), ("sharp_W256", 256, True), ("init_W512", 512, False), ("sharp_W512", 512, True), ("ground_W64", 64, None)):
    net = sn.T_NeRF(W, 4)
    net.load_state_dict(ground_sd() if sharp is None else (bench.sharp_state_dict(W)[0] if sharp else sn.synthetic_state_dict(net, 0)))
    net = net.to(dev).eval()
    r = bench.exact_solar_bench(dev, net, sizes=((256, 256, 96),))
    row = r["256x256x96"]
    out[name] = {"ms": row["ms"], "precision": r["precision_resolved"], "mean_visibility": row["mean_visibility"], "frac": row["roofline"]["frac"]}
    from season_nerf_amd import render as R_
    WC, H4 = np.array([41.29, -95.9, 300.0]), np.array([[310.0, 12.0, 0.0, -11650.0], [-9.0, 240.0, 0.0, 23390.0], [0.0, 0.0, 0.01, -3.0], [0, 0, 0, 1.0]])
    with torch.no_grad():
        dd = R_._render_by_dir_device(net, (80, 0), (30, 90), 0.25, (64, 64, 96), WC, H4, dev, False)
        sunv = R_.world_angle_2_local_vec(30, 90, WC, H4)
        vis = R_._exact_solar_visibility(net, dd["World_Points"].reshape(-1, 3), torch.tensor(sunv, dtype=torch.float32, device=dev), 96, zero_oob=True, sun64=sunv)
    torch.save(vis.cpu(), os.environ["RV_OUT"] + "_" + name + ".pt")
print(json.dumps(out))
""" % (REPO, REPO)
res = {}
for arm, env in (("early_out", {}), ("sun_first", {"SNERF_RAYVIS_SUN_FIRST": "1"}), ("all_passes", {"SNERF_RAYVIS_NO_EARLY_OUT": "1"})):
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, RV_OUT="/tmp/rv_" + arm, **env), capture_output=True, text=True, timeout=900)
    if r.returncode:
        print(r.stderr[-2000:]); sys.exit(1)
    res[arm] = json.loads(r.stdout.strip().splitlines()[-1])
import torch
for k in res["early_out"]:
    a, b = torch.load(f"/tmp/rv_early_out_{k}.pt"), torch.load(f"/tmp/rv_all_passes_{k}.pt")
    e, f, o = res["early_out"][k], res["all_passes"][k], res["sun_first"][k]
    print(f"{k:11s} [{e['precision']}] mean visibility {e['mean_visibility']:.3f}: all passes {f['ms']:8.1f} ms  early-out, sun side first {o['ms']:8.1f} ms ({f['ms'] / o['ms']:.2f}x)  "
          f"early-out, point first (shipped) {e['ms']:8.1f} ms ({f['ms'] / e['ms']:.2f}x)   largest change of a visibility (64x64x96 image) {float((a - b).abs().max()):.2e}")
