"""GPU diagnostic: is the width-512 field kernel bit-reproducible launch to launch?  Reports, per precision, how many outputs differ between launches, by how much and where
(tile, position in the 64-point tile) - a race in the ring or in the partial-sum exchange shows up as a pattern."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import season_nerf_amd as sn
from oracle import season_nerf_oracle as orc

W, R, S = 512, 4096, 96
rng = np.random.Generator(np.random.PCG64(15))
T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)
top = T(np.concatenate([rng.uniform(-1, 1, (R, 2)), np.ones((R, 1))], 1)).cuda()
bot = T(np.concatenate([rng.uniform(-1, 1, (R, 2)), -np.ones((R, 1))], 1)).cuda()
sun = rng.uniform(0, 1, (R, 3)); sun = T(sun / np.linalg.norm(sun, axis=1, keepdims=True)).cuda()
tim = T(rng.uniform(-1, 1, (R, 4))).cuda()
tv = sn.sample_parameters(S, eval_mode=True).cuda()
variant = int(os.environ.get("KS_VARIANT", "0"))
for prec in sys.argv[1:] or ("bf16x3", "i8x3"):
    net = sn.T_NeRF(W, 4); net.load_state_dict(orc.init_weights(W, 4, 12)); net.precision = prec; net = net.to("cuda").eval()
    cls, _, _ = net._groups(tim, sun)
    runs = []
    for it in range(6):
        z = lambda *sh: torch.zeros(*sh, device="cuda")
        rho, sv, col, craw, adj, adjc = z(R * S), z(R * S), z(R * S, 3), z(R * S, 3), z(R * S, 12), z(R * S, 3)
        fo = sn._lib.FieldOut(d_rho=rho.data_ptr(), d_solar_vis=sv.data_ptr(), d_col=col.data_ptr(), d_col_raw=craw.data_ptr(), d_adjust=adj.data_ptr(), d_adjust_col=adjc.data_ptr())
        sn._lib.check(sn._lib.lib().snerf_field_forward_rays(net.device_model(), variant, R, S, top.data_ptr(), bot.data_ptr(), tv.data_ptr(), 1, sun.data_ptr(),
                                                             cls.data_ptr(), C.byref(fo), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "field")
        torch.cuda.synchronize()
        runs.append((rho, sv, col.reshape(-1, 3).sum(1), craw.sum(1), adjc.sum(1)) + tuple(adj[:, i].clone() for i in range(12)))
    tile = 64 if prec == "bf16x3" else 128
    for it in range(1, 3):
        for name, a, b in zip(("rho", "sv", "col", "col_raw", "adjust_col") + tuple(f"adj{i}" for i in range(12)), runs[0], runs[it]):
            d = (a - b).abs()
            idx = torch.nonzero(d > 0).reshape(-1)
            if idx.numel() == 0:
                continue
            tiles = torch.unique(idx // tile)
            pos = torch.bincount(idx % tile, minlength=tile)
            print(f"{prec} launch {it} vs 0: {name}: {idx.numel()} of {a.numel()} differ, max abs {float(d.max()):.3e} (value scale {float(a.abs().max()):.2e}), {tiles.numel()} tiles "
                  f"(first {tiles[:6].tolist()}), workgroups {torch.unique(tiles % 256)[:8].tolist()}, positions in tile: first half {int(pos[:tile // 2].sum())} second half {int(pos[tile // 2:].sum())}")
    print(prec, "done")
