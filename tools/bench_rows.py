"""Row-GEMM timing at the training step's shapes (tuning helper, GPU box): python3 tools/bench_rows.py [reps]
forward 256->256 with activation on load + BatchNorm sums, plain forward, dgrad with the activation-backward epilogue, M = 393216."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import season_nerf_amd as sn
L = sn._lib.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
M, K, N = 393216, 256, 256
A = torch.randn(M, K, device="cuda") * 4
W_ = torch.randn(N, K, device="cuda") / 16
b = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda")
tab = torch.rand(2 * K, device="cuda")
stats = torch.zeros(2 * N + 1024 + 3072, dtype=torch.float64, device="cuda")      # (+ the stamps of the -DSNERF_STAMP16 / -DSNERF_PHASE16 diagnostic builds)
sc = torch.empty(L.snerf_linear_scratch_bytes(N, K), dtype=torch.uint8, device="cuda")
mu, istd = torch.randn(N, device="cuda"), torch.rand(N, device="cuda") + 0.5
cases = {
    "fwd aol+stats": lambda: L.snerf_linear_forward(M, K, N, A.data_ptr(), K, W_.data_ptr(), b.data_ptr(), 30.0, out.data_ptr(), N, stats.data_ptr(), 1, sc.data_ptr(), sc.numel(), tab.data_ptr(), K, st),
    "fwd aol": lambda: L.snerf_linear_forward(M, K, N, A.data_ptr(), K, W_.data_ptr(), b.data_ptr(), 30.0, out.data_ptr(), N, None, 1, sc.data_ptr(), sc.numel(), tab.data_ptr(), K, st),
    "fwd plain+stats": lambda: L.snerf_linear_forward(M, K, N, A.data_ptr(), K, W_.data_ptr(), b.data_ptr(), 30.0, out.data_ptr(), N, stats.data_ptr(), 1, sc.data_ptr(), sc.numel(), None, 0, st),
    "fwd plain": lambda: L.snerf_linear_forward(M, K, N, A.data_ptr(), K, W_.data_ptr(), b.data_ptr(), 30.0, out.data_ptr(), N, None, 1, sc.data_ptr(), sc.numel(), None, 0, st),
    "dgrad act+bn": lambda: L.snerf_linear_dgrad(M, K, N, A.data_ptr(), N, W_.data_ptr(), K, 30.0, 0, out.data_ptr(), K, 1, sc.data_ptr(), sc.numel(), A.data_ptr(), K, tab.data_ptr(), mu.data_ptr(), istd.data_ptr(), stats.data_ptr(), st),
    "dgrad plain": lambda: L.snerf_linear_dgrad(M, K, N, A.data_ptr(), N, W_.data_ptr(), K, 30.0, 0, out.data_ptr(), K, 1, sc.data_ptr(), sc.numel(), None, 0, None, None, None, None, st),
}
only = os.environ.get("ROWS_CASES")
for name, f in list(cases.items()) * int(os.environ.get("ROWS_ROUNDS", "2")):
    if only and name not in only.split(","):
        continue
    for _ in range(3):
        sn._lib.check(f(), name)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"{name:16s} {us:7.1f} us  (incl. ~5 us weight split)", flush=True)
    if os.environ.get("ROWS_PHASES") and "stats" in name:       # -DSNERF_PHASE16 build: sums of the LAST launch (the kernel overwrites them)
        ph = stats[2 * N + 1024:2 * N + 1024 + 3072].cpu().numpy().reshape(256, 2, 6)
        for half in (0, 1):
            q = ph[:, half]
            q = q[q[:, 5] > 0]
            if len(q):
                m = q.mean(0)
                print(f"    wave {4 * half}: per k-step  load wait {m[0] / m[5]:7.0f}  convert + refill {m[1] / m[5]:7.0f}  LDS + MFMA issue {m[2] / m[5]:7.0f} cycles;  "
                      f"epilogue per tile {m[3] / (m[5] / 8):7.0f};  tile loop {m[4]:9.0f} cycles = {100 * (m[0] + m[1] + m[2] + m[3]) / m[4]:.0f} % accounted, {m[5]:.0f} k-steps")
    if os.environ.get("ROWS_STAMPS") and "stats" in name:
        fin = stats[2 * N + 768:2 * N + 1024].cpu().numpy() * 0.01
        s_ = stats[2 * N:2 * N + 768].cpu().numpy().reshape(-1, 3)
        s_ = s_[s_[:, 1] > 0]
        if len(s_):
            import numpy as np
            te, t0, t1 = s_[:, 0] * 0.01, s_[:, 1] * 0.01, s_[:, 2] * 0.01                  # us: kernel entry, tile loop start, tile loop end
            dur = t1 - t0
            print(f"    {len(s_)} workgroups: entry -> loop start min / median / max {(t0 - te).min():.1f} / {np.median(t0 - te):.1f} / {(t0 - te).max():.1f} us; "
                  f"tile loop {dur.min():.1f} / {np.median(dur):.1f} / {dur.max():.1f} us; first entry -> last entry {te.max() - te.min():.1f}, "
                  f"first entry -> last loop end {t1.max() - te.min():.1f} us; -> last wave's stores and atomics acknowledged {fin.max() - te.min():.1f} us")
        stats.zero_()
