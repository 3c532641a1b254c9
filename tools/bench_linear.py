"""Per-shape timing of the training engine's three Linear products at the benchmark size (393 216 points), through the C-ABI
(snerf_linear_forward / dgrad / wgrad).  Run on the GPU box:  python3 tools/bench_linear.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib, ctypes as C
import torch
sn = importlib.import_module("season_nerf_amd")
L = sn._lib.lib()
M = 4096 * 96
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
dev = "cuda"


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def scratch(n_out, n_in):
    return torch.empty(L.snerf_linear_scratch_bytes(n_out, n_in), dtype=torch.uint8, device=dev)


rows = []
for (K, N, aol, stats) in [(256, 256, True, True), (256, 256, False, False), (320, 256, True, True), (64, 256, False, False), (256, 128, True, True),
                           (128, 3, True, False), (156, 128, True, False), (128, 128, True, False)]:
    lda = (K + 3) // 4 * 4
    A = torch.randn(M, lda, device=dev)
    W_ = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    out = torch.empty(M, max(N, 4), device=dev)
    ac = min(K, 256) // 8 * 8 if aol else 0
    if K == 156: ac = 128
    tab = torch.rand(2 * max(ac, 8), device=dev)
    sc = scratch(N, K)
    stt = torch.zeros(2 * N, dtype=torch.float64, device=dev)
    f = lambda: sn._lib.check(L.snerf_linear_forward(M, K, N, A.data_ptr(), lda, W_.data_ptr(), b.data_ptr(), 30.0, out.data_ptr(), max(N, 4),
                                                    stt.data_ptr() if stats else None, 1, sc.data_ptr(), sc.numel(), tab.data_ptr() if aol else None, ac, st), "fwd")
    us = timed(f)
    gb = 4.0 * M * (K + N) / 1e9
    rows.append(("forward", K, N, "aol" if aol else "-", us, gb / us * 1e3))
for (n_out, n_cols, act) in [(256, 256, True), (256, 256, False), (128, 256, True), (256, 128, True), (128, 128, True)]:
    dZ = torch.randn(M, n_out, device=dev)
    W_ = torch.randn(n_out, n_cols, device=dev) / n_out ** 0.5
    gi = torch.empty(M, n_cols, device=dev)
    z = torch.randn(M, n_cols, device=dev)
    tab = torch.rand(2 * n_cols, device=dev)
    mu, istd = torch.randn(n_cols, device=dev), torch.rand(n_cols, device=dev) + 0.5
    sums = torch.zeros(2 * n_cols, dtype=torch.float64, device=dev)
    sc = scratch(n_out, n_cols)
    f = lambda: sn._lib.check(L.snerf_linear_dgrad(M, n_cols, n_out, dZ.data_ptr(), n_out, W_.data_ptr(), n_cols, 30.0, 0, gi.data_ptr(), n_cols, 1,
                                                  sc.data_ptr(), sc.numel(), z.data_ptr() if act else None, n_cols, tab.data_ptr() if act else None,
                                                  mu.data_ptr() if act else None, istd.data_ptr() if act else None, sums.data_ptr() if act else None, st), "dgrad")
    us = timed(f)
    gb = 4.0 * M * (n_out + n_cols * (2 if act else 1)) / 1e9
    rows.append(("dgrad", n_out, n_cols, "act" if act else "-", us, gb / us * 1e3))
for (n_in, n_out, aol) in [(256, 256, True), (256, 256, False), (64, 256, False), (256, 128, True), (128, 128, True)]:
    dZ = torch.randn(M, n_out, device=dev)
    X = torch.randn(M, n_in, device=dev)
    dW = torch.zeros(n_out, n_in, device=dev)
    tab = torch.rand(2 * n_in, device=dev)
    f = lambda: sn._lib.check(L.snerf_linear_wgrad(M, n_in, n_out, dZ.data_ptr(), n_out, X.data_ptr(), n_in, 1.0, dW.data_ptr(), 1,
                                                  tab.data_ptr() if aol else None, n_in if aol else 0, st), "wgrad")
    us = timed(f)
    gb = 4.0 * M * (n_out + n_in) / 1e9
    rows.append(("wgrad", n_in, n_out, "aol" if aol else "-", us, gb / us * 1e3))
for r in rows:
    print("%-8s K=%4d N=%4d %-4s %8.1f us  %6.0f GB/s" % r)
