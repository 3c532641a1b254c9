"""Comparison of the pipelined full-tile row GEMMs against the general row kernel: the 32x32x16 form bit for bit (same arithmetic,
same summation order), the 16x16x32 form to fp32 summation rounding.  One child process per mode (read once per process).
usage (GPU box): python3 tools/compare_gemm_paths.py"""
import os, subprocess, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.dirname(HERE))
    import importlib, ctypes as C
    import torch
    sn = importlib.import_module("season_nerf_amd")
    L = sn._lib.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = {}
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    for (M, K, N, lda, aol, stats) in [(4096 * 8, 256, 256, 256, 256, True), (4096 * 8 + 77, 256, 256, 256, 0, False), (30001, 320, 256, 320, 256, True),
                                       (25600, 64, 256, 64, 0, True), (25600, 256, 128, 256, 256, True), (25611, 160, 128, 160, 128, False), (4096, 128, 128, 156, 128, False),
                                       (20000, 128, 256, 128, 128, False), (20000, 128, 3, 128, 128, False), (33333, 256, 12, 256, 256, False), (20000, 128, 1, 132, 0, False)]:
        A = rnd(M, lda); W_ = rnd(N, K) / K ** 0.5; b = rnd(N)
        o = torch.full((M, N), -7.0, device="cuda")      # untouched cells must stay -7 in both paths
        tab = torch.rand(2 * max(aol, 8), device="cuda", generator=g)
        sc = torch.empty(L.snerf_linear_scratch_bytes(N, K), dtype=torch.uint8, device="cuda")
        stt = torch.zeros(2 * N, dtype=torch.float64, device="cuda")
        sn._lib.check(L.snerf_linear_forward(M, K, N, A.data_ptr(), lda, W_.data_ptr(), b.data_ptr(), 30.0, o.data_ptr(), N, stt.data_ptr() if stats else None, 1,
                                             sc.data_ptr(), sc.numel(), tab.data_ptr() if aol else None, aol, st), "fwd")
        out[f"fwd_{M}_{K}_{N}_{aol}"] = o.cpu().numpy()
        if stats: out[f"fwdstats_{M}_{K}_{N}_{aol}"] = stt.cpu().numpy()
    for (M, n_out, n_cols, act, bn) in [(4096 * 8, 256, 256, True, True), (30001, 256, 256, True, False), (25600, 128, 256, True, True), (25611, 256, 128, True, True),
                                        (25600, 256, 256, False, False), (20000, 128, 128, True, False)]:
        dZ = rnd(M, n_out); W_ = rnd(n_out, n_cols) / n_out ** 0.5
        gi = torch.full((M, n_cols), -7.0, device="cuda"); z = rnd(M, n_cols)
        tab = torch.rand(2 * n_cols, device="cuda", generator=g); mu = rnd(n_cols); istd = torch.rand(n_cols, device="cuda", generator=g) + 0.5
        sums = torch.zeros(2 * n_cols, dtype=torch.float64, device="cuda")
        sc = torch.empty(L.snerf_linear_scratch_bytes(n_out, n_cols), dtype=torch.uint8, device="cuda")
        sn._lib.check(L.snerf_linear_dgrad(M, n_cols, n_out, dZ.data_ptr(), n_out, W_.data_ptr(), n_cols, 30.0, 0, gi.data_ptr(), n_cols, 1, sc.data_ptr(), sc.numel(),
                                           z.data_ptr() if act else None, n_cols, tab.data_ptr() if act else None, mu.data_ptr() if (act and bn) else None,
                                           istd.data_ptr() if (act and bn) else None, sums.data_ptr() if act else None, st), "dgrad")
        out[f"dgrad_{M}_{n_out}_{n_cols}_{act}_{bn}"] = gi.cpu().numpy()
        if act: out[f"dgradsums_{M}_{n_out}_{n_cols}_{bn}"] = sums.cpu().numpy()
    n_fuzz = int(os.environ.get("SNERF_CMP_FUZZ", "0"))      # extra random shapes (same seed in both modes)
    rs = np.random.RandomState(11)
    for it in range(n_fuzz):
        M = int(rs.randint(1024, 40000))
        K = 16 * int(rs.randint(1, 21))
        N = int(rs.choice([1, 3, 12, 31, 32, 64, 96, 128, 160, 192, 224, 256]))
        lda = K + 4 * int(rs.randint(0, 3))
        aol = 16 * int(rs.randint(0, K // 16 + 1))
        stats = bool(rs.randint(0, 2)) and N % 32 == 0
        A = rnd(M, lda); W_ = rnd(N, K) / K ** 0.5; b = rnd(N)
        ldc = N + int(rs.randint(0, 2)) * 4
        o = torch.full((M, ldc), -7.0, device="cuda")
        tab = torch.rand(2 * max(aol, 8), device="cuda", generator=g)
        sc = torch.empty(L.snerf_linear_scratch_bytes(N, K), dtype=torch.uint8, device="cuda")
        stt = torch.zeros(2 * N, dtype=torch.float64, device="cuda")
        sn._lib.check(L.snerf_linear_forward(M, K, N, A.data_ptr(), lda, W_.data_ptr(), b.data_ptr(), 30.0, o.data_ptr(), ldc, stt.data_ptr() if stats else None, 1,
                                             sc.data_ptr(), sc.numel(), tab.data_ptr() if aol else None, aol, st), "fwd")
        out[f"fz{it}fwd_{M}_{K}_{N}_{aol}_{lda}_{ldc}"] = o.cpu().numpy()
        n_out, n_cols = 16 * int(rs.randint(1, 17)), int(rs.choice([32, 64, 128, 160, 256]))
        act, bn = bool(rs.randint(0, 2)), bool(rs.randint(0, 2))
        dZ = rnd(M, n_out); W2_ = rnd(n_out, n_cols) / n_out ** 0.5
        gi = torch.full((M, n_cols), -7.0, device="cuda"); z = rnd(M, n_cols)
        tab2 = torch.rand(2 * n_cols, device="cuda", generator=g); mu = rnd(n_cols); istd = torch.rand(n_cols, device="cuda", generator=g) + 0.5
        sums = torch.zeros(2 * n_cols, dtype=torch.float64, device="cuda")
        sc2 = torch.empty(L.snerf_linear_scratch_bytes(n_out, n_cols), dtype=torch.uint8, device="cuda")
        sn._lib.check(L.snerf_linear_dgrad(M, n_cols, n_out, dZ.data_ptr(), n_out, W2_.data_ptr(), n_cols, 30.0, 0, gi.data_ptr(), n_cols, 1, sc2.data_ptr(), sc2.numel(),
                                           z.data_ptr() if act else None, n_cols, tab2.data_ptr() if act else None, mu.data_ptr() if (act and bn) else None,
                                           istd.data_ptr() if (act and bn) else None, sums.data_ptr() if act else None, st), "dgrad")
        out[f"fz{it}dgrad_{M}_{n_out}_{n_cols}_{act}_{bn}"] = gi.cpu().numpy()
    np.savez(sys.argv[1], **out)
    sys.exit(0)
os.makedirs("/tmp/cmp", exist_ok=True)
# m0: general kernel; m1: full-tile 32x32x16 kernel (bit-identical to m0); m2: full-tile 16x16x32 kernel where the launcher picks it
# (same products, the k terms of a 32-k step summed in another order: equal to fp32 summation rounding, never bitwise by design)
# (SNERF_CMP_WREG=1 with SNERF_LIB pointing at a library built with -DSNERF_WITH_WREG=1 (python tools/variants.py build gemm16.hip wreg="-DSNERF_WITH_WREG=1"):
#  mode 2 additionally routes the eligible shapes to the experimental gemm_wreg_kernel, weights in registers - not part of the shipped library)
for mode, env_ in (("0", {"SNERF_GEMM_FULL": "0"}), ("1", {"SNERF_GEMM_FULL": "1", "SNERF_GEMM16": "0"}),
                   ("2", {"SNERF_GEMM_FULL": "1", "SNERF_GEMM16": "1", "SNERF_GEMM_WREG": os.environ.get("SNERF_CMP_WREG", "0")})):
    subprocess.check_call([sys.executable, __file__, f"/tmp/cmp/m{mode}.npz"], env=dict(os.environ, **env_))
a, b, c = np.load("/tmp/cmp/m0.npz"), np.load("/tmp/cmp/m1.npz"), np.load("/tmp/cmp/m2.npz")
bad = n16 = 0
for k in a.files:
    if "stats" in k or "sums" in k:      # double atomics over workgroups: order-dependent in the last bits
        ok = np.allclose(a[k], b[k], rtol=1e-6, atol=1e-5 * float(k.split("_")[1]))      # fp32 per-lane partial sums, grouped differently
        ok16 = np.allclose(a[k], c[k], rtol=1e-5, atol=2e-5 * float(k.split("_")[1]))
    else:
        ok = np.array_equal(a[k], b[k])
        scale = float(np.abs(a[k][a[k] != -7.0]).max()) if (a[k] != -7.0).any() else 1.0
        ok16 = np.array_equal(a[k] == -7.0, c[k] == -7.0) and float(np.abs(a[k] - c[k]).max()) <= 4e-6 * scale
        n16 += not np.array_equal(a[k], c[k])
    print(("ok   " if ok else "DIFF ") + ("ok16   " if ok16 else "DIFF16 ") + k, "" if (ok and ok16) else (float(np.abs(a[k] - b[k]).max()), float(np.abs(a[k] - c[k]).max())))
    bad += (not ok) + (not ok16)
print("outputs the 16x16x32 kernel produced (differ in the last bits):", n16)
print("mismatches:", bad)
sys.exit(1 if (bad or n16 == 0) else 0)
