"""gemm_areg_kernel (csrc/gemm_areg.hip: K = 256 / 512 row GEMM, activations resident in AGPRs, weights streamed through the LDS ring) against the
column-group row GEMMs of gemm.hip: same fragments, same product order, same k order - bit for bit on the outputs, fp32-partial-sum rounding on the
BatchNorm column sums - and timed at the training step's size.  One child process per mode (the switch is read once per process).
usage (GPU box): python3 tools/areg_check.py"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SHAPES = [(4096 * 96, 512, 512, 512, 512, True, 512), (4096 * 96, 512, 512, 512, 0, False, 512), (4096 * 8 + 77, 512, 512, 516, 512, True, 520),
          (1280, 512, 256, 512, 512, True, 256), (129, 512, 64, 512, 0, True, 64), (25600, 512, 128, 512, 512, False, 128), (3001, 512, 512, 512, 512, True, 512)]
SHAPES_K256 = [(4096 * 96, 256, 256, 256, 256, True, 256), (25611, 256, 512, 256, 256, True, 512), (4096 * 8, 256, 128, 260, 0, False, 128)]

if len(sys.argv) > 1:
    sys.path.insert(0, os.path.dirname(HERE))
    import ctypes as C
    import importlib
    import torch
    sn = importlib.import_module("season_nerf_amd")
    L = sn._lib.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = {}
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    for (M, K, N, lda, aol, stats, ldc) in SHAPES + SHAPES_K256:
        A = rnd(M, lda); W_ = rnd(N, K) / K ** 0.5; b = rnd(N)
        o = torch.full((M, ldc), -7.0, device="cuda")      # untouched cells must stay -7 in both paths
        tab = torch.rand(2 * max(aol, 8), device="cuda", generator=g)
        sc = torch.empty(L.snerf_linear_scratch_bytes(N, K), dtype=torch.uint8, device="cuda")
        stt = torch.zeros(2 * N, dtype=torch.float64, device="cuda")
        run = lambda: sn._lib.check(L.snerf_linear_forward(M, K, N, A.data_ptr(), lda, W_.data_ptr(), b.data_ptr(), 30.0, o.data_ptr(), ldc,
                                                          stt.data_ptr() if stats else None, 1, sc.data_ptr(), sc.numel(), tab.data_ptr() if aol else None, aol, st), "fwd")
        run()
        torch.cuda.synchronize()
        out[f"fwd_{M}_{K}_{N}_{aol}_{lda}_{ldc}"] = o[:8192].cpu().numpy() if M > 100000 else o.cpu().numpy()
        out[f"sum_{M}_{K}_{N}_{aol}_{lda}_{ldc}"] = np.array([float(o.double().sum()), float((o.double() ** 2).sum())])
        if stats:
            out[f"fwdstats_{M}_{K}_{N}_{aol}_{lda}_{ldc}"] = stt.cpu().numpy()
        if M >= 100000:
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            print(f"  mode {os.environ.get('SNERF_GEMM_AREG')}: M={M} K={K} N={N} aol={aol} stats={stats}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call (incl. the weight split)", flush=True)
    np.savez(sys.argv[1], **out)
    sys.exit(0)

os.makedirs("/tmp/cmp", exist_ok=True)
for mode in ("0", "2"):
    subprocess.check_call([sys.executable, __file__, f"/tmp/cmp/areg{mode}.npz"], env=dict(os.environ, SNERF_GEMM_AREG=mode))
a, b = np.load("/tmp/cmp/areg0.npz"), np.load("/tmp/cmp/areg2.npz")
bad = 0
for k in a.files:
    if k.startswith("fwdstats") or k.startswith("sum_"):
        ok = np.allclose(a[k], b[k], rtol=2e-6, atol=1e-5 * float(k.split("_")[1]))
    else:
        ok = np.array_equal(a[k], b[k])
    print(("ok   " if ok else "DIFF ") + k, "" if ok else float(np.abs(a[k] - b[k]).max()))
    bad += not ok
print("mismatches:", bad)
sys.exit(1 if bad else 0)
