"""Times snerf_linear_forward at the training step's size on whatever library SNERF_LIB names (tools/variants.py run --script tools/areg_time.py)."""
import ctypes as C, os, sys
os.environ.setdefault("SNERF_GEMM_AREG", "2")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import season_nerf_amd as sn
L = sn._lib.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cuda"); g.manual_seed(5)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
for (M, K, N, aol, stats) in [(4096 * 96, 512, 512, 512, True), (4096 * 96, 512, 512, 0, False), (4096 * 96, 256, 256, 256, True)]:
    A = rnd(M, K); W_ = rnd(N, K) / K ** 0.5; b = rnd(N)
    o = torch.empty(M, N, device="cuda")
    tab = torch.rand(2 * max(aol, 8), device="cuda", generator=g)
    sc = torch.empty(L.snerf_linear_scratch_bytes(N, K), dtype=torch.uint8, device="cuda")
    stt = torch.zeros(2 * N, dtype=torch.float64, device="cuda")
    run = lambda: sn._lib.check(L.snerf_linear_forward(M, K, N, A.data_ptr(), K, W_.data_ptr(), b.data_ptr(), 30.0, o.data_ptr(), N, stt.data_ptr() if stats else None, 1,
                                                      sc.data_ptr(), sc.numel(), tab.data_ptr() if aol else None, aol, st), "fwd")
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f"   K={K} N={N} aol={aol} stats={stats}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)
