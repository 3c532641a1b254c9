"""forward 256->256 only (tuning helper): python3 tools/bench_linear_fwd.py [aol|plain] [row tiles per worker ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib, ctypes as C
import torch
sn = importlib.import_module("season_nerf_amd")
L = sn._lib.lib()
K, N = 256, 256
aol = len(sys.argv) > 1 and sys.argv[1] == "aol"
tiles = [int(x) for x in sys.argv[2:]] or [12]
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for t in tiles:
    M = 32768 * t
    A = torch.randn(M, K, device="cuda"); W_ = torch.randn(N, K, device="cuda") / 16; b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda"); tab = torch.rand(2 * K, device="cuda")
    sc = torch.empty(L.snerf_linear_scratch_bytes(N, K), dtype=torch.uint8, device="cuda")
    f = lambda: sn._lib.check(L.snerf_linear_forward(M, K, N, A.data_ptr(), K, W_.data_ptr(), b.data_ptr(), 30.0, out.data_ptr(), N, None, 1, sc.data_ptr(), sc.numel(),
                                                    tab.data_ptr() if aol else None, K if aol else 0, st), "fwd")
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print("aol" if aol else "plain", "tiles/worker", t, "M", M, "%.1f us" % (e0.elapsed_time(e1) * 100))
