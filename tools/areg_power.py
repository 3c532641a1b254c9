"""Shader clock and package power (rocm-smi) while snerf_linear_forward runs back to back at the W = 512 training size - for the in-tree library in both forms of
gemm_areg_kernel (SNERF_AREG_HV = 2 / 1) and for every ablation build under build/variants (tools/variants.py build gemm_areg.hip ...).  The question it answers: is
the row GEMM of the reference's default width issue-bound or power-bound (the parts of its time ADD UP in every structure tried: profiles/r5/areg_ablation*.txt).
usage (GPU box): python3 tools/areg_power.py"""
import glob
import os
import re
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)

if len(sys.argv) > 1 and sys.argv[1] == "loop":
    import ctypes as C
    os.environ.setdefault("SNERF_GEMM_AREG", "2")
    sys.path.insert(0, REPO)
    import torch
    import season_nerf_amd as sn
    L = sn._lib.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    M, K, N = 4096 * 96, 512, 512
    aol = 512 if sys.argv[2] == "aol" else 0
    A = torch.randn(M, K, device="cuda", generator=g); W_ = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5; b = torch.randn(N, device="cuda", generator=g)
    o = torch.empty(M, N, device="cuda")
    tab = torch.rand(2 * 512, device="cuda", generator=g)
    sc = torch.empty(L.snerf_linear_scratch_bytes(N, K), dtype=torch.uint8, device="cuda")
    stt = torch.zeros(2 * N, dtype=torch.float64, device="cuda")
    run = lambda: sn._lib.check(L.snerf_linear_forward(M, K, N, A.data_ptr(), K, W_.data_ptr(), b.data_ptr(), 30.0, o.data_ptr(), N, stt.data_ptr() if aol else None, 1,
                                                      sc.data_ptr(), sc.numel(), tab.data_ptr() if aol else None, aol, st), "fwd")
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    print("READY", flush=True)
    t0 = time.time()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < 5.0:
        for _ in range(50):
            run()
        n += 50
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    print(f"US {e0.elapsed_time(e1) / n * 1e3:.1f}", flush=True)
    sys.exit(0)


def sample():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
    pw = re.search(r"Power \(W\): ([\d.]+)", out)
    return (int(sclk.group(1)) if sclk else -1, float(pw.group(1)) if pw else -1.0)


def one(name, env, mode):
    p = subprocess.Popen([sys.executable, __file__, "loop", mode], env=dict(os.environ, **env), stdout=subprocess.PIPE, text=True)
    assert p.stdout.readline().strip() == "READY"
    time.sleep(1.0)
    s = [sample() for _ in range(3) if not time.sleep(0.8)]
    us = p.stdout.readline().strip()
    p.wait()
    print(f"{name:<22} {mode:<5} {us:<12} sclk {[a for a, _ in s]} MHz   power {[b for _, b in s]} W", flush=True)


for mode in ("plain", "aol"):
    one("in-tree HV=2", {}, mode)
    one("in-tree HV=1", {"SNERF_AREG_HV": "1"}, mode)
    for lib in sorted(glob.glob(os.path.join(REPO, "build", "variants", "lib_*.so"))):
        one(os.path.basename(lib)[4:-3], {"SNERF_LIB": lib}, mode)
