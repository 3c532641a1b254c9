"""A/B builds of the width-512 bf16x3 kernel (csrc/kernels_ks.hip): timing-only ablations and tuning switches.

  python3 tools/ks_variants.py build   (here, no GPU: one extra .so per variant under build/variants/, only kernels_ks.hip recompiled)
  python3 tools/ks_variants.py run     (on the GPU box: every variant in a child process through the C ABI, ms per 4096 x 96 field launch)

A variant is a name -> extra compiler flags; results of ablation builds are WRONG by construction (they answer "what does this part cost").
Variants whose build spills to scratch are refused (the hand-counted vmcnt waits assume no compiler-issued memory traffic in the chain)."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
OUT = os.path.join(REPO, "build", "variants")
SRC = os.environ.get("KS_SRC", "kernels_ks.hip")      # the translation unit a variant recompiles (KS_SRC=kernels_i8_w512.hip: the int8 kernel of width 512)

VARIANTS = {
    "base": [],
    "no_ring": ["-DSNERF_ABLATE", "-DABL=4"],
    "no_sin": ["-DSNERF_ABLATE", "-DABL=8"],
    "no_xchg": ["-DSNERF_ABLATE", "-DABL=16"],
    "no_ldsread": ["-DSNERF_ABLATE", "-DABL=2"],
    "mfma_only": ["-DSNERF_ABLATE", "-DABL=30"],
    "no_ring_no_lds": ["-DSNERF_ABLATE", "-DABL=6"],
}
for a in sys.argv[2:]:
    if "=" in a and not a.startswith("-"):            # name=-Dflag,-Dflag
        n, f = a.split("=", 1)
        VARIANTS[n] = [x for x in f.split(",") if x]


def build():
    import importlib.util
    spec = importlib.util.spec_from_file_location("snerf_build", os.path.join(REPO, "season_nerf_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    b.build(verbose=False)
    os.makedirs(OUT, exist_ok=True)
    want = [a.split("=", 1)[0] for a in sys.argv[2:]] or list(VARIANTS)
    procs = []
    for name in want:
        if name == "base":
            continue
        obj = os.path.join(OUT, name + ".o")
        cmd = [b._hipcc()] + b.FLAGS + b.EXTRA.get(SRC, []) + VARIANTS[name] + ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(b.CSRC, SRC), "-o", obj]
        procs.append((name, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for name, obj, p in procs:
        log = p.communicate()[0]
        if p.returncode:
            print(log)
            raise SystemExit(f"{name}: build failed")
        scratch = [l for l in log.splitlines() if "ScratchSize" in l and "ScratchSize [bytes/lane]: 0" not in l]
        regs = [l.split("remark:")[1].strip() for l in log.splitlines() if "VGPRs:" in l or "AGPRs:" in l][:2]
        if scratch:
            print(f"{name}: REFUSED (scratch) {scratch[0]}")
            continue
        lib = os.path.join(OUT, f"lib_{name}.so")
        objs = [b._obj(s) if s != SRC else obj for s in b.SOURCES]
        subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
        print(f"{name}: {lib} {regs}")


CHILD = r"""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, %r)
from season_nerf_amd import _lib
from season_nerf_amd.synthetic import synthetic_state_dict
import season_nerf_amd as sn
L = _lib.lib()
W, NC, R, S = 512, 4, 4096, 96
net = sn.T_NeRF(W, NC)
sd = synthetic_state_dict(net, 0)
m = L.snerf_model_create(W, NC)
assert L.snerf_model_set_precision(m, int(os.environ.get("KS_PREC", "0"))) == 0
for k, v in sd.items():
    if v.is_floating_point():
        a = np.ascontiguousarray(v.numpy(), dtype=np.float32)
        assert L.snerf_model_set_tensor(m, k.encode(), a.ctypes.data, a.size) == 0
assert L.snerf_model_finalize(m) == 0, L.snerf_last_error()
rng = np.random.Generator(np.random.PCG64(0))
t = lambda a: torch.tensor(a, dtype=torch.float32, device="cuda")
top = t(np.concatenate([rng.uniform(-1, 1, (R, 2)), np.ones((R, 1))], 1)); bot = t(np.concatenate([rng.uniform(-1, 1, (R, 2)), -np.ones((R, 1))], 1))
sun = rng.uniform(0, 1, (R, 3)); sun = t(sun / np.linalg.norm(sun, axis=1, keepdims=True))
tv = torch.linspace(0, 1, S + 1)[:-1].float().cuda()
cls = torch.softmax(torch.rand(R, NC, device="cuda"), 1)
rho, sv, col = torch.empty(R * S, device="cuda"), torch.empty(R * S, device="cuda"), torch.empty(R * S, 3, device="cuda")
fo = _lib.FieldOut(d_rho=rho.data_ptr(), d_solar_vis=sv.data_ptr(), d_col=col.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
variant = int(os.environ.get("KS_VARIANT", "0"))
run = lambda: _lib.check(L.snerf_field_forward_rays(m, variant, R, S, top.data_ptr(), bot.data_ptr(), tv.data_ptr(), 1, sun.data_ptr(), cls.data_ptr(), C.byref(fo), st), "field")
for _ in range(5): run()
torch.cuda.synchronize()
best = []
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    best.append(e0.elapsed_time(e1) / 20)
ref = col.clone(); run(); torch.cuda.synchronize(); nd = int((ref != col).sum())          # launch-to-launch reproducibility of the colour output
print("%%-16s %%.3f ms (runs %%s) finite=%%s  col elements differing between two launches: %%d" %% (os.environ.get("KS_NAME", "?"), min(best), " ".join("%%.3f" %% b for b in best), bool(torch.isfinite(rho).all()), nd), flush=True)
t0 = time.time()
while time.time() - t0 < float(os.environ.get("KS_LOOP", "0")):
    for _ in range(50): run()
    torch.cuda.synchronize()
""" % REPO


def run():
    want = [a.split("=", 1)[0] for a in sys.argv[2:]] or sorted(f[4:-3] for f in os.listdir(OUT) if f.startswith("lib_")) + ["base"]
    for name in want:
        lib = os.path.join(OUT, f"lib_{name}.so") if name != "base" else os.path.join(REPO, "season_nerf_amd", "libseason_nerf_hip.so")
        if not os.path.exists(lib):
            print(f"{name}: not built")
            continue
        env = dict(os.environ, SNERF_LIB=lib, KS_NAME=name)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
        print((r.stdout.strip().splitlines() or ["(no output)"])[-1] if r.returncode == 0 else f"{name}: FAILED\n{r.stderr[-800:]}", flush=True)


def power():
    """`run` with the shader clock and package power (rocm-smi twice a second) while each variant loops for 6 s"""
    import re
    import time
    want = [a.split("=", 1)[0] for a in sys.argv[2:]] or sorted(f[4:-3] for f in os.listdir(OUT) if f.startswith("lib_")) + ["base"]
    for name in want:
        lib = os.path.join(OUT, f"lib_{name}.so") if name != "base" else os.path.join(REPO, "season_nerf_amd", "libseason_nerf_hip.so")
        if not os.path.exists(lib):
            continue
        p = subprocess.Popen([sys.executable, "-c", CHILD], env=dict(os.environ, SNERF_LIB=lib, KS_NAME=name, KS_LOOP="6"), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        line = p.stdout.readline().strip()
        rows = []
        while p.poll() is None:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
            sclk, pw = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out), re.search(r"Power \(W\): ([\d.]+)", out)
            if sclk and pw:
                rows.append((int(sclk.group(1)), float(pw.group(1))))
            time.sleep(0.5)
        rows = rows[2:-1] or rows
        med = lambda v: sorted(v)[len(v) // 2] if v else float("nan")
        print(f"{line}   sclk {med([r[0] for r in rows])} MHz  power {med([r[1] for r in rows])} W ({len(rows)} samples)", flush=True)


if __name__ == "__main__":
    {"build": build, "run": run, "power": power}[sys.argv[1]]()
