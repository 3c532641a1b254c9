"""Build the gfx950 shared library in-tree:  python season_nerf_amd/build.py  [--force]

Every source is compiled to its own object (in parallel, cached under build/obj by modification time of the source
and of the headers) and linked into libseason_nerf_hip.so next to this file (git-ignored, but it travels to the
GPU box).  `build(force=True)` - what `__graft_entry__.build()` asks for - recompiles everything: a fresh .so on
disk proves nothing about the toolchain."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(REPO, "build", "obj")
LIB = os.path.join(HERE, "libseason_nerf_hip.so")
SOURCES = ["kernels.hip", "kernels_i8.hip", "kernels_i8x2.hip", "kernels_i8_w512.hip", "kernels_ks.hip", "api.cpp", "pack.cpp", "gemm.hip", "gemm16.hip", "gemm_areg.hip", "train_kernels.hip", "train.cpp", "dsm.hip"]
HEADERS = ["gemm_common.h", "kernels.h", "mlp_device.h", "mlp_bf16_device.h", "mlp_i8_device.h", "pack.h", "program.h", "train.h", os.path.join("..", "..", "include", "season_nerf_hip.h")]
FLAGS = ["-std=c++17", "-O3", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off",
         "-mllvm", "-amdgpu-mfma-vgpr-form=1",    # MFMA accumulators in VGPRs: no v_accvgpr_read per epilogue element
         "-Wno-unused-command-line-argument"]


EXTRA = {"kernels_i8_w512.hip": ["-mllvm", "-pragma-unroll-threshold=1000000"],      # per-source flags
         "kernels_ks.hip": ["-mllvm", "-pragma-unroll-threshold=1000000"],           # (a 512 -> 512 layer is 256 pairs per wave, fully unrolled)
         "gemm_areg.hip": ["-mllvm", "-pragma-unroll-threshold=1000000"]}           # (its 32 k-steps per n-tile must unroll: register numbers are immediates)
INCLUDED = {"kernels_i8_w512.hip": ["kernels_i8.hip"]}                               # sources a source #includes


def _hipcc():
    return os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _newest_header():
    return max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS if os.path.exists(os.path.join(CSRC, h)))


def _obj(src):
    return os.path.join(OBJ, src + ".o")


def _deps(src):
    """Files the object was built from, as the compiler recorded them (-MD): the source, its #includes, the headers."""
    d = _obj(src) + ".d"
    if not os.path.exists(d):
        return None
    txt = open(d).read().replace("\\\n", " ")
    return [f for f in txt.split(":", 1)[1].split() if f.startswith(REPO)] if ":" in txt else None


def _stale(src, hdr_t):
    o = _obj(src)
    if not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    deps = _deps(src)
    if deps is not None:          # exact: only what this object really includes
        return any((not os.path.exists(f)) or os.path.getmtime(f) > t for f in deps)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in [src] + INCLUDED.get(src, [])) or hdr_t > t


def needs_build():
    if not os.path.exists(LIB):
        return True
    hdr_t = _newest_header()
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, s)) > t for s in SOURCES) or hdr_t > t


def _compile(src, verbose):
    cmd = [_hipcc()] + FLAGS + EXTRA.get(src, []) + ["-MD", "-MF", _obj(src) + ".d", "-c", os.path.join(CSRC, src), "-o", _obj(src)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build(force=False, verbose=True, jobs=None):
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    hdr_t = _newest_header()
    todo = [s for s in SOURCES if force or _stale(s, hdr_t)]
    jobs = jobs or min(len(todo) or 1, max(1, (os.cpu_count() or 2) - 1), 6)
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        list(ex.map(lambda s: _compile(s, verbose), todo))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [_obj(s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


OPS_LIB = os.path.join(HERE, "libseason_nerf_ops.so")
OPS_SRC = os.path.join(CSRC, "ops.cpp")


def build_ops(force=False, verbose=True):
    """The PyTorch custom-op layer (csrc/ops.cpp: TORCH_LIBRARY(season_nerf)): host C++ only, linked against the installed
    torch and the HIP library above; lands next to it (torch.ops.load_library picks it up in season_nerf_amd/ops.py)."""
    import torch
    if not force and os.path.exists(OPS_LIB) and os.path.getmtime(OPS_LIB) > max(os.path.getmtime(OPS_SRC), os.path.getmtime(LIB),
                                                                                  os.path.getmtime(os.path.join(REPO, "include", "season_nerf_hip.h"))):
        return OPS_LIB
    tl = os.path.dirname(torch.__file__)
    cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-I" + os.path.join(tl, "include"),
           "-I" + os.path.join(tl, "include", "torch", "csrc", "api", "include"), "-I/opt/rocm/include", OPS_SRC, "-o", OPS_LIB,
           "-L" + os.path.join(tl, "lib"), "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-L" + HERE, "-lseason_nerf_hip",
           "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OPS_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    build_ops(force="--force" in sys.argv)
    print(LIB)
    print(OPS_LIB)
