"""Build the gfx950 shared library in-tree:  python season-nerf_amd/build.py  [--force]

One hipcc invocation; the .so lands next to this file (git-ignored, but it travels to the GPU box)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libseason_nerf_hip.so")
SOURCES = ["kernels.hip", "api.cpp", "pack.cpp", "gemm.hip", "train_kernels.hip", "train.cpp", "dsm.hip"]
DEPS = SOURCES + ["kernels.h", "pack.h", "program.h", "train.h", os.path.join("..", "..", "include", "season_nerf_hip.h")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-std=c++17", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-ffp-contract=off",
           "-mllvm", "-amdgpu-mfma-vgpr-form=1",    # MFMA accumulators in VGPRs: no v_accvgpr_read per epilogue element
           "-Wno-unused-command-line-argument",
           "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
