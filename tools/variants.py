"""A/B builds of one kernel source, cross-compiled HERE (the .so files travel to the GPU box), timed THERE.

    python tools/variants.py build kernels_i8.hip name1="-DFOO=1" name2="-DSNERF_ABLATE -DABL=8" ...
    python tools/variants.py run [--precision i8x3] [--width 256]        # on the GPU box: one process per library

Each variant = build/variants/lib_<name>.so: the named source compiled with the extra flags, linked with the other objects of
the regular build (build/obj, made by season_nerf_amd/build.py).  A build in which a kernel with hand-issued loads spills is refused
(see `one` below); VARIANTS_UNUSED=<substring of the mangled name>[,...] exempts kernels the timed script will not launch.  Only the
ctypes path (`SNERF_LIB`) sees a variant: the torch op library links the in-tree build."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "season_nerf_amd"))
import build as B  # noqa: E402

VAR = os.path.join(REPO, "build", "variants")


def build(src, variants):
    B.build(verbose=False)
    os.makedirs(VAR, exist_ok=True)
    for f in glob.glob(os.path.join(VAR, "*")):
        os.remove(f)

    def one(nv):
        name, flags = nv
        o = os.path.join(VAR, name + ".o")
        subprocess.check_call([B._hipcc()] + B.FLAGS + flags.split() + ["-c", os.path.join(B.CSRC, src), "-o", o])
        # A kernel that keeps hand-issued loads in flight (gemm_rows_full / gemm_rows16 / gemm_wreg: a8_issue / a16_issue) must not touch scratch: a spilled register
        # with a load still landing in it corrupts whatever it was reallocated to - addresses included (GPU memory faults).  Refuse.
        meta = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", o], capture_output=True, text=True).stdout
        asm = subprocess.run([B._hipcc()] + B.FLAGS + flags.split() + ["-S", "--cuda-device-only", "-o", "-", os.path.join(B.CSRC, src)],
                             capture_output=True, text=True).stdout
        import re
        bad = [m.group(1) for m in re.finditer(r"\.amdhsa_kernel (\S*gemm_(?:rows_full|rows16|wreg)\S*)(?:.*?)\.amdhsa_private_segment_fixed_size (\d+)", asm, re.S)
               if int(m.group(2)) > 0 and "ELi1EEEv" not in m.group(1)       # (the activation-backward forms drain their loads before they spill)
               and not any(x and x in m.group(1) for x in os.environ.get("VARIANTS_UNUSED", "").split(","))]      # kernels the run will not launch
        if bad:
            os.remove(o)
            print(f"REFUSED {name}: scratch in {bad} while hand-issued loads are pending", flush=True)
            return
        objs = [o if s == src else B._obj(s) for s in B.SOURCES]
        subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(VAR, f"lib_{name}.so")] + objs)
        os.remove(o)
        print("built", name, flush=True)
    with ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(one, variants))


def run(extra):
    libs = sorted(glob.glob(os.path.join(VAR, "lib_*.so")))
    script = None
    if extra and extra[0] == "--script":      # e.g. --script tools/bench_rows.py 30   (prints its own lines; the variant name is printed first)
        script, extra = extra[1], extra[2:]
    for rep in range(2 if script is None else 1):
        for lib in libs:
            env = dict(os.environ, SNERF_LIB=lib)
            if script:
                print("==", os.path.basename(lib)[4:-3], flush=True)
                subprocess.call([sys.executable, os.path.join(REPO, script)] + extra, env=env)
            else:
                subprocess.call([sys.executable, os.path.join(REPO, "tools", "time_field.py"), "--tag", os.path.basename(lib)[4:-3]] + extra, env=env)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2], [tuple(a.split("=", 1)) for a in sys.argv[3:]])
    else:
        run(sys.argv[2:])
