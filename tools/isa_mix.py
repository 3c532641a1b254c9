#!/usr/bin/env python3
"""Static instruction mix of one kernel in a hipcc --save-temps assembly listing, priced with the issue costs of
MI355X_MICROARCH.md ('vector-instruction ISSUE cost': MFMA holds the SIMD's vector issue 8 of its 32 cycles, a transcendental 8,
any other VALU 4).  The fused MLP kernels are fully unrolled per tile, so the static mix of the kernel body IS the per-tile
dynamic mix (a few hundred prologue instructions aside).

    hipcc <FLAGS of season_nerf_amd/build.py> -I include --save-temps -c season_nerf_amd/csrc/kernels_i8x2.hip -o /tmp/x.o
    python3 tools/isa_mix.py kernels_i8x2-hip-amdgcn-amd-amdhsa-gfx950.s mlp_i8x2_kernelILi256ELi0
"""
import collections
import re
import sys

TRANS = ("v_sin_f32", "v_cos_f32", "v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32")


def kernel_body(lines, needle):
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and needle in l)
    end = next(i for i in range(start + 1, len(lines)) if lines[i].lstrip().startswith("s_endpgm"))
    return lines[start + 1:end]


def main():
    lines = open(sys.argv[1]).read().splitlines()
    c, ops = collections.Counter(), collections.Counter()
    for l in kernel_body(lines, sys.argv[2]):
        l = l.strip()
        if not l or l[0] in ";." or l.endswith(":"):
            continue
        op = l.split()[0]
        ops[op] += 1
        if op.startswith("v_mfma"):
            c["mfma"] += 1
        elif op.startswith(TRANS):
            c["trans"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith("s_waitcnt"):
            c["waitcnt"] += 1
        elif op.startswith("s_nop"):
            c["nop"] += 1
        elif op.startswith(("global_", "buffer_", "scratch_")):
            c["vmem"] += 1
        else:
            c["scalar"] += 1
    m = max(1, c["mfma"])
    print({k: c[k] for k in sorted(c)})
    print("per MFMA: " + ", ".join(f"{k} {c[k] / m:.2f}" for k in ("valu", "trans", "lds", "waitcnt", "scalar", "nop", "vmem")))
    vec = 8 + 4 * c["valu"] / m + 8 * c["trans"] / m
    slots = 2 + (c["valu"] + 2 * c["trans"] + c["lds"] + c["waitcnt"] + c["scalar"] + c["nop"] + c["vmem"]) / m
    print(f"vector-issue cycles per MFMA of one wave's stream: {vec:.1f} (the matrix pipe needs 32); "
          f"4-cycle issue slots of the wave per MFMA: {slots:.1f}")
    for op, n in ops.most_common(16):
        print(f"{n:7d} {op}")


if __name__ == "__main__":
    main()
