#!/usr/bin/env python3
"""Static issue model of a kernel's MFMA stream (one wave per SIMD): for each gap between consecutive MFMAs sum the
issue cost of the instructions in it (MI355X_MICROARCH.md cycle constants: VALU 4, transcendental 8, MFMA holds 8 of
its 32, s_nop N+1) and compare max(32, cost) with the 32-cycle MFMA floor."""
import re
import sys
from collections import Counter


def cost(ins):
    op = ins.split()[0]
    if op.startswith("v_mfma"):
        return 8
    if op in ("v_sin_f32_e32", "v_cos_f32_e32", "v_exp_f32_e32", "v_log_f32_e32", "v_rcp_f32_e32", "v_sqrt_f32_e32"):
        return 8
    if op.startswith("v_") and "f64" in op:
        return 8
    if op.startswith("v_"):
        return 4
    if op.startswith("ds_"):
        return 4
    if op.startswith("global_") or op.startswith("buffer_"):
        return 4
    if op == "s_nop":
        return int(ins.split()[1]) + 1
    if op.startswith("s_waitcnt") or op.startswith("s_barrier"):
        return 0
    if op.startswith("s_"):
        return 1
    return 0


def main(path, kernel):
    lines = open(path).read().split("\n")
    a = [i for i, l in enumerate(lines) if l.startswith(kernel)][0]
    body = []
    for l in lines[a + 1:]:
        if l.startswith("_ZN") or ".end_amdhsa_kernel" in l:
            break
        t = l.split(";")[0].strip()
        if t and not t.startswith(".") and not t.endswith(":"):
            body.append(t)
    gaps, cur, ops = [], 0, Counter()
    n_mfma = 0
    for ins in body:
        op = ins.split()[0]
        ops[re.sub(r"_e32|_e64", "", op)] += 1
        if op.startswith("v_mfma"):
            if n_mfma:
                gaps.append(cur + 8)
            n_mfma += 1
            cur = 0
        else:
            cur += cost(ins)
    tot = sum(max(32, g) for g in gaps)
    print(f"{kernel[:48]}: MFMAs {n_mfma}, floor {32 * len(gaps)} cyc, modelled {tot} cyc ({tot / (32 * len(gaps)):.2f}x), "
          f"gaps >32: {sum(g > 32 for g in gaps)}, sum of excess {sum(max(0, g - 32) for g in gaps)}")
    hist = Counter(min(g // 16 * 16, 160) for g in gaps)
    print("  gap histogram (cycles: count):", dict(sorted(hist.items())))
    print("  top ops:", ops.most_common(22))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "_ZN5snerf10mlp_kernelILi0ELi256ELi0EEE")
