#!/usr/bin/env python3
"""Static scan of a hipcc --save-temps listing for the hazard hipcc does NOT pad: a hand-issued (inline-asm) MFMA whose result registers are read
or overwritten by vector code before the matrix pipe can have delivered them.  hipcc inserts the wait states the ISA requires around its OWN MFMAs;
an asm MFMA is opaque to it (csrc/kernels_i8.hip mfma_asm: the B operand is an AGPR tuple addressed by number - that is how they are recognised
here), so the source places consumers by hand: >= 2 further MFMAs (>= 64 cycles of pipe time) or an explicit s_nop pad in between.

    python3 tools/isa_hazards.py <listing.s> <kernel-name-substring>      -> prints the violations, exit code 1 if any

Rule checked, per asm MFMA P with result registers D: until 2 more MFMAs have been issued, no VALU / LDS-store / global-store instruction may read
a register of D, and none may write one, unless at least PAD wait states of s_nop lie between P and it (PAD = 18: a 16-pass result)."""
import re
import sys

PAD = 18


def kernel_body(lines, needle):
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and needle in l)
    end = next(i for i in range(start + 1, len(lines)) if lines[i].lstrip().startswith("s_endpgm"))
    return [l.strip() for l in lines[start + 1:end] if l.strip() and l.strip()[0] not in ";." and not l.strip().endswith(":")]


def vregs(tok):
    """registers named in an operand: ('v', n) architectural, ('a', n) accumulation"""
    out = set()
    for m in re.finditer(r"\b([va])\[(\w+):(\w+)\]", tok):
        out |= {(m.group(1), k) for k in range(int(m.group(2), 0), int(m.group(3), 0) + 1)}
    for m in re.finditer(r"\b([va])\[(\w+)\]", tok):
        out.add((m.group(1), int(m.group(2), 0)))
    for m in re.finditer(r"\b([va])(\d+)\b", tok):
        out.add((m.group(1), int(m.group(2))))
    return out


def scan(body):
    pending, bad, n_asm = [], [], 0          # pending: [index, result registers, MFMAs issued since, s_nop wait states since, text]
    for i, l in enumerate(body):
        op = l.split()[0]
        parts = [p.strip() for p in l[len(op):].split(",")]
        if op == "s_nop":
            for p in pending:
                p[3] += int(parts[0], 0) + 1
            continue
        is_mfma = op.startswith("v_mfma")
        reads, writes = set(), set()
        if is_mfma:
            for p in parts[1:3]:                # A and B operands (an identical SrcC is forwarded by the hardware)
                reads |= vregs(p)
        elif op.startswith(("ds_write", "global_store", "buffer_store", "global_atomic")):
            for p in parts:
                reads |= vregs(p)
        elif op.startswith("v_") or op.startswith(("ds_read", "global_load", "buffer_load")):
            writes = vregs(parts[0])
            if op.startswith("v_"):
                for p in parts[1:]:
                    reads |= vregs(p)
        for p in pending:
            if p[3] >= PAD:
                continue
            if reads & p[1]:
                bad.append((p[0], i, "read", p[4], l))
            elif writes & p[1] and not is_mfma:
                bad.append((p[0], i, "write", p[4], l))
        if is_mfma:
            for p in pending:
                p[2] += 1
            pending = [p for p in pending if p[2] < 2]
            if re.search(r"\ba\[|\ba\d", l):          # an AGPR operand: issued through inline asm (build.py: -amdgpu-mfma-vgpr-form=1 keeps hipcc's own in VGPRs)
                n_asm += 1
                pending.append([i, vregs(parts[0]), 0, 0, l])
    return n_asm, bad


def main():
    body = kernel_body(open(sys.argv[1]).read().splitlines(), sys.argv[2])
    n_asm, bad = scan(body)
    print(f"{sys.argv[2]}: {n_asm} asm MFMAs, {len(bad)} early accesses of their results")
    for p, i, kind, t, l in bad[:12]:
        print(f"  [{p}] {t[:90]}\n      -> [{i}] {kind}: {l[:100]}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
