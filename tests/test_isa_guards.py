"""Build-time guard for the hand-issued A stream of gemm_rows_full_kernel (csrc/gemm.hip): between `a8_issue` and `a8_wait` the
destination registers hold nothing, so the compiler must never spill them.  The instantiations without the activation-backward
epilogue must therefore compile without scratch traffic at all; the activation-backward ones do spill in their epilogue and are
protected in the source by `a8_wait<0>` on every slot before it (checked here as text).  Cross-compiles for gfx950; no GPU."""
import os
import re
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "season_nerf_amd", "csrc", "gemm.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _vregs(text):
    out = set()
    for a, b in re.findall(r"v\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", text):
        out.add(int(a))
    return out


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not available")
def test_full_tile_kernels_do_not_spill_pending_loads(tmp_path):
    out = tmp_path / "gemm.s"
    subprocess.check_call([HIPCC if os.path.exists(HIPCC) else "hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "-ffp-contract=off",
                           "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-Wno-unused-command-line-argument", "-S", "--cuda-device-only", "-o", str(out), SRC],
                          timeout=900)
    lines = out.read_text().split("\n")
    seen, i = 0, 0
    while i < len(lines):
        m = re.match(r"^_ZN5snerf21gemm_rows_full_kernelILi(\d)ELi(\d)ELi(\d)ELi(\d)EEEvNS_5GemmXE:", lines[i])
        if not m:
            i += 1
            continue
        j = i
        while "s_endpgm" not in lines[j]:
            j += 1
        body = lines[i:j]
        nt, pf, aol, act = (int(x) for x in m.groups())
        scratch = sum("scratch_" in l for l in body)
        waits = sum("a8_wait" in l for l in body)
        assert waits >= pf, (nt, pf, aol, act, waits)
        if not act:
            assert scratch == 0, f"gemm_rows_full_kernel<{nt},{pf},{aol},{act}> spills ({scratch} scratch ops) while A loads are pending"
        # every variant: no spill and no register move may touch a register with a hand-issued load in flight (program-order scan)
        pending, in_asm = set(), False
        for l in body:
            t = l.strip()
            code = t.split(";")[0]
            if t.startswith(";;#ASMSTART"):
                in_asm = True
            elif t.startswith(";;#ASMEND"):
                in_asm = False
            elif in_asm and t.startswith("global_load_dwordx4"):
                pending |= _vregs(t.split(",")[0])
            elif in_asm and "a8_wait" in t:
                pending -= _vregs(t.split("a8_wait")[1])
            elif "scratch_" in code or code.startswith("v_mov") or code.startswith("v_accvgpr"):
                hit = _vregs(code) & pending
                assert not hit, f"gemm_rows_full_kernel<{nt},{pf},{aol},{act}> moves / spills v{sorted(hit)} while its load is in flight: {t}"
        seen += 1
        i = j
    assert seen >= 18
    src = open(SRC).read()
    k = src.index("void gemm_rows_full_kernel")
    assert re.search(r"if \(ACT\) \{[^}]*a8_wait<0>\(px\[d\], py\[d\]\);", src[k:], re.S), "the activation-backward variant must drain its prefetch before the epilogue"
