"""Build-time guard for the hand-issued A stream of gemm_rows_full_kernel and gemm_rows16_kernel (csrc/gemm.hip): between `a8_issue` and `a8_wait` the
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
SRC16 = os.path.join(REPO, "season_nerf_amd", "csrc", "gemm16.hip")
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
    lines = []
    for n, src_file in enumerate((SRC, SRC16)):
        out = tmp_path / f"gemm{n}.s"
        subprocess.check_call([HIPCC if os.path.exists(HIPCC) else "hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "-ffp-contract=off",
                               "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-Wno-unused-command-line-argument", "-S", "--cuda-device-only", "-o", str(out), src_file],
                              timeout=900)
        lines += out.read_text().split("\n")
    seen, i = 0, 0
    while i < len(lines):
        m = re.match(r"^_ZN5snerf(?:21gemm_rows_full_kernel|18gemm_rows16_kernel)ILi(\d)ELi(\d)ELi(\d)ELi(\d)EEEvNS_5GemmXE:", lines[i])
        if not m:
            i += 1
            continue
        kname = "gemm_rows16_kernel" if "rows16" in lines[i] else "gemm_rows_full_kernel"      # (gemm_wreg_kernel loads by LDS-DMA: no register holds a load in flight)
        wait = "a8_wait" if kname == "gemm_rows_full_kernel" else "a16_wait"
        j = i
        while "s_endpgm" not in lines[j]:
            j += 1
        body = lines[i:j]
        nt, pf, aol, act = (int(x) for x in m.groups())
        scratch = sum("scratch_" in l for l in body)
        waits = sum(wait in l for l in body)
        assert waits >= pf, (kname, nt, pf, aol, act, waits)
        if not act:
            assert scratch == 0, f"{kname}<{nt},{pf},{aol},{act}> spills ({scratch} scratch ops) while A loads are pending"
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
            elif in_asm and wait in t:
                pending -= _vregs(t.split(wait)[1])
            elif "scratch_" in code or code.startswith("v_mov") or code.startswith("v_accvgpr"):
                hit = _vregs(code) & pending
                assert not hit, f"{kname}<{nt},{pf},{aol},{act}> moves / spills v{sorted(hit)} while its load is in flight: {t}"
        seen += 1
        i = j
    assert seen >= 18 + 7
    src = open(SRC).read()
    src16 = open(SRC16).read()
    k16 = src16.index("void gemm_rows16_kernel")
    assert re.search(r"if \(ACT\) \{[^}]*a16_wait(?:_slot)?<0>\(", src16[k16:], re.S), "the activation-backward variant of the 16x16x32 kernel must drain its prefetch before the epilogue"
    k = src.index("void gemm_rows_full_kernel")
    assert re.search(r"if \(ACT\) \{[^}]*a8_wait<0>\(px\[d\], py\[d\]\);", src[k:], re.S), "the activation-backward variant must drain its prefetch before the epilogue"


# ---- register contracts of the SHIPPED code objects (read from libseason_nerf_hip.so: no compile) ----
def _device_code_objects(path):
    """gfx950 ELF images inside the library's .hip_fatbin section (one clang offload bundle per translation unit)."""
    import struct
    data = open(path, "rb").read()
    assert data[:4] == b"\x7fELF" and data[4] == 2
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    sec = [struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize) for i in range(shnum)]
    strtab = sec[shstrndx]
    fat = next(s for s in sec if data[strtab[4] + s[0]:].split(b"\0", 1)[0] == b".hip_fatbin")
    blob = data[fat[4]:fat[4] + fat[5]]
    magic, pos, out = b"__CLANG_OFFLOAD_BUNDLE__", 0, []
    while (i := blob.find(magic, pos)) >= 0:
        n, = struct.unpack_from("<Q", blob, i + 24)
        p = i + 32
        for _ in range(n):
            off, size, ts = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + ts].decode()
            p += 24 + ts
            if "gfx950" in triple and size:
                out.append(blob[i + off:i + off + size])
        pos = i + 24
    return out


def _kernel_metadata(elf):
    """amdhsa.kernels of one device ELF (NT_AMDGPU_METADATA note, msgpack)."""
    import struct
    import msgpack
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, _ = struct.unpack_from("<HHH", elf, 0x3A)
    for i in range(shnum):
        _, typ, _, _, off, size, *_ = struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize)
        if typ != 7:            # SHT_NOTE
            continue
        p = off
        while p < off + size:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            name = elf[p:p + namesz]
            p += (namesz + 3) & ~3
            if ntype == 32 and name.startswith(b"AMDGPU"):
                return msgpack.unpackb(elf[p:p + descsz], raw=False)["amdhsa.kernels"]
            p += (descsz + 3) & ~3
    return []


def test_shipped_fused_kernels_hold_their_register_contracts():
    """The fused MLP kernels are built around facts about hipcc's register allocation: none may touch scratch,
    the two-waves-per-SIMD kernel must fit 256 registers, the W = 512 kernel owns all 256 AGPRs by number.  Checked on the library
    the tests (and the GPU box) actually load, so a compiler or source change that breaks one shows up without a GPU."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("snerf_build", os.path.join(REPO, "season_nerf_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    b.build()
    kernels = {}
    for elf in _device_code_objects(b.LIB):
        for k in _kernel_metadata(elf):
            kernels[k[".name"]] = k
    fused = {n: k for n, k in kernels.items() if re.match(r"_ZN5snerf\d+mlp_(i8x2_|i8_|ks_)?kernelI", n)}
    assert len(fused) >= 24, sorted(fused)
    for n, k in fused.items():
        assert k[".private_segment_fixed_size"] == 0, (n, "uses scratch")      # (.vgpr_spill_count > 0 with no scratch = parked in AGPRs: fine)
    x2 = {n: k for n, k in fused.items() if "mlp_i8x2_kernel" in n}
    assert len(x2) == 8            # widths 64 / 256 x variants 0..3 (3 = ray visibility)
    for n, k in x2.items():       # two waves per SIMD: 256 registers each, nothing parked
        assert k[".vgpr_count"] <= 256 and k[".vgpr_spill_count"] == 0 and k[".max_flat_workgroup_size"] == 512, (n, k[".vgpr_count"])
    w512 = {n: k for n, k in fused.items() if re.search(r"mlp_i8_kernelILi0ELi512E", n)}
    assert len(w512) == 4
    for n, k in w512.items():     # hidden activations live in AGPRs addressed by number: the whole AGPR file is reserved
        assert k[".agpr_count"] == 256 and k[".vgpr_count"] <= 512 and k[".vgpr_spill_count"] == 0, (n, k[".agpr_count"], k[".vgpr_count"])
    ks = {n: k for n, k in fused.items() if "mlp_ks_kernel" in n}
    assert len(ks) == 4            # width 512, bf16x3, K split over wave pairs: variants 0..3
    for n, k in ks.items():       # one wave per SIMD, ~440 of the 512 registers (a handful parked in AGPRs by hipcc: fine - scratch-free is asserted above)
        assert k[".vgpr_count"] <= 512 and k[".vgpr_spill_count"] <= 16 and k[".max_flat_workgroup_size"] == 256, (n, k[".vgpr_count"], k[".vgpr_spill_count"])
    # row GEMMs of the training engine: the variants without the activation-backward epilogue are scratch-free
    full = {n: k for n, k in kernels.items() if "gemm_rows_full_kernel" in n}
    assert len(full) >= 18
    for n, k in full.items():
        act = int(re.search(r"ILi\dELi\dELi\dELi(\d)E", n).group(1))
        if not act:
            assert k[".private_segment_fixed_size"] == 0, n


def test_w512_kernel_agprs_are_touched_only_by_the_hand_written_instructions(tmp_path):
    """mlp_i8_kernel<FIELD, 512, *> parks its hidden activations in AGPRs addressed BY NUMBER (v_accvgpr_write a[n] at the digit
    split, v_mfma_i32_32x32x32_i8 reading a[n:n+3] as its B operand, all through inline asm).  The one asm clobber that reserves
    a0..a255 does not stop the register allocator from placing its own values there afterwards (AV-class operands, copies,
    rematerialisation) - which would silently corrupt the parked activations.  So: disassemble the shipped code objects and allow
    an AGPR operand nowhere but in those two instruction forms."""
    import importlib.util
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    spec = importlib.util.spec_from_file_location("snerf_build", os.path.join(REPO, "season_nerf_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    b.build()
    seen = 0
    for n, elf in enumerate(_device_code_objects(b.LIB)):
        names = [k[".name"] for k in _kernel_metadata(elf) if re.search(r"mlp_i8_kernelILi0ELi512E", k[".name"])]
        if not names:
            continue
        f = tmp_path / f"co{n}.elf"
        f.write_bytes(elf)
        dis = subprocess.run([objdump, "-d", "--mcpu=gfx950", str(f)], capture_output=True, text=True, timeout=600).stdout
        for name in names:
            m = re.search(r"^[0-9a-f]+ <" + re.escape(name) + r">:\n(.*?)(?=^[0-9a-f]+ <|\Z)", dis, re.S | re.M)
            assert m, name
            writes = mfma_b = 0
            for line in m.group(1).split("\n"):
                code = line.split("//")[0].strip()
                if not code or not re.search(r"\ba(\d+|\[\d+:\d+\])", code):
                    continue
                op = code.split()[0]
                args = code[len(op):]
                if op == "v_accvgpr_write_b32":
                    assert re.match(r"\s*a\d+, v\d+\s*$", args), (name, code)        # park: AGPR <- VGPR
                    writes += 1
                elif op == "v_mfma_i32_32x32x32_i8":
                    d, a_, b_, c_ = [x.strip() for x in args.split(",")[:4]]
                    # weights (A) and accumulators in VGPRs, the parked activations (B) in AGPRs; C = 0 starts an accumulation
                    assert d.startswith("v[") and a_.startswith("v[") and b_.startswith("a[") and (c_.startswith("v[") or c_ == "0"), (name, code)
                    mfma_b += 1
                else:
                    raise AssertionError(f"{name}: AGPR operand outside the hand-written forms: {code}")
            assert writes > 100 and mfma_b > 100, (name, writes, mfma_b)
            seen += 1
    assert seen == 4               # variants 0..3 (3 = ray visibility)


def test_asm_mfma_results_are_not_read_early(tmp_path):
    """hipcc pads the wait states the ISA asks for around its OWN MFMAs; an MFMA issued through inline asm is opaque to it, so csrc/kernels_i8.hip places the
    consumers of its hand-issued MFMAs by hand (>= 2 further MFMAs or an s_nop pad in between).  Round 6 found the pad of every layer's LAST block ineffective in
    the shipped width-512 kernels: a bare `asm volatile("s_nop ..." ::: "memory")` orders nothing against register-only VALU code, hipcc hoisted the epilogue
    above it, and the first elements were read 3 instructions behind the MFMA still forming them - seen on the GPU as launch-to-launch differences of the
    seasonal-adjust outputs (tools/ks_race.py).  The pad now carries the accumulators; this test scans the disassembly of the shipped library
    (tools/isa_hazards.py) so that the hazard cannot come back unseen, with a hand-written negative control for the scanner itself."""
    import importlib.util
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    spec = importlib.util.spec_from_file_location("isa_hazards", os.path.join(REPO, "tools", "isa_hazards.py"))
    hz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hz)
    # negative control: the shipped bug in miniature (result read 2 instructions behind the asm MFMA), its fixed form, and the pipelined form
    mf = "v_mfma_i32_32x32x32_i8 v[32:47], v[72:75], a[120:123], v[32:47]"
    other = "v_mfma_i32_32x32x32_i8 v[0:15], v[72:75], v[84:87], v[0:15]"
    assert len(hz.scan([mf, "v_add_u32_e32 v52, 0x17700, v103", "v_lshl_add_u32 v0, v0, 8, v32"])[1]) == 1
    assert len(hz.scan([mf, "v_sin_f32_e32 v33, v7"])[1]) == 1                                   # overwriting a result register is as bad
    assert hz.scan([mf, "s_nop 15", "s_nop 7", "v_lshl_add_u32 v0, v0, 8, v32"])[1] == []
    assert hz.scan([mf, other, other, "v_lshl_add_u32 v0, v0, 8, v32"])[1] == []
    assert len(hz.scan([mf, other, "v_lshl_add_u32 v0, v0, 8, v32"])[1]) == 1
    spec = importlib.util.spec_from_file_location("snerf_build", os.path.join(REPO, "season_nerf_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    b.build()
    seen = 0
    for n, elf in enumerate(_device_code_objects(b.LIB)):
        names = [k[".name"] for k in _kernel_metadata(elf) if re.search(r"mlp_i8_kernelILi0ELi512E", k[".name"])]
        if not names:
            continue
        f = tmp_path / f"hz{n}.elf"
        f.write_bytes(elf)
        dis = subprocess.run([objdump, "-d", "--mcpu=gfx950", str(f)], capture_output=True, text=True, timeout=600).stdout
        for name in names:
            m = re.search(r"^[0-9a-f]+ <" + re.escape(name) + r">:\n(.*?)(?=^[0-9a-f]+ <|\Z)", dis, re.S | re.M)
            assert m, name
            body = [c for c in (line.split("//")[0].strip() for line in m.group(1).split("\n")) if c]
            n_asm, bad = hz.scan(body)
            assert n_asm > 5000 and not bad, (name, n_asm, bad[:3])
            seen += 1
    assert seen == 4


def test_areg_gemm_agprs_are_touched_only_by_the_hand_written_instructions(tmp_path):
    """gemm_areg_kernel (csrc/gemm_areg.hip) keeps the 32 x N accumulator tile of a wave in AGPRs addressed BY NUMBER (v_mfma_f32_32x32x16_bf16 with
    C / D = a[16 T : 16 T + 15], read out by v_accvgpr_read in the epilogue, both through inline asm).  The register allocator may park its own values in
    AGPRs whenever the architectural VGPRs run short (it did, twice, while the kernel was written: an address register per table row hoisted out of the tile
    loop, and a four-deep weight prefetch) - silently corrupting the accumulators.  So: no scratch, and no AGPR operand anywhere but in those two forms."""
    import importlib.util
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    spec = importlib.util.spec_from_file_location("snerf_build", os.path.join(REPO, "season_nerf_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    b.build()
    seen = 0
    for n, elf in enumerate(_device_code_objects(b.LIB)):
        meta = {k[".name"]: k for k in _kernel_metadata(elf) if "gemm_areg_kernel" in k[".name"]}
        if not meta:
            continue
        f = tmp_path / f"areg{n}.elf"
        f.write_bytes(elf)
        dis = subprocess.run([objdump, "-d", "--mcpu=gfx950", str(f)], capture_output=True, text=True, timeout=600).stdout
        for name, k in meta.items():
            assert k[".private_segment_fixed_size"] == 0 and k[".vgpr_spill_count"] == 0, (name, "scratch / spills")
            nt, _, pfa = (int(v) for v in re.search(r"kernelILi(\d+)ELi(\d+)ELi(\d+)E", name).groups())
            m = re.search(r"^[0-9a-f]+ <" + re.escape(name) + r">:\n(.*?)(?=^[0-9a-f]+ <|\Z)", dis, re.S | re.M)
            assert m, name
            reads = mfmas = 0
            for line in m.group(1).split("\n"):
                code = line.split("//")[0].strip()
                if not code or not re.search(r"\ba(\d+|\[\d+:\d+\])", code):
                    continue
                op = code.split()[0]
                args = code[len(op):]
                if op == "v_accvgpr_read_b32":
                    assert re.match(r"\s*v\d+, a\d+\s*$", args), (name, code)
                    reads += 1
                elif op == "v_mfma_f32_32x32x16_bf16":
                    d, a_, b_, c_ = [x.strip() for x in args.split(",")[:4]]
                    assert d.startswith("a[") and a_.startswith("v[") and b_.startswith("v[") and (c_ == d or c_ == "0"), (name, code)
                    mfmas += 1
                else:
                    raise AssertionError(f"{name}: AGPR operand outside the hand-written forms: {code}")
            # epilogue inside the first k-step of a row tile + the last tile's own: 2 x 16 registers per n-tile; 1 + PFA k-step bodies of 3 MFMAs per n-tile
            assert reads == 2 * 16 * nt and mfmas == (1 + pfa) * 3 * nt, (name, reads, mfmas)
            seen += 1
    # {N = 256, 512} x {plain, activation on load} x {8, 4 k-steps in flight} + the four activation-backward forms + the two forward forms with two waves per SIMD
    # (N = 512 as two column halves of eight n-tiles: 128 AGPRs and only 128 architectural registers per wave - where hipcc's parking is one live value away)
    assert seen == 14


def test_no_runtime_memory_operation_can_reach_a_captured_step():
    """a hipMemsetAsync / hipMemcpyAsync inside a captured training step is a hipGraph MEMSET / MEMCPY node, and MEMSET nodes replay wrongly on ROCm 7.2
    (tools/graph_memset_min.py).  Every copy and fill of the engine goes through the two wrappers of csrc/train.cpp, which launch kernels; the only stream-ordered
    runtime memory operations left in the C++ / HIP sources are the two calls inside those wrappers behind SNERF_TRAIN_MEMOPS=1 (the reproduction switch).  The graph-level guard on the GPU is tests/test_gpu_graph_nodes.py; this one fails at the source, without a GPU."""
    import glob
    import re
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "season_nerf_amd", "csrc")
    hits = []
    for f in sorted(glob.glob(os.path.join(csrc, "*"))):
        if not f.endswith((".cpp", ".hip", ".h")):
            continue
        for n, line in enumerate(open(f), 1):
            code = line.split("//")[0]
            if re.search(r"\bhip(Memset|Memcpy)\w*Async\s*\(", code):
                hits.append((os.path.basename(f), n, code.strip()))
    assert len(hits) == 2 and all(h[0] == "train.cpp" and h[2].startswith("if (train_memops()) return hip") for h in hits), hits
