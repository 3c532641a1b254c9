"""CPU-only checks of the C-ABI library: it builds/loads, exports every symbol include/season_nerf_hip.h declares,
its host-side error behaviour, and the host packer (no compute calls - those need a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from oracle import season_nerf_oracle as orc

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="session")
def lib():
    import importlib.util
    spec = importlib.util.spec_from_file_location("snerf_build", os.path.join(REPO, "season_nerf_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    b.build()                      # no-op when the .so is up to date (__graft_entry__.build() forces a full compile);
    b.build_ops()                  # hipcc cross-compiles without a GPU
    import season_nerf_amd as sn
    return sn._lib.lib()


def test_exports_match_header(lib):
    hdr = open(os.path.join(REPO, "include", "season_nerf_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(snerf_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    import season_nerf_amd as sn
    assert sorted(sn._lib.EXPORTS) == declared
    assert lib.snerf_abi_version() == 8


def test_custom_op_library_registers_without_a_gpu():
    """TORCH_LIBRARY(season_nerf): the op layer loads on a CPU-only box, exposes its schemas, and a model object packs on the host."""
    import season_nerf_amd as sn
    ns = sn.ops.load()
    for op in ["group_fwd", "points_fwd", "render_fwd", "composite", "composite_sweep", "fused_adam_", "model_from_handle"]:
        assert hasattr(ns, op), op
    assert "Tensor(a!) param" in str(torch.ops.season_nerf.fused_adam_.default._schema)
    m = torch.classes.season_nerf.Model(64, 4, "i8x3")
    assert (m.width(), m.classes(), m.precision()) == (64, 4, 2)
    with pytest.raises(RuntimeError):
        m.set_tensor("adjust_col.bias", torch.zeros(12, dtype=torch.float64))
    m.set_tensor("adjust_col.bias", torch.zeros(12))
    if not torch.cuda.is_available():
        with pytest.raises((RuntimeError, NotImplementedError)):          # no CPU backend: the ops never compute on the host
            torch.ops.season_nerf.group_fwd(m, torch.zeros(2, 4), torch.zeros(2, 3))


def test_error_paths(lib):
    assert not lib.snerf_model_create(96, 4)                       # no kernel compiled for this width
    assert b"96" in lib.snerf_last_error()
    assert not lib.snerf_model_create(64, 9)
    m = lib.snerf_model_create(64, 4)
    assert m
    n = C.c_size_t()
    rc = lib.snerf_model_pack_host(m, 0, None, C.byref(n), None, None)
    assert rc == -2 and b"missing tensor" in lib.snerf_last_error()       # SNERF_E_MISSING
    a = np.zeros(5, dtype=np.float32)
    assert lib.snerf_model_set_tensor(m, b"G_NeRF_net.fc1.linear.weight", a.ctypes.data, a.size) == 0
    rc = lib.snerf_model_pack_host(m, 0, None, C.byref(n), None, None)
    assert rc == -1 and b"elements" in lib.snerf_last_error()             # wrong shape -> SNERF_E_INVALID
    # forward before finalize
    assert lib.snerf_group_forward(m, 4, None, None, None, None, None, None) == -4
    if not torch.cuda.is_available():
        sd = orc.init_weights(64, 4, 0)
        for k, v in sd.items():
            if v.is_floating_point():
                arr = np.ascontiguousarray(v.numpy())
                assert lib.snerf_model_set_tensor(m, k.encode(), arr.ctypes.data, arr.size) == 0
        assert lib.snerf_model_finalize(m) == -3                    # SNERF_E_HIP: no device here
    lib.snerf_model_destroy(m)


def test_product_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import season_nerf_amd as sn
    net = sn.T_NeRF(64, 4).eval()
    with pytest.raises(RuntimeError):
        net.forward(torch.zeros(2, 3), torch.ones(2, 3), torch.ones(2, 4))


# ------------------------------------------------------------------------------------------------------------
# Double-entry check of the packer: decode the fragment streams with an independent Python statement of the layout
# rules (program.h) and evaluate the folded network in float64.  Must reproduce the oracle's forward.
def bf16_to_f64(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32).astype(np.float64)


def slot_H(kk):
    s, h, j = kk // 16, (kk % 16) // 8, kk % 8
    return 32 * (s >> 1) + 16 * (s & 1) + 8 * (j >> 2) + 4 * h + (j & 3)


def slot_pepos(kk):
    s, h, j = kk // 16, (kk % 16) // 8, kk % 8
    e = 8 * s + j
    if e < 30:
        d, r = divmod(e, 10)
        return 3 + 20 * d + 10 * (r & 1) + 5 * h + r // 2
    return (0 if h == 0 else 2) if e == 30 else (1 if h == 0 else -1)


def slot_pesun(kk):
    s, h, j = kk // 16, (kk % 16) // 8, kk % 8
    e = 8 * s + j
    if e < 12:
        d, r = divmod(e, 4)
        return 3 + 8 * d + 4 * (r & 1) + 2 * h + r // 2
    if e == 12:
        return 0 if h == 0 else 2
    return (1 if h == 0 else -1) if e == 13 else -1


def slot_petime(kk):
    s, h, j = kk // 16, (kk % 16) // 8, kk % 8
    if s != 0 or j > 4:
        return -1
    if j == 0:
        return h
    q = j - 1
    return 2 + 4 * h + 2 * (q & 1) + (q >> 1)


def acc_row(i, h):
    return (i & 3) + 8 * (i >> 2) + 4 * h


class Decoder:
    def __init__(self, lib, model, prog):
        ns, nb = C.c_size_t(), C.c_size_t()
        assert lib.snerf_model_pack_host(model, prog, None, C.byref(ns), None, C.byref(nb)) == 0
        self.stream = np.zeros(ns.value, dtype=np.uint8)
        self.bias = np.zeros(nb.value, dtype=np.float32)
        assert lib.snerf_model_pack_host(model, prog, self.stream.ctypes.data, C.byref(ns), self.bias.ctypes.data, C.byref(nb)) == 0
        self.chunk, self.boff = 0, 0

    def layer(self, n_out, ks):
        """dense folded matrix [n_out, 16*ks] in slot order and bias [n_out] of the next layer of the stream"""
        nbk = n_out // 32
        base = self.chunk * 16384
        Wd = np.zeros((n_out, 16 * ks))
        u16 = self.stream.view(np.uint16)
        for b in range(nbk):
            for s in range(ks):
                o = (base + (b * ks + s) * 2048) // 2
                hi = bf16_to_f64(u16[o:o + 512]).reshape(64, 8)
                lo = bf16_to_f64(u16[o + 512:o + 1024]).reshape(64, 8)
                v = hi + lo
                for lane in range(64):
                    r, h = lane & 31, lane >> 5
                    Wd[32 * b + r, 16 * s + 8 * h:16 * s + 8 * h + 8] = v[lane]
        bias = np.zeros(n_out)
        for b in range(nbk):
            for h in range(2):
                for i in range(16):
                    bias[32 * b + acc_row(i, h)] = self.bias[self.boff + b * 32 + h * 16 + i]
        self.chunk += (nbk * ks + 7) // 8
        self.boff += n_out
        return Wd, bias


def gather(feat, slot_fn, n_slots):
    """activation/encoding vector in slot order: [N, n_slots]"""
    out = np.zeros((feat.shape[0], n_slots))
    for kk in range(n_slots):
        f = slot_fn(kk)
        if 0 <= f < feat.shape[1]:
            out[:, kk] = feat[:, f]
    return out


@pytest.mark.parametrize("W", [64, 256])
def test_packed_streams_reproduce_the_network(lib, W):
    Cn = 4
    sd = orc.init_weights(W, Cn, seed=7)
    m = lib.snerf_model_create(W, Cn)
    for k, v in sd.items():
        if v.is_floating_point():
            arr = np.ascontiguousarray(v.numpy())
            assert lib.snerf_model_set_tensor(m, k.encode(), arr.ctypes.data, arr.size) == 0
    rng = np.random.Generator(np.random.PCG64(3))
    N = 48
    X = torch.tensor(rng.uniform(-1, 1, (N, 3)))
    sun = rng.uniform(0, 1, (N, 3)); sun /= np.linalg.norm(sun, axis=1, keepdims=True)
    sun = torch.tensor(sun)
    tim = torch.tensor(rng.uniform(-1, 1, (N, 4)))
    sd64 = orc.cast_weights(sd, torch.float64)
    with torch.no_grad():
        ref = orc.forward_separate(sd64, X, sun, tim)
        pe_pos = orc.pe_encode(X, 10).numpy()
        pe_sun = orc.pe_encode(sun, 4).numpy()
        pe_tim = orc.pe_encode(tim[:, 0:2], 2).numpy()
    act = lambda z: np.sin(2 * np.pi * z)
    W2 = W // 2
    # ---------------- field program
    D = Decoder(lib, m, 0)
    P = gather(pe_pos, slot_pepos, 64)
    Wd, b = D.layer(W, 4); h = act(P @ Wd.T + b)
    for _ in range(3):
        Wd, b = D.layer(W, W // 16); h = act(gather(h, slot_H, W) @ Wd.T + b)
    Wd, b = D.layer(W, W // 16 + 4); h = act(np.concatenate([gather(h, slot_H, W), P], 1) @ Wd.T + b)
    for _ in range(3):
        Wd, b = D.layer(W, W // 16); h = act(gather(h, slot_H, W) @ Wd.T + b)
    Wd, b = D.layer(W2, W // 16); x1 = act(gather(h, slot_H, W) @ Wd.T + b)
    Wd, b = D.layer(32, W2 // 16); head = gather(x1, slot_H, W2) @ Wd.T + b
    S_ = gather(pe_sun, slot_pesun, 32)
    Wd, b = D.layer(W2, W2 // 16 + 2); a = act(np.concatenate([gather(x1, slot_H, W2), S_], 1) @ Wd.T + b)
    for _ in range(2):
        Wd, b = D.layer(W2, W2 // 16); a = act(gather(a, slot_H, W2) @ Wd.T + b)
    Wd, b = D.layer(32, W2 // 16); sv = gather(a, slot_H, W2) @ Wd.T + b
    Wd, b = D.layer(W, W2 // 16); y = act(gather(x1, slot_H, W2) @ Wd.T + b)
    for _ in range(2):
        Wd, b = D.layer(W, W // 16); y = act(gather(y, slot_H, W) @ Wd.T + b)
    Wd, b = D.layer(32, W // 16); adj = gather(y, slot_H, W) @ Wd.T + b
    tol = dict(rtol=2e-4, atol=3e-5)      # bf16 hi+lo weights carry ~2^-17 relative representation error
    np.testing.assert_allclose(head[:, 0:3], ref[1].numpy(), **tol)                                   # Col_raw
    np.testing.assert_allclose(np.log1p(np.exp(head[:, 3:4])), ref[0].numpy(), **tol)                 # Rho
    np.testing.assert_allclose(1 / (1 + np.exp(-sv[:, 0:1])), ref[2].numpy(), **tol)                  # Solar_Vis
    adj_rows = np.stack([adj[:, acc_row(i, 0)] for i in range(3 * Cn)], 1).reshape(N, Cn, 3)
    np.testing.assert_allclose(adj_rows, ref[5].numpy(), **tol)
    # ---------------- group program
    D = Decoder(lib, m, 1)
    Wd, b = D.layer(W, 2); t1 = act(gather(pe_tim, slot_petime, 32) @ Wd.T + b)
    Wd, b = D.layer(W, W // 16); t2 = act(gather(t1, slot_H, W) @ Wd.T + b)
    Wd, b = D.layer(32, W // 16); logits = gather(t2, slot_H, W) @ Wd.T + b
    lg = np.stack([logits[:, acc_row(i, 0)] for i in range(Cn)], 1)
    pr = np.exp(lg - lg.max(1, keepdims=True)); pr /= pr.sum(1, keepdims=True)
    np.testing.assert_allclose(pr, ref[4].numpy(), **tol)
    W4p = max(32, (W // 4 + 31) // 32 * 32)
    Wd, b = D.layer(W4p, 2); k1 = act(gather(pe_sun, slot_pesun, 32) @ Wd.T + b)
    Wd, b = D.layer(32, W4p // 16); sky = gather(k1, slot_H, W4p) @ Wd.T + b
    np.testing.assert_allclose(1 / (1 + np.exp(-sky[:, 0:3])), ref[3].numpy(), **tol)
    lib.snerf_model_destroy(m)


# ---- the int8-digit format (program.h FMT_I8): decode digits and tables, run the chain in exact integer arithmetic as the
# kernel does (q = round(32767 h) = 256 a + u, digits (a, u - 128); M = sum T a, X = sum (T b + L a); z = sc (256 M + X) + bias)
def slot8_H(s, h, j):
    return 32 * s + acc_row(j, h)


def slot8_pepos(s, h, j):
    e = 16 * s + j
    if e < 30:
        d, r = divmod(e, 10)
        return 3 + 20 * d + 10 * (r & 1) + 5 * h + r // 2
    return (0 if h == 0 else 2) if e == 30 else (1 if h == 0 else -1)


def slot8_pesun(s, h, j):
    if s != 0:
        return -1
    if j < 12:
        d, r = divmod(j, 4)
        return 3 + 8 * d + 4 * (r & 1) + 2 * h + r // 2
    if j == 12:
        return 0 if h == 0 else 2
    return (1 if h == 0 else -1) if j == 13 else -1


class Decoder8:
    def __init__(self, lib, model):
        ns, nb = C.c_size_t(), C.c_size_t()
        assert lib.snerf_model_pack_host(model, 2, None, C.byref(ns), None, C.byref(nb)) == 0
        self.stream = np.zeros(ns.value, dtype=np.int8)
        self.tab = np.zeros(nb.value, dtype=np.float32)
        assert lib.snerf_model_pack_host(model, 2, self.stream.ctypes.data, C.byref(ns), self.tab.ctypes.data, C.byref(nb)) == 0
        self.chunk, self.toff = 0, 0

    def layer(self, n_out, ks, raw=False):
        """digit matrices T, L [n_out, 32*ks] in slot order (slot = 32 s + 16 h + j), per-row scale / bias, and - for layers
        that read an encoding (raw=True) - the fp32 weights [n_out, 3] of its raw coordinates"""
        nbk = n_out // 32
        base = self.chunk * 16384
        T = np.zeros((n_out, 32 * ks), dtype=np.int64)
        Lo = np.zeros((n_out, 32 * ks), dtype=np.int64)
        for b in range(nbk):
            for s in range(ks):
                o = base + (b * ks + s) * 2048
                t = self.stream[o:o + 1024].reshape(64, 16)
                l = self.stream[o + 1024:o + 2048].reshape(64, 16)
                for lane in range(64):
                    r, h = lane & 31, lane >> 5
                    T[32 * b + r, 32 * s + 16 * h:32 * s + 16 * h + 16] = t[lane]
                    Lo[32 * b + r, 32 * s + 16 * h:32 * s + 16 * h + 16] = l[lane]
        sc, bi = np.zeros(n_out), np.zeros(n_out)
        for b in range(nbk):
            for h in range(2):
                for i in range(16):
                    sc[32 * b + acc_row(i, h)] = self.tab[self.toff + (b * 2 + h) * 32 + i]
                    bi[32 * b + acc_row(i, h)] = self.tab[self.toff + (b * 2 + h) * 32 + 16 + i]
        self.chunk += (nbk * ks + 7) // 8
        self.toff += 2 * n_out
        Wraw = None
        if raw:                                    # [block][lane-half][quad][dim][4 elements]
            Wraw = np.zeros((n_out, 3))
            for b in range(nbk):
                for h in range(2):
                    for i in range(16):
                        for dd in range(3):
                            Wraw[32 * b + acc_row(i, h), dd] = self.tab[self.toff + (((b * 2 + h) * 4 + i // 4) * 3 + dd) * 4 + i % 4]
            self.toff += 3 * n_out
        return T, Lo, sc, bi, Wraw


def gather8(feat, slot_fn, ks):
    out = np.zeros((feat.shape[0], 32 * ks))
    for s in range(ks):
        for h in range(2):
            for j in range(16):
                f = slot_fn(s, h, j)
                if 0 <= f < feat.shape[1]:
                    out[:, 32 * s + 16 * h + j] = feat[:, f]
    return out


def layer8(x_slots, T, Lo, sc, bi, Wraw=None, raw=None):
    q = np.rint(np.clip(x_slots, -1, 1) * 32767).astype(np.int64)
    a = q >> 8
    b = (q & 255) - 128
    M = a @ T.T
    X = b @ T.T + a @ Lo.T
    acc = 256 * M + X
    assert np.abs(acc).max() < 2 ** 31
    z = acc.astype(np.float64) * sc + bi
    if Wraw is not None:
        assert raw is not None
        z = z + raw @ Wraw.T                       # the raw coordinates enter in fp32, whatever their range
    return z


@pytest.mark.parametrize("W", [64, 256])
def test_packed_int8_stream_reproduces_the_network(lib, W):
    Cn = 4
    sd = orc.init_weights(W, Cn, seed=7)
    m = lib.snerf_model_create(W, Cn)
    ns = C.c_size_t()
    assert lib.snerf_model_pack_host(m, 2, None, C.byref(ns), None, None) == -4       # SNERF_E_STATE without the precision mode
    assert lib.snerf_model_set_precision(m, 7) == -1
    assert lib.snerf_model_set_precision(m, 2) == 0 and lib.snerf_model_precision(m) == 2
    for k, v in sd.items():
        if v.is_floating_point():
            arr = np.ascontiguousarray(v.numpy())
            assert lib.snerf_model_set_tensor(m, k.encode(), arr.ctypes.data, arr.size) == 0
    rng = np.random.Generator(np.random.PCG64(3))
    N = 48
    X = torch.tensor(rng.uniform(-1.7, 1.7, (N, 3)))        # beyond the cube: the raw coordinates must not saturate
    sun = rng.uniform(0, 1, (N, 3)); sun /= np.linalg.norm(sun, axis=1, keepdims=True)
    sun = torch.tensor(sun)
    tim = torch.tensor(rng.uniform(-1, 1, (N, 4)))
    sd64 = orc.cast_weights(sd, torch.float64)
    with torch.no_grad():
        ref = orc.forward_separate(sd64, X, sun, tim)
        pe_pos = orc.pe_encode(X, 10).numpy()
        pe_sun = orc.pe_encode(sun, 4).numpy()
    act = lambda z: np.sin(2 * np.pi * z)
    W2 = W // 2
    D = Decoder8(lib, m)
    Xn, Sn = X.numpy(), sun.numpy()
    P = gather8(pe_pos, slot8_pepos, 2)                      # the raw slots gather the (possibly out-of-range) coordinates, on zero weights
    h = act(layer8(P, *D.layer(W, 2, raw=True), raw=Xn))
    for _ in range(3):
        h = act(layer8(gather8(h, slot8_H, W // 32), *D.layer(W, W // 32)))
    h = act(layer8(np.concatenate([gather8(h, slot8_H, W // 32), P], 1), *D.layer(W, W // 32 + 2, raw=True), raw=Xn))
    for _ in range(3):
        h = act(layer8(gather8(h, slot8_H, W // 32), *D.layer(W, W // 32)))
    x1 = act(layer8(gather8(h, slot8_H, W // 32), *D.layer(W2, W // 32)))
    head = layer8(gather8(x1, slot8_H, W2 // 32), *D.layer(32, W2 // 32))
    S_ = gather8(pe_sun, slot8_pesun, 1)
    a = act(layer8(np.concatenate([gather8(x1, slot8_H, W2 // 32), S_], 1), *D.layer(W2, W2 // 32 + 1, raw=True), raw=Sn))
    for _ in range(2):
        a = act(layer8(gather8(a, slot8_H, W2 // 32), *D.layer(W2, W2 // 32)))
    sv = layer8(gather8(a, slot8_H, W2 // 32), *D.layer(32, W2 // 32))
    y = act(layer8(gather8(x1, slot8_H, W2 // 32), *D.layer(W, W2 // 32)))
    for _ in range(2):
        y = act(layer8(gather8(y, slot8_H, W // 32), *D.layer(W, W // 32)))
    adj = layer8(gather8(y, slot8_H, W // 32), *D.layer(32, W // 32))
    tol = dict(rtol=5e-4, atol=2e-4)      # 16-bit fixed point on both operands, fourteen layers deep
    np.testing.assert_allclose(head[:, 0:3], ref[1].numpy(), **tol)                                   # Col_raw
    np.testing.assert_allclose(np.log1p(np.exp(head[:, 3:4])), ref[0].numpy(), **tol)                 # Rho
    np.testing.assert_allclose(1 / (1 + np.exp(-sv[:, 0:1])), ref[2].numpy(), **tol)                  # Solar_Vis
    adj_rows = np.stack([adj[:, acc_row(i, 0)] for i in range(3 * Cn)], 1).reshape(N, Cn, 3)
    np.testing.assert_allclose(adj_rows, ref[5].numpy(), **tol)
    err = np.abs(np.log1p(np.exp(head[:, 3:4])) - ref[0].numpy()).max() / np.abs(ref[0].numpy()).max()
    print(f"W={W}: int8-digit chain vs fp64: density max err / max {err:.2e}")
    lib.snerf_model_destroy(m)


def test_sun_ray_generator_matches_reference_draws(golden_dir):
    """create_solor_rays_uniform (host code, Eval_Tools_2.py:42-108): same numpy / torch RNG draws in the same order as the
    reference's generator, so seeded runs reproduce its random sun rays (micro.npz: the reference's output for seed 5)."""
    import numpy as np
    import torch
    import season_nerf_amd as sn
    g = np.load(os.path.join(golden_dir, "micro.npz"))
    WC = np.array([41.29, -95.9, 300.0])
    H4 = np.array([[310.0, 12.0, 0.0, -11650.0], [-9.0, 240.0, 0.0, 23390.0], [0.0, 0.0, 0.01, -3.0], [0, 0, 0, 1.0]])
    np.random.seed(5)
    torch.manual_seed(5)
    st, en, ve, ti, ae = sn.create_solor_rays_uniform(H4, WC)(48, include_times=True)
    assert st.dtype == en.dtype == ve.dtype == ti.dtype == torch.float32
    np.testing.assert_array_equal(st.numpy(), g["sungen_starts"])
    np.testing.assert_array_equal(en.numpy(), g["sungen_ends"])
    np.testing.assert_array_equal(ve.numpy(), g["sungen_vec"])
    np.testing.assert_allclose(ti.numpy(), g["sungen_times"], rtol=0, atol=1.2e-7)        # numpy vs torch cos/sin: 1 ulp
    np.testing.assert_array_equal(ae, g["sungen_az_el"])
    np.random.seed(5)
    torch.manual_seed(5)
    three = sn.create_solor_rays_uniform(H4, WC)(48)
    assert len(three) == 3 and torch.equal(three[0], st)


def test_synthetic_state_dict_follows_the_init_law():
    """season_nerf_amd.synthetic_state_dict: keys / shapes of T_NeRF.state_dict, ranges of misc.SineLayer.init_weights (misc.py:176-186)."""
    import season_nerf_amd as sn
    net = sn.T_NeRF(64, 4)
    sd = sn.synthetic_state_dict(net, 3)
    ref = net.state_dict()
    assert set(sd) == set(ref) and all(tuple(sd[k].shape) == tuple(ref[k].shape) for k in ref)
    net.load_state_dict(sd)
    w1, w2 = sd["G_NeRF_net.fc1.linear.weight"], sd["G_NeRF_net.fc2.linear.weight"]
    assert float(w1.abs().max()) <= 1 / 63 and float(w1.abs().max()) > 0.9 / 63                       # first layer: U(+-1/in)
    lim = (6 / 64) ** 0.5 / 30
    assert float(w2.abs().max()) <= lim and float(w2.abs().max()) > 0.9 * lim                         # others: U(+-sqrt(6/in)/30)
    assert float(sd["adjust_col.weight"].abs().max()) <= 1 / 8                                        # plain Linear: U(+-1/sqrt(in))
    assert float(sd["G_NeRF_net.fc3.norm.running_var"].min()) >= 0.5
    fresh = sn.synthetic_state_dict(net, 3, bn_stats="identity")
    assert float(fresh["G_NeRF_net.fc3.norm.running_mean"].abs().max()) == 0.0
    assert torch.equal(sn.synthetic_state_dict(net, 3)["time_layer_2.linear.bias"], sd["time_layer_2.linear.bias"])   # deterministic
    names = [n for n, *_ in sn.per_point_layer_shapes(net)]
    assert len(names) == 19 and "time_layer_1" not in names and "G_NeRF_net.fc5" in names


# ------------------------------------------------------------------------------------------------------------
# Pack-time error model of the int8-digit format and SNERF_PREC_AUTO (include/season_nerf_hip.h snerf_i8_estimate): host only.
def _host_model(lib, W, sd, precision):
    m = lib.snerf_model_create(W, 4)
    assert m and lib.snerf_model_set_precision(m, precision) == 0
    for k, v in sd.items():
        if v.is_floating_point():
            arr = np.ascontiguousarray(v.numpy(), dtype=np.float32)
            assert lib.snerf_model_set_tensor(m, k.encode(), arr.ctypes.data, arr.size) == 0
    return m


def _estimate(lib, m):
    import season_nerf_amd as sn
    e = sn._lib.I8Estimate()
    assert lib.snerf_model_i8_estimate(m, C.byref(e)) == 0, lib.snerf_last_error()
    return e


@pytest.mark.parametrize("W", [64, 256, 512])
def test_auto_precision_follows_the_error_bound(lib, W):
    """Weights of the init law clear the bound and run in int8 digits (so do the really-trained fixtures: test_trained_fixtures_clear_the_bound);
    synthetic families with per-row outliers (x4, x16), heavy tails or a mix of gains and outliers do not - since round 4's refit on really
    trained weights the guard is conservative for them - and are routed to bf16x3 (at 512: the K-split kernel of csrc/kernels_ks.hip, round 6)."""
    AUTO, I8, BF3 = 3, 2, 0
    for kind, want in [("init", I8), ("outlier4", BF3), ("laplace", BF3), ("outlier16", BF3), ("trained", BF3)]:
        sd = orc.init_weights(W, 4, 0) if kind == "init" else orc.stress_weights(W, 4, 0, kind)
        m = _host_model(lib, W, sd, AUTO)
        e = _estimate(lib, m)
        assert e.budget == pytest.approx(1e-4) and e.acc_bound < 2 ** 31
        assert bool(e.ok) == (want == I8), (kind, e.rgb_pred)
        if kind == "init":
            assert 5e-5 < e.rgb_pred < 9.5e-5 and max(e.head_rms) == pytest.approx(e.worst)
        r = lib.snerf_model_resolve_precision(m)
        assert r == want and lib.snerf_model_precision(m) == want
        ns = C.c_size_t()
        assert lib.snerf_model_pack_host(m, 2, None, C.byref(ns), None, None) == (0 if want == I8 else -4)
        lib.snerf_model_destroy(m)
    m = _host_model(lib, W, orc.init_weights(W, 4, 0), 1)          # the one-term fast mode: widths 64 / 256 only
    assert lib.snerf_model_resolve_precision(m) == (1 if W != 512 else -1)
    lib.snerf_model_destroy(m)


@pytest.mark.parametrize("W", [64, 256, 512])
def test_trained_fixtures_clear_the_bound(lib, golden_dir, W):
    """The reference's own training loop, 400-600 steps (tests/golden/trained_W*.npz): `auto` keeps these on the int8 pipe, and the prediction
    is above what the GPU measures against the reference for them (3.5-3.7e-5, tests/test_gpu_stress.py)."""
    import os
    g = dict(np.load(os.path.join(golden_dir, f"trained_W{W}.npz"), allow_pickle=False))
    sd = {k[3:]: torch.tensor(v) for k, v in g.items() if k.startswith("sd_")}
    m = _host_model(lib, W, sd, 3)
    e = _estimate(lib, m)
    assert e.ok and 3.5e-5 < e.rgb_pred < 1e-4, e.rgb_pred
    assert lib.snerf_model_resolve_precision(m) == 2
    lib.snerf_model_destroy(m)


# forced int8 digits against the REFERENCE on the density-gain ladder (trained fixture, density head x g), measured on the GPU: tools/sharp_modes.py ->
# profiles/r5/sharp_modes_before_refit.txt, column "i8x3 vs ref" (worst of RGB / albedo / depth over 128 rays)
LADDER_OBSERVED = {64: {1: 6.25e-5, 2: 8.15e-5, 4: 9.66e-5, 8: 9.97e-5, 16: 1.58e-4, 32: 1.87e-4, 64: 3.02e-4, 128: 5.55e-4, 256: 1.10e-3},
                   256: {1: 3.83e-5, 2: 5.69e-5, 4: 1.06e-4, 8: 1.04e-4, 16: 1.21e-4, 32: 1.54e-4, 64: 2.01e-4, 128: 3.24e-4, 256: 6.25e-4},
                   512: {1: 3.17e-5, 2: 6.95e-5, 4: 8.91e-5, 8: 1.07e-4, 16: 1.21e-4, 32: 1.61e-4, 64: 2.14e-4, 128: 4.52e-4, 256: 7.15e-4}}


@pytest.mark.parametrize("W", [64, 256, 512])
def test_error_model_covers_the_gain_ladder(lib, golden_dir, W):
    """the pack-time error model against weights WITH SURFACES - the trained fixtures with the density head scaled by g = 1 ... 256 (mean
    max-PS per ray 0.03 ... 0.87).  The scale of round 5 (x1.5 over round 4's weights) was chosen on W = 64 / 256; W = 512 is the held-out check.  Asserted for every rung: the
    prediction is not below what the GPU measured in forced int8 digits, and `auto` never sends a set to int8 digits that was measured outside the bar."""
    import os
    g = dict(np.load(os.path.join(golden_dir, f"trained_W{W}.npz"), allow_pickle=False))
    head = ("G_NeRF_net.fc10Sigma.weight", "G_NeRF_net.fc10Sigma.bias")
    on_i8 = []
    for gain, seen in LADDER_OBSERVED[W].items():
        sd = {k[3:]: torch.tensor(v) * (float(gain) if k[3:] in head else 1.0) for k, v in g.items() if k.startswith("sd_")}
        m = _host_model(lib, W, sd, 3)
        e = _estimate(lib, m)
        lib.snerf_model_destroy(m)
        print(f"  W={W} g={gain:3d}: predicted {e.rgb_pred:.2e}, observed {seen:.2e} ({e.rgb_pred / seen:.2f}x) -> {'int8 digits' if e.ok else 'rejected'}")
        assert e.rgb_pred >= 1.05 * seen, (W, gain, e.rgb_pred, seen)
        assert not (e.ok and seen > 1e-4), (W, gain)
        if e.ok:
            on_i8.append(gain)
    assert on_i8 and max(on_i8) <= 2               # what stays on the int8 pipe: fog and near-fog only


def test_int8_accumulator_bound_is_enforced(lib):
    """(M << 8) + X is formed in int32: a row whose digits could carry it past 2^31 (K = 572 digit slots at W = 512, every weight
    at the row maximum) is refused under an explicit SNERF_PREC_I8X3 and not chosen by SNERF_PREC_AUTO."""
    W = 512
    sd = orc.init_weights(W, 4, 1)
    sd["G_NeRF_net.fc5.linear.weight"] = torch.full_like(sd["G_NeRF_net.fc5.linear.weight"], 0.004)
    m = _host_model(lib, W, sd, 2)
    e = _estimate(lib, m)
    assert e.acc_bound >= 2 ** 31 and not e.ok
    assert lib.snerf_model_resolve_precision(m) == -1 and b"int32" in lib.snerf_last_error()
    lib.snerf_model_destroy(m)
    m = _host_model(lib, 256, {k: (v[:256, :319] if k == "G_NeRF_net.fc5.linear.weight" else v) for k, v in
                               {**orc.init_weights(256, 4, 1), "G_NeRF_net.fc5.linear.weight": torch.full((256, 319), 0.004)}.items()}, 2)
    assert _estimate(lib, m).acc_bound < 2 ** 31          # 316 digit slots: cannot wrap whatever the weights
    lib.snerf_model_destroy(m)


def test_network_class_resolves_its_precision_on_the_host():
    import season_nerf_amd as sn
    net = sn.T_NeRF(256, 4)
    net.load_state_dict(orc.init_weights(256, 4, 0))
    assert net.precision == "auto" and net.resolved_precision == "i8x3" and net.fused
    est = net.i8_estimate()
    assert est["ok"] and est["rgb_pred"] < est["budget"]
    net.load_state_dict(orc.stress_weights(256, 4, 0, "outlier16"))
    assert net.resolved_precision == "bf16x3" and not net.i8_estimate()["ok"]
    net512 = sn.T_NeRF(512, 4)                              # the reference's default width (main_lite.py:80)
    net512.load_state_dict(orc.init_weights(512, 4, 0))
    assert net512.resolved_precision == "i8x3"
    net512.load_state_dict(orc.stress_weights(512, 4, 0, "outlier16"))
    assert net512.resolved_precision == "bf16x3" and net512.fused       # the K-split kernel (round 6; the layer-wise engine before)
    net512.precision = "bf16x3"
    assert net512.fused
    net512.precision = "bf16"
    assert not net512.fused                                             # the fast mode has no kernel at 512: layer-wise engine
    # a buffer replaced by Module._apply (.double().float()) is still tracked
    net.load_state_dict(orc.init_weights(256, 4, 0))
    assert net.resolved_precision == "i8x3"
    net.double().float()
    with torch.no_grad():
        net.G_NeRF_net.fc2.norm.running_var.mul_(1e-6)       # BatchNorm gain x1000: far outside the bound
    assert net.resolved_precision == "bf16x3"


def test_threaded_pack_is_the_serial_pack():
    """pack.cpp packs the layers of a program on a few host threads (disjoint regions of the stream and the tables) and runs the per-weight pass of the int8
    error model in parallel: both programs' streams and tables, the int8 stream and the estimate must be what SNERF_PACK_THREADS=1 produces (the switch is
    read once per process: two children).  A missing tensor is still reported from inside a worker."""
    import subprocess
    import sys
    code = r'''
import ctypes as C, hashlib, sys, numpy as np
sys.path.insert(0, %r)
from season_nerf_amd import _lib
from oracle import season_nerf_oracle as orc
lib = _lib.lib()
for W in (64, 256):
    sd = orc.init_weights(W, 4, 3)
    m = lib.snerf_model_create(W, 4)
    keep = {}
    for k, v in sd.items():
        if v.is_floating_point():
            keep[k] = np.ascontiguousarray(v.numpy(), dtype=np.float32).ravel()
            assert lib.snerf_model_set_tensor(m, k.encode(), keep[k].ctypes.data, keep[k].size) == 0
    assert lib.snerf_model_set_precision(m, 2) == 0
    h = hashlib.sha256()
    for prog in (0, 1, 2):
        ns, nb = C.c_size_t(), C.c_size_t()
        assert lib.snerf_model_pack_host(m, prog, None, C.byref(ns), None, C.byref(nb)) == 0
        s, b = np.zeros(ns.value, np.uint8), np.zeros(nb.value, np.float32)
        assert lib.snerf_model_pack_host(m, prog, s.ctypes.data, C.byref(ns), b.ctypes.data, C.byref(nb)) == 0
        h.update(s.tobytes()); h.update(b.tobytes())
    e = _lib.I8Estimate()
    assert lib.snerf_model_i8_estimate(m, C.byref(e)) == 0
    print(W, h.hexdigest(), "%%.9e %%.9e %%d" %% (e.rgb_pred, e.hidden_rms, e.acc_bound))
m = lib.snerf_model_create(64, 4)
n = C.c_size_t()
print("missing", lib.snerf_model_pack_host(m, 0, None, C.byref(n), None, None), b"missing tensor" in lib.snerf_last_error())
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),)
    outs = []
    for threads in ("1", "8"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SNERF_PACK_THREADS=threads), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout)
    # (the estimate's pass over the inputs' own error runs in layer order in both; its per-weight pass sums per row, so not a digit differs)
    assert outs[0] == outs[1] and "missing -2 True" in outs[0] and len(outs[0].split()) >= 12, outs


def _ks_sequence(shape, a):
    """Independent restatement of the order wave `a` of a pair consumes the (block, k-step) pairs of a layer in (csrc/kernels_ks.hip header):
    shape = (n_blocks, ks_hidden, ks_encoding, raw_head, encoding_only)."""
    nb, ksh_full, ksx, raw, enc_only = shape
    if raw:
        return [(0, a * (ksh_full // 2) + s) for s in range(ksh_full // 2)]
    nbh = nb // 2
    if enc_only:
        return [(a * nbh + i, s) for i in range(nbh) for s in range(ksx)]
    ksh, seq = ksh_full // 2, []
    for i in range(nbh):
        seq += [((1 - a) * nbh + i, a * ksh + s) for s in range(ksh)]                       # F: the partner's block over the own K-half
        seq += [(a * nbh + i, a * ksh + s) for s in range(ksh)] + [(a * nbh + i, ksh_full + s) for s in range(ksx)]   # O: own block, own K-half, then the encoding
    return seq


def test_ksplit_stream_is_a_permutation_of_the_bf16_pairs(lib):
    """Width 512, bf16x3 (round 6): program 3 (the stream the K-split kernel reads) holds exactly the pairs of the canonical bf16 field program
    (program 0, itself pinned by test_packed_stream_reproduces_the_network at the other widths), each once, in the order the two waves of a pair
    consume them - 4 pairs of parity 0, then 4 of parity 1 per 16 KiB chunk, every layer on a chunk boundary; same bias table."""
    W, Cn = 512, 4
    m = _host_model(lib, W, orc.init_weights(W, Cn, 5), 0)
    def fetch(prog):
        ns, nb = C.c_size_t(), C.c_size_t()
        assert lib.snerf_model_pack_host(m, prog, None, C.byref(ns), None, C.byref(nb)) == 0, lib.snerf_last_error()
        st, bi = np.zeros(ns.value, np.uint8), np.zeros(nb.value, np.float32)
        assert lib.snerf_model_pack_host(m, prog, st.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(ns), bi.ctypes.data_as(C.POINTER(C.c_float)), C.byref(nb)) == 0
        return st, bi
    canon, bias0 = fetch(0)
    ks, bias3 = fetch(3)
    assert np.array_equal(bias0, bias3) and len(ks) % 16384 == 0
    W2, PP, PS = W // 2, 4, 2
    # (n_blocks, hidden k-steps, encoding k-steps, raw head, encoding only) per field layer, program.h field_layer
    shapes = [(W // 32, 0, PP, False, True)] + [(W // 32, W // 16, 0, False, False)] * 3 + [(W // 32, W // 16, PP, False, False)] + [(W // 32, W // 16, 0, False, False)] * 3 + \
             [(W2 // 32, W // 16, 0, False, False), (1, W2 // 16, 0, True, False), (W2 // 32, W2 // 16, PS, False, False)] + [(W2 // 32, W2 // 16, 0, False, False)] * 2 + \
             [(1, W2 // 16, 0, True, False), (W // 32, W2 // 16, 0, False, False)] + [(W // 32, W // 16, 0, False, False)] * 2 + [(1, W // 16, 0, True, False)]
    co = ko = 0
    seen = 0
    for sh in shapes:
        nb, ksh, ksx, raw, enc = sh
        kst = ksx if enc else ksh + ksx
        n_pairs = nb * kst
        seq = [_ks_sequence(sh, a) for a in (0, 1)]
        assert len(seq[0]) == len(seq[1]) and sorted(seq[0] + seq[1]) == [(b, k) for b in range(nb) for k in range(kst)]      # every pair exactly once
        chunks = -(-len(seq[0]) // 4)
        used = np.zeros(chunks * 16384, bool)
        for a in (0, 1):
            for q, (b, k) in enumerate(seq[a]):
                src = co + (b * kst + k) * 2048
                dst = ko + (q // 4) * 16384 + a * 8192 + (q % 4) * 2048
                assert np.array_equal(ks[dst:dst + 2048], canon[src:src + 2048]), (sh, a, q)
                used[dst - ko:dst - ko + 2048] = True
                seen += 1
        assert not ks[ko:ko + chunks * 16384][~used].any()                                      # padding is zero
        co += -(-n_pairs // 8) * 16384
        ko += chunks * 16384
    assert co == len(canon) and ko == len(ks) and seen * 2048 <= len(ks)
    # the per-ray networks (program 1 canonical, program 4 K-split): the same pairs, each once (T1 / K1 read an encoding only: own blocks, two k-steps)
    g0, gb0 = fetch(1)
    g4, gb4 = fetch(4)
    assert np.array_equal(gb0, gb4)
    pairs = lambda st: sorted(bytes(st[o:o + 2048]) for o in range(0, len(st), 2048) if st[o:o + 2048].any())
    assert pairs(g0) == pairs(g4) and len(pairs(g4)) > 280
    lib.snerf_model_destroy(m)

