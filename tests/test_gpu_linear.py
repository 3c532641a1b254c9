"""The three Linear-layer products of the training engine (snerf_linear_forward / dgrad / wgrad: the bf16x3 row-owner and
weight-gradient kernels and the exact-fp32 MFMA kernel) against float64 torch matmuls at awkward shapes: row counts that are
not tile multiples, K / N that are not multiples of 16 / 32, unaligned leading dimensions, the 64-column-group variant
(K > 256), several column groups per row tile, partial 256 x 256 blocks of dW."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = {1: 3e-5, 0: 2e-6}          # max |err| / max |ref|: bf16x3 (3 x 2^-17-ish per product, fp32 accumulate) / exact fp32


def _env():
    import season_nerf_amd as sn
    L = sn._lib.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    return sn, L, st


def _scratch(L, n_out, n_in):
    return torch.empty(L.snerf_linear_scratch_bytes(n_out, n_in), dtype=torch.uint8, device="cuda")


def _rel(got, ref):
    return float((got.double() - ref).abs().max() / ref.abs().max())


FWD = [  # M, K (n_in), N (n_out), lda, ldc
    (1000, 64, 64, 64, 64), (1536, 319, 256, 320, 256), (777, 63, 256, 64, 260), (2048, 256, 256, 256, 256), (513, 128, 12, 128, 12),
    (130, 283, 128, 284, 128), (4101, 256, 319, 256, 320), (600, 63, 100, 63, 101), (1100, 512, 512, 512, 512), (700, 528, 40, 528, 40),
    (1, 16, 16, 16, 16), (65, 7, 5, 7, 5),
    # gemm_areg_kernel (csrc/gemm_areg.hip: accumulators in AGPRs, A and weights streamed; N = 256 / 512, K in 16-k steps, their number a multiple of 4):
    # whole and ragged row tiles, padded leading dimensions, 8 / 4 k-steps of A in flight (K / 16 a multiple of 8 or only of 4), more rows than one CU takes
    (1177, 512, 256, 516, 260), (3001, 320, 256, 320, 256), (2049, 576, 512, 576, 512), (640, 256, 512, 256, 512), (40000, 512, 512, 512, 512), (129, 128, 512, 132, 512),
    # the two-waves-per-SIMD form (N = 512): fewer rows than a wave pair shares, one row, exactly one tile
    (1, 512, 512, 512, 512), (17, 256, 512, 256, 516), (128, 512, 512, 512, 512)]


@pytest.mark.parametrize("precision", [1, 0])
@pytest.mark.parametrize("shape", FWD)
def test_linear_forward(shape, precision):
    sn, L, st = _env()
    M, K, N, lda, ldc = shape
    g = torch.Generator(device="cpu").manual_seed(M + K)
    A = torch.randn(M, lda, generator=g).cuda()
    Wt = (torch.randn(N, K, generator=g) / np.sqrt(K)).cuda()
    b = torch.randn(N, generator=g).cuda()
    out = torch.full((M, ldc), 7.0, device="cuda")
    use_stats = precision == 1 and K <= 512
    stats = torch.zeros(2, N, dtype=torch.float64, device="cuda")
    sc = _scratch(L, N, K)
    sn._lib.check(L.snerf_linear_forward(M, K, N, A.data_ptr(), lda, Wt.data_ptr(), b.data_ptr(), 30.0, out.data_ptr(), ldc,
                                         stats.data_ptr() if use_stats else None, precision, sc.data_ptr(), sc.numel(), None, 0, st), "linear_forward")
    ref = 30.0 * (A[:, :K].double() @ Wt.double().T + b.double())
    assert _rel(out[:, :N], ref) < TOL[precision]
    assert bool((out[:, N:] == 7.0).all())                       # padding columns of the output untouched
    if use_stats:
        d = ref - 30.0 * b.double()
        s = stats.cpu()
        np.testing.assert_allclose(s[0].numpy(), d.sum(0).cpu().numpy(), rtol=0, atol=2e-4 * float(d.abs().sum(0).max()))
        np.testing.assert_allclose(s[1].numpy(), (d * d).sum(0).cpu().numpy(), rtol=2e-4, atol=2e-4 * float((d * d).sum(0).max()))


DG = [  # M, n_in, n_out, n_cols, ld_go, ld_gi
    (1500, 319, 256, 256, 256, 256), (1500, 319, 256, 319, 256, 320), (900, 128, 128, 128, 128, 128), (2049, 256, 12, 256, 12, 256),
    (640, 283, 128, 283, 132, 283), (1030, 63, 256, 63, 256, 64)]


@pytest.mark.parametrize("precision", [1, 0])
@pytest.mark.parametrize("shape", DG)
@pytest.mark.parametrize("accumulate", [0, 1])
def test_linear_dgrad(shape, precision, accumulate):
    sn, L, st = _env()
    M, n_in, n_out, n_cols, ld_go, ld_gi = shape
    g = torch.Generator(device="cpu").manual_seed(M + n_in + accumulate)
    dZ = torch.randn(M, ld_go, generator=g).cuda()
    Wt = (torch.randn(n_out, n_in, generator=g) / np.sqrt(n_out)).cuda()
    base = torch.randn(M, ld_gi, generator=g).cuda()
    dIn = base.clone()
    sc = _scratch(L, n_out, n_in)
    sn._lib.check(L.snerf_linear_dgrad(M, n_in, n_out, dZ.data_ptr(), ld_go, Wt.data_ptr(), n_cols, 30.0, accumulate, dIn.data_ptr(), ld_gi,
                                       precision, sc.data_ptr(), sc.numel(), None, 0, None, None, None, None, st), "linear_dgrad")
    ref = 30.0 * (dZ[:, :n_out].double() @ Wt.double()[:, :n_cols]) + (base[:, :n_cols].double() if accumulate else 0.0)
    assert _rel(dIn[:, :n_cols], ref) < TOL[precision]
    assert torch.equal(dIn[:, n_cols:], base[:, n_cols:])


WG = [  # M, n_in, n_out, ld_go, ld_in
    (5000, 256, 256, 256, 256), (3001, 319, 256, 256, 320), (1024, 283, 128, 128, 284), (100, 63, 64, 64, 64), (1234, 512, 512, 512, 512),
    (257, 128, 12, 12, 128), (33, 16, 16, 20, 17)]


@pytest.mark.parametrize("precision", [1, 0])
@pytest.mark.parametrize("shape", WG)
def test_linear_wgrad(shape, precision):
    sn, L, st = _env()
    M, n_in, n_out, ld_go, ld_in = shape
    g = torch.Generator(device="cpu").manual_seed(M + n_in)
    dZ = torch.randn(M, ld_go, generator=g).cuda()
    X = torch.randn(M, ld_in, generator=g).cuda()
    base = torch.randn(n_out, n_in, generator=g).cuda()
    dW = base.clone()
    sn._lib.check(L.snerf_linear_wgrad(M, n_in, n_out, dZ.data_ptr(), ld_go, X.data_ptr(), ld_in, 0.5, dW.data_ptr(), precision, None, 0, st), "linear_wgrad")
    ref = base.double() + 0.5 * (dZ[:, :n_out].double().T @ X[:, :n_in].double())
    scale = float((0.5 * (dZ[:, :n_out].double().T @ X[:, :n_in].double())).abs().max())
    assert float((dW.double() - ref).abs().max()) / scale < TOL[precision] * 3        # split-K fp32 atomics on top


THIN = [  # M, n_in, n_out, ld_go, ld_in, offset of the gradient's first column, act_cols (0: none)   - the heads of the network: K <= 4 outputs, >= 1024 rows
    (5000, 128, 3, 4, 128, 0, 0), (5000, 128, 1, 4, 128, 3, 0), (4096, 64, 3, 3, 64, 0, 0), (1024, 256, 4, 4, 256, 0, 0), (393216, 128, 1, 1, 128, 0, 0),
    (3001, 128, 3, 4, 132, 0, 128), (2050, 128, 1, 4, 128, 3, 96), (1025, 12, 2, 2, 12, 0, 0)]


@pytest.mark.parametrize("shape", THIN)
def test_thin_head_wgrad(shape):
    """The weight gradient of a head with at most four outputs runs as a stream over its input (thin_wgrad_kernel: exact fp32 FMAs, one atomic add per
    block and element), with the gradient read as 16-byte rows where its address allows (the colour head) and as scalars where not (the density head:
    column 3 of the same [N, 4] buffer), and with the activation applied on load over the leading columns."""
    sn, L, st = _env()
    M, n_in, n_out, ld_go, ld_in, off, ac = shape
    g = torch.Generator(device="cpu").manual_seed(M + n_in + n_out)
    dZ = torch.randn(M, ld_go, generator=g).cuda()
    X = (torch.randn(M, ld_in, generator=g) * (4 if ac else 1)).cuda()
    base = torch.randn(n_out, n_in, generator=g).cuda()
    dW = base.clone()
    Hd = X[:, :n_in].double().clone()
    tab = None
    if ac:
        a_, b_ = (torch.rand(ac, generator=g) * 0.2 + 0.05).double(), torch.randn(ac, generator=g).double()
        tab = torch.stack([a_, b_]).float().cuda().contiguous()
        Hd[:, :ac] = torch.sin(2 * np.pi * (tab[0].double() * Hd[:, :ac] + tab[1].double()))
    dz = dZ.reshape(-1)[off:]
    sn._lib.check(L.snerf_linear_wgrad(M, n_in, n_out, dz.data_ptr(), ld_go, X.data_ptr(), ld_in, 0.5, dW.data_ptr(), 1, tab.data_ptr() if ac else None, ac, st), "linear_wgrad")
    cols = torch.stack([dZ.reshape(-1)[off + k::ld_go][:M] for k in range(n_out)], 1).double()      # (the last row of an offset view is short by `off`: M rows exist for off + k < ld_go)
    prod = 0.5 * (cols.T @ Hd)
    assert float((dW.double() - (base.double() + prod)).abs().max()) / float(prod.abs().max()) < 2e-5


@pytest.mark.parametrize("shape", [(5000, 128, 3, 128, 4, 0), (5000, 128, 1, 132, 1, 0), (1024, 32, 4, 32, 4, 0), (393216, 128, 4, 128, 4, 128), (2050, 256, 2, 256, 5, 96), (1025, 12, 3, 12, 3, 0)])
def test_thin_head_forward(shape):
    """Heads with at most four outputs run as a stream over their input (thin_fwd_kernel): exact fp32 FMAs, the lanes of a row reduced by shuffles, with and
    without the activation on load; against float64."""
    sn, L, st = _env()
    M, n_in, n_out, ld_in, ld_out, ac = shape
    g = torch.Generator(device="cpu").manual_seed(M + n_in + n_out)
    X = (torch.randn(M, ld_in, generator=g) * (4 if ac else 1)).cuda()
    Wt = (torch.randn(n_out, n_in, generator=g) / np.sqrt(n_in)).cuda()
    b = torch.randn(n_out, generator=g).cuda()
    Hd = X[:, :n_in].double().clone()
    tab = None
    if ac:
        a_, b_ = (torch.rand(ac, generator=g) * 0.2 + 0.05).double(), torch.randn(ac, generator=g).double()
        tab = torch.stack([a_, b_]).float().cuda().contiguous()
        Hd[:, :ac] = torch.sin(2 * np.pi * (tab[0].double() * Hd[:, :ac] + tab[1].double()))
    out = torch.full((M, ld_out), 7.0, device="cuda")
    sc = _scratch(L, n_out, n_in)
    sn._lib.check(L.snerf_linear_forward(M, n_in, n_out, X.data_ptr(), ld_in, Wt.data_ptr(), b.data_ptr(), 1.5, out.data_ptr(), ld_out, None, 1,
                                         sc.data_ptr(), sc.numel(), tab.data_ptr() if ac else None, ac, st), "linear_forward")
    ref = 1.5 * (Hd @ Wt.double().T + b.double())
    err = float((out[:, :n_out].double() - ref).abs().max()) / float(ref.abs().max())
    assert err < (2e-5 if ac else 2e-6), err                    # (activation on load: the hardware sine's ~1e-6 argument error times the weights)
    assert bool((out[:, n_out:] == 7.0).all())                  # columns past n_out untouched


def test_linear_argument_errors():
    sn, L, st = _env()
    a = torch.zeros(8, 8, device="cuda")
    assert L.snerf_linear_forward(8, 8, 8, a.data_ptr(), 4, a.data_ptr(), None, 1.0, a.data_ptr(), 8, None, 0, None, 0, None, 0, st) != 0      # ld < n_in
    assert b"bad argument" in L.snerf_last_error()
    assert L.snerf_linear_forward(8, 8, 8, a.data_ptr(), 8, a.data_ptr(), None, 1.0, a.data_ptr(), 8, None, 1, None, 0, None, 0, st) != 0      # no scratch
    assert L.snerf_linear_forward(0, 8, 8, None, 8, None, None, 1.0, None, 8, None, 1, None, 0, None, 0, st) == 0                                  # empty batch
    assert L.snerf_linear_dgrad(8, 8, 8, a.data_ptr(), 8, a.data_ptr(), 9, 1.0, 0, a.data_ptr(), 8, 0, None, 0, None, 0, None, None, None, None, st) != 0                 # n_cols > n_in
    assert L.snerf_linear_wgrad(-1, 8, 8, a.data_ptr(), 8, a.data_ptr(), 8, 1.0, a.data_ptr(), 0, None, 0, st) != 0


ACT = [  # M, K, N, lda, act_cols   (In5-like concat input: transformed leading columns + raw tail; thin head; K = 128)
    (1300, 319, 256, 320, 256), (2048, 256, 256, 256, 256), (777, 128, 3, 128, 128), (1025, 156, 128, 156, 128), (64, 16, 16, 16, 8),
    # gemm_areg_kernel: table over all of K, and over the leading 512 of 576 columns (the [h | PE] concat input of fc5 at the reference's default width)
    (1300, 512, 512, 512, 512), (3333, 576, 512, 576, 512), (900, 320, 256, 320, 256), (700, 256, 512, 260, 256), (15, 512, 512, 512, 512), (1, 256, 512, 256, 256)]


@pytest.mark.parametrize("shape", ACT)
def test_activation_on_load(shape):
    """forward and wgrad with the activation applied while the operand is loaded: in = [sin(gamma*((z-mu)*istd)+beta) | raw],
    through the folded table a = gamma*istd/2pi, b = (beta - gamma*mu*istd)/2pi and the hardware sine (revolutions)."""
    sn, L, st = _env()
    M, K, N, lda, ac = shape
    g = torch.Generator(device="cpu").manual_seed(M + K)
    Zp = (torch.randn(M, lda, generator=g) * 4).cuda()                       # SIREN-sized pre-activations
    mu, istd = torch.randn(ac, generator=g).double(), (torch.rand(ac, generator=g) + 0.5).double()
    gam, bet = (torch.rand(ac, generator=g) + 0.5).double(), torch.randn(ac, generator=g).double()
    tab = torch.stack([gam * istd / (2 * np.pi), (bet - gam * mu * istd) / (2 * np.pi)]).float().cuda().contiguous()
    Wt = (torch.randn(N, K, generator=g) / np.sqrt(K)).cuda()
    b = torch.randn(N, generator=g).cuda()
    zd = Zp[:, :K].double()
    Hd = zd.clone()
    Hd[:, :ac] = torch.sin(gam.cuda() * ((zd[:, :ac] - mu.cuda()) * istd.cuda()) + bet.cuda())
    out = torch.zeros(M, N, device="cuda")
    sc = _scratch(L, N, K)
    sn._lib.check(L.snerf_linear_forward(M, K, N, Zp.data_ptr(), lda, Wt.data_ptr(), b.data_ptr(), 30.0, out.data_ptr(), N, None, 1,
                                         sc.data_ptr(), sc.numel(), tab.data_ptr(), ac, st), "linear_forward(act)")
    ref = 30.0 * (Hd @ Wt.double().T + b.double())
    assert _rel(out, ref) < 2 * TOL[1]                             # + the argument error of the folded fp32 table (~|arg| * 1e-7)
    dZ = torch.randn(M, N, generator=g).cuda()
    dW = torch.zeros(N, K, device="cuda")
    sn._lib.check(L.snerf_linear_wgrad(M, K, N, dZ.data_ptr(), N, Zp.data_ptr(), lda, 1.0, dW.data_ptr(), 1, tab.data_ptr(), ac, st), "linear_wgrad(act)")
    refw = dZ.double().T @ Hd
    assert float((dW.double() - refw).abs().max() / refw.abs().max()) < 3 * TOL[1]
    # precision 0 cannot do it: loud error
    assert L.snerf_linear_forward(M, K, N, Zp.data_ptr(), lda, Wt.data_ptr(), b.data_ptr(), 30.0, out.data_ptr(), N, None, 0,
                                  sc.data_ptr(), sc.numel(), tab.data_ptr(), ac, st) != 0


@pytest.mark.parametrize("shape", [(1500, 256, 256, True), (1029, 128, 128, False), (2048, 319, 256, True), (700, 256, 12, False),
                                   # gemm_areg_kernel's activation-backward form (512-wide layers): whole and ragged row tiles, with and without BatchNorm
                                   (1500, 512, 512, True), (2049, 512, 512, False), (1031, 256, 512, True), (40000, 512, 512, True)])
def test_dgrad_activation_backward_epilogue(shape):
    """dgrad whose epilogue already applies the activation backward of the SineLayer below (x cos) and reduces the column sums."""
    sn, L, st = _env()
    M, n_in, n_out, bn = shape
    n_cols = 256 if n_in == 319 else n_in
    g = torch.Generator(device="cpu").manual_seed(M)
    dZ = torch.randn(M, n_out, generator=g).cuda()
    Wt = (torch.randn(n_out, n_in, generator=g) / np.sqrt(n_out)).cuda()
    Zb = (torch.randn(M, n_cols + 4, generator=g) * 4).cuda()                              # pre-activation of the layer below
    mu, istd = torch.randn(n_cols, generator=g).double(), (torch.rand(n_cols, generator=g) + 0.5).double()
    gam, bet = (torch.rand(n_cols, generator=g) + 0.5).double(), torch.randn(n_cols, generator=g).double()
    if not bn:
        mu, istd, gam, bet = torch.zeros(n_cols).double(), torch.ones(n_cols).double(), torch.ones(n_cols).double(), torch.zeros(n_cols).double()
    tab = torch.stack([gam * istd / (2 * np.pi), (bet - gam * mu * istd) / (2 * np.pi)]).float().cuda().contiguous()
    mu_f, is_f = mu.float().cuda(), istd.float().cuda()
    out = torch.zeros(M, n_cols, device="cuda")
    sums = torch.zeros(2, n_cols, dtype=torch.float64, device="cuda")
    sc = _scratch(L, n_out, n_in)
    sn._lib.check(L.snerf_linear_dgrad(M, n_in, n_out, dZ.data_ptr(), n_out, Wt.data_ptr(), n_cols, 30.0, 0, out.data_ptr(), n_cols, 1,
                                       sc.data_ptr(), sc.numel(), Zb.data_ptr(), n_cols + 4, tab.data_ptr(),
                                       mu_f.data_ptr() if bn else None, is_f.data_ptr() if bn else None, sums.data_ptr(), st), "dgrad(act)")
    zd = Zb[:, :n_cols].double()
    xh = (zd - mu.cuda()) * istd.cuda()
    ref = 30.0 * (dZ.double() @ Wt.double()[:, :n_cols]) * torch.cos(gam.cuda() * xh + bet.cuda())
    assert _rel(out, ref) < 2 * TOL[1]
    s_ = sums.cpu().numpy()
    np.testing.assert_allclose(s_[0], ref.sum(0).cpu().numpy(), rtol=0, atol=3e-4 * float(ref.abs().sum(0).max()))
    want1 = (ref * xh).sum(0).cpu().numpy() if bn else np.zeros(n_cols)
    np.testing.assert_allclose(s_[1], want1, rtol=0, atol=3e-4 * float((ref * xh).abs().sum(0).max()) + 1e-12)
    # with accumulation (the last of several producers of dL/dH): (previous + this product) * cos
    prev = torch.randn(M, n_cols, generator=g).cuda()
    out2 = prev.clone()
    sums.zero_()
    sn._lib.check(L.snerf_linear_dgrad(M, n_in, n_out, dZ.data_ptr(), n_out, Wt.data_ptr(), n_cols, 30.0, 1, out2.data_ptr(), n_cols, 1,
                                       sc.data_ptr(), sc.numel(), Zb.data_ptr(), n_cols + 4, tab.data_ptr(),
                                       mu_f.data_ptr() if bn else None, is_f.data_ptr() if bn else None, sums.data_ptr(), st), "dgrad(act, accumulate)")
    ref2 = (30.0 * (dZ.double() @ Wt.double()[:, :n_cols]) + prev.double()) * torch.cos(gam.cuda() * xh + bet.cuda())
    assert _rel(out2, ref2) < 2 * TOL[1]
    np.testing.assert_allclose(sums.cpu().numpy()[0], ref2.sum(0).cpu().numpy(), rtol=0, atol=3e-4 * float(ref2.abs().sum(0).max()))


def test_full_tile_kernel_bit_identical_to_general_kernel():
    """(Also: gemm_rows16_kernel, the 16x16x32 form the wide layers run on, against the same outputs to 4e-6 of the output scale -
    it sums the k terms of a 32-k step in another order - with identical untouched cells and column sums.)
    gemm_rows_full_kernel (hand-issued A stream, cross-tile prefetch, buffer-addressed epilogue) against gemm_rows_kernel on
    the same inputs: same fragments, same summation order -> identical bits in every output (forward with and without activation
    on load, thin heads, zero-padded K, dgrad with the activation-backward epilogue); column sums agree to fp32 partial-sum
    rounding.  The path is chosen per process (SNERF_GEMM_FULL), so the two runs are child processes."""
    import subprocess, sys, os
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "compare_gemm_paths.py")
    r = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "mismatches: 0" in r.stdout
    print(r.stdout[-400:])


@pytest.mark.parametrize("shape", [(32768 + 5077, 256, 256), (40000 + 311, 128, 256)])      # (launches below 32768 rows always walk forwards)
def test_row_tiles_in_both_orders(shape):
    """gemm_rows16_kernel walks its row tiles forwards on one launch and backwards on the next (launch_gemm_rows16: the rows the producer
    wrote last first).  Four consecutive launches on a ragged row count cover both orders: every one must give the same product, the same
    BatchNorm sums, and leave the rows past M alone."""
    sn, L, st = _env()
    M, K, N = shape
    g = torch.Generator(device="cpu").manual_seed(M)
    A = torch.randn(M, K, generator=g).cuda()
    Wt = (torch.randn(N, K, generator=g) / np.sqrt(K)).cuda()
    b = torch.randn(N, generator=g).cuda()
    sc = _scratch(L, N, K)
    ref = 30.0 * (A.double() @ Wt.double().T + b.double())
    d = ref - 30.0 * b.double()
    outs = []
    for _ in range(4):
        out = torch.full((M + 300, N), 7.0, device="cuda")
        stats = torch.zeros(2, N, dtype=torch.float64, device="cuda")
        sn._lib.check(L.snerf_linear_forward(M, K, N, A.data_ptr(), K, Wt.data_ptr(), b.data_ptr(), 30.0, out.data_ptr(), N,
                                             stats.data_ptr(), 1, sc.data_ptr(), sc.numel(), None, 0, st), "linear_forward")
        assert _rel(out[:M], ref) < TOL[1]
        assert bool((out[M:] == 7.0).all())
        np.testing.assert_allclose(stats[0].cpu().numpy(), d.sum(0).cpu().numpy(), rtol=0, atol=2e-4 * float(d.abs().sum(0).max()))
        outs.append(out[:M].clone())
    for o in outs[1:]:
        assert torch.equal(o, outs[0])          # a row's product does not depend on the order of the tiles
