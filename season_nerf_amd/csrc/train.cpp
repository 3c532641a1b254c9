// Layer-wise training engine: forward with batch-statistics BatchNorm (+ running-stat EMA), backward, Adam.
// Reference semantics: T_NeRF.forward / forward_Solar in .train() mode (T_NeRF_net_v2.py:75-157, misc.py:148-189),
// Eval_Tools_2.py:165-215 (compositing) and :297-337 (sun-ray pass, trunk under no_grad), mg_run_NeRF.py:288-326.
// The caller (PyTorch) owns every buffer: parameter / gradient / Adam arenas and the activation workspace.
#include "../../include/season_nerf_hip.h"

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <mutex>
#include <set>
#include <vector>

#include "kernels.h"
#include "train.h"

// Every device-to-device copy and zero-fill of the engine is a KERNEL of ours, never hipMemcpyAsync / hipMemsetAsync.  Inside a captured training step a
// runtime memory operation becomes a hipGraph MEMCPY / MEMSET node, and on ROCm 7.2 (graph nodes recorded as AQL packets, DEBUG_CLR_GRAPH_PACKET_CAPTURE
// default on) replays of a graph that holds such nodes computed garbage once eager work of the same process (a save point's validation, a second capture)
// had run between them: gradients of 1e5..1e36 in whole layers, Adam moments to match - the failure family rounds 4-6 chased.  With kernel
// nodes only, 0 of 68 driver runs differ from the eager run; with the memory operations back (SNERF_TRAIN_MEMOPS=1, kept for the reproduction:
// tools/graph_wait_probe4.py) 29 of 29 do; with the memory operations AND packet capture off, 0 of 6.  The defect is the runtime's and reproduces without any
// code of this repository: tools/graph_memset_min.py (25 lines), profiles/r6/graph_memop_repro.txt.
static bool train_memops() {
    static const bool on = [] { const char* e = getenv("SNERF_TRAIN_MEMOPS"); return e && e[0] == '1'; }();
    return on;
}
static hipError_t snerf_zero_async(void* p, int value, size_t bytes, hipStream_t st) {
    if (train_memops()) return hipMemsetAsync(p, value, bytes, st);
    if (value != 0) return hipErrorInvalidValue;                     // (the engine only ever zeroes)
    if (bytes % 4 != 0 || ((uintptr_t)p & 3)) return snerf::launch_fill_zero_bytes(p, bytes, st);
    return snerf::launch_fill_zero((float*)p, (int64_t)(bytes / 4), st);
}
static hipError_t snerf_copy_async(void* d, const void* s_, size_t bytes, hipStream_t st) {
    if (train_memops()) return hipMemcpyAsync(d, s_, bytes, hipMemcpyDeviceToDevice, st);
    if (bytes % 4 != 0 || (((uintptr_t)d | (uintptr_t)s_) & 3)) return snerf::launch_copy_bytes(d, s_, bytes, st);
    return snerf::launch_copy_f32((float*)d, (const float*)s_, (int64_t)(bytes / 4), st);
}

using namespace snerf;

extern int snerf_set_error(int code, const std::string& msg);   // api.cpp

namespace {

struct LayerP {
    std::string name;
    int n_out, n_in;
    bool sine, bn;
    int64_t w, b, g, beta;     // offsets (floats) in the parameter arena; g/beta only with bn
    int64_t rm, rv;            // offsets in the buffer arena (running mean/var); only with bn
};

enum L {
    L_FC1, L_FC2, L_FC3, L_FC4, L_FC5, L_FC6, L_FC7, L_FC8, L_FC9, L_COL, L_SIG, L_S1, L_S2, L_S3, L_S4, L_K1, L_K2,
    L_T1, L_T2, L_CL, L_A1, L_A2, L_A3, L_AC, L_DEAD0, L_DEAD1, L_DEAD2, L_NUM
};

struct Act {           // a [rows, cols] fp32 matrix inside the workspace
    float* p = nullptr;
    int64_t ld = 0;
    // activation on load: when tab != nullptr the first `cols` columns hold the PRE-activation of a SineLayer and every
    // consumer (row GEMM, wgrad) applies sin(2 pi (a z + b)) with tab = [a | b] while loading - the post-activation is not stored
    const float* tab = nullptr;
    int cols = 0;
    // the columns between the layer's n_in and the next multiple of 16 exist and hold zeros (row GEMM: no K tail to guard)
    bool padded = false;
};

}  // namespace

struct snerf_trainer {
    int W, C, W2, W4;
    std::vector<LayerP> layers;
    int64_t n_params = 0, n_buffers = 0;
    float *params = nullptr, *grads = nullptr, *adam_m = nullptr, *adam_v = nullptr, *buffers = nullptr;
    char* ws = nullptr;
    size_t ws_bytes = 0;
    // state of the last forward passes (needed by backward)
    int64_t R = 0, Rs = 0;
    int S = 0;
    int img_flags = 0;                              // flags of the last image-pass forward (bit 0: classic solar model)
    // workspace carve (set by carve())
    struct Pass {
        int64_t R = 0, N = 0;
        Act E, In5, Z[9], H[9], head, Za[3], Ha[3], adj, In_s1, Zs[3], Hs[3], sv_raw, tmpA, tmpB;
        float *pts, *rho, *col, *sv, *pe_sun, *pe_time, *Zt1, *Ht1, *Zt2, *Ht2, *logits, *cls, *Zk1, *Hk1, *sky_raw, *sky;
        float* bn;             // [8 layers][4][W]: colsum, m2, mean, istd
        float *top, *bot, *tvals;   // engine-owned copies: the caller's tensors may be recycled before backward
        // what the consumers of each per-point SineLayer read: the materialised H, or (activation on load) Z + its table
        Act Hc[9], Hac[3], Hsc[3], In5c, In_s1c, Zc3, Zc8;
        float* tabs = nullptr;      // [15 layers][2][W] activation-on-load tables of this pass
        bool aol = false;
    } img, sol;
    Act dA, dB, dX1;
    float *d_head, *d_adj, *d_rho, *d_col, *d_sky, *d_cls, *d_sv_raw, *rayA, *rayB, *bn_bwd;   // bn_bwd: [2][W]
    uint16_t* w_frag = nullptr;                     // scratch: one weight matrix split into bf16 hi/lo MFMA fragments
    double* bn_stats = nullptr;                     // [2][W] shifted column sums from the GEMM epilogue
    int gemm_mode = 1;                              // 1 = bf16x3 MFMA for forward/dgrad (default), 0 = exact fp32 MFMA everywhere
    int aol_mode = 1;                               // 1 = activation on load where possible (SNERF_TRAIN_AOL=0: always store H)
    // data-parallel BatchNorm over the global batch: sum-all-reduce of the statistics buffers (snerf_trainer_set_allreduce)
    snerf_allreduce_fn ar_fn = nullptr;
    void* ar_user = nullptr;
    int world = 1;
    float* wgrad_partial = nullptr;                 // partial dW blocks of the two-stage weight-gradient reduction (carved: no hipMalloc in a step)
    size_t wgrad_partial_floats = 0;
    unsigned launch_parity = 0;                     // direction of the next streaming launch; reset at the start of every pass
};

// Every pass runs under its trainer's launch context (gemm.hip): the row direction of a streaming launch - and with it the summation order of the
// weight gradients' fp32 stages - depends on the launch's position in the pass only, and the partial sums land in this trainer's workspace.
struct CtxGuard {
    explicit CtxGuard(snerf_trainer* t) { t->launch_parity = 0; gemm_launch_context(t->wgrad_partial, t->wgrad_partial_floats, &t->launch_parity); }
    ~CtxGuard() { gemm_launch_context(nullptr, 0, nullptr); }
};

// sum the buffer over all ranks (no-op without a registered collective)
static int all_reduce(snerf_trainer* t, void* buf, int64_t count, bool is_double, hipStream_t st) {
    if (!t->ar_fn) return SNERF_OK;
    if (t->ar_fn(t->ar_user, buf, count, is_double ? 1 : 0, (void*)st) != 0)
        return snerf_set_error(SNERF_E_STATE, "the registered all-reduce callback failed");
    return SNERF_OK;
}

static int64_t align64(int64_t x) { return (x + 63) / 64 * 64; }

static void build_layers(snerf_trainer* t) {
    const int W = t->W, C = t->C, W2 = t->W2, W4 = t->W4;
    auto add = [&](const char* name, int n_out, int n_in, bool sine, bool bn) {
        LayerP l;
        l.name = name; l.n_out = n_out; l.n_in = n_in; l.sine = sine; l.bn = bn;
        l.w = t->n_params; t->n_params += (int64_t)n_out * n_in;
        l.b = t->n_params; t->n_params += n_out;
        l.g = l.beta = l.rm = l.rv = -1;
        if (bn) {
            l.g = t->n_params; t->n_params += n_out;
            l.beta = t->n_params; t->n_params += n_out;
            l.rm = t->n_buffers; t->n_buffers += n_out;
            l.rv = t->n_buffers; t->n_buffers += n_out;
        }
        t->layers.push_back(l);
    };
    add("G_NeRF_net.fc1", W, 63, true, false);
    add("G_NeRF_net.fc2", W, W, true, true);
    add("G_NeRF_net.fc3", W, W, true, true);
    add("G_NeRF_net.fc4", W, W, true, true);
    add("G_NeRF_net.fc5", W, W + 63, true, true);
    add("G_NeRF_net.fc6", W, W, true, true);
    add("G_NeRF_net.fc7", W, W, true, true);
    add("G_NeRF_net.fc8", W, W, true, true);
    add("G_NeRF_net.fc9", W2, W, true, true);
    add("G_NeRF_net.fc10Col", 3, W2, false, false);
    add("G_NeRF_net.fc10Sigma", 1, W2, false, false);
    add("G_NeRF_net.fc_solar_1", W2, W2 + 27, true, false);
    add("G_NeRF_net.fc_solar_2", W2, W2, true, false);
    add("G_NeRF_net.fc_solar_3", W2, W2, true, false);
    add("G_NeRF_net.fc_solar_4", 1, W2, false, false);
    add("G_NeRF_net.fc_sky_color_1", W4, 27, true, false);
    add("G_NeRF_net.fc_sky_color_2", 3, W4, false, false);
    add("time_layer_1", W, 10, true, false);
    add("time_layer_2", W, W, true, false);
    add("get_class_layer", C, W, false, false);
    add("adjust_layer_1", W, W2, true, false);
    add("adjust_layer_2", W, W, true, false);
    add("adjust_layer_3", W, W, true, false);
    add("adjust_col", 3 * C, W, false, false);
    add("adjust_rho", C, W, false, false);            // dead heads: parameters only (never used, never get a gradient)
    add("adjust_solar_vis", C, W, false, false);
    add("adjust_sky_col", 3 * C, W, false, false);
}

// ---- workspace carving ------------------------------------------------------------------------------
struct Carver {
    char* base;
    size_t off = 0;
    float* take(int64_t floats) {
        float* p = base ? (float*)(base + off) : nullptr;
        off += (size_t)align64(floats) * 4;
        return p;
    }
    Act mat(int64_t rows, int64_t ld) { Act a; a.p = take(rows * ld); a.ld = ld; return a; }
};

static size_t carve(snerf_trainer* t, char* base, int64_t R, int64_t Rs, int S) {
    Carver c{base};
    const int W = t->W, W2 = t->W2, W4 = t->W4, C = t->C;
    for (int pass = 0; pass < 2; ++pass) {
        snerf_trainer::Pass& P = pass == 0 ? t->img : t->sol;
        const int64_t Rr = pass == 0 ? R : Rs, N = Rr * S;
        P.R = Rr; P.N = N;
        P.E = c.mat(N, 64);
        P.In5 = c.mat(N, W + 64);
        for (int l = 0; l < 9; ++l) {
            const int w = l == 8 ? W2 : W;
            P.Z[l] = c.mat(N, w);
            if (l == 3) { P.H[l].p = P.In5.p; P.H[l].ld = W + 64; }     // fc4's output lives inside fc5's concat input
            else P.H[l] = c.mat(N, w);
        }
        P.head = c.mat(N, 4);
        for (int l = 0; l < 3; ++l) { P.Za[l] = c.mat(N, W); P.Ha[l] = c.mat(N, W); }
        P.adj = c.mat(N, 3 * C);
        P.In_s1 = c.mat(N, W2 + 32);         // [fc9 | PE(sun) 27 + 1 | 4 zero columns]: 16-column granules for the row GEMM
        for (int l = 0; l < 3; ++l) { P.Zs[l] = c.mat(N, W2); P.Hs[l] = c.mat(N, W2); }
        P.sv_raw = c.mat(N, 1);
        P.pts = c.take(N * 3); P.rho = c.take(N); P.col = c.take(N * 3); P.sv = c.take(N);
        P.pe_sun = c.take(Rr * 28); P.pe_time = c.take(Rr * 12);
        P.Zt1 = c.take(Rr * W); P.Ht1 = c.take(Rr * W); P.Zt2 = c.take(Rr * W); P.Ht2 = c.take(Rr * W);
        P.logits = c.take(Rr * C); P.cls = c.take(Rr * C);
        P.Zk1 = c.take(Rr * W4); P.Hk1 = c.take(Rr * W4); P.sky_raw = c.take(Rr * 3); P.sky = c.take(Rr * 3);
        P.bn = c.take(8 * 4 * W);
        P.top = c.take(Rr * 3); P.bot = c.take(Rr * 3); P.tvals = c.take(S);
        P.tabs = c.take(15 * 2 * W);
    }
    const int64_t Nmax = (R > Rs ? R : Rs) * S, Rmax = R > Rs ? R : Rs;
    t->dA = c.mat(Nmax, W + 64);
    t->dB = c.mat(Nmax, W + 64);
    t->dX1 = c.mat(Nmax, W2);
    t->d_head = c.take(Nmax * 4); t->d_adj = c.take(Nmax * 3 * C); t->d_rho = c.take(Nmax); t->d_col = c.take(Nmax * 3);
    t->d_sky = c.take(Rmax * 3); t->d_cls = c.take(Rmax * C); t->d_sv_raw = c.take(Nmax);
    t->rayA = c.take(Rmax * W); t->rayB = c.take(Rmax * W);
    t->bn_bwd = c.take(2 * W);
    const int64_t frag_bytes = (int64_t)((W + 64 + 31) / 32) * ((W + 64 + 15) / 16) * 2048;
    t->w_frag = (uint16_t*)c.take(frag_bytes / 4);
    t->bn_stats = (double*)c.take(4 * W + 2);
    t->wgrad_partial_floats = gemm_wgrad_partial_floats();
    t->wgrad_partial = c.take((int64_t)t->wgrad_partial_floats);
    return c.off;
}

// ---- building blocks ----------------------------------------------------------------------------------
#define HIPCK(x)                                                                                   \
    do {                                                                                           \
        hipError_t _e = (x);                                                                       \
        if (_e != hipSuccess) return snerf_set_error(SNERF_E_HIP, std::string(#x) + ": " + hipGetErrorString(_e)); \
    } while (0)

// bf16x3 row-owner kernel usable for an [M x K] x [K x N] product?
static bool rows_ok(const snerf_trainer* t, int64_t M, int K, int N) {
    // thin layers included (heads with 1..12 outputs): the kernels mask partial tiles, and every such GEMM is a single pass
    // over a [points x W] activation - HBM-bound either way
    return t->gemm_mode == 1 && M >= 1024 && K >= 1 && N >= 1 && gemm_rows_group_tiles((K + 15) / 16) > 0;
}
#define RCI(x)                 \
    do {                      \
        int _rci = (x);       \
        if (_rci) return _rci; \
    } while (0)

// Z[M, n_out] = alpha * (In[M, K] W^T + b);  colsum (fp32 path) / stats (bf16x3 path) optional
static hipError_t linear_fwd(snerf_trainer* t, const LayerP& L, Act InA, int64_t M, float* Z, int64_t ldz,
                             float alpha, float* colsum, hipStream_t st, double* stats = nullptr) {
    const float* In = InA.p;
    const int64_t ld_in = InA.ld;
    if (InA.tab && !rows_ok(t, M, L.n_in, L.n_out)) return hipErrorInvalidValue;        // activation on load needs the row kernel
    if (rows_ok(t, M, L.n_in, L.n_out)) {
        GemmX x{};
        x.n_tiles = (L.n_out + 31) / 32; x.ksteps = (L.n_in + 15) / 16;
        x.W = t->params + L.w; x.w_rows = L.n_out; x.w_cols = L.n_in; x.w_transpose = 0;      // split by the launcher, in its kernel's fragment order
        x.A = In; x.frag = t->w_frag; x.C = Z; x.M = M; x.N = L.n_out; x.K = L.n_in; x.lda = ld_in; x.ldc = ldz;
        x.alpha = alpha; x.bias = t->params + L.b; x.stats = stats; x.accumulate = 0;
        x.act_tab = InA.tab; x.act_cols = InA.tab ? InA.cols : 0;
        x.a_padded = (InA.padded && ld_in >= (int64_t)x.ksteps * 16) ? 1 : 0;
        return launch_gemm_bf16x3(x, st);
    }
    GemmArgs g{};
    g.A = In; g.B = t->params + L.w; g.C = Z;
    g.M = M; g.N = L.n_out; g.K = L.n_in;
    g.sAm = ld_in; g.sAk = 1; g.sBk = 1; g.sBn = L.n_in; g.ldc = ldz;
    g.alpha = alpha; g.bias = t->params + L.b; g.colsum = colsum; g.flags = 0; g.splitk = 1;
    return launch_gemm(g, st);
}
// dIn[M, n_cols] (+)= alpha * dZ[M, n_out] W[:, :n_cols]
// `below` (optional): the SineLayer whose output gradient dIn is.  When the bf16x3 kernel runs and that layer's [a | b] table
// exists (activation on load), its activation backward (x cos, column sums into t->bn_stats) happens in the epilogue and
// *fused reports it: the consumer then skips its own reduction sweep.
struct ActBelow {
    const LayerP* L;
    Act Z;
    const float* tab;
    float* bnslot;
};
static hipError_t linear_dgrad(snerf_trainer* t, const LayerP& L, const float* dZ, int64_t ldz, int64_t M, float* dIn, int64_t ld_in,
                               int n_cols, float alpha, bool accumulate, hipStream_t st, const ActBelow* below = nullptr,
                               bool* fused = nullptr) {
    if (fused) *fused = false;
    static const int thin = [] { const char* e = getenv("SNERF_THIN_DGRAD"); return (e && e[0] == '0') ? 0 : 1; }();
    if (thin && t->gemm_mode == 1 && M >= 1024 && L.n_out <= 4) {      // a thin head: a rank-K update of dIn, streamed (exact fp32)
        ThinDgradArgs a{};
        a.D = dZ; a.ldd = ldz; a.W = t->params + L.w; a.ldw = L.n_in; a.C = dIn; a.ldc = ld_in; a.M = M; a.K = L.n_out; a.N = n_cols;
        a.accumulate = accumulate ? 1 : 0; a.alpha = alpha;
        const bool act = below && below->tab && below->L->n_out == n_cols;
        if (act) {
            a.ez = below->Z.p; a.eld = below->Z.ld; a.etab = below->tab; a.stats = t->bn_stats;
            if (below->L->bn) { a.emu = below->bnslot + 2 * t->W; a.eistd = below->bnslot + 3 * t->W; }
        }
        if (thin_dgrad_ok(a)) {
            if (act && fused) *fused = true;
            return launch_thin_dgrad(a, st);
        }
    }
    if (rows_ok(t, M, L.n_out, n_cols)) {
        GemmX x{};
        x.n_tiles = (n_cols + 31) / 32; x.ksteps = (L.n_out + 15) / 16;
        // Bt[n = input feature][k = output feature] = W[k][n]: transposed split
        x.W = t->params + L.w; x.w_rows = L.n_out; x.w_cols = L.n_in; x.w_transpose = 1;
        x.A = dZ; x.frag = t->w_frag; x.C = dIn; x.M = M; x.N = n_cols; x.K = L.n_out; x.lda = ldz; x.ldc = ld_in;
        x.alpha = alpha; x.bias = nullptr; x.stats = nullptr; x.accumulate = accumulate ? 1 : 0;
        if (below && below->tab && below->L->n_out == n_cols) {      // with `accumulate` the caller guarantees this is the last producer
            x.stats = t->bn_stats; x.ez = below->Z.p; x.eld = below->Z.ld; x.etab = below->tab;      // sums: zero here (cleared by their last consumer)
            if (below->L->bn) { x.emu = below->bnslot + 2 * t->W; x.eistd = below->bnslot + 3 * t->W; }
            if (fused) *fused = true;
        }
        return launch_gemm_bf16x3(x, st);
    }
    GemmArgs g{};
    g.A = dZ; g.B = t->params + L.w; g.C = dIn;
    g.M = M; g.N = n_cols; g.K = L.n_out;
    g.sAm = ldz; g.sAk = 1; g.sBk = L.n_in; g.sBn = 1; g.ldc = ld_in;
    g.alpha = alpha; g.bias = nullptr; g.colsum = nullptr; g.flags = accumulate ? GEMM_ACCUM : 0; g.splitk = 1;
    return launch_gemm(g, st);
}
// dW[n_out, n_in] += alpha * dZ^T In   (split over the point dimension, fp32 atomics)
// can the BatchNorm dZ pass ride in the weight-gradient kernel?  (bf16x3 path, one block column over the inputs: in-place dZ)
// A layer with more than 256 inputs (fc5: [fc4 | PE]) goes as two launches: the first 256 input columns with the dZ pass, then the
// rest on the finished dZ.
static bool wgrad_bn_ok(const snerf_trainer* t, const LayerP& L, int64_t M, const Act& In) {
    (void)L; (void)In;
    return t->gemm_mode == 1 && M >= 1024;
}
static hipError_t linear_wgrad(snerf_trainer* t, const LayerP& L, float* dZ, int64_t ldz, Act InA, int64_t M, float alpha, hipStream_t st,
                               const WgradBN* bn = nullptr) {
    const float* In = InA.p;
    const int64_t ld_in = InA.ld;
    static const int thin = [] { const char* e = getenv("SNERF_THIN_WGRAD"); return (e && e[0] == '0') ? 0 : 1; }();
    if (thin && !bn && t->gemm_mode == 1 && M >= 1024 && L.n_out <= 4) {      // a thin head: K x n_in sums, streamed over the input (exact fp32)
        ThinWgradArgs a{};
        a.D = dZ; a.ldd = ldz; a.In = In; a.ldi = ld_in; a.M = M; a.K = L.n_out; a.N = L.n_in; a.alpha = alpha;
        a.dW = t->grads + L.w; a.ldw = L.n_in; a.tab = InA.tab; a.tab_cols = InA.tab ? InA.cols : 0; a.tab_stride = a.tab_cols;
        if (thin_wgrad_ok(a)) return launch_thin_wgrad(a, st);
    }
    if (t->gemm_mode == 1 && M >= 1024) {
        if (bn && L.n_in > 256) {
            // the activation-on-load table covers the first tc input columns: each launch gets its window of it ([a | b], b at distance tc)
            const int n0 = 256, tc = InA.tab ? InA.cols : 0;
            hipError_t e = launch_wgrad_bf16x3(dZ, ldz, In, ld_in, M, L.n_out, n0, alpha, t->grads + L.w, L.n_in, st, InA.tab, tc < n0 ? tc : n0, bn, tc);
            if (e != hipSuccess) return e;
            const int rest = tc > n0 ? tc - n0 : 0;
            return launch_wgrad_bf16x3(dZ, ldz, In + n0, ld_in, M, L.n_out, L.n_in - n0, alpha, t->grads + L.w + n0, L.n_in, st,
                                       rest ? InA.tab + n0 : nullptr, rest, nullptr, tc);
        }
        return launch_wgrad_bf16x3(dZ, ldz, In, ld_in, M, L.n_out, L.n_in, alpha, t->grads + L.w, L.n_in, st, InA.tab, InA.tab ? InA.cols : 0, bn);
    }
    if (bn) return hipErrorInvalidValue;
    if (InA.tab) return hipErrorInvalidValue;
    GemmArgs g{};
    g.A = dZ; g.B = In; g.C = t->grads + L.w;
    g.M = L.n_out; g.N = L.n_in; g.K = M;
    g.sAm = 1; g.sAk = ldz; g.sBk = ld_in; g.sBn = 1; g.ldc = L.n_in;
    g.alpha = alpha; g.bias = nullptr; g.colsum = nullptr; g.flags = GEMM_ATOMIC;
    int64_t sk = (M + 2047) / 2048;
    if (sk > 1024) sk = 1024;
    if (sk < 1) sk = 1;
    g.splitk = (int)sk;
    return launch_gemm(g, st);
}

// SineLayer forward in train mode: Z = 30(In W^T + b) [stashed], H = sin(BN_batch(Z)) or sin(Z)
// tab_out != nullptr: activation on load - no sin pass; the layer's [a | b] table is written instead and H stays unwritten
static int sine_fwd(snerf_trainer* t, const LayerP& L, Act In, int64_t M, Act Z, Act H, float* bnslot,
                    bool train_bn, hipStream_t st, float* tab_out = nullptr) {
    const int C = L.n_out;
    const int64_t Mg = M * (t->ar_fn ? t->world : 1);      // rows of the global batch (equal shards per rank)
    if (L.bn) {
        float *colsum = bnslot, *m2 = bnslot + t->W, *mean = bnslot + 2 * t->W, *istd = bnslot + 3 * t->W;
        if (train_bn && rows_ok(t, M, L.n_in, L.n_out)) {
            // batch statistics from the GEMM epilogue: shifted sums (shift = 30 b) in double, no extra pass over Z
            // t->bn_stats is zero here: every consumer of the sums clears them (and each pass starts with one memset)
            HIPCK(linear_fwd(t, L, In, M, Z.p, Z.ld, 30.f, nullptr, st, t->bn_stats));
            RCI(all_reduce(t, t->bn_stats, 2 * C, true, st));          // the shift 30 b is the same on every rank: the sums add
            HIPCK(launch_bn_finalize_shifted(t->bn_stats, t->params + L.b, 30.f, Mg, C, mean, istd, t->buffers + L.rm, t->buffers + L.rv,
                                             t->params + L.g, t->params + L.beta, tab_out, st));
            if (tab_out) return SNERF_OK;                               // the table is written; no sin pass with activation on load
        } else if (train_bn) {
            HIPCK(snerf_zero_async(bnslot, 0, 2 * t->W * sizeof(float), st));
            HIPCK(linear_fwd(t, L, In, M, Z.p, Z.ld, 30.f, colsum, st));
            RCI(all_reduce(t, colsum, C, false, st));
            HIPCK(launch_bn_finalize(colsum, m2, Mg, C, mean, istd, nullptr, nullptr, 0, st));
            ColArgs ca{};
            ca.mode = 0; ca.M = M; ca.C = C; ca.ld = Z.ld; ca.Z = Z.p; ca.mu = mean; ca.out0 = m2; ca.alpha0 = 1.f;
            HIPCK(launch_colreduce(ca, st));
            RCI(all_reduce(t, m2, C, false, st));
            HIPCK(launch_bn_finalize(colsum, m2, Mg, C, mean, istd, t->buffers + L.rm, t->buffers + L.rv, 1, st));
        } else {      // eval-mode statistics (running estimates)
            HIPCK(linear_fwd(t, L, In, M, Z.p, Z.ld, 30.f, nullptr, st));
            HIPCK(snerf_copy_async(mean, t->buffers + L.rm, C * sizeof(float), st));
            // istd from running var: reuse finalize stage 1 arithmetic with M = 1 via a tiny dedicated path
            HIPCK(launch_bn_finalize(nullptr, t->buffers + L.rv, 1, C, mean, istd, nullptr, nullptr, 2, st));
        }
        if (tab_out) HIPCK(launch_act_table(mean, istd, t->params + L.g, t->params + L.beta, C, tab_out, st));
        else HIPCK(launch_sin_fwd(Z.p, H.p, M, C, Z.ld, H.ld, mean, istd, t->params + L.g, t->params + L.beta, st));
    } else {
        HIPCK(linear_fwd(t, L, In, M, Z.p, Z.ld, 30.f, nullptr, st));
        // without BatchNorm the table is the constant [1/(2 pi) | 0]: written once at bind time (const_tables)
        if (!tab_out) HIPCK(launch_sin_fwd(Z.p, H.p, M, C, Z.ld, H.ld, nullptr, nullptr, nullptr, nullptr, st));
    }
    return SNERF_OK;
}

// SineLayer backward.  D holds dL/dH on entry ([M, n_out], ld = D.ld) and dL/dZ on exit (in place).
// Accumulates weight/bias (and BN affine) gradients; if dIn.p != null writes dL/dIn[:, :n_in_cols].
static int sine_bwd(snerf_trainer* t, const LayerP& L, Act D, Act Z, Act In, int64_t M, float* bnslot,
                    Act dIn, int n_in_cols, bool accumulate_in, hipStream_t st, bool pre_activated = false,
                    const ActBelow* below = nullptr, bool* fused_below = nullptr) {
    // pre_activated: the dgrad that produced D already applied this layer's activation backward (D = dL/dH * cos) and left the
    // column sums in t->bn_stats.  below / fused_below: the same offer for the layer that will consume dIn.
    const int C = L.n_out;
    if (fused_below) *fused_below = false;
    // EXPERIMENT (never on by default): the inter-layer gradient as bf16.  Rounds dL/dY (as the dgrad above would have stored it) and, behind the
    // weight-gradient kernel, dL/dZ (as it would be written back) to bf16 precision in place: the accuracy side of the idea on the real kernels.
    static const bool dy_bf16 = getenv("SNERF_TRAIN_DY_BF16") != nullptr;
    if (dy_bf16 && pre_activated) HIPCK(launch_round_bf16(D.p, D.ld, M, C, st));
    if (L.bn) {
        float *mean = bnslot + 2 * t->W, *istd = bnslot + 3 * t->W;
        float *sdy = t->bn_bwd, *sdyx = t->bn_bwd + t->W;
        HIPCK(snerf_zero_async(t->bn_bwd, 0, 2 * t->W * sizeof(float), st));      // [2][W]: a narrower layer leaves the tails zero (the
                                                                                 // whole buffer goes through the sync-BatchNorm all-reduce)
        if (pre_activated) {
            // d beta += sum dY, d gamma += sum dY*xhat (per-rank sums), and the fp32 copies the dZ pass reads
            HIPCK(launch_act_sums_finalize(t->bn_stats, C, 1.f, sdy, sdyx, t->grads + L.beta, t->grads + L.g, st));
        } else {
            ColArgs ca{};
            ca.mode = 1; ca.M = M; ca.C = C; ca.ld = Z.ld; ca.Z = Z.p; ca.D = D.p; ca.mu = mean; ca.istd = istd;
            ca.gamma = t->params + L.g; ca.beta = t->params + L.beta; ca.out0 = sdy; ca.out1 = sdyx; ca.alpha0 = 1.f;
            ca.ldd = D.ld;
            HIPCK(launch_colreduce(ca, st));
            // d beta = sum dY, d gamma = sum dY*xhat
            HIPCK(launch_copy_cols(sdy, C, t->grads + L.beta, C, 1, C, true, st));       // parameter gradients stay per-rank sums
            HIPCK(launch_copy_cols(sdyx, C, t->grads + L.g, C, 1, C, true, st));
        }
        RCI(all_reduce(t, t->bn_bwd, 2 * t->W, false, st));                               // the BatchNorm backward means are global
        const int64_t Mg = M * (t->ar_fn ? t->world : 1);
        if (pre_activated && wgrad_bn_ok(t, L, M, In)) {
            // D holds dL/dY: the dZ sweep rides in the weight-gradient kernel (dY -> dZ in registers, written back in place)
            WgradBN bn{Z.p, Z.ld, t->params + L.g, mean, istd, sdy, sdyx, 1.f / (float)Mg, 30.f, t->grads + L.b};
            HIPCK(linear_wgrad(t, L, D.p, D.ld, In, M, 30.f, st, &bn));
            if (dy_bf16) HIPCK(launch_round_bf16(D.p, D.ld, M, C, st));
            if (dIn.p) HIPCK(linear_dgrad(t, L, D.p, D.ld, M, dIn.p, dIn.ld, n_in_cols, 30.f, accumulate_in, st, below, fused_below));
            return SNERF_OK;
        }
        HIPCK(launch_bn_bwd2(Z.p, D.p, M, C, Z.ld, D.ld, mean, istd, t->params + L.g, t->params + L.beta, sdy, sdyx, t->grads + L.b, 30.f,
                             Mg, st, pre_activated));
    } else if (pre_activated) {
        HIPCK(launch_act_sums_finalize(t->bn_stats, C, 30.f, nullptr, nullptr, t->grads + L.b, nullptr, st));     // d bias = 30 * sum dZ
    } else {
        ColArgs ca{};
        ca.mode = 2; ca.M = M; ca.C = C; ca.ld = Z.ld; ca.ldd = D.ld; ca.Z = Z.p; ca.D = D.p; ca.out0 = t->grads + L.b; ca.alpha0 = 30.f;
        HIPCK(launch_colreduce(ca, st));
    }
    HIPCK(linear_wgrad(t, L, D.p, D.ld, In, M, 30.f, st));
    if (dIn.p) HIPCK(linear_dgrad(t, L, D.p, D.ld, M, dIn.p, dIn.ld, n_in_cols, 30.f, accumulate_in, st, below, fused_below));
    return SNERF_OK;
}

// plain Linear (heads): Out[M, n_out] = In W^T + b
static int plain_fwd(snerf_trainer* t, const LayerP& L, Act In, int64_t M, float* Out, int64_t ldo, hipStream_t st) {
    static const int thin = [] { const char* e = getenv("SNERF_FUSED_HEADS"); return (e && e[0] == '0') ? 0 : 1; }();
    if (thin && t->gemm_mode == 1 && M >= 1024 && L.n_out <= 4) {      // a head with at most four outputs: a stream over its input (thin_fwd_kernel), not a 32-column MFMA tile
        ThinFwdArgs f{};
        f.In = In.p; f.ldi = In.ld; f.M = M; f.K = L.n_out; f.N = L.n_in; f.alpha = 1.f; f.Out = Out; f.ldo = ldo;
        f.W = t->params + L.w; f.ldw = L.n_in; f.bias = t->params + L.b; f.tab = In.tab; f.tab_cols = In.tab ? In.cols : 0; f.tab_stride = f.tab_cols;
        if (thin_fwd_ok(f)) { HIPCK(launch_thin_fwd(f, st)); return SNERF_OK; }
    }
    HIPCK(linear_fwd(t, L, In, M, Out, ldo, 1.f, nullptr, st));
    return SNERF_OK;
}
static int plain_bwd(snerf_trainer* t, const LayerP& L, float* dOut, int64_t ldo, Act In, int64_t M,
                     float* dIn, int64_t ld_din, bool accumulate, hipStream_t st, const ActBelow* below = nullptr, bool* fused_below = nullptr) {
    if (fused_below) *fused_below = false;
    HIPCK(linear_wgrad(t, L, dOut, ldo, In, M, 1.f, st));
    HIPCK(launch_colsum(dOut, M, L.n_out, ldo, 1.f, t->grads + L.b, st));
    if (dIn) HIPCK(linear_dgrad(t, L, dOut, ldo, M, dIn, ld_din, L.n_in, 1.f, accumulate, st, below, fused_below));
    return SNERF_OK;
}

#define RC(x)                 \
    do {                      \
        int _rc = (x);        \
        if (_rc) return _rc;  \
    } while (0)

static int forward_pass(snerf_trainer* t, snerf_trainer::Pass& P, bool solar, int64_t R, int S, const float* top, const float* bot,
                        const float* tvals, const float* sun, const float* time, bool train_bn, hipStream_t st,
                        float* adjust_col_out = nullptr) {
    const int W = t->W, W2 = t->W2, W4 = t->W4, C = t->C;
    const int64_t N = R * S;
    // the column-sum scratch starts every pass at zero (its consumers clear what they read: no memset per layer)
    HIPCK(snerf_zero_async(t->bn_stats, 0, (size_t)(4 * W + 2) * sizeof(float), st));
    HIPCK(snerf_copy_async(P.top, top, R * 3 * sizeof(float), st));
    HIPCK(snerf_copy_async(P.bot, bot, R * 3 * sizeof(float), st));
    HIPCK(snerf_copy_async(P.tvals, tvals, S * sizeof(float), st));
    auto& Ls = t->layers;
    PeArgs pa{};
    pa.n = N; pa.n_samples = S; pa.top = top; pa.bot = bot; pa.tvals = tvals; pa.pe = P.E.p; pa.pts = P.pts;
    pa.pe2 = P.In5.p + W; pa.ld2 = W + 64;                  // second copy: the PE columns of fc5's concat input
    HIPCK(launch_pe_points(pa, st));
    // Activation on load (per-point layers): store only the pre-activations Z; every consumer applies sin(BN(.)) while loading.
    // Needs the bf16x3 kernels for every consumer (K <= 512, widths multiples of 16) and enough points for them to be used.
    const bool aol = t->aol_mode == 1 && t->gemm_mode == 1 && N >= 1024 && W % 16 == 0 && gemm_rows_group_tiles((W + 64 + 15) / 16) > 0;
    P.aol = aol;
    auto tab_of = [&](int slot) { return aol ? P.tabs + (int64_t)slot * 2 * W : (float*)nullptr; };       // slots: 0-8 trunk, 9-11 adjust, 12-14 solar
    auto view = [&](Act Z, Act H, int slot, int n) { return aol ? Act{Z.p, Z.ld, tab_of(slot), n} : H; };
    // fc4's output lives inside fc5's concat input [ . | PE]: its H without, its Z with activation on load
    // likewise fc9's inside fc_solar_1's concat input [ . | PE(sun)]
    P.Zc3 = aol ? Act{P.In5.p, W + 64} : P.Z[3];
    P.Zc8 = aol ? Act{P.In_s1.p, P.In_s1.ld} : P.Z[8];
    auto zof = [&](int l) { return l == 3 ? P.Zc3 : (l == 8 ? P.Zc8 : P.Z[l]); };
    for (int l = 0; l < 9; ++l) P.Hc[l] = view(zof(l), P.H[l], l, l == 8 ? W2 : W);
    for (int l = 0; l < 3; ++l) { P.Hac[l] = view(P.Za[l], P.Ha[l], 9 + l, W); P.Hsc[l] = view(P.Zs[l], P.Hs[l], 12 + l, W2); }
    P.In5c = aol ? Act{P.In5.p, W + 64, tab_of(3), W, true} : Act{P.In5.p, W + 64, nullptr, 0, true};      // column W+63 is the PE's zero pad
    P.In_s1c = aol ? Act{P.In_s1.p, P.In_s1.ld, tab_of(8), W2, true} : Act{P.In_s1.p, P.In_s1.ld, nullptr, 0, true};
    // trunk (G_NeRF.py:80-91)
    RC(sine_fwd(t, Ls[L_FC1], Act{P.E.p, 64, nullptr, 0, true}, N, P.Z[0], P.H[0], nullptr, train_bn, st, tab_of(0)));
    for (int l = 1; l < 9; ++l) {
        const Act In = l == 4 ? P.In5c : P.Hc[l - 1];
        RC(sine_fwd(t, Ls[L_FC1 + l], In, N, zof(l), P.H[l], P.bn + (l - 1) * 4 * W, train_bn, st, tab_of(l)));
    }
    const Act X1 = P.Hc[8];
    {   // colour (3) and density (1) heads: one stream over X1 for both where it applies (thin_fwd_kernel), else one row GEMM each
        static const int fused = [] { const char* e = getenv("SNERF_FUSED_HEADS"); return (e && e[0] == '0') ? 0 : 1; }();
        ThinFwdArgs f{};
        f.In = X1.p; f.ldi = X1.ld; f.M = N; f.K = 4; f.N = Ls[L_COL].n_in; f.alpha = 1.f; f.Out = P.head.p; f.ldo = 4;
        f.W = t->params + Ls[L_COL].w; f.ldw = Ls[L_COL].n_in; f.W3 = t->params + Ls[L_SIG].w; f.bias = t->params + Ls[L_COL].b; f.bias3 = t->params + Ls[L_SIG].b;
        f.tab = X1.tab; f.tab_cols = X1.tab ? X1.cols : 0; f.tab_stride = f.tab_cols;
        if (fused && t->gemm_mode == 1 && N >= 1024 && Ls[L_COL].n_out == 3 && Ls[L_SIG].n_out == 1 && Ls[L_SIG].n_in == f.N && thin_fwd_ok(f)) {
            HIPCK(launch_thin_fwd(f, st));
        } else {
            RC(plain_fwd(t, Ls[L_COL], X1, N, P.head.p, 4, st));
            RC(plain_fwd(t, Ls[L_SIG], X1, N, P.head.p + 3, 4, st));
        }
    }
    // solar visibility branch (G_NeRF.py:100-108)
    HIPCK(launch_pe_small(sun, 3, 3, 4, R, P.pe_sun, 28, st));
    if (!aol) HIPCK(launch_copy_cols(X1.p, X1.ld, P.In_s1.p, P.In_s1.ld, N, W2, false, st));
    HIPCK(launch_bcast_rows(P.pe_sun, 28, P.In_s1.p, P.In_s1.ld, W2, N, S, st));
    RC(sine_fwd(t, Ls[L_S1], P.In_s1c, N, P.Zs[0], P.Hs[0], nullptr, train_bn, st, tab_of(12)));
    RC(sine_fwd(t, Ls[L_S2], P.Hsc[0], N, P.Zs[1], P.Hs[1], nullptr, train_bn, st, tab_of(13)));
    RC(sine_fwd(t, Ls[L_S3], P.Hsc[1], N, P.Zs[2], P.Hs[2], nullptr, train_bn, st, tab_of(14)));
    RC(plain_fwd(t, Ls[L_S4], P.Hsc[2], N, P.sv_raw.p, 1, st));
    // sky colour head (G_NeRF.py:110-111), per ray
    RC(sine_fwd(t, Ls[L_K1], Act{P.pe_sun, 28}, R, Act{P.Zk1, W4}, Act{P.Hk1, W4}, nullptr, train_bn, st));
    RC(plain_fwd(t, Ls[L_K2], Act{P.Hk1, W4}, R, P.sky_raw, 3, st));
    HIPCK(launch_sigmoid(P.sky_raw, P.sky, R * 3, st));
    PointOutArgs po{};
    po.n = N; po.n_samples = S; po.C = C; po.head = P.head.p; po.sv_raw = P.sv_raw.p; po.rho = P.rho; po.sv = P.sv;
    if (!solar) {
        // seasonal colour-adjust branch (T_NeRF_net_v2.py:83-88) and time -> class softmax (:77-78, per ray)
        RC(sine_fwd(t, Ls[L_A1], X1, N, P.Za[0], P.Ha[0], nullptr, train_bn, st, tab_of(9)));
        RC(sine_fwd(t, Ls[L_A2], P.Hac[0], N, P.Za[1], P.Ha[1], nullptr, train_bn, st, tab_of(10)));
        RC(sine_fwd(t, Ls[L_A3], P.Hac[1], N, P.Za[2], P.Ha[2], nullptr, train_bn, st, tab_of(11)));
        RC(plain_fwd(t, Ls[L_AC], P.Hac[2], N, P.adj.p, 3 * C, st));
        HIPCK(launch_pe_small(time, 4, 2, 2, R, P.pe_time, 12, st));
        RC(sine_fwd(t, Ls[L_T1], Act{P.pe_time, 12}, R, Act{P.Zt1, W}, Act{P.Ht1, W}, nullptr, train_bn, st));
        RC(sine_fwd(t, Ls[L_T2], Act{P.Ht1, W}, R, Act{P.Zt2, W}, Act{P.Ht2, W}, nullptr, train_bn, st));
        RC(plain_fwd(t, Ls[L_CL], Act{P.Ht2, W}, R, P.logits, C, st));
        HIPCK(launch_softmax(P.logits, P.cls, R, C, st));
        po.adj = P.adj.p; po.cls = P.cls; po.col = P.col; po.adjust_col = adjust_col_out;
    }
    HIPCK(launch_point_out(po, false, st));
    return SNERF_OK;
}

extern "C" {

static std::mutex g_live_mu;
static std::set<const snerf_trainer*> g_live;          // trainers created and not yet destroyed

snerf_trainer* snerf_trainer_create(int layer_width, int n_classes) {
    if (layer_width < 16 || layer_width % 4 != 0 || n_classes < 1 || n_classes > 5) {
        snerf_set_error(SNERF_E_INVALID, "snerf_trainer_create: width must be a multiple of 4 (>=16), classes in [1,5]");
        return nullptr;
    }
    snerf_trainer* t = new snerf_trainer();
    t->W = layer_width; t->C = n_classes; t->W2 = layer_width / 2; t->W4 = layer_width / 4;
    if (const char* e = getenv("SNERF_TRAIN_GEMM")) t->gemm_mode = std::strcmp(e, "fp32") == 0 ? 0 : 1;
    if (const char* e = getenv("SNERF_TRAIN_AOL")) t->aol_mode = std::strcmp(e, "0") == 0 ? 0 : 1;
    build_layers(t);
    {
        std::lock_guard<std::mutex> lock(g_live_mu);
        g_live.insert(t);
    }
    return t;
}
void snerf_trainer_destroy(snerf_trainer* t) {
    {
        std::lock_guard<std::mutex> lock(g_live_mu);
        g_live.erase(t);
    }
    delete t;
}
// Class count of a LIVE trainer, -1 for a pointer that is not one (never created here, or destroyed): what a caller that holds the handle as
// a plain integer (the PyTorch custom ops, csrc/ops.cpp) checks before it dereferences it or sizes an output by a class count.
int snerf_trainer_classes(const snerf_trainer* t) {
    std::lock_guard<std::mutex> lock(g_live_mu);
    return g_live.count(t) ? t->C : -1;
}
int snerf_trainer_bound_sizes(const snerf_trainer* t, int64_t* n_rays, int64_t* n_solar_rays, int* n_samples) {
    if (!t || !t->ws) return snerf_set_error(SNERF_E_STATE, "trainer not bound (call snerf_trainer_bind)");
    if (n_rays) *n_rays = t->R;
    if (n_solar_rays) *n_solar_rays = t->Rs;
    if (n_samples) *n_samples = t->S;
    return SNERF_OK;
}
int64_t snerf_trainer_param_floats(const snerf_trainer* t) { return t ? t->n_params : 0; }
int64_t snerf_trainer_buffer_floats(const snerf_trainer* t) { return t ? t->n_buffers : 0; }
int snerf_trainer_tensor_count(const snerf_trainer* t) {
    if (!t) return 0;
    int n = 0;
    for (const LayerP& l : t->layers) n += 2 + (l.bn ? 4 : 0);
    return n;
}
int snerf_trainer_tensor_info(const snerf_trainer* t, int index, char* key, int key_cap, int* is_buffer, int64_t* offset,
                              int64_t* numel, int* rows, int* cols) {
    if (!t || !key) return snerf_set_error(SNERF_E_INVALID, "snerf_trainer_tensor_info: bad argument");
    int i = 0;
    for (const LayerP& l : t->layers) {
        const std::string lin = l.sine ? l.name + ".linear" : l.name;
        struct Ent { std::string k; int buf; int64_t off, n; int r, c; };
        std::vector<Ent> ents = {{lin + ".weight", 0, l.w, (int64_t)l.n_out * l.n_in, l.n_out, l.n_in}, {lin + ".bias", 0, l.b, l.n_out, l.n_out, 1}};
        if (l.bn) {
            ents.push_back({l.name + ".norm.weight", 0, l.g, l.n_out, l.n_out, 1});
            ents.push_back({l.name + ".norm.bias", 0, l.beta, l.n_out, l.n_out, 1});
            ents.push_back({l.name + ".norm.running_mean", 1, l.rm, l.n_out, l.n_out, 1});
            ents.push_back({l.name + ".norm.running_var", 1, l.rv, l.n_out, l.n_out, 1});
        }
        for (const Ent& e : ents) {
            if (i == index) {
                std::strncpy(key, e.k.c_str(), key_cap - 1);
                key[key_cap - 1] = 0;
                if (is_buffer) *is_buffer = e.buf;
                if (offset) *offset = e.off;
                if (numel) *numel = e.n;
                if (rows) *rows = e.r;
                if (cols) *cols = e.c;
                return SNERF_OK;
            }
            ++i;
        }
    }
    return snerf_set_error(SNERF_E_INVALID, "snerf_trainer_tensor_info: index out of range");
}

size_t snerf_trainer_workspace_bytes(snerf_trainer* t, int64_t n_rays, int64_t n_solar_rays, int n_samples) {
    if (!t) return 0;
    snerf_trainer tmp = *t;
    return carve(&tmp, nullptr, n_rays, n_solar_rays, n_samples);
}

int snerf_trainer_bind(snerf_trainer* t, float* d_params, float* d_grads, float* d_adam_m, float* d_adam_v, float* d_buffers,
                       void* d_workspace, size_t workspace_bytes, int64_t n_rays, int64_t n_solar_rays, int n_samples) {
    if (!t || !d_params || !d_grads || !d_buffers || !d_workspace || n_rays < 0 || n_solar_rays < 0 || n_samples < 1)
        return snerf_set_error(SNERF_E_INVALID, "snerf_trainer_bind: bad argument");
    const size_t need = snerf_trainer_workspace_bytes(t, n_rays, n_solar_rays, n_samples);
    if (workspace_bytes < need) return snerf_set_error(SNERF_E_INVALID, "snerf_trainer_bind: workspace too small");
    t->params = d_params; t->grads = d_grads; t->adam_m = d_adam_m; t->adam_v = d_adam_v; t->buffers = d_buffers;
    t->ws = (char*)d_workspace; t->ws_bytes = workspace_bytes;
    t->R = n_rays; t->Rs = n_solar_rays; t->S = n_samples;
    carve(t, t->ws, n_rays, n_solar_rays, n_samples);
    for (snerf_trainer::Pass* P : {&t->img, &t->sol}) {
        // the zero pad columns of the solar branch's concat input
        if (P->N > 0) HIPCK(hipMemset(P->In_s1.p, 0, (size_t)P->N * P->In_s1.ld * sizeof(float)));
        // activation-on-load tables of the layers without BatchNorm (fc1, adjust 1-3, solar 1-3) are constants: [1/(2 pi) | 0]
        const int slots[7] = {0, 9, 10, 11, 12, 13, 14};
        for (int sl : slots) HIPCK(launch_act_table(nullptr, nullptr, nullptr, nullptr, sl >= 12 ? t->W2 : t->W, P->tabs + (int64_t)sl * 2 * t->W, nullptr));
    }
    HIPCK(hipMemset(t->bn_stats, 0, (size_t)(4 * t->W + 2) * sizeof(float)));
    HIPCK(hipDeviceSynchronize());
    return SNERF_OK;
}

static int check_bound(const snerf_trainer* t, int64_t R, int S, bool solar) {
    if (!t || !t->ws) return snerf_set_error(SNERF_E_STATE, "trainer not bound (call snerf_trainer_bind)");
    if (S != t->S || R != (solar ? t->Rs : t->R)) return snerf_set_error(SNERF_E_INVALID, "ray/sample counts differ from the bound sizes");
    return SNERF_OK;
}

int snerf_trainer_forward_image(snerf_trainer* t, int64_t n_rays, int n_samples, const float* d_top, const float* d_bot,
                                const float* d_tvals, const float* d_sun, const float* d_time, int train_bn, int flags,
                                const snerf_composite_out* out, float* d_sky, float* d_classes, const snerf_field_out* per_sample,
                                void* stream) {
    RC(check_bound(t, n_rays, n_samples, false));
    if (!d_top || !d_bot || !d_tvals || !d_sun || !d_time || !out) return snerf_set_error(SNERF_E_INVALID, "snerf_trainer_forward_image: bad argument");
    CtxGuard ctx(t);
    t->img_flags = flags;
    hipStream_t st = (hipStream_t)stream;
    RC(forward_pass(t, t->img, false, n_rays, n_samples, d_top, d_bot, d_tvals, d_sun, d_time, train_bn != 0, st,
                    per_sample ? per_sample->d_adjust_col : nullptr));
    snerf_trainer::Pass& P = t->img;
    RC(snerf_composite_rays(n_rays, n_samples, d_top, d_bot, d_tvals, P.rho, P.col, P.sv, P.sky, flags, nullptr, 1.f, out, stream));
    const int64_t N = n_rays * n_samples;
    if (d_sky) HIPCK(snerf_copy_async(d_sky, P.sky, n_rays * 3 * sizeof(float), st));
    if (d_classes) HIPCK(snerf_copy_async(d_classes, P.cls, n_rays * t->C * sizeof(float), st));
    if (per_sample) {
        if (per_sample->d_rho) HIPCK(snerf_copy_async(per_sample->d_rho, P.rho, N * 4, st));
        if (per_sample->d_solar_vis) HIPCK(snerf_copy_async(per_sample->d_solar_vis, P.sv, N * 4, st));
        if (per_sample->d_col) HIPCK(snerf_copy_async(per_sample->d_col, P.col, N * 12, st));
        if (per_sample->d_points) HIPCK(snerf_copy_async(per_sample->d_points, P.pts, N * 12, st));
        if (per_sample->d_col_raw) HIPCK(launch_copy_cols(P.head.p, 4, per_sample->d_col_raw, 3, N, 3, false, st));
        if (per_sample->d_adjust) HIPCK(snerf_copy_async(per_sample->d_adjust, P.adj.p, N * 3 * t->C * 4, st));
    }
    return SNERF_OK;
}

// Backward of the image pass through the network, from the per-sample gradients already sitting in the workspace:
// t->d_rho [N], t->d_col [N,3], t->d_sky [R,3] and - when `classic` (the solar visibility carries gradient) - dL/dSolar_Vis
// in t->d_sv_raw [N].  d_g_classes (optional, [R,C]) is added to the class-probability gradient.
static int network_backward_image(snerf_trainer* t, bool classic, const float* d_g_classes, hipStream_t st);

static int backward_image(snerf_trainer* t, const float* d_g_rgb, const float* d_g_albedo, const float* d_g_sky,
                          const float* d_g_pe, const float* d_rho_prior, float trust, const float* d_trust, const float* d_g_rgb_merged,
                          const float* d_g_albedo_merged, void* stream) {
    if (!t || !t->ws) return snerf_set_error(SNERF_E_STATE, "trainer not bound");
    CtxGuard ctx(t);
    hipStream_t st = (hipStream_t)stream;
    snerf_trainer::Pass& P = t->img;
    const int S = t->S;
    const int64_t R = P.R;
    CompBwdArgs cb{};
    cb.n_rays = R; cb.n_samples = S; cb.top = P.top; cb.bot = P.bot; cb.rho = P.rho; cb.col = P.col; cb.sv = P.sv; cb.sky = P.sky;
    cb.g_rgb = d_g_rgb; cb.g_albedo = d_g_albedo; cb.g_pe = d_g_pe; cb.d_rho = t->d_rho; cb.d_col = t->d_col; cb.d_sky = t->d_sky;
    cb.rho_prior = d_rho_prior; cb.trust = trust; cb.trust_dev = d_trust; cb.g_rgb_m = d_g_rgb_merged; cb.g_albedo_m = d_g_albedo_merged;
    const bool classic = (t->img_flags & 1) != 0;
    cb.classic = classic ? 1 : 0; cb.d_sv = t->d_sv_raw;       // dL/dSolar_Vis, turned into dL/d(raw) in place below
    HIPCK(launch_composite_bwd(cb, st));
    if (d_g_sky) HIPCK(launch_copy_cols(d_g_sky, 3, t->d_sky, 3, R, 3, true, st));
    return network_backward_image(t, classic, nullptr, st);
}

int snerf_trainer_backward_image(snerf_trainer* t, const float* d_g_rgb, const float* d_g_albedo, const float* d_g_sky,
                                 const float* d_g_pe, const float* d_rho_prior, float trust, const float* d_g_rgb_merged,
                                 const float* d_g_albedo_merged, void* stream) {
    return backward_image(t, d_g_rgb, d_g_albedo, d_g_sky, d_g_pe, d_rho_prior, trust, nullptr, d_g_rgb_merged, d_g_albedo_merged, stream);
}

int snerf_trainer_backward_image_dt(snerf_trainer* t, const float* d_g_rgb, const float* d_g_albedo, const float* d_g_sky,
                                    const float* d_g_pe, const float* d_rho_prior, const float* d_trust, const float* d_g_rgb_merged,
                                    const float* d_g_albedo_merged, void* stream) {
    if (d_rho_prior && !d_trust) return snerf_set_error(SNERF_E_INVALID, "snerf_trainer_backward_image_dt: d_trust is NULL");
    return backward_image(t, d_g_rgb, d_g_albedo, d_g_sky, d_g_pe, d_rho_prior, 1.f, d_trust, d_g_rgb_merged, d_g_albedo_merged, stream);
}

// Seam B1 in train mode (T_NeRF.forward called on points with autograd, T_NeRF_net_v2.py:75-105): backward from gradients with
// respect to the per-sample network outputs instead of the composited colours.  Any of the pointers may be NULL (= zero).
int snerf_trainer_backward_points(snerf_trainer* t, const float* d_g_rho, const float* d_g_col, const float* d_g_solar_vis,
                                  const float* d_g_sky, const float* d_g_classes, void* stream) {
    if (!t || !t->ws) return snerf_set_error(SNERF_E_STATE, "trainer not bound");
    CtxGuard ctx(t);
    hipStream_t st = (hipStream_t)stream;
    snerf_trainer::Pass& P = t->img;
    const int64_t R = P.R, N = P.N;
    auto put = [&](float* dst, const float* src, int64_t n) -> hipError_t {
        return src ? snerf_copy_async(dst, src, n * sizeof(float), st) : snerf_zero_async(dst, 0, n * sizeof(float), st);
    };
    HIPCK(put(t->d_rho, d_g_rho, N));
    HIPCK(put(t->d_col, d_g_col, N * 3));
    HIPCK(put(t->d_sky, d_g_sky, R * 3));
    if (d_g_solar_vis) HIPCK(put(t->d_sv_raw, d_g_solar_vis, N));
    return network_backward_image(t, d_g_solar_vis != nullptr, d_g_classes, st);
}

// The colour head (3 outputs) and the density head (1) read the same input X1 and their output gradients sit side by side in d_head [N, 4]: ONE stream over X1
// for both weight gradients and ONE read-modify-write of dX1 for both input gradients (K = 4, row 3 of the weights / of the result at the density head's
// address) instead of two of each.  Returns false - nothing launched - where the streams do not apply (SNERF_FUSED_HEADS=0, small batches, exact-fp32 mode).
static bool heads_bwd_fused(snerf_trainer* t, const Act& X1, int64_t N, const ActBelow* below, bool* fused_below, hipStream_t st) {
    static const int on = [] { const char* e = getenv("SNERF_FUSED_HEADS"); return (e && e[0] == '0') ? 0 : 1; }();
    const LayerP &Lc = t->layers[L_COL], &Ls_ = t->layers[L_SIG];
    if (fused_below) *fused_below = false;
    if (!on || t->gemm_mode != 1 || N < 1024 || Lc.n_out != 3 || Ls_.n_out != 1 || Lc.n_in != Ls_.n_in) return false;
    ThinWgradArgs w{};
    w.D = t->d_head; w.ldd = 4; w.In = X1.p; w.ldi = X1.ld; w.M = N; w.K = 4; w.N = Lc.n_in; w.alpha = 1.f;
    w.dW = t->grads + Lc.w; w.ldw = Lc.n_in; w.dW3 = t->grads + Ls_.w; w.tab = X1.tab; w.tab_cols = X1.tab ? X1.cols : 0; w.tab_stride = w.tab_cols;
    ThinDgradArgs a{};
    a.D = t->d_head; a.ldd = 4; a.W = t->params + Lc.w; a.W3 = t->params + Ls_.w; a.ldw = Lc.n_in; a.C = t->dX1.p; a.ldc = t->W2; a.M = N; a.K = 4; a.N = Lc.n_in;
    a.accumulate = 1; a.alpha = 1.f;
    const bool act = below && below->tab && below->L->n_out == Lc.n_in;
    if (act) {
        a.ez = below->Z.p; a.eld = below->Z.ld; a.etab = below->tab; a.stats = t->bn_stats;
        if (below->L->bn) { a.emu = below->bnslot + 2 * t->W; a.eistd = below->bnslot + 3 * t->W; }
    }
    if (!thin_wgrad_ok(w) || !thin_dgrad_ok(a)) return false;
    if (launch_thin_wgrad(w, st) != hipSuccess) return false;
    if (launch_colsum(t->d_head, N, 3, 4, 1.f, t->grads + Lc.b, st) != hipSuccess) return false;
    if (launch_colsum(t->d_head + 3, N, 1, 4, 1.f, t->grads + Ls_.b, st) != hipSuccess) return false;
    if (launch_thin_dgrad(a, st) != hipSuccess) return false;
    if (act && fused_below) *fused_below = true;
    return true;
}

static int network_backward_image(snerf_trainer* t, bool classic, const float* d_g_classes, hipStream_t st) {
    snerf_trainer::Pass& P = t->img;
    const int W = t->W, W2 = t->W2, W4 = t->W4, C = t->C, S = t->S;
    const int64_t R = P.R, N = P.N;
    auto& Ls = t->layers;
    if (d_g_classes) HIPCK(snerf_copy_async(t->d_cls, d_g_classes, R * C * sizeof(float), st));
    else HIPCK(snerf_zero_async(t->d_cls, 0, R * C * sizeof(float), st));
    HIPCK(snerf_zero_async(t->bn_stats, 0, (size_t)(4 * W + 2) * sizeof(float), st));      // column-sum scratch: zero at the start of the pass
    PointOutArgs po{};
    po.n = N; po.n_samples = S; po.C = C; po.head = P.head.p; po.adj = P.adj.p; po.cls = P.cls; po.col = P.col; po.sv = P.sv;
    po.d_rho = t->d_rho; po.d_col = t->d_col; po.d_head = t->d_head; po.d_adj = t->d_adj; po.d_cls = t->d_cls;
    if (classic) { po.d_sv = t->d_sv_raw; po.d_sv_raw = t->d_sv_raw; }
    HIPCK(launch_point_out(po, true, st));
    const Act X1 = P.Hc[8];
    // adjust branch
    // `pre`: did the dgrad that produced this layer's output gradient already apply its activation backward (ActBelow)?
    auto tab_of = [&](int slot) { return P.aol ? P.tabs + (int64_t)slot * 2 * W : (const float*)nullptr; };   // slots as in forward_pass
    bool pre = false;
    {
        const ActBelow b3{&Ls[L_A3], P.Za[2], tab_of(11), nullptr}, b2{&Ls[L_A2], P.Za[1], tab_of(10), nullptr}, b1{&Ls[L_A1], P.Za[0], tab_of(9), nullptr};
        RC(plain_bwd(t, Ls[L_AC], t->d_adj, 3 * C, P.Hac[2], N, t->dA.p, W, false, st, &b3, &pre));
        RC(sine_bwd(t, Ls[L_A3], Act{t->dA.p, W}, P.Za[2], P.Hac[1], N, nullptr, Act{t->dB.p, W}, W, false, st, pre, &b2, &pre));
        RC(sine_bwd(t, Ls[L_A2], Act{t->dB.p, W}, P.Za[1], P.Hac[0], N, nullptr, Act{t->dA.p, W}, W, false, st, pre, &b1, &pre));
        RC(sine_bwd(t, Ls[L_A1], Act{t->dA.p, W}, P.Za[0], X1, N, nullptr, Act{t->dX1.p, W2}, W2, false, st, pre));
    }
    // sigma / colour heads
    // dL/dX1 is summed from the adjust branch, the two heads and (classic solar) the solar branch: the LAST of these dgrads
    // also applies fc9's activation backward in its epilogue
    const ActBelow b9{&Ls[L_FC9], P.Zc8, tab_of(8), P.bn + 7 * 4 * W};
    bool pre9 = false;
    if (!heads_bwd_fused(t, X1, N, classic ? nullptr : &b9, classic ? nullptr : &pre9, st)) {
        RC(plain_bwd(t, Ls[L_COL], t->d_head, 4, X1, N, t->dX1.p, W2, true, st));
        RC(plain_bwd(t, Ls[L_SIG], t->d_head + 3, 4, X1, N, t->dX1.p, W2, true, st, classic ? nullptr : &b9, classic ? nullptr : &pre9));
    }
    if (classic) {      // the solar-visibility branch carries gradient from the image (G_NeRF.py:100-108), on into X1
        const ActBelow s3{&Ls[L_S3], P.Zs[2], tab_of(14), nullptr}, s2{&Ls[L_S2], P.Zs[1], tab_of(13), nullptr}, s1{&Ls[L_S1], P.Zs[0], tab_of(12), nullptr};
        RC(plain_bwd(t, Ls[L_S4], t->d_sv_raw, 1, P.Hsc[2], N, t->dA.p, W2, false, st, &s3, &pre));
        RC(sine_bwd(t, Ls[L_S3], Act{t->dA.p, W2}, P.Zs[2], P.Hsc[1], N, nullptr, Act{t->dB.p, W2}, W2, false, st, pre, &s2, &pre));
        RC(sine_bwd(t, Ls[L_S2], Act{t->dB.p, W2}, P.Zs[1], P.Hsc[0], N, nullptr, Act{t->dA.p, W2}, W2, false, st, pre, &s1, &pre));
        RC(sine_bwd(t, Ls[L_S1], Act{t->dA.p, W2}, P.Zs[0], P.In_s1c, N, nullptr, Act{t->dX1.p, W2}, W2, true, st, pre, &b9, &pre9));
    }
    // trunk
    float* cur = t->dA.p;
    float* nxt = t->dB.p;
    auto below_of = [&](int l) {        // trunk layer l (fc{l+1}) as the consumer of a dgrad's output
        return ActBelow{&Ls[L_FC1 + l], l == 3 ? P.Zc3 : (l == 8 ? P.Zc8 : P.Z[l]), tab_of(l), l >= 1 ? P.bn + (l - 1) * 4 * W : nullptr};
    };
    {
        const ActBelow b = below_of(7);
        RC(sine_bwd(t, Ls[L_FC9], Act{t->dX1.p, W2}, P.Zc8, P.Hc[7], N, P.bn + 7 * 4 * W, Act{cur, W}, W, false, st, pre9, &b, &pre));
    }
    for (int l = 7; l >= 1; --l) {
        const Act In = l == 4 ? P.In5c : P.Hc[l - 1];
        const ActBelow b = below_of(l - 1);
        RC(sine_bwd(t, Ls[L_FC1 + l], Act{cur, W}, l == 3 ? P.Zc3 : P.Z[l], In, N, P.bn + (l - 1) * 4 * W, Act{nxt, W}, W, false, st, pre, &b, &pre));
        float* tmp = cur; cur = nxt; nxt = tmp;
    }
    RC(sine_bwd(t, Ls[L_FC1], Act{cur, W}, P.Z[0], Act{P.E.p, 64}, N, nullptr, Act{nullptr, 0}, 0, false, st, pre));
    // time -> class branch (per ray)
    HIPCK(launch_softmax_bwd(P.cls, t->d_cls, t->rayA, R, C, st));
    RC(plain_bwd(t, Ls[L_CL], t->rayA, C, Act{P.Ht2, W}, R, t->rayB, W, false, st));
    RC(sine_bwd(t, Ls[L_T2], Act{t->rayB, W}, Act{P.Zt2, W}, Act{P.Ht1, W}, R, nullptr, Act{t->rayA, W}, W, false, st));
    RC(sine_bwd(t, Ls[L_T1], Act{t->rayA, W}, Act{P.Zt1, W}, Act{P.pe_time, 12}, R, nullptr, Act{nullptr, 0}, 0, false, st));
    // sky colour head (per ray)
    HIPCK(launch_sigmoid_bwd(P.sky, t->d_sky, t->rayB, R * 3, st));
    RC(plain_bwd(t, Ls[L_K2], t->rayB, 3, Act{P.Hk1, W4}, R, t->rayA, W4, false, st));
    RC(sine_bwd(t, Ls[L_K1], Act{t->rayA, W4}, Act{P.Zk1, W4}, Act{P.pe_sun, 28}, R, nullptr, Act{nullptr, 0}, 0, false, st));
    return SNERF_OK;
}

int snerf_trainer_forward_solar(snerf_trainer* t, int64_t n_rays, int n_samples, const float* d_top, const float* d_bot,
                                const float* d_tvals, const float* d_sun, int train_bn, float* d_solar_vis, float* d_pv,
                                float* d_pe, float* d_sky_raw, float* d_rho, float* d_points, float* d_delta, void* stream) {
    RC(check_bound(t, n_rays, n_samples, true));
    if (!d_top || !d_bot || !d_tvals || !d_sun) return snerf_set_error(SNERF_E_INVALID, "snerf_trainer_forward_solar: bad argument");
    CtxGuard ctx(t);
    hipStream_t st = (hipStream_t)stream;
    snerf_trainer::Pass& P = t->sol;
    RC(forward_pass(t, P, true, n_rays, n_samples, d_top, d_bot, d_tvals, d_sun, nullptr, train_bn != 0, st));
    snerf_composite_out co{};
    co.d_pv = d_pv; co.d_pe = d_pe; co.d_delta = d_delta;
    // col is not produced by the sun-ray pass: composite only needs rho (col/sv operands are dummies of the right size)
    RC(snerf_composite_rays(n_rays, n_samples, d_top, d_bot, d_tvals, P.rho, t->d_col, P.sv, P.sky, 0, nullptr, 1.f, &co, stream));
    const int64_t N = n_rays * n_samples;
    if (d_solar_vis) HIPCK(snerf_copy_async(d_solar_vis, P.sv, N * 4, st));
    if (d_sky_raw) HIPCK(snerf_copy_async(d_sky_raw, P.sky_raw, n_rays * 12, st));
    if (d_rho) HIPCK(snerf_copy_async(d_rho, P.rho, N * 4, st));
    if (d_points) HIPCK(snerf_copy_async(d_points, P.pts, N * 12, st));
    return SNERF_OK;
}

int snerf_trainer_backward_solar(snerf_trainer* t, const float* d_g_solar_vis, void* stream) {
    if (!t || !t->ws || !d_g_solar_vis) return snerf_set_error(SNERF_E_STATE, "trainer not bound / NULL gradient");
    CtxGuard ctx(t);
    hipStream_t st = (hipStream_t)stream;
    snerf_trainer::Pass& P = t->sol;
    const int W2 = t->W2;
    const int64_t N = P.N;
    auto& Ls = t->layers;
    HIPCK(snerf_zero_async(t->bn_stats, 0, (size_t)(4 * t->W + 2) * sizeof(float), st));
    PointOutArgs po{};
    po.n = N; po.n_samples = t->S; po.C = t->C; po.head = P.head.p; po.sv = P.sv; po.d_sv = d_g_solar_vis; po.d_sv_raw = t->d_sv_raw;
    HIPCK(launch_point_out(po, true, st));
    auto tab_of = [&](int slot) { return P.aol ? P.tabs + (int64_t)slot * 2 * t->W : (const float*)nullptr; };
    const ActBelow s3{&Ls[L_S3], P.Zs[2], tab_of(14), nullptr}, s2{&Ls[L_S2], P.Zs[1], tab_of(13), nullptr}, s1{&Ls[L_S1], P.Zs[0], tab_of(12), nullptr};
    bool pre = false;
    RC(plain_bwd(t, Ls[L_S4], t->d_sv_raw, 1, P.Hsc[2], N, t->dA.p, W2, false, st, &s3, &pre));
    RC(sine_bwd(t, Ls[L_S3], Act{t->dA.p, W2}, P.Zs[2], P.Hsc[1], N, nullptr, Act{t->dB.p, W2}, W2, false, st, pre, &s2, &pre));
    RC(sine_bwd(t, Ls[L_S2], Act{t->dB.p, W2}, P.Zs[1], P.Hsc[0], N, nullptr, Act{t->dA.p, W2}, W2, false, st, pre, &s1, &pre));
    RC(sine_bwd(t, Ls[L_S1], Act{t->dA.p, W2}, P.Zs[0], P.In_s1c, N, nullptr, Act{nullptr, 0}, 0, false, st, pre));
    return SNERF_OK;
}

// Introspection for tests: copy an internal buffer of the last image-pass backward to the host (synchronous).
int snerf_trainer_debug_read(snerf_trainer* t, const char* name, float* host_out, int64_t n_floats) {
    if (!t || !t->ws || !name || !host_out) return snerf_set_error(SNERF_E_INVALID, "snerf_trainer_debug_read: bad argument");
    const std::string k = name;
    const float* src = k == "d_rho" ? t->d_rho : k == "d_col" ? t->d_col : k == "d_head" ? t->d_head : k == "d_adj" ? t->d_adj
                     : k == "d_sky" ? t->d_sky : k == "d_cls" ? t->d_cls : k == "rho" ? t->img.rho : k == "col" ? t->img.col
                     : k == "sv" ? t->img.sv : k == "sky" ? t->img.sky : k == "head" ? t->img.head.p : nullptr;
    if (!src) return snerf_set_error(SNERF_E_INVALID, "snerf_trainer_debug_read: unknown buffer " + k);
    HIPCK(hipDeviceSynchronize());
    HIPCK(hipMemcpy(host_out, src, n_floats * sizeof(float), hipMemcpyDeviceToHost));
    return SNERF_OK;
}

// ---- the three products of a Linear layer as stand-alone calls (building blocks of the training engine; tests)
size_t snerf_linear_scratch_bytes(int n_out, int n_in) {
    const int64_t d = n_out > n_in ? n_out : n_in;
    return (size_t)(((d + 31) / 32) * ((d + 15) / 16) * 2048 + 256);
}

static int linear_mode(int precision, int64_t M, int K, int N) {      // 1 = bf16x3 kernel applicable and requested
    return (precision == 1 && M >= 1 && K >= 1 && N >= 1 && gemm_rows_group_tiles((K + 15) / 16) > 0) ? 1 : 0;
}

int snerf_linear_forward(int64_t n_points, int n_in, int n_out, const float* d_in, int64_t ld_in, const float* d_weight,
                         const float* d_bias, float alpha, float* d_out, int64_t ld_out, double* d_stats, int precision,
                         void* d_scratch, size_t scratch_bytes, const float* d_act_tab, int act_cols, void* stream) {
    if (n_points < 0 || n_in < 1 || n_out < 1) return snerf_set_error(SNERF_E_INVALID, "snerf_linear_forward: bad shape");
    if (n_points == 0) return SNERF_OK;
    if (!d_in || !d_weight || !d_out || ld_in < n_in || ld_out < n_out) return snerf_set_error(SNERF_E_INVALID, "snerf_linear_forward: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (precision == 1 && n_out <= 4 && n_points >= 1024 && d_bias && !d_stats) {      // a thin head: the stream the training engine uses for the colour + density heads
        ThinFwdArgs f{};
        f.In = d_in; f.ldi = ld_in; f.M = n_points; f.K = n_out; f.N = n_in; f.alpha = alpha; f.Out = d_out; f.ldo = ld_out;
        f.W = d_weight; f.ldw = n_in; f.bias = d_bias; f.tab = d_act_tab; f.tab_cols = d_act_tab ? act_cols : 0; f.tab_stride = f.tab_cols;
        static const int fused = [] { const char* e = getenv("SNERF_FUSED_HEADS"); return (e && e[0] == '0') ? 0 : 1; }();
        if (fused && thin_fwd_ok(f)) { HIPCK(launch_thin_fwd(f, st)); return SNERF_OK; }
    }
    if (linear_mode(precision, n_points, n_in, n_out)) {
        if (!d_scratch || scratch_bytes < snerf_linear_scratch_bytes(n_out, n_in)) return snerf_set_error(SNERF_E_INVALID, "snerf_linear_forward: scratch too small");
        GemmX x{};
        x.n_tiles = (n_out + 31) / 32; x.ksteps = (n_in + 15) / 16;
        x.W = d_weight; x.w_rows = n_out; x.w_cols = n_in; x.w_transpose = 0;
        x.A = d_in; x.frag = (const uint16_t*)d_scratch; x.C = d_out; x.M = n_points; x.N = n_out; x.K = n_in; x.lda = ld_in; x.ldc = ld_out;
        x.alpha = alpha; x.bias = d_bias; x.stats = d_stats; x.accumulate = 0;
        if (d_act_tab) {
            if (act_cols < 8 || act_cols % 8 || act_cols > n_in) return snerf_set_error(SNERF_E_INVALID, "snerf_linear_forward: act_cols must be a multiple of 8 within n_in");
            x.act_tab = d_act_tab; x.act_cols = act_cols;
        }
        HIPCK(launch_gemm_bf16x3(x, st));
        return SNERF_OK;
    }
    if (d_stats || d_act_tab) return snerf_set_error(SNERF_E_INVALID, "snerf_linear_forward: column statistics / activation on load need the bf16x3 path");
    GemmArgs g{};
    g.A = d_in; g.B = d_weight; g.C = d_out; g.M = n_points; g.N = n_out; g.K = n_in;
    g.sAm = ld_in; g.sAk = 1; g.sBk = 1; g.sBn = n_in; g.ldc = ld_out;
    g.alpha = alpha; g.bias = d_bias; g.colsum = nullptr; g.flags = 0; g.splitk = 1;
    HIPCK(launch_gemm(g, st));
    return SNERF_OK;
}

int snerf_linear_dgrad(int64_t n_points, int n_in, int n_out, const float* d_grad_out, int64_t ld_go, const float* d_weight,
                       int n_cols, float alpha, int accumulate, float* d_grad_in, int64_t ld_gi, int precision, void* d_scratch,
                       size_t scratch_bytes, const float* d_below_z, int64_t ld_below_z, const float* d_below_tab,
                       const float* d_below_mu, const float* d_below_istd, double* d_sums, void* stream) {
    if (n_points < 0 || n_in < 1 || n_out < 1 || n_cols < 1 || n_cols > n_in) return snerf_set_error(SNERF_E_INVALID, "snerf_linear_dgrad: bad shape");
    if (n_points == 0) return SNERF_OK;
    if (!d_grad_out || !d_weight || !d_grad_in || ld_go < n_out || ld_gi < n_cols) return snerf_set_error(SNERF_E_INVALID, "snerf_linear_dgrad: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (linear_mode(precision, n_points, n_out, n_cols)) {
        if (!d_scratch || scratch_bytes < snerf_linear_scratch_bytes(n_out, n_in)) return snerf_set_error(SNERF_E_INVALID, "snerf_linear_dgrad: scratch too small");
        GemmX x{};
        x.n_tiles = (n_cols + 31) / 32; x.ksteps = (n_out + 15) / 16;
        x.W = d_weight; x.w_rows = n_out; x.w_cols = n_in; x.w_transpose = 1;
        x.A = d_grad_out; x.frag = (const uint16_t*)d_scratch; x.C = d_grad_in; x.M = n_points; x.N = n_cols; x.K = n_out; x.lda = ld_go; x.ldc = ld_gi;
        x.alpha = alpha; x.bias = nullptr; x.stats = nullptr; x.accumulate = accumulate ? 1 : 0;
        if (d_below_z) {
            if (!d_below_tab || !d_sums || ld_below_z < n_cols || (!d_below_mu) != (!d_below_istd))
                return snerf_set_error(SNERF_E_INVALID, "snerf_linear_dgrad: the activation-backward epilogue needs table and sums");
            x.ez = d_below_z; x.eld = ld_below_z; x.etab = d_below_tab; x.emu = d_below_mu; x.eistd = d_below_istd; x.stats = d_sums;
        }
        HIPCK(launch_gemm_bf16x3(x, st));
        return SNERF_OK;
    }
    if (d_below_z) return snerf_set_error(SNERF_E_INVALID, "snerf_linear_dgrad: the activation-backward epilogue needs the bf16x3 path");
    GemmArgs g{};
    g.A = d_grad_out; g.B = d_weight; g.C = d_grad_in; g.M = n_points; g.N = n_cols; g.K = n_out;
    g.sAm = ld_go; g.sAk = 1; g.sBk = n_in; g.sBn = 1; g.ldc = ld_gi;
    g.alpha = alpha; g.bias = nullptr; g.colsum = nullptr; g.flags = accumulate ? GEMM_ACCUM : 0; g.splitk = 1;
    HIPCK(launch_gemm(g, st));
    return SNERF_OK;
}

int snerf_linear_wgrad(int64_t n_points, int n_in, int n_out, const float* d_grad_out, int64_t ld_go, const float* d_in,
                       int64_t ld_in, float alpha, float* d_grad_weight, int precision, const float* d_act_tab, int act_cols,
                       void* stream) {
    if (n_points < 0 || n_in < 1 || n_out < 1) return snerf_set_error(SNERF_E_INVALID, "snerf_linear_wgrad: bad shape");
    if (n_points == 0) return SNERF_OK;
    if (!d_grad_out || !d_in || !d_grad_weight || ld_go < n_out || ld_in < n_in) return snerf_set_error(SNERF_E_INVALID, "snerf_linear_wgrad: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (d_act_tab && (precision != 1 || act_cols < 1 || act_cols > n_in))
        return snerf_set_error(SNERF_E_INVALID, "snerf_linear_wgrad: activation on load needs the bf16x3 path and 1 <= act_cols <= n_in");
    if (precision == 1) {
        if (n_out <= 4 && n_points >= 1024) {      // the thin heads' stream, as the training engine routes them (linear_wgrad)
            ThinWgradArgs a{};
            a.D = d_grad_out; a.ldd = ld_go; a.In = d_in; a.ldi = ld_in; a.M = n_points; a.K = n_out; a.N = n_in; a.alpha = alpha;
            a.dW = d_grad_weight; a.ldw = n_in; a.tab = d_act_tab; a.tab_cols = d_act_tab ? act_cols : 0; a.tab_stride = a.tab_cols;
            static const int thin = [] { const char* e = getenv("SNERF_THIN_WGRAD"); return (e && e[0] == '0') ? 0 : 1; }();
            if (thin && thin_wgrad_ok(a)) { HIPCK(launch_thin_wgrad(a, st)); return SNERF_OK; }
        }
        HIPCK(launch_wgrad_bf16x3(const_cast<float*>(d_grad_out), ld_go, d_in, ld_in, n_points, n_out, n_in, alpha, d_grad_weight, n_in, st, d_act_tab, act_cols));      // read-only without the BatchNorm option
        return SNERF_OK;
    }
    GemmArgs g{};
    g.A = d_grad_out; g.B = d_in; g.C = d_grad_weight; g.M = n_out; g.N = n_in; g.K = n_points;
    g.sAm = 1; g.sAk = ld_go; g.sBk = ld_in; g.sBn = 1; g.ldc = n_in;
    g.alpha = alpha; g.bias = nullptr; g.colsum = nullptr; g.flags = GEMM_ATOMIC;
    int64_t sk = (n_points + 2047) / 2048;
    g.splitk = (int)(sk > 1024 ? 1024 : (sk < 1 ? 1 : sk));
    HIPCK(launch_gemm(g, st));
    return SNERF_OK;
}

int snerf_trainer_set_allreduce(snerf_trainer* t, snerf_allreduce_fn fn, void* user, int world_size) {
    if (!t) return snerf_set_error(SNERF_E_INVALID, "NULL trainer");
    if (fn && world_size < 1) return snerf_set_error(SNERF_E_INVALID, "snerf_trainer_set_allreduce: world_size must be >= 1");
    t->ar_fn = fn; t->ar_user = user; t->world = fn ? world_size : 1;
    return SNERF_OK;
}

int snerf_trainer_zero_grad(snerf_trainer* t, void* stream) {
    if (!t || !t->grads) return snerf_set_error(SNERF_E_STATE, "trainer not bound");
    HIPCK(snerf_zero_async(t->grads, 0, t->n_params * sizeof(float), (hipStream_t)stream));
    return SNERF_OK;
}

int snerf_trainer_adam_step(snerf_trainer* t, float lr, float beta1, float beta2, float eps, int step, void* stream) {
    if (!t || !t->params || !t->adam_m || !t->adam_v) return snerf_set_error(SNERF_E_STATE, "trainer not bound (Adam state)");
    if (step < 1) return snerf_set_error(SNERF_E_INVALID, "Adam step counts from 1");
    HIPCK(launch_adam(t->params, t->grads, t->adam_m, t->adam_v, t->n_params, lr, beta1, beta2, eps, step, (hipStream_t)stream));
    return SNERF_OK;
}

/* the same update with lr / betas / eps / bias corrections read from device memory: what a captured (hipGraph) training step launches */
int snerf_trainer_adam_step_dev(snerf_trainer* t, const float* d_hyper6, void* stream) {
    if (!t || !t->params || !t->adam_m || !t->adam_v) return snerf_set_error(SNERF_E_STATE, "trainer not bound (Adam state)");
    if (!d_hyper6) return snerf_set_error(SNERF_E_INVALID, "snerf_trainer_adam_step_dev: NULL hyper-parameter vector");
    HIPCK(launch_adam_dev(t->params, t->grads, t->adam_m, t->adam_v, t->n_params, d_hyper6, (hipStream_t)stream));
    return SNERF_OK;
}

/* The scalar terms of All_in_One_Eval.get_loss (Eval_Tools_2.py:340-420) for the default training configuration - MSE colour loss, solar rays
 * on, default solar model, no DSM prior - in two launches, and their gradients in one (see include/season_nerf_hip.h). */
size_t snerf_loss_scratch_bytes(void) { return 4 * sizeof(double) + 3 * 1024 * sizeof(unsigned long long); }      // sums + per-block minima (<= 1024 blocks)
int snerf_loss_scratch_init(void* d_scratch, void* stream) {
    if (!d_scratch) return snerf_set_error(SNERF_E_INVALID, "snerf_loss_scratch_init: NULL scratch");
    HIPCK(launch_loss_scratch_init(d_scratch, (hipStream_t)stream));
    return SNERF_OK;
}
static int loss_args(LossArgs* a, int64_t n_rays, int64_t n_solar_rays, int n_samples, const float* d_rgb, const float* d_gt, const float* d_albedo,
                     const float* d_sky, const float* d_solar_vis, const float* d_pv_exact, const float* d_pe, const float* d_albedo_min_global, int world) {
    if (n_rays < 1 || n_solar_rays < 1 || n_samples < 1 || world < 1 || !d_rgb || !d_gt || !d_albedo || !d_sky || !d_solar_vis || !d_pv_exact)
        return snerf_set_error(SNERF_E_INVALID, "snerf_loss_terms: bad argument");
    *a = LossArgs{n_rays, n_solar_rays, n_samples, d_rgb, d_gt, d_albedo, d_sky, d_solar_vis, d_pv_exact, d_pe, d_albedo_min_global, d_albedo_min_global ? world : 1};
    return SNERF_OK;
}
int snerf_loss_terms_forward(int64_t n_rays, int64_t n_solar_rays, int n_samples, const float* d_rgb, const float* d_gt, const float* d_albedo,
                             const float* d_sky, const float* d_solar_vis, const float* d_pv_exact, const float* d_pe, const float* d_albedo_min_global,
                             int world, void* d_scratch, float* d_vals5, float* d_min3, void* stream) {
    LossArgs a;
    RC(loss_args(&a, n_rays, n_solar_rays, n_samples, d_rgb, d_gt, d_albedo, d_sky, d_solar_vis, d_pv_exact, d_pe, d_albedo_min_global, world));
    if (!d_pe || !d_scratch || !d_vals5 || !d_min3) return snerf_set_error(SNERF_E_INVALID, "snerf_loss_terms_forward: bad argument");
    HIPCK(launch_loss_terms(a, d_scratch, d_vals5, d_min3, (hipStream_t)stream));
    return SNERF_OK;
}
int snerf_loss_terms_backward(int64_t n_rays, int64_t n_solar_rays, int n_samples, const float* d_rgb, const float* d_gt, const float* d_albedo,
                              const float* d_sky, const float* d_solar_vis, const float* d_pv_exact, const float* d_min3, int world, const float* d_g_vals5,
                              float* d_g_rgb, float* d_g_albedo, float* d_g_sky, float* d_g_solar_vis, void* stream) {
    LossArgs a;
    RC(loss_args(&a, n_rays, n_solar_rays, n_samples, d_rgb, d_gt, d_albedo, d_sky, d_solar_vis, d_pv_exact, nullptr, nullptr, 1));
    (void)world;      // the value's denominator only: the gradient is per local ray count (see loss_bwd_kernel)
    if (!d_min3 || !d_g_vals5 || !d_g_rgb || !d_g_albedo || !d_g_sky || !d_g_solar_vis) return snerf_set_error(SNERF_E_INVALID, "snerf_loss_terms_backward: bad argument");
    HIPCK(launch_loss_terms_bwd(a, d_g_vals5, d_min3, d_g_rgb, d_g_albedo, d_g_sky, d_g_solar_vis, (hipStream_t)stream));
    return SNERF_OK;
}

/* the same update on caller-owned arenas (no trainer object): what season_nerf::fused_adam_ binds */
int snerf_adam_step(float* d_params, const float* d_grads, float* d_m, float* d_v, int64_t n, float lr, float beta1, float beta2, float eps,
                    int step, void* stream) {
    if (n < 0 || (n && (!d_params || !d_grads || !d_m || !d_v))) return snerf_set_error(SNERF_E_INVALID, "snerf_adam_step: bad argument");
    if (step < 1) return snerf_set_error(SNERF_E_INVALID, "Adam step counts from 1");
    if (n == 0) return SNERF_OK;
    HIPCK(launch_adam(d_params, d_grads, d_m, d_v, n, lr, beta1, beta2, eps, step, (hipStream_t)stream));
    return SNERF_OK;
}

}  // extern "C"
