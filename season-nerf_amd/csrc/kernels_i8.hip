// gfx950 (MI355X / CDNA4): the fused field network on the int8 matrix pipe ("i8x3", program.h FMT_I8).
//
//  mlp_i8_kernel<W, VARIANT>   same register-resident chain as mlp_kernel<PROG_FIELD, W, VARIANT> (kernels.hip), but every
//                              operand is 16-bit fixed point carried as two signed int8 digits on v_mfma_i32_32x32x32_i8:
//     * per 32-feature k-step three MFMAs (T a -> M;  T b, L a -> X; the L b term is dropped) instead of six bf16 ones
//       (three split products for each of the two 16-slot halves): half the matrix-pipe time, exact integer accumulation;
//     * an activation costs 2 bytes of register state instead of 4 (bf16 hi + lo), a weight 2 bytes of LDS / L2 traffic
//       instead of 4, and one 32x32 output block is exactly one k-step of the next layer;
//     * epilogue per element: (M << 8) + X, v_cvt_f32_i32, fma with the row's scale and bias (both fold BatchNorm, the
//       factor 30, 1/(2 pi) and the +128 digit offset, pack.cpp), v_sin_f32, then v_cvt_pknorm_i16_f32 (two elements per
//       instruction) and two v_perm_b32 + one v_xor per four elements to split the int16 into the two digit streams.
//  Accuracy: 16-bit fixed point on both operands: RGB within ~2e-5 relative of the fp32 reference (bf16x3: ~3e-6), density
//  and the other per-sample outputs within ~1e-4 (tools/numerics_i8.py, tests/test_gpu_precision.py).  Inputs must lie in
//  [-1,1] (sample positions inside the scene cube, unit sun vectors): v_cvt_pknorm saturates.
#include "mlp_device.h"

namespace snerf {

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;
typedef __attribute__((ext_vector_type(2))) short i16x2;
typedef const __attribute__((address_space(3))) i32x4 lds_ci32x4;

struct Frag8 {          // B operand of one 32-slot k-step: 16 high digits + 16 low digits of this lane's point
    i32x4 hi, lo;
};
struct Acc8 {           // M = sum T a (weight 2^16), X = sum (T b + L a) (weight 2^8)
    i32x16 M, X;
};
struct Tab8 {           // per-row scale and bias of one 32-row block, accumulator order
    f32x16 sc, bi;
};

// four values in [-1,1] -> their four high digits and four low digits (byte t of each dword = value t)
__device__ __forceinline__ void digits4(float v0, float v1, float v2, float v3, int& hi, int& lo) {
    const uint32_t p0 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pknorm_i16(v0, v1));
    const uint32_t p1 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pknorm_i16(v2, v3));
    hi = (int)__builtin_amdgcn_perm(p1, p0, 0x07050301u);                 // signed high bytes of the four int16
    lo = (int)(__builtin_amdgcn_perm(p1, p0, 0x06040200u) ^ 0x80808080u); // low bytes - 128 (the +128 lives in the bias)
}
__device__ __forceinline__ void pack16(const float* v, Frag8& f) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int h, l;
        digits4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3], h, l);
        f.hi[q] = h;
        f.lo[q] = l;
    }
}

__device__ __forceinline__ void mfma_i8x3(const i32x4& aT, const i32x4& aL, const Frag8& b, Acc8& acc) {
    acc.M = __builtin_amdgcn_mfma_i32_32x32x32_i8(aT, b.hi, acc.M, 0, 0, 0);
    acc.X = __builtin_amdgcn_mfma_i32_32x32x32_i8(aT, b.lo, acc.X, 0, 0, 0);
    acc.X = __builtin_amdgcn_mfma_i32_32x32x32_i8(aL, b.hi, acc.X, 0, 0, 0);
}

__device__ __forceinline__ Tab8 load_tab(lds_cfloat* tab_l, int b, int h) {
    lds_cf32x4* tp = (lds_cf32x4*)(tab_l + (b * 2 + h) * 32);
    Tab8 t;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 a = tp[q], c = tp[4 + q];
        t.sc[4 * q] = a[0]; t.sc[4 * q + 1] = a[1]; t.sc[4 * q + 2] = a[2]; t.sc[4 * q + 3] = a[3];
        t.bi[4 * q] = c[0]; t.bi[4 * q + 1] = c[1]; t.bi[4 * q + 2] = c[2]; t.bi[4 * q + 3] = c[3];
    }
    return t;
}

// pre-activation (revolutions for sine layers) of accumulator element i
__device__ __forceinline__ float preact(const Acc8& acc, const Tab8& t, int i) {
    const int m = (int)(((uint32_t)acc.M[i] << 8) + (uint32_t)acc.X[i]);
    return __builtin_fmaf((float)m, t.sc[i], t.bi[i]);
}
// epilogue pieces of a 32x32 block: A(e) = elements 2e, 2e+1 through the sine; Q(g) = digits of elements 4g..4g+3
__device__ __forceinline__ void epi_A(const Acc8& acc, const Tab8& t, int e, float* ev) {
#if defined(SNERF_ABLATE) && (ABL & 8)      // timing-only: no transcendental
    ev[2 * e] = preact(acc, t, 2 * e) * 1e-9f;
    ev[2 * e + 1] = preact(acc, t, 2 * e + 1) * 1e-9f;
#else
    ev[2 * e] = sin2pi(preact(acc, t, 2 * e));
    ev[2 * e + 1] = sin2pi(preact(acc, t, 2 * e + 1));
#endif
}
__device__ __forceinline__ void epi_Q(int g, const float* ev, Frag8& o) {
    int h, l;
    digits4(ev[4 * g], ev[4 * g + 1], ev[4 * g + 2], ev[4 * g + 3], h, l);
    o.hi[g] = h;
    o.lo[g] = l;
}

#define SG_VALU 0x002
#define SG_MFMA 0x008
#define SG_DSREAD 0x100
#define SG_TRANS 0x400

#ifndef SNERF_PF8
#define SNERF_PF8 3
#endif
constexpr int PF8 = SNERF_PF8;

// One fused layer: out^T[n x 32 pts] = act(W[n x k] in^T[k x 32 pts]), digits in registers.  Same software pipeline as
// run_layer (kernels.hip): weight fragments PF8 pairs ahead, ring step PF8 pairs before a chunk's first MFMA, accumulators
// ping-pong between blocks and the epilogue of block b-1 is spread over block b's k-steps: element pair e runs its sine
// at k-step sA(e) = 1 + e (KS-2) / 8, the digit split of quad g one step after its second pair.
template <int NB, int KS0, int KS1, bool SIN, int D>
__device__ __forceinline__ void run_layer8(Ring& rg, const uint8_t* stream, uint32_t stream_bytes, lds_char* lds, lds_cfloat* tab_l,
                                           const Frag8* in0, const Frag8* in1, Frag8* out, f32x16* raw, int wave, int lane) {
    constexpr int KS = KS0 + KS1, NP = NB * KS;
    constexpr bool PIPE = KS >= 4;
    const int h = lane >> 5;
    i32x4 fT[PF8], fL[PF8];
#pragma unroll
    for (int q = 0; q < PF8; ++q) {
        if (q < NP) {
            if (q % kChunkPairs == 0) ring_step<D>(rg, stream, stream_bytes, lds, wave, lane);
            lds_char* ap = lds + rg.cur + (q % kChunkPairs) * kPairBytes + lane * 16;
            fT[q] = *(lds_ci32x4*)ap;
            fL[q] = *(lds_ci32x4*)(ap + kFragBytes);
        }
    }
    Acc8 accs[2];
    float ev[16];
    Tab8 tab;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        Acc8 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc.M[i] = 0; acc.X[i] = 0; }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int q = b * KS + s;
            const i32x4 aT = fT[q % PF8], aL = fL[q % PF8];
            if (q + PF8 < NP) {
                const int qn = q + PF8;
                if (qn % kChunkPairs == 0) ring_step<D>(rg, stream, stream_bytes, lds, wave, lane);
#if defined(SNERF_ABLATE) && (ABL & 2)     // timing-only: A fragments stay in registers, no LDS reads
                asm volatile("" : "+v"(fT[q % PF8]), "+v"(fL[q % PF8]));
#else
                lds_char* ap = lds + rg.cur + (qn % kChunkPairs) * kPairBytes + lane * 16;
                fT[q % PF8] = *(lds_ci32x4*)ap;
                fL[q % PF8] = *(lds_ci32x4*)(ap + kFragBytes);
#endif
            }
            mfma_i8x3(aT, aL, s < KS0 ? in0[s] : in1[s - KS0], acc);
            if (SIN && b > 0) {
                if (s == 0) tab = load_tab(tab_l, b - 1, h);
                if (PIPE) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int sA = 1 + (e * (KS - 2)) / 8;
                        if (s == sA) epi_A(accs[(b - 1) & 1], tab, e, ev);
                        if ((e & 1) && s == sA + 1) epi_Q(e >> 1, ev, out[b - 1]);
                    }
                } else if (s == 0) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) epi_A(accs[(b - 1) & 1], tab, e, ev);
#pragma unroll
                    for (int g = 0; g < 4; ++g) epi_Q(g, ev, out[b - 1]);
                }
            }
#ifndef SNERF_NO_SCHED_GROUPS
            // pin the interleave: per MFMA at most 1 LDS read, 1 transcendental (8 cycles) and 4 plain VALU (16) in its shadow
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
                __builtin_amdgcn_sched_group_barrier(SG_DSREAD, 1, 0);
                __builtin_amdgcn_sched_group_barrier(SG_TRANS, 1, 0);
                __builtin_amdgcn_sched_group_barrier(SG_VALU, 4, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        accs[b & 1] = acc;
    }
    tab = load_tab(tab_l, NB - 1, h);
    if (SIN) {
#pragma unroll
        for (int e = 0; e < 8; ++e) epi_A(accs[(NB - 1) & 1], tab, e, ev);
#pragma unroll
        for (int g = 0; g < 4; ++g) epi_Q(g, ev, out[NB - 1]);
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) (*raw)[i] = preact(accs[(NB - 1) & 1], tab, i);
    }
}

// PE(pos): 32 values per lane-half (program.h pepos_feature), two k-steps
__device__ __forceinline__ void make_pe_pos8(float x0, float x1, float x2, int h, Frag8* pe) {
    float v[32];
    const float xs[3] = {x0, x1, x2};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const PeArg a = pe_arg(xs[d]);
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const double scale = h ? (double)(1 << (5 + q)) : (double)(1 << q);
            pe_sincos(a, scale, v[10 * d + 2 * q], v[10 * d + 2 * q + 1]);
        }
    }
    v[30] = h ? x2 : x0;
    v[31] = h ? 0.f : x1;
    pack16(v, pe[0]);
    pack16(v + 16, pe[1]);
}
// PE(sun): 16 values per lane-half (pesun_feature), one k-step
__device__ __forceinline__ void make_pe_sun8(float x0, float x1, float x2, int h, Frag8* pe) {
    float v[16];
    const float xs[3] = {x0, x1, x2};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const PeArg a = pe_arg(xs[d]);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const double scale = h ? (double)(1 << (2 + q)) : (double)(1 << q);
            pe_sincos(a, scale, v[4 * d + 2 * q], v[4 * d + 2 * q + 1]);
        }
    }
    v[12] = h ? x2 : x0;
    v[13] = h ? 0.f : x1;
    v[14] = 0.f;
    v[15] = 0.f;
    pack16(v, pe[0]);
}

constexpr int ring_depth8(int W) { return W > 256 ? 6 : RING_D; }    // W = 512: the 54 KB table leaves room for 6 slots

template <int W, int VARIANT>
__global__ __launch_bounds__(256, 1) void mlp_i8_kernel(const MlpArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int C_MAX = kMaxClasses;
    constexpr int W2 = W / 2;
    constexpr int D = ring_depth8(W);
    lds_char* lds = (lds_char*)smem;
    __attribute__((address_space(3))) float* tab_lds = (__attribute__((address_space(3))) float*)(lds + D * kChunkBytes);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5;
    const int C = A.n_classes;

    for (int i = threadIdx.x; i < A.bias_floats; i += 256) tab_lds[i] = A.bias[i];

    Ring rg;
    rg.rd = 0;
    rg.cur = 0;
    rg.goff = 0;
    {
        uint32_t wr = 0;
#pragma unroll
        for (int c = 0; c < D - 2; ++c) {
            dma_chunk(A.stream, rg.goff, lds, wr, wave, lane);
            rg.goff += kChunkBytes;
            if (rg.goff >= A.stream_bytes) rg.goff = 0;
            wr += kChunkBytes;
        }
        rg.wr = wr;
    }
    __syncthreads();

    const int64_t n_tiles = (A.n + TILE_PTS - 1) / TILE_PTS;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t n = tile * TILE_PTS + wave * 32 + (lane & 31);
        const bool valid = n < A.n;
        const int64_t nc = valid ? n : A.n - 1;
        const int64_t g = nc / A.group_size;

        // ---- sample position (misc.py:234-247 fused): top*(1-t) + bot*t, two roundings + one add, no fma
        float x0, x1, x2;
        if (A.points) {
            x0 = A.points[nc * 3]; x1 = A.points[nc * 3 + 1]; x2 = A.points[nc * 3 + 2];
        } else {
            const int64_t r = nc / A.n_samples;
            const int s = (int)(nc - r * A.n_samples);
            const float t = A.tvals[s], omt = __fsub_rn(1.f, t);
            x0 = __fadd_rn(__fmul_rn(A.top[r * 3], omt), __fmul_rn(A.bot[r * 3], t));
            x1 = __fadd_rn(__fmul_rn(A.top[r * 3 + 1], omt), __fmul_rn(A.bot[r * 3 + 1], t));
            x2 = __fadd_rn(__fmul_rn(A.top[r * 3 + 2], omt), __fmul_rn(A.bot[r * 3 + 2], t));
        }
        // every per-tile input is loaded before the MFMA chain (a plain load inside it drains the LDS-DMA pipeline)
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        float pcls[C_MAX];
#pragma unroll
        for (int c = 0; c < C_MAX; ++c) pcls[c] = 0.f;
        if constexpr (VARIANT <= 1) { s0 = A.sun[g * 3]; s1 = A.sun[g * 3 + 1]; s2 = A.sun[g * 3 + 2]; }
        if constexpr (VARIANT == 0) {
            if (A.classes) {
#pragma unroll
                for (int c = 0; c < C_MAX; ++c) if (c < C) pcls[c] = A.classes[g * C + c];
            }
        }
        Frag8 pe[PEPOS_KS8];
        make_pe_pos8(x0, x1, x2, h, pe);

        constexpr int KW = W / 32, KW2 = W2 / 32;
        Frag8 hA[KW], hB[KW];
        f32x16 raw;
#define LAYER(L, NBv, K0, K1, SINv, IN0, IN1, OUT, RAW)                                                                 \
    run_layer8<NBv, K0, K1, SINv, D>(rg, A.stream, A.stream_bytes, lds, tab_lds + prog_table_start(PROG_FIELD, W, C_MAX, L), \
                                     IN0, IN1, OUT, RAW, wave, lane)
        // trunk (G_NeRF.py:80-91)
        LAYER(F_FC1, W / 32, PEPOS_KS8, 0, true, pe, nullptr, hA, nullptr);
        LAYER(F_FC2, W / 32, KW, 0, true, hA, nullptr, hB, nullptr);
        LAYER(F_FC3, W / 32, KW, 0, true, hB, nullptr, hA, nullptr);
        LAYER(F_FC4, W / 32, KW, 0, true, hA, nullptr, hB, nullptr);
        LAYER(F_FC5, W / 32, KW, PEPOS_KS8, true, hB, pe, hA, nullptr);
        LAYER(F_FC6, W / 32, KW, 0, true, hA, nullptr, hB, nullptr);
        LAYER(F_FC7, W / 32, KW, 0, true, hB, nullptr, hA, nullptr);
        LAYER(F_FC8, W / 32, KW, 0, true, hA, nullptr, hB, nullptr);
        Frag8 x1f[KW2];
        LAYER(F_FC9, W2 / 32, KW, 0, true, hB, nullptr, x1f, nullptr);
        // sigma / colour head (G_NeRF.py:93-98): regs 0..2 colour, 3 density (lane-half 0)
        LAYER(F_HEAD, 1, KW2, 0, false, x1f, nullptr, nullptr, &raw);
        const float col_r = raw[0], col_g = raw[1], col_b = raw[2], rho_raw = raw[3];
        float sv_raw = 0.f;
        float adj[3 * C_MAX];
#pragma unroll
        for (int i = 0; i < 3 * C_MAX; ++i) adj[i] = 0.f;
        if constexpr (VARIANT <= 1) {
            // solar visibility branch (G_NeRF.py:100-108)
            Frag8 ps[PESUN_KS8];
            make_pe_sun8(s0, s1, s2, h, ps);
            Frag8 sA[KW2], sB[KW2];
            LAYER(F_S1, W2 / 32, KW2, PESUN_KS8, true, x1f, ps, sA, nullptr);
            LAYER(F_S2, W2 / 32, KW2, 0, true, sA, nullptr, sB, nullptr);
            LAYER(F_S3, W2 / 32, KW2, 0, true, sB, nullptr, sA, nullptr);
            LAYER(F_S4, 1, KW2, 0, false, sA, nullptr, nullptr, &raw);
            sv_raw = raw[0];
        }
        if constexpr (VARIANT == 0) {
            // seasonal colour-adjust branch (T_NeRF_net_v2.py:83-87)
            LAYER(F_A1, W / 32, KW2, 0, true, x1f, nullptr, hA, nullptr);
            LAYER(F_A2, W / 32, KW, 0, true, hA, nullptr, hB, nullptr);
            LAYER(F_A3, W / 32, KW, 0, true, hB, nullptr, hA, nullptr);
            LAYER(F_AC, 1, KW, 0, false, hA, nullptr, nullptr, &raw);
#pragma unroll
            for (int i = 0; i < 3 * C_MAX; ++i) adj[i] = raw[i];
        }
#undef LAYER
        if (h == 0 && valid) store_field_outputs<VARIANT>(A.out, n, C, x0, x1, x2, col_r, col_g, col_b, rho_raw, sv_raw, adj, pcls);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
}

template <int W, int VARIANT>
static hipError_t launch_mlp_i8_t(const MlpArgs& a, int n_cu, hipStream_t st) {
    const int lds_bytes = ring_depth8(W) * kChunkBytes + a.bias_floats * 4;
    const int64_t n_tiles = (a.n + TILE_PTS - 1) / TILE_PTS;
    int grid = (int)(n_tiles < n_cu ? n_tiles : n_cu);
    if (grid < 1) grid = 1;
    auto k = mlp_i8_kernel<W, VARIANT>;
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds_bytes, st, a);
    return hipGetLastError();
}

hipError_t launch_mlp_i8(int W, int variant, const MlpArgs& a, int n_cu, hipStream_t st) {
#define CASE(Wv)                                                             \
    if (W == Wv) {                                                           \
        if (variant == 0) return launch_mlp_i8_t<Wv, 0>(a, n_cu, st);        \
        if (variant == 1) return launch_mlp_i8_t<Wv, 1>(a, n_cu, st);        \
        return launch_mlp_i8_t<Wv, 2>(a, n_cu, st);                          \
    }
    CASE(64)
    CASE(256)
#undef CASE
    return hipErrorInvalidValue;
}

// chunks consumed per tile by a variant of the int8 field program (the DMA stream is cyclic over exactly these)
int field_variant_chunks_i8(int W, int C, int variant) {
    const int last = variant == 0 ? (int)F_NUM : variant == 1 ? (int)F_A1 : (int)F_S1;
    return prog_chunk_start(PROG_FIELD, W, C, last, FMT_I8);
}

}  // namespace snerf
