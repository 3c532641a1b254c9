// Fused bf16x3 field kernel of the reference's DEFAULT width (main_lite.py:80, opt2.py:79: fc_units = 512): the chain of kernels.hip with the
// K dimension of every layer split over a PAIR of waves (round 6).
//
// Why a second structure.  At W = 512 the activations of 32 points are 2 x 256 registers per lane (input + output of a 512 -> 512 layer as bf16
// hi / lo fragments): more than a wave owns.  Here two waves (on two SIMDs) share 32 points.  Wave `a` (0 / 1) of a pair
//   * holds the K-half `a` of every hidden activation (features [256 a, 256 a + 256): 16 k-steps = 128 registers, as a W = 256 wave does),
//   * OWNS the output blocks a*NBH .. a*NBH + NBH - 1 of every layer (they are exactly its K-half of the next layer),
//   * multiplies its K-half against ALL output blocks: an "F" phase (a block the partner owns: the fp32 partial sums go to the partner through
//     4 KiB of LDS) and an "O" phase (a block it owns: the partner's partial sums are added in front of the bias / sin / split epilogue).
// Both waves walk F(i), O(i), i = 0 .. NBH-1 in step: while one forms the partial sums of block i for its partner, the partner forms the partial
// sums of block NBH + i for it.  The exchange needs no synchronisation of its own - it is ordered by the weight ring's barriers: a partial
// written in the first k-steps of an O phase is read by the partner in the first k-step of its NEXT F phase, >= 2 ring steps later, and
// overwritten one phase after that.  The encodings (PE(pos) of fc1 / fc5, PE(sun) of fc_solar_1) enter on the OWN blocks only, in full, so the
// first layer needs no exchange at all.  The raw heads (one block) are symmetric: both waves add the partner's half and both hold the result.
// Epilogue work (sin + split) exists in every second phase only: per MFMA half the vector work of the W = 256 kernel.
//
// Weight stream (pack.cpp pack_program_ks): the pairs of the canonical bf16 stream in the order each wave consumes them, a 16 KiB chunk =
// 4 pairs for parity 0 | 4 pairs for parity 1; every chunk is read by two waves (one per pair), i.e. the L2 -> LDS stream per point is twice
// the W = 256 kernel's (64 points share a chunk, not 128) - the price of the width; the ring (7 slots, 5 in flight) is the same, its refill spread
// over the k-steps (ks_ring below).
#include "mlp_bf16_device.h"

namespace snerf {

constexpr int KS_TILE_PTS = 64;                       // points per workgroup tile: 2 wave pairs x 32
constexpr int KS_PAR_PAIRS = 4;                       // pairs per parity and chunk
constexpr int KS_PAR_BYTES = KS_PAR_PAIRS * kPairBytes;   // 8 KiB
constexpr int KS_XBUF_BYTES = 4096;                   // one 32 x 32 fp32 block of partial sums
static_assert(kChunkPairs == 2 * KS_PAR_PAIRS, "a chunk holds 4 pairs per parity");

// LDS: ring | bias table | 32 zero floats (bias of the partner's half of a raw head) | 4 exchange buffers (pair, parity)
__host__ __device__ constexpr int ks_lds_bytes(int bias_floats) { return RING_BYTES + (bias_floats + 32) * 4 + 4 * KS_XBUF_BYTES + kVoteBytes; }

struct KsCtx {
    lds_char* lds;            // ring base
    uint32_t par_off;         // a * 8 KiB: this wave's half of every chunk
    lds_char* xw;             // exchange buffer this wave writes (read by the partner)
    lds_char* xr;             // ... the partner writes
};

// The ring of mlp_device.h with the refill SPREAD: a ring step (counted vmcnt + barrier) only names the chunk to fetch; its four 1 KiB pieces per
// wave are issued one per k-step over the four k-steps up to the next ring step.  An LDS-DMA instruction holds its wave's issue for ~30 cycles:
// four in a row behind every barrier cost this kernel - a ring step every 12 MFMAs, twice the W = 256 kernel's rate - 0.9 ms of 5.5 (the `dma_only`
// build of tools/ks_variants.py); one per k-step hides in an MFMA's shadow.  The vmcnt count is unchanged: every piece of a chunk is issued before
// the next ring step (the first chunk of a layer, whose ring step has one k-step to the next, flushes its last three pieces in front of it).
struct Pend {
    uint32_t goff, wr;        // the chunk whose pieces are being issued: offset in the global stream, LDS slot
};
__device__ __forceinline__ void dma_piece(const uint8_t* stream, const Pend& pd, lds_char* lds, int wave, int lane, int p) {
#if defined(SNERF_ABLATE) && (ABL & (4 | 32))
    return;
#endif
#if defined(SNERF_ABLATE) && (ABL & 256)      // timing-only: half the DMA bytes
    if (p >= 2) return;
#endif
    const uint8_t* bp = stream + pd.goff + (wave + 4 * p) * kFragBytes;                          // wave-uniform
    const uint32_t dst = (uint32_t)(uintptr_t)(lds + pd.wr + (wave + 4 * p) * kFragBytes);      // wave-uniform LDS byte address
    const uint32_t voff = lane * 16;
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %3\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(dst), "s"(bp)
        : "memory", "scc");
}
__device__ __forceinline__ void ring_step_ks(Ring& rg, Pend& pd, uint32_t stream_bytes) {
#if defined(SNERF_ABLATE) && (ABL & 4)
    return;
#endif
#if defined(SNERF_ABLATE) && (ABL & 64)
    asm volatile("" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((RING_D - 3) * DMA_PER_WAVE) : "memory");
#endif
    pd.goff = rg.goff;
    pd.wr = rg.wr;
    rg.goff += kChunkBytes;
#if defined(SNERF_ABLATE) && (ABL & 128)      // timing-only: the stream cycles over 64 KiB (always cache-resident)
    if (rg.goff >= 4 * kChunkBytes) rg.goff = 0;
#endif
    if (rg.goff >= stream_bytes) rg.goff = 0;
    rg.cur = rg.rd;
    rg.rd = ring_next<RING_D>(rg.rd);
    rg.wr = ring_next<RING_D>(rg.wr);
}
// k-step q of a layer of NP pairs: the ring step of chunk (q + PF) / 4 opens it where (q + PF) % 4 == 0 (q = 1, 5, 9, ...; the layer's first chunk in
// its prologue).  Piece to issue in k-step q: q = 0 -> piece 0 of the first chunk (pieces 1..3 are flushed at q = 1); q >= 1 -> piece (q - 1) % 4 of the
// chunk opened at k-step q - (q - 1) % 4, if that k-step had a ring step.
template <int NP>
__device__ __forceinline__ void ks_ring(Ring& rg, Pend& pd, const uint8_t* stream, uint32_t stream_bytes, lds_char* lds, int wave, int lane, int q) {
    static_assert(PF == 3 && KS_PAR_PAIRS == 4, "piece schedule written for 3 pairs of prefetch, 4 pairs per chunk and parity");
    if (q == 1) {
        dma_piece(stream, pd, lds, wave, lane, 1);
        dma_piece(stream, pd, lds, wave, lane, 2);
        dma_piece(stream, pd, lds, wave, lane, 3);
    }
    if (q + PF < NP && (q + PF) % KS_PAR_PAIRS == 0) ring_step_ks(rg, pd, stream_bytes);
    if (q == 0) dma_piece(stream, pd, lds, wave, lane, 0);
    else if ((q - (q - 1) % 4) + PF < NP) dma_piece(stream, pd, lds, wave, lane, (q - 1) % 4);
}

__device__ __forceinline__ void xbuf_write(lds_char* xw, const f32x16& v, int lane) {
#if defined(SNERF_ABLATE) && (ABL & 16)       // timing-only: no partial-sum exchange (the values stay live: the F phases must not become dead code)
    _Pragma("unroll") for (int q = 0; q < 16; ++q) asm volatile("" ::"v"(v[q]));
    return;
#endif
    __attribute__((address_space(3))) f32x4* p = (__attribute__((address_space(3))) f32x4*)(xw + lane * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) p[q * 64] = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
}
__device__ __forceinline__ f32x16 xbuf_read(lds_char* xr, int lane) {
    lds_cf32x4* p = (lds_cf32x4*)(xr + lane * 16);
    f32x16 v;
#if defined(SNERF_ABLATE) && (ABL & 16)
    _Pragma("unroll") for (int q = 0; q < 16; ++q) v[q] = 0.f;
    return v;
#endif
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 t = p[q * 64];
        v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
    }
    return v;
}
// phase A of the epilogue with the partner's partial sums added in front of the sine
__device__ __forceinline__ void epi_A2(const f32x16& acc, const f32x16& part, int e, EpiTmp& t) {
    t.v0 = sin2pi(acc[2 * e] + part[2 * e]);
    t.v1 = sin2pi(acc[2 * e + 1] + part[2 * e + 1]);
}

// One layer over a wave pair.  NBH own (= foreign) blocks per wave, KSH k-steps of this wave's K-half of the hidden input (0: none - the layer
// reads an encoding only and needs no exchange), KSX k-steps of an encoding that enter the own blocks in full.  bias_own: bias rows of the own blocks.
template <int NBH, int KSH, int KSX>
__device__ __forceinline__ void run_layer_ks(Ring& rg, Pend& pd, const uint8_t* stream, uint32_t stream_bytes, const KsCtx& cx, lds_cfloat* bias_own,
                                             const Frag* in0, const Frag* in1, Frag* out, int wave, int lane) {
    constexpr bool EX = KSH > 0;
    constexpr int KO = KSH + KSX;                  // k-steps of an O phase
    constexpr int SL = (EX ? KSH : 0) + KO;        // pairs per step (F + O)
    constexpr int NP = NBH * SL;
    static_assert(!EX || KSH >= 8, "the exchange is ordered by >= 2 ring steps per phase");
    constexpr bool PIPE = KO >= 4;                 // the sliced epilogue needs 4 k-steps; a two-step encoding layer (per-ray networks) runs it in one piece
    static_assert(PIPE || !EX, "only an encoding-only layer may be that short");
    const int h = lane >> 5;
    u32x4 fh[PF], fl[PF];
#pragma unroll
    for (int q = 0; q < PF; ++q) {
        if (q % KS_PAR_PAIRS == 0) ring_step_ks(rg, pd, stream_bytes);       // the layer's first chunk; its pieces go out in k-steps 0 and 1 (ks_ring)
        lds_char* ap = cx.lds + rg.cur + cx.par_off + (q % KS_PAR_PAIRS) * kPairBytes + lane * 16;
        fh[q] = *(lds_cu32x4*)ap;
        fl[q] = *(lds_cu32x4*)(ap + kFragBytes);
    }
    f32x16 accF, accO[2], part;
    EpiTmp et[8];
    f32x16 next_init = load_bias(bias_own, 0, h);
    // one k-step: consume pair q, prefetch pair q + PF (ring step where it opens a chunk)
#if defined(SNERF_ABLATE) && (ABL & 2)        // timing-only: the A fragments stay in registers, no LDS reads
#define KS_FRAG_READ(q) asm volatile("" : "+v"(fh[(q) % PF]), "+v"(fl[(q) % PF]));
#else
#define KS_FRAG_READ(q)                                                                                                 \
    {                                                                                                                   \
        lds_char* ap = cx.lds + rg.cur + cx.par_off + (((q) + PF) % KS_PAR_PAIRS) * kPairBytes + lane * 16;             \
        fh[(q) % PF] = *(lds_cu32x4*)ap;                                                                                \
        fl[(q) % PF] = *(lds_cu32x4*)(ap + kFragBytes);                                                                 \
    }
#endif
#define KS_STEP(q, ACC, BFRAG)                                                                                          \
    {                                                                                                                   \
        const u32x4 a_hi = fh[(q) % PF], a_lo = fl[(q) % PF];                                                           \
        ks_ring<NP>(rg, pd, stream, stream_bytes, cx.lds, wave, lane, (q));                                             \
        if ((q) + PF < NP) KS_FRAG_READ(q)                                                                              \
        ACC = mfma3(a_hi, a_lo, BFRAG, ACC);                                                                            \
    }
#ifdef KS_NO_SCHED
#define KS_SCHED() __builtin_amdgcn_sched_barrier(0);
#else
#define KS_SCHED()                                                                                                      \
    {                                                                                                                   \
        _Pragma("unroll") for (int m = 0; m < 3; ++m) {                                                                 \
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);                                                        \
            __builtin_amdgcn_sched_group_barrier(SG_DSREAD, 1, 0);                                                      \
            __builtin_amdgcn_sched_group_barrier(SG_TRANS, 1, 0);                                                       \
            __builtin_amdgcn_sched_group_barrier(SG_VALU, 3, 0);                                                        \
        }                                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    }
#endif
#pragma unroll
    for (int i = 0; i < NBH; ++i) {
        if constexpr (EX) {
            // ---- F phase: partial sums of the partner's block i over this wave's K-half; the epilogue of the own block i - 1 rides along
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < KSH; ++s) {
                const int q = i * SL + s;
                if (i > 0 && s == 0) part = xbuf_read(cx.xr, lane);       // written by the partner in the first k-steps of its last O phase
                KS_STEP(q, acc, in0[s]);
                if (i > 0) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int sA = 1 + (e * (KSH - 4)) / 8;
                        if (s == sA + 2) epi_C(e, et[e], out + 2 * (i - 1));
                        if (s == sA + 1) epi_B(e, et[e], out + 2 * (i - 1));
                        if (s == sA) epi_A2(accO[(i - 1) & 1], part, e, et[e]);
                    }
                }
                KS_SCHED();
            }
            accF = acc;
        }
        {
            // ---- O phase: the own block i over this wave's K-half and, in full, the encoding
            f32x16 acc = next_init;
#pragma unroll
            for (int s = 0; s < KO; ++s) {
                const int q = i * SL + (EX ? KSH : 0) + s;
                if (EX && s == 1) xbuf_write(cx.xw, accF, lane);           // in front of this k-step's ring step (if any): see the header
                KS_STEP(q, acc, (s < KSH ? in0[s] : in1[s - KSH]));
                if (i + 1 < NBH && s == KO - 1) next_init = load_bias(bias_own, i + 1, h);
                if (!EX && i > 0) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (PIPE) {
                            const int sA = 1 + (e * (KO - 4)) / 8;
                            if (s == sA + 2) epi_C(e, et[e], out + 2 * (i - 1));
                            if (s == sA + 1) epi_B(e, et[e], out + 2 * (i - 1));
                            if (s == sA) epi_A(accO[(i - 1) & 1], e, et[e]);
                        } else if (s == 0) {
                            epi_A(accO[(i - 1) & 1], e, et[e]);
                            epi_B(e, et[e], out + 2 * (i - 1));
                            epi_C(e, et[e], out + 2 * (i - 1));
                        }
                    }
                }
                KS_SCHED();
            }
            accO[i & 1] = acc;
        }
    }
    // ---- the last own block: nothing left to hide its epilogue behind
    if constexpr (EX) {
        part = xbuf_read(cx.xr, lane);
#pragma unroll
        for (int e = 0; e < 8; ++e) epi_A2(accO[(NBH - 1) & 1], part, e, et[e]);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) epi_A(accO[(NBH - 1) & 1], e, et[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) epi_B(e, et[e], out + 2 * (NBH - 1));
#pragma unroll
    for (int e = 0; e < 8; ++e) epi_C(e, et[e], out + 2 * (NBH - 1));
}

// A raw head (one 32-row block): each wave multiplies its K-half (parity 0 starts from the bias, parity 1 from zero), both publish their half and
// both add the partner's - symmetric, no branch; one extra workgroup barrier per head.
template <int KSH>
__device__ __forceinline__ f32x16 run_head_ks(Ring& rg, Pend& pd, const uint8_t* stream, uint32_t stream_bytes, const KsCtx& cx, lds_cfloat* bias_sel,
                                              const Frag* in0, int wave, int lane) {
    constexpr int NP = KSH;
    const int h = lane >> 5;
    u32x4 fh[PF], fl[PF];
#pragma unroll
    for (int q = 0; q < PF; ++q) {
        if (q % KS_PAR_PAIRS == 0) ring_step_ks(rg, pd, stream_bytes);       // the layer's first chunk; its pieces go out in k-steps 0 and 1 (ks_ring)
        lds_char* ap = cx.lds + rg.cur + cx.par_off + (q % KS_PAR_PAIRS) * kPairBytes + lane * 16;
        fh[q] = *(lds_cu32x4*)ap;
        fl[q] = *(lds_cu32x4*)(ap + kFragBytes);
    }
    f32x16 acc = load_bias(bias_sel, 0, h);
#pragma unroll
    for (int s = 0; s < KSH; ++s) {
        KS_STEP(s, acc, in0[s]);
        __builtin_amdgcn_sched_barrier(0);
    }
    xbuf_write(cx.xw, acc, lane);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const f32x16 part = xbuf_read(cx.xr, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += part[r];
    return acc;
}
#undef KS_STEP
#undef KS_SCHED
#undef KS_FRAG_READ

template <int W, int VARIANT>
__global__ __launch_bounds__(256, 1) void mlp_ks_kernel(const MlpArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int C_MAX = kMaxClasses;
    constexpr int W2 = W / 2;
    lds_char* lds = (lds_char*)smem;
    __attribute__((address_space(3))) float* bias_lds = (__attribute__((address_space(3))) float*)(lds + RING_BYTES);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = wave >> 1, par = wave & 1;
    const int h = lane >> 5;
    const int C = A.n_classes;

    for (int i = threadIdx.x; i < A.bias_floats + 32; i += 256) bias_lds[i] = i < A.bias_floats ? A.bias[i] : 0.f;
    lds_cfloat* zero_bias = bias_lds + A.bias_floats;
    KsCtx cx;
    cx.lds = lds;
    cx.par_off = par * KS_PAR_BYTES;
    {
        lds_char* xb = lds + RING_BYTES + (A.bias_floats + 32) * 4;
        cx.xw = xb + (pair * 2 + par) * KS_XBUF_BYTES;
        cx.xr = xb + (pair * 2 + (par ^ 1)) * KS_XBUF_BYTES;
    }
    __attribute__((address_space(3))) float* vote_lds = (__attribute__((address_space(3))) float*)(lds + RING_BYTES + (A.bias_floats + 32) * 4 + 4 * KS_XBUF_BYTES);

    Ring rg;
    Pend pd;
    pd.goff = 0;
    pd.wr = 0;
    rg.rd = 0;
    rg.cur = 0;
    rg.goff = 0;
    {
        uint32_t wr = 0;
#pragma unroll
        for (int c = 0; c < RING_D - 2; ++c) {
            dma_chunk(A.stream, rg.goff, lds, wr, wave, lane);
            rg.goff += kChunkBytes;
            if (rg.goff >= A.stream_bytes) rg.goff = 0;
            wr += kChunkBytes;
        }
        rg.wr = wr;
    }
    __syncthreads();

    // VARIANT 3 (ray visibility, mlp_device.h RaySum): a "tile" is a group of 2 rays (one per wave pair), walked in `passes` steps of 32 samples
    const int64_t n_tiles = VARIANT == 3 ? (A.n + 1) / 2 : (A.n + KS_TILE_PTS - 1) / KS_TILE_PTS;
    const int passes = VARIANT == 3 ? (A.n_samples + 31) / 32 : 1;
    int pass = 0;
    RaySum rs;
    for (int64_t tile = blockIdx.x; tile < n_tiles;) {
        const int64_t n = tile * KS_TILE_PTS + pair * 32 + (lane & 31);
        const bool valid = n < A.n;
        const int64_t nc = valid ? n : A.n - 1;
        const int64_t g = VARIANT == 3 ? 0 : nc / A.group_size;

        float x0, x1, x2;
        if constexpr (VARIANT == 3) {
            raysum_point(rs, A, tile, 2, pair, pass, lane, x0, x1, x2);
        } else if (A.points) {
            x0 = A.points[nc * 3]; x1 = A.points[nc * 3 + 1]; x2 = A.points[nc * 3 + 2];
        } else {
            const int64_t r = nc / A.n_samples;
            const int s = (int)(nc - r * A.n_samples);
            const float t = A.tvals[s], omt = __fsub_rn(1.f, t);
            x0 = __fadd_rn(__fmul_rn(A.top[r * 3], omt), __fmul_rn(A.bot[r * 3], t));
            x1 = __fadd_rn(__fmul_rn(A.top[r * 3 + 1], omt), __fmul_rn(A.bot[r * 3 + 1], t));
            x2 = __fadd_rn(__fmul_rn(A.top[r * 3 + 2], omt), __fmul_rn(A.bot[r * 3 + 2], t));
        }
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
        Frag pe[PEPOS_KS];
        make_pe_pos(x0, x1, x2, h, pe);

        constexpr int KH = W / 32, KH2 = W2 / 32;        // k-steps of this wave's K-half of a W- / W/2-wide activation
        constexpr int NBW = W / 64, NBW2 = W2 / 64;      // own blocks of a W- / W/2-wide layer
        static_assert(ks_layer_pairs(field_layer(W, C_MAX, F_FC1)) == NBW * PEPOS_KS && ks_layer_pairs(field_layer(W, C_MAX, F_FC2)) == NBW * 2 * KH &&
                      ks_layer_pairs(field_layer(W, C_MAX, F_FC5)) == NBW * (2 * KH + PEPOS_KS) && ks_layer_pairs(field_layer(W, C_MAX, F_FC9)) == NBW2 * 2 * KH &&
                      ks_layer_pairs(field_layer(W, C_MAX, F_HEAD)) == KH2 && ks_layer_pairs(field_layer(W, C_MAX, F_S1)) == NBW2 * (2 * KH2 + PESUN_KS) &&
                      ks_layer_pairs(field_layer(W, C_MAX, F_S2)) == NBW2 * 2 * KH2 && ks_layer_pairs(field_layer(W, C_MAX, F_A1)) == NBW * 2 * KH2 &&
                      ks_layer_pairs(field_layer(W, C_MAX, F_AC)) == KH, "kernel and packer (program.h ks_*) disagree about the stream");
        Frag hA[KH], hB[KH];
        // bias rows of this wave's own blocks of layer L (n_out = 64 NBH)
#define OWNB(L, NBH) (bias_lds + prog_bias_start(PROG_FIELD, W, C_MAX, L) + par * (NBH) * 32)
#define LAYER(L, NBH, KSHv, KSXv, IN0, IN1, OUT) run_layer_ks<NBH, KSHv, KSXv>(rg, pd, A.stream, A.stream_bytes, cx, OWNB(L, NBH), IN0, IN1, OUT, wave, lane)
#define HEADL(L, KSHv, IN0) run_head_ks<KSHv>(rg, pd, A.stream, A.stream_bytes, cx, par ? zero_bias : bias_lds + prog_bias_start(PROG_FIELD, W, C_MAX, L), IN0, wave, lane)
        LAYER(F_FC1, NBW, 0, PEPOS_KS, nullptr, pe, hA);
        LAYER(F_FC2, NBW, KH, 0, hA, nullptr, hB);
        LAYER(F_FC3, NBW, KH, 0, hB, nullptr, hA);
        LAYER(F_FC4, NBW, KH, 0, hA, nullptr, hB);
        LAYER(F_FC5, NBW, KH, PEPOS_KS, hB, pe, hA);
        LAYER(F_FC6, NBW, KH, 0, hA, nullptr, hB);
        LAYER(F_FC7, NBW, KH, 0, hB, nullptr, hA);
        LAYER(F_FC8, NBW, KH, 0, hA, nullptr, hB);
        Frag x1f[KH2];
        LAYER(F_FC9, NBW2, KH, 0, hB, nullptr, x1f);
        f32x16 raw = HEADL(F_HEAD, KH2, x1f);
        const float col_r = raw[0], col_g = raw[1], col_b = raw[2], rho_raw = raw[3];
        float sv_raw = 0.f;
        float adj[3 * C_MAX];
#pragma unroll
        for (int i = 0; i < 3 * C_MAX; ++i) adj[i] = 0.f;
        if constexpr (VARIANT <= 1) {
            Frag ps[PESUN_KS];
            make_pe_sun(s0, s1, s2, h, ps);
            Frag sA[KH2], sB[KH2];
            LAYER(F_S1, NBW2, KH2, PESUN_KS, x1f, ps, sA);
            LAYER(F_S2, NBW2, KH2, 0, sA, nullptr, sB);
            LAYER(F_S3, NBW2, KH2, 0, sB, nullptr, sA);
            raw = HEADL(F_S4, KH2, sA);
            sv_raw = raw[0];
        }
        if constexpr (VARIANT == 0) {
            LAYER(F_A1, NBW, KH2, 0, x1f, nullptr, hA);
            LAYER(F_A2, NBW, KH, 0, hA, nullptr, hB);
            LAYER(F_A3, NBW, KH, 0, hB, nullptr, hA);
            raw = HEADL(F_AC, KH, hA);
#pragma unroll
            for (int i = 0; i < 3 * C_MAX; ++i) adj[i] = raw[i];
        }
#undef LAYER
#undef HEADL
#undef OWNB
        if constexpr (VARIANT == 3) {
            raysum_add(rs, A, tile, 2, pair, pass, lane, rho_raw, x0, x1, x2);
            if (++pass == passes || raysum_saturated(rs, A, tile * 2 + pair, wave, 4, lane, vote_lds)) {      // both waves of a pair hold the same sum and vote alike
                if (par == 0) raysum_end(rs, A, tile, 2, pair, lane);
                pass = 0;
                tile += gridDim.x;
            }
        } else {
            if (par == 0 && h == 0 && valid) store_field_outputs<VARIANT>(A.out, n, C, x0, x1, x2, col_r, col_g, col_b, rho_raw, sv_raw, adj, pcls);
            tile += gridDim.x;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
}

// The per-ray networks at this width (time -> class softmax, T_NeRF_net_v2.py:77-78,160-163; sun -> sky colour, G_NeRF.py:110-111) on the same wave-pair
// structure: one "point" per ray, 64 rays per tile.  Until round 6 they ran layer by layer in exact fp32 - five GEMM launches and seven small kernels,
// ~0.3 ms in front of every width-512 render step; bf16x3 is what the per-ray networks of the other widths run in (never int8 digits: their error is
// per ray, not averaged over a ray's samples).
template <int W>
__global__ __launch_bounds__(256, 1) void mlp_ks_group_kernel(const MlpArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int C_MAX = kMaxClasses;
    lds_char* lds = (lds_char*)smem;
    __attribute__((address_space(3))) float* bias_lds = (__attribute__((address_space(3))) float*)(lds + RING_BYTES);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = wave >> 1, par = wave & 1;
    const int h = lane >> 5;
    const int C = A.n_classes;
    for (int i = threadIdx.x; i < A.bias_floats + 32; i += 256) bias_lds[i] = i < A.bias_floats ? A.bias[i] : 0.f;
    lds_cfloat* zero_bias = bias_lds + A.bias_floats;
    KsCtx cx;
    cx.lds = lds;
    cx.par_off = par * KS_PAR_BYTES;
    {
        lds_char* xb = lds + RING_BYTES + (A.bias_floats + 32) * 4;
        cx.xw = xb + (pair * 2 + par) * KS_XBUF_BYTES;
        cx.xr = xb + (pair * 2 + (par ^ 1)) * KS_XBUF_BYTES;
    }
    Ring rg;
    Pend pd;
    pd.goff = 0;
    pd.wr = 0;
    rg.rd = 0;
    rg.cur = 0;
    rg.goff = 0;
    {
        uint32_t wr = 0;
#pragma unroll
        for (int c = 0; c < RING_D - 2; ++c) {
            dma_chunk(A.stream, rg.goff, lds, wr, wave, lane);
            rg.goff += kChunkBytes;
            if (rg.goff >= A.stream_bytes) rg.goff = 0;
            wr += kChunkBytes;
        }
        rg.wr = wr;
    }
    __syncthreads();
    const int64_t n_tiles = (A.n + KS_TILE_PTS - 1) / KS_TILE_PTS;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t n = tile * KS_TILE_PTS + pair * 32 + (lane & 31);
        const bool valid = n < A.n;
        const int64_t nc = valid ? n : A.n - 1;
        const float t0 = A.time[nc * 4], t1 = A.time[nc * 4 + 1];
        const float s0 = A.sun[nc * 3], s1 = A.sun[nc * 3 + 1], s2 = A.sun[nc * 3 + 2];
        constexpr int KH = W / 32, W4P = pad32(W / 4), KH4 = W4P / 32, NBW = W / 64, NB4 = W4P / 64;
        static_assert(ks_layer_pairs(group_layer(W, C_MAX, G_T1)) == NBW * PETIME_KS && ks_layer_pairs(group_layer(W, C_MAX, G_T2)) == NBW * 2 * KH &&
                      ks_layer_pairs(group_layer(W, C_MAX, G_CL)) == KH && ks_layer_pairs(group_layer(W, C_MAX, G_K1)) == NB4 * PESUN_KS &&
                      ks_layer_pairs(group_layer(W, C_MAX, G_K2)) == KH4, "kernel and packer (program.h ks_*) disagree about the stream");
#define OWNB(L, NBH) (bias_lds + prog_bias_start(PROG_GROUP, W, C_MAX, L) + par * (NBH) * 32)
#define LAYER(L, NBH, KSHv, KSXv, IN0, IN1, OUT) run_layer_ks<NBH, KSHv, KSXv>(rg, pd, A.stream, A.stream_bytes, cx, OWNB(L, NBH), IN0, IN1, OUT, wave, lane)
#define HEADL(L, KSHv, IN0) run_head_ks<KSHv>(rg, pd, A.stream, A.stream_bytes, cx, par ? zero_bias : bias_lds + prog_bias_start(PROG_GROUP, W, C_MAX, L), IN0, wave, lane)
        Frag pt[PETIME_KS];
        make_pe_time(t0, t1, h, pt);
        Frag hA[KH], hB[KH];
        LAYER(G_T1, NBW, 0, PETIME_KS, nullptr, pt, hA);
        LAYER(G_T2, NBW, KH, 0, hA, nullptr, hB);
        f32x16 raw = HEADL(G_CL, KH, hB);
        float logit[C_MAX];
#pragma unroll
        for (int c = 0; c < C_MAX; ++c) logit[c] = raw[c];
        Frag ps[PESUN_KS];
        make_pe_sun(s0, s1, s2, h, ps);
        Frag kA[KH4];
        LAYER(G_K1, NB4, 0, PESUN_KS, nullptr, ps, kA);
        raw = HEADL(G_K2, KH4, kA);
#undef LAYER
#undef HEADL
#undef OWNB
        if (par == 0 && h == 0 && valid) {
            float m = -3.0e38f;
#pragma unroll
            for (int c = 0; c < C_MAX; ++c) if (c < C) m = fmaxf(m, logit[c]);
            float e[C_MAX], sum = 0.f;
#pragma unroll
            for (int c = 0; c < C_MAX; ++c) { e[c] = c < C ? expf(logit[c] - m) : 0.f; sum += e[c]; }
#pragma unroll
            for (int c = 0; c < C_MAX; ++c) if (c < C && A.g_classes) A.g_classes[n * C + c] = e[c] / sum;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (A.g_sky_raw) A.g_sky_raw[n * 3 + k] = raw[k];
                if (A.g_sky) A.g_sky[n * 3 + k] = sigmoid_f(raw[k]);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
}

hipError_t launch_mlp_ks_group(int W, const MlpArgs& a, int n_cu, hipStream_t st) {
    if (W != 512) return hipErrorInvalidValue;
    const int lds_bytes = ks_lds_bytes(a.bias_floats);
    const int64_t n_tiles = (a.n + KS_TILE_PTS - 1) / KS_TILE_PTS;
    int grid = (int)(n_tiles < n_cu ? n_tiles : n_cu);
    if (grid < 1) grid = 1;
    auto k = mlp_ks_group_kernel<512>;
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds_bytes, st, a);
    return hipGetLastError();
}
int group_chunks_ks(int W, int C) { return ks_chunk_start_g(W, C, G_NUM); }

template <int W, int VARIANT>
static hipError_t launch_ks_t(const MlpArgs& a, int n_cu, hipStream_t st) {
    const int lds_bytes = ks_lds_bytes(a.bias_floats);
    if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
    const int64_t n_tiles = VARIANT == 3 ? (a.n + 1) / 2 : (a.n + KS_TILE_PTS - 1) / KS_TILE_PTS;
    int grid = (int)(n_tiles < n_cu ? n_tiles : n_cu);
    if (grid < 1) grid = 1;
    auto k = mlp_ks_kernel<W, VARIANT>;
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds_bytes, st, a);
    return hipGetLastError();
}

hipError_t launch_mlp_ks(int W, int variant, const MlpArgs& a, int n_cu, hipStream_t st) {
    if (W != 512) return hipErrorInvalidValue;
    switch (variant) {
        case 0: return launch_ks_t<512, 0>(a, n_cu, st);
        case 1: return launch_ks_t<512, 1>(a, n_cu, st);
        case 2: return launch_ks_t<512, 2>(a, n_cu, st);
        case 3: return launch_ks_t<512, 3>(a, n_cu, st);
    }
    return hipErrorInvalidValue;
}

// chunks consumed per tile by a variant (the DMA stream is cyclic over exactly these); variant 3 = the layers of variant 2
int field_variant_chunks_ks(int W, int C, int variant) {
    const int last = variant == 0 ? (int)F_NUM : variant == 1 ? (int)F_A1 : (int)F_S1;
    return ks_chunk_start(W, C, last);
}
int mlp_ks_lds_bytes(int bias_floats) { return ks_lds_bytes(bias_floats); }
int mlp_ks_tile_points() { return KS_TILE_PTS; }

}  // namespace snerf
