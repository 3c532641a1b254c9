// gfx950 (MI355X / CDNA4): the fused field network on the int8 matrix pipe ("i8x3", program.h FMT_I8).
//
//  mlp_i8_kernel<PROG, W, VARIANT>   same register-resident chain as mlp_kernel<PROG_FIELD, W, VARIANT> (kernels.hip), but every
//                              operand is 16-bit fixed point carried as two signed int8 digits on v_mfma_i32_32x32x32_i8:
//     * per 32-feature k-step three MFMAs (T a -> M;  T b, L a -> X; the L b term is dropped) instead of six bf16 ones
//       (three split products for each of the two 16-slot halves): half the matrix-pipe time, exact integer accumulation;
//     * an activation costs 2 bytes of register state instead of 4 (bf16 hi + lo), a weight 2 bytes of LDS / L2 traffic
//       instead of 4, and one 32x32 output block is exactly one k-step of the next layer;
//     * epilogue per element: (M << 8) + X, v_cvt_f32_i32, fma with the row's scale and bias (both fold BatchNorm, the
//       factor 30, 1/(2 pi) and the +128 digit offset, pack.cpp), v_sin_f32, then v_cvt_pknorm_i16_f32 (two elements per
//       instruction) and two v_perm_b32 + one v_xor per four elements to split the int16 into the two digit streams.
//  Accuracy: 16-bit fixed point on both operands: RGB within ~2e-5 relative of the fp32 reference (bf16x3: ~3e-6), density
//  and the other per-sample outputs within ~1e-4 (tools/numerics_i8.py, tests/test_gpu_precision.py).  No input range: the
//  digit operands are sines / cosines and hidden activations, all in [-1,1] by construction; the raw coordinates the
//  encodings carry along (x, y, z; the sun vector; the time code) enter in fp32, three FMAs per element of the layers that
//  read them (fc1, fc5, fc_solar_1; time_layer_1, fc_sky_color_1).
#include "mlp_i8_device.h"

namespace snerf {


// ---- W = 512: the hidden activations (2 x 128 registers of B operands) cannot share the 256 architectural VGPRs with
// everything else, and hipcc neither places an MFMA B operand in an AGPR on its own nor keeps "a"-constrained values there
// (it spills them to scratch and reloads before every MFMA: 13.7 ms per 4096 x 96 batch).  So the activations live in
// AGPRs ADDRESSED BY NUMBER: a[base + 8 k .. base + 8 k + 7] = high / low digits of k-step k (two 4-register tuples),
// written by v_accvgpr_write at the digit split and read by MFMAs issued through inline asm.  The compiler does not see
// these registers: reserve_agprs() clobbers all 256 once so that it allocates none of them (the kernel uses no AGPR for
// itself: MFMA results are in VGPRs, and nothing spills), volatile asm keeps parks and MFMAs in program order.  hipcc pads
// no hazards around inline asm: the accumulators of an asm MFMA are read by VALU code no earlier than three MFMAs later
// (pipelined epilogue: k-step >= 1 of the next block) or behind an explicit s_nop pad (the last block of a layer), and
// parked digits are consumed one block (the last block: one ring step) after they were written.
constexpr int AG_R0 = 0, AG_R1 = 128;      // two 128-register regions; who lives where: see the layer list of the kernel
// register numbers are "i" operands: loop indices that are constants once the layer is unrolled (a number that did not fold
// fails the build in the backend, it cannot reach the GPU)
#ifdef SNERF_DBG_NOP_PARK
#define DBG_PARK_PAD "\n\ts_nop 7"
#else
#define DBG_PARK_PAD ""
#endif
#ifdef SNERF_DBG_NOP_MFMA
#define DBG_MFMA_PAD "\n\ts_nop 15\n\ts_nop 15"
#else
#define DBG_MFMA_PAD ""
#endif
#ifdef SNERF_DBG_EARLYCLOBBER
#define DBG_EC "=&v"
#else
#define DBG_EC "=v"
#endif
static __device__ __forceinline__ void park(int v, int idx) {
    asm volatile("v_accvgpr_write_b32 a[%1], %0" DBG_PARK_PAD ::"v"(v), "i"(idx));
}
template <bool FIRST>
__device__ __forceinline__ void mfma_asm(i32x16& acc, const i32x4& a, int base) {
    if (FIRST) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, a[%2:%3], 0" DBG_MFMA_PAD : DBG_EC(acc) : "v"(a), "i"(base), "i"(base + 3));
    else asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, a[%2:%3], %0" DBG_MFMA_PAD : "+v"(acc) : "v"(a), "i"(base), "i"(base + 3));
}
template <bool FIRST>      // base: first register of the k-step (4 high-digit dwords, then 4 low-digit dwords)
__device__ __forceinline__ void mfma_i8x3_agpr(const i32x4& aT, const i32x4& aL, Acc8& acc, int base) {
    mfma_asm<FIRST>(acc.M, aT, base);
    mfma_asm<FIRST>(acc.X, aT, base + 4);
    mfma_asm<false>(acc.X, aL, base);
}
#define A8(n) "a" #n
#define A8x8(n) A8(n##0), A8(n##1), A8(n##2), A8(n##3), A8(n##4), A8(n##5), A8(n##6), A8(n##7), A8(n##8), A8(n##9)
__device__ __forceinline__ void reserve_agprs() {
    asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", A8x8(1), A8x8(2), A8x8(3), A8x8(4), A8x8(5), A8x8(6),
                 A8x8(7), A8x8(8), A8x8(9), A8x8(10), A8x8(11), A8x8(12), A8x8(13), A8x8(14), A8x8(15), A8x8(16), A8x8(17), A8x8(18),
                 A8x8(19), A8x8(20), A8x8(21), A8x8(22), A8x8(23), A8x8(24), "a250", "a251", "a252", "a253", "a254", "a255");
}

template <bool AG>       // AG: the block's 8 registers start at AGPR ago (o is not touched)
__device__ __forceinline__ void epi_Q(int g, const float* ev, Frag8* o, int ago) {
    int h, l;
    digits4(ev[4 * g], ev[4 * g + 1], ev[4 * g + 2], ev[4 * g + 3], h, l);
    if constexpr (AG) {
        park(h, ago + g);
        park(l, ago + 4 + g);
    } else {
        o->hi[g] = h;
        o->lo[g] = l;
    }
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
//   AG_IN0 / AG_OUT >= 0: the first input block / the output live in AGPRs from that register number on (W = 512; the
//   pointer is then unused); the second input block, where there is one, is always an encoding in VGPRs.
//   RAWL: the layer reads an encoding; rawx = its three raw coordinates, added in fp32 (mlp_i8_device.h add_raw).
template <int NB, int KS0, int KS1, bool SIN, int D, int AG_IN0 = -1, int AG_OUT = -1, bool AG = false, bool RAWL = false, bool OPQ = false>
__device__ __forceinline__ void run_layer8(Ring& rg, const uint8_t* stream, uint32_t stream_bytes, lds_char* lds, lds_cfloat* tab_l,
                                           const Frag8* in0, const Frag8* in1, Frag8* out, f32x16* raw, int wave, int lane,
                                           const float* rawx = nullptr) {
    static_assert(!RAWL || SIN, "raw coordinates only enter sine layers");
    // OPQ (ray-visibility variant): the table base is made opaque per layer so that hipcc does not hoist one address register per table row out of
    // the persistent tile loop (kernels_i8x2.hip run_layer8x2: the same measure; without it the W = 512 instance of the variant needs 32 bytes of scratch)
    if constexpr (OPQ) asm volatile("" : "+v"(tab_l));
    lds_cfloat* raw_l = tab_l + 2 * 32 * NB;         // the layer's raw-weight table follows its scale / bias table
    Raw8 rw;
    constexpr int KS = KS0 + KS1, NP = NB * KS;
    constexpr bool PIPE = KS >= 4;
    constexpr int S_EPI = (AG && KS > 1) ? 1 : 0;      // un-pipelined epilogue: behind the block's first MFMAs (see mfma_asm)
    static_assert(!AG || KS > 1 || NB == 1, "AGPR path: a one-step layer with several blocks would read accumulators too early");
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
        if (!AG) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc.M[i] = 0; acc.X[i] = 0; }
        }
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
            if (AG) {
                // encodings live in VGPRs (compiler-issued MFMA), hidden activations in AGPRs (asm MFMA)
                if (s >= KS0 || AG_IN0 < 0) {
                    if (s == 0) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) { acc.M[i] = 0; acc.X[i] = 0; }
                    }
                    mfma_i8x3(aT, aL, s < KS0 ? in0[s] : in1[s - KS0], acc);
                } else if (s == 0) {
                    mfma_i8x3_agpr<true>(aT, aL, acc, AG_IN0 + 8 * s);
                } else {
                    mfma_i8x3_agpr<false>(aT, aL, acc, AG_IN0 + 8 * s);
                }
            } else {
                mfma_i8x3(aT, aL, s < KS0 ? in0[s] : in1[s - KS0], acc);
            }
            if (SIN && b > 0) {
                if (s == 0) {
                    tab = load_tab(tab_l, b - 1, h);
                    if (RAWL) rw = load_raw(raw_l, b - 1, h);
                }
                if (PIPE) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int sA = 1 + (e * (KS - 2)) / 8;
                        if (s == sA) epi_A<RAWL>(accs[(b - 1) & 1], tab, e, ev, &rw, rawx);
                        if ((e & 1) && s == sA + 1) epi_Q<(AG_OUT >= 0)>(e >> 1, ev, AG_OUT >= 0 ? nullptr : out + (b - 1), AG_OUT + 8 * (b - 1));
                    }
                } else if (s == S_EPI) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) epi_A<RAWL>(accs[(b - 1) & 1], tab, e, ev, &rw, rawx);
#pragma unroll
                    for (int g = 0; g < 4; ++g) epi_Q<(AG_OUT >= 0)>(g, ev, AG_OUT >= 0 ? nullptr : out + (b - 1), AG_OUT + 8 * (b - 1));
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
    if (RAWL) rw = load_raw(raw_l, NB - 1, h);
    // The last asm MFMA's result: 18 wait states before a VALU read, by hand.  The pad must CARRY the accumulators ("+v"): a bare asm volatile with a
    // memory clobber orders nothing against register-only VALU code, and hipcc hoisted the whole epilogue above it - the first elements of every
    // layer's last block were read 3 instructions behind the MFMA that was still forming them (stale by its product, and by how far the read ran ahead:
    // found in round 6 as launch-to-launch differences of the seasonal-adjust outputs, tools/ks_race.py; tests/test_isa_guards.py now scans the ISA).
    if (AG) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(accs[(NB - 1) & 1].M), "+v"(accs[(NB - 1) & 1].X)::"memory");
    if (SIN) {
#pragma unroll
        for (int e = 0; e < 8; ++e) epi_A<RAWL>(accs[(NB - 1) & 1], tab, e, ev, &rw, rawx);
#pragma unroll
        for (int g = 0; g < 4; ++g) epi_Q<(AG_OUT >= 0)>(g, ev, AG_OUT >= 0 ? nullptr : out + (NB - 1), AG_OUT + 8 * (NB - 1));
        if (AG) asm volatile("s_nop 3" ::: "memory");             // parked digits -> the next layer's first MFMA
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) (*raw)[i] = preact(accs[(NB - 1) & 1], tab, i);
    }
}


constexpr int ring_depth8(int W) { return W > 256 ? 5 : RING_D; }    // W = 512: the 72 KB of tables leave room for 5 slots

template <int PROG, int W, int VARIANT>
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

    // VARIANT 3 (ray visibility, mlp_device.h RaySum): a "tile" is a group of 4 rays (one per wave), walked in `passes` steps of 32 samples
    const int64_t n_tiles = VARIANT == 3 ? (A.n + 3) / 4 : (A.n + TILE_PTS - 1) / TILE_PTS;
    const int passes = VARIANT == 3 ? (A.n_samples + 31) / 32 : 1;
    int pass = 0;
    RaySum rs;
    for (int64_t tile = blockIdx.x; tile < n_tiles;) {
        const int64_t n = tile * TILE_PTS + wave * 32 + (lane & 31);
        const bool valid = n < A.n;
        const int64_t nc = valid ? n : A.n - 1;
        const int64_t g = VARIANT == 3 ? 0 : nc / A.group_size;

        if constexpr (PROG == PROG_GROUP) {
            // ---- group program: class softmax (T_NeRF_net_v2.py:77-78) and sky colour (G_NeRF.py:110-111); used where the
            // bf16 group kernel has no instance (W = 512: its activations do not fit the register file)
            constexpr int KW = W / 32, W4P = pad32(W / 4), KW4 = W4P / 32;
            f32x16 raw;
#define LAYER(L, NBv, K0, K1, SINv, IN0, IN1, OUT, RAW)                                                                 \
    run_layer8<NBv, K0, K1, SINv, D>(rg, A.stream, A.stream_bytes, lds, tab_lds + prog_table_start(PROG_GROUP, W, C_MAX, L), \
                                     IN0, IN1, OUT, RAW, wave, lane)
#define LAYER_RAW(L, NBv, K0, K1, IN0, IN1, OUT, RX)                                                                    \
    run_layer8<NBv, K0, K1, true, D, -1, -1, false, true>(rg, A.stream, A.stream_bytes, lds,                             \
        tab_lds + prog_table_start(PROG_GROUP, W, C_MAX, L), IN0, IN1, OUT, nullptr, wave, lane, RX)
            const float t0 = A.time[nc * 4], t1 = A.time[nc * 4 + 1];
            const float s0 = A.sun[nc * 3], s1 = A.sun[nc * 3 + 1], s2 = A.sun[nc * 3 + 2];
            Frag8 pt[PETIME_KS8];
            make_pe_time8(t0, t1, h, pt);
            Frag8 hA[KW], hB[KW];
            const float rx_t[3] = {t0, t1, 0.f}, rx_s[3] = {s0, s1, s2};
            LAYER_RAW(G_T1, W / 32, PETIME_KS8, 0, pt, nullptr, hA, rx_t);
            LAYER(G_T2, W / 32, KW, 0, true, hA, nullptr, hB, nullptr);
            LAYER(G_CL, 1, KW, 0, false, hB, nullptr, nullptr, &raw);
            float logit[C_MAX];
#pragma unroll
            for (int c = 0; c < C_MAX; ++c) logit[c] = raw[c];
            Frag8 ps[PESUN_KS8];
            make_pe_sun8(s0, s1, s2, h, ps);
            Frag8 kA[KW4];
            LAYER_RAW(G_K1, W4P / 32, PESUN_KS8, 0, ps, nullptr, kA, rx_s);
            LAYER(G_K2, 1, KW4, 0, false, kA, nullptr, nullptr, &raw);
#undef LAYER
#undef LAYER_RAW
            if (h == 0 && valid) {
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
            tile += gridDim.x;
            continue;
        }
        // ---- sample position (misc.py:234-247 fused): top*(1-t) + bot*t, two roundings + one add, no fma
        float x0, x1, x2;
        if constexpr (VARIANT == 3) {
            raysum_point(rs, A, tile, 4, wave, pass, lane, x0, x1, x2);
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
        make_pe_pos8<VARIANT == 3>(x0, x1, x2, h, pe);

        constexpr int KW = W / 32, KW2 = W2 / 32;
        Frag8 hA[KW], hB[KW];
        f32x16 raw;
        // AG (W = 512): hidden activations in AGPRs by number (see mfma_asm).  Two regions of 128 registers: the trunk
        // ping-pongs R0 / R1; fc9 reads R1 and writes x1 into the first half of R0; the solar branch uses the two halves
        // of R1; the adjust branch starts from x1 (R0) into R1, so its ping-pong is R1 / R0 / R1.  The Frag8 arrays below
        // are the same buffers for the widths whose activations stay in compiler-allocated VGPRs.
        constexpr bool AG = W > 256;
        constexpr int xA = AG_R0, xB = AG_R1, xX1 = AG_R0, xSA = AG_R1, xSB = AG_R1 + 64, NOAG = -1;
        if constexpr (AG) reserve_agprs();
#define LAYER(L, NBv, K0, K1, SINv, IN0, IN0AG, IN1, OUT, OUTAG, RAW)                                                      \
    run_layer8<NBv, K0, K1, SINv, D, AG ? IN0AG : -1, AG ? OUTAG : -1, AG, false, VARIANT == 3>(rg, A.stream, A.stream_bytes, lds, \
        tab_lds + prog_table_start(PROG_FIELD, W, C_MAX, L), IN0, IN1, OUT, RAW, wave, lane)
#define LAYER_RAW(L, NBv, K0, K1, IN0, IN0AG, IN1, OUT, OUTAG, RX)                                                         \
    run_layer8<NBv, K0, K1, true, D, AG ? IN0AG : -1, AG ? OUTAG : -1, AG, true, VARIANT == 3>(rg, A.stream, A.stream_bytes, lds, \
        tab_lds + prog_table_start(PROG_FIELD, W, C_MAX, L), IN0, IN1, OUT, nullptr, wave, lane, RX)
        const float rx_p[3] = {x0, x1, x2}, rx_s[3] = {s0, s1, s2};      // raw coordinates: fp32, no digit range
        // trunk (G_NeRF.py:80-91)
        LAYER_RAW(F_FC1, W / 32, PEPOS_KS8, 0, pe, NOAG, nullptr, hA, xA, rx_p);
        LAYER(F_FC2, W / 32, KW, 0, true, hA, xA, nullptr, hB, xB, nullptr);
        LAYER(F_FC3, W / 32, KW, 0, true, hB, xB, nullptr, hA, xA, nullptr);
        LAYER(F_FC4, W / 32, KW, 0, true, hA, xA, nullptr, hB, xB, nullptr);
        LAYER_RAW(F_FC5, W / 32, KW, PEPOS_KS8, hB, xB, pe, hA, xA, rx_p);
        LAYER(F_FC6, W / 32, KW, 0, true, hA, xA, nullptr, hB, xB, nullptr);
        LAYER(F_FC7, W / 32, KW, 0, true, hB, xB, nullptr, hA, xA, nullptr);
        LAYER(F_FC8, W / 32, KW, 0, true, hA, xA, nullptr, hB, xB, nullptr);
        Frag8 x1f[KW2];
        LAYER(F_FC9, W2 / 32, KW, 0, true, hB, xB, nullptr, x1f, xX1, nullptr);
        // sigma / colour head (G_NeRF.py:93-98): regs 0..2 colour, 3 density (lane-half 0)
        LAYER(F_HEAD, 1, KW2, 0, false, x1f, xX1, nullptr, nullptr, NOAG, &raw);
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
            LAYER_RAW(F_S1, W2 / 32, KW2, PESUN_KS8, x1f, xX1, ps, sA, xSA, rx_s);
            LAYER(F_S2, W2 / 32, KW2, 0, true, sA, xSA, nullptr, sB, xSB, nullptr);
            LAYER(F_S3, W2 / 32, KW2, 0, true, sB, xSB, nullptr, sA, xSA, nullptr);
            LAYER(F_S4, 1, KW2, 0, false, sA, xSA, nullptr, nullptr, NOAG, &raw);
            sv_raw = raw[0];
        }
        if constexpr (VARIANT == 0) {
            // seasonal colour-adjust branch (T_NeRF_net_v2.py:83-87)
            LAYER(F_A1, W / 32, KW2, 0, true, x1f, xX1, nullptr, hB, xB, nullptr);
            LAYER(F_A2, W / 32, KW, 0, true, hB, xB, nullptr, hA, xA, nullptr);
            LAYER(F_A3, W / 32, KW, 0, true, hA, xA, nullptr, hB, xB, nullptr);
            LAYER(F_AC, 1, KW, 0, false, hB, xB, nullptr, nullptr, NOAG, &raw);
#pragma unroll
            for (int i = 0; i < 3 * C_MAX; ++i) adj[i] = raw[i];
        }
#undef LAYER
#undef LAYER_RAW
        if constexpr (VARIANT == 3) {
            raysum_add(rs, A, tile, 4, wave, pass, lane, rho_raw, x0, x1, x2);
            if (++pass == passes || raysum_saturated(rs, A, tile * 4 + wave, wave, 4, lane, (__attribute__((address_space(3))) float*)(tab_lds + A.bias_floats))) {
                raysum_end(rs, A, tile, 4, wave, lane);
                pass = 0;
                tile += gridDim.x;
            }
        } else {
            if (h == 0 && valid) store_field_outputs<VARIANT>(A.out, n, C, x0, x1, x2, col_r, col_g, col_b, rho_raw, sv_raw, adj, pcls);
            tile += gridDim.x;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
}

template <int PROG, int W, int VARIANT>
static hipError_t launch_mlp_i8_t(const MlpArgs& a, int n_cu, hipStream_t st) {
    const int lds_bytes = ring_depth8(W) * kChunkBytes + a.bias_floats * 4 + kVoteBytes;
    const int64_t n_tiles = VARIANT == 3 ? (a.n + 3) / 4 : (a.n + TILE_PTS - 1) / TILE_PTS;
    int grid = (int)(n_tiles < n_cu ? n_tiles : n_cu);
    if (grid < 1) grid = 1;
    auto k = mlp_i8_kernel<PROG, W, VARIANT>;
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds_bytes, st, a);
    return hipGetLastError();
}

// The W = 512 field instances are compiled in their own translation unit (kernels_i8_w512.hip: this file again, with
// SNERF_I8_W512_TU defined) under a larger pragma-unroll threshold: their layers (16 blocks x 16-18 k-steps) exceed
// hipcc's default cap, and a layer loop left rolled would index the register arrays dynamically (= scratch).
hipError_t launch_mlp_i8_w512(int variant, const MlpArgs& a, int n_cu, hipStream_t st);
#ifdef SNERF_I8_W512_TU
hipError_t launch_mlp_i8_w512(int variant, const MlpArgs& a, int n_cu, hipStream_t st) {
    if (variant == 0) return launch_mlp_i8_t<PROG_FIELD, 512, 0>(a, n_cu, st);
    if (variant == 1) return launch_mlp_i8_t<PROG_FIELD, 512, 1>(a, n_cu, st);
    if (variant == 3) return launch_mlp_i8_t<PROG_FIELD, 512, 3>(a, n_cu, st);
    return launch_mlp_i8_t<PROG_FIELD, 512, 2>(a, n_cu, st);
}
#else
hipError_t launch_mlp_i8(int prog, int W, int variant, const MlpArgs& a, int n_cu, hipStream_t st) {
#define CASE(Wv)                                                                          \
    if (W == Wv) {                                                                        \
        if (prog != PROG_FIELD) return hipErrorInvalidValue;                              \
        if (variant == 0) return launch_mlp_i8_t<PROG_FIELD, Wv, 0>(a, n_cu, st);         \
        if (variant == 1) return launch_mlp_i8_t<PROG_FIELD, Wv, 1>(a, n_cu, st);         \
        if (variant == 3) return launch_mlp_i8_t<PROG_FIELD, Wv, 3>(a, n_cu, st);         \
        return launch_mlp_i8_t<PROG_FIELD, Wv, 2>(a, n_cu, st);                           \
    }
    CASE(64)
    CASE(256)
#undef CASE
    if (W == 512) {
        // the per-ray networks have no int8 instance: one row per ray, their error is not averaged over a ray's samples
        // (api.cpp group_forward_f32 runs them in exact fp32 at this width)
        if (prog != PROG_FIELD) return hipErrorInvalidValue;
        return launch_mlp_i8_w512(variant, a, n_cu, st);
    }
    return hipErrorInvalidValue;
}

// chunks consumed per tile by a variant of the int8 field program (the DMA stream is cyclic over exactly these)
int field_variant_chunks_i8(int W, int C, int variant) {
    const int last = variant == 0 ? (int)F_NUM : variant == 1 ? (int)F_A1 : (int)F_S1;      // variant 3 = the layers of variant 2
    return prog_chunk_start(PROG_FIELD, W, C, last, FMT_I8);
}
#endif

}  // namespace snerf
