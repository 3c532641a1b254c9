// gfx950 (MI355X / CDNA4): the int8-digit fused field network with TWO waves per SIMD (widths <= 256).
//
//  mlp_i8x2_kernel<W, VARIANT>   the arithmetic of mlp_i8_kernel (kernels_i8.hip, mlp_i8_device.h), re-shaped so that a wave
//                                needs < 256 registers and eight waves (256 points) share every weight chunk:
//     * one wave alone leaves the matrix pipe idle 45 % of the time (profiles/r2/a_pmc.txt: 57 % of its cycles issue, 27 % are
//       issue stalls, 15 % waits) - the epilogue's VALU work, LDS latency and the per-chunk barrier cannot all hide behind
//       its own MFMAs.  With a partner on the same SIMD the pipe is fed by whichever wave has an MFMA ready;
//     * no software-pipelined epilogue and no accumulator ping-pong (that is what the registers went to): a block is
//       24 MFMAs with the next fragments prefetched, then its epilogue (sine + digit split) in two halves;
//     * ring protocol as in the one-wave kernels (mlp_device.h): one counted vmcnt wait + one workgroup barrier per 16 KiB chunk,
//       now shared by eight waves - each moves two of the chunk's sixteen 1 KiB pieces - so the L2 -> LDS weight traffic per
//       point halves.
//  Measured (profiles/r2): 0.74 ms against 0.82 ms for the one-wave kernel.  The matrix pipe is busy 59 % of the time: the two
//  waves of a SIMD run in step (the per-chunk barrier re-aligns all eight every block), so the pipe saturates while both are
//  in their MFMA phase and idles while both run epilogues.  Tried against that, each without gain: waves 4-7 taking every
//  barrier half a chunk later in their own stream (a protocol in which barrier m publishes chunk m + 1), s_setprio 2 / 3 for
//  the MFMA phase, fragment prefetch depth 1 / 3, MFMA / VALU interleave pinned inside a k-step (sched_group_barrier, the
//  SNERF_X2_GROUPS switch below), waves 4-7 running their epilogue slices in the odd k-steps and waves 0-3 in the even ones.
//  So the two waves running in step is not what leaves the matrix pipe 40 % idle; the chip holds 1.8 GHz in this kernel.
//  Weight stream, tables, digit formats, accuracy: exactly those of kernels_i8.hip (same packed model, bit-identical results).
#include "mlp_i8_device.h"

namespace snerf {

constexpr int NW2 = 8;                                      // waves per workgroup: two per SIMD
constexpr int TILE2 = 32 * NW2;                             // 256 points per workgroup tile
constexpr int PIECES2 = kChunkBytes / kFragBytes / NW2;     // 1 KiB DMA pieces per wave and chunk (2)
constexpr int RING2_D = 7;
#ifndef SNERF_PFX
#define SNERF_PFX 2
#endif
constexpr int PFX = SNERF_PFX;                                // weight-fragment pairs requested ahead of their MFMAs
static_assert(PIECES2 == 2 && kChunkPairs == 8, "the DMA below moves two pieces per wave; barrier positions assume 8-pair chunks");

#ifdef SNERF_STAMP
// Diagnostic build only (tools/variants.py ... -DSNERF_STAMP): cycle stamps of workgroup 0, second tile: per layer the s_memtime
// at entry, and per wave the cycles spent inside ring_step2 (vmcnt wait + barrier).  Read back with snerf_debug_stamps().
__device__ unsigned long long g_stamps[8 * 64];
__device__ __forceinline__ void stamp(int wave, int lane, int slot, bool on) {
    if (on && lane == 0) g_stamps[wave * 64 + slot] = __builtin_amdgcn_s_memtime();
}
#define STAMP(slot) stamp(wave, lane, slot, stamp_on)
#else
#define STAMP(slot)
#endif

struct Ring2 {
    uint32_t wr;       // LDS offset of the slot the next DMA fills
    uint32_t goff;     // byte offset in the (cyclic) global stream of the next chunk to fetch
    uint32_t cur;      // LDS offset of the chunk this wave is reading
};
__device__ __forceinline__ uint32_t ring2_next(uint32_t off) {
    off += kChunkBytes;
    return off == RING2_D * kChunkBytes ? 0u : off;
}
__device__ __forceinline__ void dma_chunk2(const uint8_t* stream, uint32_t goff, lds_char* lds, uint32_t wr, int wave, int lane);
__device__ __forceinline__ void dma_chunk4(const uint8_t* stream, uint32_t goff, lds_char* lds, uint32_t wr, int wave, int lane);
#ifndef SNERF_X2_ALL_LOAD
#define SNERF_X2_OLD_LOADS 1      // waves 0-3 (which reach every barrier early, see run_layer8x2) move all sixteen pieces of a chunk
#endif
// Hand the next chunk to the consumers and refill the slot released two chunks ago (ring_step of mlp_device.h for eight waves):
// vmcnt((D-3)*2): all but the (D-3) youngest chunks this wave fetched have landed => the chunk about to be read is complete.
template <int PHASE>
__device__ __forceinline__ void ring_step2(Ring2& rg, const uint8_t* stream, uint32_t stream_bytes, lds_char* lds, int wave, int lane) {
#if defined(SNERF_ABLATE) && (ABL & 4)     // timing-only: no ring at all
    return;
#endif
#if defined(SNERF_X2_OLD_LOADS) && !defined(SNERF_STAMP)
    if (PHASE == 0) {
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((RING2_D - 3) * 4) : "memory");
        dma_chunk4(stream, rg.goff, lds, rg.wr, wave, lane);
    } else {
        asm volatile("s_barrier" ::: "memory");      // the loading waves waited for their pieces before they arrived here
    }
    rg.goff += kChunkBytes;
    if (rg.goff >= stream_bytes) rg.goff = 0;
    rg.wr = ring2_next(rg.wr);
    return;
#endif
#ifdef SNERF_STAMP
    const unsigned long long t0_ = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING2_D - 3) * PIECES2) : "memory");
    const unsigned long long t1_ = __builtin_amdgcn_s_memtime();
    asm volatile("s_barrier" ::: "memory");
    const unsigned long long t2_ = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && lane == 0) { g_stamps[wave * 64 + 62] += t1_ - t0_; g_stamps[wave * 64 + 63] += t2_ - t1_; }
#else
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((RING2_D - 3) * PIECES2) : "memory");
#endif
    dma_chunk2(stream, rg.goff, lds, rg.wr, wave, lane);
    rg.goff += kChunkBytes;
    if (rg.goff >= stream_bytes) rg.goff = 0;
    rg.wr = ring2_next(rg.wr);
}
// wave w moves pieces 2w and 2w + 1 of the chunk (see dma_chunk in mlp_device.h for why this is inline asm)
__device__ __forceinline__ void dma_chunk2(const uint8_t* stream, uint32_t goff, lds_char* lds, uint32_t wr, int wave, int lane) {
    const uint8_t* b0 = stream + goff + wave * (PIECES2 * kFragBytes);
    const uint32_t dst = (uint32_t)(uintptr_t)(lds + wr + wave * (PIECES2 * kFragBytes));
    const uint32_t voff = lane * 16;
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %3\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %4\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(dst), "s"(b0), "s"(b0 + kFragBytes)
        : "memory", "scc");
}

// wave w (0..3) moves pieces 4w .. 4w+3
__device__ __forceinline__ void dma_chunk4(const uint8_t* stream, uint32_t goff, lds_char* lds, uint32_t wr, int wave, int lane) {
    const uint8_t* b0 = stream + goff + wave * (4 * kFragBytes);
    const uint32_t dst = (uint32_t)(uintptr_t)(lds + wr + wave * (4 * kFragBytes));
    const uint32_t voff = lane * 16;
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %3\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %4\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %5\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %6\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(dst), "s"(b0), "s"(b0 + kFragBytes), "s"(b0 + 2 * kFragBytes), "s"(b0 + 3 * kFragBytes)
        : "memory", "scc");
}

// per-row scale and bias of accumulator elements 4g .. 4g+3 of block b (table layout: [16 scales | 16 biases] per block and lane-half)
struct TabQ {
    f32x4 sc, bi;
};
// per-row scale and bias from the layer's table start (block b, lane-half h, quad g)
__device__ __forceinline__ TabQ load_tab_quad(lds_cfloat* tab_l, int b, int h, int g) {
    typedef volatile const __attribute__((address_space(3))) f32x4 lds_vf32x4;      // volatile: see load_tab_quad_h
    lds_vf32x4* tp = (lds_vf32x4*)(tab_l + (b * 2 + h) * 32);
    TabQ t;
    t.sc = tp[g];
    t.bi = tp[4 + g];
    return t;
}
// tab_h: a per-lane base that already points at this lane-half's rows of block 0 (tab_l + 32 h): block b, quad g are then IMMEDIATE offsets
__device__ __forceinline__ TabQ load_tab_quad_h(lds_cfloat* tab_h, int b, int g) {
    // volatile: the table is loop-invariant data, and the optimiser otherwise gathers the loads of whole layers ahead of
    // the chain and parks them in scratch (1.7 KB per lane measured) - they must stay where the epilogue needs them
    typedef volatile const __attribute__((address_space(3))) f32x4 lds_vf32x4;
    lds_vf32x4* tp = (lds_vf32x4*)(tab_h + b * 64);
    TabQ t;
    t.sc = tp[g];
    t.bi = tp[4 + g];
    return t;
}
// One fused layer (see run_layer8 in kernels_i8.hip for the arithmetic).  Chunk boundaries of the read sequence are compile-time
// positions; every layer starts on one.  The whole layer is ONE basic block for the compiler (no branch): with branches inside
// it hipcc sinks every block's epilogue to the end of the layer and spills the accumulators meanwhile (measured, 2.4 KB scratch).
//   RAWL: the layer reads an encoding; rawx = its three raw coordinates, added in fp32 (table behind the scale / bias table).
//   PHASE (0 / 1) = the wave group (waves 0-3 / 4-7: waves w and w + 4 share a SIMD).  The matrix pipe is arbitrated by age: left
//   alone, the older wave of a SIMD takes every MFMA slot it wants, reaches the chunk barrier early and waits there for the
//   younger one (in-kernel stamps, -DSNERF_STAMP: wave 0 spent 30 % of the launch inside s_barrier, wave 4 3 %).  So the
//   priority alternates per k-step (s_setprio 3 in the k-steps of one's own parity, 0 in the others): wave 0's barrier time
//   drops to 13 %, the kernel gains 2.5 %.  Tried on top without gain: the epilogue slices in the k-steps where a wave yields,
//   biased levels ({0,2} against {1,3}: the younger wave then takes the older one's place), alternating the tie-break per
//   k-step pair / per block (the two waves' barrier times even out, their sum and the kernel time do not change).
template <int NB, int KS0, int KS1, bool SIN, bool RAWL = false, int PHASE = 0, bool OPQ = false>
__device__ __forceinline__ void run_layer8x2(Ring2& rg, const uint8_t* stream, uint32_t stream_bytes, lds_char* lds, lds_cfloat* tab_l,
                                             const Frag8* in0, const Frag8* in1, Frag8* out, f32x16* raw, int wave, int lane,
                                             const float* rawx = nullptr) {
    constexpr int KS = KS0 + KS1, NP = NB * KS;
    typedef volatile const __attribute__((address_space(3))) f32x4 lds_vf32x4;
    f32x4 rq[4][3];        // raw-coordinate weights of the previous block's quads (RAWL), requested with the table entries
    const int h = lane >> 5;
    // OPQ (the ray-visibility variant): table addresses from ONE opaque per-lane base per layer (this lane-half's rows of block 0), every block / quad an
    // immediate offset from it.  Written as absolute addresses (LDS base + layer start + block + h-term: constants above the 64 KiB reach of a ds_read
    // offset) hipcc gave every table row of THAT variant its own address register and hoisted all of them out of the persistent tile loop - ~25 registers
    // alive across the whole chain, 72-104 bytes of scratch (tests/test_isa_guards.py); the asm makes the base un-hoistable, so it lives for one layer.
    // The other variants compile without scratch as they are and keep their addressing (with the opaque bases the full program spills 116 bytes).
    lds_cfloat* raw_l = tab_l + 2 * 32 * NB;
    lds_cfloat* tab_h = tab_l + 32 * h;
    lds_cfloat* raw_h = raw_l + 48 * h;
    if constexpr (OPQ) asm volatile("" : "+v"(tab_h), "+v"(raw_h));
    auto tabq = [&](int b_, int g_) { return OPQ ? load_tab_quad_h(tab_h, b_, g_) : load_tab_quad(tab_l, b_, h, g_); };
    i32x4 fT[PFX], fL[PFX];
#if defined(SNERF_ABLATE) && (ABL & 2)     // timing-only: weight fragments stay in registers, no LDS reads
    constexpr bool ABL2_ = true;
#pragma unroll
    for (int q = 0; q < PFX; ++q) { fT[q] = i32x4{lane, 1, 2, 3}; fL[q] = i32x4{4, lane, 6, 7}; }
#else
    constexpr bool ABL2_ = false;
#endif
#define REQUEST(QN, SLOT)                                                                              \
    do {                                                                                               \
        if ((QN) % kChunkPairs == 0) {                                                                 \
            ring_step2<PHASE>(rg, stream, stream_bytes, lds, wave, lane);                                \
            if ((QN) > 0) rg.cur = ring2_next(rg.cur);                                                 \
        }                                                                                              \
        lds_char* ap_ = lds + rg.cur + ((QN) % kChunkPairs) * kPairBytes + lane * 16;                  \
        if (!ABL2_) {                                                                                  \
            fT[SLOT] = *(lds_ci32x4*)ap_;                                                              \
            fL[SLOT] = *(lds_ci32x4*)(ap_ + kFragBytes);                                               \
        } else {                                                                                       \
            asm volatile("" : "+v"(fT[SLOT]), "+v"(fL[SLOT]));                                         \
        }                                                                                              \
    } while (0)
#pragma unroll
    for (int q = 0; q < PFX; ++q) {
        if (q < NP) REQUEST(q, q);
    }
    // Software pipeline: when block b starts, the finished accumulators of block b-1 are merged into m[i] = (M << 8) + X
    // (16 registers instead of 32: no second accumulator set), and their epilogue - a quad of elements per slot - runs
    // inside block b's k-steps (quad g at k-step g KS / 4, its table entries requested at the top of that step).  Each
    // wave's stream is then a uniform mix of MFMA and VALU work, so it does not matter that the two waves of a SIMD run
    // in step: whichever has an MFMA ready feeds the pipe.
    Acc8 acc;
    int m[16];
    TabQ tq[4];        // table entries of the previous block's quads: requested one k-step before their slice runs
    auto load_rq = [&](int g, int b_of) {
        if constexpr (RAWL) {
            lds_vf32x4* tp = OPQ ? (lds_vf32x4*)(raw_h + (b_of * 8 + g) * 12) : (lds_vf32x4*)(raw_l + ((b_of * 2 + h) * 4 + g) * 12);
            rq[g][0] = tp[0]; rq[g][1] = tp[1]; rq[g][2] = tp[2];
        }
    };
    auto quad = [&](int g, int b_of, const TabQ& t) {       // epilogue of elements 4g .. 4g+3 of block b_of from m[]
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = __builtin_fmaf((float)m[4 * g + j], t.sc[j], t.bi[j]);
        if constexpr (RAWL) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                v[j] = __builtin_fmaf(rq[g][0][j], rawx[0], __builtin_fmaf(rq[g][1][j], rawx[1], __builtin_fmaf(rq[g][2][j], rawx[2], v[j])));
        }
        if (SIN) {
            int hi, lo;
            digits4(sin2pi(v[0]), sin2pi(v[1]), sin2pi(v[2]), sin2pi(v[3]), hi, lo);
            asm volatile("" : "+v"(hi), "+v"(lo));       // the digits exist HERE (nothing may sink the epilogue towards their first use)
            out[b_of].hi[g] = hi;
            out[b_of].lo[g] = lo;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) (*raw)[4 * g + j] = v[j];
        }
    };
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (b > 0) {
            tq[0] = tabq(b - 1, 0);
            load_rq(0, b - 1);
#pragma unroll
            for (int i = 0; i < 16; ++i) m[i] = (int)(((uint32_t)acc.M[i] << 8) + (uint32_t)acc.X[i]);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc.M[i] = 0; acc.X[i] = 0; }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int q = b * KS + s;
            const i32x4 aT = fT[q % PFX], aL = fL[q % PFX];
#ifndef SNERF_X2_NOPRIO
            // own parity: this wave's MFMAs first (it runs no epilogue slice in this k-step); else yield to the partner
            if (((s + PHASE) & 1) == 0) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(3);
#endif
            if (q + PFX < NP) REQUEST(q + PFX, q % PFX);
            if (b > 0) {
#pragma unroll
                for (int g = 1; g < 4; ++g) {
                    const int sg = (g * KS) / 4, lg = sg > 0 ? sg - 1 : 0;
                    if (lg == s) { tq[g] = tabq(b - 1, g); load_rq(g, b - 1); }
                }
            }
            mfma_i8x3(aT, aL, s < KS0 ? in0[s] : in1[s - KS0], acc);
            if (b > 0) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if ((g * KS) / 4 == s) quad(g, b - 1, tq[g]);
            }
            asm volatile("" ::: "memory");          // table loads stay in their k-step (the optimiser would gather them up front)
#ifdef SNERF_X2_GROUPS
            // interleave inside the k-step: each MFMA followed by its share of the epilogue slice, so that two waves running in
            // step still alternate on the matrix pipe instead of meeting in a VALU-only stretch
#pragma unroll
            for (int mm = 0; mm < 3; ++mm) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // DS read
                __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);      // transcendental
                __builtin_amdgcn_sched_group_barrier(0x002, SNERF_X2_GROUPS, 0);      // VALU
            }
#endif
            __builtin_amdgcn_sched_barrier(0);      // ... and so do requests, MFMAs and epilogue slices (register pressure)
        }
    }
    // the last block of the layer: nothing to hide its epilogue behind but the partner wave
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = (int)(((uint32_t)acc.M[i] << 8) + (uint32_t)acc.X[i]);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        asm volatile("" ::: "memory");
        const TabQ t = tabq(NB - 1, g);
        load_rq(g, NB - 1);
        quad(g, NB - 1, t);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    rg.cur = ring2_next(rg.cur);          // the next layer starts on the next chunk
#undef REQUEST
}

template <int W, int VARIANT, int PHASE>
__device__ __forceinline__ void field_tiles2(const MlpArgs& A, Ring2& rg, lds_char* lds, __attribute__((address_space(3))) float* tab_lds,
                                             int wave, int lane);

template <int W, int VARIANT>
__global__ __launch_bounds__(64 * NW2, 1) void mlp_i8x2_kernel(const MlpArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int C_MAX = kMaxClasses;
    constexpr int W2 = W / 2;
    lds_char* lds = (lds_char*)smem;
    __attribute__((address_space(3))) float* tab_lds = (__attribute__((address_space(3))) float*)(lds + RING2_D * kChunkBytes);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5;
    const int C = A.n_classes;

    for (int i = threadIdx.x; i < A.bias_floats; i += 64 * NW2) tab_lds[i] = A.bias[i];

    Ring2 rg;
    rg.cur = 0;
    rg.goff = 0;
    {
        uint32_t wr = 0;
#pragma unroll
        for (int c = 0; c < RING2_D - 2; ++c) {
#if defined(SNERF_X2_OLD_LOADS) && !defined(SNERF_STAMP)
            if (wave < 4) dma_chunk4(A.stream, rg.goff, lds, wr, wave, lane);
#else
            dma_chunk2(A.stream, rg.goff, lds, wr, wave, lane);
#endif
            rg.goff += kChunkBytes;
            if (rg.goff >= A.stream_bytes) rg.goff = 0;
            wr += kChunkBytes;
        }
        rg.wr = wr;
    }
    __syncthreads();   // table visible (drains the prologue DMAs once; harmless)

    if (wave < 4) field_tiles2<W, VARIANT, 0>(A, rg, lds, tab_lds, wave, lane);
    else field_tiles2<W, VARIANT, 1>(A, rg, lds, tab_lds, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
}

// the persistent tile loop of one wave group (PHASE: see run_layer8x2); the two groups run two copies of the code
template <int W, int VARIANT, int PHASE>
__device__ __forceinline__ void field_tiles2(const MlpArgs& A, Ring2& rg, lds_char* lds, __attribute__((address_space(3))) float* tab_lds,
                                             int wave, int lane) {
    constexpr int C_MAX = kMaxClasses;
    constexpr int W2 = W / 2;
    const int h = lane >> 5;
    const int C = A.n_classes;
    // VARIANT 3 (ray visibility, mlp_device.h RaySum): a "tile" is a group of NW2 rays, walked in `passes` steps of 32 samples
    const int64_t n_tiles = VARIANT == 3 ? (A.n + NW2 - 1) / NW2 : (A.n + TILE2 - 1) / TILE2;
    const int passes = VARIANT == 3 ? (A.n_samples + 31) / 32 : 1;
    int pass = 0;
    RaySum rs;
    for (int64_t tile = blockIdx.x; tile < n_tiles;) {
        const int64_t n = tile * TILE2 + wave * 32 + (lane & 31);
        const bool valid = n < A.n;
        const int64_t nc = valid ? n : A.n - 1;
        const int64_t g = VARIANT == 3 ? 0 : nc / A.group_size;

        // ---- sample position (misc.py:234-247 fused): top*(1-t) + bot*t, two roundings + one add, no fma
        float x0, x1, x2;
        if constexpr (VARIANT == 3) {
            raysum_point(rs, A, tile, NW2, wave, pass, lane, x0, x1, x2);
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
#ifdef SNERF_STAMP
        const bool stamp_on = blockIdx.x == 0 && tile == (int64_t)blockIdx.x + gridDim.x;      // the workgroup's second tile
#endif
        STAMP(0);
        Frag8 pe[PEPOS_KS8];
        make_pe_pos8<VARIANT == 3>(x0, x1, x2, h, pe);
        STAMP(1);

        constexpr int KW = W / 32, KW2 = W2 / 32;
        Frag8 hA[KW], hB[KW];
        f32x16 raw;
#define LAYER(L, NBv, K0, K1, SINv, IN0, IN1, OUT, RAW)                                                                       \
    run_layer8x2<NBv, K0, K1, SINv, false, PHASE, VARIANT == 3>(rg, A.stream, A.stream_bytes, lds, tab_lds + prog_table_start(PROG_FIELD, W, C_MAX, L), \
                                                  IN0, IN1, OUT, RAW, wave, lane)
#define LAYER_RAW(L, NBv, K0, K1, IN0, IN1, OUT, RX)                                                                          \
    run_layer8x2<NBv, K0, K1, true, true, PHASE, VARIANT == 3>(rg, A.stream, A.stream_bytes, lds, tab_lds + prog_table_start(PROG_FIELD, W, C_MAX, L), \
                                                 IN0, IN1, OUT, nullptr, wave, lane, RX)
        const float rx_p[3] = {x0, x1, x2}, rx_s[3] = {s0, s1, s2};      // raw coordinates: fp32, no digit range
        // trunk (G_NeRF.py:80-91)
        LAYER_RAW(F_FC1, W / 32, PEPOS_KS8, 0, pe, nullptr, hA, rx_p); STAMP(2);
        LAYER(F_FC2, W / 32, KW, 0, true, hA, nullptr, hB, nullptr); STAMP(3);
        LAYER(F_FC3, W / 32, KW, 0, true, hB, nullptr, hA, nullptr); STAMP(4);
        LAYER(F_FC4, W / 32, KW, 0, true, hA, nullptr, hB, nullptr); STAMP(5);
        LAYER_RAW(F_FC5, W / 32, KW, PEPOS_KS8, hB, pe, hA, rx_p); STAMP(6);
        LAYER(F_FC6, W / 32, KW, 0, true, hA, nullptr, hB, nullptr); STAMP(7);
        LAYER(F_FC7, W / 32, KW, 0, true, hB, nullptr, hA, nullptr); STAMP(8);
        LAYER(F_FC8, W / 32, KW, 0, true, hA, nullptr, hB, nullptr); STAMP(9);
        Frag8 x1f[KW2];
        LAYER(F_FC9, W2 / 32, KW, 0, true, hB, nullptr, x1f, nullptr); STAMP(10);
        // sigma / colour head (G_NeRF.py:93-98): regs 0..2 colour, 3 density (lane-half 0)
        LAYER(F_HEAD, 1, KW2, 0, false, x1f, nullptr, nullptr, &raw); STAMP(11);
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
            LAYER_RAW(F_S1, W2 / 32, KW2, PESUN_KS8, x1f, ps, sA, rx_s); STAMP(12);
            LAYER(F_S2, W2 / 32, KW2, 0, true, sA, nullptr, sB, nullptr); STAMP(13);
            LAYER(F_S3, W2 / 32, KW2, 0, true, sB, nullptr, sA, nullptr); STAMP(14);
            LAYER(F_S4, 1, KW2, 0, false, sA, nullptr, nullptr, &raw); STAMP(15);
            sv_raw = raw[0];
        }
        if constexpr (VARIANT == 0) {
            // seasonal colour-adjust branch (T_NeRF_net_v2.py:83-87)
            LAYER(F_A1, W / 32, KW2, 0, true, x1f, nullptr, hA, nullptr); STAMP(16);
            LAYER(F_A2, W / 32, KW, 0, true, hA, nullptr, hB, nullptr); STAMP(17);
            LAYER(F_A3, W / 32, KW, 0, true, hB, nullptr, hA, nullptr); STAMP(18);
            LAYER(F_AC, 1, KW, 0, false, hA, nullptr, nullptr, &raw); STAMP(19);
#pragma unroll
            for (int i = 0; i < 3 * C_MAX; ++i) adj[i] = raw[i];
        }
#undef LAYER
#undef LAYER_RAW
        if constexpr (VARIANT == 3) {
            raysum_add(rs, A, tile, NW2, wave, pass, lane, rho_raw, x0, x1, x2);
            if (++pass == passes || raysum_saturated(rs, A, tile * NW2 + wave, wave, NW2, lane, (__attribute__((address_space(3))) float*)(tab_lds + A.bias_floats))) {
                raysum_end(rs, A, tile, NW2, wave, lane);
                pass = 0;
                tile += gridDim.x;
            }
        } else {
            if (h == 0 && valid) store_field_outputs<VARIANT>(A.out, n, C, x0, x1, x2, col_r, col_g, col_b, rho_raw, sv_raw, adj, pcls);
            tile += gridDim.x;
        }
        STAMP(20);
    }
    __builtin_amdgcn_s_setprio(0);
}

template <int W, int VARIANT>
static hipError_t launch_mlp_i8x2_t(const MlpArgs& a, int n_cu, hipStream_t st) {
    const int lds_bytes = RING2_D * kChunkBytes + a.bias_floats * 4 + kVoteBytes;
    const int64_t n_tiles = VARIANT == 3 ? (a.n + NW2 - 1) / NW2 : (a.n + TILE2 - 1) / TILE2;
    int grid = (int)(n_tiles < n_cu ? n_tiles : n_cu);
    if (grid < 1) grid = 1;
    auto k = mlp_i8x2_kernel<W, VARIANT>;
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * NW2), lds_bytes, st, a);
    return hipGetLastError();
}

hipError_t launch_mlp_i8x2(int W, int variant, const MlpArgs& a, int n_cu, hipStream_t st) {
#define CASE(Wv)                                                              \
    if (W == Wv) {                                                            \
        if (variant == 0) return launch_mlp_i8x2_t<Wv, 0>(a, n_cu, st);       \
        if (variant == 1) return launch_mlp_i8x2_t<Wv, 1>(a, n_cu, st);       \
        if (variant == 3) return launch_mlp_i8x2_t<Wv, 3>(a, n_cu, st);       \
        return launch_mlp_i8x2_t<Wv, 2>(a, n_cu, st);                         \
    }
    CASE(64)
    CASE(256)
#undef CASE
    return hipErrorInvalidValue;
}

}  // namespace snerf

#ifdef SNERF_STAMP
extern "C" int snerf_debug_stamps(unsigned long long* host, int n) {
    if (n > 8 * 64) n = 8 * 64;
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(snerf::g_stamps), (size_t)n * 8);
    unsigned long long zero[8 * 64] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(snerf::g_stamps), zero, sizeof(zero));
    return e == hipSuccess ? 0 : -3;
}
#endif
