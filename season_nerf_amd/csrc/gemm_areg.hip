// gfx950 (MI355X / CDNA4): row GEMM of the layer-wise engine for WIDE layers (N = 256 / 512 outputs, any K) - output-stationary, accumulators in AGPRs.
//
//   C[M, N] = alpha * (act(A)[M, K] x Bt[N, K]^T) + alpha * bias      bf16x3 products (lo*hi + hi*lo + hi*hi, fp32 accumulate) as gemm.hip / gemm16.hip
//
// Why a third row GEMM.  The reference's DEFAULT width is 512 (main_lite.py:80).  There the two kernels of gemm.hip / gemm16.hip keep a 64-column slice of
// the split weights resident in LDS (64 x 512 x 4 B = 128 KiB is all 160 KiB hold), so eight column groups read every A row - and redo its activation
// (fma + v_sin) and bf16 hi / lo split - EIGHT times: 931-1012 us per 512 -> 512 layer against a ~300 us copy floor, vector-issue bound.
// Here a wave owns 32 rows x ALL N columns: its 32 x 512 accumulator tile IS the 256 AGPRs (16 n-tiles x 16 registers, addressed by number: the MFMAs
// are issued through inline asm with C / D = a[16 T : 16 T + 15]), and BOTH operands stream:
//   * A: one k-step (16 k = two 16-byte loads per lane) at a time, PFA k-steps in flight in staging registers, activated (AOL: fma + v_sin against the
//     [a | b] table in LDS) and split into bf16 hi / lo ONCE per element, used by all N / 32 n-tiles of that k-step; the stream runs on into the next row
//     tile without a seam, so loads, conversion work and HBM demand are spread evenly over the whole tile.  (A first version kept the ACTIVATIONS resident
//     in the AGPRs and walked the n-tiles: it had to fetch and convert the next tile's A inside its last n-tile - a 14 TB/s burst chip-wide, 110 us of a
//     785 us layer by ablation - and its dependent three-MFMA chains left no room to hide anything: profiles/r5/areg_ablation.txt.)
//   * weights: L2 -> LDS through the 16 KiB ring of the fused kernels (LDS-DMA, one counted vmcnt wait + one barrier per chunk), k-major
//     ([k-step][n-tile] pairs: areg_split_weights_kernel), every fragment shared by the four waves of the workgroup;
//   * n-tiles in pairs: six MFMAs (lo*hi, hi*lo, hi*hi of two accumulators, interleaved: no MFMA waits for the one before it), the next pair's
//     fragments requested one pair ahead, and between the pairs the slices of the NEXT k-step's conversion;
//   * epilogue (bias, BatchNorm column sums, stores through a buffer descriptor: no address arithmetic, rows past M dropped by the bounds check) of row
//     tile t inside the FIRST k-step of tile t+1: each n-tile's 16 accumulators are read out right before that k-step's first MFMA restarts them (C = 0);
//     ACT (input gradients): the activation-backward epilogue of gemm_rows_full_kernel - times cos(2 pi (a z + b)) of the layer below, its pre-activations
//     loaded a pair of n-tiles ahead - and the column sums sum v, sum v xhat;
//   * one workgroup (4 waves = one per SIMD, all 512 registers) per CU, persistent over row tiles of 128 rows;
//   * HV = 2 (the forward forms at N = 512): EIGHT waves - waves w and w + 4 share a row group and own a column half each (128 AGPRs, 128 architectural
//     registers), convert every k-step of A once between them (16 rows each) and exchange the bf16 operands through LDS behind the ring's barriers.  The
//     second wave fills the matrix pipe (MFMAs alone: 277 us instead of 380) - and the kernel as a whole does not move, because every complete build of it
//     runs at the package power limit (tools/areg_power.py): -4 % with activation on load, 0 without.
// Same products, same k order and the same order of the three partial products per accumulator as gemm_rows_full_kernel (tools/areg_check.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "gemm_common.h"
#include "mlp_device.h"
#include "train.h"

#ifndef SNERF_ABLA
#define SNERF_ABLA 0       // timing-only ablations of scratch builds (tools/variants.py): 1 no stores, 2 no A stream, 4 no LDS weight reads, 8 no ring, 16 no MFMAs
#endif
#ifndef SNERF_AR_D
#define SNERF_AR_D 7
#endif

namespace snerf {

constexpr int AR_WAVES = 4, AR_ROWS = 32 * AR_WAVES;
constexpr int AR_D = SNERF_AR_D;              // ring slots (16 KiB chunks of 8 weight pairs)
static_assert(kChunkPairs == 8 && DMA_PER_WAVE == 4, "ring arithmetic below assumes 8-pair chunks moved by four waves");

// k-major fragment stream: pair (ks, T) = 1 KiB hi then 1 KiB lo; inside, lane (r, h) owns 16 bytes = bf16 of Bt[32 T + r][16 ks + 8 h + 0..7]
// (the lane layout of split_weights_kernel, gemm.hip; only the order of the pairs differs: there n-tile-major).  Bt[n][k] = W[n][k] or W[k][n] (transpose).
__global__ void areg_split_weights_kernel(const float* W, int rows, int cols, int transpose, uint16_t* frag, int n_tiles, int ksteps) {
    const int64_t total = (int64_t)n_tiles * ksteps * 512;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const int64_t tk = i >> 9;
        const int T = (int)(tk % n_tiles), ks = (int)(tk / n_tiles);
        const int n = T * 32 + (lane & 31), k = ks * 16 + (lane >> 5) * 8 + e;
        float v = 0.f;
        if (!transpose) { if (n < rows && k < cols) v = W[(int64_t)n * cols + k]; }
        else { if (k < rows && n < cols) v = W[(int64_t)k * cols + n]; }
        const __bf16 h = (__bf16)v;
        const __bf16 l = (__bf16)(v - (float)h);
        uint16_t* dst = frag + tk * 1024 + lane * 8 + e;
        dst[0] = __builtin_bit_cast(uint16_t, h);
        dst[512] = __builtin_bit_cast(uint16_t, l);
    }
}

#define AR8(n) "a" #n
#define AR8x8(n) AR8(n##0), AR8(n##1), AR8(n##2), AR8(n##3), AR8(n##4), AR8(n##5), AR8(n##6), AR8(n##7), AR8(n##8), AR8(n##9)
__device__ __forceinline__ void ar_reserve_agprs128() {   // two waves per SIMD: 128 AGPRs per wave, all of them accumulators
    asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", AR8x8(1), AR8x8(2), AR8x8(3), AR8x8(4), AR8x8(5), AR8x8(6),
                 AR8x8(7), AR8x8(8), AR8x8(9), AR8x8(10), AR8x8(11), "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127");
}
__device__ __forceinline__ void ar_reserve_agprs() {      // the compiler must allocate none of a0..a255 (see kernels_i8.hip reserve_agprs)
    asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", AR8x8(1), AR8x8(2), AR8x8(3), AR8x8(4), AR8x8(5), AR8x8(6),
                 AR8x8(7), AR8x8(8), AR8x8(9), AR8x8(10), AR8x8(11), AR8x8(12), AR8x8(13), AR8x8(14), AR8x8(15), AR8x8(16), AR8x8(17), AR8x8(18),
                 AR8x8(19), AR8x8(20), AR8x8(21), AR8x8(22), AR8x8(23), AR8x8(24), "a250", "a251", "a252", "a253", "a254", "a255");
}
// a[base : base + 15] (+)= A x B   (A: 32 rows x 16 k of activations, B: 16 k x 32 columns of weights, both in VGPRs)
template <bool FIRST>
__device__ __forceinline__ void ar_mfma(int base, const u32x4& a, const u32x4& b) {
    if (SNERF_ABLA & 16) { asm volatile("" ::"v"(a), "v"(b)); return; }
    if (FIRST) asm volatile("v_mfma_f32_32x32x16_bf16 a[%2:%3], %0, %1, 0" ::"v"(a), "v"(b), "i"(base), "i"(base + 15));
    else asm volatile("v_mfma_f32_32x32x16_bf16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(a), "v"(b), "i"(base), "i"(base + 15));
}
__device__ __forceinline__ float ar_read(int idx) {
    float v;
    asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(v) : "i"(idx));
    return v;
}

// hand-issued A loads: the 8 consecutive k of this lane's row for one k-step (two 16-byte loads); completion is awaited by count
__device__ __forceinline__ void ar_issue(const float* p, f32x4& x, f32x4& y) {
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16" : "=&v"(x), "=&v"(y) : "v"(p));
}
template <int N>
__device__ __forceinline__ void ar_wait(f32x4& x, f32x4& y) {
    asm volatile("s_waitcnt vmcnt(%2) ; ar_wait %0 %1" : "+v"(x), "+v"(y) : "n"(N > 63 ? 63 : N));
}

// HV = 2: a lane stages FOUR consecutive k of one row per k-step (one 16-byte load): the wave pair that shares 32 rows splits them, 16 rows each
__device__ __forceinline__ void ar_issue1(const float* p, f32x4& x) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(x) : "v"(p));
}
template <int N>
__device__ __forceinline__ void ar_wait1(f32x4& x) {
    asm volatile("s_waitcnt vmcnt(%1) ; ar_wait1 %0" : "+v"(x) : "n"(N > 63 ? 63 : N));
}

struct ArRing {
    uint32_t rd, wr, cur, goff;
};
// ring_step of mlp_device.h; YOUNGER = vector-memory operations of this wave issued after the chunk handed over here (loads, stores and LDS-DMA are
// counted together and retire in issue order, MI355X_MICROARCH.md): a count BELOW the true number only waits for more, never for less
// a chunk's sixteen 1 KiB pieces over 4 HV waves (HV = 2: wave w moves pieces w and w + 8)
template <int HV>
__device__ __forceinline__ void ar_dma_chunk(const uint8_t* stream, uint32_t goff, lds_char* lds, uint32_t wr, int wave, int lane) {
    if (HV == 1) { dma_chunk(stream, goff, lds, wr, wave, lane); return; }
    const uint8_t* b0 = stream + goff + wave * kFragBytes;                    // wave-uniform
    const uint32_t dst = (uint32_t)(uintptr_t)(lds + wr + wave * kFragBytes); // wave-uniform LDS byte address
    const uint32_t voff = lane * 16;
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %3\n\t"
        "s_add_u32 m0, m0, 0x2000\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %4\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(dst), "s"(b0), "s"(b0 + 8 * kFragBytes)
        : "memory", "scc");
}
template <int YOUNGER, int HV = 1>
__device__ __forceinline__ void ar_ring_step(ArRing& rg, const uint8_t* stream, uint32_t stream_bytes, lds_char* lds, int wave, int lane) {
    if (SNERF_ABLA & 8) return;
    if (HV == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(YOUNGER > 63 ? 63 : YOUNGER) : "memory");      // + the operand exchange's LDS writes
    else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(YOUNGER > 63 ? 63 : YOUNGER) : "memory");
    // (the DMA statement takes its addresses in SGPRs: hipcc keeps `wave` in a vector register in this kernel unless told again that it is uniform)
    ar_dma_chunk<HV>(stream, __builtin_amdgcn_readfirstlane(rg.goff), lds, __builtin_amdgcn_readfirstlane(rg.wr), __builtin_amdgcn_readfirstlane(wave), lane);
    rg.goff += kChunkBytes;
    if (rg.goff >= stream_bytes) rg.goff = 0;
    rg.cur = rg.rd;
    rg.rd = ring_next<AR_D>(rg.rd);
    rg.wr = ring_next<AR_D>(rg.wr);
}

struct ArFrag {       // weight fragments of a pair of n-tiles
    u32x4 h0, l0, h1, l1;
};
struct ArTile {       // what the epilogue of a row tile needs
    int64_t row0;     // first row of the wave's 32 rows
    bool on;
};
struct ArStream {     // the A stream of a wave: the next k-step to request
    const float* p;   // this lane's row of the tile being requested (+ 8 h)
    int64_t t;        // that tile
    int ks;           // k-step to request next
};

// epilogue of n-tile T of the previous row tile: accumulators read out of a[16 T ..] (the caller restarts them right after).
// ACT (input gradients): the value is dL/dH of the SineLayer below - multiplied by cos(2 pi (a z + b)) of that layer's pre-activation z (staged by the
// caller: zs[i] = ez[row i][column]) it becomes dL/d(arg); the column sums are sum v and sum v * xhat, xhat = (z - mu) istd (see gemm_rows_full_kernel).
template <int ACT>
__device__ __forceinline__ void ar_epilogue(const GemmX& g, int T, const __amdgpu_buffer_rsrc_t& rs_c, int lc, int so0, int rows_left, bool on, lds_cfloat* col_l,
                                            __attribute__((address_space(3))) float* stat_l, int r, int h, const float* zs = nullptr) {
    const int N = (int)g.N;
    const float ab = ACT ? 0.f : g.alpha * col_l[32 * T + r];
    const float ea = ACT ? stat_l[2 * N + 32 * T + r] : 0.f, eb = ACT ? stat_l[3 * N + 32 * T + r] : 0.f;
    const float mu = ACT ? stat_l[4 * N + 32 * T + r] : 0.f, istd = ACT ? stat_l[5 * N + 32 * T + r] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    switch (T) {
#define AR_EPI(T_) case T_: {                                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                                                                                    \
            const int ro = (i & 3) + 8 * (i >> 2);                                                                                                          \
            const float a = ar_read(16 * T_ + i);                                                                                                           \
            float v, w;                                                                                                                                     \
            if (ACT) {                                                                                                                                      \
                const float z = zs[i];                                                                                                                      \
                v = g.alpha * a * __builtin_amdgcn_cosf(__builtin_fmaf(ea, z, eb));                                                                         \
                w = v * ((z - mu) * istd);                                                                                                                  \
            } else {                                                                                                                                        \
                v = __builtin_fmaf(g.alpha, a, ab);                                                                                                         \
                w = a;                                                                                                                                      \
            }                                                                                                                                               \
            if (!(SNERF_ABLA & 1))                                                                                                                          \
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rs_c, lc, so0 + (ro * (int)g.ldc + 32 * T_) * 4, 0);                   \
            const bool ok = ro < rows_left;                                                                                                                 \
            if (ACT) { s1 += ok ? v : 0.f; s2 += ok ? w : 0.f; }                                                                                            \
            else { const float am = ok ? a : 0.f; s1 += am; s2 = __builtin_fmaf(am, am, s2); }                                                              \
        } } break;
        AR_EPI(0) AR_EPI(1) AR_EPI(2) AR_EPI(3) AR_EPI(4) AR_EPI(5) AR_EPI(6) AR_EPI(7) AR_EPI(8) AR_EPI(9) AR_EPI(10) AR_EPI(11) AR_EPI(12) AR_EPI(13) AR_EPI(14) AR_EPI(15)
#undef AR_EPI
        default: break;
    }
    if (g.stats) {      // column sums: the two lane-halves hold the same column (rows differ), the four waves too
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (h == 0 && on) {
            __builtin_amdgcn_ds_faddf(stat_l + 32 * T + r, s1, __ATOMIC_RELAXED, __MEMORY_SCOPE_WRKGRP, false);      // ds_add_f32: a flat atomic would count in vmcnt
            __builtin_amdgcn_ds_faddf(stat_l + N + 32 * T + r, s2, __ATOMIC_RELAXED, __MEMORY_SCOPE_WRKGRP, false);
        }
    }
}
// ACT: the pre-activations of n-tile T of the row tile at byte offset sz0 (buffer descriptor over ez; rows past M read as 0): 16 loads per lane,
// compiler-issued - it waits for them by its own count of the vector-memory operations IT issued, which is below the true number (the hand-issued
// DMA and A loads are invisible to it): the wait is merely conservative
__device__ __forceinline__ void ar_load_z(const GemmX& g, const __amdgpu_buffer_rsrc_t& rs_z, int lz, int sz0, int T, float* zs) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int ro = (i & 3) + 8 * (i >> 2);
        zs[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_z, lz, sz0 + (ro * (int)g.eld + 32 * T) * 4, 0));
    }
}

// One k-step over all NT n-tiles, in NT / 2 pairs.  On entry `cur` holds the converted operands of this k-step and `fb0` the fragments of its first pair;
// on exit `nxt` those of the next k-step (converted here from staging slot `sx / sy`, which is refilled) and `fb0` the first pair of the next k-step.
// FIRST: the first k-step of a row tile - every accumulator is read out (the epilogue of the previous row tile `pv`) right before it is restarted.
// HV = 2 (two waves per SIMD): waves w and w + 4 share the rows 32 (w & 3) .. and split the columns - wave half `hf` owns the n-tiles hf NT .. hf NT + NT - 1
// (column base cb = 32 NT hf; col_l / stat_l arrive shifted by cb); a k-step's weights are then HV chunks, handed over back to back, each wave reading its own.
template <int NT, int AOL, int PFA, bool FIRST, int ACT = 0, int HV = 1>
__device__ __forceinline__ void ar_kstep(const GemmX& g, ArRing& rg, lds_char* lds, const uint8_t* stream, uint32_t stream_bytes, const u32x4& chi, const u32x4& clo,
                                         u32x4& nhi, u32x4& nlo, ArFrag& fb0, f32x4& sx, f32x4& sy, ArStream& as, int ks_next, int KS, lds_cfloat* tab_h,
                                         const ArTile& pv, lds_cfloat* col_l, __attribute__((address_space(3))) float* stat_l, int64_t n_tiles, int wave, int lane,
                                         int hf = 0, lds_char* xb = nullptr, int par = 0) {
    static_assert(HV == 1 || (HV == 2 && NT == kChunkPairs), "two column halves: one chunk of eight n-tiles each");
    constexpr int G = NT / 2;
    constexpr int DPW = DMA_PER_WAVE / HV;             // LDS-DMA instructions per wave and chunk
    constexpr int SPK = NT * HV / kChunkPairs;         // ring steps per k-step
    // HV = 2: r, h and everything derived from them are recomputed from a FRESH lane id here (an opaque v_mbcnt pair) - carried through the k-step loop they
    // are seven more live registers than the 128 a wave has, and hipcc parks the excess in the accumulators' AGPRs
    int lane_f = lane;
    if (HV == 2) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_f));
    const int r = lane_f & 31, h = lane_f >> 5;
    const int cb = HV == 2 ? 32 * NT * hf : 0;
    const int wrow = HV == 2 ? (wave & 3) : wave;
    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc((void*)g.C, 0, (int)(g.M * g.ldc * 4), 0x00020000);
    const int lc = (FIRST && pv.on) ? (int)((4 * h) * g.ldc + r + cb) * 4 : (int)0x80000000;
    const int so0 = FIRST ? (int)(pv.row0 * g.ldc * 4) : 0;
    const int rows_left = (FIRST && pv.on) ? (int)(g.M - pv.row0) - 4 * h : 0;      // element i is a real row iff (i & 3) + 8 (i >> 2) < rows_left
    // ACT: the pre-activations of the previous row tile, a pair of n-tiles ahead of the epilogue that needs them (64 staging registers)
    const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc((void*)(ACT ? g.ez : g.A), 0, ACT ? (int)(g.M * g.eld * 4) : 0, 0x00020000);
    const int lz = (FIRST && ACT && pv.on) ? (int)((4 * h) * g.eld + r + cb) * 4 : (int)0x80000000;
    const int sz0 = (FIRST && ACT) ? (int)(pv.row0 * g.eld * 4) : 0;
    float zt[2][HV == 2 ? 1 : 2][16];                  // HV = 2 (128 architectural registers): ONE n-tile ahead instead of a pair
    if (FIRST && ACT) {
        ar_load_z(g, rs_z, lz, sz0, 0, zt[0][0]);
        if constexpr (HV == 1) ar_load_z(g, rs_z, lz, sz0, 1, zt[0][1]);
    }
    ArFrag fb[2];
    fb[0] = fb0;
    float a8[8];
    f32x4 ta[2] = {}, tb[2] = {};                      // (HV = 2 reads the table entries where they are used: 16 registers less)
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
        const int T0 = 2 * gi, T1 = T0 + 1;
        ArFrag& f = fb[gi & 1];
        ArFrag& fn = fb[(gi + 1) & 1];
        // ---- request the next pair's fragments (the first pair of the next k-step behind the last one: the stream is k-major, contiguous and cyclic)
        {
            const int Tn = (T0 + 2) % NT;
            if (Tn % kChunkPairs == 0) {
                // younger than the chunk handed over (issued AR_D - 2 ring steps ago): the AR_D - 3 younger chunks' DMA loads, and the A loads of the k-steps
                // since: a ring step every 8 / NT k-steps, one A request (two loads) per k-step - (AR_D - 2) * 8 / NT k-steps, rounded DOWN less one for the
                // position inside the k-step (an under-count only waits for more)
                constexpr int kyoung = ((AR_D - 2) * kChunkPairs) / (NT * HV) - 1;
                constexpr int Y = (AR_D - 3) * DPW + (HV == 2 ? 1 : 2) * (kyoung > 0 ? (kyoung < PFA ? kyoung : PFA) : 0);
                ar_ring_step<Y, HV>(rg, stream, stream_bytes, lds, wave, lane);
                if (HV == 2) {      // the other half's chunk: same count (the two DMA loads just issued are exactly the two more that may stay in flight)
                    const uint32_t c0 = rg.cur;
                    ar_ring_step<Y, HV>(rg, stream, stream_bytes, lds, wave, lane);
                    if (hf == 0) rg.cur = c0;
                }
            }
            lds_char* ap = lds + rg.cur + (Tn % kChunkPairs) * kPairBytes + lane * 16;
            if (SNERF_ABLA & 4) {
                fn = f;
                asm volatile("" : "+v"(fn.h0), "+v"(fn.l0), "+v"(fn.h1), "+v"(fn.l1));
            } else {
                fn.h0 = *(lds_cu32x4*)ap;
                fn.l0 = *(lds_cu32x4*)(ap + kFragBytes);
                fn.h1 = *(lds_cu32x4*)(ap + kPairBytes);
                fn.l1 = *(lds_cu32x4*)(ap + kPairBytes + kFragBytes);
            }
        }
        if (FIRST) {      // the previous row tile's n-tiles T0, T1 leave the accumulators (their last MFMAs issued >= 90 MFMAs ago)
            if constexpr (HV == 1) {
                if (ACT && gi + 1 < G) {
                    ar_load_z(g, rs_z, lz, sz0, T0 + 2, zt[(gi + 1) & 1][0]);
                    ar_load_z(g, rs_z, lz, sz0, T1 + 2, zt[(gi + 1) & 1][1]);
                }
                ar_epilogue<ACT>(g, T0, rs_c, lc, so0, rows_left, pv.on, col_l, stat_l, r, h, zt[gi & 1][0]);
                ar_epilogue<ACT>(g, T1, rs_c, lc, so0, rows_left, pv.on, col_l, stat_l, r, h, zt[gi & 1][HV == 1 ? 1 : 0]);
            } else {
                if (ACT) ar_load_z(g, rs_z, lz, sz0, T1, zt[1][0]);
                ar_epilogue<ACT>(g, T0, rs_c, lc, so0, rows_left, pv.on, col_l, stat_l, r, h, zt[0][0]);
                if (ACT && gi + 1 < G) ar_load_z(g, rs_z, lz, sz0, T0 + 2, zt[0][0]);
                ar_epilogue<ACT>(g, T1, rs_c, lc, so0, rows_left, pv.on, col_l, stat_l, r, h, zt[1][0]);
            }
            asm volatile("s_nop 1" ::: "memory");      // accumulator reads -> the MFMAs that overwrite them
        }
        __builtin_amdgcn_sched_barrier(0);
        ar_mfma<FIRST>(16 * T0, clo, f.h0);             // a_lo x w_hi
        ar_mfma<FIRST>(16 * T1, clo, f.h1);
        // ---- a slice of the next k-step's conversion between the MFMAs (six slices: await + table, four element pairs, refill; NT = 8: two per pair)
        constexpr int SPG = G >= 6 ? 1 : 2;            // slices per pair of n-tiles
        if constexpr (HV == 2) {
            // The wave pair of a row group converts each k-step ONCE: this wave its 16 rows (lane = row (lane >> 2), four k (lane & 3)), both halves of the operand
            // go through LDS (xb: two buffers of 1 KiB hi + 1 KiB lo per row group, lane-linear for the readers).  During k-step j: the operands of k-step j + 1
            // are read (written during j - 1, the ring barriers at its end in between), the staged values of k-step j + 2 converted and written to the buffer
            // j & 1 (last read during j - 1), the staging slot refilled PFA k-steps ahead.  `ks_next` = the k-step being CONVERTED, `par` = j & 1.
            const int q4 = lane_f & 3, row16 = lane_f >> 2;
            if (gi == 0) {
                lds_char* xr = xb + (par ^ 1) * 2048 + lane_f * 16;
                nhi = *(lds_cu32x4*)xr;
                nlo = *(lds_cu32x4*)(xr + 1024);
                if (!(SNERF_ABLA & 2)) ar_wait1<(PFA - 1) + DPW * SPK * (PFA - 1)>(sx);
            } else if (gi == 1) {
                float v[4] = {sx[0], sx[1], sx[2], sx[3]};
                if (AOL && 16 * ks_next < g.act_cols) {
                    lds_cfloat* pa = tab_h + 16 * ks_next + 4 * q4;        // (tab_h: the table's base here)
                    lds_cfloat* pb = pa + 16 * KS;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_sinf(__builtin_fmaf(pa[e], v[e], pb[e]));
                }
                uint32_t h0, l0, h1, l1;
                split2_bf16(v[0], v[1], h0, l0);
                split2_bf16(v[2], v[3], h1, l1);
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                __attribute__((address_space(3))) char* xw = (__attribute__((address_space(3))) char*)xb + par * 2048 + (((q4 >> 1) * 32 + 16 * hf + row16) * 16 + (q4 & 1) * 8);
                *(__attribute__((address_space(3))) u32x2*)xw = u32x2{h0, h1};
                *(__attribute__((address_space(3))) u32x2*)(xw + 1024) = u32x2{l0, l1};
            } else if (gi == 2) {
                if (!(SNERF_ABLA & 2)) ar_issue1(as.p + 16 * as.ks, sx);
                if (++as.ks == KS) {
                    as.ks = 0;
                    as.t += gridDim.x;
                    const int64_t tt = as.t < n_tiles ? as.t : n_tiles - 1;
                    int64_t m = (g.reverse ? n_tiles - 1 - tt : tt) * AR_ROWS + wrow * 32 + 16 * hf + row16;
                    m = m < g.M ? m : g.M - 1;
                    as.p = g.A + m * g.lda + 4 * q4;
                }
            }
        } else
#pragma unroll
        for (int q = 0; q < SPG; ++q) {
            const int sl = gi * SPG + q;
            if (sl == 0) {
                // the k-steps requested after this one (PFA - 1 of them) and the DMA loads of the ring steps since (NT / 8 per k-step, at least PFA - 1
                // k-steps' worth) stay in flight
                if (!(SNERF_ABLA & 2)) ar_wait<2 * (PFA - 1) + DPW * SPK * (PFA - 1)>(sx, sy);
                a8[0] = sx[0]; a8[1] = sx[1]; a8[2] = sx[2]; a8[3] = sx[3]; a8[4] = sy[0]; a8[5] = sy[1]; a8[6] = sy[2]; a8[7] = sy[3];
                if (AOL && HV == 1) {
                    lds_cf32x4* pa = (lds_cf32x4*)(tab_h + 16 * ks_next);
                    lds_cf32x4* pb = (lds_cf32x4*)(tab_h + 16 * KS + 16 * ks_next);
                    ta[0] = pa[0]; ta[1] = pa[1]; tb[0] = pb[0]; tb[1] = pb[1];
                }
            } else if (sl >= 1 && sl <= 4) {
                const int e0 = 2 * (sl - 1);
                float v0 = a8[e0], v1 = a8[e0 + 1];
                if (AOL && 16 * ks_next < g.act_cols) {      // (uniform: the table covers the leading act_cols columns, a multiple of 16)
                    if (HV == 1) {
                        v0 = __builtin_amdgcn_sinf(__builtin_fmaf(ta[e0 >> 2][e0 & 3], v0, tb[e0 >> 2][e0 & 3]));
                        v1 = __builtin_amdgcn_sinf(__builtin_fmaf(ta[(e0 + 1) >> 2][(e0 + 1) & 3], v1, tb[(e0 + 1) >> 2][(e0 + 1) & 3]));
                    } else {
                        lds_cfloat* pa = tab_h + 16 * ks_next + e0;
                        lds_cfloat* pb = tab_h + 16 * KS + 16 * ks_next + e0;
                        v0 = __builtin_amdgcn_sinf(__builtin_fmaf(pa[0], v0, pb[0]));
                        v1 = __builtin_amdgcn_sinf(__builtin_fmaf(pa[1], v1, pb[1]));
                    }
                }
                uint32_t hh, ll;
                split2_bf16(v0, v1, hh, ll);
                nhi[sl - 1] = hh;
                nlo[sl - 1] = ll;
            } else if (sl == 5) {
                // refill the staging slot: k-step PFA ahead in the stream (of this tile or the next; past the last tile: the last tile again, never used)
                if (!(SNERF_ABLA & 2)) ar_issue(as.p + 16 * as.ks, sx, sy);
                if (++as.ks == KS) {
                    as.ks = 0;
                    as.t += gridDim.x;
                    const int64_t tt = as.t < n_tiles ? as.t : n_tiles - 1;
                    int64_t m = (g.reverse ? n_tiles - 1 - tt : tt) * AR_ROWS + wrow * 32 + r;
                    m = m < g.M ? m : g.M - 1;
                    as.p = g.A + m * g.lda + 8 * h;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        ar_mfma<false>(16 * T0, chi, f.l0);             // a_hi x w_lo
        ar_mfma<false>(16 * T1, chi, f.l1);
        ar_mfma<false>(16 * T0, chi, f.h0);             // a_hi x w_hi
        ar_mfma<false>(16 * T1, chi, f.h1);
        __builtin_amdgcn_sched_barrier(0);
    }
    fb0 = fb[G & 1];
}

// NT: 32-column n-tiles (8: N = 256, 16: N = 512).  AOL: activation on load from the table g.act_tab ([a | b] x 16 KS).  PFA: k-steps of A in flight
// (divides the number of k-steps: staging slots are compile-time).
// HV: column halves = waves per SIMD (1: a wave owns all N = 32 NT columns in up to 256 AGPRs; 2: N = 64 NT, waves w and w + 4 own a half each in 128 AGPRs,
// 128 architectural registers: the second wave's MFMAs run while the first issues its loads, stores and DMA - what bounds the one-wave form)
template <int NT, int AOL, int PFA, int ACT = 0, int HV = 1>
__global__ __launch_bounds__(64 * AR_WAVES * HV, 1) void gemm_areg_kernel(const GemmX g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    lds_char* lds = (lds_char*)smem;
    const int KS = g.ksteps;
    __attribute__((address_space(3))) float* tab_l = (__attribute__((address_space(3))) float*)(lds + AR_D * kChunkBytes);      // [a | b] x 16 KS
    constexpr int NC = 32 * NT * HV;                                                                                          // N
    constexpr int NTH = 64 * AR_WAVES * HV;
    __attribute__((address_space(3))) float* col_l = tab_l + 2 * 16 * KS;                                                    // bias x N
    __attribute__((address_space(3))) float* stat_l = col_l + NC;                                                            // [sum | sum of squares] x N
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wrow = HV == 2 ? (wave & 3) : wave, hf = HV == 2 ? (wave >> 2) : 0, cb = 32 * NT * hf;
    const int r = lane & 31, h = lane >> 5;
    const uint8_t* stream = (const uint8_t*)g.frag;
    const uint32_t stream_bytes = (uint32_t)NT * HV * KS * kPairBytes;

    if (HV == 2) ar_reserve_agprs128(); else ar_reserve_agprs();
    if (AOL) for (int i = tid; i < 2 * 16 * KS; i += NTH) tab_l[i] = i % (16 * KS) < g.act_cols ? g.act_tab[(i / (16 * KS)) * g.act_cols + i % (16 * KS)] : 0.f;
    for (int i = tid; i < NC; i += NTH) {
        col_l[i] = g.bias ? g.bias[i] : 0.f;
        stat_l[i] = 0.f;
        stat_l[NC + i] = 0.f;
        if (ACT) {      // [a | b | mu | istd] of the SineLayer below, behind the sums (zeros for a layer without BatchNorm: its xhat sums are defined as 0)
            stat_l[2 * NC + i] = g.etab[i];
            stat_l[3 * NC + i] = g.etab[g.N + i];
            stat_l[4 * NC + i] = g.emu ? g.emu[i] : 0.f;
            stat_l[5 * NC + i] = g.eistd ? g.eistd[i] : 0.f;
        }
    }
    ArRing rg;
    rg.rd = 0; rg.cur = 0; rg.goff = 0;
    {
        uint32_t wr = 0;
#pragma unroll
        for (int c = 0; c < AR_D - 2; ++c) {
            ar_dma_chunk<HV>(stream, rg.goff, lds, wr, __builtin_amdgcn_readfirstlane(wave), lane);
            rg.goff += kChunkBytes;
            if (rg.goff >= stream_bytes) rg.goff = 0;
            wr += kChunkBytes;
        }
        rg.wr = wr;
    }
    __syncthreads();                                      // tables in LDS (drains the ring prologue once: harmless)

    const int64_t n_tiles = (g.M + AR_ROWS - 1) / AR_ROWS;
    auto row_of = [&](int64_t t) { return (g.reverse ? n_tiles - 1 - t : t) * AR_ROWS + wrow * 32; };
    // The A stream: k-step j of the workgroup's tile sequence (tile j / KS, k-step j % KS), PFA of them in flight; slot j % PFA (KS % PFA == 0: static).
    f32x4 sx[PFA], sy[HV == 2 ? 1 : PFA];
    ArStream as;
    as.t = blockIdx.x;
    as.ks = 0;
    // this lane's row inside the wave's 32 and its first k: HV = 1 (r, 8 h); HV = 2 (16 hf + (lane >> 2), 4 (lane & 3)): the pair splits the rows
    const int arow = HV == 2 ? 16 * hf + (lane >> 2) : r, ak = HV == 2 ? 4 * (lane & 3) : 8 * h;
    {
        int64_t m = row_of(as.t < n_tiles ? as.t : n_tiles - 1) + arow;
        m = m < g.M ? m : g.M - 1;                        // loads stay in bounds, stores are masked
        as.p = g.A + m * g.lda + ak;
    }
    auto advance = [&]() {
        if (++as.ks == KS) {
            as.ks = 0;
            as.t += gridDim.x;
            int64_t m = row_of(as.t < n_tiles ? as.t : n_tiles - 1) + arow;
            m = m < g.M ? m : g.M - 1;
            as.p = g.A + m * g.lda + ak;
        }
    };
#pragma unroll
    for (int d = 0; d < PFA; ++d) {
        if (!(SNERF_ABLA & 2)) { if constexpr (HV == 2) ar_issue1(as.p + 16 * as.ks, sx[d]); else ar_issue(as.p + 16 * as.ks, sx[d], sy[d]); }
        advance();
    }
    lds_cfloat* tab_h = (lds_cfloat*)tab_l + (HV == 2 ? 0 : 8 * h);
    // HV = 2: the operand exchange of the wave pairs, behind the tables ([row group][k-step parity] x (1 KiB hi + 1 KiB lo))
    lds_char* xb = (lds_char*)(stat_l + (ACT ? 6 : 2) * NC) + wrow * 4096;

    // the first k-step's operands: nothing to hide their conversion behind (once per workgroup)
    u32x4 ohi[2], olo[2];
    if constexpr (HV == 2) {      // k-steps 0 and 1 through the exchange; k-step 0's operands back into registers
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const int q4 = lane & 3, row16 = lane >> 2;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            if (!(SNERF_ABLA & 2)) ar_wait1<0>(sx[d]);
            float v[4] = {sx[d][0], sx[d][1], sx[d][2], sx[d][3]};
            if (AOL && 16 * d < g.act_cols) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_sinf(__builtin_fmaf(tab_h[16 * d + 4 * q4 + e], v[e], tab_h[16 * KS + 16 * d + 4 * q4 + e]));
            }
            uint32_t h0, l0, h1, l1;
            split2_bf16(v[0], v[1], h0, l0);
            split2_bf16(v[2], v[3], h1, l1);
            lds_char* xw = xb + d * 2048 + (((q4 >> 1) * 32 + 16 * hf + row16) * 16 + (q4 & 1) * 8);
            *(__attribute__((address_space(3))) u32x2*)xw = u32x2{h0, h1};
            *(__attribute__((address_space(3))) u32x2*)(xw + 1024) = u32x2{l0, l1};
            if (!(SNERF_ABLA & 2)) ar_issue1(as.p + 16 * as.ks, sx[d]);
            advance();
        }
        __syncthreads();
        ohi[0] = *(lds_cu32x4*)(xb + lane * 16);
        olo[0] = *(lds_cu32x4*)(xb + 1024 + lane * 16);
        __syncthreads();      // (the partner's read of buffer 0 is over before k-step 0 writes k-step 2's operands into it)
    } else {
        if (!(SNERF_ABLA & 2)) ar_wait<0>(sx[0], sy[0]);
        float a8[8] = {sx[0][0], sx[0][1], sx[0][2], sx[0][3], sy[0][0], sy[0][1], sy[0][2], sy[0][3]};
        if (AOL) {
#pragma unroll
            for (int e = 0; e < 8; ++e) a8[e] = __builtin_amdgcn_sinf(__builtin_fmaf(tab_h[e], a8[e], tab_h[16 * KS + e]));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t hh, ll;
            split2_bf16(a8[2 * q], a8[2 * q + 1], hh, ll);
            ohi[0][q] = hh;
            olo[0][q] = ll;
        }
        if (!(SNERF_ABLA & 2)) ar_issue(as.p + 16 * as.ks, sx[0], sy[0]);
        advance();
    }
    // the first pair's fragments
    ArFrag fb0;
    {
        ar_ring_step<(AR_D - 3) * (DMA_PER_WAVE / HV), HV>(rg, stream, stream_bytes, lds, wave, lane);
        if (HV == 2) {
            const uint32_t c0 = rg.cur;
            ar_ring_step<(AR_D - 3) * (DMA_PER_WAVE / HV), HV>(rg, stream, stream_bytes, lds, wave, lane);
            if (hf == 0) rg.cur = c0;
        }
        lds_char* ap = lds + rg.cur + lane * 16;
        fb0.h0 = *(lds_cu32x4*)ap;
        fb0.l0 = *(lds_cu32x4*)(ap + kFragBytes);
        fb0.h1 = *(lds_cu32x4*)(ap + kPairBytes);
        fb0.l1 = *(lds_cu32x4*)(ap + kPairBytes + kFragBytes);
    }
    ArTile pv{0, false};
    // k-steps in blocks of PFA: k-step j uses the operand set j & 1 (PFA is even) and converts staging slot (j + 1) % PFA into the other set
#define AR_STEP(D_, FIRST_)                                                                                                                              \
    ar_kstep<NT, AOL, PFA, FIRST_, ACT, HV>(g, rg, lds, stream, stream_bytes, ohi[(D_) & 1], olo[(D_) & 1], ohi[((D_) + 1) & 1], olo[((D_) + 1) & 1], fb0,      \
                                   sx[((D_) + HV) % PFA], sy[HV == 2 ? 0 : ((D_) + 1) % PFA], as, (ks0 + (D_) + HV) % KS, KS, tab_h, pv, (lds_cfloat*)col_l + cb, stat_l + cb,  \
                                   n_tiles, wave, lane, hf, xb, (D_) & 1);
    for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        for (int ks0 = 0; ks0 < KS; ks0 += PFA) {
            if (ks0 == 0) { AR_STEP(0, true) } else { AR_STEP(0, false) }
            AR_STEP(1, false)
            if (PFA > 2) { AR_STEP(2, false) AR_STEP(3, false) }
            if (PFA > 4) { AR_STEP(4, false) AR_STEP(5, false) AR_STEP(6, false) AR_STEP(7, false) }
        }
        pv = ArTile{row_of(t), true};
    }
#undef AR_STEP
    // the epilogue of the last row tile: nothing to hide it behind
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    if (pv.on) {
        const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc((void*)g.C, 0, (int)(g.M * g.ldc * 4), 0x00020000);
        const int lc = (int)((4 * h) * g.ldc + r + cb) * 4, so0 = (int)(pv.row0 * g.ldc * 4);
        const int rows_left = (int)(g.M - pv.row0) - 4 * h;
        const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc((void*)(ACT ? g.ez : g.A), 0, ACT ? (int)(g.M * g.eld * 4) : 0, 0x00020000);
        const int lz = ACT ? (int)((4 * h) * g.eld + r + cb) * 4 : 0, sz0 = ACT ? (int)(pv.row0 * g.eld * 4) : 0;
#pragma unroll
        for (int T = 0; T < NT; ++T) {
            float zs[16];
            if (ACT) ar_load_z(g, rs_z, lz, sz0, T, zs);
            ar_epilogue<ACT>(g, T, rs_c, lc, so0, rows_left, true, (lds_cfloat*)col_l + cb, stat_l + cb, r, h, zs);
        }
    }
    // the never-consumed A loads of the stream's tail and the ring's must land before the wave ends
#pragma unroll
    for (int q = 0; q < PFA; ++q) { if constexpr (HV == 2) ar_wait1<0>(sx[q]); else ar_wait<0>(sx[q], sy[q]); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (g.stats) {
        __syncthreads();
        for (int i = tid; i < NC; i += NTH) {      // (ACT: the sums are of the finished values, nothing to scale)
            atomicAdd(g.stats + i, (ACT ? 1.0 : (double)g.alpha) * (double)stat_l[i]);
            atomicAdd(g.stats + g.N + i, (ACT ? 1.0 : (double)g.alpha * (double)g.alpha) * (double)stat_l[NC + i]);
        }
    }
}

static int areg_blocks() {
    static int n = 0;
    if (!n) {
        hipDeviceProp_t p;
        int dev = 0;
        n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ? p.multiProcessorCount : 256;
        if (n < 1) n = 256;
    }
    return n;
}

// shapes this kernel takes: N = 256 or 512 exactly (8 / 16 n-tiles in 128 / 256 AGPRs); K in whole 16-k steps (or zero-padded to one: a_padded), their
// number a multiple of 4 between 8 and 64; whole 16-byte aligned rows; no accumulate, no activation-backward epilogue; an activation table covers K
bool gemm_areg_ok(const GemmX& g) {
    const bool aol = g.act_tab != nullptr && g.act_cols > 0;
    const bool k_ok = g.K % 16 == 0 || (g.a_padded && g.lda >= (int64_t)g.ksteps * 16);
    const bool act = g.ez != nullptr;
    return (g.N == 256 || g.N == 512) && g.n_tiles * 32 == g.N && k_ok && g.ksteps >= 8 && g.ksteps <= 64 && g.ksteps % 4 == 0 && !g.accumulate &&
           (!act || (!aol && g.stats && g.etab && g.M * g.eld < (1ll << 29))) &&
           (!aol || (g.act_cols % 16 == 0 && g.act_cols <= g.ksteps * 16)) && ((uintptr_t)g.A % 16 == 0) && g.lda % 4 == 0 && g.frag != nullptr &&
           g.M * g.ldc < (1ll << 29);      // 32-bit byte offsets of the buffer stores
}

hipError_t launch_areg_split_weights(const float* W, int rows, int cols, bool transpose, uint16_t* frag, int n_tiles, int ksteps, hipStream_t st) {
    const int64_t total = (int64_t)n_tiles * ksteps * 512;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(areg_split_weights_kernel, dim3(blocks), dim3(256), 0, st, W, rows, cols, transpose ? 1 : 0, frag, n_tiles, ksteps);
    return hipGetLastError();
}

// g.frag must hold the k-major stream (launch_areg_split_weights)
hipError_t launch_gemm_areg(const GemmX& g, hipStream_t st) {
    if (!gemm_areg_ok(g)) return hipErrorInvalidValue;
    const bool aol = g.act_tab != nullptr && g.act_cols > 0;
    const int KS = g.ksteps;
    const bool act = g.ez != nullptr;
    static const int hv = [] { const char* e = getenv("SNERF_AREG_HV"); return e ? atoi(e) : 2; }();      // SNERF_AREG_HV=1: the one-wave-per-SIMD form at N = 512 too (A/B)
    // (not for the activation-backward form: with its pre-activation staging the 128 architectural registers of a wave overflow by four - hipcc would park
    // them in the accumulators' AGPRs - and the forward's gain, -4 % with activation on load, 0 without, would not pay for a third staging scheme)
    const bool two = g.N == 512 && hv >= 2 && !act;
    const size_t lds = (size_t)AR_D * kChunkBytes + (size_t)(2 * 16 * KS + (act ? 7 : 3) * g.N) * 4 + (two ? 4 * 4096 : 0);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const int64_t n_tiles = (g.M + AR_ROWS - 1) / AR_ROWS;
    int grid = (int)(n_tiles < areg_blocks() ? n_tiles : areg_blocks());
    if (grid < 1) grid = 1;
#define AR_LAUNCH_HV(HV_, NT_, AOL_, PFA_, ACT_)                                                                                  \
    do {                                                                                                                         \
        auto k = gemm_areg_kernel<NT_, AOL_, PFA_, ACT_, HV_>;                                                                   \
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
        if (e != hipSuccess) return e;                                                                                           \
        hipLaunchKernelGGL(k, dim3(grid), dim3(64 * AR_WAVES * HV_), lds, st, g);                                                \
        return hipGetLastError();                                                                                                \
    } while (0)
#define AR_LAUNCH(NT_, AOL_, PFA_, ...) AR_LAUNCH_HV(1, NT_, AOL_, PFA_, (0 __VA_OPT__(+) __VA_ARGS__))
    const bool p8 = KS % 8 == 0;
    if (two) {      // two waves per SIMD, a column half each
        if (aol) AR_LAUNCH_HV(2, 8, 1, 4, 0);
        AR_LAUNCH_HV(2, 8, 0, 4, 0);
    }
    if (act) {      // input gradient with the activation-backward epilogue
        if (g.N == 512) { if (p8) AR_LAUNCH(16, 0, 8, 1); else AR_LAUNCH(16, 0, 4, 1); }
        if (p8) AR_LAUNCH(8, 0, 8, 1); else AR_LAUNCH(8, 0, 4, 1);
    }
    if (g.N == 512) {
        if (aol) { if (p8) AR_LAUNCH(16, 1, 8); else AR_LAUNCH(16, 1, 4); }
        if (p8) AR_LAUNCH(16, 0, 8); else AR_LAUNCH(16, 0, 4);
    }
    if (aol) { if (p8) AR_LAUNCH(8, 1, 8); else AR_LAUNCH(8, 1, 4); }
    if (p8) AR_LAUNCH(8, 0, 8); else AR_LAUNCH(8, 0, 4);
#undef AR_LAUNCH
#undef AR_LAUNCH_HV
}

}  // namespace snerf
