// gfx950 (MI355X / CDNA4): row GEMM of the layer-wise engine for WIDE K (K = 256 / 512) - "one layer of the fused kernel as a kernel".
//
//   C[M, N] = alpha * (act(A)[M, K] x Bt[N, K]^T) + alpha * bias      bf16x3 products (hi*hi + lo*hi + hi*lo, fp32 accumulate) as gemm.hip / gemm16.hip
//
// Why a third row GEMM.  The reference's DEFAULT width is 512 (main_lite.py:80).  There the two kernels of gemm.hip / gemm16.hip keep a 64-column slice of
// the split weights resident in LDS (64 x 512 x 4 B = 128 KiB is all 160 KiB hold), so eight column groups read every A row - and redo its activation
// (fma + v_sin) and bf16 hi / lo split - EIGHT times: 931 us per 512 -> 512 layer against a 304 us copy floor and a 250 us matrix floor, vector-issue bound
// (DESIGN 5.4c).  This kernel turns the roles round, exactly as the fused inference kernels do (kernels.hip): the ACTIVATIONS of a 32-row wave tile are
// resident - activated and split ONCE, parked in the 256 AGPRs addressed by number (K = 512: 32 k-steps x (4 hi + 4 lo) registers; kernels_i8.hip has the
// technique) - and the WEIGHTS stream L2 -> LDS through the 16 KiB ring of the fused kernels (LDS-DMA, one counted vmcnt wait + one barrier per chunk),
// every fragment shared by the four waves of the workgroup.  N is walked in 32-column n-tiles; per n-tile and k-step three MFMAs (A operand = a[n:n+3]).
//   * the next row tile's A stream is software-pipelined INTO the last n-tile: loads AR_PFA k-steps ahead into staging registers, and k-step s of the next
//     tile is activated, split and parked right after k-step s+1 of the last n-tile has issued its MFMAs (the AGPRs of k-step s are free from then on);
//   * the epilogue of n-tile T (bias, BatchNorm column sums, 2 x 128-byte row segments per store instruction) runs inside n-tile T+1's k-steps;
//   * one workgroup (4 waves = one per SIMD, all 512 registers) per CU, persistent over row tiles of 128 rows.
// Same fragment stream as gemm_rows_full_kernel (split_weights_kernel: [n-tile][k-step][hi | lo][512] bf16), same product order, same k order:
// results are bit-identical to it (tools/compare_gemm_paths.py, tests/test_gpu_linear.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "gemm_common.h"
#include "mlp_device.h"
#include "train.h"

#ifndef SNERF_ABLA
#define SNERF_ABLA 0       // timing-only ablations of scratch builds (tools/variants.py): 1 no stores, 2 no next-tile A stream, 4 no LDS weight reads, 8 no ring, 16 no MFMAs
#endif

namespace snerf {

constexpr int AR_WAVES = 4, AR_ROWS = 32 * AR_WAVES;
#ifndef SNERF_AR_D
#define SNERF_AR_D 7
#endif
#ifndef SNERF_AR_PFA
#define SNERF_AR_PFA 8
#endif
constexpr int AR_D = SNERF_AR_D;              // ring slots (16 KiB chunks of 8 weight pairs)
constexpr int AR_PFW = 2;                     // weight pairs requested ahead of their MFMAs (divides every KS: slot index static; 4 costs 16 more registers)
constexpr int AR_PFA = SNERF_AR_PFA;          // k-steps of the NEXT row tile's A in flight during the last n-tile (2 loads per k-step and lane)
static_assert(kChunkPairs == 8 && DMA_PER_WAVE == 4, "ring arithmetic below assumes 8-pair chunks moved by four waves");

#define AR8(n) "a" #n
#define AR8x8(n) AR8(n##0), AR8(n##1), AR8(n##2), AR8(n##3), AR8(n##4), AR8(n##5), AR8(n##6), AR8(n##7), AR8(n##8), AR8(n##9)
__device__ __forceinline__ void ar_reserve_agprs() {      // the compiler must allocate none of a0..a255 (see kernels_i8.hip reserve_agprs)
    asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", AR8x8(1), AR8x8(2), AR8x8(3), AR8x8(4), AR8x8(5), AR8x8(6),
                 AR8x8(7), AR8x8(8), AR8x8(9), AR8x8(10), AR8x8(11), AR8x8(12), AR8x8(13), AR8x8(14), AR8x8(15), AR8x8(16), AR8x8(17), AR8x8(18),
                 AR8x8(19), AR8x8(20), AR8x8(21), AR8x8(22), AR8x8(23), AR8x8(24), "a250", "a251", "a252", "a253", "a254", "a255");
}
__device__ __forceinline__ void ar_park(uint32_t v, int idx) { asm volatile("v_accvgpr_write_b32 a[%1], %0" ::"v"(v), "i"(idx)); }
// acc (+)= A[a[base : base+3]] x B   (A: 32 rows x 16 k from the parked activations, B: 16 k x 32 columns of weights)
template <bool FIRST>
__device__ __forceinline__ void ar_mfma(f32x16& acc, int base, const u32x4& b) {
    if (SNERF_ABLA & 16) { asm volatile("" : "+v"(acc) : "v"(b)); return; }
    if (FIRST) asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%2:%3], %1, 0" : "=v"(acc) : "v"(b), "i"(base), "i"(base + 3));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%2:%3], %1, %0" : "+v"(acc) : "v"(b), "i"(base), "i"(base + 3));
}

// hand-issued A loads: the 8 consecutive k of this lane's row for one k-step (two 16-byte loads); completion is awaited by count
template <int OFF>
__device__ __forceinline__ void ar_issue(const float* p, f32x4& x, f32x4& y) {
    asm volatile("global_load_dwordx4 %0, %2, off offset:%3\n\tglobal_load_dwordx4 %1, %2, off offset:%4" : "=&v"(x), "=&v"(y) : "v"(p), "n"(OFF), "n"(OFF + 16));
}
template <int N>
__device__ __forceinline__ void ar_wait(f32x4& x, f32x4& y) {
    asm volatile("s_waitcnt vmcnt(%2) ; ar_wait %0 %1" : "+v"(x), "+v"(y) : "n"(N));
}

// the same with the count a value that folds to a constant once the k-loop is unrolled (the immediate must be chosen by dispatch); capped at the 6-bit
// counter's 63 (a smaller count only waits for more)
#define AR_CASES8(M_, B_) M_(B_ + 0) M_(B_ + 1) M_(B_ + 2) M_(B_ + 3) M_(B_ + 4) M_(B_ + 5) M_(B_ + 6) M_(B_ + 7)
#define AR_CASES64(M_) AR_CASES8(M_, 0) AR_CASES8(M_, 8) AR_CASES8(M_, 16) AR_CASES8(M_, 24) AR_CASES8(M_, 32) AR_CASES8(M_, 40) AR_CASES8(M_, 48) AR_CASES8(M_, 56)
__device__ __forceinline__ void ar_wait_n(int n, f32x4& x, f32x4& y) {
    switch (n > 63 ? 63 : n) {
#define AR_W(N_) case N_: ar_wait<N_>(x, y); break;
        AR_CASES64(AR_W)
#undef AR_W
        default: ar_wait<0>(x, y); break;
    }
}

// ring steps issued in k-steps j0 .. j1 of an n-tile body: the request for pair j + PFW crosses into a new chunk
constexpr int ar_ring_steps(int j0, int j1) {
    int n = 0;
    for (int j = j0 < 0 ? 0 : j0; j <= j1; ++j) n += ((j + AR_PFW) % kChunkPairs == 0) ? 1 : 0;
    return n;
}

struct ArRing {
    uint32_t rd, wr, cur, goff;
};
// ring_step of mlp_device.h with the count of this wave's vector-memory operations YOUNGER than the chunk handed over given by the caller: loads,
// stores and LDS-DMA count together and retire in issue order (MI355X_MICROARCH.md, vmcnt), so the count is exact - the four younger chunks' DMA loads,
// the epilogue stores of the last five chunk periods, and in the LAST body the staged A loads of the next tile.  (Counting the loads only - the first
// version - is correct too, but then every wait also sits out the acknowledgements of the ~10 stores in flight: 100 us per layer in the ring steps and
// 110 us in the staged loads, found by ablation.)
template <int YOUNGER>
__device__ __forceinline__ void ar_ring_step(ArRing& rg, const uint8_t* stream, uint32_t stream_bytes, lds_char* lds, int wave, int lane) {
    if (SNERF_ABLA & 8) return;
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(YOUNGER) : "memory");
    // (the DMA statement takes its addresses in SGPRs: hipcc keeps `wave` in a vector register in this kernel unless told again that it is uniform)
    dma_chunk(stream, __builtin_amdgcn_readfirstlane(rg.goff), lds, __builtin_amdgcn_readfirstlane(rg.wr), __builtin_amdgcn_readfirstlane(wave), lane);
    rg.goff += kChunkBytes;
    if (rg.goff >= stream_bytes) rg.goff = 0;
    rg.cur = rg.rd;
    rg.rd = ring_next<AR_D>(rg.rd);
    rg.wr = ring_next<AR_D>(rg.wr);
}
__device__ __forceinline__ void ar_ring_step_n(int younger, ArRing& rg, const uint8_t* stream, uint32_t stream_bytes, lds_char* lds, int wave, int lane) {
    switch (younger > 63 ? 63 : younger) {
#define AR_R(N_) case N_: ar_ring_step<N_>(rg, stream, stream_bytes, lds, wave, lane); break;
        AR_CASES64(AR_R)
#undef AR_R
        default: ar_ring_step<0>(rg, stream, stream_bytes, lds, wave, lane); break;
    }
}
// the epilogue store of element i of the previous n-tile is issued in k-step ar_store_kstep(i) of every n-tile body (behind its second MFMA)
constexpr int ar_store_kstep(int i, int KS) { return 1 + (i * (KS - 1)) / 16; }
// stores issued in the k-steps a .. b (inclusive) counted from the start of the current body; negative k-steps lie in the bodies before (same schedule)
constexpr int ar_stores_in(int KS, int a, int b) {
    int n = 0;
    for (int t = a; t <= b; ++t) {
        const int s = ((t % KS) + KS) % KS;
        for (int i = 0; i < 16; ++i) n += ar_store_kstep(i, KS) == s ? 1 : 0;
    }
    return (SNERF_ABLA & 1) ? 0 : n;
}

// activation on load (AOL: sin(2 pi (a z + b)), table [a | b] in LDS) + bf16 hi / lo split of k-step S of this lane's row, parked in a[8 S .. 8 S + 7]
// (tab_h / tab_hb: this lane-half's [a] and [b] rows, OPAQUE per-lane bases - the k-step is then an immediate offset; written as absolute LDS addresses,
//  which lie above the 64 KiB reach of a ds_read offset, hipcc keeps one address register per k-step alive across the whole tile loop and parks its own
//  values in the AGPRs this kernel addresses by number: tests/test_isa_guards.py)
template <int S, int AOL>
__device__ __forceinline__ void ar_convert_park(const f32x4& x, const f32x4& y, lds_cfloat* tab_h, lds_cfloat* tab_hb) {
    float a8[8] = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
    if (AOL) {
        lds_cf32x4* ta = (lds_cf32x4*)(tab_h + 16 * S);
        lds_cf32x4* tb = (lds_cf32x4*)(tab_hb + 16 * S);
        const f32x4 a0 = ta[0], a1 = ta[1], b0 = tb[0], b1 = tb[1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a8[e] = __builtin_amdgcn_sinf(__builtin_fmaf(a0[e], a8[e], b0[e]));
            a8[4 + e] = __builtin_amdgcn_sinf(__builtin_fmaf(a1[e], a8[4 + e], b1[e]));
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t hh, ll;
        split2_bf16(a8[2 * q], a8[2 * q + 1], hh, ll);
        ar_park(hh, 8 * S + q);
        ar_park(ll, 8 * S + 4 + q);
    }
}

// everything the epilogue of one n-tile needs besides its accumulator
struct ArPrev {
    int64_t row0;      // first row of the wave's 32-row tile
    int T;             // n-tile
    bool on;           // false: nothing to write (the very first body of a workgroup)
};

// One 32-column n-tile: KS k-steps of three MFMAs on `acc`; inside them the epilogue of the PREVIOUS n-tile (`pacc`, `pv`) and - LAST - the A stream
// of the next row tile.  Pair index inside the n-tile = k-step (KS % 8 == 0: chunk boundaries are compile-time positions).
template <int KS, int AOL, bool LAST>
__device__ __forceinline__ void ar_ntile(const GemmX& g, ArRing& rg, lds_char* lds, const uint8_t* stream, uint32_t stream_bytes, f32x16& acc, const f32x16& pacc,
                                         const ArPrev& pv, lds_cfloat* col_l, __attribute__((address_space(3))) float* stat_l, u32x4 (&fH)[AR_PFW],
                                         u32x4 (&fL)[AR_PFW], const float* pnext, f32x4 (&sx)[AR_PFA], f32x4 (&sy)[AR_PFA], lds_cfloat* tab_h, int wave,
                                         int lane) {
    const int r = lane & 31, h = lane >> 5;
    lds_cfloat* tab_hb = tab_h + g.act_cols;
    if (LAST) asm volatile("" : "+v"(tab_h), "+v"(tab_hb));
    float s1 = 0.f, s2 = 0.f;
    // The previous n-tile's stores go through a buffer descriptor (as gemm_rows_full_kernel's): one per-lane byte offset ((4 h) ldc + r), the element's
    // row and the n-tile as a SCALAR offset, rows past M dropped by the bounds check - no 64-bit vector address arithmetic and no branch per element
    // (the first version spent ~20 instructions and an exec-mask branch on each of the 16 elements: 19 k cycles per row tile, measured by ablation).
    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc((void*)g.C, 0, (int)(g.M * g.ldc * 4), 0x00020000);
    const int lc = pv.on ? (int)((4 * h) * g.ldc + r) * 4 : (int)0x80000000;
    const int so0 = (int)((pv.row0 * g.ldc + 32 * pv.T) * 4);
    const int rows_left = pv.on ? (int)(g.M - pv.row0) - 4 * h : 0;      // element i is a real row iff (i & 3) + 8 (i >> 2) < rows_left
    const float ab = g.alpha * col_l[32 * pv.T + r];
    if (LAST && !(SNERF_ABLA & 2)) {
        // prime the A stream of the next row tile: k-steps 0 .. PFA-1
#define AR_PRIME(d) if (d < AR_PFA) ar_issue<(d) * 64>(pnext, sx[d < AR_PFA ? d : 0], sy[d < AR_PFA ? d : 0]);
        AR_PRIME(0) AR_PRIME(1) AR_PRIME(2) AR_PRIME(3) AR_PRIME(4) AR_PRIME(5) AR_PRIME(6) AR_PRIME(7)
        AR_PRIME(8) AR_PRIME(9) AR_PRIME(10) AR_PRIME(11) AR_PRIME(12) AR_PRIME(13) AR_PRIME(14) AR_PRIME(15)
#undef AR_PRIME
    }
    // A k-step is a chain of three DEPENDENT MFMAs (same accumulator): the wave - alone on its SIMD, in order - stalls at each of them until the one
    // before has left the pipe, so anything placed AFTER the three overlaps one MFMA at best.  Everything else a k-step does is therefore dealt out
    // into the gaps BETWEEN them (scheduling barriers pin it there): the weight request behind the first, the epilogue slice behind the second, the
    // next tile's A stream behind the third.
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const u32x4 bh = fH[s % AR_PFW], bl = fL[s % AR_PFW];
        if (s == 0) {
            ar_mfma<true>(acc, 8 * s + 4, bh);        // a_lo x w_hi
        } else {
            ar_mfma<false>(acc, 8 * s + 4, bh);
        }
        {   // request pair s + PFW (of this n-tile or the next: the stream is contiguous and cyclic)
            const int qn = s + AR_PFW;
            if (qn % kChunkPairs == 0) {
                // Loads of this wave YOUNGER than the chunk handed over here (issued five ring steps ago): the four younger chunks' DMA loads and - in
                // the LAST body - the staged A loads of the next tile not yet awaited: k-steps s-1 .. min(s-2+PFA, KS-1), i.e. min(PFA, KS+1-s) of them.
                // The count must never EXCEED the younger loads really in flight (else the wait could pass with the chunk still on its way): exact.
                const int staged = (LAST && !(SNERF_ABLA & 2)) ? ((AR_PFA < KS + 1 - s) ? AR_PFA : KS + 1 - s) : 0;
                // ... and the epilogue stores since that chunk's DMA was issued, (AR_D - 2) ring steps = 8 (AR_D - 2) k-steps ago, behind the first MFMA of its
                // k-step as this one: the stores of k-steps s - 8 (AR_D - 2) .. s - 1
                const int stores = ar_stores_in(KS, s - kChunkPairs * (AR_D - 2), s - 1);
                ar_ring_step_n((AR_D - 3) * DMA_PER_WAVE + 2 * staged + stores, rg, stream, stream_bytes, lds, wave, lane);
            }
            lds_char* ap = lds + rg.cur + (qn % kChunkPairs) * kPairBytes + lane * 16;
            if (SNERF_ABLA & 4) {
                asm volatile("" : "+v"(fH[s % AR_PFW]), "+v"(fL[s % AR_PFW]));
            } else {
                fH[s % AR_PFW] = *(lds_cu32x4*)ap;
                fL[s % AR_PFW] = *(lds_cu32x4*)(ap + kFragBytes);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        ar_mfma<false>(acc, 8 * s, bl);               // a_hi x w_lo
        // epilogue slices of the previous n-tile: 16 elements over the k-steps (from k-step 1 on: its last MFMA has long retired by then)
#pragma unroll
        for (int i = 0; i < 16; ++i) {                // element i in k-step 1 + i (KS - 1) / 16 (KS = 16: two elements share k-step 1)
            if (1 + (i * (KS - 1)) / 16 != s) continue;
            const int ro = (i & 3) + 8 * (i >> 2);
            const float a = pacc[i];
            if (!(SNERF_ABLA & 1))
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, __builtin_fmaf(g.alpha, a, ab)), rs_c, lc, so0 + ro * (int)g.ldc * 4, 0);
            const float am = ro < rows_left ? a : 0.f;
            s1 += am;
            s2 = __builtin_fmaf(am, am, s2);
        }
        __builtin_amdgcn_sched_barrier(0);
        ar_mfma<false>(acc, 8 * s, bh);               // a_hi x w_hi
        if (LAST && s >= 1 && !(SNERF_ABLA & 2)) {
            // k-step s-1 of the NEXT row tile: its loads were issued PFA k-steps ago; the MFMAs that read a[8 (s-1) ..] issued one k-step ago.
            // Younger loads of this wave: the PFA-1 staged k-steps behind it, and the one ring step (4 DMA loads) of the last PFA k-steps.
            constexpr int dummy2 = 0;
            (void)dummy2;
            const int k = s - 1;
            // ring steps between the issue of staging(k) and here: k-steps j in (k - PFA + 1, k + 1] with (j + PFW) % 8 == 0 (k >= PFA: exactly one);
            // k < PFA (primed at the top of the body): j in [0, k + 1]
            int rs = 0;
#pragma unroll
            for (int j = (k >= AR_PFA ? k - AR_PFA + 2 : 0); j <= k + 1; ++j) rs += ((j + AR_PFW) % kChunkPairs == 0) ? 1 : 0;
            // younger staged k-steps: k+1 .. min(k + PFA - 1, KS - 1) (staging(k + PFA) is issued below, after this wait)
            const int young = (AR_PFA - 1 < KS - 1 - k) ? AR_PFA - 1 : KS - 1 - k;
            // ... and the stores issued since: k-steps k - PFA + 2 .. k + 1 (k >= PFA) or 0 .. k + 1 (primed at the top of the body)
            const int st = ar_stores_in(KS, k >= AR_PFA ? k - AR_PFA + 2 : 0, k + 1);
            ar_wait_n(2 * young + DMA_PER_WAVE * rs + st, sx[k % AR_PFA], sy[k % AR_PFA]);      // (constants after unrolling: one immediate survives)
            switch (k) {
#define AR_CP(K_) case K_: if (K_ < KS) { ar_convert_park<(K_ < KS ? K_ : 0), AOL>(sx[K_ % AR_PFA], sy[K_ % AR_PFA], tab_h, tab_hb); \
                                            if (K_ + AR_PFA < KS) ar_issue<((K_ + AR_PFA) < KS ? (K_ + AR_PFA) : 0) * 64>(pnext, sx[K_ % AR_PFA], sy[K_ % AR_PFA]); } break;
                AR_CP(0) AR_CP(1) AR_CP(2) AR_CP(3) AR_CP(4) AR_CP(5) AR_CP(6) AR_CP(7) AR_CP(8) AR_CP(9) AR_CP(10) AR_CP(11) AR_CP(12) AR_CP(13) AR_CP(14) AR_CP(15)
                AR_CP(16) AR_CP(17) AR_CP(18) AR_CP(19) AR_CP(20) AR_CP(21) AR_CP(22) AR_CP(23) AR_CP(24) AR_CP(25) AR_CP(26) AR_CP(27) AR_CP(28) AR_CP(29) AR_CP(30)
#undef AR_CP
                default: break;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (LAST && !(SNERF_ABLA & 2)) {
        // the last k-step of the next tile: its AGPRs were read by the three MFMAs just issued - let them start before the registers change
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
        // younger loads: the ring steps since staging(KS-1) was issued (at the end of k-step KS-PFA)
        ar_wait<DMA_PER_WAVE * ar_ring_steps(KS - AR_PFA + 1, KS - 1) + ar_stores_in(KS, KS - AR_PFA + 1, KS - 1)>(sx[(KS - 1) % AR_PFA], sy[(KS - 1) % AR_PFA]);
        ar_convert_park<KS - 1, AOL>(sx[(KS - 1) % AR_PFA], sy[(KS - 1) % AR_PFA], tab_h, tab_hb);
        asm volatile("s_nop 3" ::: "memory");           // parked values -> the next tile's first MFMA
    }
    // BatchNorm column sums of the previous n-tile: the two lane-halves hold the same column (rows differ), the four waves too
    if (g.stats) {
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (h == 0 && pv.on) {
            __builtin_amdgcn_ds_faddf(stat_l + 32 * pv.T + r, s1, __ATOMIC_RELAXED, __MEMORY_SCOPE_WRKGRP, false);      // ds_add_f32: a flat atomic would count in vmcnt
            __builtin_amdgcn_ds_faddf(stat_l + g.N + 32 * pv.T + r, s2, __ATOMIC_RELAXED, __MEMORY_SCOPE_WRKGRP, false);
        }
    }
}

// KS: k-steps of 16 (K = 16 KS; 16 or 32).  AOL: activation on load from the table g.act_tab ([a | b] x K).
template <int KS, int AOL>
__global__ __launch_bounds__(64 * AR_WAVES, 1) void gemm_areg_kernel(const GemmX g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    lds_char* lds = (lds_char*)smem;
    __attribute__((address_space(3))) float* tab_l = (__attribute__((address_space(3))) float*)(lds + AR_D * kChunkBytes);      // [a | b] x K
    __attribute__((address_space(3))) float* col_l = tab_l + 2 * 16 * KS;                                                    // bias x N
    __attribute__((address_space(3))) float* stat_l = col_l + g.N;                                                           // [sum | sum of squares] x N
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int NT = (int)(g.N / 32);
    const uint8_t* stream = (const uint8_t*)g.frag;
    const uint32_t stream_bytes = (uint32_t)NT * KS * kPairBytes;

    ar_reserve_agprs();
    if (AOL) for (int i = tid; i < 2 * 16 * KS; i += 64 * AR_WAVES) tab_l[i] = g.act_tab[i];
    for (int i = tid; i < (int)g.N; i += 64 * AR_WAVES) {
        col_l[i] = g.bias ? g.bias[i] : 0.f;
        stat_l[i] = 0.f;
        stat_l[g.N + i] = 0.f;
    }
    ArRing rg;
    rg.rd = 0; rg.cur = 0; rg.goff = 0;
    {
        uint32_t wr = 0;
#pragma unroll
        for (int c = 0; c < AR_D - 2; ++c) {
            dma_chunk(stream, rg.goff, lds, wr, __builtin_amdgcn_readfirstlane(wave), lane);
            rg.goff += kChunkBytes;
            if (rg.goff >= stream_bytes) rg.goff = 0;
            wr += kChunkBytes;
        }
        rg.wr = wr;
    }
    lds_cfloat* tab_h = (lds_cfloat*)tab_l + 8 * h;
    lds_cfloat* tab_hb0 = tab_h + g.act_cols;
    asm volatile("" : "+v"(tab_h), "+v"(tab_hb0));

    const int64_t n_tiles = (g.M + AR_ROWS - 1) / AR_ROWS;
    auto row_of = [&](int64_t t) { return (g.reverse ? n_tiles - 1 - t : t) * AR_ROWS + wave * 32; };
    auto a_ptr = [&](int64_t t) {
        int64_t m = row_of(t) + r;
        m = m < g.M ? m : g.M - 1;                       // loads stay in bounds, stores are masked
        return g.A + m * g.lda + 8 * h;
    };
    f32x4 sx[AR_PFA], sy[AR_PFA];
    int64_t t = blockIdx.x;
    if (t < n_tiles) {
        // the first row tile's A: nothing to hide it behind (once per workgroup)
        const float* p0 = a_ptr(t);
#define AR_FIRST(S_) if (S_ < KS) { ar_issue<(S_ < KS ? S_ : 0) * 64>(p0, sx[S_ % AR_PFA], sy[S_ % AR_PFA]); }
#define AR_FIRSTC(S_) if (S_ < KS) { ar_wait<0>(sx[S_ % AR_PFA], sy[S_ % AR_PFA]); ar_convert_park<(S_ < KS ? S_ : 0), AOL>(sx[S_ % AR_PFA], sy[S_ % AR_PFA], tab_h, tab_hb0); }
        __syncthreads();                                  // the activation table is in LDS (drains the ring prologue once: harmless)
        AR_FIRST(0) AR_FIRST(1) AR_FIRST(2) AR_FIRST(3) AR_FIRST(4) AR_FIRST(5) AR_FIRST(6) AR_FIRST(7)
        AR_FIRSTC(0) AR_FIRSTC(1) AR_FIRSTC(2) AR_FIRSTC(3) AR_FIRSTC(4) AR_FIRSTC(5) AR_FIRSTC(6) AR_FIRSTC(7)
        AR_FIRST(8) AR_FIRST(9) AR_FIRST(10) AR_FIRST(11) AR_FIRST(12) AR_FIRST(13) AR_FIRST(14) AR_FIRST(15)
        AR_FIRSTC(8) AR_FIRSTC(9) AR_FIRSTC(10) AR_FIRSTC(11) AR_FIRSTC(12) AR_FIRSTC(13) AR_FIRSTC(14) AR_FIRSTC(15)
        AR_FIRST(16) AR_FIRST(17) AR_FIRST(18) AR_FIRST(19) AR_FIRST(20) AR_FIRST(21) AR_FIRST(22) AR_FIRST(23)
        AR_FIRSTC(16) AR_FIRSTC(17) AR_FIRSTC(18) AR_FIRSTC(19) AR_FIRSTC(20) AR_FIRSTC(21) AR_FIRSTC(22) AR_FIRSTC(23)
        AR_FIRST(24) AR_FIRST(25) AR_FIRST(26) AR_FIRST(27) AR_FIRST(28) AR_FIRST(29) AR_FIRST(30) AR_FIRST(31)
        AR_FIRSTC(24) AR_FIRSTC(25) AR_FIRSTC(26) AR_FIRSTC(27) AR_FIRSTC(28) AR_FIRSTC(29) AR_FIRSTC(30) AR_FIRSTC(31)
#undef AR_FIRST
#undef AR_FIRSTC
        asm volatile("s_nop 3" ::: "memory");
    } else {
        __syncthreads();
    }
    // (the ring prologue's DMA loads have been waited for by the ar_wait<0> above; the ring protocol below only ever asks for "all but N youngest")

    // the first PFW weight pairs
    u32x4 fH[AR_PFW], fL[AR_PFW];
#pragma unroll
    for (int q = 0; q < AR_PFW; ++q) {
        if (q % kChunkPairs == 0) ar_ring_step<(AR_D - 3) * DMA_PER_WAVE>(rg, stream, stream_bytes, lds, wave, lane);
        lds_char* ap = lds + rg.cur + (q % kChunkPairs) * kPairBytes + lane * 16;
        fH[q] = *(lds_cu32x4*)ap;
        fL[q] = *(lds_cu32x4*)(ap + kFragBytes);
    }
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    ArPrev pv{0, 0, false};
    for (; t < n_tiles; t += gridDim.x) {
        const int64_t tn = t + gridDim.x;
        const float* pnext = a_ptr(tn < n_tiles ? tn : t);            // no next tile: reload this one (parked, never used)
        const int64_t row0 = row_of(t);
        for (int T = 0; T < NT; T += 2) {
            ar_ntile<KS, AOL, false>(g, rg, lds, stream, stream_bytes, acc0, acc1, pv, (lds_cfloat*)col_l, stat_l, fH, fL, pnext, sx, sy, tab_h, wave, lane);
            pv = ArPrev{row0, T, true};
            if (T + 2 < NT) ar_ntile<KS, AOL, false>(g, rg, lds, stream, stream_bytes, acc1, acc0, pv, (lds_cfloat*)col_l, stat_l, fH, fL, pnext, sx, sy, tab_h, wave, lane);
            else ar_ntile<KS, AOL, true>(g, rg, lds, stream, stream_bytes, acc1, acc0, pv, (lds_cfloat*)col_l, stat_l, fH, fL, pnext, sx, sy, tab_h, wave, lane);
            pv = ArPrev{row0, T + 1, true};
        }
    }
    // the epilogue of the very last n-tile: nothing to hide it behind
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    if (pv.on) {
        float s1 = 0.f, s2 = 0.f;
        const float ab = g.alpha * col_l[32 * pv.T + r];
        float* cbase = g.C + (pv.row0 + 4 * h) * g.ldc + 32 * pv.T + r;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ro = (i & 3) + 8 * (i >> 2);
            const float a = acc1[i];
            if (pv.row0 + 4 * h + ro < g.M) {
                cbase[(int64_t)ro * g.ldc] = __builtin_fmaf(g.alpha, a, ab);
                s1 += a;
                s2 = __builtin_fmaf(a, a, s2);
            }
        }
        if (g.stats) {
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (h == 0) {
                __builtin_amdgcn_ds_faddf(stat_l + 32 * pv.T + r, s1, __ATOMIC_RELAXED, __MEMORY_SCOPE_WRKGRP, false);
                __builtin_amdgcn_ds_faddf(stat_l + g.N + 32 * pv.T + r, s2, __ATOMIC_RELAXED, __MEMORY_SCOPE_WRKGRP, false);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA may outlive the workgroup
    if (g.stats) {
        __syncthreads();
        for (int i = tid; i < (int)g.N; i += 64 * AR_WAVES) {
            atomicAdd(g.stats + i, (double)g.alpha * (double)stat_l[i]);
            atomicAdd(g.stats + g.N + i, (double)g.alpha * (double)g.alpha * (double)stat_l[g.N + i]);
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

// shapes this kernel takes: K = 256 or 512 exactly (16 / 32 k-steps in 128 / 256 AGPRs), N a multiple of 64, whole 16-byte aligned rows,
// no accumulate, no activation-backward epilogue; an activation table must cover all of K
bool gemm_areg_ok(const GemmX& g) {
    const bool aol = g.act_tab != nullptr && g.act_cols > 0;
    return (g.K == 256 || g.K == 512) && g.ksteps * 16 == g.K && g.N >= 64 && g.N % 64 == 0 && g.N <= 1024 && g.n_tiles * 32 == g.N && !g.accumulate && !g.ez &&
           (!aol || g.act_cols == g.K) && ((uintptr_t)g.A % 16 == 0) && g.lda % 4 == 0 && g.frag != nullptr && (!aol || (uintptr_t)g.act_tab % 16 == 0) &&
           g.M * g.ldc < (1ll << 29);      // 32-bit byte offsets of the buffer stores
}

hipError_t launch_gemm_areg(const GemmX& g, hipStream_t st) {
    if (!gemm_areg_ok(g)) return hipErrorInvalidValue;
    const bool aol = g.act_tab != nullptr && g.act_cols > 0;
    const int KS = g.ksteps;
    const size_t lds = (size_t)AR_D * kChunkBytes + (size_t)(2 * 16 * KS + 3 * g.N) * 4;
    const int64_t n_tiles = (g.M + AR_ROWS - 1) / AR_ROWS;
    int grid = (int)(n_tiles < areg_blocks() ? n_tiles : areg_blocks());
    if (grid < 1) grid = 1;
#define AR_LAUNCH(KS_, AOL_)                                                                                                     \
    do {                                                                                                                         \
        auto k = gemm_areg_kernel<KS_, AOL_>;                                                                                    \
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
        if (e != hipSuccess) return e;                                                                                           \
        hipLaunchKernelGGL(k, dim3(grid), dim3(64 * AR_WAVES), lds, st, g);                                                      \
        return hipGetLastError();                                                                                                \
    } while (0)
    if (KS == 32) { if (aol) AR_LAUNCH(32, 1); else AR_LAUNCH(32, 0); }
    if (aol) AR_LAUNCH(16, 1);
    AR_LAUNCH(16, 0);
#undef AR_LAUNCH
}

}  // namespace snerf
