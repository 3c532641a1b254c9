// Device helpers shared by the fused-MLP kernels (kernels.hip: bf16 split products; kernels_i8.hip: int8 digits):
// vector types, the LDS-DMA weight ring, the exact positional-encoding argument reduction, output non-linearities.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "program.h"
#include "kernels.h"
#if defined(SNERF_ABLATE) && !defined(ABL)
#define ABL 0
#endif

namespace snerf {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;
typedef const __attribute__((address_space(3))) float lds_cfloat;
typedef const __attribute__((address_space(3))) u32x4 lds_cu32x4;
typedef const __attribute__((address_space(3))) f32x4 lds_cf32x4;
typedef const __attribute__((address_space(1))) void glb_void;

#ifndef SNERF_RING_D
#define SNERF_RING_D 7
#endif
constexpr int RING_D = SNERF_RING_D;            // ring slots (one chunk each)
constexpr int DMA_PER_WAVE = kChunkBytes / kFragBytes / 4;   // 1 KiB pieces each wave moves per chunk
constexpr int RING_BYTES = RING_D * kChunkBytes;
constexpr int TILE_PTS = 128;                   // points per workgroup tile (4 waves x 32)

struct Frag {          // B operand of one k-step: 8 bf16 hi + 8 bf16 lo of this lane's point
    u32x4 hi, lo;
};

struct Ring {
    uint32_t rd;       // LDS offset of the slot the NEXT ring_step hands to the consumers
    uint32_t wr;       // LDS offset of the slot the next DMA fills
    uint32_t cur;      // LDS offset of the chunk being consumed
    uint32_t goff;     // byte offset in the (cyclic) global stream of the next chunk to fetch
};

__device__ __forceinline__ float sin2pi(float r) { return __builtin_amdgcn_sinf(r); }   // v_sin_f32: revolutions,
__device__ __forceinline__ float cos2pi(float r) { return __builtin_amdgcn_cosf(r); }   // 1.25e-7 abs err (probe_hw)

// Fetch one 16 KiB chunk: 16 pieces of 1 KiB, wave w moves pieces w, w+4, w+8, w+12 (LDS-DMA: each lane's 16 bytes land
// at M0 + lane*16).  Issued through inline asm on purpose: the __builtin_amdgcn_global_load_lds form is FLAT-encoded
// and makes hipcc (ROCm 7.2) treat every later LDS read as dependent on a "pending flat" access, i.e. it emits
// s_waitcnt lgkmcnt(0) in front of every MFMA instead of counted waits (measured: 685 of 685 waits).  hipcc neither
// counts these loads nor waits for them; ring_step's hand-counted vmcnt does (cdna_hip_programming.md 5.7).
// M0 is written and restored inside the one statement; saddr form: address = sgpr base + lane*16.
__device__ __forceinline__ void dma_chunk(const uint8_t* stream, uint32_t goff, lds_char* lds, uint32_t wr, int wave, int lane) {
    const uint8_t* b0 = stream + goff + wave * kFragBytes;                    // wave-uniform
    const uint32_t dst = (uint32_t)(uintptr_t)(lds + wr + wave * kFragBytes); // wave-uniform LDS byte address
    const uint32_t voff = lane * 16;
    uint32_t keep;
#pragma unroll
    for (int part = 0; part < DMA_PER_WAVE / 4; ++part) {
        const uint8_t* bp = b0 + part * 16 * kFragBytes;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %2\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %3\n\t"
            "s_add_u32 m0, m0, 0x1000\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %4\n\t"
            "s_add_u32 m0, m0, 0x1000\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %5\n\t"
            "s_add_u32 m0, m0, 0x1000\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %6\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff), "s"(dst + part * 16 * kFragBytes), "s"(bp), "s"(bp + 4 * kFragBytes), "s"(bp + 8 * kFragBytes), "s"(bp + 12 * kFragBytes)
            : "memory", "scc");
    }
}

template <int D = RING_D>
__device__ __forceinline__ uint32_t ring_next(uint32_t off) {
    off += kChunkBytes;
    return off == D * kChunkBytes ? 0u : off;
}

// Hand the next chunk to the consumers and refill the slot released TWO chunks ago.
//  - vmcnt((D-3)*DMA_PER_WAVE): all but the (D-3) youngest chunks this wave fetched have landed => chunk `rd` is complete
//    (counted in DMA instructions of THIS wave; extra older loads/stores only make the wait stricter);
//  - s_barrier: every wave's pieces of chunk `rd` have landed, and every wave has issued the MFMAs that consumed the
//    chunk two steps back (its LDS reads are therefore complete) - so the refill needs no lgkmcnt drain, and the
//    software-pipelined fragment reads of the previous chunk stay in flight across the barrier.
template <int D = RING_D>
__device__ __forceinline__ void ring_step(Ring& rg, const uint8_t* stream, uint32_t stream_bytes, lds_char* lds, int wave, int lane) {
#if defined(SNERF_ABLATE) && (ABL & 4)     // timing-only: no ring at all
    return;
#endif
#if defined(SNERF_ABLATE) && (ABL & 64)    // timing-only: the DMA stream without its rendezvous
    asm volatile("" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((D - 3) * DMA_PER_WAVE) : "memory");
#endif
#if !(defined(SNERF_ABLATE) && (ABL & 32)) // timing-only (32): the rendezvous without the DMA stream
    dma_chunk(stream, rg.goff, lds, rg.wr, wave, lane);
#endif
    rg.goff += kChunkBytes;
    if (rg.goff >= stream_bytes) rg.goff = 0;
    rg.cur = rg.rd;
    rg.rd = ring_next<D>(rg.rd);
    rg.wr = ring_next<D>(rg.wr);
}

// sin/cos of k_j * x exactly as the reference evaluates them (misc.py:109,127-131): the fp32 argument is
// 2^j * fl32(fl32(pi/2) * x); it is reduced in fp64 (exact power-of-two scaling, exact fract) before v_sin/v_cos.
struct PeArg {
    double u;   // fl32(fl32(pi/2)*x) / (2*pi), revolutions at j = 0
};
__device__ __forceinline__ PeArg pe_arg(float x) {
    const float a0 = __fmul_rn(x, 1.57079637050628662109375f);
    PeArg r;
    r.u = (double)a0 * 0.15915494309189533576888;
    return r;
}
// the same with the power of two given as its exponent (v_ldexp_f64: exact, and no per-lane table of fp64 scales to keep live)
__device__ __forceinline__ void pe_sincos_exp(const PeArg& a, int e, float& c, float& s) {
    const double r = __builtin_ldexp(a.u, e);
    const float f = (float)(r - __builtin_floor(r));
    c = cos2pi(f);
    s = sin2pi(f);
}
__device__ __forceinline__ void pe_sincos(const PeArg& a, double scale, float& c, float& s) {
    const double r = a.u * scale;                 // exact: scale = 2^j
    const float f = (float)(r - __builtin_floor(r));
    c = cos2pi(f);
    s = sin2pi(f);
}

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }   // torch Softplus(beta 1, thr 20)
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// VARIANT 3 ("ray visibility", Eval_Tools_2.py:255-271 / mg_Img_Eval.py:57-70): the density-only program over the samples of a ray, the
// optical depth sum_{j < S-1} rho_j delta_j kept in a register across the ray's ceil(S / 32) passes, one float out per ray:
// vis = exp(-sum).  A wave owns ray `group * waves + wave`; lane l & 31 of pass p evaluates sample 32 p + (l & 31).
struct RaySum {
    float sum;                      // running optical depth of this lane's samples (lives across the ray's passes)
};
// Which 32 samples the p-th pass of a ray evaluates.  The secondary ray runs from `top` (where it leaves the cube towards the sun, t = 0) to `bot` (the point
// whose visibility is asked for, t = 1); the passes walk it FROM THE POINT OUTWARDS: block ceil(S / 32) - 1 first.  The sum does not care about the order, the
// early-out below does: matter that shadows a point of a real scene is the ground / the building the point sits in, i.e. next to the point, and a ray that
// meets it in its first pass is finished after one evaluation of the network instead of three (ray_flags bit 3 = the old order, sun side first: A/B).
__device__ __forceinline__ int raysum_block(const MlpArgs& A, int p) { return (A.ray_flags & 8) ? p : (A.n_samples + 31) / 32 - 1 - p; }
// sample position of this lane in pass p (misc.py:234-247: top (1 - t) + bot t, two roundings + one add).  The ray's end points are re-read
// every pass (cached loads before the MFMA chain) and once more after it for the segment length: nothing but `sum` lives across the chain.
__device__ __forceinline__ void raysum_point(RaySum& q, const MlpArgs& A, int64_t group, int waves, int wave, int p, int lane, float& x0, float& x1, float& x2) {
    const int64_t ray = group * waves + wave;
    const int64_t r = ray < A.n ? ray : A.n - 1;
    if (p == 0) q.sum = 0.f;
    const int s = raysum_block(A, p) * 32 + (lane & 31);
    const float t = A.tvals[s < A.n_samples ? s : A.n_samples - 1], omt = __fsub_rn(1.f, t);
    x0 = __fadd_rn(__fmul_rn(A.top[r * 3], omt), __fmul_rn(A.bot[r * 3], t));
    x1 = __fadd_rn(__fmul_rn(A.top[r * 3 + 1], omt), __fmul_rn(A.bot[r * 3 + 1], t));
    x2 = __fadd_rn(__fmul_rn(A.top[r * 3 + 2], omt), __fmul_rn(A.bot[r * 3 + 2], t));
}
// lane-half 0 holds the density row; the last sample never counts (PV at the last sample is the EXCLUSIVE prefix, Eval_Tools_2.py:13-16);
// delta = ||top - bot|| / S (misc.py:243) with the composite kernel's operations
__device__ __forceinline__ void raysum_add(RaySum& q, const MlpArgs& A, int64_t group, int waves, int wave, int p, int lane, float rho_raw, float x0, float x1, float x2) {
    // its own basic block (as the `if (h == 0 && valid)` around the other variants' stores): straight-line code here is scheduled up into the last
    // layer's epilogue, where the softplus / division expansions cost 26 registers of scratch (measured)
    if (lane >= 32) return;
    const int64_t ray = group * waves + wave;
    const int64_t r = ray < A.n ? ray : A.n - 1;
    const float dx = A.top[r * 3] - A.bot[r * 3], dy = A.top[r * 3 + 1] - A.bot[r * 3 + 1], dz = A.top[r * 3 + 2] - A.bot[r * 3 + 2];
    const float delta = __fdiv_rn(__fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz))), (float)A.n_samples);
    const int s = raysum_block(A, p) * 32 + (lane & 31);
    {   // the sample position again (the registers that held it during the chain are long gone: fewer values live across it)
        const float t = A.tvals[s < A.n_samples ? s : A.n_samples - 1], omt = __fsub_rn(1.f, t);
        x0 = __fadd_rn(__fmul_rn(A.top[r * 3], omt), __fmul_rn(A.bot[r * 3], t));
        x1 = __fadd_rn(__fmul_rn(A.top[r * 3 + 1], omt), __fmul_rn(A.bot[r * 3 + 1], t));
        x2 = __fadd_rn(__fmul_rn(A.top[r * 3 + 2], omt), __fmul_rn(A.bot[r * 3 + 2], t));
    }
    const bool oob = x0 > 1.f || x0 < -1.f || x1 > 1.f || x1 < -1.f || x2 > 1.f || x2 < -1.f;
    const bool on = s < A.n_samples - 1 && !((A.ray_flags & 2) && oob);
    q.sum += on ? __fmul_rn(softplus_f(rho_raw), delta) : 0.f;
}
// Early-out (round 6): behind a surface the optical depth only grows, and exp(-18) = 1.5e-8 is below the last bit of a visibility - once EVERY ray of the
// workgroup's group has passed 18, the ray's remaining passes (whole evaluations of the density network) change no result by more than that.  The vote is
// workgroup-wide because the waves share the weight ring: all of them skip the same passes, at a pass boundary, where the cyclic stream stands at its
// start either way.  Consecutive secondary rays start at consecutive samples of one primary ray, so a group saturates together.  `vote`: one LDS float per wave
// (kVoteBytes behind each kernel's LDS image); the store is drained and the barrier passed by all waves before the loads; the next vote is a whole pass later.
constexpr float kSaturatedDepth = 18.f;
// `ray`: the ray this wave walks; `slot` / `n_slots`: this wave's vote word and how many the workgroup casts
__device__ __forceinline__ bool raysum_saturated(const RaySum& q, const MlpArgs& A, int64_t ray, int slot, int n_slots, int lane,
                                                 __attribute__((address_space(3))) float* vote) {
    if (A.ray_flags & 4) return false;                        // A/B switch (SNERF_RAYVIS_NO_EARLY_OUT=1): every pass of every ray
    float v = q.sum;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (ray >= A.n) v = 1e30f;                                // a ray past the end never holds the group back
    if (lane == 0) vote[slot] = v;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    bool all = true;
    for (int w = 0; w < n_slots; ++w) all = all && vote[w] > kSaturatedDepth;
    return all;
}
__device__ __forceinline__ void raysum_end(RaySum& q, const MlpArgs& A, int64_t group, int waves, int wave, int lane) {
    float v = q.sum;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int64_t ray = group * waves + wave;
    if (lane == 0 && ray < A.n) A.out.vis[ray] = expf(-v);
}

// output non-linearities of the field program (T_NeRF_net_v2.py:91-98) for one point; called by the lanes that hold the head rows
template <int VARIANT>
__device__ __forceinline__ void store_field_outputs(const snerf_field_out_dev& O, int64_t n, int C, float x0, float x1, float x2,
                                                    float col_r, float col_g, float col_b, float rho_raw, float sv_raw,
                                                    const float* adj, const float* pcls) {
    if (O.rho) O.rho[n] = softplus_f(rho_raw);
    if (O.points) { O.points[n * 3] = x0; O.points[n * 3 + 1] = x1; O.points[n * 3 + 2] = x2; }
    if constexpr (VARIANT <= 1) {
        if (O.solar_vis) O.solar_vis[n] = sigmoid_f(sv_raw);
    }
    if constexpr (VARIANT == 0) {
        if (O.col_raw) { O.col_raw[n * 3] = col_r; O.col_raw[n * 3 + 1] = col_g; O.col_raw[n * 3 + 2] = col_b; }
        float ac0 = 0.f, ac1 = 0.f, ac2 = 0.f;
#pragma unroll
        for (int c = 0; c < kMaxClasses; ++c) {
            if (c < C) {
                if (O.adjust) {
                    O.adjust[(n * C + c) * 3] = adj[3 * c];
                    O.adjust[(n * C + c) * 3 + 1] = adj[3 * c + 1];
                    O.adjust[(n * C + c) * 3 + 2] = adj[3 * c + 2];
                }
                const float pc = pcls[c];
                ac0 = __fadd_rn(ac0, __fmul_rn(adj[3 * c], pc));
                ac1 = __fadd_rn(ac1, __fmul_rn(adj[3 * c + 1], pc));
                ac2 = __fadd_rn(ac2, __fmul_rn(adj[3 * c + 2], pc));
            }
        }
        if (O.adjust_col) { O.adjust_col[n * 3] = ac0; O.adjust_col[n * 3 + 1] = ac1; O.adjust_col[n * 3 + 2] = ac2; }
        if (O.col) {
            O.col[n * 3] = sigmoid_f(col_r + ac0);
            O.col[n * 3 + 1] = sigmoid_f(col_g + ac1);
            O.col[n * 3 + 2] = sigmoid_f(col_b + ac2);
        }
    }
}

}  // namespace snerf
