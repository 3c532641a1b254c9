// bf16 hi/lo split arithmetic shared by the fused bf16x3 kernels (kernels.hip: one wave per 32 points; kernels_ks.hip: K split over a wave pair):
// the 3-term product, the sliced sin/split epilogue, the bias-initialised accumulator, the positional encodings as B fragments.
#pragma once
#include "mlp_device.h"

namespace snerf {

// two fp32 -> packed bf16 hi and packed bf16 lo (x = hi + lo to ~2^-17 relative)
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    bf16x2 hv;
    hv[0] = (__bf16)a;
    hv[1] = (__bf16)b;
    hi = __builtin_bit_cast(uint32_t, hv);
    const float ha = __builtin_bit_cast(float, hi << 16);
    const float hb = __builtin_bit_cast(float, hi & 0xffff0000u);
    bf16x2 lv;
    lv[0] = (__bf16)(a - ha);
    lv[1] = (__bf16)(b - hb);
    lo = __builtin_bit_cast(uint32_t, lv);
}

// 8 consecutive values -> one Frag
__device__ __forceinline__ void pack8(const float* v, Frag& f) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t h, l;
        split2(v[2 * q], v[2 * q + 1], h, l);
        f.hi[q] = h;
        f.lo[q] = l;
    }
}

__device__ __forceinline__ f32x16 mfma3(const u32x4& a_hi, const u32x4& a_lo, const Frag& b, f32x16 acc) {
    const bf16x8 ah = __builtin_bit_cast(bf16x8, a_hi), al = __builtin_bit_cast(bf16x8, a_lo);
    const bf16x8 bh = __builtin_bit_cast(bf16x8, b.hi), bl = __builtin_bit_cast(bf16x8, b.lo);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    return acc;
}

// "bf16-fast": one MFMA per product on the rounded operands (2-3e-3 on RGB, outside the parity bar)
__device__ __forceinline__ f32x16 mfma1(const u32x4& a_hi, const Frag& b, f32x16 acc) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_hi), __builtin_bit_cast(bf16x8, b.hi), acc, 0, 0, 0);
}

// One fused layer: out^T[n x 32 pts] = act(W[n x k] * in^T[k x 32 pts] + b), activations in registers.
//   in0/in1: B fragments of the (concatenated) input blocks; out: 2 fragments per 32-row output block;
//   raw: fp32 accumulator of block 0 for OUT_RAW layers.
// Software pipelining, all at source level (every index is static after unrolling):
//   * the A fragments (weights) are read PF pairs ahead of the MFMAs that consume them, and the ring step that
//     publishes a chunk runs PF pairs before the chunk's first MFMA: LDS latency and the barrier hide behind MFMAs;
//   * accumulators ping-pong between blocks and the epilogue of block b-1 (bias, sin, bf16 hi/lo split) is emitted
//     in four slices inside block b's MFMA sequence, so VALU/transcendental work fills the MFMA shadow instead
//     of serialising at every block boundary.
#ifndef SNERF_PF
#define SNERF_PF 3
#endif
constexpr int PF = SNERF_PF;

// Epilogue of element pair e (0..7) of a 32x32 accumulator block, cut into three phases that run in three
// consecutive k-steps, so that the VALU work sharing an MFMA shadow is always three *independent* short chains
// (a single in-order wave cannot hide a 7-deep add->sin->cvt->shift->sub->cvt chain behind 3 MFMAs: measured -25 %):
//   A: t = acc + bias, v = sin(2*pi*t)        B: hi = bf16x2(v), ha/hb = hi as fp32        C: lo = bf16x2(v - h)
struct EpiTmp {
    float v0, v1, ha, hb;
};
__device__ __forceinline__ void epi_A(const f32x16& acc, int e, EpiTmp& t) {
    const int i0 = 2 * e, i1 = 2 * e + 1;
#if defined(SNERF_ABLATE) && (ABL & 8)      // timing-only: no transcendental
    t.v0 = acc[i0] * 0.5f;
    t.v1 = acc[i1] * 0.5f;
#else
    t.v0 = sin2pi(acc[i0]);                 // the bias is already in the accumulator (it was its initial value)
    t.v1 = sin2pi(acc[i1]);
#endif
}
// accumulator initialised with the layer bias: 4 ds_read_b128 straight into the accumulator registers, no VALU
__device__ __forceinline__ f32x16 load_bias(lds_cfloat* bias_l, int b, int h) {
    lds_cf32x4* bp = (lds_cf32x4*)(bias_l + b * 32 + h * 16);
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 t = bp[q];
        acc[4 * q] = t[0]; acc[4 * q + 1] = t[1]; acc[4 * q + 2] = t[2]; acc[4 * q + 3] = t[3];
    }
    return acc;
}
__device__ __forceinline__ void epi_B(int e, EpiTmp& t, Frag* out2) {
    // compiler-generated VALU only: an inline-asm v_cvt_pk/v_sub here read stale v_sin results (wrong low parts,
    // 7e-4 instead of 7e-6 on Rho) - hipcc pads the transcendental-use hazard for its own instructions, not for asm
    bf16x2 hv;
    hv[0] = (__bf16)t.v0;
    hv[1] = (__bf16)t.v1;
    const uint32_t hi = __builtin_bit_cast(uint32_t, hv);
    t.ha = __builtin_bit_cast(float, hi << 16);
    t.hb = __builtin_bit_cast(float, hi & 0xffff0000u);
    out2[e >> 2].hi[e & 3] = hi;
}
__device__ __forceinline__ void epi_C(int e, const EpiTmp& t, Frag* out2) {
    bf16x2 lv;
    lv[0] = (__bf16)(t.v0 - t.ha);
    lv[1] = (__bf16)(t.v1 - t.hb);
    out2[e >> 2].lo[e & 3] = __builtin_bit_cast(uint32_t, lv);
}

// LLVM SchedGroupMask bits
#define SG_VALU 0x002
#define SG_MFMA 0x008
#define SG_DSREAD 0x100
#define SG_TRANS 0x400

// PE(pos): 32 slots per lane-half, see slot_feature_PEPOS
__device__ __forceinline__ void make_pe_pos(float x0, float x1, float x2, int h, Frag* pe) {
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
#pragma unroll
    for (int s = 0; s < 4; ++s) pack8(v + 8 * s, pe[s]);
}

// PE(sun): 16 slots per lane-half, see slot_feature_PESUN
__device__ __forceinline__ void make_pe_sun(float x0, float x1, float x2, int h, Frag* pe) {
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
    pack8(v, pe[0]);
    pack8(v + 8, pe[1]);
}

// PE(time[:,0:2]): lane-half h owns coordinate h, see slot_feature_PETIME
__device__ __forceinline__ void make_pe_time(float t0, float t1, int h, Frag* pe) {
    float v[8];
    const float x = h ? t1 : t0;
    const PeArg a = pe_arg(x);
    v[0] = x;
    pe_sincos(a, 1.0, v[1], v[2]);
    pe_sincos(a, 2.0, v[3], v[4]);
    v[5] = v[6] = v[7] = 0.f;
    pack8(v, pe[0]);
    pe[1].hi = u32x4{0, 0, 0, 0};
    pe[1].lo = u32x4{0, 0, 0, 0};
}

}  // namespace snerf
