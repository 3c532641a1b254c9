// Device helpers of the int8-digit fused kernels (kernels_i8.hip: one wave per SIMD; kernels_i8x2.hip: two): digit
// fragments, the three-MFMA digit product, per-row tables, the epilogue arithmetic, the encodings as digit fragments.
#pragma once
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

// fp32 weights of the raw coordinates (layers that read an encoding): table [quad][dim 0..2][4 elements] behind the layer's
// scale / bias table (program.h prog_table_start); element i = 4 g + j
struct Raw8 {
    f32x16 w[3];
};
__device__ __forceinline__ Raw8 load_raw(lds_cfloat* raw_l, int b, int h) {
    lds_cf32x4* tp = (lds_cf32x4*)(raw_l + (b * 2 + h) * 48);
    Raw8 r;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const f32x4 t = tp[g * 3 + d];
            r.w[d][4 * g] = t[0]; r.w[d][4 * g + 1] = t[1]; r.w[d][4 * g + 2] = t[2]; r.w[d][4 * g + 3] = t[3];
        }
    return r;
}
__device__ __forceinline__ float add_raw(float z, const Raw8& r, int i, const float* x) {
    return __builtin_fmaf(r.w[0][i], x[0], __builtin_fmaf(r.w[1][i], x[1], __builtin_fmaf(r.w[2][i], x[2], z)));
}

// pre-activation (revolutions for sine layers) of accumulator element i
__device__ __forceinline__ float preact(const Acc8& acc, const Tab8& t, int i) {
    const int m = (int)(((uint32_t)acc.M[i] << 8) + (uint32_t)acc.X[i]);
    return __builtin_fmaf((float)m, t.sc[i], t.bi[i]);
}
// epilogue pieces of a 32x32 block: A(e) = elements 2e, 2e+1 through the sine; Q(g) = digits of elements 4g..4g+3
template <bool RAW = false>
__device__ __forceinline__ void epi_A(const Acc8& acc, const Tab8& t, int e, float* ev, const Raw8* rw = nullptr, const float* rx = nullptr) {
    if constexpr (RAW) {
        ev[2 * e] = sin2pi(add_raw(preact(acc, t, 2 * e), *rw, 2 * e, rx));
        ev[2 * e + 1] = sin2pi(add_raw(preact(acc, t, 2 * e + 1), *rw, 2 * e + 1, rx));
        return;
    }
#if defined(SNERF_ABLATE) && (ABL & 8)      // timing-only: no transcendental; 2 fract(z) - 1 keeps the data as random as sin does
    ev[2 * e] = __builtin_fmaf(__builtin_amdgcn_fractf(preact(acc, t, 2 * e)), 2.f, -1.f);
    ev[2 * e + 1] = __builtin_fmaf(__builtin_amdgcn_fractf(preact(acc, t, 2 * e + 1)), 2.f, -1.f);
#elif defined(SNERF_ABLATE) && (ABL & 16)   // timing-only: no epilogue arithmetic at all but the digit split
    ev[2 * e] = __builtin_bit_cast(float, (acc.X[2 * e] & 0x007fffff) | 0x3f000000) - 0.75f;
    ev[2 * e + 1] = __builtin_bit_cast(float, (acc.X[2 * e + 1] & 0x007fffff) | 0x3f000000) - 0.75f;
#else
    ev[2 * e] = sin2pi(preact(acc, t, 2 * e));
    ev[2 * e + 1] = sin2pi(preact(acc, t, 2 * e + 1));
#endif
}

// PE(pos): 32 values per lane-half (program.h pepos_feature), two k-steps
template <bool OPQ = false>
__device__ __forceinline__ void make_pe_pos8(float x0, float x1, float x2, int h, Frag8* pe) {
    float v[32];
    const float xs[3] = {x0, x1, x2};
    // lane-half h evaluates frequencies 2^(5h) .. 2^(5h+4).  The five exponents are loop-invariant per lane: left alone, hipcc hoists them out of the
    // persistent tile loop and keeps five registers alive across the whole MFMA chain (the ray-visibility variant of the two-wave kernel spilled them);
    // OPQ makes them opaque: recomputed per tile (five v_add per 32 points).  Off for the other variants (their allocation is at its edge as it is).
    int hv = h;
    if constexpr (OPQ) asm volatile("" : "+v"(hv));
    const int e0 = 5 * hv;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const PeArg a = pe_arg(xs[d]);
#pragma unroll
        for (int q = 0; q < 5; ++q) pe_sincos_exp(a, e0 + q, v[10 * d + 2 * q], v[10 * d + 2 * q + 1]);
        __builtin_amdgcn_sched_barrier(0);        // one coordinate's fp64 reductions at a time (register pressure)
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
    const int e0 = 2 * h;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const PeArg a = pe_arg(xs[d]);
#pragma unroll
        for (int q = 0; q < 2; ++q) pe_sincos_exp(a, e0 + q, v[4 * d + 2 * q], v[4 * d + 2 * q + 1]);
    }
    v[12] = h ? x2 : x0;
    v[13] = h ? 0.f : x1;
    v[14] = 0.f;
    v[15] = 0.f;
    pack16(v, pe[0]);
}

// PE(time[:,0:2]): lane-half h owns coordinate h (petime_feature), one k-step
__device__ __forceinline__ void make_pe_time8(float t0, float t1, int h, Frag8* pe) {
    float v[16];
    const float x = h ? t1 : t0;
    const PeArg a = pe_arg(x);
    v[0] = x;
    pe_sincos(a, 1.0, v[1], v[2]);
    pe_sincos(a, 2.0, v[3], v[4]);
#pragma unroll
    for (int i = 5; i < 16; ++i) v[i] = 0.f;
    pack16(v, pe[0]);
}

}  // namespace snerf
