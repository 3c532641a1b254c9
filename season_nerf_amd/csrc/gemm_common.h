// Shared device helpers of the bf16x3 row GEMMs (gemm.hip, gemm16.hip): vector types, the bf16 hi/lo split, tile constants.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace snerf {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

constexpr int RO_WAVES = 8, RO_MT = 1;                       // 8 waves x 32 rows = 256 rows per workgroup tile (64 accumulators per lane)
constexpr int RO_ROWS = RO_WAVES * RO_MT * 32;

__device__ __forceinline__ void split2_bf16(float a, float b, uint32_t& hi, uint32_t& lo) {
    bf16x2_t hv;
    hv[0] = (__bf16)a;
    hv[1] = (__bf16)b;
    hi = __builtin_bit_cast(uint32_t, hv);
    const float ha = __builtin_bit_cast(float, hi << 16);
    const float hb = __builtin_bit_cast(float, hi & 0xffff0000u);
    bf16x2_t lv;
    lv[0] = (__bf16)(a - ha);
    lv[1] = (__bf16)(b - hb);
    lo = __builtin_bit_cast(uint32_t, lv);
}

}  // namespace snerf
