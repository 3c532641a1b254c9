// How well do MFMA and VALU work overlap on one SIMD?  One workgroup per CU (LDS-bound occupancy, as in the row GEMM),
// WAVES waves per workgroup (4 = one per SIMD, 8 = two per SIMD); every wave runs ITER iterations of
//   NM x v_mfma_f32_32x32x16_bf16  interleaved with  NV x (the bf16 hi/lo split's VALU mix)   [+ NL x ds_read_b128]
// and reports cycles per iteration (s_memtime around the loop, wave 0 of block 0).  No global memory traffic in the loop.
// build: hipcc -O3 --offload-arch=gfx950 -o mfma_valu_overlap mfma_valu_overlap.hip ; run: ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int NM, int NV, int NL>
__global__ __launch_bounds__(512) void probe(float* out, uint64_t* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((uint32_t*)lds)[i] = i * 2654435761u;
    __syncthreads();
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = 1.0f + lane * 0.001f + e;
    u32x4 a = {1u, 2u, 3u, 4u}, b = {5u, 6u, 7u, 8u};
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        // NV "split" groups: 7 VALU each on a pair of values (cvt_pk, shift, and, 2 sub, cvt_pk + feedback add)
        uint32_t hi[4], lo[4];
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            float x = v[(2 * q) & 7], y = v[(2 * q + 1) & 7];
            typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
            bf2 hv; hv[0] = (__bf16)x; hv[1] = (__bf16)y;
            const uint32_t h = __builtin_bit_cast(uint32_t, hv);
            const float hx = __builtin_bit_cast(float, h << 16), hy = __builtin_bit_cast(float, h & 0xffff0000u);
            bf2 lv; lv[0] = (__bf16)(x - hx); lv[1] = (__bf16)(y - hy);
            if (q < 4) { hi[q] = h; lo[q] = __builtin_bit_cast(uint32_t, lv); }
            else { hi[q & 3] ^= h; lo[q & 3] ^= __builtin_bit_cast(uint32_t, lv); }
            v[(2 * q) & 7] = x * 1.0001f;
        }
        if (NV) {
#pragma unroll
            for (int q = 0; q < (NV < 4 ? NV : 4); ++q) { a[q] ^= hi[q] & 1u; b[q] ^= lo[q] & 1u; }
        }
        u32x4 w[NL ? NL : 1];
#pragma unroll
        for (int l = 0; l < NL; ++l) w[l] = *(const u32x4*)(lds + ((it * 8 + l) & 63) * 1024 + lane * 16);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const bf16x8 A = __builtin_bit_cast(bf16x8, a), B = __builtin_bit_cast(bf16x8, NL ? w[m % (NL ? NL : 1)] : b);
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[m & 3], 0, 0, 0);
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    for (int e = 0; e < 8; ++e) s += v[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }
}

template <int NM, int NV, int NL>
static void run(int waves, float* out, uint64_t* cyc) {
    const int iters = 2000;
    auto k = probe<NM, NV, NL>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 128 * 1024, 0, out, cyc, iters);
    hipDeviceSynchronize();
    uint64_t c[32];
    hipMemcpy(c, cyc, 16 * waves, hipMemcpyDeviceToHost);
    uint64_t s0 = c[0], e1 = 0;
    for (int w = 0; w < waves; ++w) { if (c[2 * w] < s0) s0 = c[2 * w]; if (c[2 * w + 1] > e1) e1 = c[2 * w + 1]; }
    printf("waves/WG %d  MFMA %2d  VALU-groups %2d (x7 ops)  ds_read_b128 %d : per iteration  wave0 %7.1f  wave%d %7.1f  workgroup %7.1f ticks\n", waves, NM, NV, NL,
           (double)(c[1] - c[0]) / iters, waves - 4, (double)(c[2 * (waves - 4) + 1] - c[2 * (waves - 4)]) / iters, (double)(e1 - s0) / iters);
}

int main() {
    float* out; uint64_t* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256);
    for (int waves : {4, 8}) {
        run<12, 0, 0>(waves, out, cyc);
        run<0, 8, 0>(waves, out, cyc);
        run<12, 4, 0>(waves, out, cyc);
        run<12, 8, 0>(waves, out, cyc);
        run<12, 8, 8>(waves, out, cyc);
        run<12, 0, 8>(waves, out, cyc);
    }
    // memtime tick rate: compare with a known-duration loop
    return 0;
}
