// Probe: how many plain VALU instructions hide in the shadow of one MFMA (one wave per SIMD, dependent accumulator chain),
// for v_mfma_f32_32x32x16_bf16 against v_mfma_i32_32x32x32_i8.  Fillers are independent v_mul_f32 issued through inline asm
// (the compiler can neither pack nor move them).  Prints s_memtime ticks per MFMA for 0..8 fillers per gap.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_shadow.hip -o build/probes/mfma_shadow
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int KIND, int NF>
__global__ __launch_bounds__(256, 1) void probe(int iters, float* sink, long long* cyc) {
    const int l = threadIdx.x & 63;
    i32x4 a = {l, l + 1, l + 2, l + 3}, b = {l * 3, l * 5, l * 7, l * 9};
    i32x16 M;
    f32x16 F;
    for (int i = 0; i < 16; ++i) { M[i] = 0; F[i] = 0; }
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = 1.0f + i + l;
    const bf16x8 ab = __builtin_bit_cast(bf16x8, a), bb = __builtin_bit_cast(bf16x8, b);
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND == 0) F = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, F, 0, 0, 0);
            else M = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, M, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NF; ++k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[k]) : "v"(1.0001f));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += f[i];
    for (int i = 0; i < 16; ++i) s += (float)M[i] + F[i];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int NF>
double run(float* sink, long long* cyc) {
    const int nb = 256, iters = 1000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((probe<KIND, NF>), dim3(nb), dim3(256), 0, 0, iters, sink, cyc);
        CK(hipDeviceSynchronize());
    }
    std::vector<long long> c(nb);
    CK(hipMemcpy(c.data(), cyc, nb * 8, hipMemcpyDeviceToHost));
    double m = 0; for (auto v : c) m += (double)v; m /= nb;
    return m / (iters * 16.0);
}

int main() {
    float* sink; long long* cyc;
    CK(hipMalloc(&sink, 256 * 256 * 4)); CK(hipMalloc(&cyc, 256 * 8));
    printf("fillers/gap:              0      1      2      3      4      5      6      8\n");
    printf("bf16 32x32x16:       %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f\n", run<0, 0>(sink, cyc), run<0, 1>(sink, cyc), run<0, 2>(sink, cyc),
           run<0, 3>(sink, cyc), run<0, 4>(sink, cyc), run<0, 5>(sink, cyc), run<0, 6>(sink, cyc), run<0, 8>(sink, cyc));
    printf("i8   32x32x32:       %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f\n", run<1, 0>(sink, cyc), run<1, 1>(sink, cyc), run<1, 2>(sink, cyc),
           run<1, 3>(sink, cyc), run<1, 4>(sink, cyc), run<1, 5>(sink, cyc), run<1, 6>(sink, cyc), run<1, 8>(sink, cyc));
    return 0;
}
