// Probe: cycles per v_mfma_i32_32x32x32_i8 (one wave per SIMD) for the accumulation patterns of the int8-digit kernel
//   0: one accumulator, every MFMA depends on the previous one
//   1: the kernel's pattern  M, X, X  (two accumulators, X -> X back to back)
//   2: pattern 1 with the X pair split around M:  X, M, X
//   3: pattern 1 + per MFMA one v_sin_f32 and three v_fma_f32 on independent registers (the epilogue's filler load)
//   4: pattern 1 + per MFMA one v_sin and six plain VALU
//   5: pattern 1 + per MFMA two v_sin and two v_fma
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_i8_chain.hip -o build/probes/mfma_i8_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int KIND>
__global__ __launch_bounds__(256, 1) void probe(int iters, float* sink, long long* cyc, float seed) {
    const int l = threadIdx.x & 63;
    i32x4 a = {l, l + 1, l + 2, l + 3}, b = {l * 3, l * 5, l * 7, l * 9};
    i32x16 M, X;
    f32x16 F;
    for (int i = 0; i < 16; ++i) { M[i] = 0; X[i] = 0; F[i] = 0; }
    float sn[24], fm[72];      // independent chains: a filler's result is consumed 8 fillers later (no latency chain)
    for (int i = 0; i < 24; ++i) sn[i] = seed * (i + 1) + l;
    for (int i = 0; i < 72; ++i) fm[i] = seed * (i + 3);
    int fi = 0;
    const bf16x8 ab = __builtin_bit_cast(bf16x8, a), bb = __builtin_bit_cast(bf16x8, b);
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#define FILL3(k) { sn[k] = __builtin_amdgcn_sinf(sn[k]); fm[3*(k)] = __builtin_fmaf(fm[3*(k)], 1.0001f, 0.5f); fm[3*(k)+1] = __builtin_fmaf(fm[3*(k)+1], 1.0001f, 0.25f); fm[3*(k)+2] = __builtin_fmaf(fm[3*(k)+2], 0.9999f, 0.125f); }
#define FILL6(k) { FILL3(k); fm[3*(k)] = __builtin_fmaf(fm[3*(k)], 1.0002f, 0.5f); fm[3*(k)+1] = __builtin_fmaf(fm[3*(k)+1], 1.0002f, 0.25f); fm[3*(k)+2] = __builtin_fmaf(fm[3*(k)+2], 0.9998f, 0.125f); }
#define FILLS2(k) { sn[k] = __builtin_amdgcn_sinf(sn[k]); sn[(k)+12] = __builtin_amdgcn_sinf(sn[(k)+12]); fm[3*(k)] = __builtin_fmaf(fm[3*(k)], 1.0001f, 0.5f); fm[3*(k)+1] = __builtin_fmaf(fm[3*(k)+1], 1.0001f, 0.25f); }
            if (KIND == 0) {
                M = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, M, 0, 0, 0);
                M = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, M, 0, 0, 0);
                M = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, M, 0, 0, 0);
            } else if (KIND == 1) {
                M = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, M, 0, 0, 0);
                X = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, X, 0, 0, 0);
                X = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, X, 0, 0, 0);
            } else if (KIND == 2) {
                X = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, X, 0, 0, 0);
                M = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, M, 0, 0, 0);
                X = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, X, 0, 0, 0);
            } else if (KIND == 3) {
                M = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, M, 0, 0, 0); FILL3(3*(u%4))
                X = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, X, 0, 0, 0); FILL3(3*(u%4)+1)
                X = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, X, 0, 0, 0); FILL3(3*(u%4)+2)
            } else if (KIND == 4) {
                M = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, M, 0, 0, 0); FILL6(3*(u%4))
                X = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, X, 0, 0, 0); FILL6(3*(u%4)+1)
                X = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, X, 0, 0, 0); FILL6(3*(u%4)+2)
            } else {
                M = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, M, 0, 0, 0); FILLS2(3*(u%4))
                X = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, X, 0, 0, 0); FILLS2(3*(u%4)+1)
                X = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, X, 0, 0, 0); FILLS2(3*(u%4)+2)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 24; ++i) s += sn[i];
    for (int i = 0; i < 72; ++i) s += fm[i];
    (void)fi; (void)ab; (void)bb;
    for (int i = 0; i < 16; ++i) s += (float)M[i] + (float)X[i] + F[i];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float* sink; long long* cyc;
    const int nb = 256, iters = 1000;
    CK(hipMalloc(&sink, nb * 256 * 4)); CK(hipMalloc(&cyc, nb * 8));
    const char* names[] = {"i8 one accumulator (dependent chain)", "i8 M,X,X (kernel pattern)", "i8 X,M,X", "i8 M,X,X + 1 sin + 3 fma per MFMA",
                           "i8 M,X,X + 1 sin + 6 fma per MFMA", "i8 M,X,X + 2 sin + 2 fma per MFMA"};
    for (int kind = 0; kind < 6; ++kind) {
        for (int rep = 0; rep < 2; ++rep) {
            switch (kind) {
                case 0: hipLaunchKernelGGL(probe<0>, dim3(nb), dim3(256), 0, 0, iters, sink, cyc, 0.001f); break;
                case 1: hipLaunchKernelGGL(probe<1>, dim3(nb), dim3(256), 0, 0, iters, sink, cyc, 0.001f); break;
                case 2: hipLaunchKernelGGL(probe<2>, dim3(nb), dim3(256), 0, 0, iters, sink, cyc, 0.001f); break;
                case 3: hipLaunchKernelGGL(probe<3>, dim3(nb), dim3(256), 0, 0, iters, sink, cyc, 0.001f); break;
                case 4: hipLaunchKernelGGL(probe<4>, dim3(nb), dim3(256), 0, 0, iters, sink, cyc, 0.001f); break;
                default: hipLaunchKernelGGL(probe<5>, dim3(nb), dim3(256), 0, 0, iters, sink, cyc, 0.001f); break;
            }
            CK(hipDeviceSynchronize());
        }
        std::vector<long long> c(nb);
        CK(hipMemcpy(c.data(), cyc, nb * 8, hipMemcpyDeviceToHost));
        double m = 0; for (auto v : c) m += (double)v; m /= nb;
        printf("%-52s %.2f s_memtime ticks per MFMA\n", names[kind], m / (iters * 24.0));
    }
    return 0;
}
