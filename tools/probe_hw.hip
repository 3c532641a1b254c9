// Hardware probe for design decisions (run on the MI355X box):
//   1. accuracy of v_sin_f32 / v_cos_f32 (input in revolutions) vs double, and of a polynomial alternative
//   2. lane maps of v_mfma_f32_32x32x16_bf16 (A, B, C/D) checked with exact small-integer data
//   3. the "accumulator as next B operand" k-permutation used by the fused MLP chain
// Build: hipcc -O3 --offload-arch=gfx950 tools/probe_hw.hip -o gpurun_out/probe_hw
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ float poly_sin2pi(float r) {
    // r in revolutions, any magnitude within fp32 exactness of fract
    float t = r - rintf(r);                 // [-0.5, 0.5]
    float a = fabsf(t);
    float u = (a > 0.25f) ? (0.5f - a) : a; // fold to [0, 0.25]
    u = copysignf(u, t);
    float z = u * u;
    // sin(2*pi*u) = u * P(z), minimax-ish Taylor coefficients in (2pi)^(2k+1)/(2k+1)!
    float p = -15.094642576822990f;          // -(2pi)^11/11!
    p = fmaf(p, z, 42.058693944897651f);     //  (2pi)^9/9!
    p = fmaf(p, z, -76.705859753061385f);    // -(2pi)^7/7!
    p = fmaf(p, z, 81.605249276075054f);     //  (2pi)^5/5!
    p = fmaf(p, z, -41.341702240399755f);    // -(2pi)^3/3!
    p = fmaf(p, z, 6.283185307179586f);
    return u * p;
}

__global__ void sin_probe(const float* in, float* hw_sin, float* hw_cos, float* hw_fract_sin, float* poly, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r = in[i];
    hw_sin[i] = __builtin_amdgcn_sinf(r);
    hw_cos[i] = __builtin_amdgcn_cosf(r);
    hw_fract_sin[i] = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(r));
    poly[i] = poly_sin2pi(r);
}

// timing: many sin per thread
template <int MODE>
__global__ void sin_bench(float* out, float seed, int iters) {
    float x0 = seed + threadIdx.x * 1e-3f, x1 = x0 + 0.1f, x2 = x0 + 0.2f, x3 = x0 + 0.3f;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { x0 = __builtin_amdgcn_sinf(x0); x1 = __builtin_amdgcn_sinf(x1); x2 = __builtin_amdgcn_sinf(x2); x3 = __builtin_amdgcn_sinf(x3); }
        else if (MODE == 1) { x0 = poly_sin2pi(x0); x1 = poly_sin2pi(x1); x2 = poly_sin2pi(x2); x3 = poly_sin2pi(x3); }
        else { x0 = fmaf(x0, 0.999f, 0.001f); x1 = fmaf(x1, 0.999f, 0.001f); x2 = fmaf(x2, 0.999f, 0.001f); x3 = fmaf(x3, 0.999f, 0.001f); }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
}

// ---- MFMA layout probe: one wave, D = A(32x16) * B(16x32) with integer-valued bf16 data
__global__ void mfma_probe(const float* A, const float* B, float* D) {
    int l = threadIdx.x, r = l & 31, h = l >> 5;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16)A[r * 16 + 8 * h + j];        // lane holds A[row r][k = 8h + j]
        b[j] = (__bf16)B[(8 * h + j) * 32 + r];      // lane holds B[k = 8h + j][col r]
    }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 16; ++i) {
        int row = (i & 3) + 8 * (i >> 2) + 4 * h;    // C/D: col = lane&31, row = (i&3)+8(i>>2)+4h
        D[row * 32 + r] = c[i];
    }
}

// ---- chain probe: Y = W2 * (W1 * X) with the intermediate kept in registers.
// X: [16 k][32 pts], W1: [32 n][16 k], W2: [32 n2][32 k2]; the second product consumes the accumulator
// of the first as B operand with k-slot (s, 8h+j) <-> row 16s + 8(j>>2) + 4h + (j&3).
__global__ void chain_probe(const float* W1, const float* X, const float* W2, float* Y) {
    int l = threadIdx.x, r = l & 31, h = l >> 5;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)W1[r * 16 + 8 * h + j]; b[j] = (__bf16)X[(8 * h + j) * 32 + r]; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);   // c = (W1 X)[n][pt], n in regs, pt on lane
    f32x16 y = {0};
    for (int s = 0; s < 2; ++s) {
        bf16x8 bb, aa;
        for (int j = 0; j < 8; ++j) {
            bb[j] = (__bf16)c[8 * s + j];
            int k = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);          // feature carried by element j
            aa[j] = (__bf16)W2[r * 32 + k];
        }
        y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa, bb, y, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) Y[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = y[i];
}

int main() {
    // ---------------- sin accuracy
    {
        const int n = 1 << 22;
        std::vector<float> in(n);
        // mix: dense in [-1,1], wide in [-40,40], and [0,1)
        for (int i = 0; i < n; ++i) {
            double u = (i + 0.5) / n;
            if (i % 3 == 0) in[i] = (float)(2 * u - 1);
            else if (i % 3 == 1) in[i] = (float)(80 * u - 40);
            else in[i] = (float)u;
        }
        float *d_in, *d_a, *d_b, *d_c, *d_d;
        CK(hipMalloc(&d_in, n * 4)); CK(hipMalloc(&d_a, n * 4)); CK(hipMalloc(&d_b, n * 4)); CK(hipMalloc(&d_c, n * 4)); CK(hipMalloc(&d_d, n * 4));
        CK(hipMemcpy(d_in, in.data(), n * 4, hipMemcpyHostToDevice));
        sin_probe<<<n / 256, 256>>>(d_in, d_a, d_b, d_c, d_d, n);
        CK(hipDeviceSynchronize());
        std::vector<float> a(n), b(n), c(n), d(n);
        CK(hipMemcpy(a.data(), d_a, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), d_b, n * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(c.data(), d_c, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(d.data(), d_d, n * 4, hipMemcpyDeviceToHost));
        double e_sin[3] = {0, 0, 0}, e_cos[3] = {0, 0, 0}, e_fs[3] = {0, 0, 0}, e_p[3] = {0, 0, 0};
        for (int i = 0; i < n; ++i) {
            double x = (double)in[i] * 2.0 * M_PI;
            double s = sin(x), co = cos(x);
            int g = i % 3;
            e_sin[g] = fmax(e_sin[g], fabs(a[i] - s)); e_cos[g] = fmax(e_cos[g], fabs(b[i] - co));
            e_fs[g] = fmax(e_fs[g], fabs(c[i] - s)); e_p[g] = fmax(e_p[g], fabs(d[i] - s));
        }
        const char* nm[3] = {"[-1,1]", "[-40,40]", "[0,1)"};
        for (int g = 0; g < 3; ++g)
            printf("SIN range %-9s max abs err: v_sin %.3e  v_cos %.3e  fract+v_sin %.3e  poly %.3e\n", nm[g], e_sin[g], e_cos[g], e_fs[g], e_p[g]);
        // small-argument relative behaviour
        double rel = 0;
        for (int i = 0; i < n; i += 3) { double x = (double)in[i] * 2 * M_PI; if (fabs(x) > 1e-3 && fabs(x) < 0.3) rel = fmax(rel, fabs(a[i] - sin(x)) / fabs(sin(x))); }
        printf("SIN v_sin max rel err on 1e-3<|x|<0.3 rad: %.3e\n", rel);
    }
    // ---------------- sin timing
    {
        float* d_out; CK(hipMalloc(&d_out, 256 * 1024 * 4 * 4));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int iters = 4096, blocks = 256 * 16;
        for (int mode = 0; mode < 3; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) sin_bench<0><<<blocks, 256>>>(d_out, 0.3f, iters);
                if (mode == 1) sin_bench<1><<<blocks, 256>>>(d_out, 0.3f, iters);
                if (mode == 2) sin_bench<2><<<blocks, 256>>>(d_out, 0.3f, iters);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                double ops = (double)blocks * 256 * iters * 4;
                if (rep == 1) printf("TIMING mode %d (0=v_sin,1=poly,2=fma): %.3f ms, %.2f Gop/s\n", mode, ms, ops / ms * 1e-6);
            }
        }
    }
    // ---------------- MFMA maps
    {
        std::vector<float> A(32 * 16), B(16 * 32), D(32 * 32), Dref(32 * 32, 0.f);
        for (int i = 0; i < 32 * 16; ++i) A[i] = (float)((i * 7 + 3) % 13 - 6);
        for (int i = 0; i < 16 * 32; ++i) B[i] = (float)((i * 5 + 1) % 11 - 5);
        for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) { float s = 0; for (int k = 0; k < 16; ++k) s += A[r * 16 + k] * B[k * 32 + c]; Dref[r * 32 + c] = s; }
        float *dA, *dB, *dD; CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dD, D.size() * 4));
        CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
        mfma_probe<<<1, 64>>>(dA, dB, dD); CK(hipDeviceSynchronize());
        CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0; for (int i = 0; i < 32 * 32; ++i) bad += (D[i] != Dref[i]);
        printf("MFMA 32x32x16 bf16 lane maps: %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
        // chain
        std::vector<float> W1(32 * 16), X(16 * 32), W2(32 * 32), Y(32 * 32), Yref(32 * 32, 0.f), Hh(32 * 32, 0.f);
        for (int i = 0; i < 32 * 16; ++i) W1[i] = (float)((i * 3 + 2) % 5 - 2);
        for (int i = 0; i < 16 * 32; ++i) X[i] = (float)((i * 7 + 1) % 3 - 1);
        for (int i = 0; i < 32 * 32; ++i) W2[i] = (float)((i * 11 + 5) % 7 - 3);
        for (int n = 0; n < 32; ++n) for (int p = 0; p < 32; ++p) { float s = 0; for (int k = 0; k < 16; ++k) s += W1[n * 16 + k] * X[k * 32 + p]; Hh[n * 32 + p] = s; }
        for (int n = 0; n < 32; ++n) for (int p = 0; p < 32; ++p) { float s = 0; for (int k = 0; k < 32; ++k) s += W2[n * 32 + k] * Hh[k * 32 + p]; Yref[n * 32 + p] = s; }
        float *dW1, *dX, *dW2, *dY; CK(hipMalloc(&dW1, W1.size() * 4)); CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dW2, W2.size() * 4)); CK(hipMalloc(&dY, Y.size() * 4));
        CK(hipMemcpy(dW1, W1.data(), W1.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW2, W2.data(), W2.size() * 4, hipMemcpyHostToDevice));
        chain_probe<<<1, 64>>>(dW1, dX, dW2, dY); CK(hipDeviceSynchronize());
        CK(hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost));
        bad = 0; for (int i = 0; i < 32 * 32; ++i) bad += (Y[i] != Yref[i]);
        printf("MFMA accumulator-as-B chain permutation: %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
    }
    return 0;
}
