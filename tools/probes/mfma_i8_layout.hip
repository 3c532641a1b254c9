// Probe for the int8-digit fused MLP:
//   1. operand lane maps of v_mfma_i32_32x32x32_i8 checked with exact asymmetric integer data (two hypotheses for k)
//   2. "accumulator as the next B operand": which k order a lane's 16 accumulator rows give as 16 packed bytes
//   3. v_cvt_pknorm_i16_f32 rounding (nearest? ties?), saturation, and the byte split by v_perm_b32
//   4. cycles per i8 MFMA (back to back, one wave per SIMD) against the bf16 32x32x16 form
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_i8_layout.hip -o gpurun_out/mfma_i8 && gpurun_out/mfma_i8
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// each lane hands over its 16 operand bytes as given (host prepares them per hypothesis)
__global__ void mfma_i8_probe(const i32x4* a_frag, const i32x4* b_frag, int* D) {
    const int l = threadIdx.x;
    i32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0;
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_frag[l], b_frag[l], acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) D[l * 16 + i] = acc[i];
}

__global__ void cvt_probe(const float* in, int n, uint32_t* pk, uint32_t* d1, uint32_t* d2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i * 4 + 3 >= n) return;
    const float a = in[4 * i], b = in[4 * i + 1], c = in[4 * i + 2], d = in[4 * i + 3];
    uint32_t p0, p1;
    asm volatile("v_cvt_pknorm_i16_f32 %0, %1, %2" : "=v"(p0) : "v"(a), "v"(b));
    asm volatile("v_cvt_pknorm_i16_f32 %0, %1, %2" : "=v"(p1) : "v"(c), "v"(d));
    pk[2 * i] = p0;
    pk[2 * i + 1] = p1;
    // digits: x = 256*hi + lo_u, lo_u in [0,255];  d2' = lo_u ^ 0x80 (signed), the +128 is a constant folded elsewhere
    const uint32_t q0 = p0 ^ 0x00800080u, q1 = p1 ^ 0x00800080u;
    // v_perm_b32 D = bytes selected from {S0 (bytes 7..4), S1 (bytes 3..0)}
    d1[i] = __builtin_amdgcn_perm(q1, q0, 0x07050301u);   // high bytes of the four int16
    d2[i] = __builtin_amdgcn_perm(q1, q0, 0x06040200u);   // low bytes
}

template <int KIND>
__global__ __launch_bounds__(256, 1) void rate_probe(int iters, int* sink, long long* cyc) {
    const int l = threadIdx.x & 63;
    i32x4 a = {l, l + 1, l + 2, l + 3}, b = {l * 3, l * 5, l * 7, l * 9};
    i32x16 acc0, acc1;
    f32x16 f0, f1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0; acc1[i] = 0; f0[i] = 0; f1[i] = 0; }
    const bf16x8 ab = __builtin_bit_cast(bf16x8, a), bb = __builtin_bit_cast(bf16x8, b);
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) {
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, acc1, 0, 0, 0);
            } else {
                f0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, f0, 0, 0, 0);
                f1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bb, ab, f1, 0, 0, 0);
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + (int)f0[i] + (int)f1[i];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static int Aval(int r, int k) { return ((r * 7 + k * 3) % 11) - 5; }
static int Bval(int k, int c) { return ((k * 5 + c * 2) % 13) - 6; }

int main() {
    // ---------- 1. lane maps
    // hypothesis 0: lane (r, h) byte j  <->  k = 16h + j          (16 contiguous k per lane half)
    // hypothesis 1: lane (r, h) byte j  <->  k = 8h + (j & 7) + 16 (j >> 3)   (two 32x32x16-style halves)
    std::vector<int> Cref(32 * 32, 0);
    for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) { int s = 0; for (int k = 0; k < 32; ++k) s += Aval(r, k) * Bval(k, c); Cref[r * 32 + c] = s; }
    i32x4 *da, *db; int* dD;
    CK(hipMalloc(&da, 64 * 16)); CK(hipMalloc(&db, 64 * 16)); CK(hipMalloc(&dD, 64 * 16 * 4));
    for (int hyp = 0; hyp < 2; ++hyp) {
        std::vector<int8_t> fa(64 * 16), fb(64 * 16);
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 16; ++j) {
            const int r = l & 31, h = l >> 5;
            const int k = hyp == 0 ? 16 * h + j : 8 * h + (j & 7) + 16 * (j >> 3);
            fa[l * 16 + j] = (int8_t)Aval(r, k);
            fb[l * 16 + j] = (int8_t)Bval(k, r);
        }
        CK(hipMemcpy(da, fa.data(), 64 * 16, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, fb.data(), 64 * 16, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(mfma_i8_probe, dim3(1), dim3(64), 0, 0, da, db, dD);
        std::vector<int> D(64 * 16);
        CK(hipMemcpy(D.data(), dD, 64 * 16 * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5), col = l & 31;
            if (D[l * 16 + i] != Cref[row * 32 + col]) ++bad;
        }
        printf("i8 32x32x32 operand map hypothesis %d (%s): %d of 1024 outputs wrong\n", hyp,
               hyp == 0 ? "k = 16h + j" : "k = 8h + (j&7) + 16(j>>3)", bad);
    }
    // ---------- 3. conversions
    {
        std::vector<float> in;
        const float sc = 1.0f / 32767.0f;
        for (int v = -5; v <= 5; ++v) { in.push_back((v + 0.5f) * sc); in.push_back((v + 0.49f) * sc); in.push_back((v + 0.51f) * sc); in.push_back(v * sc); }
        const float extra[] = {1.0f, -1.0f, 1.5f, -1.5f, 0.999f, -0.999f, 0.5f, -0.5f, 0.25f, 0.1f, -0.1f, 32639.0f / 32767.0f};
        for (float e : extra) in.push_back(e);
        while (in.size() % 4) in.push_back(0.f);
        const int n = (int)in.size();
        float* din; uint32_t *dpk, *d1, *d2;
        CK(hipMalloc(&din, n * 4)); CK(hipMalloc(&dpk, n * 2)); CK(hipMalloc(&d1, n)); CK(hipMalloc(&d2, n));
        CK(hipMemcpy(din, in.data(), n * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(cvt_probe, dim3(1), dim3(64), 0, 0, din, n, dpk, d1, d2);
        std::vector<uint32_t> pk(n / 2), h1(n / 4), h2(n / 4);
        CK(hipMemcpy(pk.data(), dpk, n * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h1.data(), d1, n, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h2.data(), d2, n, hipMemcpyDeviceToHost));
        int bad_digits = 0;
        for (int i = 0; i < n; ++i) {
            const int16_t x = (int16_t)((pk[i / 2] >> (16 * (i & 1))) & 0xffff);
            const int8_t a = (int8_t)((h1[i / 4] >> (8 * (i & 3))) & 0xff), b = (int8_t)((h2[i / 4] >> (8 * (i & 3))) & 0xff);
            const int recon = 256 * (int)a + (int)b + 128;
            if (recon != (int)x) ++bad_digits;
            if (i < 48 || i >= n - 16)
                printf("  pknorm(%+.8f = %+9.3f/32767) = %6d   digits (%4d, %4d) -> %6d%s\n", in[i], in[i] * 32767.0, (int)x, (int)a, (int)b, recon,
                       recon == (int)x ? "" : "  MISMATCH");
        }
        printf("digit split: %d of %d values do not reconstruct\n", bad_digits, n);
    }
    // ---------- 4. rate
    {
        int* sink; long long* cyc;
        const int nb = 256;
        CK(hipMalloc(&sink, nb * 256 * 4)); CK(hipMalloc(&cyc, nb * 8));
        for (int kind = 0; kind < 2; ++kind) {
            const int iters = 2000;
            for (int rep = 0; rep < 2; ++rep) {
                if (kind == 0) hipLaunchKernelGGL(rate_probe<0>, dim3(nb), dim3(256), 0, 0, iters, sink, cyc);
                else hipLaunchKernelGGL(rate_probe<1>, dim3(nb), dim3(256), 0, 0, iters, sink, cyc);
                CK(hipDeviceSynchronize());
            }
            std::vector<long long> c(nb);
            CK(hipMemcpy(c.data(), cyc, nb * 8, hipMemcpyDeviceToHost));
            double m = 0; for (auto v : c) m += (double)v; m /= nb;
            printf("%s: %.2f s_memtime ticks per MFMA (one wave per SIMD, 16 back to back per iteration)\n",
                   kind == 0 ? "v_mfma_i32_32x32x32_i8 " : "v_mfma_f32_32x32x16_bf16", m / (iters * 16.0));
        }
    }
    return 0;
}
