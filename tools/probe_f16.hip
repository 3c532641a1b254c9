// Probe for the fp16 split: (1) does hipcc emit v_cvt_pkrtz + v_fma_mixlo/hi_f16 for the hi/lo split written in C++?
// (2) does v_mfma_f32_32x32x16_f16 keep subnormal fp16 inputs?  (3) accuracy of hi+lo vs the fp32 value.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ void split2_f16(float a, float b, unsigned& hi, unsigned& lo) {
    const f16x2 h = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a, b));
    f16x2 l;
    l[0] = (_Float16)__builtin_fmaf(a, 1.0f, -(float)h[0]);
    l[1] = (_Float16)__builtin_fmaf(b, 1.0f, -(float)h[1]);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
__global__ void split_probe(const float* in, unsigned* hi, unsigned* lo, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 < n) split2_f16(in[2 * i], in[2 * i + 1], hi[i], lo[i]);
}
__global__ void mfma_denorm_probe(float a_val, float b_val, float* out) {
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)a_val; b[j] = (_Float16)b_val; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    const int n = 1 << 16;
    float* h_in = new float[n];
    for (int i = 0; i < n; ++i) h_in[i] = (float)sin(0.37 * i) * (i % 7 == 0 ? 1e-3f : 1.f);
    float* d_in; unsigned *d_hi, *d_lo; float* d_out;
    hipMalloc(&d_in, n * 4); hipMalloc(&d_hi, n * 2); hipMalloc(&d_lo, n * 2); hipMalloc(&d_out, 16);
    hipMemcpy(d_in, h_in, n * 4, hipMemcpyHostToDevice);
    split_probe<<<n / 2 / 256, 256>>>(d_in, d_hi, d_lo, n);
    unsigned short* hi = new unsigned short[n]; unsigned short* lo = new unsigned short[n];
    hipMemcpy(hi, d_hi, n * 2, hipMemcpyDeviceToHost); hipMemcpy(lo, d_lo, n * 2, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < n; ++i) {
        _Float16 a, b; memcpy(&a, &hi[i], 2); memcpy(&b, &lo[i], 2);
        double rec = (double)(float)a + (double)(float)b;
        double err = fabs(rec - (double)h_in[i]) / fmax(fabs((double)h_in[i]), 1e-3);
        if (err > worst) worst = err;
    }
    printf("fp16 hi+lo reconstruction: worst error relative to max(|x|,1e-3) = %.3e\n", worst);
    // 1e-6 is subnormal in fp16 (min normal 6.1e-5): 16 products of 1e-6 * 1.0
    mfma_denorm_probe<<<1, 64>>>(1e-6f, 1.0f, d_out);
    float r; hipMemcpy(&r, d_out, 4, hipMemcpyDeviceToHost);
    printf("MFMA f16 subnormal A (1e-6 x 1.0 x16): %.6e (expected ~1.6e-5 if not flushed)\n", r);
    mfma_denorm_probe<<<1, 64>>>(1.0f, 1e-6f, d_out);
    hipMemcpy(&r, d_out, 4, hipMemcpyDeviceToHost);
    printf("MFMA f16 subnormal B (1.0 x 1e-6 x16): %.6e\n", r);
    return 0;
}
