// Elementwise / reduction kernels of the layer-wise training path (see train.h).  All HBM-bound: rows of
// [points x features] fp32 matrices are read with consecutive lanes on consecutive features (coalesced), per-column
// reductions go block-partial -> one fp32 atomic per column per block.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "train.h"

namespace snerf {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define LAUNCH_1D(kernel, n, st, ...)                                                    \
    do {                                                                                 \
        const int64_t _n = (n);                                                          \
        if (_n > 0) {                                                                    \
            int64_t _b = (_n + 255) / 256;                                               \
            if (_b > 65536 * 16) _b = 65536 * 16;                                        \
            hipLaunchKernelGGL(kernel, dim3((unsigned)_b), dim3(256), 0, st, __VA_ARGS__); \
        }                                                                                \
    } while (0)

__device__ __forceinline__ float sigmoid_t(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float softplus_t(float x) { return x > 20.f ? x : log1pf(expf(x)); }

// sin/cos of 2^j * fl32(fl32(pi/2)*x), reduced in fp64 (same construction as the inference kernel, misc.py:109-131)
__device__ __forceinline__ void pe_sc(float x, int j, float& c, float& s) {
    const float a0 = __fmul_rn(x, 1.57079637050628662109375f);
    const double r = (double)a0 * 0.15915494309189533576888 * (double)(1 << j);
    const double f = r - floor(r);
    const double ang = f * 6.283185307179586476925287;
    c = (float)cos(ang);
    s = (float)sin(ang);
}

// one lane per output feature: a wavefront writes one 256-byte row of PE(pos) per pass (coalesced; the thread-per-point form
// scattered 64 dword stores over 64 rows and moved 13x the bytes).  cos(a) is evaluated as sin(a + 1/4 revolution): the reduced
// argument f is exact in fp64, so is f + 0.25
__global__ void pe_points_kernel(const PeArgs A) {
    const int64_t total = A.n * 64;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e >> 6;
        const int f = (int)(e & 63);
        const int k = f - 3;
        const int d = f < 3 ? f : (f == 63 ? 0 : k / 20);
        float x;
        if (A.points) {
            x = A.points[i * 3 + d];
        } else {
            const int64_t r = i / A.n_samples;
            const int s = (int)(i - r * A.n_samples);
            const float t = A.tvals[s], omt = __fsub_rn(1.f, t);
            x = __fadd_rn(__fmul_rn(A.top[r * 3 + d], omt), __fmul_rn(A.bot[r * 3 + d], t));
        }
        float v;
        if (f < 3) {
            v = x;
            if (A.pts) A.pts[i * 3 + f] = x;
        } else if (f == 63) {
            v = 0.f;
        } else {
            const int q = k - 20 * d;               // 0..9 cos, 10..19 sin
            const int j = q < 10 ? q : q - 10;
            const float a0 = __fmul_rn(x, 1.57079637050628662109375f);
            const double r = (double)a0 * 0.15915494309189533576888 * (double)(1 << j);
            double fr = r - floor(r);
            if (q < 10) fr += 0.25;
            // fold the exact fraction of a revolution into [-1/4, 1/4] (still exact), then one fp32 sinpi: the only rounding is the
            // fp32 conversion of the folded argument (relative 2^-24)
            double g = fr - rint(fr);
            if (fabs(g) > 0.25) g = copysign(0.5, g) - g;
            v = sinpif(2.f * (float)g);
        }
        A.pe[e] = v;
        if (A.pe2) A.pe2[i * A.ld2 + f] = v;
    }
}
hipError_t launch_pe_points(const PeArgs& a, hipStream_t st) {
    LAUNCH_1D(pe_points_kernel, a.n * 64, st, a);
    return hipGetLastError();
}

__global__ void pe_small_kernel(const float* in, int in_stride, int D, int F, int64_t rows, float* out, int out_stride) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < rows; i += (int64_t)gridDim.x * blockDim.x) {
        float* o = out + i * out_stride;
        for (int d = 0; d < D; ++d) {
            const float x = in[i * in_stride + d];
            o[d] = x;
            for (int j = 0; j < F; ++j) {
                float c, s;
                pe_sc(x, j, c, s);
                o[D + 2 * F * d + j] = c;
                o[D + 2 * F * d + F + j] = s;
            }
        }
        for (int k = D * (2 * F + 1); k < out_stride; ++k) o[k] = 0.f;
    }
}
hipError_t launch_pe_small(const float* in, int in_stride, int n_dims, int n_freq, int64_t rows, float* out, int out_stride, hipStream_t st) {
    LAUNCH_1D(pe_small_kernel, rows, st, in, in_stride, n_dims, n_freq, rows, out, out_stride);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// column reductions: block = 256 threads, thread t owns column (t % CP) and row phase (t / CP), CP = min(C, 256) rounded
constexpr int ROWS_PER_BLOCK = 512;
__global__ __launch_bounds__(256) void colreduce_kernel(const ColArgs A) {
    __shared__ float red0[256], red1[256];
    const int C = A.C;
    const int64_t ldd = A.ldd ? A.ldd : A.ld;
    const int cp = C >= 256 ? 256 : (C >= 128 ? 128 : (C >= 64 ? 64 : 32));   // columns handled per pass
    const int phases = 256 / cp;
    const int tc = threadIdx.x % cp, tp = threadIdx.x / cp;
    const int64_t r0 = (int64_t)blockIdx.x * ROWS_PER_BLOCK;
    const int64_t r1 = r0 + ROWS_PER_BLOCK < A.M ? r0 + ROWS_PER_BLOCK : A.M;
    for (int c0 = 0; c0 < C; c0 += cp) {
        const int c = c0 + tc;
        float s0 = 0.f, s1 = 0.f;
        if (c < C) {
            const float mu = A.mu ? A.mu[c] : 0.f, istd = A.istd ? A.istd[c] : 1.f;
            const float gm = A.gamma ? A.gamma[c] : 1.f, bt = A.beta ? A.beta[c] : 0.f;
            for (int64_t r = r0 + tp; r < r1; r += phases) {
                const float z = A.Z[r * A.ld + c];
                if (A.mode == 0) {
                    const float d = z - mu;
                    s0 += d * d;
                } else if (A.mode == 1) {
                    const float xh = (z - mu) * istd;
                    const float dy = A.D[r * ldd + c] * cosf(gm * xh + bt);       // not stored: the dZ pass recomputes it
                    s0 += dy;
                    s1 += dy * xh;
                } else {
                    const float dz = A.D[r * ldd + c] * cosf(z);
                    A.D[r * ldd + c] = dz;
                    s0 += dz;
                }
            }
        }
        red0[threadIdx.x] = s0;
        red1[threadIdx.x] = s1;
        __syncthreads();
        if (tp == 0 && c < C) {
            for (int p = 1; p < phases; ++p) { s0 += red0[p * cp + tc]; s1 += red1[p * cp + tc]; }
            atomicAdd(A.out0 + c, (A.mode == 2 ? A.alpha0 : 1.f) * s0);
            if (A.mode == 1) atomicAdd(A.out1 + c, s1);
        }
        __syncthreads();
    }
}
// 16-byte variant of the column passes of a SineLayer backward (C, ld multiples of 4, aligned bases): a thread owns 4
// columns (constants in registers) and walks down a 512-row chunk; partial sums meet in LDS, one atomic per column and block.
//   MODE 1  BatchNorm reduction: sum dY, sum dY*xhat with dY = dH*cos(gamma*xhat+beta)   (reads Z, D; writes nothing)
//   MODE 2  plain layer: D <- dH*cos(z), out0 += alpha0 * sum
//   MODE 3  BatchNorm dZ: D <- gamma*istd*(dY - mean(dY) - xhat*mean(dY*xhat)), out0 += alpha0 * sum, dY = dH*cos(.) recomputed
//   MODE 4  the same with D already holding dY (a dgrad epilogue applied the cosine)
template <int MODE>
__global__ __launch_bounds__(256) void colpass_vec_kernel(const ColArgs A, int C4, int cpt, const float* sdy, const float* sdyx, int rows_per_block) {
    __shared__ float red[2][256][4];
    const int tc = threadIdx.x % cpt, tr = threadIdx.x / cpt, rows_pass = 256 / cpt;
    const bool live = tc < C4;
    const int64_t ldd = A.ldd ? A.ldd : A.ld;
    float mu[4] = {0.f, 0.f, 0.f, 0.f}, is[4] = {1.f, 1.f, 1.f, 1.f}, gm[4] = {1.f, 1.f, 1.f, 1.f}, bt[4] = {0.f, 0.f, 0.f, 0.f};
    float ma[4] = {0.f, 0.f, 0.f, 0.f}, mb[4] = {0.f, 0.f, 0.f, 0.f};
    if (live && MODE != 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = tc * 4 + q;
            mu[q] = A.mu[c]; is[q] = A.istd[c]; gm[q] = A.gamma[c]; bt[q] = A.beta[c];
            if (MODE >= 3) { const float invM = 1.f / (float)A.M_global; ma[q] = sdy[c] * invM; mb[q] = sdyx[c] * invM; }
        }
    }
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < A.M ? r0 + rows_per_block : A.M;
    if (live) {
        for (int64_t r = r0 + tr; r < r1; r += rows_pass) {
            const f32x4_t z = *(const f32x4_t*)(A.Z + r * A.ld + tc * 4);
            f32x4_t d = *(const f32x4_t*)(A.D + r * ldd + tc * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (MODE == 2) {
                    d[q] *= cosf(z[q]);
                    s0[q] += d[q];
                } else {
                    const float xh = (z[q] - mu[q]) * is[q];
                    const float dy = MODE == 4 ? d[q] : d[q] * cosf(gm[q] * xh + bt[q]);
                    if (MODE == 1) {
                        s0[q] += dy;
                        s1[q] += dy * xh;
                    } else {
                        d[q] = (gm[q] * is[q]) * (dy - ma[q] - xh * mb[q]);
                        s0[q] += d[q];
                    }
                }
            }
            if (MODE != 1) *(f32x4_t*)(A.D + r * ldd + tc * 4) = d;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { red[0][threadIdx.x][q] = s0[q]; red[1][threadIdx.x][q] = s1[q]; }
    __syncthreads();
    if (tr == 0 && live) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float a = 0.f, b = 0.f;
            for (int p = 0; p < rows_pass; ++p) { a += red[0][p * cpt + tc][q]; b += red[1][p * cpt + tc][q]; }
            const int c = tc * 4 + q;
            if (A.out0) atomicAdd(A.out0 + c, (MODE == 1 ? 1.f : A.alpha0) * a);
            if (MODE == 1) atomicAdd(A.out1 + c, b);
        }
    }
}
static bool colpass_vec_ok(const ColArgs& a) {
    return a.C % 4 == 0 && a.C <= 1024 && a.ld % 4 == 0 && a.ldd % 4 == 0 && (uintptr_t)a.Z % 16 == 0 && (uintptr_t)a.D % 16 == 0;
}
template <int MODE>
static hipError_t launch_colpass_vec(const ColArgs& a, const float* sdy, const float* sdyx, hipStream_t st) {
    const int C4 = a.C / 4;
    int cpt = 1;
    while (cpt < C4) cpt <<= 1;
    // 512 rows per block for the per-point arrays; the per-ray branches (a few thousand rows) get shorter blocks so that they still fill the chip
    int rpb = ROWS_PER_BLOCK;
    while (rpb > 32 && rpb > 256 / cpt && (a.M + rpb - 1) / rpb < 512) rpb >>= 1;
    const int64_t blocks = (a.M + rpb - 1) / rpb;
    hipLaunchKernelGGL(colpass_vec_kernel<MODE>, dim3((unsigned)blocks), dim3(256), 0, st, a, C4, cpt, sdy, sdyx, rpb);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Input gradient of a head with K <= 4 outputs (colour 3, density 1, solar visibility 1):
//   C[m, n] = (accumulate ? C[m, n] : 0) + alpha * sum_k D[m, k] W[k, n],   then optionally the activation backward of the layer below
//   (C *= cos(2 pi (a z + b)), column sums of C and C * xhat as the row GEMMs' epilogue leaves them).
// A rank-K update is a stream over C, not a GEMM: exact fp32 FMAs, a thread owns four columns (its K x 4 weights in registers) and walks
// down the rows; the row GEMM spent 230-260 us on each of these at 4096 x 96, the stream 90-135 us.  (With K = 12 - the class adjustments,
// 256 columns - the same stream took 400 us against the GEMM's 186: 48 FMAs per 16 bytes; those stay on the GEMM.)
template <bool ACT, int KMAX>      // KMAX: 4
__global__ __launch_bounds__(256) void thin_dgrad_kernel(const ThinDgradArgs A, int C4, int cpt, int rows_per_block) {
    __shared__ float red[2][256][4];
    const int tc = threadIdx.x % cpt, tr = threadIdx.x / cpt, rows_pass = 256 / cpt;
    const bool live = tc < C4;
    float w[KMAX][4];
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) w[k][q] = (live && k < A.K) ? ((k == 3 && A.W3) ? A.W3[tc * 4 + q] : A.W[(int64_t)k * A.ldw + tc * 4 + q]) : 0.f;
    float ea[4] = {0.f, 0.f, 0.f, 0.f}, eb[4] = {0.f, 0.f, 0.f, 0.f}, mu[4] = {0.f, 0.f, 0.f, 0.f}, is[4] = {0.f, 0.f, 0.f, 0.f};
    if (ACT && live) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = tc * 4 + q;
            ea[q] = A.etab[c]; eb[q] = A.etab[A.N + c];
            if (A.emu) { mu[q] = A.emu[c]; is[q] = A.eistd[c]; }
        }
    }
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    // persistent blocks over chunks of rows_per_block rows (the column sums leave each block once); U rows of a thread in flight at a time
    constexpr int U = KMAX <= 4 ? 4 : 2;                    // (register budget: three waves per SIMD)
    const bool vec_d = (A.ldd & 3) == 0 && ((uintptr_t)A.D & 15) == 0 && ((A.K + 3) & ~3) <= A.ldd;
    const int64_t n_chunks = (A.M + rows_per_block - 1) / rows_per_block;
    if (live) {
        for (int64_t ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
            const int64_t r0 = ch * rows_per_block;
            const int64_t r1 = r0 + rows_per_block < A.M ? r0 + rows_per_block : A.M;
            for (int64_t rb = r0 + tr; rb < r1; rb += (int64_t)U * rows_pass) {
                float d[U][KMAX];
                f32x4_t cold[U], z[U];
                bool ok[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int64_t r = rb + (int64_t)u * rows_pass;
                    ok[u] = r < r1;
                    const int64_t rc = ok[u] ? r : r1 - 1;
                    if (vec_d) {                                         // uniform: the K values of a row as 16-byte loads
#pragma unroll
                        for (int k4 = 0; k4 < KMAX / 4; ++k4) {
                            f32x4_t t4 = {0.f, 0.f, 0.f, 0.f};
                            if (4 * k4 < A.K) t4 = *(const f32x4_t*)(A.D + rc * A.ldd + 4 * k4);      // (columns past K: multiplied by zero weights)
#pragma unroll
                            for (int q = 0; q < 4; ++q) d[u][4 * k4 + q] = t4[q];
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < KMAX; ++k) d[u][k] = k < A.K ? A.D[rc * A.ldd + k] : 0.f;
                    }
                    if (A.accumulate) cold[u] = *(const f32x4_t*)(A.C + rc * A.ldc + tc * 4);
                    if (ACT) z[u] = *(const f32x4_t*)(A.ez + rc * A.eld + tc * 4);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) {
                        if (k < A.K) {                                  // uniform
#pragma unroll
                            for (int q = 0; q < 4; ++q) acc[q] = __builtin_fmaf(d[u][k], w[k][q], acc[q]);
                        }
                    }
                    f32x4_t v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = A.alpha * acc[q];
                    if (A.accumulate) v += cold[u];
                    if (ACT) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            v[q] *= __builtin_amdgcn_cosf(__builtin_fmaf(ea[q], z[u][q], eb[q]));
                            s0[q] += ok[u] ? v[q] : 0.f;
                            s1[q] += ok[u] ? v[q] * ((z[u][q] - mu[q]) * is[q]) : 0.f;
                        }
                    }
                    if (ok[u]) *(f32x4_t*)(A.C + (rb + (int64_t)u * rows_pass) * A.ldc + tc * 4) = v;
                }
            }
        }
    }
    if (ACT) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { red[0][threadIdx.x][q] = s0[q]; red[1][threadIdx.x][q] = s1[q]; }
        __syncthreads();
        if (tr == 0 && live) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                double a = 0.0, b = 0.0;
                for (int p = 0; p < rows_pass; ++p) { a += (double)red[0][p * cpt + tc][q]; b += (double)red[1][p * cpt + tc][q]; }
                const int c = tc * 4 + q;
                atomicAdd(A.stats + c, a);
                atomicAdd(A.stats + A.N + c, b);
            }
        }
    }
}
bool thin_dgrad_ok(const ThinDgradArgs& a) {
    return a.K >= 1 && a.K <= 4 && a.N % 4 == 0 && a.N <= 1024 && a.ldc % 4 == 0 && (uintptr_t)a.C % 16 == 0 &&
           (!a.ez || (a.eld % 4 == 0 && (uintptr_t)a.ez % 16 == 0 && a.etab && a.stats));
}
hipError_t launch_thin_dgrad(const ThinDgradArgs& a, hipStream_t st) {
    if (a.M <= 0 || a.N <= 0) return hipSuccess;
    if (!thin_dgrad_ok(a)) return hipErrorInvalidValue;
    const int C4 = a.N / 4;
    int cpt = 1;
    while (cpt < C4) cpt <<= 1;
    int rpb = 256;
    while (rpb > 32 && rpb > 256 / cpt && (a.M + rpb - 1) / rpb < 1024) rpb >>= 1;
    int64_t blocks = (a.M + rpb - 1) / rpb;
    if (blocks > 1024) blocks = 1024;                       // four resident blocks per CU, each leaves its column sums once
    if (a.ez) hipLaunchKernelGGL((thin_dgrad_kernel<true, 4>), dim3((unsigned)blocks), dim3(256), 0, st, a, C4, cpt, rpb);
    else hipLaunchKernelGGL((thin_dgrad_kernel<false, 4>), dim3((unsigned)blocks), dim3(256), 0, st, a, C4, cpt, rpb);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Forward of heads with K <= 4 outputs in all (colour 3 + density 1 share their input): Out[m, k] = alpha * (In[m, :] . W[k, :] + bias[k]).
// One stream over In instead of one row GEMM per head (58 us each at 4096 x 96 on a 32-column MFMA tile of which 1 or 3 columns are real): a thread owns four
// input columns, its K x 4 weights live in registers, the lanes of a row add up their partial dot products with width-limited shuffles (exact fp32 FMAs).
template <bool ACT>
__global__ __launch_bounds__(256) void thin_fwd_kernel(const ThinFwdArgs A, int C4, int cpt) {
    const int tc = threadIdx.x % cpt, tr = threadIdx.x / cpt, rows_pass = 256 / cpt;
    const bool live = tc < C4;
    float w[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            w[k][q] = (live && k < A.K) ? ((k == 3 && A.W3) ? A.W3[tc * 4 + q] : A.W[(int64_t)k * A.ldw + tc * 4 + q]) : 0.f;
    float bias[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) bias[k] = k < A.K ? ((k == 3 && A.bias3) ? A.bias3[0] : A.bias[k]) : 0.f;
    float ta[4] = {0.f, 0.f, 0.f, 0.f}, tb[4] = {0.f, 0.f, 0.f, 0.f};
    bool act[4] = {false, false, false, false};
    if (ACT && live) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = tc * 4 + q;
            act[q] = c < A.tab_cols;
            if (act[q]) { ta[q] = A.tab[c]; tb[q] = A.tab[A.tab_stride + c]; }
        }
    }
    constexpr int U = 4;
    const int64_t stride = (int64_t)gridDim.x * rows_pass * U;
    for (int64_t rb = (int64_t)blockIdx.x * rows_pass * U + tr; rb < A.M; rb += stride) {      // (uniform per row group: every lane of a row takes the same trips)
        f32x4_t x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = rb + (int64_t)u * rows_pass;
            const int64_t rc = r < A.M ? r : A.M - 1;
            x[u] = live ? *(const f32x4_t*)(A.In + rc * A.ldi + tc * 4) : f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (ACT) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (act[q]) x[u][q] = __builtin_amdgcn_sinf(__builtin_fmaf(ta[q], x[u][q], tb[q]));
            }
            float p[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                p[k] = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) p[k] = __builtin_fmaf(w[k][q], x[u][q], p[k]);
            }
            const int64_t r = rb + (int64_t)u * rows_pass;
            if (cpt >= 4) {
                // butterfly with a transpose: the first step leaves each lane two of the four sums, the second one - 6 shuffles for the row instead of 20
                const int h1 = cpt >> 1, h2 = cpt >> 2;
                const bool up1 = (tc & h1) != 0, up2 = (tc & h2) != 0;
                const float s01 = up1 ? p[0] : p[2], s23 = up1 ? p[1] : p[3];            // what this lane hands over: the pair it does not keep
                float a = (up1 ? p[2] : p[0]) + __shfl_xor(s01, h1, 64);                  // lanes with bit h1 clear keep sums 0 and 1, the others 2 and 3
                float b = (up1 ? p[3] : p[1]) + __shfl_xor(s23, h1, 64);
                float v = (up2 ? b : a) + __shfl_xor(up2 ? a : b, h2, 64);                 // ... and of its pair: bit h2 clear keeps the first, set the second
                for (int o = h2 >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                const int k = (up1 ? 2 : 0) + (up2 ? 1 : 0);                               // which sum ended up here
                if ((tc & (h2 - 1)) == 0 && r < A.M && k < A.K) A.Out[r * A.ldo + k] = A.alpha * (v + (k == 0 ? bias[0] : k == 1 ? bias[1] : k == 2 ? bias[2] : bias[3]));
            } else {
                for (int o = cpt >> 1; o > 0; o >>= 1) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) p[k] += __shfl_xor(p[k], o, 64);
                }
                if (tc == 0 && r < A.M) {
                    for (int k = 0; k < A.K; ++k) A.Out[r * A.ldo + k] = A.alpha * (p[k] + bias[k]);
                }
            }
        }
    }
}
bool thin_fwd_ok(const ThinFwdArgs& a) {
    return a.K >= 1 && a.K <= 4 && a.N % 4 == 0 && a.N >= 4 && a.N <= 256 && a.ldi % 4 == 0 && (uintptr_t)a.In % 16 == 0 && a.ldo >= a.K &&
           (!a.tab || (a.tab_cols >= 0 && a.tab_stride >= a.tab_cols));
}
hipError_t launch_thin_fwd(const ThinFwdArgs& a, hipStream_t st) {
    if (a.M <= 0) return hipSuccess;
    if (!thin_fwd_ok(a)) return hipErrorInvalidValue;
    const int C4 = a.N / 4;
    int cpt = 1;
    while (cpt < C4) cpt <<= 1;                             // <= 64: the lanes of a row sit in one wave (the shuffles never cross a row: cpt divides 64)
    const int rows = 4 * (256 / cpt);
    int64_t blocks = (a.M + rows - 1) / rows;
    if (blocks > 2048) blocks = 2048;
    if (a.tab && a.tab_cols > 0) hipLaunchKernelGGL((thin_fwd_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, st, a, C4, cpt);
    else hipLaunchKernelGGL((thin_fwd_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, st, a, C4, cpt);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradient of a head with K <= 4 outputs (colour 3, density 1, solar visibility 1, sky colour 3):
//   dW[k, n] += alpha * sum_m D[m, k] In[m, n]
// K x N is at most 4 x 1024 numbers: the MFMA kernel spends a 128 x 64 block of accumulators per workgroup on it, gathers D with 4 live lanes of 64 and
// reached 53 % of the copy rate over In at 4096 x 96 (74 us per launch, 7 launches per step).  Like the input gradient of these heads (thin_dgrad_kernel)
// it is a stream, not a GEMM: a thread owns four input columns and a row phase, keeps its K x 4 sums in registers (exact fp32 FMAs), U rows in flight;
// the row phases of a block meet in LDS, the block leaves its K x N sums in the partial-sum scratch and a second small kernel adds the blocks up in a fixed
// order (as the two-stage reduction of the MFMA weight-gradient kernel does: no atomics, the same bits every step).
template <bool ACT>
__global__ __launch_bounds__(256) void thin_wgrad_kernel(const ThinWgradArgs A, int C4, int cpt, int rows_per_block) {
    __shared__ float red[256][16];
    const int tc = threadIdx.x % cpt, tr = threadIdx.x / cpt, rows_pass = 256 / cpt;
    const bool live = tc < C4;
    float ta[4] = {0.f, 0.f, 0.f, 0.f}, tb[4] = {0.f, 0.f, 0.f, 0.f};
    bool act[4] = {false, false, false, false};
    if (ACT && live) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = tc * 4 + q;
            act[q] = c < A.tab_cols;
            if (act[q]) { ta[q] = A.tab[c]; tb[q] = A.tab[A.tab_stride + c]; }
        }
    }
    float acc[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[k][q] = 0.f;
    constexpr int U = 8;
    const bool vec_d = (A.ldd & 3) == 0 && ((uintptr_t)A.D & 15) == 0;      // the K values of a row as one 16-byte load (columns past K: dropped below)
    const int64_t n_chunks = (A.M + rows_per_block - 1) / rows_per_block;
    if (live) {
        for (int64_t ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
            const int64_t r0 = ch * rows_per_block;
            const int64_t r1 = r0 + rows_per_block < A.M ? r0 + rows_per_block : A.M;
            for (int64_t rb = r0 + tr; rb < r1; rb += (int64_t)U * rows_pass) {
                float d[U][4];
                f32x4_t x[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int64_t r = rb + (int64_t)u * rows_pass;
                    const bool ok = r < r1;
                    const int64_t rc = ok ? r : r1 - 1;
                    if (vec_d) {
                        const f32x4_t t4 = *(const f32x4_t*)(A.D + rc * A.ldd);
#pragma unroll
                        for (int k = 0; k < 4; ++k) d[u][k] = (ok && k < A.K) ? t4[k] : 0.f;
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) d[u][k] = (ok && k < A.K) ? A.D[rc * A.ldd + k] : 0.f;
                    }
                    x[u] = *(const f32x4_t*)(A.In + rc * A.ldi + tc * 4);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (ACT) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (act[q]) x[u][q] = __builtin_amdgcn_sinf(__builtin_fmaf(ta[q], x[u][q], tb[q]));
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[k][q] = __builtin_fmaf(d[u][k], x[u][q], acc[k][q]);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) red[threadIdx.x][k * 4 + q] = acc[k][q];
    __syncthreads();
    if (tr == 0 && live) {
        float* out = A.partial + (int64_t)blockIdx.x * A.K * A.N;
        for (int k = 0; k < A.K; ++k) {
            f32x4_t v;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float s = 0.f;
                for (int p = 0; p < rows_pass; ++p) s += red[p * cpt + tc][k * 4 + q];
                v[q] = s;
            }
            *(f32x4_t*)(out + k * A.N + tc * 4) = v;
        }
    }
}
__global__ __launch_bounds__(256) void thin_wgrad_reduce_kernel(const float* __restrict__ partial, int blocks, int K, int N, float* dW, int64_t ldw, float alpha, float* dW3) {
    // 32 elements of dW per workgroup, 8 threads per element: thread (e, seg) adds the blocks seg, seg + 8, ... (8 loads in flight), LDS adds the 8 segments
    __shared__ float red[8][32];
    const int el = threadIdx.x & 31, seg = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + el;
    const int64_t KN = (int64_t)K * N;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (e < KN) {
        int b = seg;
        for (; b + 56 < blocks; b += 64) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += partial[(int64_t)(b + 8 * u) * KN + e];
        }
        for (; b < blocks; b += 8) s[0] += partial[(int64_t)b * KN + e];
    }
    red[seg][el] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (seg == 0 && e < KN) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += red[q][el];
        float* out = (dW3 && e / N == 3) ? dW3 + e % N : dW + (int64_t)(e / N) * ldw + e % N;
        *out += alpha * t;
    }
}
bool thin_wgrad_ok(const ThinWgradArgs& a) {
    return a.K >= 1 && a.K <= 4 && a.N % 4 == 0 && a.N >= 4 && a.N <= 1024 && a.ldi % 4 == 0 && (uintptr_t)a.In % 16 == 0 &&
           (!a.tab || (a.tab_cols >= 0 && a.tab_stride >= a.tab_cols));
}
hipError_t launch_thin_wgrad(const ThinWgradArgs& a, hipStream_t st) {
    if (a.M <= 0 || a.N <= 0) return hipSuccess;
    if (!thin_wgrad_ok(a)) return hipErrorInvalidValue;
    const int C4 = a.N / 4;
    int cpt = 1;
    while (cpt < C4) cpt <<= 1;
    const int rpb = 8 * 2 * (256 / cpt);                    // two rounds of U = 8 rows per row phase and chunk
    int64_t blocks = (a.M + rpb - 1) / rpb;
    if (blocks > 512) blocks = 512;                         // two resident blocks per CU: 512 partial rows for the second stage
    ThinWgradArgs b = a;
    b.partial = gemm_partial_scratch(st, (size_t)blocks * a.K * a.N);
    if (!b.partial) return hipErrorOutOfMemory;
    if (a.tab && a.tab_cols > 0) hipLaunchKernelGGL((thin_wgrad_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, st, b, C4, cpt, rpb);
    else hipLaunchKernelGGL((thin_wgrad_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, st, b, C4, cpt, rpb);
    hipLaunchKernelGGL(thin_wgrad_reduce_kernel, dim3((unsigned)((a.K * a.N + 31) / 32)), dim3(256), 0, st, b.partial, (int)blocks, a.K, a.N, a.dW, a.ldw, a.alpha, a.dW3);
    return hipGetLastError();
}

hipError_t launch_colreduce(const ColArgs& a, hipStream_t st) {
    if (a.M <= 0) return hipSuccess;
    if (colpass_vec_ok(a) && a.mode == 1) return launch_colpass_vec<1>(a, nullptr, nullptr, st);
    if (colpass_vec_ok(a) && a.mode == 2) return launch_colpass_vec<2>(a, nullptr, nullptr, st);
    const int64_t blocks = (a.M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    hipLaunchKernelGGL(colreduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
    return hipGetLastError();
}

__global__ void bn_finalize_kernel(const float* colsum, const float* m2, int64_t M, int C, float* mean, float* istd,
                                   float* running_mean, float* running_var, int stage) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (stage == 0) {
        mean[c] = colsum[c] / (float)M;
    } else if (stage == 2) {                                        // eval mode: m2 points at the running variance
        istd[c] = 1.f / sqrtf(m2[c] + 1e-5f);
    } else {
        const float var_b = m2[c] / (float)M;                       // biased: normalisation (torch BatchNorm1d)
        istd[c] = 1.f / sqrtf(var_b + 1e-5f);
        const float var_u = M > 1 ? m2[c] / (float)(M - 1) : var_b; // unbiased: running estimate
        running_mean[c] = 0.99f * running_mean[c] + 0.01f * mean[c];
        running_var[c] = 0.99f * running_var[c] + 0.01f * var_u;
    }
}
hipError_t launch_bn_finalize(const float* colsum, const float* m2, int64_t M, int C, float* mean, float* istd,
                              float* running_mean, float* running_var, int stage, hipStream_t st) {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, colsum, m2, M, C, mean, istd,
                       running_mean, running_var, stage);
    return hipGetLastError();
}

// train-mode statistics from the shifted sums of the GEMM epilogue: S1 = sum(z - s), S2 = sum((z - s)^2), s = alpha*bias
// Also (one launch instead of three): writes the layer's activation-on-load table [a | b] (act_table_kernel's arithmetic on the
// rounded fp32 mean / istd just stored) when tab != nullptr, and clears the sums it consumed, so that the next producer finds
// zeros without a memset in between.
__global__ void bn_finalize_shifted_kernel(double* stats, const float* bias, float alpha, int64_t M, int C, float* mean, float* istd,
                                           float* running_mean, float* running_var, const float* gamma, const float* beta, float* tab) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double m1 = stats[c] / (double)M, m2 = stats[C + c] / (double)M;
    stats[c] = 0.0;
    stats[C + c] = 0.0;
    const double mu = (double)(alpha * bias[c]) + m1;
    double var_b = m2 - m1 * m1;                                    // biased: normalisation (torch BatchNorm1d)
    if (var_b < 0.0) var_b = 0.0;
    const float mean_f = (float)mu, istd_f = 1.f / sqrtf((float)var_b + 1e-5f);
    mean[c] = mean_f;
    istd[c] = istd_f;
    const double var_u = M > 1 ? var_b * (double)M / (double)(M - 1) : var_b;    // unbiased: running estimate
    running_mean[c] = 0.99f * running_mean[c] + 0.01f * (float)mu;
    running_var[c] = 0.99f * running_var[c] + 0.01f * (float)var_u;
    if (tab) {
        const double inv2pi = 0.15915494309189535;
        double a = (double)gamma[c] * (double)istd_f;
        const double b = ((double)beta[c] - a * (double)mean_f) * inv2pi;
        a *= inv2pi;
        tab[c] = (float)a;
        tab[C + c] = (float)b;
    }
}
hipError_t launch_bn_finalize_shifted(double* stats, const float* bias, float alpha, int64_t M, int C, float* mean, float* istd,
                                      float* running_mean, float* running_var, const float* gamma, const float* beta, float* tab,
                                      hipStream_t st) {
    hipLaunchKernelGGL(bn_finalize_shifted_kernel, dim3((C + 255) / 256), dim3(256), 0, st, stats, bias, alpha, M, C, mean, istd,
                       running_mean, running_var, gamma, beta, tab);
    return hipGetLastError();
}

// sums of a fused activation-backward epilogue (double [2][C]) -> fp32 vectors out0 = scale0*S0, out1 = S1 (optional) and
// accumulated into parameter gradients acc0 += scale0*S0, acc1 += S1 (optional)
__global__ void act_sums_finalize_kernel(double* stats, int C, float scale0, float* out0, float* out1, float* acc0, float* acc1) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float s0 = scale0 * (float)stats[c], s1 = (float)stats[C + c];
    stats[c] = 0.0;                                  // consumed: the next producer finds zeros
    stats[C + c] = 0.0;
    if (out0) out0[c] = s0;
    if (out1) out1[c] = s1;
    if (acc0) acc0[c] += s0;
    if (acc1) acc1[c] += s1;
}
hipError_t launch_act_sums_finalize(double* stats, int C, float scale0, float* out0, float* out1, float* acc0, float* acc1, hipStream_t st) {
    hipLaunchKernelGGL(act_sums_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, stats, C, scale0, out0, out1, acc0, acc1);
    return hipGetLastError();
}

__global__ void act_table_kernel(const float* mu, const float* istd, const float* gamma, const float* beta, int n, float* dst) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const double inv2pi = 0.15915494309189535;
    double a = inv2pi, b = 0.0;
    if (mu) {
        a = (double)gamma[c] * (double)istd[c];
        b = ((double)beta[c] - a * (double)mu[c]) * inv2pi;
        a *= inv2pi;
    }
    dst[c] = (float)a;
    dst[n + c] = (float)b;
}
hipError_t launch_act_table(const float* mu, const float* istd, const float* gamma, const float* beta, int n, float* dst, hipStream_t st) {
    hipLaunchKernelGGL(act_table_kernel, dim3((n + 255) / 256), dim3(256), 0, st, mu, istd, gamma, beta, n, dst);
    return hipGetLastError();
}

__global__ void sin_fwd_kernel(const float* Z, float* H, int64_t M, int C, int64_t ldz, int64_t ldh, const float* mu,
                               const float* istd, const float* gamma, const float* beta) {
    const int64_t total = M * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / C;
        const int c = (int)(i - r * C);
        float z = Z[r * ldz + c];
        if (mu) z = gamma[c] * ((z - mu[c]) * istd[c]) + beta[c];
        H[r * ldh + c] = sinf(z);
    }
}
// 16-byte variant (C, ldz, ldh multiples of 4, aligned bases): a thread keeps the BatchNorm constants of its 4 columns in
// registers and walks down the rows - no index division, full-width loads and stores
__global__ __launch_bounds__(256) void sin_fwd_vec_kernel(const float* Z, float* H, int64_t M, int C4, int cpt, int64_t ldz, int64_t ldh,
                                                          const float* mu, const float* istd, const float* gamma, const float* beta) {
    const int tc = threadIdx.x % cpt, tr = threadIdx.x / cpt, rows_pass = 256 / cpt;
    if (tc >= C4) return;
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, m_[4] = {0.f, 0.f, 0.f, 0.f}, is_[4] = {1.f, 1.f, 1.f, 1.f}, be[4] = {0.f, 0.f, 0.f, 0.f};
    if (mu) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { sc[q] = gamma[tc * 4 + q]; m_[q] = mu[tc * 4 + q]; is_[q] = istd[tc * 4 + q]; be[q] = beta[tc * 4 + q]; }
    }
    for (int64_t r = (int64_t)blockIdx.x * rows_pass + tr; r < M; r += (int64_t)gridDim.x * rows_pass) {
        const f32x4_t z = *(const f32x4_t*)(Z + r * ldz + tc * 4);
        f32x4_t o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float y = z[q];
            if (mu) y = sc[q] * ((y - m_[q]) * is_[q]) + be[q];
            o[q] = sinf(y);
        }
        *(f32x4_t*)(H + r * ldh + tc * 4) = o;
    }
}
hipError_t launch_sin_fwd(const float* Z, float* H, int64_t M, int C, int64_t ldz, int64_t ldh, const float* mu, const float* istd,
                          const float* gamma, const float* beta, hipStream_t st) {
    if (M <= 0 || C <= 0) return hipSuccess;
    if (C % 4 == 0 && C <= 1024 && ldz % 4 == 0 && ldh % 4 == 0 && (uintptr_t)Z % 16 == 0 && (uintptr_t)H % 16 == 0) {
        const int C4 = C / 4;
        int cpt = 1;
        while (cpt < C4) cpt <<= 1;                       // threads per row, power of two <= 256
        const int rows_pass = 256 / cpt;
        int64_t blocks = (M + rows_pass - 1) / rows_pass;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(sin_fwd_vec_kernel, dim3((unsigned)blocks), dim3(256), 0, st, Z, H, M, C4, cpt, ldz, ldh, mu, istd, gamma, beta);
        return hipGetLastError();
    }
    LAUNCH_1D(sin_fwd_kernel, M * C, st, Z, H, M, C, ldz, ldh, mu, istd, gamma, beta);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void bn_bwd2_kernel(const float* Z, float* D, int64_t M, int C, int64_t ld, int64_t ldd, const float* mu,
                                                      const float* istd, const float* gamma, const float* beta, const float* sdy,
                                                      const float* sdyx, float* dbias_sum, float alpha, float invM, int d_is_dy) {
    __shared__ float red0[256];
    const int cp = C >= 256 ? 256 : (C >= 128 ? 128 : (C >= 64 ? 64 : 32));
    const int phases = 256 / cp;
    const int tc = threadIdx.x % cp, tp = threadIdx.x / cp;
    const int64_t r0 = (int64_t)blockIdx.x * ROWS_PER_BLOCK;
    const int64_t r1 = r0 + ROWS_PER_BLOCK < M ? r0 + ROWS_PER_BLOCK : M;
    for (int c0 = 0; c0 < C; c0 += cp) {
        const int c = c0 + tc;
        float s0 = 0.f;
        if (c < C) {
            const float m = mu[c], is = istd[c], gm = gamma[c], bt = beta[c], k = gm * is, a = sdy[c] * invM, b = sdyx[c] * invM;
            for (int64_t r = r0 + tp; r < r1; r += phases) {
                const float xh = (Z[r * ld + c] - m) * is;
                const float dy = d_is_dy ? D[r * ldd + c] : D[r * ldd + c] * cosf(gm * xh + bt);     // D holds dL/dH (or dL/dY) on entry
                const float dz = k * (dy - a - xh * b);
                D[r * ldd + c] = dz;
                s0 += dz;
            }
        }
        red0[threadIdx.x] = s0;
        __syncthreads();
        if (tp == 0 && c < C && dbias_sum) {
            for (int p = 1; p < phases; ++p) s0 += red0[p * cp + tc];
            atomicAdd(dbias_sum + c, alpha * s0);
        }
        __syncthreads();
    }
}
hipError_t launch_bn_bwd2(const float* Z, float* D, int64_t M, int C, int64_t ld, int64_t ldd, const float* mu, const float* istd,
                          const float* gamma, const float* beta, const float* sdy, const float* sdyx, float* dbias_sum, float alpha,
                          int64_t M_global, hipStream_t st, bool d_is_dy) {
    if (M <= 0) return hipSuccess;
    ColArgs a{};
    a.mode = 3; a.M = M; a.M_global = M_global; a.C = C; a.ld = ld; a.ldd = ldd; a.Z = Z; a.D = D; a.mu = mu; a.istd = istd; a.gamma = gamma; a.beta = beta;
    a.out0 = dbias_sum; a.alpha0 = alpha;
    if (colpass_vec_ok(a)) return d_is_dy ? launch_colpass_vec<4>(a, sdy, sdyx, st) : launch_colpass_vec<3>(a, sdy, sdyx, st);
    const int64_t blocks = (M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    hipLaunchKernelGGL(bn_bwd2_kernel, dim3((unsigned)blocks), dim3(256), 0, st, Z, D, M, C, ld, ldd ? ldd : ld, mu, istd, gamma, beta, sdy, sdyx, dbias_sum, alpha,
                       1.f / (float)M_global, d_is_dy ? 1 : 0);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
__global__ void point_out_fwd_kernel(const PointOutArgs A) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < A.n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t g = i / A.n_samples;
        if (A.rho) A.rho[i] = softplus_t(A.head[i * 4 + 3]);
        if (A.sv && A.sv_raw) A.sv[i] = sigmoid_t(A.sv_raw[i]);
        if (A.col) {
            float ac[3] = {0.f, 0.f, 0.f};
            for (int c = 0; c < A.C; ++c) {
                const float p = A.cls[g * A.C + c];
#pragma unroll
                for (int k = 0; k < 3; ++k) ac[k] = __fadd_rn(ac[k], __fmul_rn(A.adj[i * 3 * A.C + 3 * c + k], p));
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) A.col[i * 3 + k] = sigmoid_t(A.head[i * 4 + k] + ac[k]);
            if (A.adjust_col) { A.adjust_col[i * 3] = ac[0]; A.adjust_col[i * 3 + 1] = ac[1]; A.adjust_col[i * 3 + 2] = ac[2]; }
        }
    }
}
__global__ __launch_bounds__(256) void point_out_bwd_kernel(const PointOutArgs A) {
    // d_cls[ray, c] collects one term per sample: the samples of a ray sit next to each other, so a block first adds them up in LDS (the
    // 256 points of one pass cover at most 256 / n_samples + 2 rays) and sends one atomic per ray and class to memory - 96 samples of a
    // ray hammering one address cost 160 us at 4096 x 96
    constexpr int CMAX = 8;
    __shared__ float bins[258 * CMAX];
    const bool binned = A.d_cls && A.d_head && A.C <= CMAX;
    for (int64_t i0 = blockIdx.x * (int64_t)blockDim.x; i0 < A.n; i0 += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = i0 + threadIdx.x;
        const int64_t g0 = i0 / A.n_samples;
        const int64_t i_last = i0 + 255 < A.n ? i0 + 255 : A.n - 1;
        const int n_bins = (int)(i_last / A.n_samples - g0 + 1) * A.C;
        if (binned) {
            for (int b = threadIdx.x; b < n_bins; b += 256) bins[b] = 0.f;
            __syncthreads();
        }
        if (i < A.n) {
        const int64_t g = i / A.n_samples;
        if (A.d_head) {
            const float raw = A.head[i * 4 + 3];
            A.d_head[i * 4 + 3] = A.d_rho ? A.d_rho[i] * (raw > 20.f ? 1.f : sigmoid_t(raw)) : 0.f;
            float dpre[3] = {0.f, 0.f, 0.f};
            if (A.d_col) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { const float y = A.col[i * 3 + k]; dpre[k] = A.d_col[i * 3 + k] * y * (1.f - y); }
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) A.d_head[i * 4 + k] = dpre[k];
            for (int c = 0; c < A.C; ++c) {
                const float p = A.cls[g * A.C + c];
                float dc = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    if (A.d_adj) A.d_adj[i * 3 * A.C + 3 * c + k] = dpre[k] * p;
                    dc += dpre[k] * A.adj[i * 3 * A.C + 3 * c + k];
                }
                if (binned) atomicAdd(bins + (g - g0) * A.C + c, dc);
                else if (A.d_cls) atomicAdd(A.d_cls + g * A.C + c, dc);
            }
        }
        if (A.d_sv_raw && A.d_sv) { const float y = A.sv[i]; A.d_sv_raw[i] = A.d_sv[i] * y * (1.f - y); }
        }
        if (binned) {
            __syncthreads();
            for (int b = threadIdx.x; b < n_bins; b += 256) atomicAdd(A.d_cls + g0 * A.C + b, bins[b]);
            __syncthreads();
        }
    }
}
hipError_t launch_point_out(const PointOutArgs& a, bool backward, hipStream_t st) {
    if (backward) LAUNCH_1D(point_out_bwd_kernel, a.n, st, a);
    else LAUNCH_1D(point_out_fwd_kernel, a.n, st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// compositing backward, one wavefront per ray (forward: Eval_Tools_2.py:187-215, default solar model):
//   y = rho*delta, PV_s = exp(-sum_{j<s} y_j), PE_s = 1 - exp(-y_s), PS = PV*PE
//   Albedo = sum PS*col ; u = sum PS*sv (sv detached) ; SV3 = sigmoid(30(u-.2)) ; F = SV3 + (1-SV3)*sky ; RGB = Albedo*F
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wscan_incl(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}
__global__ __launch_bounds__(256) void composite_bwd_kernel(const CompBwdArgs A) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= A.n_rays) return;
    const int S = A.n_samples;
    const float dx = A.top[r * 3] - A.bot[r * 3], dy = A.top[r * 3 + 1] - A.bot[r * 3 + 1], dz = A.top[r * 3 + 2] - A.bot[r * 3 + 2];
    const float delta = __fdiv_rn(__fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz))), (float)S);
    const float sky[3] = {A.sky[r * 3], A.sky[r * 3 + 1], A.sky[r * 3 + 2]};
    const bool prior = A.rho_prior != nullptr;
    const float tr = A.trust_dev ? A.trust_dev[0] : A.trust;
    // ---- pass 1: forward sums (albedo, merged albedo, u)
    float alb[3] = {0.f, 0.f, 0.f}, albm[3] = {0.f, 0.f, 0.f}, u = 0.f, carry = 0.f, carry_m = 0.f;
    for (int base = 0; base < S; base += 64) {
        const int s = base + lane;
        const bool in = s < S;
        const int64_t idx = r * S + (in ? s : S - 1);
        const float rho = A.rho[idx];
        const float y = in ? rho * delta : 0.f;
        const float incl = wscan_incl(y, lane);
        const float pv = expf(-(carry + incl - y));
        carry += __shfl(incl, 63, 64);
        const float ps = in ? pv * (1.f - expf(-y)) : 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) alb[k] += ps * A.col[idx * 3 + k];
        u += ps * A.sv[idx];
        if (prior) {
            const float ym = in ? (rho * tr + A.rho_prior[idx] * (1.f - tr)) * delta : 0.f;
            const float inclm = wscan_incl(ym, lane);
            const float pvm = expf(-(carry_m + inclm - ym));
            carry_m += __shfl(inclm, 63, 64);
            const float psm = in ? pvm * (1.f - expf(-ym)) : 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) albm[k] += psm * A.col[idx * 3 + k];
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { alb[k] = wsum(alb[k]); albm[k] = wsum(albm[k]); }
    u = wsum(u);
    const bool classic = A.classic != 0;
    const float sv3 = sigmoid_t((u - 0.2f) * 30.f);
    float dalb[3], dalbm[3], dsky[3], dsv3 = 0.f, grgb[3], grgbm[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float F = classic ? 0.f : sv3 + (1.f - sv3) * sky[k];       // classic: the shading factor is per sample (pass 2)
        const float g = A.g_rgb ? A.g_rgb[r * 3 + k] : 0.f;
        const float gm = (prior && A.g_rgb_m) ? A.g_rgb_m[r * 3 + k] : 0.f;
        grgb[k] = g;
        grgbm[k] = classic ? gm : 0.f;                 // classic: Rendered_Col_Merged is shaded per sample too
        dalb[k] = g * F + (A.g_albedo ? A.g_albedo[r * 3 + k] : 0.f);
        dalbm[k] = gm * F + ((prior && A.g_albedo_m) ? A.g_albedo_m[r * 3 + k] : 0.f);
        const float dF = g * alb[k] + gm * albm[k];
        dsv3 += dF * (1.f - sky[k]);
        dsky[k] = classic ? 0.f : dF * (1.f - sv3);
    }
    const float du = classic ? 0.f : dsv3 * sv3 * (1.f - sv3) * 30.f;
    if (!classic && lane == 0) { A.d_sky[r * 3] = dsky[0]; A.d_sky[r * 3 + 1] = dsky[1]; A.d_sky[r * 3 + 2] = dsky[2]; }
    // ---- pass 2: dPS, then dy_s = dPE_s*exp(-y_s) - sum_{k>s} dPV_k*PV_k  (suffix sums, chunks walked backwards)
    float suffix = 0.f, suffix_m = 0.f;
    const int nchunk = (S + 63) / 64;
    for (int ch = nchunk - 1; ch >= 0; --ch) {
        float pre = 0.f, pre_m = 0.f;          // prefix of y (and merged y) before this chunk
        for (int b2 = 0; b2 < ch; ++b2) {
            const int s2 = b2 * 64 + lane;
            if (s2 < S) {
                const float rr = A.rho[r * S + s2];
                pre += rr * delta;
                if (prior) pre_m += (rr * tr + A.rho_prior[r * S + s2] * (1.f - tr)) * delta;
            }
        }
        pre = wsum(pre);
        pre_m = wsum(pre_m);
        const int s = ch * 64 + lane;
        const bool in = s < S;
        const int64_t idx = r * S + (in ? s : S - 1);
        const float rho = A.rho[idx];
        const float y = in ? rho * delta : 0.f;
        const float incl = wscan_incl(y, lane);
        const float pv = expf(-(pre + incl - y));
        const float ey = expf(-y);
        const float pe = 1.f - ey;
        float dps = 0.f;
        const float svs = A.sv[idx];
        float shade[3] = {0.f, 0.f, 0.f};
        if (in) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                shade[k] = classic ? svs + (1.f - svs) * sky[k] : 0.f;
                dps += (dalb[k] + grgb[k] * shade[k]) * A.col[idx * 3 + k];
            }
            dps += du * svs;
        }
        const float dpv_pv = in ? dps * pe * pv : 0.f;          // dPV_s * PV_s
        float dpe = in ? dps * pv : 0.f;
        if (in && A.g_pe) dpe += A.g_pe[idx];
        const float incl2 = wscan_incl(dpv_pv, lane);
        const float tot = __shfl(incl2, 63, 64);
        const float later = suffix + (tot - incl2);
        float d_rho = (dpe * ey - later) * delta;
        float dc[3];
        const float ps = pv * pe;
#pragma unroll
        for (int k = 0; k < 3; ++k) dc[k] = (dalb[k] + grgb[k] * shade[k]) * ps;
        float dsv = 0.f;
        if (classic) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float t_ = in ? grgb[k] * ps * A.col[idx * 3 + k] : 0.f;
                dsv += t_ * (1.f - sky[k]);
                dsky[k] += t_ * (1.f - svs);
            }
        }
        suffix += tot;
        if (prior) {
            const float ym = in ? (rho * tr + A.rho_prior[idx] * (1.f - tr)) * delta : 0.f;
            const float inclm = wscan_incl(ym, lane);
            const float pvm = expf(-(pre_m + inclm - ym));
            const float eym = expf(-ym);
            const float pem = 1.f - eym;
            float dpsm = 0.f;
            if (in) {
#pragma unroll
                for (int k = 0; k < 3; ++k) dpsm += (dalbm[k] + grgbm[k] * shade[k]) * A.col[idx * 3 + k];
            }
            const float dpvm_pvm = in ? dpsm * pem * pvm : 0.f;
            const float incl3 = wscan_incl(dpvm_pvm, lane);
            const float totm = __shfl(incl3, 63, 64);
            const float later_m = suffix_m + (totm - incl3);
            d_rho += tr * (dpsm * pvm * eym - later_m) * delta;
            const float psm = pvm * pem;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                dc[k] += (dalbm[k] + grgbm[k] * shade[k]) * psm;
                if (classic) {
                    const float t_ = in ? grgbm[k] * psm * A.col[idx * 3 + k] : 0.f;
                    dsv += t_ * (1.f - sky[k]);
                    dsky[k] += t_ * (1.f - svs);
                }
            }
            suffix_m += totm;
        }
        if (classic && in) A.d_sv[idx] = dsv;
        if (in) {
            A.d_rho[idx] = d_rho;
#pragma unroll
            for (int k = 0; k < 3; ++k) A.d_col[idx * 3 + k] = dc[k];
        }
    }
    if (classic) {
#pragma unroll
        for (int k = 0; k < 3; ++k) dsky[k] = wsum(dsky[k]);
        if (lane == 0) { A.d_sky[r * 3] = dsky[0]; A.d_sky[r * 3 + 1] = dsky[1]; A.d_sky[r * 3 + 2] = dsky[2]; }
    }
}
hipError_t launch_composite_bwd(const CompBwdArgs& a, hipStream_t st) {
    if (a.n_rays <= 0) return hipSuccess;
    hipLaunchKernelGGL(composite_bwd_kernel, dim3((unsigned)((a.n_rays + 3) / 4)), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
__global__ void softmax_kernel(const float* x, float* p, int64_t rows, int C) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < rows; i += (int64_t)gridDim.x * blockDim.x) {
        float m = -3.0e38f;
        for (int c = 0; c < C; ++c) m = fmaxf(m, x[i * C + c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) { const float e = expf(x[i * C + c] - m); p[i * C + c] = e; s += e; }
        for (int c = 0; c < C; ++c) p[i * C + c] /= s;
    }
}
__global__ void softmax_bwd_kernel(const float* p, const float* dp, float* dx, int64_t rows, int C) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < rows; i += (int64_t)gridDim.x * blockDim.x) {
        float dot = 0.f;
        for (int c = 0; c < C; ++c) dot += p[i * C + c] * dp[i * C + c];
        for (int c = 0; c < C; ++c) dx[i * C + c] = p[i * C + c] * (dp[i * C + c] - dot);
    }
}
__global__ void sigmoid_kernel(const float* x, float* y, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = sigmoid_t(x[i]);
}
__global__ void sigmoid_bwd_kernel(const float* y, const float* dy, float* dx, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dx[i] = dy[i] * y[i] * (1.f - y[i]);
}
hipError_t launch_softmax(const float* x, float* p, int64_t rows, int C, hipStream_t st) { LAUNCH_1D(softmax_kernel, rows, st, x, p, rows, C); return hipGetLastError(); }
hipError_t launch_softmax_bwd(const float* p, const float* dp, float* dx, int64_t rows, int C, hipStream_t st) { LAUNCH_1D(softmax_bwd_kernel, rows, st, p, dp, dx, rows, C); return hipGetLastError(); }
hipError_t launch_sigmoid(const float* x, float* y, int64_t n, hipStream_t st) { LAUNCH_1D(sigmoid_kernel, n, st, x, y, n); return hipGetLastError(); }
hipError_t launch_sigmoid_bwd(const float* y, const float* dy, float* dx, int64_t n, hipStream_t st) { LAUNCH_1D(sigmoid_bwd_kernel, n, st, y, dy, dx, n); return hipGetLastError(); }

__global__ __launch_bounds__(256) void colsum_kernel(const float* X, int64_t M, int C, int64_t ld, float alpha, float* out) {
    // small C (<= 16): every thread walks rows and keeps all C column sums in registers - one pass over X
    const int lane = threadIdx.x & 63;
    float s[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) s[c] = 0.f;
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < M; r += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c < C) s[c] += X[r * ld + c];
    }
    __shared__ float red[4][16];               // same-address atomics serialise: one per column and workgroup
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const float v = c < C ? wsum(s[c]) : 0.f;
        if (lane == 0) red[threadIdx.x >> 6][c] = v;
    }
    __syncthreads();
    if (threadIdx.x < C) atomicAdd(out + threadIdx.x, alpha * (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]));
}
hipError_t launch_colsum(const float* X, int64_t M, int C, int64_t ld, float alpha, float* out, hipStream_t st) {
    if (M <= 0) return hipSuccess;
    int64_t b = (M + 255) / 256;
    if (b > 512) b = 512;
    for (int c0 = 0; c0 < C; c0 += 16)          // 16 columns per pass (heads have 1..12 outputs at the default 4 classes)
        hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)b), dim3(256), 0, st, X + c0, M, C - c0 < 16 ? C - c0 : 16, ld, alpha, out + c0);
    return hipGetLastError();
}

__global__ void copy_cols_kernel(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int64_t M, int C, int acc) {
    const int64_t total = M * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / C;
        const int c = (int)(i - r * C);
        const float v = src[r * ld_src + c];
        if (acc) dst[r * ld_dst + c] += v; else dst[r * ld_dst + c] = v;
    }
}
// Experiment switch only (SNERF_TRAIN_DY_BF16=1, train.cpp sine_bwd): round an fp32 array to bf16 precision IN PLACE - what storing the inter-layer gradient
// as bf16 would do to its values, without the kernels that would read it as such (the byte saving is priced from the layer table, not measured here).
__global__ void round_bf16_kernel(float* p, int64_t ld, int64_t M, int C) {
    const int64_t total = M * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / C;
        float* q = p + r * ld + (i - r * C);
        *q = (float)(__bf16)*q;
    }
}
// The engine's own fill and copy (train.cpp snerf_zero_async / snerf_copy_async: no runtime memory operations inside a step): 16 bytes per lane where the
// addresses allow, a scalar tail.
__global__ void fill_zero_kernel(float* __restrict__ p, int64_t n) {
    const int64_t n4 = (((uintptr_t)p & 15) == 0) ? n / 4 : 0;
    const int64_t t0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = t0; i < n4; i += step) ((float4*)p)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = n4 * 4 + t0; i < n; i += step) p[i] = 0.f;
}
hipError_t launch_fill_zero(float* p, int64_t n, hipStream_t st) {
    LAUNCH_1D(fill_zero_kernel, (n + 3) / 4, st, p, n);
    return hipGetLastError();
}
__global__ void copy_f32_kernel(float* __restrict__ d, const float* __restrict__ s, int64_t n) {
    const int64_t n4 = (((((uintptr_t)d) | ((uintptr_t)s)) & 15) == 0) ? n / 4 : 0;
    const int64_t t0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = t0; i < n4; i += step) ((float4*)d)[i] = ((const float4*)s)[i];
    for (int64_t i = n4 * 4 + t0; i < n; i += step) d[i] = s[i];
}
hipError_t launch_copy_f32(float* d, const float* s, int64_t n, hipStream_t st) {
    LAUNCH_1D(copy_f32_kernel, (n + 3) / 4, st, d, s, n);
    return hipGetLastError();
}
__global__ void fill_zero_bytes_kernel(unsigned char* p, int64_t n) {      // (odd sizes / addresses: the engine has none today; no path falls back to the runtime's memset)
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0;
}
__global__ void copy_bytes_kernel(unsigned char* __restrict__ d, const unsigned char* __restrict__ s, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = s[i];
}
hipError_t launch_fill_zero_bytes(void* p, size_t bytes, hipStream_t st) {
    LAUNCH_1D(fill_zero_bytes_kernel, (int64_t)bytes, st, (unsigned char*)p, (int64_t)bytes);
    return hipGetLastError();
}
hipError_t launch_copy_bytes(void* d, const void* s, size_t bytes, hipStream_t st) {
    LAUNCH_1D(copy_bytes_kernel, (int64_t)bytes, st, (unsigned char*)d, (const unsigned char*)s, (int64_t)bytes);
    return hipGetLastError();
}
hipError_t launch_round_bf16(float* p, int64_t ld, int64_t M, int C, hipStream_t st) {
    if (M <= 0 || C <= 0) return hipSuccess;
    int64_t b = (M * C + 255) / 256;
    if (b > 8192) b = 8192;
    hipLaunchKernelGGL(round_bf16_kernel, dim3((unsigned)b), dim3(256), 0, st, p, ld, M, C);
    return hipGetLastError();
}
hipError_t launch_copy_cols(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int64_t M, int C, bool accumulate, hipStream_t st) {
    LAUNCH_1D(copy_cols_kernel, M * C, st, src, ld_src, dst, ld_dst, M, C, accumulate ? 1 : 0);
    return hipGetLastError();
}

__global__ void bcast_rows_kernel(const float* src, int C, float* dst, int64_t ld_dst, int col0, int64_t n, int S) {
    const int64_t total = n * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / C;
        const int c = (int)(i - r * C);
        dst[r * ld_dst + col0 + c] = src[(r / S) * C + c];
    }
}
hipError_t launch_bcast_rows(const float* src, int C, float* dst, int64_t ld_dst, int col0, int64_t n, int n_samples, hipStream_t st) {
    LAUNCH_1D(bcast_rows_kernel, n * C, st, src, C, dst, ld_dst, col0, n, n_samples);
    return hipGetLastError();
}

__global__ void reduce_rows_kernel(const float* src, int64_t ld_src, int col0, int C, float* dst, int64_t G, int S) {
    const int64_t total = G * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t g = i / C;
        const int c = (int)(i - g * C);
        float s = 0.f;
        for (int k = 0; k < S; ++k) s += src[(g * S + k) * ld_src + col0 + c];
        dst[g * C + c] = s;
    }
}
hipError_t launch_reduce_rows(const float* src, int64_t ld_src, int col0, int C, float* dst, int64_t n_groups, int n_samples, hipStream_t st) {
    LAUNCH_1D(reduce_rows_kernel, n_groups * C, st, src, ld_src, col0, C, dst, n_groups, n_samples);
    return hipGetLastError();
}

__global__ void adam_kernel(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                            float bc1, float bc2) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        // torch.optim.Adam: step_size = lr / bias_correction1; denom = sqrt(v)/sqrt(bias_correction2) + eps
        p[i] = p[i] - (lr / bc1) * (mi / (sqrtf(vi) / sqrtf(bc2) + eps));
    }
}
// the same update with its per-step scalars read from DEVICE memory - hyper = [lr, beta1, beta2, eps, 1 - beta1^step, 1 - beta2^step] - so that a
// captured launch (hipGraph: kernel arguments are frozen at capture time) follows the learning-rate schedule and the bias corrections
__global__ void adam_dev_kernel(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper) {
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], bc1 = hyper[4], bc2 = hyper[5];
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] - (lr / bc1) * (mi / (sqrtf(vi) / sqrtf(bc2) + eps));
    }
}
hipError_t launch_adam_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, hipStream_t st) {
    LAUNCH_1D(adam_dev_kernel, n, st, p, g, m, v, n, hyper);
    return hipGetLastError();
}
hipError_t launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, int step, hipStream_t st) {
    const float bc1 = 1.f - powf(b1, (float)step), bc2 = 1.f - powf(b2, (float)step);
    LAUNCH_1D(adam_kernel, n, st, p, g, m, v, n, lr, b1, b2, eps, bc1, bc2);
    return hipGetLastError();
}

// ---- the scalar loss terms of a training step in three launches (get_loss, Eval_Tools_2.py:340-420: MSE colour loss, solar rays on, default solar
// model, no DSM prior).  The reference forms them with ~45 small tensor ops (and as many again in autograd's backward): ~110 launches of 3-5 us
// each per step here, a tenth of a step in launch gaps alone.  scratch: 4 doubles (sums) + 3 64-bit keys per block of the partial kernel (albedo minima: float bits << 32 | row; at most 1024 blocks);
// self-cleaning - the finalize kernel leaves it in its initial state (sums 0, minima +inf), loss_scratch_init sets that state once.
// mins: 64-bit keys (albedo as float bits << 32 | row): the minimum AND the lowest row that attains it in one comparison - torch.min(albedo, 0)
// hands its gradient to ONE row (with ties - a saturated albedo - every tied row used to receive it)
constexpr unsigned long long kMinInit = 0x7f800000ffffffffull;
__global__ void loss_scratch_init_kernel(double* sums, unsigned long long* mins) {
    if (threadIdx.x < 4) sums[threadIdx.x] = 0.0;
    (void)mins;
}
__global__ __launch_bounds__(256) void loss_partial_kernel(const LossArgs A, double* sums, unsigned long long* mins) {
    float color = 0.f, sk = 0.f, sc = 0.f, ab = 0.f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    // rows, not elements: a thread sees all three channels of its rays, so the channel minima reduce in registers and the wave (one atomic per
    // wave and channel: thousands of atomics on three addresses cost 85 us when every element issued its own)
    unsigned long long mn[3] = {kMinInit, kMinInit, kMinInit};
    for (int64_t rr = t0; rr < A.R; rr += stride) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int64_t i = rr * 3 + c;
            const float d = A.rgb[i] - A.gt[i];
            color += d * d;
            const float x = (A.sky[i] - .5f) / .5f;
            if (x > 0.f) sk += x * x;
            // non-negative floats order as their bits; the row in the low half breaks ties towards the lowest row
            const unsigned long long key = ((unsigned long long)__float_as_uint(fmaxf(A.albedo[i], 0.f)) << 32) | (unsigned long long)(unsigned)rr;
            mn[c] = key < mn[c] ? key : mn[c];
        }
    }
    // per-block minima, no atomics (a 64-bit atomicMin under contention - 12 288 of them on three addresses - cost the step 6 ms: measured 20.2 against
    // 13.8 ms): wave reduction, the four waves through LDS, one plain store per block and channel; loss_finalize_kernel reduces the blocks
    __shared__ unsigned long long bmin[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        unsigned long long v = mn[c];
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long w = __shfl_xor(v, o);
            v = w < v ? w : v;
        }
        if ((threadIdx.x & 63) == 0) bmin[c][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        unsigned long long v = bmin[threadIdx.x][0];
        for (int k = 1; k < 4; ++k) v = bmin[threadIdx.x][k] < v ? bmin[threadIdx.x][k] : v;
        mins[(size_t)blockIdx.x * 3 + threadIdx.x] = v;
    }
    for (int64_t j = t0; j < A.Rs * A.S; j += stride) {
        const float v = A.sv[j], p = A.pv[j], d = v - p;
        sc += d * d;
        ab += A.pe[j] * p * v;
    }
    __shared__ float red[4][4];
    float vals[4] = {sc, ab, sk, color};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float v = vals[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4) atomicAdd(sums + threadIdx.x, (double)red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}
// minv [6]: the three minima the Albedo_Color term used, then (as int bits) the row that owns each one on THIS rank (-1: the global minimum lives elsewhere)
__global__ void loss_finalize_kernel(const LossArgs A, double* sums, unsigned long long* mins, int nblocks, float* vals, float* minv) {
    // one wave: the per-block minima of the partial kernel (3 keys per block) reduced by 64 lanes, then lane 0 writes the terms
    unsigned long long key[3] = {kMinInit, kMinInit, kMinInit};
    for (int b = threadIdx.x; b < nblocks; b += 64)
#pragma unroll
        for (int c = 0; c < 3; ++c) key[c] = mins[(size_t)b * 3 + c] < key[c] ? mins[(size_t)b * 3 + c] : key[c];
#pragma unroll
    for (int c = 0; c < 3; ++c)
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long w = __shfl_xor(key[c], o);
            key[c] = w < key[c] ? w : key[c];
        }
    if (threadIdx.x != 0) return;
    const double Rs = (double)A.Rs, R = (double)A.R;
    vals[0] = (float)(sums[0] / Rs);                          // Solar_Correction   = mean_r sum_s (Solar_Vis - PV_Exact)^2      (:361)
    vals[1] = (float)(1.0 - sums[1] / Rs);                    // Solar_Correction_2 = mean_r (1 - sum_s PE PV_Exact Solar_Vis)  (:366, detached)
    vals[2] = (float)(sums[2] / (3.0 * R));                   // Sky_Color_Var      = sum_{x > 0} x^2 / numel over [R, S, 3]: S copies of each ray's sky (:381-388)
    float h = 0.f;
    for (int c = 0; c < 3; ++c) {
        const float local = __uint_as_float((unsigned)(key[c] >> 32));
        const float a = A.alb_min_in ? A.alb_min_in[c] : local;
        minv[c] = a;
        minv[3 + c] = __int_as_float(local == a ? (int)(unsigned)(key[c] & 0xffffffffull) : -1);
        if (a < .2f) { const float u = 1.f - a / .2f; h += u * u; }
    }
    vals[3] = h / (float)(R * A.world);                       // Albedo_Color       = sum_c [a_c < .2] (1 - a_c / .2)^2 / R, a = min over the batch (:374-379)
    vals[4] = (float)(sums[3] / (3.0 * R));                   // Color              = MSE(Rendered_Col, GT_Color)                (:413)
    for (int k = 0; k < 4; ++k) sums[k] = 0.0;      // (the per-block minima are rewritten by every launch)
}
__global__ __launch_bounds__(256) void loss_bwd_kernel(const LossArgs A, const float* g, const float* minv, float* d_rgb, float* d_albedo, float* d_sky,
                                                       float* d_sv) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const float g_sc = g[0], g_sk = g[2], g_al = g[3], g_col = g[4];
    const float inv3r = 1.f / (3.f * (float)A.R), invr = 1.f / (float)A.R, invrs = 1.f / (float)A.Rs;
    for (int64_t i = t0; i < A.R * 3; i += stride) {
        d_rgb[i] = g_col * 2.f * (A.rgb[i] - A.gt[i]) * inv3r;
        const float x = (A.sky[i] - .5f) / .5f;
        d_sky[i] = x > 0.f ? g_sk * 4.f * x * inv3r : 0.f;                              // d x^2 / d sky = 2 x / .5
        const float a = minv[i % 3];
        const int own = __float_as_int(minv[3 + i % 3]);
        // the minimum's gradient goes to the ONE row that attains it - the lowest such row, as torch.min(albedo, 0) (on the rank that owns the global
        // minimum, divided by the LOCAL ray count: the rank average of the gradients is then the global-batch gradient, training.albedo_min_loss)
        d_albedo[i] = ((int)(i / 3) == own && a < .2f) ? g_al * 2.f * (1.f - a / .2f) * (-1.f / .2f) * invr : 0.f;
    }
    for (int64_t j = t0; j < A.Rs * A.S; j += stride) d_sv[j] = g_sc * 2.f * (A.sv[j] - A.pv[j]) * invrs;
}
hipError_t launch_loss_scratch_init(void* scratch, hipStream_t st) {
    hipLaunchKernelGGL(loss_scratch_init_kernel, dim3(1), dim3(64), 0, st, (double*)scratch, (unsigned long long*)((double*)scratch + 4));
    return hipGetLastError();
}
hipError_t launch_loss_terms(const LossArgs& a, void* scratch, float* vals, float* minv, hipStream_t st) {
    double* sums = (double*)scratch;
    unsigned long long* mins = (unsigned long long*)(sums + 4);
    int64_t n = a.Rs * a.S > a.R * 3 ? a.Rs * a.S : a.R * 3;
    int64_t b = (n + 255) / 256;
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    hipLaunchKernelGGL(loss_partial_kernel, dim3((unsigned)b), dim3(256), 0, st, a, sums, mins);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, st, a, sums, mins, (int)b, vals, minv);
    return hipGetLastError();
}
hipError_t launch_loss_terms_bwd(const LossArgs& a, const float* g, const float* minv, float* d_rgb, float* d_albedo, float* d_sky, float* d_sv, hipStream_t st) {
    int64_t n = a.Rs * a.S > a.R * 3 ? a.Rs * a.S : a.R * 3;
    int64_t b = (n + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    hipLaunchKernelGGL(loss_bwd_kernel, dim3((unsigned)b), dim3(256), 0, st, a, g, minv, d_rgb, d_albedo, d_sky, d_sv);
    return hipGetLastError();
}

}  // namespace snerf
