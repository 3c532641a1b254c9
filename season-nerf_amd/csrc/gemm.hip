// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate - bitwise a
// k-ordered fmaf chain), used by the layer-wise TRAINING path where batch-statistics BatchNorm forces a global
// reduction between layers (reference: misc.py:169-170,188-189).
//
//   C[m,n] (+)= alpha * sum_k A(m,k) * B(k,n) [+ bias[n]]      A(m,k) = A[m*sAm + k*sAk],  B(k,n) = B[k*sBk + n*sBn]
//
// One kernel covers the three products of a Linear layer by strides:
//   forward   Z  = H  W^T     A = H [pts,K]  (k contiguous)      B(k,n) = W[n,k]  (k contiguous)
//   dgrad     dH = dZ W       A = dZ [pts,N] (k contiguous)      B(k,n) = W[k,n]  (n contiguous)
//   wgrad     dW = dZ^T H     A(m,k) = dZ[k,m] (m contiguous)    B(k,n) = H[k,n]  (n contiguous), K = #points, split
//                             over blockIdx.z with fp32 atomics (dW is 256 KB: ~1e3 adders per element, Guideline 12)
// Tile 128x128x16, 4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles; LDS tiles are k-major ([k][m], [k][n]) so the
// one-float-per-lane operands (A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]) are conflict-free ds_read_b32.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "train.h"

namespace snerf {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int GBM = 128, GBN = 128, GBK = 16, GPAD = 4;

// load a [rows x GBK] (K_CONTIG) or [GBK x rows] tile into k-major LDS: dst[k][r]
template <bool K_CONTIG>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, int64_t sR, int64_t sK, int64_t r0, int64_t rows_total,
                                          int64_t k0, int64_t k_total, float (*dst)[GBM + GPAD], int tid) {
    // 128 x 16 elements, 256 threads, 8 per thread
    if (K_CONTIG) {
        // thread -> (row = tid/2, 8 consecutive k)
        const int r = tid >> 1, kb = (tid & 1) * 8;
        const int64_t gr = r0 + r;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t gk = k0 + kb + i;
            float v = 0.f;
            if (gr < rows_total && gk < k_total) v = P[gr * sR + gk * sK];
            dst[kb + i][r] = v;
        }
    } else {
        // rows contiguous in memory: thread -> (k = tid/16, 8 consecutive rows)
        const int k = tid >> 4, rb = (tid & 15) * 8;
        const int64_t gk = k0 + k;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t gr = r0 + rb + i;
            float v = 0.f;
            if (gr < rows_total && gk < k_total) v = P[gr * sR + gk * sK];
            dst[k][rb + i] = v;
        }
    }
}

template <bool A_K_CONTIG, bool B_K_CONTIG>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs g) {
    __shared__ float As[2][GBK][GBM + GPAD];
    __shared__ float Bs[2][GBK][GBN + GPAD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * GBM, n0 = (int64_t)blockIdx.y * GBN;
    // split-K range of this block
    const int64_t kchunk = (g.K + gridDim.z - 1) / gridDim.z;
    const int64_t kbeg = (int64_t)blockIdx.z * kchunk;
    const int64_t kend = kbeg + kchunk < g.K ? kbeg + kchunk : g.K;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    if (kbeg < kend) {
        int buf = 0;
        load_tile<A_K_CONTIG>(g.A, g.sAm, g.sAk, m0, g.M, kbeg, kend, As[0], tid);
        load_tile<B_K_CONTIG>(g.B, g.sBn, g.sBk, n0, g.N, kbeg, kend, Bs[0], tid);
        __syncthreads();
        for (int64_t k0 = kbeg; k0 < kend; k0 += GBK) {
            if (k0 + GBK < kend) {
                load_tile<A_K_CONTIG>(g.A, g.sAm, g.sAk, m0, g.M, k0 + GBK, kend, As[buf ^ 1], tid);
                load_tile<B_K_CONTIG>(g.B, g.sBn, g.sBk, n0, g.N, k0 + GBK, kend, Bs[buf ^ 1], tid);
            }
#pragma unroll
            for (int kk = 0; kk < GBK; kk += 2) {
                const float a0 = As[buf][kk + h][wm * 64 + r], a1 = As[buf][kk + h][wm * 64 + 32 + r];
                const float b0 = Bs[buf][kk + h][wn * 64 + r], b1 = Bs[buf][kk + h][wn * 64 + 32 + r];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
            __syncthreads();
            buf ^= 1;
        }
    }
    // epilogue: C/D layout col = lane&31, row = (e&3) + 8*(e>>2) + 4*h
    const bool atomic = gridDim.z > 1 || (g.flags & GEMM_ATOMIC);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int64_t n = n0 + wn * 64 + j * 32 + r;
        const bool nok = n < g.N;
        const float bias = (g.bias && nok && blockIdx.z == 0) ? g.bias[n] : 0.f;
        float colsum = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (nok && m < g.M) {
                    float v = g.alpha * (acc[i][j][e] + bias);
                    float* c = g.C + m * g.ldc + n;
                    if (atomic) atomicAdd(c, v);
                    else {
                        if (g.flags & GEMM_ACCUM) v += *c;
                        *c = v;
                    }
                    colsum += v;
                }
            }
        }
        if (g.colsum) {     // per-column sum of the values just written (BatchNorm mean): lanes l and l+32 share a column
            colsum += __shfl_xor(colsum, 32, 64);
            if (h == 0 && nok) atomicAdd(g.colsum + n, colsum);
        }
    }
}

hipError_t launch_gemm(const GemmArgs& g, hipStream_t st) {
    if (g.M <= 0 || g.N <= 0) return hipSuccess;
    dim3 grid((unsigned)((g.M + GBM - 1) / GBM), (unsigned)((g.N + GBN - 1) / GBN), (unsigned)(g.splitk > 0 ? g.splitk : 1));
    const bool akc = g.sAk == 1, bkc = g.sBk == 1;
    if (akc && bkc) hipLaunchKernelGGL((gemm_kernel<true, true>), grid, dim3(256), 0, st, g);
    else if (akc && !bkc) hipLaunchKernelGGL((gemm_kernel<true, false>), grid, dim3(256), 0, st, g);
    else if (!akc && bkc) hipLaunchKernelGGL((gemm_kernel<false, true>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((gemm_kernel<false, false>), grid, dim3(256), 0, st, g);
    return hipGetLastError();
}

}  // namespace snerf
