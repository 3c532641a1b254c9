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

typedef __attribute__((ext_vector_type(4))) float f32x4;

// A [rows x GBK] (K_CONTIG) or [GBK x rows] tile is moved in two phases (register staging, T14): `fetch` issues the
// global loads of the NEXT tile before the MFMA block, `stash` writes them to the k-major LDS image dst[k][r] after it,
// so HBM/L2 latency hides behind the matrix work.  8 elements per thread; 16-byte loads when the layout allows.
template <bool K_CONTIG>
__device__ __forceinline__ void fetch_tile(const float* __restrict__ P, int64_t sR, int64_t sK, int64_t r0, int64_t rows_total,
                                           int64_t k0, int64_t k_total, bool vec_ok, float (&v)[8], int tid) {
    if (K_CONTIG) {
        const int r = tid >> 1, kb = (tid & 1) * 8;           // thread -> (row, 8 consecutive k)
        const int64_t gr = r0 + r, gk = k0 + kb;
        if (vec_ok && gr < rows_total && gk + 8 <= k_total) {
            const f32x4* p = (const f32x4*)(P + gr * sR + gk);
            const f32x4 a = p[0], b = p[1];
            v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (gr < rows_total && gk + i < k_total) ? P[gr * sR + (gk + i) * sK] : 0.f;
        }
    } else {
        const int k = tid >> 4, rb = (tid & 15) * 8;          // thread -> (k, 8 consecutive rows)
        const int64_t gk = k0 + k, gr = r0 + rb;
        if (vec_ok && gk < k_total && gr + 8 <= rows_total) {
            const f32x4* p = (const f32x4*)(P + gk * sK + gr);
            const f32x4 a = p[0], b = p[1];
            v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (gk < k_total && gr + i < rows_total) ? P[(gr + i) * sR + gk * sK] : 0.f;
        }
    }
}
template <bool K_CONTIG>
__device__ __forceinline__ void stash_tile(const float (&v)[8], float (*dst)[GBM + GPAD], int tid) {
    if (K_CONTIG) {
        const int r = tid >> 1, kb = (tid & 1) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[kb + i][r] = v[i];
    } else {
        const int k = tid >> 4, rb = (tid & 15) * 8;
        f32x4* d = (f32x4*)&dst[k][rb];                       // (GBM+GPAD)*4 and rb*4 are multiples of 16 bytes
        d[0] = f32x4{v[0], v[1], v[2], v[3]};
        d[1] = f32x4{v[4], v[5], v[6], v[7]};
    }
}

template <bool A_K_CONTIG, bool B_K_CONTIG>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[2][GBK][GBM + GPAD];
    __shared__ __attribute__((aligned(16))) float Bs[2][GBK][GBN + GPAD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * GBM, n0 = (int64_t)blockIdx.y * GBN;
    // split-K range of this block
    const int64_t kchunk = (g.K + gridDim.z - 1) / gridDim.z;
    const int64_t kbeg = (int64_t)blockIdx.z * kchunk;
    const int64_t kend = kbeg + kchunk < g.K ? kbeg + kchunk : g.K;
    // 16-byte loads need: unit stride along the vectorised axis (by construction), the other stride a multiple of 4,
    // a 16-byte aligned base and a 4-aligned first index of this block
    const bool a_vec = ((uintptr_t)g.A % 16 == 0) && ((A_K_CONTIG ? g.sAm : g.sAk) % 4 == 0) && ((A_K_CONTIG ? kbeg : m0) % 4 == 0);
    const bool b_vec = ((uintptr_t)g.B % 16 == 0) && ((B_K_CONTIG ? g.sBn : g.sBk) % 4 == 0) && ((B_K_CONTIG ? kbeg : n0) % 4 == 0);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    if (kbeg < kend) {
        int buf = 0;
        float va[8], vb[8];
        fetch_tile<A_K_CONTIG>(g.A, g.sAm, g.sAk, m0, g.M, kbeg, kend, a_vec, va, tid);
        fetch_tile<B_K_CONTIG>(g.B, g.sBn, g.sBk, n0, g.N, kbeg, kend, b_vec, vb, tid);
        stash_tile<A_K_CONTIG>(va, As[0], tid);
        stash_tile<B_K_CONTIG>(vb, Bs[0], tid);
        __syncthreads();
        for (int64_t k0 = kbeg; k0 < kend; k0 += GBK) {
            const bool more = k0 + GBK < kend;
            if (more) {
                fetch_tile<A_K_CONTIG>(g.A, g.sAm, g.sAk, m0, g.M, k0 + GBK, kend, a_vec, va, tid);
                fetch_tile<B_K_CONTIG>(g.B, g.sBn, g.sBk, n0, g.N, k0 + GBK, kend, b_vec, vb, tid);
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
            if (more) {
                stash_tile<A_K_CONTIG>(va, As[buf ^ 1], tid);
                stash_tile<B_K_CONTIG>(vb, Bs[buf ^ 1], tid);
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

// =====================================================================================================
// bf16x3 "NT" GEMM:  C[m,n] (+)= alpha * (sum_k A[m,k] * Bt[n,k] + bias[n])
//   A  : fp32 activations / gradients [M, lda], k contiguous - split into bf16 hi/lo while staging into LDS
//   Bt : weights already split (split_weights_kernel) into bf16 hi / lo [N, Kp], Kp = K rounded up to 32, zero padded
// 3-term error-compensated product on v_mfma_f32_32x32x16_bf16 (hi*hi + lo*hi + hi*lo, fp32 accumulate): ~1e-5 relative,
// 16/3 x the fp32-MFMA rate - with 403 MB in and out per layer these GEMMs become HBM-bound.
namespace snerf {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

constexpr int XBM = 128, XBN = 128, XBK = 32, XLD = 40;      // LDS row stride in bf16 elements (80 B: conflict-free b128 reads)

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

__global__ void split_weights_kernel(const float* W, int rows, int cols, int transpose, uint16_t* hi, uint16_t* lo, int out_rows, int kp) {
    // out[r][k] = W[r][k] (transpose = 0, rows x cols) or W[k][r] (transpose = 1); zero padded to [out_rows, kp]
    const int64_t total = (int64_t)out_rows * kp;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / kp), k = (int)(i - (int64_t)r * kp);
        float v = 0.f;
        if (!transpose) { if (r < rows && k < cols) v = W[(int64_t)r * cols + k]; }
        else { if (k < rows && r < cols) v = W[(int64_t)k * cols + r]; }
        const __bf16 h = (__bf16)v;
        const float hf = (float)h;
        const __bf16 l = (__bf16)(v - hf);
        hi[i] = __builtin_bit_cast(uint16_t, h);
        lo[i] = __builtin_bit_cast(uint16_t, l);
    }
}

__global__ __launch_bounds__(256) void gemm_bf16x3_kernel(const GemmX g) {
    __shared__ __attribute__((aligned(16))) uint16_t Ah[XBM][XLD], Al[XBM][XLD], Bh[XBN][XLD], Bl[XBN][XLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * XBM, n0 = (int64_t)blockIdx.y * XBN;
    const int srow = tid >> 1, skh = (tid & 1) * 16;          // staging: thread -> (row, 16 consecutive k)
    const bool a_vec = ((uintptr_t)g.A % 16 == 0) && (g.lda % 4 == 0);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float va[16];
    u32x4 vbh[2], vbl[2];
    auto fetch = [&](int64_t k0) {
        const int64_t gr = m0 + srow, gk = k0 + skh;
        if (gr < g.M && a_vec && gk + 16 <= g.K) {
            const f32x4* p = (const f32x4*)(g.A + gr * g.lda + gk);
#pragma unroll
            for (int q = 0; q < 4; ++q) { const f32x4 t = p[q]; va[4 * q] = t[0]; va[4 * q + 1] = t[1]; va[4 * q + 2] = t[2]; va[4 * q + 3] = t[3]; }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) va[i] = (gr < g.M && gk + i < g.K) ? g.A[gr * g.lda + gk + i] : 0.f;
        }
        const int64_t gn = n0 + srow;
        if (gn < g.N) {
            const u32x4* ph = (const u32x4*)(g.Bh + gn * g.kp + gk);
            const u32x4* pl = (const u32x4*)(g.Bl + gn * g.kp + gk);
            vbh[0] = ph[0]; vbh[1] = ph[1]; vbl[0] = pl[0]; vbl[1] = pl[1];
        } else {
            vbh[0] = vbh[1] = vbl[0] = vbl[1] = u32x4{0, 0, 0, 0};
        }
    };
    auto stash = [&]() {
        u32x4 hq[2], lq[2];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            uint32_t hh, ll;
            split2_bf16(va[2 * q], va[2 * q + 1], hh, ll);
            hq[q >> 2][q & 3] = hh;
            lq[q >> 2][q & 3] = ll;
        }
        u32x4* dh = (u32x4*)&Ah[srow][skh];
        u32x4* dl = (u32x4*)&Al[srow][skh];
        dh[0] = hq[0]; dh[1] = hq[1]; dl[0] = lq[0]; dl[1] = lq[1];
        u32x4* eh = (u32x4*)&Bh[srow][skh];
        u32x4* el = (u32x4*)&Bl[srow][skh];
        eh[0] = vbh[0]; eh[1] = vbh[1]; el[0] = vbl[0]; el[1] = vbl[1];
    };
    const int64_t ksteps = (g.K + XBK - 1) / XBK;
    fetch(0);
    for (int64_t ks = 0; ks < ksteps; ++ks) {
        __syncthreads();                 // previous step's fragment reads are done
        stash();
        __syncthreads();
        if (ks + 1 < ksteps) fetch((ks + 1) * XBK);          // next tile's global loads fly during the MFMAs
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            u32x4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *(const u32x4*)&Ah[wm * 64 + i * 32 + r][kk * 16 + h * 8];
                al[i] = *(const u32x4*)&Al[wm * 64 + i * 32 + r][kk * 16 + h * 8];
                bh[i] = *(const u32x4*)&Bh[wn * 64 + i * 32 + r][kk * 16 + h * 8];
                bl[i] = *(const u32x4*)&Bl[wn * 64 + i * 32 + r][kk * 16 + h * 8];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bf16x8 Ahi = __builtin_bit_cast(bf16x8, ah[i]), Alo = __builtin_bit_cast(bf16x8, al[i]);
                    const bf16x8 Bhi = __builtin_bit_cast(bf16x8, bh[j]), Blo = __builtin_bit_cast(bf16x8, bl[j]);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Alo, Bhi, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ahi, Blo, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ahi, Bhi, acc[i][j], 0, 0, 0);
                }
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int64_t n = n0 + wn * 64 + j * 32 + r;
        const bool nok = n < g.N;
        const float bias = (g.bias && nok) ? g.bias[n] : 0.f;
        float colsum = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (nok && m < g.M) {
                    float v = g.alpha * (acc[i][j][e] + bias);
                    float* c = g.C + m * g.ldc + n;
                    if (g.accumulate) v += *c;
                    *c = v;
                    colsum += v;
                }
            }
        }
        if (g.colsum) {
            colsum += __shfl_xor(colsum, 32, 64);
            if (h == 0 && nok) atomicAdd(g.colsum + n, colsum);
        }
    }
}

hipError_t launch_split_weights(const float* W, int rows, int cols, bool transpose, uint16_t* hi, uint16_t* lo, int out_rows, int kp, hipStream_t st) {
    const int64_t total = (int64_t)out_rows * kp;
    if (total <= 0) return hipSuccess;
    int64_t b = (total + 255) / 256;
    if (b > 4096) b = 4096;
    hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)b), dim3(256), 0, st, W, rows, cols, transpose ? 1 : 0, hi, lo, out_rows, kp);
    return hipGetLastError();
}
hipError_t launch_gemm_bf16x3(const GemmX& g, hipStream_t st) {
    if (g.M <= 0 || g.N <= 0) return hipSuccess;
    dim3 grid((unsigned)((g.M + XBM - 1) / XBM), (unsigned)((g.N + XBN - 1) / XBN));
    hipLaunchKernelGGL(gemm_bf16x3_kernel, grid, dim3(256), 0, st, g);
    return hipGetLastError();
}

}  // namespace snerf
