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
#include <atomic>
#include <map>
#include <mutex>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "gemm_common.h"
#include "train.h"

namespace snerf {


constexpr int GBM = 128, GBN = 128, GBK = 16, GPAD = 4;


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
// bf16x3 "row-owner" GEMM (forward and dgrad of the training engine):
//     C[m,n] (+)= alpha * (sum_k A[m,k] * Bt[n,k] + bias[n])          [+ per-column shifted sums for train-mode BatchNorm]
//   A  : fp32 activations / gradients [M, lda], k contiguous - read straight from HBM into MFMA operand registers
//        (lane (r,h) of a 32x32x16 MFMA holds 8 consecutive k of row r: two 16-byte loads), split into bf16 hi/lo in registers
//   Bt : weights, pre-split by split_weights_kernel into MFMA fragment order, resident in LDS for the whole kernel
// Every wave owns its 32 rows: no barrier in the main loop, activations never touch LDS, four k-steps of loads are kept
// in flight.  Persistent: 1 workgroup (8 waves) per CU loops over 256-row tiles; the n-groups of
// one row tile sit on the same XCD so the second reader of a tile hits that XCD's L2.
// 3-term error-compensated product (hi*hi + lo*hi + hi*lo, fp32 accumulate): ~1e-5 relative, 16/3 x the fp32-MFMA rate,
// which makes these GEMMs HBM-bound (403 MB in + 403 MB out per layer at 4096 x 96).
namespace snerf {


constexpr int RO_PF = 4;                                     // k-steps of A loads in flight ahead of the MFMAs

// Fragment-order split: tile T (32 output columns), k-step ks (16 k): 1 KiB hi then 1 KiB lo; inside, lane (r,h) owns 16 bytes =
// bf16 of Bt[T*32 + r][ks*16 + h*8 + 0..7].  Bt[n][k] = W[n][k] (transpose = 0, W is [rows x cols]) or W[k][n] (transpose = 1).
__global__ void split_weights_kernel(const float* W, int rows, int cols, int transpose, uint16_t* frag, int n_tiles, int ksteps) {
    const int64_t total = (int64_t)n_tiles * ksteps * 512;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const int64_t tk = i >> 9;
        const int ks = (int)(tk % ksteps), T = (int)(tk / ksteps);
        const int n = T * 32 + (lane & 31), k = ks * 16 + (lane >> 5) * 8 + e;
        float v = 0.f;
        if (!transpose) { if (n < rows && k < cols) v = W[(int64_t)n * cols + k]; }
        else { if (k < rows && n < cols) v = W[(int64_t)k * cols + n]; }
        const __bf16 h = (__bf16)v;
        const __bf16 l = (__bf16)(v - (float)h);
        uint16_t* dst = frag + tk * 1024 + lane * 8 + e;
        dst[0] = __builtin_bit_cast(uint16_t, h);
        dst[512] = __builtin_bit_cast(uint16_t, l);
    }
}

// AOL ("activation on load"): the A operand is the stored PRE-activation Z of the layer below; its first act_cols columns
// become sin(2 pi (a z + b)) = one fma + one v_sin_f32 (which takes revolutions) while they sit in registers.  The table
// act_tab = [a | b], each act_cols long, folds BatchNorm and the 1/(2 pi): a = gamma*istd/(2 pi), b = (beta - gamma*mu*istd)/(2 pi)
// (a = 1/(2 pi), b = 0 for a plain SineLayer) - the post-activation H is never written to HBM.
// ACT (dgrad only): the value produced is dL/dH of the SineLayer below, whose pre-activation Z (ez) and [a | b] table (etab) are
// at hand: the epilogue multiplies by cos(2 pi (a z + b)) = dH/d(arg) - one fma + one v_cos_f32 - so that what reaches HBM is
// dL/d(arg) already, and accumulates the column sums sum v and sum v*xhat (xhat = (z - mu)*istd, BatchNorm only) that the
// bias / BatchNorm gradients and the BatchNorm backward need: the separate reduction sweep over [points x width] disappears.
template <int NT, bool AOL, bool ACT>
__global__ __launch_bounds__(512) void gemm_rows_kernel(const GemmX g) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_w[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int KS = g.ksteps;
    const int n_groups = (g.n_tiles + NT - 1) / NT;
    // block -> (XCD, slot on the XCD) -> (n-group, worker): all n-groups of a worker share an XCD (and its L2)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int workers_per_xcd = slots / n_groups;
    if (slot >= workers_per_xcd * n_groups) return;
    const int grp = slot % n_groups, worker = (slot / n_groups) * 8 + xcd, n_workers = workers_per_xcd * 8;
    const int tiles_here = (g.n_tiles - grp * NT) < NT ? (g.n_tiles - grp * NT) : NT;

    {   // weights of this n-group -> LDS (fragment order, straight copy)
        const u32x4* src = (const u32x4*)(g.frag + (int64_t)grp * NT * KS * 1024);
        u32x4* dst = (u32x4*)lds_w;
        const int n16 = tiles_here * KS * 128;
        for (int i = tid; i < n16; i += 512) dst[i] = src[i];
    }
    // AOL: the [a | b] table goes behind the weights when the 160 KiB allow it (every lane of a half-wave reads the same
    // 16 bytes: an LDS broadcast instead of an L1 round trip per k-step); otherwise it is read from global memory
    const uint8_t* lds_tab = lds_w + (size_t)NT * KS * 2048;
    const bool tab_in_lds = AOL && g.tab_lds;
    if (tab_in_lds) {
        float* dst = (float*)lds_tab;
        for (int i = tid; i < 2 * g.act_cols; i += 512) dst[i] = g.act_tab[i];
    }
    __syncthreads();

    const bool a_vec = ((uintptr_t)g.A % 16 == 0) && (g.lda % 4 == 0);
    const int64_t n_row_tiles = (g.M + RO_ROWS - 1) / RO_ROWS;
    float st1[NT], st2[NT];                                          // shifted column sums over this block's rows
#pragma unroll
    for (int j = 0; j < NT; ++j) st1[j] = st2[j] = 0.f;

    for (int64_t rt = worker; rt < n_row_tiles; rt += n_workers) {
        const int64_t row0 = rt * RO_ROWS + wave * (RO_MT * 32);
        const float* arow[RO_MT];
#pragma unroll
        for (int i = 0; i < RO_MT; ++i) {
            int64_t m = row0 + i * 32 + r;
            if (m > g.M - 1) m = g.M - 1;                            // loads stay in bounds, stores are masked
            arow[i] = g.A + m * g.lda + h * 8;
        }
        auto load_a = [&](int ks, float (&v)[RO_MT][8]) {
            const int k0 = ks * 16 + h * 8;
#pragma unroll
            for (int i = 0; i < RO_MT; ++i) {
                const float* p = arow[i] + ks * 16;
                if (a_vec && k0 + 8 <= g.K) {
                    const f32x4 x = *(const f32x4*)p, y = *(const f32x4*)(p + 4);
                    v[i][0] = x[0]; v[i][1] = x[1]; v[i][2] = x[2]; v[i][3] = x[3];
                    v[i][4] = y[0]; v[i][5] = y[1]; v[i][6] = y[2]; v[i][7] = y[3];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[i][e] = (k0 + e < g.K) ? p[e] : 0.f;
                }
            }
        };
        f32x16 acc[RO_MT][NT];
#pragma unroll
        for (int i = 0; i < RO_MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        float pf[RO_PF][RO_MT][8];
        f32x4 tb[2][4];                                        // AOL: table rows of two k-steps (this one and the next)
        auto load_tab = [&](int ks, f32x4 (&t_)[4]) {
            const int k0 = ks * 16 + h * 8;
            if (k0 < g.act_cols) {                             // act_cols is a multiple of 8 (checked by the launcher)
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const f32x4* p = tab_in_lds ? (const f32x4*)(lds_tab + ((size_t)a * g.act_cols + k0) * 4)
                                                : (const f32x4*)(g.act_tab + (int64_t)a * g.act_cols + k0);
                    t_[2 * a] = p[0];
                    t_[2 * a + 1] = p[1];
                }
            }
        };
        if (AOL) load_tab(0, tb[0]);
#pragma unroll
        for (int d = 0; d < RO_PF; ++d)
            if (d < KS) load_a(d, pf[d]);
        for (int ks0 = 0; ks0 < KS; ks0 += RO_PF) {
#pragma unroll
            for (int d = 0; d < RO_PF; ++d) {
                const int ks = ks0 + d;
                if (ks < KS) {
                    u32x4 ahi[RO_MT], alo[RO_MT];
                    if (AOL) {
                        if (ks + 1 < KS) load_tab(ks + 1, tb[(d + 1) & 1]);
                        if (ks * 16 + h * 8 < g.act_cols) {
                            const f32x4(&t_)[4] = tb[d & 1];
#pragma unroll
                            for (int i = 0; i < RO_MT; ++i)
#pragma unroll
                                for (int e = 0; e < 8; ++e)
                                    pf[d][i][e] = __builtin_amdgcn_sinf(__builtin_fmaf(t_[e >> 2][e & 3], pf[d][i][e], t_[2 + (e >> 2)][e & 3]));
                        }
                    }
#pragma unroll
                    for (int i = 0; i < RO_MT; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            uint32_t hh, ll;
                            split2_bf16(pf[d][i][2 * q], pf[d][i][2 * q + 1], hh, ll);
                            ahi[i][q] = hh;
                            alo[i][q] = ll;
                        }
                    if (ks + RO_PF < KS) load_a(ks + RO_PF, pf[d]);          // refill the slot just consumed
                    const uint32_t base = (uint32_t)ks * 2048u + (uint32_t)lane * 16u;
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        if (j < tiles_here) {
                            const u32x4 bh = *(const u32x4*)(lds_w + base + (uint32_t)j * KS * 2048u);
                            const u32x4 bl = *(const u32x4*)(lds_w + base + (uint32_t)j * KS * 2048u + 1024u);
                            const bf16x8 Bhi = __builtin_bit_cast(bf16x8, bh), Blo = __builtin_bit_cast(bf16x8, bl);
#pragma unroll
                            for (int i = 0; i < RO_MT; ++i) {
                                const bf16x8 Ahi = __builtin_bit_cast(bf16x8, ahi[i]), Alo = __builtin_bit_cast(bf16x8, alo[i]);
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Alo, Bhi, acc[i][j], 0, 0, 0);
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ahi, Blo, acc[i][j], 0, 0, 0);
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ahi, Bhi, acc[i][j], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        }
        // epilogue: D[row = (e&3) + 8(e>>2) + 4h, col = r]
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int64_t n = (int64_t)(grp * NT + j) * 32 + r;
            const bool nok = j < tiles_here && n < g.N;
            const float bias = (g.bias && nok) ? g.bias[n] : 0.f;
            const float shift = (!ACT && g.stats && nok) ? g.alpha * bias : 0.f;
            float zt[RO_MT][16];
            float e_a = 0.f, e_b = 0.f, e_mu = 0.f, e_is = 0.f;
            if (ACT) {      // the 16 pre-activations of this column are all in flight before the first cosine (clamped, no branches)
                const int64_t nc = nok ? n : 0;
                e_a = g.etab[nc]; e_b = g.etab[g.N + nc];
                if (g.emu) { e_mu = g.emu[nc]; e_is = g.eistd[nc]; }
#pragma unroll
                for (int i = 0; i < RO_MT; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        int64_t m = row0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                        m = m < g.M ? m : g.M - 1;
                        zt[i][e] = g.ez[m * g.eld + nc];
                    }
            }
#pragma unroll
            for (int i = 0; i < RO_MT; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t m = row0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (nok && m < g.M) {
                        float v = g.alpha * (acc[i][j][e] + bias);
                        float* c = g.C + m * g.ldc + n;
                        if (g.accumulate) v += *c;                       // ACT: this dgrad is the last of several producers of dL/dH
                        if (ACT) {
                            const float z = zt[i][e];
                            v *= __builtin_amdgcn_cosf(__builtin_fmaf(e_a, z, e_b));
                            *c = v;
                            st1[j] += v;
                            st2[j] += v * ((z - e_mu) * e_is);
                        } else {
                            *c = v;
                            const float d = v - shift;
                            st1[j] += d;
                            st2[j] += d * d;
                        }
                    }
                }
        }
    }
    if (g.stats) {      // per-column sums (forward: sum(v - shift), sum((v - shift)^2); ACT: sum v, sum v*xhat) -> double atomics
        __syncthreads();                                   // weights no longer needed: reuse LDS for the cross-wave reduction
        float* red = (float*)lds_w;                        // [8 waves][NT][2][32]
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float a = st1[j] + __shfl_xor(st1[j], 32, 64), b = st2[j] + __shfl_xor(st2[j], 32, 64);
            if (h == 0) {
                red[((wave * NT + j) * 2 + 0) * 32 + r] = a;
                red[((wave * NT + j) * 2 + 1) * 32 + r] = b;
            }
        }
        __syncthreads();
        if (tid < NT * 64) {
            const int j = tid >> 6, which = (tid >> 5) & 1, c = tid & 31;
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < RO_WAVES; ++w) s += (double)red[((w * NT + j) * 2 + which) * 32 + c];
            const int64_t n = (int64_t)(grp * NT + j) * 32 + c;
            if (j < tiles_here && n < g.N) atomicAdd(g.stats + which * g.N + n, s);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Full-tile form of gemm_rows_kernel, the one the training step runs on: K a multiple of 16 (or zero-padded to one), every
// n-group complete, 16-byte aligned rows.  Same arithmetic, same fragment layout; what changes is that NO global load sits inside a
// branch, so hipcc counts them precisely (`s_waitcnt vmcnt(2*(PF-1))` instead of the `vmcnt(0)` at the loop head that the
// guarded loads of the general kernel force), and the A stream becomes a real software pipeline:
//   * PF k-steps of A in flight per wave, refilled slot by slot right after the slot is split;
//   * the refills of a tile's last PF k-steps fetch the first PF k-steps of the wave's NEXT tile, so the epilogue (stores,
//     activation backward) overlaps their latency and a tile starts with its operands landed;
//   * the weights reach LDS eight 16-byte loads per thread at a time instead of one round trip per 8 KiB.
// AOL: 0 none, 1 [a | b] table in LDS.  ACT: 0 none, 1 activation backward in the epilogue (without `accumulate`).
// The A stream of the full-tile kernel is issued and awaited by hand.  With compiler-visible loads hipcc emits `s_waitcnt vmcnt(0)`
// at the head of the k-loop for the plain and LDS-table variants (every refill of the previous round awaited at once: no
// overlap), whatever the surrounding code looks like; it does count precisely for the variants that have other global loads in
// the tile loop.  Hand-issued loads sidestep the question: loads complete in issue order, so `vmcnt(N)` with N = the number of
// A loads issued after the awaited pair is exact in the k-loop and merely conservative when compiler-issued stores/loads are
// younger still (they only add to the count).  The registers are handed to the compiler by the wait (its "+v" operands).
__device__ __forceinline__ void a8_issue(const float* p, f32x4& x, f32x4& y) {
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16" : "=&v"(x), "=&v"(y) : "v"(p));
}
template <int N>
__device__ __forceinline__ void a8_wait(f32x4& x, f32x4& y) {
    asm volatile("s_waitcnt vmcnt(%2) ; a8_wait %0 %1" : "+v"(x), "+v"(y) : "n"(N));
}

template <int NT, int PF, int AOL, int ACT>
__global__ __launch_bounds__(512) void gemm_rows_full_kernel(const GemmX g) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_w[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int KS = g.ksteps;                                        // multiple of PF
    const int n_groups = g.n_tiles / NT;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int workers_per_xcd = slots / n_groups;
    if (slot >= workers_per_xcd * n_groups) return;
    const int grp = slot % n_groups, worker = (slot / n_groups) * 8 + xcd, n_workers = workers_per_xcd * 8;

    {
        const u32x4* src = (const u32x4*)(g.frag + (int64_t)grp * NT * KS * 1024);
        u32x4* dst = (u32x4*)lds_w;
        const int n16 = NT * KS * 128;
        int i0 = tid;
        for (; i0 + 7 * 512 < n16; i0 += 512 * 8) {                  // n16 is a multiple of 512
            u32x4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = src[i0 + q * 512];
#pragma unroll
            for (int q = 0; q < 8; ++q) dst[i0 + q * 512] = v[q];
        }
        for (; i0 < n16; i0 += 512) dst[i0] = src[i0];
    }
    const uint8_t* lds_tab = lds_w + (size_t)NT * KS * 2048;
    if (AOL == 1) {
        float* dst = (float*)lds_tab;
        for (int i = tid; i < 2 * g.act_cols; i += 512) dst[i] = g.act_tab[i];
    }
    // per-column constants of this lane's NT output columns (the activation-backward ones are fetched per column in the epilogue:
    // that variant has no registers to spare)
    float biasv[NT], shiftv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int64_t n = (int64_t)(grp * NT + j) * 32 + r;
        biasv[j] = (!ACT && g.bias && n < g.N) ? g.bias[n] : 0.f;
        shiftv[j] = (!ACT && g.stats) ? g.alpha * biasv[j] : 0.f;
    }
    __syncthreads();

    const int64_t n_row_tiles = (g.M + RO_ROWS - 1) / RO_ROWS;
    float st1[NT], st2[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) st1[j] = st2[j] = 0.f;

    auto a_ptr = [&](int64_t rt) {
        int64_t m = rt * RO_ROWS + wave * 32 + r;
        m = m < g.M ? m : g.M - 1;                                  // loads stay in bounds, stores are masked
        return g.A + m * g.lda + h * 8;
    };
    // epilogue addressing through buffer instructions: descriptor (scalar) + per-lane byte offset (lz / lc, one register each,
    // loop-invariant) + wave-uniform row offset (scalar) + 128 j as the immediate - no 64-bit vector address arithmetic
    const int lz = ACT ? (int)(4 * h * g.eld + grp * NT * 32 + r) * 4 : 0;
    // a thin layer (N < 32, one n-tile): lanes past N get an offset beyond the descriptor's range - the hardware drops their stores
    const bool nok0 = NT > 1 || grp * 32 + r < g.N;
    const int lc = nok0 ? (int)(4 * h * g.ldc + grp * NT * 32 + r) * 4 : (int)0x80000000;
    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc((void*)g.C, 0, (int)(g.M * g.ldc * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc((void*)(ACT ? g.ez : g.A), 0, -1, 0x00020000);
    int64_t rt = worker;
    const float* arow = a_ptr(rt < n_row_tiles ? rt : n_row_tiles - 1);
    f32x4 px[PF], py[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d) a8_issue(arow + d * 16, px[d], py[d]);

    for (; rt < n_row_tiles; rt += n_workers) {
        const int64_t rn = rt + n_workers;
        const float* anext = a_ptr(rn < n_row_tiles ? rn : rt);
        f32x16 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        f32x4 tb[2][4];
        auto load_tab = [&](int ks, f32x4 (&t_)[4]) {
            int k0 = ks * 16 + h * 8;
            k0 = k0 < g.act_cols ? k0 : g.act_cols - 8;            // clamped: the loads are unconditional
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const f32x4* p = AOL == 1 ? (const f32x4*)(lds_tab + ((size_t)a * g.act_cols + k0) * 4)
                                          : (const f32x4*)(g.act_tab + (int64_t)a * g.act_cols + k0);
                t_[2 * a] = p[0];
                t_[2 * a + 1] = p[1];
            }
        };
        if (AOL) load_tab(0, tb[0]);
        for (int ks0 = 0; ks0 < KS; ks0 += PF) {
            const float* src = (ks0 + PF < KS) ? arow + (ks0 + PF) * 16 : anext;
#pragma unroll
            for (int d = 0; d < PF; ++d) {
                const int ks = ks0 + d;
                a8_wait<2 * (PF - 1)>(px[d], py[d]);                 // the PF-1 younger refills stay in flight
                float a8[8] = {px[d][0], px[d][1], px[d][2], px[d][3], py[d][0], py[d][1], py[d][2], py[d][3]};
                if (AOL) {
                    load_tab(ks + 1 < KS ? ks + 1 : ks, tb[(d + 1) & 1]);
                    if (ks * 16 < g.act_cols) {                     // act_cols is a multiple of 16 here: uniform
                        const f32x4(&t_)[4] = tb[d & 1];
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            a8[e] = __builtin_amdgcn_sinf(__builtin_fmaf(t_[e >> 2][e & 3], a8[e], t_[2 + (e >> 2)][e & 3]));
                    }
                }
                u32x4 ahi, alo;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint32_t hh, ll;
                    split2_bf16(a8[2 * q], a8[2 * q + 1], hh, ll);
                    ahi[q] = hh;
                    alo[q] = ll;
                }
                a8_issue(src + d * 16, px[d], py[d]);                // refill the slot just consumed (next tile's on the last round)
                // one scheduling barrier per k-step: the MFMAs below may still interleave with the next k-step's wait,
                // activation and split VALU work, but nothing wanders further
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8 Ahi = __builtin_bit_cast(bf16x8, ahi), Alo = __builtin_bit_cast(bf16x8, alo);
                const uint32_t base = (uint32_t)ks * 2048u + (uint32_t)lane * 16u;
                bf16x8 Bhi[NT], Blo[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    Bhi[j] = __builtin_bit_cast(bf16x8, *(const u32x4*)(lds_w + base + (uint32_t)j * KS * 2048u));
                    Blo[j] = __builtin_bit_cast(bf16x8, *(const u32x4*)(lds_w + base + (uint32_t)j * KS * 2048u + 1024u));
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Alo, Bhi[j], acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ahi, Blo[j], acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ahi, Bhi[j], acc[j], 0, 0, 0);
            }
        }
        // epilogue: D[row = (e&3) + 8(e>>2) + 4h, col = r]
        const int64_t rowu = rt * RO_ROWS + wave * 32;
        const int64_t row0 = rowu + 4 * h;
        auto epilogue = [&](auto interior_tag) {
            constexpr bool INTERIOR = decltype(interior_tag)::value;      // no row of the workgroup tile is masked: branch-free
            float zt[2][16], ec[2][4];
            auto fetch = [&](int j, float (&z_)[16], float (&c_)[4]) {    // ACT: pre-activations and [a, b, mu, istd] of column j
                const int64_t n = (int64_t)(grp * NT + j) * 32 + r;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t ro = (e & 3) + 8 * (e >> 2);
                    if (INTERIOR) {
                        z_[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_z, lz + j * 128, (int)((rowu + ro) * g.eld * 4), 0));
                    } else {
                        int64_t m = row0 + ro;
                        m = m < g.M ? m : g.M - 1;
                        z_[e] = g.ez[m * g.eld + n];
                    }
                }
                c_[0] = g.etab[n]; c_[1] = g.etab[g.N + n];
                c_[2] = g.emu[n]; c_[3] = g.eistd[n];                     // the launcher substitutes zeros for a layer without BatchNorm
            };
            if (ACT) fetch(0, zt[0], ec[0]);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int64_t n = (int64_t)(grp * NT + j) * 32 + r;
                if (ACT) {      // column j+1's operands fly while column j is finished; the barriers keep hipcc from hoisting all four
                    if (j + 1 < NT) fetch(j + 1, zt[(j + 1) & 1], ec[(j + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t ro = (e & 3) + 8 * (e >> 2);
                    float v = g.alpha * (acc[j][e] + biasv[j]);
                    if (ACT) v *= __builtin_amdgcn_cosf(__builtin_fmaf(ec[j & 1][0], zt[j & 1][e], ec[j & 1][1]));
                    const bool ok = INTERIOR || row0 + ro < g.M;
                    if (INTERIOR) {
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rs_c, lc + j * 128, (int)((rowu + ro) * g.ldc * 4), 0);
                    } else if (ok && nok0) {
                        g.C[(row0 + ro) * g.ldc + n] = v;
                    }
                    if (ACT) {
                        const float s1 = v, s2 = v * ((zt[j & 1][e] - ec[j & 1][2]) * ec[j & 1][3]);
                        st1[j] += ok ? s1 : 0.f;
                        st2[j] += ok ? s2 : 0.f;
                    } else {
                        const float d = v - shiftv[j];
                        st1[j] += ok ? d : 0.f;
                        st2[j] += ok ? d * d : 0.f;
                    }
                }
                if (ACT) __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (ACT) {      // this variant's epilogue spills registers: the prefetched operands must have landed before it may touch them
#pragma unroll
            for (int d = 0; d < PF; ++d) a8_wait<0>(px[d], py[d]);
        }
        if (rt * RO_ROWS + RO_ROWS <= g.M) epilogue(std::true_type{});
        else epilogue(std::false_type{});
        arow = anext;
    }
    // The last round's refills (and the first loads of a worker without tiles) are never consumed.  They must have landed
    // before the code below may reuse their registers: a late-landing load would overwrite whatever lives there by then
    // (an address, for instance).  The wait also keeps the registers allocated up to this point.
#pragma unroll
    for (int d = 0; d < PF; ++d) a8_wait<0>(px[d], py[d]);
    if (g.stats) {
        __syncthreads();
        float* red = (float*)lds_w;                        // [8 waves][NT][2][32]
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float a = st1[j] + __shfl_xor(st1[j], 32, 64), b = st2[j] + __shfl_xor(st2[j], 32, 64);
            if (h == 0) {
                red[((wave * NT + j) * 2 + 0) * 32 + r] = a;
                red[((wave * NT + j) * 2 + 1) * 32 + r] = b;
            }
        }
        __syncthreads();
        if (tid < NT * 64) {
            const int j = tid >> 6, which = (tid >> 5) & 1, c = tid & 31;
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < RO_WAVES; ++w) s += (double)red[((w * NT + j) * 2 + which) * 32 + c];
            const int64_t n = (int64_t)(grp * NT + j) * 32 + c;
            if (n < g.N) atomicAdd(g.stats + which * g.N + n, s);
        }
    }
}

// =====================================================================================================
// bf16x3 weight-gradient GEMM:   dW[o, i] += alpha * sum_m dZ[m, o] * In[m, i]           (K = #points, split over workgroups)
// Both operands are point-major, so the 8 consecutive-k values an MFMA lane needs are 8 rows of one column: each lane gathers
// them with 8 row-coalesced dword loads (32 lanes = 128 contiguous bytes of a row), splits them into bf16 hi/lo in registers
// and publishes the finished 1 KiB fragments through LDS, where the 8 waves of the workgroup share them: the workgroup
// holds the whole dW block (up to 256 x 256, 128 accumulator registers per lane), so dZ and In are read from HBM exactly once.
// Stage = 32 points; double-buffered LDS (2 x 64 KiB), one barrier per stage; the loads of stage s+1 fly during the MFMAs of s.
// Direction of the next streaming kernel (row GEMMs on the 16x16x32 form, weight gradients): they alternate, so that a kernel starts with the
// rows its producer touched last - some of which are still in the Infinity Cache (forward 256 -> 256: 182 -> 177 us; SNERF_SNAKE=0: always forwards)
static int ro_grid_blocks_public();
// Launch context of the calling thread (set by a training pass, train.cpp CtxGuard): the launch parity of ITS trainer - reset at the start of
// every pass, so the direction of each launch (and with it the order in which wgrad_bf16x3_kernel accumulates its fp32 stages) is a function of the
// launch's position in the pass, not of the process's launch history - and that trainer's pre-allocated partial-sum scratch.
struct LaunchCtx { float* wgrad_partial = nullptr; size_t wgrad_floats = 0; unsigned* parity = nullptr; };
static thread_local LaunchCtx tl_ctx;
void gemm_launch_context(float* wgrad_partial, size_t wgrad_floats, unsigned* parity) { tl_ctx = LaunchCtx{wgrad_partial, wgrad_floats, parity}; }
size_t gemm_wgrad_partial_floats() { return (size_t)(ro_grid_blocks_public() < 256 ? 256 : ro_grid_blocks_public()) * 8 * 2 * 4 * 1024; }

int stream_direction(int64_t rows) {
    static std::atomic<unsigned> launches{0};        // stand-alone launches (snerf_linear_*): one process-wide parity
    static const int snake = [] { const char* e = getenv("SNERF_SNAKE"); return (e && e[0] == '0') ? 0 : 1; }();
    if (!snake || rows < 32768) return 0;                  // (a small launch neither gains from a direction nor takes a turn)
    if (tl_ctx.parity) return (int)((*tl_ctx.parity)++ & 1u);
    return (int)(launches.fetch_add(1, std::memory_order_relaxed) & 1u);
}

static int ro_grid_blocks();
static int ro_grid_blocks_public() { return ro_grid_blocks(); }
static int ro_grid_blocks() {
    static int n = 0;
    if (!n) {
        hipDeviceProp_t p;
        int dev = 0;
        n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ? p.multiProcessorCount : 256;
        n = n / 8 * 8;
        if (n < 8) n = 8;
    }
    return n;
}

struct WgradX {
    const float* in_tab;         // optional activation-on-load table for the first in_cols columns of In ([a | b], see gemm_rows_kernel)
    int in_cols;
    int in_tab_stride;           // distance between an input column's a and b in the table (>= in_cols: a column window of a wider table)
    // BNZ (optional): dZ holds dL/dY of a BatchNorm SineLayer on entry; the BatchNorm backward
    //   dZ = gamma*istd*(dY - mean(dY) - xhat*mean(dY*xhat)),  xhat = (z - mu)*istd
    // is applied to each gathered value, written back IN PLACE (every element is gathered by exactly one lane when the grid has
    // one z-block) and summed per column into dbias (+= bias_alpha * sum dZ): the separate dZ sweep disappears
    const float* z;
    int64_t ldzz;
    const float *bn_gamma, *bn_mu, *bn_istd, *bn_sdy, *bn_sdyx;
    float bn_inv_m, bias_alpha;
    float* dbias;
    float* dZ;
    const float* In;
    float* dW;
    int64_t M, ldz, ldi, ldw;
    int n_out, n_in;
    float alpha;
    int64_t rows_per_block;      // multiple of 32
    // two-stage reduction over the row blocks (optional): every workgroup stores its 256 x 256 block of partial sums here in
    // register order, [block x][block y][block z][wave][a][b][e][lane], and wgrad_reduce_kernel adds them up into dW.  The other
    // way - 64 Ki atomic adds per workgroup - is bound by the atomic rate of a CU (one 256-byte wave instruction per ~50 ns,
    // MI355X_MICROARCH.md): ~50 us at the tail of EVERY launch, whatever its size
    float* partial;
    int reverse;                 // stages from the last to the first (stream_direction())
};

constexpr int WG_STAGE = 32;

// FULL: every 256 x 256 block of dW is complete (no tile or column masks in the hot loop).  TA x TB: 32 x 32 tiles of dW per wave - the
// workgroup's block is 4 TA tiles of n_out by 2 TB tiles of n_in (2 x 4: 256 x 256; 1 x 2: 128 x 128, so that a 128-wide layer keeps all
// eight waves multiplying instead of two)
template <bool FULL, bool BNZ, int TA = 2, int TB = 4>
__global__ __launch_bounds__(512) void wgrad_bf16x3_kernel(const WgradX g) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_f[];          // [2 buffers][2 operands][8 tiles][2 ksteps][hi,lo][1 KiB]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    constexpr int NTO = 4 * TA, NTI = 2 * TB;              // 32-column tiles of each operand in this workgroup's block
    const int o_base = blockIdx.y * (32 * NTO), i_base = blockIdx.z * (32 * NTI);
    const int to_n = FULL ? NTO : ((g.n_out - o_base + 31) / 32 < NTO ? (g.n_out - o_base + 31) / 32 : NTO);      // valid tiles of each operand
    const int ti_n = FULL ? NTI : ((g.n_in - i_base + 31) / 32 < NTI ? (g.n_in - i_base + 31) / 32 : NTI);
    const int wo = wave >> 1, wi = wave & 1;                 // this wave's dW piece: n_out tiles {TA wo + a} x n_in tiles {TB wi + b}
    // rows: 32-row stages dealt round-robin over the row blocks (stage s of block x = rows 32 (s gridDim.x + x) ...): the workgroups walk
    // through the operands side by side, a few MiB apart in total.  With one contiguous range per workgroup (round 2) the 256 streams sat
    // 1.5 MiB apart and HBM delivered ~15 % less (tools/probes/copy_patterns.hip: 4.7 against 5.5 TB/s for that spacing)
    const int64_t stages_total = (g.M + WG_STAGE - 1) / WG_STAGE;
    if ((int64_t)blockIdx.x >= stages_total) return;
    const int n_stages = (int)((stages_total - blockIdx.x + gridDim.x - 1) / gridDim.x);

    // producer role: wave w gathers tile w of dZ and one tile of In (both k-steps of the stage) - In tile w, or, where dZ has only four
    // tiles, In tile 7 - w, so that the waves without a dZ tile take the In tiles first
    const int tile_i = TA == 1 ? 7 - wave : wave;
    const bool make_o = FULL || wave < to_n, make_i = FULL || tile_i < ti_n;
    const int col_o = o_base + wave * 32 + r, col_i = i_base + tile_i * 32 + r;
    const bool ok_o = FULL || (make_o && col_o < g.n_out), ok_i = FULL || (make_i && col_i < g.n_in);
    // branch-free gathers: a wave-uniform 64-bit stage base plus a 32-bit lane offset (row clamped to the last valid row of this
    // workgroup's range); out-of-range values are zeroed at publish time so that nothing depends on the loads before then
    const uint32_t ldz = (uint32_t)g.ldz, ldi = (uint32_t)g.ldi;
    const uint32_t co = ok_o ? (uint32_t)col_o : 0u, ci = ok_i ? (uint32_t)col_i : 0u;
    const int64_t m_last = g.M - 1;
    float vo[2][8], vi[2][8], vz[BNZ ? 2 : 1][BNZ ? 8 : 1];
    int gathered_last = 0;
    int64_t gathered_ms = 0;
    // BNZ: constants of this lane's dZ column
    const uint32_t ldzz = BNZ ? (uint32_t)g.ldzz : 0u;
    const float z_is = (BNZ && ok_o) ? g.bn_istd[col_o] : 0.f, z_mu = (BNZ && ok_o) ? g.bn_mu[col_o] : 0.f;
    const float z_k = (BNZ && ok_o) ? g.bn_gamma[col_o] * z_is : 0.f;
    const float z_ma = (BNZ && ok_o) ? g.bn_sdy[col_o] * g.bn_inv_m : 0.f, z_mb = (BNZ && ok_o) ? g.bn_sdyx[col_o] * g.bn_inv_m : 0.f;
    float z_sum = 0.f;
    // activation on load: this lane's In column is a stored pre-activation -> sin(2 pi (a z + b)) at publish time
    const bool in_act = g.in_tab != nullptr && ok_i && col_i < g.in_cols;
    const float c_a = in_act ? g.in_tab[col_i] : 0.f, c_b = in_act ? g.in_tab[g.in_tab_stride + col_i] : 0.f;
    auto gather = [&](int stage) {
        const int64_t ms = ((int64_t)(g.reverse ? n_stages - 1 - stage : stage) * gridDim.x + blockIdx.x) * WG_STAGE;      // uniform
        const int last_rel = (int)(m_last - ms < 63 ? m_last - ms : 63);         // >= 0: the stage exists
        const float* bo = g.dZ + ms * g.ldz;
        const float* bi = g.In + ms * g.ldi;
        const float* bz = BNZ ? g.z + ms * g.ldzz : nullptr;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = h * 8 + ks * 16 + e;
                const uint32_t kr = (uint32_t)(k < last_rel ? k : last_rel);
                if (FULL || make_o) vo[ks][e] = bo[kr * ldz + co];          // raw: masking waits for publish, so nothing here depends on the
                if (FULL || make_i) vi[ks][e] = bi[kr * ldi + ci];          // loads and they stay in flight across the MFMA block
                if (BNZ && (FULL || make_o)) vz[ks][e] = bz[kr * ldzz + co];      // (wave-uniform conditions: a wave without a tile loads nothing)
            }
        gathered_last = last_rel;
        gathered_ms = ms;
    };
    auto publish = [&](int buf) {
        uint8_t* base = lds_f + buf * 65536;
        if (BNZ) {      // dY -> dZ in registers, back to HBM in place (rows past the range are clamped duplicates of the last row: same value)
            float* wb = g.dZ + gathered_ms * g.ldz;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float dz = z_k * (vo[ks][e] - z_ma - ((vz[ks][e] - z_mu) * z_is) * z_mb);
                    vo[ks][e] = dz;
                    z_sum += (h * 8 + ks * 16 + e <= gathered_last) ? dz : 0.f;
                }
            if (ok_o) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int k = h * 8 + ks * 16 + e;
                        const uint32_t kr = (uint32_t)(k < gathered_last ? k : gathered_last);
                        wb[kr * ldz + co] = vo[ks][e];
                    }
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 oh, ol, ih, il;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t a, b;
                const bool k0 = h * 8 + ks * 16 + 2 * q <= gathered_last, k1 = h * 8 + ks * 16 + 2 * q + 1 <= gathered_last;
                split2_bf16((ok_o && k0) ? vo[ks][2 * q] : 0.f, (ok_o && k1) ? vo[ks][2 * q + 1] : 0.f, a, b);
                oh[q] = a; ol[q] = b;
                float i0 = vi[ks][2 * q], i1 = vi[ks][2 * q + 1];
                if (in_act) {
                    i0 = __builtin_amdgcn_sinf(__builtin_fmaf(c_a, i0, c_b));
                    i1 = __builtin_amdgcn_sinf(__builtin_fmaf(c_a, i1, c_b));
                }
                split2_bf16((ok_i && k0) ? i0 : 0.f, (ok_i && k1) ? i1 : 0.f, a, b);
                ih[q] = a; il[q] = b;
            }
            const uint32_t f = (uint32_t)((wave * 2 + ks) * 2048 + lane * 16), fi = (uint32_t)((tile_i * 2 + ks) * 2048 + lane * 16);
            if (make_o) { *(u32x4*)(base + f) = oh; *(u32x4*)(base + f + 1024) = ol; }
            if (make_i) { *(u32x4*)(base + 32768 + fi) = ih; *(u32x4*)(base + 32768 + fi + 1024) = il; }
        }
    };

    f32x16 acc[TA][TB];
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    gather(0);
    publish(0);
    __syncthreads();
    for (int s = 0; s < n_stages; ++s) {
        if (s + 1 < n_stages) gather(s + 1);                 // in flight during the MFMAs below
        const uint8_t* base = lds_f + (s & 1) * 65536 + lane * 16;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 ah[TA], al[TA];
#pragma unroll
            for (int a = 0; a < TA; ++a) {
                const uint32_t f = (uint32_t)(((TA * wo + a) * 2 + ks) * 2048);
                ah[a] = *(const u32x4*)(base + f);
                al[a] = *(const u32x4*)(base + f + 1024);
            }
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                if (FULL || TB * wi + b < ti_n) {
                    const uint32_t f = (uint32_t)(32768 + ((TB * wi + b) * 2 + ks) * 2048);
                    const bf16x8 Bhi = __builtin_bit_cast(bf16x8, *(const u32x4*)(base + f));
                    const bf16x8 Blo = __builtin_bit_cast(bf16x8, *(const u32x4*)(base + f + 1024));
#pragma unroll
                    for (int a = 0; a < TA; ++a) {
                        if (FULL || TA * wo + a < to_n) {
                            const bf16x8 Ahi = __builtin_bit_cast(bf16x8, ah[a]), Alo = __builtin_bit_cast(bf16x8, al[a]);
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Alo, Bhi, acc[a][b], 0, 0, 0);
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ahi, Blo, acc[a][b], 0, 0, 0);
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ahi, Bhi, acc[a][b], 0, 0, 0);
                        }
                    }
                }
            }
        }
        if (s + 1 < n_stages) publish((s + 1) & 1);          // that buffer was last read in stage s-1, behind the previous barrier
        __syncthreads();
    }
    if (BNZ && g.dbias) {      // d bias += bias_alpha * sum_m dZ[m, col]: the two half-waves hold the two row halves of the column
        const float tot = z_sum + __shfl_xor(z_sum, 32, 64);
        if (h == 0 && ok_o) atomicAdd(g.dbias + col_o, g.bias_alpha * tot);
    }
    if (g.partial) {
        float* out = g.partial + ((((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * gridDim.z + blockIdx.z) * 8 + wave) * (TA * TB * 1024) + lane;
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
            for (int b = 0; b < TB; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) out[((a * TB + b) * 16 + e) * 64] = acc[a][b][e];      // 256 contiguous bytes per instruction
        return;
    }
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b) {
            if (FULL || (TA * wo + a < to_n && TB * wi + b < ti_n)) {
                const int i = i_base + (TB * wi + b) * 32 + r;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int o = o_base + (TA * wo + a) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (FULL || (o < g.n_out && i < g.n_in)) atomicAdd(g.dW + (int64_t)o * g.ldw + i, g.alpha * acc[a][b][e]);
                }
            }
        }
}

// dW[o, i] += alpha * sum over the row blocks of their partial sums (one thread per element of a workgroup's block of dW, in register
// order: [wave][a < ta][b < tb][e][lane])
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int n_row_blocks, float* dW, int64_t ldw,
                                                           int n_out, int n_in, float alpha, int ta, int tb) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int lane = idx & 63, e = (idx >> 6) & 15, ab = (idx >> 10) % (ta * tb), wave = (idx >> 10) / (ta * tb), a = ab / tb, b = ab % tb;
    const int block = 8 * ta * tb * 1024;
    const int64_t blocks_yz = (int64_t)gridDim.y * gridDim.z, yz = (int64_t)blockIdx.y * gridDim.z + blockIdx.z;
    const float* p = partial + yz * block + idx;
    constexpr int U = 16;                                                  // loads in flight per thread (16 KiB per CU: enough for the full rate)
    const int64_t stride = blocks_yz * block;
    float s[U];
#pragma unroll
    for (int u = 0; u < U; ++u) s[u] = 0.f;
    int x = 0;
    for (; x + U <= n_row_blocks; x += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) s[u] += p[(int64_t)(x + u) * stride];
    }
    for (; x < n_row_blocks; ++x) s[0] += p[(int64_t)x * stride];
#pragma unroll
    for (int w = U / 2; w > 0; w >>= 1)
#pragma unroll
        for (int u = 0; u < w; ++u) s[u] += s[u + w];
    const int o = blockIdx.y * 128 * ta + (ta * (wave >> 1) + a) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
    const int i = blockIdx.z * 64 * tb + (tb * (wave & 1) + b) * 32 + (lane & 31);
    if (o < n_out && i < n_in) dW[(int64_t)o * ldw + i] += alpha * s[0];
}

// partial-sum scratch of the two-stage reduction: the calling trainer's (carved from its workspace at bind time, gemm_launch_context); for
// the stand-alone entry point snerf_linear_wgrad one block per stream (grown on demand, never freed: 64 MiB for a 256 x 256 layer on 256 CUs)
static float* wgrad_scratch(hipStream_t st, size_t floats) {
    if (tl_ctx.wgrad_partial && floats <= tl_ctx.wgrad_floats) return tl_ctx.wgrad_partial;
    struct Buf { float* p; size_t cap; };
    static std::mutex mu;
    static std::map<hipStream_t, Buf> bufs;
    std::lock_guard<std::mutex> lock(mu);
    Buf& b = bufs[st];
    if (floats > b.cap) {
        float* q = nullptr;
        if (hipMalloc(&q, floats * sizeof(float)) != hipSuccess) return nullptr;
        b.p = q; b.cap = floats;                                       // the old block stays alive for launches already queued
    }
    return b.p;
}

float* gemm_partial_scratch(hipStream_t st, size_t floats) { return wgrad_scratch(st, floats); }      // (train_kernels.hip: the thin heads' two-stage sums)

template <bool FULL, bool BNZ, int TA, int TB>
static hipError_t launch_wgrad_as(const WgradX& g, dim3 grid, hipStream_t st) {
    static bool done = false;
    auto k = wgrad_bf16x3_kernel<FULL, BNZ, TA, TB>;
    if (!done) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        if (e != hipSuccess) return e;
        done = true;
    }
    hipLaunchKernelGGL(k, grid, dim3(512), 131072, st, g);
    return hipGetLastError();
}

hipError_t launch_wgrad_bf16x3(float* dZ, int64_t ldz, const float* In, int64_t ldi, int64_t M, int n_out, int n_in, float alpha,
                               float* dW, int64_t ldw, hipStream_t st, const float* in_tab, int in_cols, const WgradBN* bn, int in_tab_stride) {
    if (M <= 0 || n_out <= 0 || n_in <= 0) return hipSuccess;
    if (ldz >= (1 << 24) || ldi >= (1 << 24)) return hipErrorInvalidValue;      // 32-bit lane offsets: 64 rows x ld
    WgradX g{};
    g.in_tab = in_tab; g.in_cols = in_tab ? in_cols : 0; g.in_tab_stride = in_tab_stride > 0 ? in_tab_stride : g.in_cols;
    g.dZ = dZ; g.In = In; g.dW = dW; g.M = M; g.ldz = ldz; g.ldi = ldi; g.ldw = ldw; g.n_out = n_out; g.n_in = n_in; g.alpha = alpha;
    // the workgroup's block of dW: 128 TA x 64 TB - the 128-wide layers get blocks of their own size (SNERF_WGRAD_SMALL=0: always 256 x 256)
    static const int small_blocks = [] { const char* e = getenv("SNERF_WGRAD_SMALL"); return (e && e[0] == '0') ? 0 : 1; }();
    const int ta = (small_blocks && n_out <= 128) ? 1 : 2, tb = (small_blocks && n_in <= 128) ? 2 : 4;
    const int bo = 128 * ta, bi = 64 * tb;
    const int by = (n_out + bo - 1) / bo, bz = (n_in + bi - 1) / bi;
    int64_t bx = ro_grid_blocks() / (by * bz);
    if (bx < 1) bx = 1;
    int64_t rows = (M + bx - 1) / bx;
    rows = (rows + WG_STAGE - 1) / WG_STAGE * WG_STAGE;
    if (rows < 4 * WG_STAGE) rows = 4 * WG_STAGE;
    bx = (M + rows - 1) / rows;
    g.rows_per_block = rows;
    g.reverse = stream_direction(M);
    const bool full = n_out % 256 == 0 && n_in % 256 == 0;
    const dim3 grid((unsigned)bx, by, bz);
    static const int two_stage = [] { const char* e = getenv("SNERF_WGRAD_ATOMIC"); return (e && e[0] == '1') ? 0 : 1; }();
    // (a few row blocks, or a thin layer whose blocks are mostly empty: the atomics are cheap enough)
    if (two_stage && bx >= 8 && 2 * (int64_t)(n_out < bo ? n_out : bo) * (n_in < bi ? n_in : bi) >= (int64_t)bo * bi) {
        g.partial = wgrad_scratch(st, (size_t)bx * by * bz * 8 * ta * tb * 1024);
        if (!g.partial) return hipErrorOutOfMemory;
    }
    if (bn) {
        if (bz != 1) return hipErrorInvalidValue;            // in-place dZ: every element must be gathered exactly once
        g.z = bn->z; g.ldzz = bn->ldz; g.bn_gamma = bn->gamma; g.bn_mu = bn->mu; g.bn_istd = bn->istd; g.bn_sdy = bn->sdy; g.bn_sdyx = bn->sdyx;
        g.bn_inv_m = bn->inv_m; g.bias_alpha = bn->bias_alpha; g.dbias = bn->dbias;
    }
    {
        static const int log_shapes = [] { const char* e = getenv("SNERF_LOG_WGRAD"); return (e && e[0] == '1') ? 1 : 0; }();      // SNERF_LOG_WGRAD=1: one line per launch (which layers end up on which block shape)
        if (log_shapes)
            fprintf(stderr, "wgrad M=%lld n_out=%d n_in=%d ldz=%lld ldi=%lld ta=%d tb=%d full=%d bn=%d in_tab=%d grid=(%lld,%d,%d) rows/block=%lld partial=%d\n", (long long)M, n_out, n_in,
                    (long long)ldz, (long long)ldi, ta, tb, (int)full, bn ? 1 : 0, in_tab ? in_cols : 0, (long long)bx, by, bz, (long long)rows, g.partial ? 1 : 0);
    }
    hipError_t e;
    if (full) e = bn ? launch_wgrad_as<true, true, 2, 4>(g, grid, st) : launch_wgrad_as<true, false, 2, 4>(g, grid, st);
    else if (ta == 2 && tb == 4) e = bn ? launch_wgrad_as<false, true, 2, 4>(g, grid, st) : launch_wgrad_as<false, false, 2, 4>(g, grid, st);
    else if (ta == 1 && tb == 4) e = bn ? launch_wgrad_as<false, true, 1, 4>(g, grid, st) : launch_wgrad_as<false, false, 1, 4>(g, grid, st);
    else if (ta == 2 && tb == 2) e = bn ? launch_wgrad_as<false, true, 2, 2>(g, grid, st) : launch_wgrad_as<false, false, 2, 2>(g, grid, st);
    else e = bn ? launch_wgrad_as<false, true, 1, 2>(g, grid, st) : launch_wgrad_as<false, false, 1, 2>(g, grid, st);
    if (e != hipSuccess) return e;
    if (g.partial)
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(8 * ta * tb * 4, by, bz), dim3(256), 0, st, g.partial, (int)bx, dW, ldw, n_out, n_in, alpha, ta, tb);
    return hipGetLastError();
}

hipError_t launch_split_weights(const float* W, int rows, int cols, bool transpose, uint16_t* frag, int n_tiles, int ksteps, hipStream_t st) {
    const int64_t total = (int64_t)n_tiles * ksteps * 512;
    if (total <= 0) return hipSuccess;
    int64_t b = (total + 255) / 256;
    if (b > 4096) b = 4096;
    hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)b), dim3(256), 0, st, W, rows, cols, transpose ? 1 : 0, frag, n_tiles, ksteps);
    return hipGetLastError();
}

// a device array of zeros (grown on demand, never freed: a few KiB), for optional per-column inputs
static const float* zeros_dev(int64_t n) {
    static float* p = nullptr;
    static int64_t cap = 0;
    if (n > cap) {
        int64_t want = n < 4096 ? 4096 : n;
        float* q = nullptr;
        if (hipMalloc(&q, want * sizeof(float)) != hipSuccess) return nullptr;
        if (hipMemset(q, 0, want * sizeof(float)) != hipSuccess) return nullptr;
        p = q; cap = want;                                         // the old block stays alive for launches already queued
    }
    return p;
}

template <int NT, int PF>
static hipError_t launch_full(const GemmX& gx, int aol_mode, int act_mode, dim3 grid, size_t lds, hipStream_t st) {
#define SNERF_GO(A_, C_)                                                                                              \
    do {                                                                                                              \
        static bool done = false;                                                                                     \
        auto k = gemm_rows_full_kernel<NT, PF, A_, C_>;                                                               \
        if (!done) {                                                                                                  \
            hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            if (e != hipSuccess) return e;                                                                            \
            done = true;                                                                                              \
        }                                                                                                             \
        hipLaunchKernelGGL(k, grid, dim3(512), lds, st, gx);                                                          \
    } while (0)
    if (act_mode == 1) {
        if constexpr (NT > 1) SNERF_GO(0, 1);
        else return hipErrorInvalidValue;
    }
    else if (aol_mode == 1) SNERF_GO(1, 0);
    else SNERF_GO(0, 0);
#undef SNERF_GO
    return hipGetLastError();
}

// n-tiles whose weights (2 KiB per tile and 16-k step) fit the 160 KiB LDS beside an activation-on-load table (8 bytes per input column):
// four up to K = 320, two up to K = 608 (W = 512's [fc4 | PE] layer has 36 k-steps)
int gemm_rows_group_tiles(int ksteps) { return ksteps <= 20 ? 4 : (ksteps <= 38 ? 2 : 0); }

hipError_t launch_gemm_bf16x3(const GemmX& g, hipStream_t st) {
    if (g.M <= 0 || g.N <= 0) return hipSuccess;
    const int nt = gemm_rows_group_tiles(g.ksteps);
    if (!nt) return hipErrorInvalidValue;
    const size_t lds = (size_t)nt * g.ksteps * 2048;
    const int groups = (g.n_tiles + nt - 1) / nt;
    int blocks = ro_grid_blocks();
    if (blocks / 8 < groups) blocks = groups * 8;            // at least one worker per XCD
    static bool attr_done = false;
    if (!attr_done) {
        const void* fns[6] = {(const void*)gemm_rows_kernel<4, false, false>, (const void*)gemm_rows_kernel<4, true, false>,
                              (const void*)gemm_rows_kernel<4, false, true>, (const void*)gemm_rows_kernel<2, false, false>,
                              (const void*)gemm_rows_kernel<2, true, false>, (const void*)gemm_rows_kernel<2, false, true>};
        for (const void* f : fns) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
        }
        attr_done = true;
    }
    const bool aol = g.act_tab != nullptr && g.act_cols > 0;
    const bool act = g.ez != nullptr;
    if (aol && (g.act_cols % 8 != 0 || g.act_cols > g.K || (uintptr_t)g.act_tab % 16 != 0)) return hipErrorInvalidValue;
    if (act && (aol || !g.stats || !g.etab)) return hipErrorInvalidValue;
    GemmX gx = g;
    gx.tab_lds = 0;
    size_t lds_total = lds;
    if (aol && lds + (size_t)g.act_cols * 8 <= 160 * 1024) { gx.tab_lds = 1; lds_total = lds + (size_t)g.act_cols * 8; }
    const dim3 grid(blocks), block(512);
    // raw weights: split them here, in the fragment order of the kernel that runs
    auto split32 = [&]() -> hipError_t {
        return g.W ? launch_split_weights(g.W, g.w_rows, g.w_cols, g.w_transpose != 0, const_cast<uint16_t*>(g.frag), g.n_tiles, g.ksteps, st) : hipSuccess;
    };
    {   // the reference's default width (512-wide layers): accumulators in AGPRs, A and weights streamed (gemm_areg.hip).  SNERF_GEMM_AREG=0 off, =2 every
        // shape the kernel takes (also the 256-wide layers of a W = 256 network, where the column-group kernels are HBM-bound already)
        static const int areg_mode = [] { const char* e = getenv("SNERF_GEMM_AREG"); return e ? atoi(e) : 1; }();
        static const int areg_act = [] { const char* e = getenv("SNERF_GEMM_AREG_ACT"); return e ? atoi(e) : 1; }();      // the activation-backward form on it: SNERF_GEMM_AREG_ACT=0 keeps the column-group kernel for those
        if (areg_mode && (!act || areg_act) && g.W && (g.N == 512 || g.K > 256 || areg_mode == 2) && gemm_areg_ok(g)) {
            hipError_t e = launch_areg_split_weights(g.W, g.w_rows, g.w_cols, g.w_transpose != 0, const_cast<uint16_t*>(g.frag), g.n_tiles, g.ksteps, st);
            if (e != hipSuccess) return e;
            return launch_gemm_areg(gx, st);
        }
    }
    {   // the pipelined full-tile kernel wherever the shape allows it (every per-point layer of the training step)
        static const int full_mode = [] { const char* e = getenv("SNERF_GEMM_FULL"); return (e && e[0] == '0') ? 0 : 1; }();
        static const int pf_force = [] { const char* e = getenv("SNERF_GEMM_PF"); return e ? atoi(e) : 0; }();
        static const int mode16 = [] { const char* e = getenv("SNERF_GEMM16"); return (e && e[0] == '0') ? 0 : 1; }();
        const int KS = g.ksteps;
        const bool k_ok = g.K % 16 == 0 || (g.a_padded && g.lda >= (int64_t)KS * 16);
        const bool a_vec = ((uintptr_t)g.A % 16 == 0) && (g.lda % 4 == 0);
        int pf = KS % 8 == 0 ? 8 : (KS % 4 == 0 ? 4 : (KS % 2 == 0 ? 2 : 0));
        if (act && pf > 4) pf = 4;                               // the activation-backward epilogue needs the registers
        if (pf_force && pf_force <= pf && KS % pf_force == 0 && (pf_force == 8 || pf_force == 4 || pf_force == 2)) pf = pf_force;
        // activation on load wants its table in LDS: with 20 k-steps four n-tiles fill the 160 KiB, so that layer runs two per group
        int ntf = nt;
        if (aol && ntf == 4 && (size_t)4 * KS * 2048 + (size_t)g.act_cols * 8 > 160 * 1024) ntf = 2;
        if (g.n_tiles == 1 && !act) ntf = 1;                      // thin heads (1..32 outputs)
        const size_t lds_f = (size_t)ntf * KS * 2048 + (aol ? (size_t)g.act_cols * 8 : 0);
        if (full_mode && k_ok && a_vec && pf && g.n_tiles % ntf == 0 && (g.N == (int64_t)g.n_tiles * 32 || ntf == 1) && (!aol || g.act_cols % 16 == 0) &&
            !g.accumulate && lds_f <= 160 * 1024 && g.M * g.ldc < (1ll << 29) && (!act || g.M * g.eld < (1ll << 29))) {      // 32-bit byte offsets
            const int groups_f = g.n_tiles / ntf;
            int blocks_f = ro_grid_blocks();
            if (blocks_f / 8 < groups_f) blocks_f = groups_f * 8;
            const dim3 grid_f(blocks_f);
            gx.tab_lds = aol ? 1 : 0;
            const int aol_mode = aol ? 1 : 0, act_mode = act ? 1 : 0;
            if (act && !gx.emu) {                                  // a layer without BatchNorm: xhat sums are defined as 0
                gx.emu = zeros_dev(g.N);
                gx.eistd = gx.emu;
                if (!gx.emu) return hipErrorOutOfMemory;
            }
            // the 16x16x32 form (quad-coalesced A loads): raw weights at hand, K in whole 32-k steps, wide layers only (the thin
            // heads and the 64-column groups of the K = 320 layer keep the 32x32x16 form)
            const bool k32 = KS % 2 == 0 && (g.K % 32 == 0 || (g.a_padded && g.lda >= (int64_t)KS * 16));
            if (mode16 && g.W && k32 && ntf == 4 && (!aol || g.act_cols % 32 == 0) && lds_f + 128 * 4 * 5 <= 160 * 1024) {
                return launch_gemm_rows16(gx, aol_mode, act_mode, grid_f, lds_f + 128 * 4 * (act_mode ? 5 : 1), st);      // + per-column constants
            }
            // the K = 320 layer ([fc4 | PE]: 64-column groups, four of them read every A row): its forward on the same kernel, whose quad-coalesced
            // A loads cost the vector-memory path half of what the lane-per-row operand layout of the 32x32x16 form does
            static const int mode16_k320 = [] { const char* e = getenv("SNERF_GEMM16_K320"); return (e && e[0] == '0') ? 0 : 1; }();
            if (mode16 && mode16_k320 && g.W && k32 && ntf == 2 && !act && g.n_tiles % 2 == 0 && (!aol || g.act_cols % 32 == 0) && lds_f + 128 * 4 <= 160 * 1024) {
                return launch_gemm_rows16(gx, aol_mode, 0, grid_f, lds_f + 128 * 4, st, 4);
            }
            {
                hipError_t e = split32();
                if (e != hipSuccess) return e;
            }
            if (ntf == 1)
                return pf == 8 ? launch_full<1, 8>(gx, aol_mode, 0, grid_f, lds_f, st)
                               : pf == 4 ? launch_full<1, 4>(gx, aol_mode, 0, grid_f, lds_f, st) : launch_full<1, 2>(gx, aol_mode, 0, grid_f, lds_f, st);
            return ntf == 4 ? (pf == 8 ? launch_full<4, 8>(gx, aol_mode, act_mode, grid_f, lds_f, st)
                                       : pf == 4 ? launch_full<4, 4>(gx, aol_mode, act_mode, grid_f, lds_f, st)
                                                 : launch_full<4, 2>(gx, aol_mode, act_mode, grid_f, lds_f, st))
                            : (pf == 8 ? launch_full<2, 8>(gx, aol_mode, act_mode, grid_f, lds_f, st)
                                       : pf == 4 ? launch_full<2, 4>(gx, aol_mode, act_mode, grid_f, lds_f, st)
                                                 : launch_full<2, 2>(gx, aol_mode, act_mode, grid_f, lds_f, st));
        }
    }
    {
        hipError_t e = split32();
        if (e != hipSuccess) return e;
    }
    if (nt == 4) {
        if (act) hipLaunchKernelGGL((gemm_rows_kernel<4, false, true>), grid, block, lds_total, st, gx);
        else if (aol) hipLaunchKernelGGL((gemm_rows_kernel<4, true, false>), grid, block, lds_total, st, gx);
        else hipLaunchKernelGGL((gemm_rows_kernel<4, false, false>), grid, block, lds_total, st, gx);
    } else {
        if (act) hipLaunchKernelGGL((gemm_rows_kernel<2, false, true>), grid, block, lds_total, st, gx);
        else if (aol) hipLaunchKernelGGL((gemm_rows_kernel<2, true, false>), grid, block, lds_total, st, gx);
        else hipLaunchKernelGGL((gemm_rows_kernel<2, false, false>), grid, block, lds_total, st, gx);
    }
    return hipGetLastError();
}

}  // namespace snerf
