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

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

constexpr int RO_WAVES = 8, RO_MT = 1;                       // 8 waves x 32 rows = 256 rows per workgroup tile (64 accumulators per lane)
constexpr int RO_ROWS = RO_WAVES * RO_MT * 32;
constexpr int RO_PF = 4;                                     // k-steps of A loads in flight ahead of the MFMAs

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

// =====================================================================================================
// bf16x3 weight-gradient GEMM:   dW[o, i] += alpha * sum_m dZ[m, o] * In[m, i]           (K = #points, split over workgroups)
// Both operands are point-major, so the 8 consecutive-k values an MFMA lane needs are 8 rows of one column: each lane gathers
// them with 8 row-coalesced dword loads (32 lanes = 128 contiguous bytes of a row), splits them into bf16 hi/lo in registers
// and publishes the finished 1 KiB fragments through LDS, where the 8 waves of the workgroup share them: the workgroup
// holds the whole dW block (up to 256 x 256, 128 accumulator registers per lane), so dZ and In are read from HBM exactly once.
// Stage = 32 points; double-buffered LDS (2 x 64 KiB), one barrier per stage; the loads of stage s+1 fly during the MFMAs of s.
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
};

constexpr int WG_STAGE = 32;

template <bool FULL, bool BNZ>  // FULL: every 256 x 256 block of dW is complete (no tile or column masks in the hot loop)
__global__ __launch_bounds__(512) void wgrad_bf16x3_kernel(const WgradX g) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_f[];          // [2 buffers][2 operands][8 tiles][2 ksteps][hi,lo][1 KiB]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int o_base = blockIdx.y * 256, i_base = blockIdx.z * 256;
    const int to_n = FULL ? 8 : ((g.n_out - o_base + 31) / 32 < 8 ? (g.n_out - o_base + 31) / 32 : 8);      // valid 32-column tiles of each operand
    const int ti_n = FULL ? 8 : ((g.n_in - i_base + 31) / 32 < 8 ? (g.n_in - i_base + 31) / 32 : 8);
    const int wo = wave >> 1, wi = wave & 1;                 // this wave's dW piece: n_out tiles {2wo, 2wo+1} x n_in tiles {4wi .. 4wi+3}
    const int64_t m_begin = (int64_t)blockIdx.x * g.rows_per_block;
    const int64_t m_end = m_begin + g.rows_per_block < g.M ? m_begin + g.rows_per_block : g.M;
    if (m_begin >= m_end) return;
    const int n_stages = (int)((m_end - m_begin + WG_STAGE - 1) / WG_STAGE);

    // producer role: wave w gathers tile w of dZ and tile w of In (both k-steps of the stage)
    const bool make_o = FULL || wave < to_n, make_i = FULL || wave < ti_n;
    const int col_o = o_base + wave * 32 + r, col_i = i_base + wave * 32 + r;
    const bool ok_o = FULL || (make_o && col_o < g.n_out), ok_i = FULL || (make_i && col_i < g.n_in);
    // branch-free gathers: a wave-uniform 64-bit stage base plus a 32-bit lane offset (row clamped to the last valid row of this
    // workgroup's range); out-of-range values are zeroed at publish time so that nothing depends on the loads before then
    const uint32_t ldz = (uint32_t)g.ldz, ldi = (uint32_t)g.ldi;
    const uint32_t co = ok_o ? (uint32_t)col_o : 0u, ci = ok_i ? (uint32_t)col_i : 0u;
    const int64_t m_last = m_end - 1;
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
    const float c_a = in_act ? g.in_tab[col_i] : 0.f, c_b = in_act ? g.in_tab[g.in_cols + col_i] : 0.f;
    auto gather = [&](int stage) {
        const int64_t ms = m_begin + (int64_t)stage * WG_STAGE;                  // uniform
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
                vo[ks][e] = bo[kr * ldz + co];          // raw: masking waits for publish, so nothing here depends on the
                vi[ks][e] = bi[kr * ldi + ci];          // loads and they stay in flight across the MFMA block
                if (BNZ) vz[ks][e] = bz[kr * ldzz + co];
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
            const uint32_t f = (uint32_t)((wave * 2 + ks) * 2048 + lane * 16);
            if (make_o) { *(u32x4*)(base + f) = oh; *(u32x4*)(base + f + 1024) = ol; }
            if (make_i) { *(u32x4*)(base + 32768 + f) = ih; *(u32x4*)(base + 32768 + f + 1024) = il; }
        }
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
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
            u32x4 ah[2], al[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const uint32_t f = (uint32_t)(((2 * wo + a) * 2 + ks) * 2048);
                ah[a] = *(const u32x4*)(base + f);
                al[a] = *(const u32x4*)(base + f + 1024);
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (FULL || 4 * wi + b < ti_n) {
                    const uint32_t f = (uint32_t)(32768 + ((4 * wi + b) * 2 + ks) * 2048);
                    const bf16x8 Bhi = __builtin_bit_cast(bf16x8, *(const u32x4*)(base + f));
                    const bf16x8 Blo = __builtin_bit_cast(bf16x8, *(const u32x4*)(base + f + 1024));
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        if (FULL || 2 * wo + a < to_n) {
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
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (FULL || (2 * wo + a < to_n && 4 * wi + b < ti_n)) {
                const int i = i_base + (4 * wi + b) * 32 + r;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int o = o_base + (2 * wo + a) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (FULL || (o < g.n_out && i < g.n_in)) atomicAdd(g.dW + (int64_t)o * g.ldw + i, g.alpha * acc[a][b][e]);
                }
            }
        }
}

hipError_t launch_wgrad_bf16x3(float* dZ, int64_t ldz, const float* In, int64_t ldi, int64_t M, int n_out, int n_in, float alpha,
                               float* dW, int64_t ldw, hipStream_t st, const float* in_tab, int in_cols, const WgradBN* bn) {
    if (M <= 0 || n_out <= 0 || n_in <= 0) return hipSuccess;
    if (ldz >= (1 << 24) || ldi >= (1 << 24)) return hipErrorInvalidValue;      // 32-bit lane offsets: 64 rows x ld
    static bool attr_done = false;
    if (!attr_done) {
        const void* fns[4] = {(const void*)wgrad_bf16x3_kernel<true, false>, (const void*)wgrad_bf16x3_kernel<false, false>,
                              (const void*)wgrad_bf16x3_kernel<true, true>, (const void*)wgrad_bf16x3_kernel<false, true>};
        for (const void* f : fns) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            if (e != hipSuccess) return e;
        }
        attr_done = true;
    }
    WgradX g{};
    g.in_tab = in_tab; g.in_cols = in_tab ? in_cols : 0;
    g.dZ = dZ; g.In = In; g.dW = dW; g.M = M; g.ldz = ldz; g.ldi = ldi; g.ldw = ldw; g.n_out = n_out; g.n_in = n_in; g.alpha = alpha;
    const int by = (n_out + 255) / 256, bz = (n_in + 255) / 256;
    int64_t bx = ro_grid_blocks() / (by * bz);
    if (bx < 1) bx = 1;
    int64_t rows = (M + bx - 1) / bx;
    rows = (rows + WG_STAGE - 1) / WG_STAGE * WG_STAGE;
    if (rows < 4 * WG_STAGE) rows = 4 * WG_STAGE;
    bx = (M + rows - 1) / rows;
    g.rows_per_block = rows;
    const bool full = n_out % 256 == 0 && n_in % 256 == 0;
    const dim3 grid((unsigned)bx, by, bz), block(512);
    if (bn) {
        if (bz != 1) return hipErrorInvalidValue;            // in-place dZ: every element must be gathered exactly once
        g.z = bn->z; g.ldzz = bn->ldz; g.bn_gamma = bn->gamma; g.bn_mu = bn->mu; g.bn_istd = bn->istd; g.bn_sdy = bn->sdy; g.bn_sdyx = bn->sdyx;
        g.bn_inv_m = bn->inv_m; g.bias_alpha = bn->bias_alpha; g.dbias = bn->dbias;
        if (full) hipLaunchKernelGGL((wgrad_bf16x3_kernel<true, true>), grid, block, 131072, st, g);
        else hipLaunchKernelGGL((wgrad_bf16x3_kernel<false, true>), grid, block, 131072, st, g);
    } else {
        if (full) hipLaunchKernelGGL((wgrad_bf16x3_kernel<true, false>), grid, block, 131072, st, g);
        else hipLaunchKernelGGL((wgrad_bf16x3_kernel<false, false>), grid, block, 131072, st, g);
    }
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

int gemm_rows_group_tiles(int ksteps) { return ksteps <= 20 ? 4 : (ksteps <= 32 ? 2 : 0); }     // n-tiles whose weights fit the 160 KiB LDS

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
