// Full-tile row GEMM on v_mfma_f32_16x16x32_bf16 (its own translation unit: see the comment at the kernel).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "gemm_common.h"
#include "train.h"

static int ro_blocks16() {
    static int n = 0;
    if (!n) {
        hipDeviceProp_t p;
        int dev = 0;
        n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ? p.multiProcessorCount : 256;
        if (n < 1) n = 256;
    }
    return n;
}

#ifndef SNERF_STORE_AUX
#define SNERF_STORE_AUX 0  // cache policy of the epilogue stores (buffer instruction aux bits: 1 sc0, 2 nt, 16 sc1)
#endif
#ifndef SNERF_ABLW
#define SNERF_ABLW 0       // the same for gemm_wreg_kernel: 1 no MFMAs, 2 no stores, 4 no LDS-DMA, 8 no produce step
#endif
#ifndef SNERF_ABL16
#define SNERF_ABL16 0      // timing-only ablations of scratch builds (tools/variants.py): 1 no MFMAs, 2 no stores, 4 no A refills, 8 no sin / split, 16 no LDS weight reads
#endif

namespace snerf {

// ---------------------------------------------------------------------------------------------------------------------
// The same full-tile row GEMM on v_mfma_f32_16x16x32_bf16.  Why a second MFMA shape: the vector-memory path, not HBM or the
// matrix pipe, bounds the 32x32x16 form.  Its A operand puts one ROW on every lane of a half-wave - a 1 KiB load
// instruction touches 64 different 128-B lines (147 cycles per instruction and CU, tools/probes/ta_rate.hip) - and every A
// byte is loaded by two column groups.  The 16x16x32 A operand has 16 rows x 4 lanes: with the k order chosen below the four
// lanes of a row read 64 contiguous bytes per instruction (quad-coalesced: 67 cycles), from the SAME row-major activations.
// The accumulator of a 16x16 tile (lane (g, j): column j, rows 4g .. 4g+3) stores as four 64-B row segments per instruction,
// at the per-byte rate of the 32x32 form's two 128-B segments (17 against 16 cycles per 256 B).  Arithmetic, summation order
// inside a product (hi*hi last) and results differ from the 32x32x16 kernel only by the order of the k terms inside a 32-k step.
//   fragment order (split_weights16_kernel): n-tile T (16 columns), k-step ks (32 k): 1 KiB hi then 1 KiB lo; lane (g, j) owns
//   16 bytes = bf16 of Bt[16 T + j][32 ks + kmap(g, e)], e = 0..7, kmap(g, e) = 4 g + e (e < 4), 16 + 4 g + (e - 4) (e >= 4) -
//   so a lane's A values are two 16-byte loads, at byte 16 g and byte 64 + 16 g of the 128-B k-step of its row.
// A wave owns 32 rows = two 16-row tiles (each weight fragment read from LDS serves both); 8 waves = 256 rows per workgroup tile.
__global__ void split_weights16_kernel(const float* W, int rows, int cols, int transpose, uint16_t* frag, int n_tiles16, int ksteps32) {
    const int64_t total = (int64_t)n_tiles16 * ksteps32 * 512;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const int64_t tk = i >> 9;
        const int ks = (int)(tk % ksteps32), T = (int)(tk / ksteps32);
        const int g = lane >> 4, n = T * 16 + (lane & 15), k = ks * 32 + (e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4));
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

__device__ __forceinline__ void a16_issue(const float* p, f32x4& x, f32x4& y) {       // k = 4g .. 4g+3 and 16+4g .. 16+4g+3 of a 32-k step
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:64" : "=&v"(x), "=&v"(y) : "v"(p));
}
template <int N>
__device__ __forceinline__ void a16_wait(f32x4& x0, f32x4& y0, f32x4& x1, f32x4& y1) {
    asm volatile("s_waitcnt vmcnt(%4) ; a16_wait %0 %1 %2 %3" : "+v"(x0), "+v"(y0), "+v"(x1), "+v"(y1) : "n"(N));
}
template <int N>
__device__ __forceinline__ void a16_wait(f32x4& x0, f32x4& y0) {
    asm volatile("s_waitcnt vmcnt(%2) ; a16_wait %0 %1" : "+v"(x0), "+v"(y0) : "n"(N));
}
#ifndef SNERF_R16_RT
#define SNERF_R16_RT 2      // 16-row tiles per wave: 2 = 8 waves x 32 rows (two waves per SIMD), 1 = 16 waves x 16 rows (four per SIMD)
#endif
constexpr int R16_RT = SNERF_R16_RT, R16_WAVES = RO_ROWS / (16 * R16_RT);
template <int N>
__device__ __forceinline__ void a16_wait_slot(f32x4 (&x)[R16_RT], f32x4 (&y)[R16_RT]) {
    if constexpr (R16_RT == 2) a16_wait<N>(x[0], y[0], x[1], y[1]);
    else a16_wait<N>(x[0], y[0]);
}

// NT: 16-column n-tiles per group (8 = 128 columns).  PF: 32-k steps of A in flight.  AOL / ACT as in gemm_rows_full_kernel.
template <int NT, int PF, int AOL, int ACT>
__global__ __launch_bounds__(64 * R16_WAVES) void gemm_rows16_kernel(const GemmX g) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_w[];
#ifdef SNERF_STAMP16
    const uint64_t stamp_entry = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int jj = lane & 15, gq = lane >> 4;
    const int KS = g.ksteps >> 1;                                   // 32-k steps (multiple of PF)
    const int n_groups = (2 * g.n_tiles) / NT;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int workers_per_xcd = slots / n_groups;
    if (slot >= workers_per_xcd * n_groups) return;
    const int grp = slot % n_groups, worker = (slot / n_groups) * 8 + xcd, n_workers = workers_per_xcd * 8;

    {
        const u32x4* src = (const u32x4*)(g.frag + (int64_t)grp * NT * KS * 1024);
        u32x4* dst = (u32x4*)lds_w;
        const int n16 = NT * KS * 128;
        int i0 = tid;
        constexpr int NTH = 64 * R16_WAVES;
        for (; i0 + 7 * NTH < n16; i0 += NTH * 8) {
            u32x4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = src[i0 + q * NTH];
#pragma unroll
            for (int q = 0; q < 8; ++q) dst[i0 + q * NTH] = v[q];
        }
        for (; i0 < n16; i0 += NTH) dst[i0] = src[i0];
    }
    const uint8_t* lds_tab = lds_w + (size_t)NT * KS * 2048;
    if (AOL == 1) {
        float* dst = (float*)lds_tab;
        for (int i = tid; i < 2 * g.act_cols; i += 64 * R16_WAVES) dst[i] = g.act_tab[i];
    }
    // per-column constants of this group's NT * 16 columns, behind the table: the epilogue reads them from LDS - a global load there
    // would make hipcc wait for vmcnt(0), i.e. for the previous tile's stores and every prefetched operand of the next one
    float* lds_col = (float*)(lds_tab + (AOL ? (size_t)g.act_cols * 8 : 0));      // [bias | etab a | etab b | mu | istd][NT * 16]
    for (int i = tid; i < NT * 16; i += 64 * R16_WAVES) {
        const int64_t n = (int64_t)grp * NT * 16 + i;
        const bool in = n < g.N;
        lds_col[i] = (!ACT && g.bias && in) ? g.bias[n] : 0.f;
        if (ACT) {
            lds_col[NT * 16 + i] = in ? g.etab[n] : 0.f;
            lds_col[2 * NT * 16 + i] = in ? g.etab[g.N + n] : 0.f;
            lds_col[3 * NT * 16 + i] = in ? g.emu[n] : 0.f;
            lds_col[4 * NT * 16 + i] = in ? g.eistd[n] : 0.f;
        }
    }
    const int col0 = grp * NT * 16 + jj;                            // this lane's column of n-tile 0
    __syncthreads();

    const int64_t n_row_tiles = (g.M + RO_ROWS - 1) / RO_ROWS;
    float st1[NT], st2[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) st1[j] = st2[j] = 0.f;
#ifdef SNERF_STAMP16      // diagnostic build: shader clock held inside the tile loop = d(s_memtime) / d(s_memrealtime) x 100 MHz
    const uint64_t stamp_c0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif

    auto a_ptr = [&](int64_t rt, int half) {
        int64_t m = (g.reverse ? n_row_tiles - 1 - rt : rt) * RO_ROWS + wave * (16 * R16_RT) + half * 16 + jj;
        m = m < g.M ? m : g.M - 1;                                  // loads stay in bounds, stores are masked
        return g.A + m * g.lda + gq * 4;
    };
    // epilogue addressing through buffer instructions (see gemm_rows_full_kernel): lane offset + scalar row offset + immediate
    const bool nok0 = NT > 1 || col0 < g.N;                       // thin head (N < 16): out-of-range lanes store nowhere
    const int lc = nok0 ? (int)(4 * gq * g.ldc + col0) * 4 : (int)0x80000000;
    const int lz = ACT ? (int)(4 * gq * g.eld + col0) * 4 : 0;
    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc((void*)g.C, 0, (int)(g.M * g.ldc * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc((void*)(ACT ? g.ez : g.A), 0, -1, 0x00020000);
    int64_t rt = worker;
    const float* arow[R16_RT];
#pragma unroll
    for (int h = 0; h < R16_RT; ++h) arow[h] = a_ptr(rt < n_row_tiles ? rt : n_row_tiles - 1, h);
    f32x4 px[PF][R16_RT], py[PF][R16_RT];
#pragma unroll
    for (int d = 0; d < PF; ++d)
#pragma unroll
        for (int h = 0; h < R16_RT; ++h) a16_issue(arow[h] + d * 32, px[d][h], py[d][h]);

#ifdef SNERF_PHASE16       // diagnostic build: where one wave's time goes (shader cycles: load wait | convert + refill | LDS + MFMA issue | epilogue)
#define SNERF_PH(x) __builtin_amdgcn_sched_barrier(0); const uint64_t x = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0)
    uint64_t ph_wait = 0, ph_conv = 0, ph_mfma = 0, ph_epi = 0, ph_steps = 0;
    const uint64_t ph_t0 = __builtin_amdgcn_s_memtime();
#else
#define SNERF_PH(x)
#endif
    for (; rt < n_row_tiles; rt += n_workers) {
        const int64_t rn = rt + n_workers;
        const float* anext[R16_RT];
#pragma unroll
        for (int h = 0; h < R16_RT; ++h) anext[h] = a_ptr(rn < n_row_tiles ? rn : rt, h);
        f32x4 acc[R16_RT][NT];
#pragma unroll
        for (int h = 0; h < R16_RT; ++h)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[h][j][e] = 0.f;
        for (int ks0 = 0; ks0 < KS; ks0 += PF) {
            const bool last = ks0 + PF >= KS;
            const float* srcs[R16_RT];
#pragma unroll
            for (int h = 0; h < R16_RT; ++h) srcs[h] = last ? anext[h] : arow[h] + (ks0 + PF) * 32;
#pragma unroll
            for (int d = 0; d < PF; ++d) {
                const int ks = ks0 + d;
                f32x4 ta[2], tb[2];                                  // AOL: [a | b] of this lane's 8 k values (shared by both row tiles)
                if (AOL) {
                    int k0 = ks * 32 + gq * 4;
                    k0 = k0 + 20 <= g.act_cols ? k0 : 0;           // clamped: the loads are unconditional (act_cols is a multiple of 32 here)
                    const float* tp = (const float*)lds_tab + k0;
                    ta[0] = *(const f32x4*)tp; ta[1] = *(const f32x4*)(tp + 16);
                    tb[0] = *(const f32x4*)(tp + g.act_cols); tb[1] = *(const f32x4*)(tp + g.act_cols + 16);
                }
                SNERF_PH(q0);
#if !(SNERF_ABL16 & 4)
                a16_wait_slot<2 * R16_RT * (PF - 1)>(px[d], py[d]);                  // the PF-1 younger k-steps stay in flight
#endif
                SNERF_PH(q1);
                u32x4 ahi[R16_RT], alo[R16_RT];
#pragma unroll
                for (int h = 0; h < R16_RT; ++h) {
                    float a8[8] = {px[d][h][0], px[d][h][1], px[d][h][2], px[d][h][3], py[d][h][0], py[d][h][1], py[d][h][2], py[d][h][3]};
#if SNERF_ABL16 & 8
                    for (int q = 0; q < 4; ++q) { ahi[h][q] = __builtin_bit_cast(uint32_t, a8[2 * q]); alo[h][q] = __builtin_bit_cast(uint32_t, a8[2 * q + 1]); }
                    continue;
#endif
                    if (AOL) {
                        if (ks * 32 < g.act_cols) {                 // uniform
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                a8[e] = __builtin_amdgcn_sinf(__builtin_fmaf(ta[e >> 2][e & 3], a8[e], tb[e >> 2][e & 3]));
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        uint32_t hh, ll;
                        split2_bf16(a8[2 * q], a8[2 * q + 1], hh, ll);
                        ahi[h][q] = hh;
                        alo[h][q] = ll;
                    }
                }
#if !(SNERF_ABL16 & 4)
#pragma unroll
                for (int h = 0; h < R16_RT; ++h) a16_issue(srcs[h] + d * 32, px[d][h], py[d][h]);      // refill the slot just consumed (next tile's on the last round)
#endif
                __builtin_amdgcn_sched_barrier(0);
                SNERF_PH(q2);
                bf16x8 Ahi[R16_RT], Alo[R16_RT];
#pragma unroll
                for (int h = 0; h < R16_RT; ++h) { Ahi[h] = __builtin_bit_cast(bf16x8, ahi[h]); Alo[h] = __builtin_bit_cast(bf16x8, alo[h]); }
                const uint32_t base = (uint32_t)ks * 2048u + (uint32_t)lane * 16u;
                constexpr int JB = NT < 4 ? NT : 4;                  // weight fragments of four n-tiles in registers at a time
#pragma unroll
                for (int j0 = 0; j0 < NT; j0 += JB) {
                    bf16x8 Bhi[JB], Blo[JB];
#pragma unroll
                    for (int j = 0; j < JB; ++j) {
#if SNERF_ABL16 & 16
                        Bhi[j] = Ahi[0]; Blo[j] = Alo[R16_RT - 1];
#else
                        Bhi[j] = __builtin_bit_cast(bf16x8, *(const u32x4*)(lds_w + base + (uint32_t)(j0 + j) * KS * 2048u));
                        Blo[j] = __builtin_bit_cast(bf16x8, *(const u32x4*)(lds_w + base + (uint32_t)(j0 + j) * KS * 2048u + 1024u));
#endif
                    }
#if SNERF_ABL16 & 1
#pragma unroll
                    for (int j = 0; j < JB; ++j) {      // keep every operand alive with one cheap VALU op per accumulator register group
#pragma unroll
                        for (int h = 0; h < R16_RT; ++h)
                            acc[h][j0 + j][0] += __builtin_bit_cast(float, (h ? __builtin_bit_cast(u32x4, Blo[j])[0] : __builtin_bit_cast(u32x4, Bhi[j])[0]) ^ ahi[h][0] ^ alo[h][1]);
                    }
                    continue;
#endif
#pragma unroll
                    for (int j = 0; j < JB; ++j) {
                        acc[0][j0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Alo[0], Bhi[j], acc[0][j0 + j], 0, 0, 0);
                        if constexpr (R16_RT == 2) acc[R16_RT - 1][j0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Alo[R16_RT - 1], Bhi[j], acc[R16_RT - 1][j0 + j], 0, 0, 0);
                    }
#pragma unroll
                    for (int j = 0; j < JB; ++j) {
                        acc[0][j0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ahi[0], Blo[j], acc[0][j0 + j], 0, 0, 0);
                        if constexpr (R16_RT == 2) acc[R16_RT - 1][j0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ahi[R16_RT - 1], Blo[j], acc[R16_RT - 1][j0 + j], 0, 0, 0);
                    }
#pragma unroll
                    for (int j = 0; j < JB; ++j) {
                        acc[0][j0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ahi[0], Bhi[j], acc[0][j0 + j], 0, 0, 0);
                        if constexpr (R16_RT == 2) acc[R16_RT - 1][j0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ahi[R16_RT - 1], Bhi[j], acc[R16_RT - 1][j0 + j], 0, 0, 0);
                    }
                }
#ifdef SNERF_PHASE16
                SNERF_PH(q3);
                ph_wait += q1 - q0; ph_conv += q2 - q1; ph_mfma += q3 - q2; ++ph_steps;
#endif
            }
        }
        // epilogue: D[row = 16 h + 4 gq + e, col = 16 j + jj]
        const int64_t rt_m = g.reverse ? n_row_tiles - 1 - rt : rt;
        const int64_t rowu = rt_m * RO_ROWS + wave * (16 * R16_RT);
        auto epilogue = [&](auto interior_tag) {
            constexpr bool INTERIOR = decltype(interior_tag)::value;      // no row of the workgroup tile is masked: branch-free
            float zt[2][4 * R16_RT], ec[2][4];
            auto fetch = [&](int j, float (&z_)[4 * R16_RT], float (&c_)[4]) {     // ACT: pre-activations and [a, b, mu, istd] of column j
                const int64_t n = col0 + 16 * j;
#pragma unroll
                for (int h = 0; h < R16_RT; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int64_t ro = 16 * h + e;
                        if (INTERIOR) {
                            z_[4 * h + e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_z, lz + j * 64, (int)((rowu + ro) * g.eld * 4), 0));
                        } else {
                            int64_t m = rowu + ro + 4 * gq;
                            m = m < g.M ? m : g.M - 1;
                            z_[4 * h + e] = g.ez[m * g.eld + n];
                        }
                    }
                c_[0] = lds_col[NT * 16 + 16 * j + jj]; c_[1] = lds_col[2 * NT * 16 + 16 * j + jj];
                c_[2] = lds_col[3 * NT * 16 + 16 * j + jj]; c_[3] = lds_col[4 * NT * 16 + 16 * j + jj];      // (zeros for a layer without BatchNorm)
            };
            if (ACT) fetch(0, zt[0], ec[0]);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int64_t n = col0 + 16 * j;
                // bias of this lane's column: fetched here (L1-resident), not held in registers across the k-loop
                // plain form: v = alpha acc + (alpha bias) in one fma, and the BatchNorm sums of v - alpha bias = alpha acc are taken from the
                // accumulator itself (sum acc, sum acc^2; scaled by alpha, alpha^2 once, after the tile loop): 3 vector instructions per element
                const float abj = ACT ? 0.f : g.alpha * lds_col[16 * j + jj];
                if (ACT) {
                    if (j + 1 < NT) fetch(j + 1, zt[(j + 1) & 1], ec[(j + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int h = 0; h < R16_RT; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int64_t ro = 16 * h + e;
                        float v = ACT ? g.alpha * acc[h][j][e] : __builtin_fmaf(g.alpha, acc[h][j][e], abj);
                        const float z = ACT ? zt[j & 1][4 * h + e] : 0.f;
                        if (ACT) v *= __builtin_amdgcn_cosf(__builtin_fmaf(ec[j & 1][0], z, ec[j & 1][1]));
                        const bool ok = INTERIOR || rowu + ro + 4 * gq < g.M;
                        if (SNERF_ABL16 & 2) {
                            if (v == 123.456f) g.C[0] = v;
                        } else if (INTERIOR) {
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rs_c, lc + j * 64, (int)((rowu + ro) * g.ldc * 4), SNERF_STORE_AUX);
                        } else if (ok && nok0) {
                            g.C[(rowu + ro + 4 * gq) * g.ldc + n] = v;
                        }
                        if (ACT) {
                            const float s1 = v, s2 = v * ((z - ec[j & 1][2]) * ec[j & 1][3]);
                            st1[j] += ok ? s1 : 0.f;
                            st2[j] += ok ? s2 : 0.f;
                        } else {
                            const float dd = acc[h][j][e];
                            st1[j] += ok ? dd : 0.f;
                            st2[j] += ok ? dd * dd : 0.f;
                        }
                    }
                if (ACT) __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (ACT) {      // this variant's epilogue may spill registers: the prefetched operands must have landed before it may touch them
#pragma unroll
            for (int d = 0; d < PF; ++d) a16_wait_slot<0>(px[d], py[d]);
        }
        SNERF_PH(q4);
        if (rt_m * RO_ROWS + RO_ROWS <= g.M) epilogue(std::true_type{});
        else epilogue(std::false_type{});
#ifdef SNERF_PHASE16
        SNERF_PH(q5);
        ph_epi += q5 - q4;
#endif
#pragma unroll
        for (int h = 0; h < R16_RT; ++h) arow[h] = anext[h];
    }
    // the never-consumed refills of the last round must land before their registers are reused (see gemm_rows_full_kernel)
#pragma unroll
    for (int d = 0; d < PF; ++d) a16_wait_slot<0>(px[d], py[d]);
#ifdef SNERF_PHASE16
    if (g.stats && lane == 0 && (wave == 0 || wave == R16_WAVES / 2) && blockIdx.x < 256) {
        double* o = g.stats + 2 * g.N + 1024 + (blockIdx.x * 2 + (wave != 0)) * 6;
        o[0] = (double)ph_wait; o[1] = (double)ph_conv; o[2] = (double)ph_mfma; o[3] = (double)ph_epi;
        o[4] = (double)(__builtin_amdgcn_s_memtime() - ph_t0); o[5] = (double)ph_steps;
    }
#endif
#ifdef SNERF_STAMP16
    if (g.stats && tid == 0 && blockIdx.x < 256) {     // per workgroup, behind the column sums: cycles of the tile loop, its start and end in 100 MHz ticks
        g.stats[2 * g.N + 3 * blockIdx.x] = (double)stamp_entry;
        g.stats[2 * g.N + 3 * blockIdx.x + 1] = (double)stamp_r0;
        g.stats[2 * g.N + 3 * blockIdx.x + 2] = (double)__builtin_amdgcn_s_memrealtime();
    }
#endif
    if (g.stats) {
        __syncthreads();
        float* red = (float*)lds_w;                        // [waves][NT][2][16]
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float a = ACT ? st1[j] : g.alpha * st1[j], b = ACT ? st2[j] : (g.alpha * g.alpha) * st2[j];
            a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
            a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
            if (gq == 0) {
                red[((wave * NT + j) * 2 + 0) * 16 + jj] = a;
                red[((wave * NT + j) * 2 + 1) * 16 + jj] = b;
            }
        }
        __syncthreads();
        if (tid < NT * 32) {
            const int j = tid >> 5, which = (tid >> 4) & 1, c = tid & 15;
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < R16_WAVES; ++w) s += (double)red[((w * NT + j) * 2 + which) * 16 + c];
            const int64_t n = (int64_t)(grp * NT + j) * 16 + c;
            if (n < g.N) atomicAdd(g.stats + which * g.N + n, s);
        }
    }
#ifdef SNERF_STAMP16
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every store and atomic of this wave acknowledged
    if (g.stats && tid == 0 && blockIdx.x < 256) g.stats[2 * g.N + 768 + blockIdx.x] = (double)__builtin_amdgcn_s_memrealtime();
#endif
}

template <int NT, int PF>
static hipError_t launch_rows16(const GemmX& gx, int aol_mode, int act_mode, dim3 grid, size_t lds, hipStream_t st) {
#define SNERF_GO16(A_, C_)                                                                                            \
    do {                                                                                                              \
        static bool done = false;                                                                                     \
        auto k = gemm_rows16_kernel<NT, PF, A_, C_>;                                                                  \
        if (!done) {                                                                                                  \
            hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            if (e != hipSuccess) return e;                                                                            \
            done = true;                                                                                              \
        }                                                                                                             \
        hipLaunchKernelGGL(k, grid, dim3(64 * R16_WAVES), lds, st, gx);                                                          \
    } while (0)
    if constexpr (PF == 4) {          // only the activation-on-load form fits four k-steps of prefetch without scratch
        if (act_mode == 1 || aol_mode != 1) return hipErrorInvalidValue;
        SNERF_GO16(1, 0);
    } else {
        if (act_mode == 1) SNERF_GO16(0, 1);
        else if (aol_mode == 1) SNERF_GO16(1, 0);
        else SNERF_GO16(0, 0);
    }
#undef SNERF_GO16
    return hipGetLastError();
}

hipError_t launch_split_weights16(const float* W, int rows, int cols, bool transpose, uint16_t* frag, int n_tiles16, int ksteps32, hipStream_t st) {
    const int64_t total = (int64_t)n_tiles16 * ksteps32 * 512;
    if (total <= 0) return hipSuccess;
    int64_t b = (total + 255) / 256;
    if (b > 4096) b = 4096;
    hipLaunchKernelGGL(split_weights16_kernel, dim3((unsigned)b), dim3(256), 0, st, W, rows, cols, transpose ? 1 : 0, frag, n_tiles16, ksteps32);
    return hipGetLastError();
}


// The register-resident-weight row GEMM of round 3 (measured: no gain over gemm_rows16_kernel) is an EXPERIMENT, not product code:
// it is compiled only with -DSNERF_WITH_WREG=1 (tools/variants.py builds such a library; SNERF_GEMM_WREG=1 then routes the eligible shapes to it,
// tools/compare_gemm_paths.py with SNERF_CMP_WREG=1 compares it bit for bit).  The shipped library does not contain it.
#ifndef SNERF_WITH_WREG
#define SNERF_WITH_WREG 0
#endif
#if SNERF_WITH_WREG
// ---------------------------------------------------------------------------------------------------------------------
// Row GEMM with the WEIGHTS IN REGISTERS and the activations shared through LDS.
// What bounds gemm_rows16_kernel (and its 32x32x16 predecessor) is the vector-memory path of the CU, ~16 B per clock
// (tools/probes/ta_rate.hip): with the weights resident in LDS a workgroup can hold half of a 256 x 256 layer, so
// every activation row is loaded - and activated and split into bf16 hi / lo - by TWO workgroups: 806 + 403 MB through that path
// per layer, a 140 us floor under the 128 us of HBM time.  Here every wave owns 32 (or 16) output COLUMNS of the layer and keeps
// their weight fragments, all of K, in its own registers (2 n-tiles x 8 k-steps x hi / lo x 4 registers = 128); the workgroup
// covers all 256 columns, so an activation row is loaded once per layer chip-wide, activated and split once, published to LDS as
// finished MFMA fragments and read from there by all eight waves (LDS has the bandwidth: 85 B / clock against 16 of the memory path).
//   stage = 64 rows x 128 k = 16 fragments (4 row tiles x 4 k-steps): wave w produces fragments (row tile w & 3, k-steps w >> 2 and
//           (w >> 2) + 2) - 16-byte pieces in the operand layout of gemm_rows16_kernel, sin, split, ds_write_b128 - all waves consume all 16;
//   two LDS slots, one barrier per stage: slot (s + 1) & 1 is written in the interval in which every wave consumes slot s & 1
//           (its last readers finished before barrier s);
//   the raw fp32 activations reach LDS by LDS-DMA (global_load_lds_dwordx4: no registers hold a load in flight), two stages =
//           64 KiB per CU ahead of their use - with the weights in registers the LDS is free for it; each wave fetches and later
//           reads only its own fragment's bytes (hand-counted vmcnt, no barrier), the stream runs across tile boundaries.
// Same products and fragment order as gemm_rows16_kernel; per output the k-steps are summed in the same order: bit-identical results.
template <int NTW, int KS, int AOL, int ACT>      // NTW: 16-column n-tiles per wave (2: N = 256, 1: N = 128); KS: 32-k steps (K = 32 KS)
__global__ __launch_bounds__(512) void gemm_wreg_kernel(const GemmX g) {
    constexpr int NST = KS / 4, RAW_D = 2, TM = 64;                         // stage = 64 rows x 128 k = 16 fragments (4 row tiles x 4 k-steps)
    constexpr int SLOT = 32768;
    // [2 slots][4 row tiles][4 k-steps][hi | lo][1 KiB] = 64 KiB of finished fragments | [RAW_D][8 waves][2 units][x | y][1 KiB] raw fp32 | table
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_a[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int jj = lane & 15, gq = lane >> 4;
    const int my_rt = wave & 3, my_kq = wave >> 2;                          // this wave produces fragments (my_rt, my_kq) and (my_rt, my_kq + 2) of every stage
    // this wave's weights: n-tiles NTW * wave + j, every k-step, hi and lo, as MFMA B operands
    bf16x8 Wh[NTW][KS], Wl[NTW][KS];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const u32x4* f = (const u32x4*)(g.frag + ((int64_t)(NTW * wave + j) * KS + ks) * 1024) + lane;
            u32x4 wh = f[0], wl = f[64];
            // opaque from here on: the compiler must KEEP the fragments in registers (left to itself it re-fetches them from global
            // memory inside the tile loop - 256 KB per 64-row tile through the very memory path this kernel exists to relieve)
            asm volatile("" : "+v"(wh), "+v"(wl));
            Wh[j][ks] = __builtin_bit_cast(bf16x8, wh);
            Wl[j][ks] = __builtin_bit_cast(bf16x8, wl);
        }
    uint8_t* lds_raw = lds_a + 2 * SLOT;
    const uint8_t* lds_tab = lds_raw + RAW_D * SLOT;
    if (AOL) {
        float* dst = (float*)lds_tab;
        for (int i = tid; i < 2 * g.act_cols; i += 512) dst[i] = g.act_tab[i];
    }
    // per-column constants behind the table (the epilogue must not load from global memory: hipcc would wait for vmcnt(0) there -
    // for the previous tile's stores and for every DMA in flight)
    float* lds_col = (float*)(lds_tab + (AOL ? (size_t)g.act_cols * 8 : 0));      // [bias | etab a | etab b | mu | istd][N]
    for (int i = tid; i < (int)g.N; i += 512) {
        lds_col[i] = (!ACT && g.bias) ? g.bias[i] : 0.f;
        if (ACT) {
            lds_col[g.N + i] = g.etab[i];
            lds_col[2 * g.N + i] = g.etab[g.N + i];
            lds_col[3 * g.N + i] = g.emu[i];
            lds_col[4 * g.N + i] = g.eistd[i];
        }
    }
    __syncthreads();
    const int col0 = NTW * wave * 16 + jj;                                  // this lane's column of the wave's n-tile 0
    const int64_t n_tiles_m = (g.M + TM - 1) / TM;
    const int worker = blockIdx.x, n_workers = gridDim.x;
    float st1[NTW], st2[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) st1[j] = st2[j] = 0.f;
    const int lc = (int)(4 * gq * g.ldc + col0) * 4;
    const int lz = ACT ? (int)(4 * gq * g.eld + col0) * 4 : 0;
    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc((void*)g.C, 0, (int)(g.M * g.ldc * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc((void*)(ACT ? g.ez : g.A), 0, -1, 0x00020000);

    // the A stream of this wave: stage st of tile t -> rows t * 64 + 16 my_rt + (0..15), k = 128 st + 32 (my_kq + 2 u) + {4 gq .., 16 + 4 gq ..},
    // u = 0, 1; LDS-DMA, saddr form: wave-uniform base (row clamped so that every lane's row offset is >= 0) + per-lane byte offset
    int64_t t_load = worker;                                                // tile / stage of the next DMA group to issue
    int st_load = 0;
    typedef __attribute__((address_space(3))) uint8_t lds_u8;
    const uint32_t raw_lds = (uint32_t)(uintptr_t)((lds_u8*)lds_raw + wave * 4096);      // this wave's 4 KiB of every raw slot (LDS byte address)
    auto dma_issue = [&](int q) {                                           // next stage of the stream -> raw slot q (4 x 1 KiB per wave)
        const int64_t t = t_load < n_tiles_m ? t_load : (n_tiles_m - 1);    // past the end: harmless re-reads of the last tile
        int64_t r0 = t * TM + 16 * my_rt;                                   // first row of this wave's row tile
        int64_t rb = r0 < g.M - 16 ? r0 : g.M - 16;
        rb = rb > 0 ? rb : 0;                                               // base row: <= every row a lane will touch
        int64_t m = r0 + jj;
        m = m < g.M ? m : g.M - 1;
        const uint32_t voff = (uint32_t)((m - rb) * g.lda * 4 + gq * 16);
        const float* base = g.A + rb * g.lda + st_load * 128 + my_kq * 32;
        if (++st_load == NST) { st_load = 0; t_load += n_workers; }
        uint32_t keep;
        if (SNERF_ABLW & 4) return;
        asm volatile(      // (no instruction offsets: they would move the LDS side too)
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %2\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %3\n\t"
            "s_add_u32 m0, m0, 0x400\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %4\n\t"
            "s_add_u32 m0, m0, 0x400\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %5\n\t"
            "s_add_u32 m0, m0, 0x400\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %6\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff), "s"(raw_lds + (uint32_t)q * (uint32_t)SLOT), "s"(base), "s"(base + 16), "s"(base + 64), "s"(base + 80)
            : "memory", "scc");
    };
#pragma unroll
    for (int q = 0; q < RAW_D; ++q) dma_issue(q);
    int rq = 0;                                                             // raw slot of the next stage to finish

    auto produce = [&](int st, int slot) {                                  // finish stage `st` of the current stream position into frag slot
        // all but the 4 (RAW_D - 1) youngest vector-memory operations of this wave have completed: this stage's four DMAs have
        // (operations issued since - epilogue stores - only make the wait stricter)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (RAW_D - 1)) : "memory");
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int kq = my_kq + 2 * u;
            const int k0 = st * 128 + kq * 32 + gq * 4;
            f32x4 ta[2], tb[2];
            if (AOL) {
                const int kc = k0 + 20 <= g.act_cols ? k0 : 0;
                const float* tp = (const float*)lds_tab + kc;
                ta[0] = *(const f32x4*)tp; ta[1] = *(const f32x4*)(tp + 16);
                tb[0] = *(const f32x4*)(tp + g.act_cols); tb[1] = *(const f32x4*)(tp + g.act_cols + 16);
            }
            const uint8_t* rp = lds_raw + rq * SLOT + wave * 4096 + u * 2048 + lane * 16;
            const f32x4 x = *(const f32x4*)rp, y = *(const f32x4*)(rp + 1024);
            float a8[8] = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
            if (AOL) {
                if (st * 128 + kq * 32 < g.act_cols) {                      // wave-uniform (act_cols is a multiple of 32)
#pragma unroll
                    for (int e = 0; e < 8; ++e) a8[e] = __builtin_amdgcn_sinf(__builtin_fmaf(ta[e >> 2][e & 3], a8[e], tb[e >> 2][e & 3]));
                }
            }
            u32x4 hi, lo;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t hh, ll;
                split2_bf16(a8[2 * q], a8[2 * q + 1], hh, ll);
                hi[q] = hh;
                lo[q] = ll;
            }
            uint8_t* f = lds_a + slot * SLOT + (my_rt * 4 + kq) * 2048 + lane * 16;
            *(u32x4*)f = hi;
            *(u32x4*)(f + 1024) = lo;
        }
        dma_issue(rq);                                                      // refill the raw slot just read (its values are in registers)
        rq = rq + 1 == RAW_D ? 0 : rq + 1;
    };

    int slot = 0;
    produce(0, slot);                                                       // stage 0 of the first tile
    __syncthreads();
    for (int64_t t = worker; t < n_tiles_m; t += n_workers) {
        f32x4 acc[4][NTW];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < NTW; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[r][j][e] = 0.f;
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            const bool more = st + 1 < NST || t + n_workers < n_tiles_m;
            const uint8_t* base = lds_a + slot * SLOT + lane * 16;
            // 16 fragment pairs (k-step kq, row tile r), read two pairs ahead of their MFMAs (LDS latency behind 12 MFMAs of the
            // pairs before); the scheduling barriers keep hipcc from sinking the reads next to their use, where each would expose it
            bf16x8 fh[3], fl[3];
            auto rd = [&](int gi, int b) {
                const int kq = gi >> 2, r = gi & 3;
                fh[b] = __builtin_bit_cast(bf16x8, *(const u32x4*)(base + (r * 4 + kq) * 2048));
                fl[b] = __builtin_bit_cast(bf16x8, *(const u32x4*)(base + (r * 4 + kq) * 2048 + 1024));
            };
            rd(0, 0);
            rd(1, 1);
#pragma unroll
            for (int gi = 0; gi < 16; ++gi) {
                const int kq = gi >> 2, r = gi & 3, b = gi % 3;
                if (gi + 2 < 16) rd(gi + 2, (gi + 2) % 3);
                // the next stage (this tile's st + 1, or stage 0 of the wave's next tile) is finished into the other slot in the middle
                // of this one's matrix work (its last readers passed the previous barrier)
                if (gi == 6 && more && !(SNERF_ABLW & 8)) produce(st + 1 < NST ? st + 1 : 0, slot ^ 1);
                __builtin_amdgcn_sched_barrier(0);
#if SNERF_ABLW & 1
#pragma unroll
                for (int j = 0; j < NTW; ++j) acc[r][j][0] += __builtin_bit_cast(float, __builtin_bit_cast(u32x4, fl[b])[0] ^ __builtin_bit_cast(u32x4, fh[b])[1] ^ __builtin_bit_cast(u32x4, Wh[j][4 * st + kq])[0]);
                __builtin_amdgcn_sched_barrier(0);
                continue;
#endif
#pragma unroll
                for (int j = 0; j < NTW; ++j) acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl[b], Wh[j][4 * st + kq], acc[r][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NTW; ++j) acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[b], Wl[j][4 * st + kq], acc[r][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NTW; ++j) acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[b], Wh[j][4 * st + kq], acc[r][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            slot ^= 1;
        }
        // epilogue: D[row = 16 r + 4 gq + e, col = 16 j + jj] of this wave's NTW n-tiles
        const int64_t rowu = t * TM;
        const bool interior = rowu + TM <= g.M;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int64_t n = col0 + 16 * j;
            const float biasj = ACT ? 0.f : lds_col[n];
            const float shiftj = (!ACT && g.stats) ? g.alpha * biasj : 0.f;
            float e_a = 0.f, e_b = 0.f, e_mu = 0.f, e_is = 0.f;
            if (ACT) { e_a = lds_col[g.N + n]; e_b = lds_col[2 * g.N + n]; e_mu = lds_col[3 * g.N + n]; e_is = lds_col[4 * g.N + n]; }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float z[4];
                if (ACT) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        int64_t m = rowu + 16 * r + 4 * gq + e;
                        m = m < g.M ? m : g.M - 1;
                        z[e] = interior ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_z, lz + j * 64, (int)((rowu + 16 * r + e) * g.eld * 4), 0))
                                        : g.ez[m * g.eld + n];
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int64_t ro = 16 * r + e;
                    float v = g.alpha * (acc[r][j][e] + biasj);
                    if (ACT) v *= __builtin_amdgcn_cosf(__builtin_fmaf(e_a, z[e], e_b));
                    const bool ok = interior || rowu + ro + 4 * gq < g.M;
                    if (SNERF_ABLW & 2) { if (v == 123.456f) g.C[0] = v; }
                    else if (interior) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rs_c, lc + j * 64, (int)((rowu + ro) * g.ldc * 4), SNERF_STORE_AUX);
                    else if (ok) g.C[(rowu + ro + 4 * gq) * g.ldc + n] = v;
                    if (ACT) {
                        st1[j] += ok ? v : 0.f;
                        st2[j] += ok ? v * ((z[e] - e_mu) * e_is) : 0.f;
                    } else {
                        const float dd = v - shiftj;
                        st1[j] += ok ? dd : 0.f;
                        st2[j] += ok ? dd * dd : 0.f;
                    }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the never-consumed DMAs of the last stages must land before the workgroup's LDS is released
    if (g.stats) {      // every wave owns its columns: reduce over the four row groups of the lanes, one double atomic per column
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            float a = st1[j], b = st2[j];
            a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
            a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
            if (gq == 0) {
                atomicAdd(g.stats + col0 + 16 * j, (double)a);
                atomicAdd(g.stats + g.N + col0 + 16 * j, (double)b);
            }
        }
    }
}

template <int NTW, int KS>
static hipError_t launch_wreg(const GemmX& gx, int aol_mode, int act_mode, hipStream_t st) {
    const size_t lds = (2 + 2) * 32768 + (aol_mode ? (size_t)gx.act_cols * 8 : 0) + (size_t)gx.N * 4 * (act_mode ? 5 : 1);
    const dim3 grid(ro_blocks16());
#define SNERF_GOW(A_, C_)                                                                                              \
    do {                                                                                                              \
        static bool done = false;                                                                                     \
        auto k = gemm_wreg_kernel<NTW, KS, A_, C_>;                                                                   \
        if (!done) {                                                                                                  \
            hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            if (e != hipSuccess) return e;                                                                            \
            done = true;                                                                                              \
        }                                                                                                             \
        hipLaunchKernelGGL(k, grid, dim3(512), lds, st, gx);                                                          \
    } while (0)
    if (act_mode) SNERF_GOW(0, 1);
    else if (aol_mode) SNERF_GOW(1, 0);
    else SNERF_GOW(0, 0);
#undef SNERF_GOW
    return hipGetLastError();
}

// shapes the (experimental) register-resident-weight kernel takes: N = 256 or 128 output columns, K = 256
bool gemm_wreg_ok(const GemmX& gx) {
    // opt-in (SNERF_GEMM_WREG=1): measured equal to gemm_rows16_kernel in the forward (213 against 200-224 us per 256 -> 256 layer) and
    // slower with the activation-backward epilogue (282 against 236 us) - see DESIGN 5.4 for what the per-stage barrier costs
    static const int mode = [] { const char* e = getenv("SNERF_GEMM_WREG"); if (SNERF_ABLW) return 1; return (e && e[0] == '1') ? 1 : 0; }();
    const int KS32 = gx.ksteps / 2;
    return mode && gx.W && KS32 == 8 && gx.ksteps == 16 && (gx.N == 256 || gx.N == 128) && gx.N == (int64_t)gx.n_tiles * 32;
}

#endif      // SNERF_WITH_WREG

// gx: as launch_gemm_bf16x3 prepared it for the full-tile path (raw weights in gx.W, K in whole 32-k steps, N = 32 n_tiles)
// nt16: 16-column n-tiles per column group - 8 (128 columns; K <= 256) or 4 (64 columns: the K = 320 layer, whose 128-column slice of hi / lo weights
// does not fit the LDS beside its table)
hipError_t launch_gemm_rows16(const GemmX& gx_in, int aol_mode, int act_mode, dim3 grid, size_t lds, hipStream_t st, int nt16) {
    GemmX gx = gx_in;
    gx.reverse = stream_direction(gx.M);      // every other streaming launch walks its row tiles backwards (gemm.hip)
    const int KS32 = gx.ksteps / 2;
    hipError_t e = launch_split_weights16(gx.W, gx.w_rows, gx.w_cols, gx.w_transpose != 0, const_cast<uint16_t*>(gx.frag), 2 * gx.n_tiles, KS32, st);
    if (e != hipSuccess) return e;
#if SNERF_WITH_WREG
    if (gemm_wreg_ok(gx)) {      // weights in registers, activations through LDS: each A byte crosses the memory path once
        return gx.N == 256 ? launch_wreg<2, 8>(gx, aol_mode, act_mode, st) : launch_wreg<1, 8>(gx, aol_mode, act_mode, st);
    }
#endif
    // 32-k steps of A in flight: 4 with activation on load (244 registers, no scratch), 2 otherwise (the plain form spills at 4,
    // the activation-backward epilogue needs the registers); must divide the k-step count
    int pf16 = (aol_mode && !act_mode && KS32 % 4 == 0) ? 4 : (KS32 % 2 == 0 ? 2 : 1);
    if (R16_RT == 1) {                      // experimental geometry (16 waves x 16 rows, 128 registers): only the scratch-free forms
        if (act_mode) return hipErrorInvalidValue;
        pf16 = 1;
    }
    if (nt16 == 4) {
        if (act_mode) return hipErrorInvalidValue;               // (the 64-column form exists for forwards only)
        if (pf16 == 4) return launch_rows16<4, 4>(gx, 1, 0, grid, lds, st);
        return pf16 == 2 ? launch_rows16<4, 2>(gx, aol_mode, 0, grid, lds, st) : launch_rows16<4, 1>(gx, aol_mode, 0, grid, lds, st);
    }
    if (pf16 == 4) return launch_rows16<8, 4>(gx, 1, 0, grid, lds, st);
    return pf16 == 2 ? launch_rows16<8, 2>(gx, aol_mode, act_mode, grid, lds, st) : launch_rows16<8, 1>(gx, aol_mode, act_mode, grid, lds, st);
}

}  // namespace snerf
