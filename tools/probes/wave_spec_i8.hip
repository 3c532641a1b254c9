// Would WAVE SPECIALISATION lift the int8-digit field kernel (csrc/kernels_i8x2.hip) off its 64 % matrix-pipe utilisation?
//
// The kernel's per-block instruction mix: a 256 -> 256 layer block of one 32-point wave tile = 8 k-steps x 3
// v_mfma_i32_32x32x32_i8 (T a -> M; T b, L a -> X) + the epilogue of 16 output elements per lane: merge (M << 8) + X, cvt, fma (scale, bias),
// v_sin, and the digit split (v_cvt_pknorm_i16_f32 per pair, two v_perm + one v_xor per quad) = 4.25 VALU + 1 transcendental per element,
// weight fragments read from LDS (2 x ds_read_b128 per k-step).  One workgroup per CU, 8 waves = two per SIMD, as in the kernel.
//
//   SYM   every wave does both: 24 MFMAs interleaved (by the compiler) with the epilogue of its previous block - today's structure
//   SPEC  waves 0-3 (one per SIMD) issue ONLY the MFMAs (+ weight / digit reads, the merge and the hand-off stores), two tiles each;
//         waves 4-7 ONLY the epilogues of those tiles: merged int32 blocks travel MFMA wave -> epilogue wave through LDS
//         (4 x ds_write_b128 / ds_read_b128 per block), the new digits travel back (2 + 2), ready flags are LDS words polled with s_sleep
//   MFMA  the 48 MFMAs per SIMD and block pair alone (the floor: 32 cycles each)
//   SYM16x64  SYM with the block on v_mfma_i32_16x16x64_i8 (48 half-size MFMAs per block: twice the issue slots between matrix instructions)
// Same matrix work per SIMD in all three (48 MFMAs per iteration); reported: cycles per MFMA and SIMD (s_memtime, 100 MHz ticks x clock
// ratio measured against MFMA-only = 32 cycles).  No global memory traffic in the loop.
// build: hipcc -O3 --offload-arch=gfx950 -o build/probes/wave_spec_i8 tools/probes/wave_spec_i8.hip ; run on an MI355X
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define MFMA(a, b, c) __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0)

// LDS map (bytes): [0, 32K) weight fragments (32 x 1 KiB, lane-linear); [32K, 96K) exchange: per SIMD 16 KiB = 2 parities x 2 tiles x 4 KiB merged block;
// [96K, 128K) digits: per SIMD 8 KiB = 2 tiles x (a 1 KiB + b 1 KiB) x 2 slots; [128K, +256) flags
constexpr int W_OFF = 0, X_OFF = 32768, D_OFF = 98304, F_OFF = 131072, LDS_BYTES = 131072 + 256;

// the epilogue of one 32 x 32 block (16 elements per lane): returns the two digit quads x 4 (a-digits, b-digits), as the kernel forms them
__device__ __forceinline__ void epilogue16(const i32x16& m, float sc, float bi, u32x4& da, u32x4& db) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float h[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float z = __builtin_fmaf((float)m[4 * q + e], sc, bi);          // cvt + fma
            h[e] = __builtin_amdgcn_sinf(z);                                        // v_sin_f32 (revolutions)
        }
        const uint32_t p0 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pknorm_i16(h[0], h[1]));
        const uint32_t p1 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pknorm_i16(h[2], h[3]));
        da[q] = __builtin_amdgcn_perm(p1, p0, 0x07050301u);                        // the four high bytes
        db[q] = __builtin_amdgcn_perm(p1, p0, 0x06040200u) ^ 0x80808080u;          // the four low bytes, re-centred
    }
}

template <int MODE>      // 0 SYM, 1 SPEC, 2 MFMA + weight reads only, 3 SYM on 16x16x64, 4 MFMA from registers (yardstick)
__global__ __launch_bounds__(512) void probe(float* out, uint64_t* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, simd = wave & 3;
    for (int i = threadIdx.x; i < (LDS_BYTES - 256) / 4; i += blockDim.x) ((uint32_t*)lds)[i] = (uint32_t)i * 2654435761u;
    if (threadIdx.x < 64) ((volatile uint32_t*)(lds + F_OFF))[threadIdx.x] = 0;
    __syncthreads();
    volatile uint32_t* flags = (volatile uint32_t*)(lds + F_OFF);      // [simd][0] = blocks published by the MFMA wave, [simd][1] = blocks consumed
    const float sc = 1.0e-9f + lane * 1e-12f, bi = 0.25f;
    uint32_t sink = 0;
    uint64_t t0 = 0, t1 = 0;

    if (MODE == 0) {
        // ---- SYM: each wave its own tile; M, X of block b, merged block of b-1 in epilogue; digits stay in registers
        i32x16 M, X, P;
        for (int e = 0; e < 16; ++e) { M[e] = 0; X[e] = 0; P[e] = lane + e; }
        u32x4 da[8], db[8];
        for (int k = 0; k < 8; ++k) { da[k] = *(const u32x4*)(lds + D_OFF + k * 1024 + lane * 16); db[k] = da[k] ^ 0x01010101u; }
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
            const uint8_t* wp = lds + W_OFF + (it & 1) * 16384 + lane * 16;         // 16 fragments per block, immediate offsets
            u32x4 na, nb;
            epilogue16(P, sc, bi, na, nb);                                          // the previous block's epilogue: independent of this block's MFMAs
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const i32x4 T = *(const i32x4*)(wp + (2 * k) * 1024);
                const i32x4 L = *(const i32x4*)(wp + (2 * k + 1) * 1024);
                M = MFMA(T, __builtin_bit_cast(i32x4, da[k]), M);
                X = MFMA(T, __builtin_bit_cast(i32x4, db[k]), X);
                X = MFMA(L, __builtin_bit_cast(i32x4, da[k]), X);
            }
            da[7] = na; db[7] = nb;                                                  // the new digits are a k-step of the next layer (kept live)
#pragma unroll
            for (int e = 0; e < 16; ++e) { P[e] = (M[e] << 8) + X[e]; M[e] = 0; X[e] = 0; }
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int k = 0; k < 8; ++k) sink ^= da[k][0] ^ db[k][3];
    } else if (MODE == 3) {
        // ---- SYM on v_mfma_i32_16x16x64_i8: the same block (32 rows x 32 points x K = 256, three digit products) as 48 MFMAs of half the size -
        // 2 row halves x 2 point halves x 4 k-steps of 64; same LDS bytes, same epilogue (16 elements per lane = 4 sub-blocks x 4 registers)
        i32x4 M[2][2], X[2][2];
        i32x16 P;
        for (int r = 0; r < 2; ++r) for (int q = 0; q < 2; ++q) for (int e = 0; e < 4; ++e) { M[r][q][e] = 0; X[r][q][e] = 0; }
        for (int e = 0; e < 16; ++e) P[e] = lane + e;
        u32x4 da[8], db[8];                                                          // [k-step of 64][point half]
        for (int k = 0; k < 8; ++k) { da[k] = *(const u32x4*)(lds + D_OFF + k * 1024 + lane * 16); db[k] = da[k] ^ 0x01010101u; }
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
            const uint8_t* wp = lds + W_OFF + (it & 1) * 16384 + lane * 16;
            u32x4 na, nb;
            epilogue16(P, sc, bi, na, nb);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                i32x4 T[2], L[2];
                T[0] = *(const i32x4*)(wp + (4 * k) * 1024); T[1] = *(const i32x4*)(wp + (4 * k + 1) * 1024);
                L[0] = *(const i32x4*)(wp + (4 * k + 2) * 1024); L[1] = *(const i32x4*)(wp + (4 * k + 3) * 1024);
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        M[r][q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(T[r], __builtin_bit_cast(i32x4, da[2 * k + q]), M[r][q], 0, 0, 0);
                        X[r][q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(T[r], __builtin_bit_cast(i32x4, db[2 * k + q]), X[r][q], 0, 0, 0);
                        X[r][q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(L[r], __builtin_bit_cast(i32x4, da[2 * k + q]), X[r][q], 0, 0, 0);
                    }
            }
            da[7] = na; db[7] = nb;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { P[8 * r + 4 * q + e] = (M[r][q][e] << 8) + X[r][q][e]; M[r][q][e] = 0; X[r][q][e] = 0; }
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int k = 0; k < 8; ++k) sink ^= da[k][0] ^ db[k][3];
    } else if (MODE == 4) {
        // ---- the yardstick: 48 independent-enough MFMAs per iteration from registers (four accumulators in rotation, no LDS), waves 0-3
        if (wave < 4) {
            i32x16 A4[4];
            for (int t = 0; t < 4; ++t) for (int e = 0; e < 16; ++e) A4[t][e] = 0;
            const i32x4 a = {lane, 1, 2, 3}, b = {5, lane, 7, 8};
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int k = 0; k < 12; ++k)
#pragma unroll
                    for (int t = 0; t < 4; ++t) A4[t] = MFMA(a, b, A4[t]);
            }
            t1 = __builtin_amdgcn_s_memtime();
            for (int t = 0; t < 4; ++t) sink ^= (uint32_t)A4[t][t];
        }
    } else if (MODE == 5) {
        // ---- the yardstick with LIVE data: the same 48 register-fed MFMAs per iteration, but every MFMA gets another pair of random operands (eight sets in
        // rotation, taken from the hashed LDS image): operand toggling as in the real kernel - what clock does the chip hold for THAT?
        if (wave < 4) {
            i32x16 A4[4];
            for (int t = 0; t < 4; ++t) for (int e = 0; e < 16; ++e) A4[t][e] = 0;
            i32x4 ra[8], rb[8];
            for (int q = 0; q < 8; ++q) { ra[q] = *(const i32x4*)(lds + W_OFF + q * 1024 + lane * 16); rb[q] = *(const i32x4*)(lds + W_OFF + (8 + q) * 1024 + lane * 16); }
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int k = 0; k < 12; ++k)
#pragma unroll
                    for (int t = 0; t < 4; ++t) A4[t] = MFMA(ra[(4 * k + t) & 7], rb[(4 * k + t + 3) & 7], A4[t]);
            }
            t1 = __builtin_amdgcn_s_memtime();
            for (int t = 0; t < 4; ++t) sink ^= (uint32_t)A4[t][t];
        }
    } else if (MODE == 2) {
        // ---- MFMA only, waves 0-3 two tiles each (48 per iteration), waves 4-7 idle
        if (wave < 4) {
            i32x16 M[2], X[2];
            for (int t = 0; t < 2; ++t) for (int e = 0; e < 16; ++e) { M[t][e] = 0; X[t][e] = 0; }
            const i32x4 a = {lane, 1, 2, 3}, b = {5, lane, 7, 8};
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < iters; ++it) {
                const uint8_t* wp = lds + W_OFF + (it & 1) * 16384 + lane * 16;
                i32x4 T[2], L[2];
                T[0] = *(const i32x4*)(wp); L[0] = *(const i32x4*)(wp + 1024);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (k < 7) { T[(k + 1) & 1] = *(const i32x4*)(wp + (2 * k + 2) * 1024); L[(k + 1) & 1] = *(const i32x4*)(wp + (2 * k + 3) * 1024); }
#pragma unroll
                    for (int t = 0; t < 2; ++t) { M[t] = MFMA(T[k & 1], a, M[t]); X[t] = MFMA(T[k & 1], b, X[t]); X[t] = MFMA(L[k & 1], a, X[t]); }
                }
            }
            t1 = __builtin_amdgcn_s_memtime();
            for (int t = 0; t < 2; ++t) sink ^= (uint32_t)(M[t][0] + X[t][5]);
        }
    } else {
        // ---- SPEC
        uint8_t* xch = lds + X_OFF + simd * 16384;                                   // [parity][tile][4 KiB]: merged block, 4 x 16 B per lane
        uint8_t* dig = lds + D_OFF + simd * 8192;                                    // [slot][tile][a 1 KiB | b 1 KiB]
        if (wave < 4) {
            i32x16 M[2], X[2];
            for (int t = 0; t < 2; ++t) for (int e = 0; e < 16; ++e) { M[t][e] = 0; X[t][e] = 0; }
            // the activations of the running layer stay in registers (2 tiles x 8 k-steps x (a, b)); one k-step per tile is refreshed per
            // block from what the epilogue wave wrote (the traffic of a real layer: every block's digits are read once)
            i32x4 da[2][8], db[2][8];
            for (int t = 0; t < 2; ++t) for (int k = 0; k < 8; ++k) { da[t][k] = *(const i32x4*)(dig + t * 2048 + lane * 16) + k; db[t][k] = da[t][k] ^ 0x01010101; }
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < iters; ++it) {
                // exchange buffer (it & 1) is free once the epilogue wave has consumed block it-2
                if (it > 1) while (flags[simd * 2 + 1] < (uint32_t)(it - 1)) __builtin_amdgcn_s_sleep(1);
                const uint8_t* wp = lds + W_OFF + (it & 1) * 16384 + lane * 16;
                i32x4 T[2], L[2];
                T[0] = *(const i32x4*)(wp); L[0] = *(const i32x4*)(wp + 1024);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (k < 7) { T[(k + 1) & 1] = *(const i32x4*)(wp + (2 * k + 2) * 1024); L[(k + 1) & 1] = *(const i32x4*)(wp + (2 * k + 3) * 1024); }   // one k-step ahead
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        M[t] = MFMA(T[k & 1], da[t][k], M[t]); X[t] = MFMA(T[k & 1], db[t][k], X[t]); X[t] = MFMA(L[k & 1], da[t][k], X[t]);
                    }
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    da[t][7] = *(const i32x4*)(dig + ((it & 1) * 2 + t) * 2048 + lane * 16);
                    db[t][7] = *(const i32x4*)(dig + ((it & 1) * 2 + t) * 2048 + 1024 + lane * 16);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        i32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] = (M[t][4 * q + e] << 8) + X[t][4 * q + e]; M[t][4 * q + e] = 0; X[t][4 * q + e] = 0; }
                        *(i32x4*)(xch + (it & 1) * 8192 + t * 4096 + q * 1024 + lane * 16) = v;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) flags[simd * 2 + 0] = (uint32_t)(it + 1);
            }
            t1 = __builtin_amdgcn_s_memtime();
            for (int t = 0; t < 2; ++t) sink ^= (uint32_t)(da[t][7][0] + db[t][3][1]);
        } else {
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < iters; ++it) {
                while (flags[simd * 2 + 0] < (uint32_t)(it + 1)) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                i32x16 P[2];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const i32x4 v = *(const i32x4*)(xch + (it & 1) * 8192 + t * 4096 + q * 1024 + lane * 16);
#pragma unroll
                        for (int e = 0; e < 4; ++e) P[t][4 * q + e] = v[e];
                    }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) flags[simd * 2 + 1] = (uint32_t)(it + 1);            // the blocks are in registers: the MFMA wave may overwrite them
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    u32x4 na, nb;
                    epilogue16(P[t], sc, bi, na, nb);
                    *(u32x4*)(dig + ((it & 1) * 2 + t) * 2048 + lane * 16) = na;
                    *(u32x4*)(dig + ((it & 1) * 2 + t) * 2048 + 1024 + lane * 16) = nb;
                }
            }
            t1 = __builtin_amdgcn_s_memtime();
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)sink;
    if (blockIdx.x == 0 && lane == 0) { cyc[2 * wave] = t0; cyc[2 * wave + 1] = t1; }
}

template <int MODE>
static double run(const char* name, float* out, uint64_t* cyc, double mfma_ticks) {
    const int iters = 4000;
    auto k = probe<MODE>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipMemset(cyc, 0, 256);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(512), LDS_BYTES, 0, out, cyc, iters);
    hipEvent_t e0, e1v;
    hipEventCreate(&e0); hipEventCreate(&e1v);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), LDS_BYTES, 0, out, cyc, iters);
    hipEventRecord(e1v, 0);
    hipError_t e = hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1v);
    if (e != hipSuccess) { printf("%s: %s\n", name, hipGetErrorString(e)); return 0; }
    uint64_t c[16];
    hipMemcpy(c, cyc, 128, hipMemcpyDeviceToHost);
    uint64_t s0 = ~0ull, e1 = 0;
    for (int w = 0; w < 8; ++w) if (c[2 * w + 1]) { if (c[2 * w] < s0) s0 = c[2 * w]; if (c[2 * w + 1] > e1) e1 = c[2 * w + 1]; }
    // matrix work per SIMD and iteration: SYM 2 waves x 24, SPEC / MFMA 1 wave x 48
    const double ticks = (double)(e1 - s0) / iters / 48.0;
    printf("%-10s %8.3f memtime ticks per MFMA and SIMD", name, ticks);
    if (mfma_ticks > 0) printf("  = %5.1f cycles (MFMA-only = 32)  -> matrix pipe %4.1f %% busy", ticks / mfma_ticks * 32.0, mfma_ticks / ticks * 100.0);
    printf("   [%.3f ms for %d x 48 MFMAs per SIMD: %.2f ns per MFMA]\n", ms, iters, ms * 1e6 / iters / 48.0);
    return ticks;
}

int main() {
    float* out; uint64_t* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256);
    const double base = run<4>("MFMA-regs", out, cyc, 0);      // 32 cycles each by construction: the tick -> cycle yardstick
    run<5>("MFMA-live", out, cyc, base);     // the same with random operands that change every MFMA
    run<2>("MFMA+LDS", out, cyc, base);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("SYM", out, cyc, base);
        run<1>("SPEC", out, cyc, base);
        run<3>("SYM16x64", out, cyc, base);      // counted in 32x32x32 equivalents (two 16x16x64 = one)
    }
    return 0;
}
