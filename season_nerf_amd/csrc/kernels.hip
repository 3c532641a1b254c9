// gfx950 (MI355X / CDNA4) kernels of the Season-NeRF per-ray hot path.  See program.h for the data layout.
//
//  mlp_kernel<PROG, W, VARIANT>   fused register-resident MFMA chain (field program: one sample point per lane
//                                 column, 32 points per wave, 128 per workgroup; group program: time/sun groups)
//  composite_kernel               exclusive-prefix transmittance + shading, one wavefront per ray
//
// Structure of mlp_kernel (persistent, 4 waves = 1 wave per SIMD, up to 512 VGPR+AGPR per lane):
//   * weights stream L2 -> LDS through a ring of 16 KiB chunks filled by LDS-DMA (global_load_lds_dwordx4),
//     RING_D-1 chunks in flight, one counted s_waitcnt vmcnt + one s_barrier per chunk (24 MFMAs);
//   * each wave reads the A fragments (weights) with conflict-free lane-linear ds_read_b128 and multiplies them
//     against its own 32 points, whose activations stay in registers from the positional encoding to the heads;
//   * 3 bf16 MFMAs per product (hi*hi + lo*hi + hi*lo, fp32 accumulate) keep the result within ~1e-5 of fp32,
//     the parity bar (1e-4 rel) being out of reach of plain bf16 on an omega_0 = 30 SIREN.
#include "mlp_bf16_device.h"

namespace snerf {


template <int NB, int KS0, int KS1, bool SIN, int TERMS = 3, bool LO_OUT = true>
__device__ __forceinline__ void run_layer(Ring& rg, const uint8_t* stream, uint32_t stream_bytes, lds_char* lds,
                                          lds_cfloat* bias_l, const Frag* in0, const Frag* in1, Frag* out,
                                          f32x16* raw, int wave, int lane) {
    constexpr int KS = KS0 + KS1, NP = NB * KS;
    constexpr bool PIPE = KS >= 4;      // tiny layers (2 k-steps per block) run the previous epilogue in one piece
    const int h = lane >> 5;
    u32x4 fh[PF], fl[PF] = {};
#pragma unroll
    for (int q = 0; q < PF; ++q) {
        if (q < NP) {
            if (q % kChunkPairs == 0) ring_step(rg, stream, stream_bytes, lds, wave, lane);
            lds_char* ap = lds + rg.cur + (q % kChunkPairs) * kPairBytes + lane * 16;
            fh[q] = *(lds_cu32x4*)ap;
            if (TERMS == 3) fl[q] = *(lds_cu32x4*)(ap + kFragBytes);
        }
    }
    f32x16 accs[2];
    EpiTmp et[8];
    f32x16 next_init = load_bias(bias_l, 0, h);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        f32x16 acc = next_init;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int q = b * KS + s;
            const u32x4 a_hi = fh[q % PF], a_lo = fl[q % PF];
            if (q + PF < NP) {
                const int qn = q + PF;
                if (qn % kChunkPairs == 0) ring_step(rg, stream, stream_bytes, lds, wave, lane);
#if defined(SNERF_ABLATE) && (ABL & 2)     // timing-only: A fragments stay in registers, no LDS reads
                asm volatile("" : "+v"(fh[q % PF]), "+v"(fl[q % PF]));
#else
                lds_char* ap = lds + rg.cur + (qn % kChunkPairs) * kPairBytes + lane * 16;
                fh[q % PF] = *(lds_cu32x4*)ap;
                if (TERMS == 3) fl[q % PF] = *(lds_cu32x4*)(ap + kFragBytes);
#endif
            }
            if (TERMS == 3) acc = mfma3(a_hi, a_lo, s < KS0 ? in0[s] : in1[s - KS0], acc);
            else acc = mfma1(a_hi, s < KS0 ? in0[s] : in1[s - KS0], acc);
            // next block's bias -> its accumulator, issued a few k-steps early (after the previous epilogue released
            // the other accumulator buffer), so the LDS latency hides behind this block's last MFMAs
            if (b + 1 < NB && s == (KS >= 4 ? KS - 1 : 0)) next_init = load_bias(bias_l, b + 1, h);
            if (SIN && b > 0) {
                // previous block's epilogue: pair e runs phase A at step sA(e) = e*(KS-3)/8, B at sA+1, C at sA+2
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (PIPE) {
                        const int sA = 1 + (e * (KS - 4)) / 8;   // starts one k-step late: the block's last MFMA has retired
                        if (LO_OUT && s == sA + 2) epi_C(e, et[e], out + 2 * (b - 1));
                        if (s == sA + 1) epi_B(e, et[e], out + 2 * (b - 1));
                        if (s == sA) epi_A(accs[(b - 1) & 1], e, et[e]);
                    } else if (s == 0) {
                        epi_A(accs[(b - 1) & 1], e, et[e]);
                        epi_B(e, et[e], out + 2 * (b - 1));
                        if (LO_OUT) epi_C(e, et[e], out + 2 * (b - 1));
                    }
                }
            }
#ifndef SNERF_NO_SCHED_GROUPS
            // pin the interleave: each MFMA gets at most 1 LDS read, 1 transcendental and 3 plain VALU in its shadow
            // (an MFMA holds vector issue for 8 of its 32 cycles: <= 6 plain-VALU-equivalents hide per gap)
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
                __builtin_amdgcn_sched_group_barrier(SG_DSREAD, 1, 0);
                __builtin_amdgcn_sched_group_barrier(SG_TRANS, 1, 0);
                __builtin_amdgcn_sched_group_barrier(SG_VALU, 3, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        accs[b & 1] = acc;
    }
    if (SIN) {
#pragma unroll
        for (int e = 0; e < 8; ++e) epi_A(accs[(NB - 1) & 1], e, et[e]);
#pragma unroll
        for (int e = 0; e < 8; ++e) epi_B(e, et[e], out + 2 * (NB - 1));
        if (LO_OUT) {
#pragma unroll
            for (int e = 0; e < 8; ++e) epi_C(e, et[e], out + 2 * (NB - 1));
        }
    } else {
        *raw = accs[0];
    }
}



// =====================================================================================================
// FAST: every layer but the first multiplies the bf16-rounded operands once (first layer / positional encoding keep the
// 3-term product); activations keep no low part.
template <int PROG, int W, int VARIANT, bool FAST = false>
__global__ __launch_bounds__(256, 1) void mlp_kernel(const MlpArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int C_MAX = kMaxClasses;
    constexpr int W2 = W / 2;
    lds_char* lds = (lds_char*)smem;
    __attribute__((address_space(3))) float* bias_lds = (__attribute__((address_space(3))) float*)(lds + RING_BYTES);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5;
    const int C = A.n_classes;

    // bias table -> LDS (once per workgroup)
    for (int i = threadIdx.x; i < A.bias_floats; i += 256) bias_lds[i] = A.bias[i];

    // prologue: RING_D-2 chunks in flight (the refill target trails the consumer by two slots)
    Ring rg;
    rg.rd = 0;
    rg.cur = 0;
    rg.goff = 0;
    {
        uint32_t wr = 0;
#pragma unroll
        for (int c = 0; c < RING_D - 2; ++c) {
            dma_chunk(A.stream, rg.goff, lds, wr, wave, lane);
            rg.goff += kChunkBytes;
            if (rg.goff >= A.stream_bytes) rg.goff = 0;
            wr += kChunkBytes;
        }
        rg.wr = wr;   // = (RING_D-2)*chunk: at the step that publishes chunk k this is the slot of chunk k-2
    }
    __syncthreads();   // bias table visible (drains the prologue DMAs once; harmless)

    // VARIANT 3 (ray visibility, mlp_device.h RaySum): a "tile" is a group of 4 rays (one per wave), walked in `passes` steps of 32 samples
    const int64_t n_tiles = VARIANT == 3 ? (A.n + 3) / 4 : (A.n + TILE_PTS - 1) / TILE_PTS;
    const int passes = VARIANT == 3 ? (A.n_samples + 31) / 32 : 1;
    int pass = 0;
    RaySum rs;
    for (int64_t tile = blockIdx.x; tile < n_tiles;) {
        const int64_t n = tile * TILE_PTS + wave * 32 + (lane & 31);
        const bool valid = n < A.n;
        const int64_t nc = valid ? n : A.n - 1;
        const int64_t g = VARIANT == 3 ? 0 : nc / A.group_size;

        if constexpr (PROG == PROG_FIELD) {
            // ---- sample position (misc.py:234-247 fused): top*(1-t) + bot*t, two roundings + one add, no fma
            float x0, x1, x2;
            if constexpr (VARIANT == 3) {
                raysum_point(rs, A, tile, 4, wave, pass, lane, x0, x1, x2);
            } else if (A.points) {
                x0 = A.points[nc * 3]; x1 = A.points[nc * 3 + 1]; x2 = A.points[nc * 3 + 2];
            } else {
                const int64_t r = nc / A.n_samples;
                const int s = (int)(nc - r * A.n_samples);
                const float t = A.tvals[s], omt = __fsub_rn(1.f, t);
                x0 = __fadd_rn(__fmul_rn(A.top[r * 3], omt), __fmul_rn(A.bot[r * 3], t));
                x1 = __fadd_rn(__fmul_rn(A.top[r * 3 + 1], omt), __fmul_rn(A.bot[r * 3 + 1], t));
                x2 = __fadd_rn(__fmul_rn(A.top[r * 3 + 2], omt), __fmul_rn(A.bot[r * 3 + 2], t));
            }
            // every per-tile input is loaded here, before the MFMA chain: a plain load in the middle of the chain
            // makes hipcc drain the LDS-DMA pipeline with vmcnt(0)
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
            float pcls[C_MAX];
#pragma unroll
            for (int c = 0; c < C_MAX; ++c) pcls[c] = 0.f;
            if constexpr (VARIANT <= 1) { s0 = A.sun[g * 3]; s1 = A.sun[g * 3 + 1]; s2 = A.sun[g * 3 + 2]; }
            if constexpr (VARIANT == 0) {
                if (A.classes) {
#pragma unroll
                    for (int c = 0; c < C_MAX; ++c) if (c < C) pcls[c] = A.classes[g * C + c];
                }
            }
            Frag pe[PEPOS_KS];
            make_pe_pos(x0, x1, x2, h, pe);

            constexpr int KW = W / 16, KW2 = W2 / 16;
            Frag hA[KW], hB[KW];
            f32x16 raw;
#define LAYER(L, NBv, K0, K1, SINv, IN0, IN1, OUT, RAW)                                                              \
    run_layer<NBv, K0, K1, SINv, (FAST && L != F_FC1) ? 1 : 3, !FAST>(rg, A.stream, A.stream_bytes, lds,                   \
                                 bias_lds + prog_bias_start(PROG_FIELD, W, C_MAX, L), IN0, IN1, OUT, RAW, wave, lane)
            // trunk (G_NeRF.py:80-91)
            LAYER(F_FC1, W / 32, PEPOS_KS, 0, true, pe, nullptr, hA, nullptr);
            LAYER(F_FC2, W / 32, KW, 0, true, hA, nullptr, hB, nullptr);
            LAYER(F_FC3, W / 32, KW, 0, true, hB, nullptr, hA, nullptr);
            LAYER(F_FC4, W / 32, KW, 0, true, hA, nullptr, hB, nullptr);
            LAYER(F_FC5, W / 32, KW, PEPOS_KS, true, hB, pe, hA, nullptr);
            LAYER(F_FC6, W / 32, KW, 0, true, hA, nullptr, hB, nullptr);
            LAYER(F_FC7, W / 32, KW, 0, true, hB, nullptr, hA, nullptr);
            LAYER(F_FC8, W / 32, KW, 0, true, hA, nullptr, hB, nullptr);
            Frag x1f[KW2];
            LAYER(F_FC9, W2 / 32, KW, 0, true, hB, nullptr, x1f, nullptr);
            // sigma / colour head (G_NeRF.py:93-98): regs 0..2 colour, 3 density (lane-half 0)
            LAYER(F_HEAD, 1, KW2, 0, false, x1f, nullptr, nullptr, &raw);
            const float col_r = raw[0], col_g = raw[1], col_b = raw[2], rho_raw = raw[3];
            float sv_raw = 0.f;
            float adj[3 * C_MAX];
#pragma unroll
            for (int i = 0; i < 3 * C_MAX; ++i) adj[i] = 0.f;
            if constexpr (VARIANT <= 1) {
                // solar visibility branch (G_NeRF.py:100-108)
                Frag ps[PESUN_KS];
                make_pe_sun(s0, s1, s2, h, ps);
                Frag sA[KW2], sB[KW2];
                LAYER(F_S1, W2 / 32, KW2, PESUN_KS, true, x1f, ps, sA, nullptr);
                LAYER(F_S2, W2 / 32, KW2, 0, true, sA, nullptr, sB, nullptr);
                LAYER(F_S3, W2 / 32, KW2, 0, true, sB, nullptr, sA, nullptr);
                LAYER(F_S4, 1, KW2, 0, false, sA, nullptr, nullptr, &raw);
                sv_raw = raw[0];
            }
            if constexpr (VARIANT == 0) {
                // seasonal colour-adjust branch (T_NeRF_net_v2.py:83-87)
                LAYER(F_A1, W / 32, KW2, 0, true, x1f, nullptr, hA, nullptr);
                LAYER(F_A2, W / 32, KW, 0, true, hA, nullptr, hB, nullptr);
                LAYER(F_A3, W / 32, KW, 0, true, hB, nullptr, hA, nullptr);
                LAYER(F_AC, 1, KW, 0, false, hA, nullptr, nullptr, &raw);
#pragma unroll
                for (int i = 0; i < 3 * C_MAX; ++i) adj[i] = raw[i];
            }
#undef LAYER
            // ---- output non-linearities (T_NeRF_net_v2.py:91-98), lane-half 0 holds the head rows
            if constexpr (VARIANT == 3) {
                raysum_add(rs, A, tile, 4, wave, pass, lane, rho_raw, x0, x1, x2);
                if (++pass == passes || raysum_saturated(rs, A, tile * 4 + wave, wave, 4, lane, (__attribute__((address_space(3))) float*)(bias_lds + A.bias_floats))) {
                    raysum_end(rs, A, tile, 4, wave, lane);
                    pass = 0;
                    tile += gridDim.x;
                }
            } else {
                if (h == 0 && valid) store_field_outputs<VARIANT>(A.out, n, C, x0, x1, x2, col_r, col_g, col_b, rho_raw, sv_raw, adj, pcls);
                tile += gridDim.x;
            }
        } else {
            // ---- group program: class softmax (T_NeRF_net_v2.py:77-78) and sky colour (G_NeRF.py:110-111)
            constexpr int KW = W / 16, W4P = pad32(W / 4), KW4 = W4P / 16;
            f32x16 raw;
#define LAYER(L, NBv, K0, K1, SINv, IN0, IN1, OUT, RAW)                                                              \
    run_layer<NBv, K0, K1, SINv>(rg, A.stream, A.stream_bytes, lds, bias_lds + prog_bias_start(PROG_GROUP, W, C_MAX, L), \
                                 IN0, IN1, OUT, RAW, wave, lane)
            const float t0 = A.time[nc * 4], t1 = A.time[nc * 4 + 1];
            const float s0 = A.sun[nc * 3], s1 = A.sun[nc * 3 + 1], s2 = A.sun[nc * 3 + 2];
            Frag pt[PETIME_KS];
            make_pe_time(t0, t1, h, pt);
            Frag hA[KW], hB[KW];
            LAYER(G_T1, W / 32, PETIME_KS, 0, true, pt, nullptr, hA, nullptr);
            LAYER(G_T2, W / 32, KW, 0, true, hA, nullptr, hB, nullptr);
            LAYER(G_CL, 1, KW, 0, false, hB, nullptr, nullptr, &raw);
            float logit[C_MAX];
#pragma unroll
            for (int c = 0; c < C_MAX; ++c) logit[c] = raw[c];
            Frag ps[PESUN_KS];
            make_pe_sun(s0, s1, s2, h, ps);
            Frag kA[KW4];
            LAYER(G_K1, W4P / 32, PESUN_KS, 0, true, ps, nullptr, kA, nullptr);
            LAYER(G_K2, 1, KW4, 0, false, kA, nullptr, nullptr, &raw);
#undef LAYER
            if (h == 0 && valid) {
                float m = -3.0e38f;
#pragma unroll
                for (int c = 0; c < C_MAX; ++c) if (c < C) m = fmaxf(m, logit[c]);
                float e[C_MAX], sum = 0.f;
#pragma unroll
                for (int c = 0; c < C_MAX; ++c) { e[c] = c < C ? expf(logit[c] - m) : 0.f; sum += e[c]; }
#pragma unroll
                for (int c = 0; c < C_MAX; ++c) if (c < C && A.g_classes) A.g_classes[n * C + c] = e[c] / sum;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    if (A.g_sky_raw) A.g_sky_raw[n * 3 + k] = raw[k];
                    if (A.g_sky) A.g_sky[n * 3 + k] = sigmoid_f(raw[k]);
                }
            }
            tile += gridDim.x;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
}

// =====================================================================================================
// compositing: one wavefront per ray (Eval_Tools_2.py:13-16,187-215; mg_run_NeRF.py:188-189)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_incl_scan(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

__global__ __launch_bounds__(256) void composite_kernel(const CompArgs A) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= A.n_rays) return;
    const int S = A.n_samples;
    const float tx = A.top[r * 3], ty = A.top[r * 3 + 1], tz = A.top[r * 3 + 2];
    const float bx = A.bot[r * 3], by = A.bot[r * 3 + 1], bz = A.bot[r * 3 + 2];
    const float dx = tx - bx, dy = ty - by, dz = tz - bz;
    // deltas = sqrt(sum((top-bot)^2)) / S   (misc.py:243)
    const float delta_ray = __fdiv_rn(__fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz))), (float)S);
    const bool classic = A.flags & 1, zero_oob = A.flags & 2;
    const float sky0 = A.sky[r * 3], sky1 = A.sky[r * 3 + 1], sky2 = A.sky[r * 3 + 2];

    float carry = 0.f, carry_m = 0.f;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, svsum = 0.f, acc = 0.f, l0 = 0.f, l1 = 0.f, l2 = 0.f, dist = 0.f;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;          // classic-solar colour
    float m0 = 0.f, m1 = 0.f, m2 = 0.f;          // prior-merged albedo (classic: merged, per-sample shaded colour)
    float ma0 = 0.f, ma1 = 0.f, ma2 = 0.f;       // classic + prior: the merged albedo itself
    for (int base = 0; base < S; base += 64) {
        const int s = base + lane;
        const bool in = s < S;
        const int64_t idx = r * S + (in ? s : S - 1);
        const float t = A.tvals[in ? s : S - 1], omt = __fsub_rn(1.f, t);
        const float px = __fadd_rn(__fmul_rn(tx, omt), __fmul_rn(bx, t));
        const float py = __fadd_rn(__fmul_rn(ty, omt), __fmul_rn(by, t));
        const float pz = __fadd_rn(__fmul_rn(tz, omt), __fmul_rn(bz, t));
        float delta = delta_ray;
        if (zero_oob && (px > 1.f || px < -1.f || py > 1.f || py < -1.f || pz > 1.f || pz < -1.f)) delta = 0.f;
        const float rho = in ? A.rho[idx] : 0.f;
        const float y = in ? rho * delta : 0.f;
        const float incl = wave_incl_scan(y, lane);
        const float excl = carry + (incl - y);
        carry += __shfl(incl, 63, 64);
        const float pv = expf(-excl);
        const float pe = 1.f - expf(-y);
        const float ps = in ? pv * pe : 0.f;
        const float sv = in ? A.solar_vis[idx] : 0.f;
        const float k0 = in ? A.col[idx * 3] : 0.f, k1 = in ? A.col[idx * 3 + 1] : 0.f, k2 = in ? A.col[idx * 3 + 2] : 0.f;
        if (in) {
            if (A.out.pv) A.out.pv[idx] = pv;
            if (A.out.pe) A.out.pe[idx] = pe;
            if (A.out.ps) A.out.ps[idx] = ps;
            if (A.out.delta) A.out.delta[idx] = delta;
        }
        a0 += ps * k0; a1 += ps * k1; a2 += ps * k2;
        svsum += ps * sv;
        acc += ps;
        l0 += ps * px; l1 += ps * py; l2 += ps * pz;
        dist += ps * (delta_ray * (float)(s + 1));
        if (classic) {
            c0 += ps * k0 * (sv + (1.f - sv) * sky0);
            c1 += ps * k1 * (sv + (1.f - sv) * sky1);
            c2 += ps * k2 * (sv + (1.f - sv) * sky2);
        }
        if (A.rho_prior) {
            const float tr = A.trust_dev ? A.trust_dev[0] : A.trust;
            const float rm = in ? (rho * tr + A.rho_prior[idx] * (1.f - tr)) : 0.f;
            const float ym = rm * delta;
            const float incl_m = wave_incl_scan(ym, lane);
            const float excl_m = carry_m + (incl_m - ym);
            carry_m += __shfl(incl_m, 63, 64);
            const float psm = in ? expf(-excl_m) * (1.f - expf(-ym)) : 0.f;
            if (classic) {
                m0 += psm * k0 * (sv + (1.f - sv) * sky0);
                m1 += psm * k1 * (sv + (1.f - sv) * sky1);
                m2 += psm * k2 * (sv + (1.f - sv) * sky2);
                ma0 += psm * k0; ma1 += psm * k1; ma2 += psm * k2;
            } else {
                m0 += psm * k0; m1 += psm * k1; m2 += psm * k2;
            }
        }
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
    svsum = wave_sum(svsum); acc = wave_sum(acc);
    l0 = wave_sum(l0); l1 = wave_sum(l1); l2 = wave_sum(l2); dist = wave_sum(dist);
    if (classic) { c0 = wave_sum(c0); c1 = wave_sum(c1); c2 = wave_sum(c2); }
    if (A.rho_prior) { m0 = wave_sum(m0); m1 = wave_sum(m1); m2 = wave_sum(m2); }
    if (A.rho_prior && classic) { ma0 = wave_sum(ma0); ma1 = wave_sum(ma1); ma2 = wave_sum(ma2); }
    if (lane == 0) {
        const float sv3 = sigmoid_f((svsum - 0.2f) * 30.f);                 // Eval_Tools_2.py:214 (un-merged PS)
        const float f0 = sv3 + (1.f - sv3) * sky0, f1 = sv3 + (1.f - sv3) * sky1, f2 = sv3 + (1.f - sv3) * sky2;
        float r0, r1, r2, al0 = a0, al1 = a1, al2 = a2;
        if (A.rho_prior) {                                                  // Rendered_Col_Merged, :243-248
            if (classic) { r0 = m0; r1 = m1; r2 = m2; al0 = ma0; al1 = ma1; al2 = ma2; } else { r0 = m0 * f0; r1 = m1 * f1; r2 = m2 * f2; al0 = m0; al1 = m1; al2 = m2; }
        } else if (classic) { r0 = c0; r1 = c1; r2 = c2; }
        else { r0 = a0 * f0; r1 = a1 * f1; r2 = a2 * f2; }
        if (A.out.rgb) { A.out.rgb[r * 3] = r0; A.out.rgb[r * 3 + 1] = r1; A.out.rgb[r * 3 + 2] = r2; }
        if (A.out.albedo) { A.out.albedo[r * 3] = al0; A.out.albedo[r * 3 + 1] = al1; A.out.albedo[r * 3 + 2] = al2; }
        if (A.out.shadow) A.out.shadow[r] = svsum;
        if (A.out.acc) A.out.acc[r] = acc;
        if (A.out.surf_loc) {
            A.out.surf_loc[r * 3] = l0 / (acc + 1e-8f); A.out.surf_loc[r * 3 + 1] = l1 / (acc + 1e-8f); A.out.surf_loc[r * 3 + 2] = l2 / (acc + 1e-8f);
        }
        if (A.out.surf_dist) A.out.surf_dist[r] = dist / acc;
    }
}

// =====================================================================================================
// seasonal sweep (mg_Img_Eval.py:123-228): the MLP ran once; for every class vector t recompute only
//   img[t,r] = sum_s PS * sigmoid(Col_raw + class_t @ Adjust)  * (shadow + (1-shadow) * sky)
// One wavefront per ray; PS by the same shuffle scan; T_CHUNK class vectors per pass (accumulators in registers).
// Reads 17 floats per sample once per T_CHUNK time-steps, but is bound by the 36 sigmoids per sample (v_exp + v_rcp),
// not by HBM: 1.7 GB in 2.2 ms = 0.78 TB/s at 512 x 512 x 96.
constexpr int T_CHUNK = 12;
// sigmoid on the transcendental unit: v_exp_f32 (base 2) + v_rcp_f32, 4 instructions instead of the ~30 of expf + an IEEE division - the sweep
// evaluates 3 + 3 T of them per sample.  Error ~2e-7 relative (1 ulp each + the rounding of x log2 e); the image tolerances are 1e-5.
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f)); }
// Lane map (round 4): the two halves of a wavefront work on the SAME 32 samples of the ray and split the T_CHUNK class vectors between them (6 each):
// no idle lanes at S = 96 (three 32-sample steps; one 64-lane step + a half-empty one before), half the accumulator registers per lane, the loads
// of both halves coalesce (same addresses); the transmittance scan runs over 32 lanes, redundantly in both halves.  The next step's 17 values per
// sample are requested before the current step is evaluated.
template <bool CLASSIC>
__global__ __launch_bounds__(256) void sweep_kernel(const SweepArgs A) {
    constexpr int TH = T_CHUNK / 2;
    const int lane = threadIdx.x & 63, sl = lane & 31, half = lane >> 5;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= A.n_rays) return;
    const int S = A.n_samples, C = A.n_classes;
    const bool explicit_delta = A.deltas != nullptr;
    float tx = 0.f, ty = 0.f, tz = 0.f, bx = 0.f, by = 0.f, bz = 0.f, delta_ray = 0.f;
    if (!explicit_delta) {
        tx = A.top[r * 3]; ty = A.top[r * 3 + 1]; tz = A.top[r * 3 + 2];
        bx = A.bot[r * 3]; by = A.bot[r * 3 + 1]; bz = A.bot[r * 3 + 2];
        const float dx = tx - bx, dy = ty - by, dz = tz - bz;
        delta_ray = __fdiv_rn(__fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz))), (float)S);
    }
    const bool zero_oob = (A.flags & 2) && !explicit_delta;
    const float sky0 = A.sky[0], sky1 = A.sky[1], sky2 = A.sky[2];
    struct In { float rho, sv, dl, k[3], ad[kMaxClasses][3]; };
    auto fetch = [&](int base) {
        In v;
        const int s = base + sl;
        const int64_t idx = r * S + (s < S ? s : S - 1);
        v.rho = A.rho[idx];
        v.sv = A.solar_vis[idx];
        v.dl = explicit_delta ? A.deltas[idx] : (zero_oob ? A.tvals[s < S ? s : S - 1] : 0.f);
#pragma unroll
        for (int k = 0; k < 3; ++k) v.k[k] = A.col_raw[idx * 3 + k];
        if (A.adjust_vec4 && kMaxClasses >= 4) {            // the usual class count: a sample's 12 adjust values are three aligned 16-byte loads
            const float4* p = reinterpret_cast<const float4*>(A.adjust + idx * 12);
            const float4 q0 = p[0], q1 = p[1], q2 = p[2];
            const float f[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
#pragma unroll
            for (int c = 0; c < kMaxClasses; ++c)
#pragma unroll
                for (int k = 0; k < 3; ++k) v.ad[c][k] = c < 4 ? f[c * 3 + k] : 0.f;
            return v;
        }
#pragma unroll
        for (int c = 0; c < kMaxClasses; ++c)
#pragma unroll
            for (int k = 0; k < 3; ++k) v.ad[c][k] = c < C ? A.adjust[(idx * C + c) * 3 + k] : 0.f;
        return v;
    };
    auto half_sum = [](float v) {
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    };
    for (int t0 = 0; t0 < A.n_times; t0 += T_CHUNK) {
        const int tb = t0 + half * TH;                     // this half's class vectors: tb .. tb + TH - 1
        float cw[TH][kMaxClasses];
#pragma unroll
        for (int t = 0; t < TH; ++t)
#pragma unroll
            for (int c = 0; c < kMaxClasses; ++c) cw[t][c] = (tb + t < A.n_times && c < C) ? A.class_vecs[(tb + t) * C + c] : 0.f;
        float acc[TH][3];
        float accc[CLASSIC ? TH : 1][3];
#pragma unroll
        for (int t = 0; t < TH; ++t) acc[t][0] = acc[t][1] = acc[t][2] = 0.f;
#pragma unroll
        for (int t = 0; t < (CLASSIC ? TH : 1); ++t) accc[t][0] = accc[t][1] = accc[t][2] = 0.f;
        float carry = 0.f, svsum = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f;
        In nxt = fetch(0);
        for (int base = 0; base < S; base += 32) {
            const In cur = nxt;
            if (base + 32 < S) nxt = fetch(base + 32);
            const int s = base + sl;
            const bool in = s < S;
            float delta = delta_ray;
            if (explicit_delta) {
                delta = cur.dl;
            } else if (zero_oob) {
                const float tt = cur.dl, omt = __fsub_rn(1.f, tt);
                const float px = __fadd_rn(__fmul_rn(tx, omt), __fmul_rn(bx, tt));
                const float py = __fadd_rn(__fmul_rn(ty, omt), __fmul_rn(by, tt));
                const float pz = __fadd_rn(__fmul_rn(tz, omt), __fmul_rn(bz, tt));
                if (px > 1.f || px < -1.f || py > 1.f || py < -1.f || pz > 1.f || pz < -1.f) delta = 0.f;
            }
            const float y = in ? cur.rho * delta : 0.f;
            float incl = y;                                // inclusive scan over the 32 lanes of the half (the same in both halves)
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) {
                const float t = __shfl_up(incl, o, 32);
                if (sl >= o) incl += t;
            }
            const float excl = carry + (incl - y);
            carry += __shfl(incl, 31, 32);
            const float ps = in ? expf(-excl) * (1.f - expf(-y)) : 0.f;
            const float sv = in ? cur.sv : 0.f;
            svsum += ps * sv;
            const float k0 = cur.k[0], k1 = cur.k[1], k2 = cur.k[2];
            b0 += ps * sigmoid_fast(k0); b1 += ps * sigmoid_fast(k1); b2 += ps * sigmoid_fast(k2);
#pragma unroll
            for (int t = 0; t < TH; ++t) {
                float m0 = 0.f, m1 = 0.f, m2 = 0.f;
#pragma unroll
                for (int c = 0; c < kMaxClasses; ++c) { m0 += cw[t][c] * cur.ad[c][0]; m1 += cw[t][c] * cur.ad[c][1]; m2 += cw[t][c] * cur.ad[c][2]; }
                const float q0 = ps * sigmoid_fast(k0 + m0), q1 = ps * sigmoid_fast(k1 + m1), q2 = ps * sigmoid_fast(k2 + m2);
                acc[t][0] += q0; acc[t][1] += q1; acc[t][2] += q2;
                if constexpr (CLASSIC) {     // per-sample shading, the use_classic_shadows branch of mg_Img_Eval.py:165-170
                    accc[t][0] += q0 * (sv + (1.f - sv) * sky0);
                    accc[t][1] += q1 * (sv + (1.f - sv) * sky1);
                    accc[t][2] += q2 * (sv + (1.f - sv) * sky2);
                }
            }
        }
        svsum = half_sum(svsum);
        b0 = half_sum(b0); b1 = half_sum(b1); b2 = half_sum(b2);
        const float mask = sigmoid_f((svsum - 0.2f) * 30.f);
        const float f0 = mask + (1.f - mask) * sky0, f1 = mask + (1.f - mask) * sky1, f2 = mask + (1.f - mask) * sky2;
#pragma unroll
        for (int t = 0; t < TH; ++t) {
            const float v0 = half_sum(acc[t][0]), v1 = half_sum(acc[t][1]), v2 = half_sum(acc[t][2]);
            float c0 = 0.f, c1 = 0.f, c2 = 0.f;
            if constexpr (CLASSIC) { c0 = half_sum(accc[t][0]); c1 = half_sum(accc[t][1]); c2 = half_sum(accc[t][2]); }
            if (sl == 0 && tb + t < A.n_times) {
                const int64_t o = ((int64_t)(tb + t) * A.n_rays + r) * 3;
                if (A.season) { A.season[o] = v0; A.season[o + 1] = v1; A.season[o + 2] = v2; }
                if (A.shaded) { A.shaded[o] = v0 * f0; A.shaded[o + 1] = v1 * f1; A.shaded[o + 2] = v2 * f2; }
                if constexpr (CLASSIC) { A.classic[o] = c0; A.classic[o + 1] = c1; A.classic[o + 2] = c2; }
            }
        }
        if (lane == 0 && t0 == 0) {
            if (A.raw_shadow) A.raw_shadow[r] = svsum;
            if (A.shadow_adjust) { A.shadow_adjust[r * 3] = f0; A.shadow_adjust[r * 3 + 1] = f1; A.shadow_adjust[r * 3 + 2] = f2; }
            if (A.base) { A.base[r * 3] = b0; A.base[r * 3 + 1] = b1; A.base[r * 3 + 2] = b2; }
        }
    }
}

hipError_t launch_sweep(const SweepArgs& a, hipStream_t st) {
    const int grid = (int)((a.n_rays + 3) / 4);
    if (a.classic) hipLaunchKernelGGL(sweep_kernel<true>, dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(sweep_kernel<false>, dim3(grid), dim3(256), 0, st, a);
    return hipGetLastError();
}

// =====================================================================================================
// ray table from a 3x4 projective camera (pre_NeRF/P_Img.py:133-147 `invert_P`, mg_Pt_holder.py:178-194): for every
// pixel (i*DS, j*DS) of the down-scaled grid, the cube xy where its ray crosses z = +1 (Top) and z = -1 (Bot): a 2x2
// solve in fp64 (as the reference), results rounded to fp32 rows [Img_Pt 2 | Top 3 | Bot 3 | View 3] + validity.
__global__ void rays_from_camera_kernel(const RayGenArgs A) {
    const int64_t n = (int64_t)A.rows * A.cols;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int pi = (int)(i / A.cols), pj = (int)(i - (int64_t)pi * A.cols);
        const double row = (double)pi * A.ds, col = (double)pj * A.ds;
        const double* P = A.P;
        const double a11 = P[0] - P[8] * row, a12 = P[1] - P[9] * row, a21 = P[4] - P[8] * col, a22 = P[5] - P[9] * col;
        const double den = a11 * a22 - a12 * a21;
        double xy[2][2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double hgt = k == 0 ? 1.0 : -1.0;
            const double c2 = P[6] * hgt + P[7] - P[10] * hgt * col - P[11] * col;     // P23 h + P24 - P33 h y - P34 y
            const double c1 = P[2] * hgt + P[3] - P[10] * hgt * row - P[11] * row;     // P13 h + P14 - P33 h x - P34 x
            xy[k][0] = (a12 * c2 - a22 * c1) / den;
            xy[k][1] = (-a11 * c2 + a21 * c1) / den;
        }
        const bool good = xy[0][0] <= 1.0 && xy[0][0] >= -1.0 && xy[0][1] <= 1.0 && xy[0][1] >= -1.0 &&
                          xy[1][0] <= 1.0 && xy[1][0] >= -1.0 && xy[1][1] <= 1.0 && xy[1][1] >= -1.0;
        if (A.valid) A.valid[i] = good ? 1 : 0;
        float* o = A.rows_out + i * 11;
        o[0] = (float)pi; o[1] = (float)pj;
        o[2] = (float)xy[0][0]; o[3] = (float)xy[0][1]; o[4] = 1.f;
        o[5] = (float)xy[1][0]; o[6] = (float)xy[1][1]; o[7] = -1.f;
        const double vx = xy[1][0] - xy[0][0], vy = xy[1][1] - xy[0][1], vz = -2.0;       // view = (bot - top) / |bot - top|
        const double vn = sqrt(vx * vx + vy * vy + vz * vz);
        o[8] = (float)(vx / vn); o[9] = (float)(vy / vn); o[10] = (float)(vz / vn);
    }
}
hipError_t launch_rays_from_camera(const RayGenArgs& a, hipStream_t st) {
    const int64_t n = (int64_t)a.rows * a.cols;
    if (n <= 0) return hipSuccess;
    int64_t b = (n + 255) / 256;
    if (b > 65536) b = 65536;
    hipLaunchKernelGGL(rays_from_camera_kernel, dim3((unsigned)b), dim3(256), 0, st, a);
    return hipGetLastError();
}

// =====================================================================================================
// Novel-view ray grids on the device, in the float64 arithmetic of the reference's numpy (one rounding per operation, no contraction: the
// library is built with -ffp-contract=off), rounded to fp32 where the reference casts: bit-identical rays.
// numpy.linspace(start, stop, num)[i] = i * ((stop - start) / (num - 1)) + start, last element = stop exactly, num == 1 -> start.
__device__ inline double linspace_at(double start, double stop, int num, int i) {
    if (num <= 1) return start;
    if (i == num - 1) return stop;
    const double step = (stop - start) / (double)(num - 1);
    return (double)i * step + start;
}
__global__ void ray_grid_kernel(const RayGridArgs A) {
    const int64_t n = A.hi - A.lo;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = A.lo + t;
        const int r = (int)(i / A.cols), c = (int)(i - (int64_t)r * A.cols);
        double tx, ty, tz, bx, by, bz;
        bool good = true;
        if (A.mode == 2) {
            // mg_Img_Eval.py:77-84: source pixel = round(linspace(0, img - 1, out)) (half to even, as numpy.round), rays by invert_P at h = +1 / -1
            const double row = rint(linspace_at(0.0, (double)(A.img_rows - 1), A.rows, r)), col = rint(linspace_at(0.0, (double)(A.img_cols - 1), A.cols, c));
            const double* P = A.P;
            const double a11 = P[0] - P[8] * row, a12 = P[1] - P[9] * row, a21 = P[4] - P[8] * col, a22 = P[5] - P[9] * col;
            const double den = a11 * a22 - a12 * a21;
            double xy[2][2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const double hgt = k == 0 ? 1.0 : -1.0;
                const double c2 = P[6] * hgt + P[7] - P[10] * hgt * col - P[11] * col;
                const double c1 = P[2] * hgt + P[3] - P[10] * hgt * row - P[11] * row;
                xy[k][0] = (a12 * c2 - a22 * c1) / den;
                xy[k][1] = (-a11 * c2 + a21 * c1) / den;
            }
            tx = xy[0][0]; ty = xy[0][1]; tz = 1.0; bx = xy[1][0]; by = xy[1][1]; bz = -1.0;
            good = tx >= -1.0 && ty <= 1.0 && bx >= -1.0 && by <= 1.0 && ty >= -1.0 && tx <= 1.0 && by >= -1.0 && bx <= 1.0;
            if (A.pix) { A.pix[t * 2] = (int32_t)row; A.pix[t * 2 + 1] = (int32_t)col; }
        } else {
            double gx, gy;
            if (A.mode == 0) {               // mg_Img_Eval.py:99-101: rows linspace(1, -1, H) (cube x), columns linspace(-1, 1, W) (cube y), z = 0
                gx = linspace_at(1.0, -1.0, A.rows, r);
                gy = linspace_at(-1.0, 1.0, A.cols, c);
            } else {                         // Quick_Run.py:82-90: XY * 2. / (size - 1) - 1, optional affine remap onto a region
                gx = ((double)r * 2.0) / (double)(A.rows - 1) - 1.0;
                gy = ((double)c * 2.0) / (double)(A.cols - 1) - 1.0;
                if (A.has_region) {
                    gx = (gx + 1.0) / 2.0 * (A.region[1] - A.region[0]) + A.region[0];
                    gy = (gy + 1.0) / 2.0 * (A.region[3] - A.region[2]) + A.region[2];
                }
            }
            tx = gx + A.q[0]; ty = gy + A.q[1]; tz = 0.0 + A.q[2];
            bx = gx - A.q[0]; by = gy - A.q[1]; bz = 0.0 - A.q[2];
            if (A.mode == 1)                 // Quick_Run.py:95: keep rays whose tops AND bots lie in [-1, 1]^3
                good = bx <= 1.0 && bx >= -1.0 && by <= 1.0 && by >= -1.0 && bz <= 1.0 && bz >= -1.0 &&
                       tx <= 1.0 && tx >= -1.0 && ty <= 1.0 && ty >= -1.0 && tz <= 1.0 && tz >= -1.0;
        }
        float* o = A.top + t * 3;
        o[0] = (float)tx; o[1] = (float)ty; o[2] = (float)tz;
        o = A.bot + t * 3;
        o[0] = (float)bx; o[1] = (float)by; o[2] = (float)bz;
        if (A.valid) A.valid[t] = good ? 1 : 0;
    }
}
hipError_t launch_ray_grid(const RayGridArgs& a, hipStream_t st) {
    const int64_t n = a.hi - a.lo;
    if (n <= 0) return hipSuccess;
    int64_t b = (n + 255) / 256;
    if (b > 65536) b = 65536;
    hipLaunchKernelGGL(ray_grid_kernel, dim3((unsigned)b), dim3(256), 0, st, a);
    return hipGetLastError();
}

// =====================================================================================================
// launchers
template <int PROG, int W, int VARIANT, bool FAST = false>
static hipError_t launch_mlp_t(const MlpArgs& a, int n_cu, hipStream_t st) {
    const int lds_bytes = RING_BYTES + a.bias_floats * 4 + kVoteBytes;
    const int64_t n_tiles = VARIANT == 3 ? (a.n + 3) / 4 : (a.n + TILE_PTS - 1) / TILE_PTS;
    int grid = (int)(n_tiles < n_cu ? n_tiles : n_cu);
    if (grid < 1) grid = 1;
    auto k = mlp_kernel<PROG, W, VARIANT, FAST>;
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds_bytes, st, a);
    return hipGetLastError();
}

hipError_t launch_mlp(int prog, int W, int variant, bool fast, const MlpArgs& a, int n_cu, hipStream_t st) {
#define CASE(Wv)                                                                                  \
    if (W == Wv) {                                                                                \
        if (prog == PROG_GROUP) return launch_mlp_t<PROG_GROUP, Wv, 0>(a, n_cu, st);              \
        if (fast && variant == 0) return launch_mlp_t<PROG_FIELD, Wv, 0, true>(a, n_cu, st);      \
        if (fast && variant == 1) return launch_mlp_t<PROG_FIELD, Wv, 1, true>(a, n_cu, st);     \
        if (fast && variant == 3) return hipErrorInvalidValue;      /* ray visibility: bf16x3 / int8 digits only */ \
        if (fast) return launch_mlp_t<PROG_FIELD, Wv, 2, true>(a, n_cu, st);                      \
        if (variant == 3) return launch_mlp_t<PROG_FIELD, Wv, 3>(a, n_cu, st);                    \
        if (variant == 0) return launch_mlp_t<PROG_FIELD, Wv, 0>(a, n_cu, st);                    \
        if (variant == 1) return launch_mlp_t<PROG_FIELD, Wv, 1>(a, n_cu, st);                    \
        return launch_mlp_t<PROG_FIELD, Wv, 2>(a, n_cu, st);                                      \
    }
    CASE(64)
    CASE(256)
#undef CASE
    return hipErrorInvalidValue;
}

int mlp_lds_bytes(int bias_floats) { return RING_BYTES + bias_floats * 4 + kVoteBytes; }
int mlp_tile_points() { return TILE_PTS; }
const char* mlp_kernel_name() { return "mlp_kernel"; }

// chunks consumed per tile by a variant of the field program (the DMA stream is cyclic over exactly these)
int field_variant_chunks(int W, int C, int variant) {
    const int last = variant == 0 ? (int)F_NUM : variant == 1 ? (int)F_A1 : (int)F_S1;      // variant 3 = the layers of variant 2
    return prog_chunk_start(PROG_FIELD, W, C, last);
}

hipError_t launch_composite(const CompArgs& a, hipStream_t st) {
    const int grid = (int)((a.n_rays + 3) / 4);
    hipLaunchKernelGGL(composite_kernel, dim3(grid), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace snerf
