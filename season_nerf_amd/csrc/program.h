// Shared description of the fused-MLP "programs" (host packer + device kernels).
//
// The Season-NeRF network (reference: T_NeRF_Full_2/T_NeRF_net_v2.py:20-105, T_NeRF_Full_2/G_NeRF.py:42-111)
// is run as two register-resident MFMA chains:
//   FIELD program  - per sample point: PE(pos) -> fc1..fc9 -> {sigma/colour head, solar branch, adjust branch}
//   GROUP program  - per (time, sun) group (a ray, or one per image): time->class softmax, sun->sky colour
//
// Data layout contract (gfx950, v_mfma_f32_32x32x16_bf16, computed transposed:  H_out^T[n x pts] = W[n x k] * H_in^T[k x pts]):
//   * weights are the MFMA A operand: one "fragment" = 32 output rows x 16 k-slots of bf16 = 1 KiB, stored
//     lane-linear (lane l = 32*h + r holds row r, k-slots 8h..8h+7 as 16 contiguous bytes), so a
//     global_load_lds_dwordx4 / ds_read_b128 pair moves it with no swizzle and no bank conflicts;
//   * every fragment exists twice, hi = bf16(w) and lo = bf16(w - hi) (3-term error-compensated product
//     hi*hi + lo*hi + hi*lo, fp32 accumulate) - a "pair" = 2 KiB, a DMA "chunk" = 8 pairs = 16 KiB;
//   * activations are the MFMA B operand and never leave registers: the 32x32 fp32 accumulator of output
//     block b (point on the lane, 16 rows in registers) becomes, after sin() and the bf16 hi/lo split, the
//     B fragments of k-steps 2b and 2b+1 of the next layer.  The price is a fixed permutation of the k order,
//     paid for at pack time:  k-slot (s, 8h+j)  <->  feature 32*(s>>1) + 16*(s&1) + 8*(j>>2) + 4*h + (j&3).
//
// Second operand format (FMT_I8, kernels_i8.hip): 16-bit fixed point in two signed int8 digits on v_mfma_i32_32x32x32_i8.
//   * an activation h in [-1,1] is q = round(32767 h) = 256 a + u (a = signed high byte, u = unsigned low byte), carried
//     as the digits (a, b = u - 128); the +128 is a per-row constant folded into the bias at pack time;
//   * a weight row n is s_n (256 T + L) with balanced digits T, L in [-128,127]; a "pair" = the T fragment and the L
//     fragment of one 32-row x 32-k tile (1 KiB each, lane l = 32h + r holds row r, 16 k-slots as 16 bytes);
//   * per pair three MFMAs: M += T a, X += T b, X += L a (L b dropped), z = sc_n (256 M + X) + bias_n;
//   * one k-step (32 slots) of the next layer = one 32x32 output block: lane-half h, byte j <-> row acc_row(j, h).
#pragma once
#include <stdint.h>

namespace snerf {

enum Format : int { FMT_BF16 = 0, FMT_I8 = 1 };
__host__ __device__ constexpr int fmt_kstep(int fmt) { return fmt == FMT_I8 ? 32 : 16; }   // k-slots per MFMA step

constexpr int kFragBytes = 1024;
constexpr int kPairBytes = 2 * kFragBytes;
#ifndef SNERF_CHUNK_PAIRS
#define SNERF_CHUNK_PAIRS 8
#endif
constexpr int kChunkPairs = SNERF_CHUNK_PAIRS;
constexpr int kChunkBytes = kChunkPairs * kPairBytes;   // 16 KiB (8 pairs) or 32 KiB (16 pairs)
constexpr int kMaxClasses = 5;                           // 3*C adjust rows must fit the 16 rows one lane owns

constexpr int PE_POS_N = 10, PE_SUN_N = 4, PE_TIME_N = 2;     // G_NeRF.py:7, T_NeRF_net_v2.py:36
constexpr int PE_POS_F = 3 * (2 * PE_POS_N + 1);              // 63
constexpr int PE_SUN_F = 3 * (2 * PE_SUN_N + 1);              // 27
constexpr int PE_TIME_F = 2 * (2 * PE_TIME_N + 1);            // 10
constexpr int PEPOS_KS = 4, PESUN_KS = 2, PETIME_KS = 2;      // k-steps (16 slots each) of the encodings
constexpr int PEPOS_KS8 = 2, PESUN_KS8 = 1, PETIME_KS8 = 1;   // ... in the int8 format (32 slots each)

enum InKind : int { IN_NONE = 0, IN_H = 1, IN_PEPOS = 2, IN_PESUN = 3, IN_PETIME = 4 };
enum OutKind : int { OUT_SIN = 0, OUT_RAW = 1 };
enum RowMap : int { ROWS_ID = 0, ROWS_HEAD = 1, ROWS_SV = 2, ROWS_ADJ = 3, ROWS_CLASS = 4, ROWS_SKY = 5 };

enum FieldLayer : int { F_FC1 = 0, F_FC2, F_FC3, F_FC4, F_FC5, F_FC6, F_FC7, F_FC8, F_FC9, F_HEAD,
                        F_S1, F_S2, F_S3, F_S4, F_A1, F_A2, F_A3, F_AC, F_NUM };
enum GroupLayer : int { G_T1 = 0, G_T2, G_CL, G_K1, G_K2, G_NUM };

struct LayerShape {
    int n_ref;      // output rows in the reference layer (before padding)
    int n_out;      // padded to a multiple of 32
    int ks0, kind0; // first input block: k-steps and kind
    int ks1, kind1; // optional second block (concatenated after the first, as the reference's torch.cat)
    int out_kind;   // OUT_SIN: sin(2*pi*acc) -> bf16 hi/lo fragments;  OUT_RAW: fp32 accumulator rows
    int row_map;
    __host__ __device__ constexpr int ks() const { return ks0 + ks1; }
    __host__ __device__ constexpr int nb() const { return n_out / 32; }
    __host__ __device__ constexpr int pairs() const { return nb() * ks(); }
    __host__ __device__ constexpr int chunks() const { return (pairs() + kChunkPairs - 1) / kChunkPairs; }
};

__host__ __device__ constexpr int pad32(int x) { return (x + 31) / 32 * 32; }

// W must be a multiple of 64 (so W/2 is a multiple of 32).  C = number of season classes.  fmt: operand format.
__host__ __device__ constexpr LayerShape field_layer(int W, int C, int l, int fmt = FMT_BF16) {
    const int W2 = W / 2, ks = fmt_kstep(fmt);
    const int pp = fmt == FMT_I8 ? PEPOS_KS8 : PEPOS_KS, ps = fmt == FMT_I8 ? PESUN_KS8 : PESUN_KS;
    switch (l) {
        case F_FC1: return {W, W, pp, IN_PEPOS, 0, IN_NONE, OUT_SIN, ROWS_ID};
        case F_FC2: case F_FC3: case F_FC4: case F_FC6: case F_FC7: case F_FC8:
            return {W, W, W / ks, IN_H, 0, IN_NONE, OUT_SIN, ROWS_ID};
        case F_FC5: return {W, W, W / ks, IN_H, pp, IN_PEPOS, OUT_SIN, ROWS_ID};
        case F_FC9: return {W2, W2, W / ks, IN_H, 0, IN_NONE, OUT_SIN, ROWS_ID};
        case F_HEAD: return {4, 32, W2 / ks, IN_H, 0, IN_NONE, OUT_RAW, ROWS_HEAD};
        case F_S1: return {W2, W2, W2 / ks, IN_H, ps, IN_PESUN, OUT_SIN, ROWS_ID};
        case F_S2: case F_S3: return {W2, W2, W2 / ks, IN_H, 0, IN_NONE, OUT_SIN, ROWS_ID};
        case F_S4: return {1, 32, W2 / ks, IN_H, 0, IN_NONE, OUT_RAW, ROWS_SV};
        case F_A1: return {W, W, W2 / ks, IN_H, 0, IN_NONE, OUT_SIN, ROWS_ID};
        case F_A2: case F_A3: return {W, W, W / ks, IN_H, 0, IN_NONE, OUT_SIN, ROWS_ID};
        case F_AC: return {3 * C, 32, W / ks, IN_H, 0, IN_NONE, OUT_RAW, ROWS_ADJ};
        default: return {0, 0, 0, 0, 0, 0, 0, 0};
    }
}

__host__ __device__ constexpr LayerShape group_layer(int W, int C, int l, int fmt = FMT_BF16) {
    const int W4 = W / 4, ks = fmt_kstep(fmt);
    const int pt = fmt == FMT_I8 ? PETIME_KS8 : PETIME_KS, ps = fmt == FMT_I8 ? PESUN_KS8 : PESUN_KS;
    switch (l) {
        case G_T1: return {W, W, pt, IN_PETIME, 0, IN_NONE, OUT_SIN, ROWS_ID};
        case G_T2: return {W, W, W / ks, IN_H, 0, IN_NONE, OUT_SIN, ROWS_ID};
        case G_CL: return {C, 32, W / ks, IN_H, 0, IN_NONE, OUT_RAW, ROWS_CLASS};
        case G_K1: return {W4, pad32(W4), ps, IN_PESUN, 0, IN_NONE, OUT_SIN, ROWS_ID};
        case G_K2: return {3, 32, pad32(W4) / ks, IN_H, 0, IN_NONE, OUT_RAW, ROWS_SKY};
        default: return {0, 0, 0, 0, 0, 0, 0, 0};
    }
}

enum Program : int { PROG_FIELD = 0, PROG_GROUP = 1 };
__host__ __device__ constexpr int prog_layers(int prog) { return prog == PROG_FIELD ? (int)F_NUM : (int)G_NUM; }
__host__ __device__ constexpr LayerShape prog_layer(int prog, int W, int C, int l, int fmt = FMT_BF16) {
    return prog == PROG_FIELD ? field_layer(W, C, l, fmt) : group_layer(W, C, l, fmt);
}
// chunk index at which layer l starts (every layer starts on a chunk boundary), and bias-table offset (floats)
__host__ __device__ constexpr int prog_chunk_start(int prog, int W, int C, int l, int fmt = FMT_BF16) {
    int c = 0;
    for (int i = 0; i < l; ++i) c += prog_layer(prog, W, C, i, fmt).chunks();
    return c;
}
__host__ __device__ constexpr int prog_bias_start(int prog, int W, int C, int l) {
    int c = 0;
    for (int i = 0; i < l; ++i) c += prog_layer(prog, W, C, i).n_out;
    return c;
}
__host__ __device__ constexpr int prog_chunks(int prog, int W, int C, int fmt = FMT_BF16) { return prog_chunk_start(prog, W, C, prog_layers(prog), fmt); }
__host__ __device__ constexpr int prog_bias_floats(int prog, int W, int C) { return prog_bias_start(prog, W, C, prog_layers(prog)); }
// int8 format: per output row a scale and a bias, stored [block][lane-half][16 scales | 16 biases] = 2 floats per row.
// Layers that read an encoding take its RAW coordinates (x, y, z of the point / the sun vector; t of the time code) not as
// digits but in fp32 - three FMAs per element in the epilogue - so that the int8 mode has no input range: the digit slots of
// the raw features carry zero weights, and a second table [block][lane-half][quad][dim 0..2][4 elements] = 3 floats per row
// follows the layer's scale / bias table.
__host__ __device__ constexpr int raw_kind(const LayerShape& s) {
    return (s.kind0 == IN_PEPOS || s.kind0 == IN_PESUN || s.kind0 == IN_PETIME) ? s.kind0
         : (s.kind1 == IN_PEPOS || s.kind1 == IN_PESUN || s.kind1 == IN_PETIME) ? s.kind1 : (int)IN_NONE;
}
__host__ __device__ constexpr int layer_table_floats(const LayerShape& s) { return (raw_kind(s) != IN_NONE ? 5 : 2) * s.n_out; }
__host__ __device__ constexpr int prog_table_start(int prog, int W, int C, int l) {
    int c = 0;
    for (int i = 0; i < l; ++i) c += layer_table_floats(prog_layer(prog, W, C, i, FMT_I8));
    return c;
}
__host__ __device__ constexpr int prog_table_floats(int prog, int W, int C) { return prog_table_start(prog, W, C, prog_layers(prog)); }

// ---- K-split stream of the field program (kernels_ks.hip, W = 512 bf16x3): the pairs of the bf16 stream in the order the two waves of a pair
// consume them.  Per parity a layer is a sequence of pairs: a raw head ks0/2 of them (its K-half); a hidden layer NBH = nb/2 steps of
// [F: ks0/2 pairs of the partner's block over the own K-half | O: ks0/2 + ks1 pairs of the own block: own K-half, then the encoding in full];
// an encoding-only layer (fc1) NBH steps of ks0 pairs (own blocks).  A 16 KiB chunk holds 4 pairs of parity 0, then 4 pairs of parity 1.
constexpr int kKsParPairs = kChunkPairs / 2;
__host__ __device__ constexpr int ks_layer_pairs(const LayerShape& s) {       // pairs per parity
    return s.out_kind == OUT_RAW ? s.ks0 / 2 : s.kind0 == IN_H ? (s.nb() / 2) * (s.ks0 + s.ks1) : (s.nb() / 2) * s.ks0;
}
__host__ __device__ constexpr int ks_layer_chunks(const LayerShape& s) { return (ks_layer_pairs(s) + kKsParPairs - 1) / kKsParPairs; }
__host__ __device__ constexpr int ks_chunk_start(int W, int C, int l) {
    int c = 0;
    for (int i = 0; i < l; ++i) c += ks_layer_chunks(field_layer(W, C, i));
    return c;
}
__host__ __device__ constexpr int ks_chunk_start_g(int W, int C, int l) {      // the per-ray (group) program in the same order
    int c = 0;
    for (int i = 0; i < l; ++i) c += ks_layer_chunks(group_layer(W, C, i));
    return c;
}
// canonical pair (block, k-step) behind pair q of parity a of a layer; returns block * 4096 + k-step
__host__ __device__ constexpr int ks_pair_source(const LayerShape& s, int a, int q) {
    if (s.out_kind == OUT_RAW) return a * (s.ks0 / 2) + q;
    const int nbh = s.nb() / 2;
    if (s.kind0 != IN_H) return (a * nbh + q / s.ks0) * 4096 + q % s.ks0;
    const int ksh = s.ks0 / 2, sl = s.ks0 + s.ks1, i = q / sl, r = q % sl;
    if (r < ksh) return ((1 - a) * nbh + i) * 4096 + a * ksh + r;                    // F phase
    const int t = r - ksh;                                                         // O phase
    return (a * nbh + i) * 4096 + (t < ksh ? a * ksh + t : s.ks0 + (t - ksh));
}

// accumulator register i of lane-half h  <->  row of the 32-row output block
__host__ __device__ constexpr int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// ---- k-slot -> input feature maps.  -1 = zero padding.
// Encodings are produced per lane as an array v[e]; lane-half h of a point evaluates its own share:
// PE(pos), reference feature order (misc.py:114-139): [x0 x1 x2 | per d: cos(k_0..k_9 x_d), sin(k_0..k_9 x_d)].
//   lane-half h evaluates frequencies 5h..5h+4 of all three coordinates, e in [0,32):
//   e < 30: d = e/10, r = e%10, freq = 5h + r/2, (r&1 ? sin : cos);  e = 30,31: raw x (h=0: x0,x1; h=1: x2, pad)
__host__ __device__ constexpr int pepos_feature(int e, int h) {
    if (e < 30) { const int d = e / 10, r = e % 10; return 3 + 20 * d + 10 * (r & 1) + 5 * h + r / 2; }
    if (e == 30) return h == 0 ? 0 : 2;
    return h == 0 ? 1 : -1;
}
// PE(sun), n = 4, e in [0,16): e < 12: d = e/4, r = e%4, freq = 2h + r/2; e = 12,13 raw; 14,15 pad
__host__ __device__ constexpr int pesun_feature(int e, int h) {
    if (e < 12) { const int d = e / 4, r = e % 4; return 3 + 8 * d + 4 * (r & 1) + 2 * h + r / 2; }
    if (e == 12) return h == 0 ? 0 : 2;
    if (e == 13) return h == 0 ? 1 : -1;
    return -1;
}
// PE(time[:, 0:2]), n = 2: lane-half h owns coordinate d = h: e = 0 raw, 1 cos k0, 2 sin k0, 3 cos k1, 4 sin k1, rest pad
__host__ __device__ constexpr int petime_feature(int e, int h) {
    if (e > 4) return -1;
    if (e == 0) return h;
    const int q = e - 1;                       // 0 cos k0, 1 sin k0, 2 cos k1, 3 sin k1
    return 2 + 4 * h + 2 * (q & 1) + (q >> 1);
}
// bf16 format: kk = 16*s + 8*h + j is the K index inside a block (v index e = 8 s + j).
__host__ __device__ constexpr int slot_feature_H(int kk) {
    const int s = kk / 16, h = (kk % 16) / 8, j = kk % 8;
    return 32 * (s >> 1) + 16 * (s & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
}
__host__ __device__ constexpr int slot_feature_PEPOS(int kk) { return pepos_feature(8 * (kk / 16) + kk % 8, (kk % 16) / 8); }
__host__ __device__ constexpr int slot_feature_PESUN(int kk) { return pesun_feature(8 * (kk / 16) + kk % 8, (kk % 16) / 8); }
__host__ __device__ constexpr int slot_feature_PETIME(int kk) { return kk / 16 != 0 ? -1 : petime_feature(kk % 8, (kk % 16) / 8); }
__host__ __device__ constexpr int slot_feature(int kind, int kk) {
    return kind == IN_H ? slot_feature_H(kk) : kind == IN_PEPOS ? slot_feature_PEPOS(kk)
         : kind == IN_PESUN ? slot_feature_PESUN(kk) : kind == IN_PETIME ? slot_feature_PETIME(kk) : -1;
}
// int8 format: k-step s, lane-half h, byte j (v index e = 16 s + j); a hidden k-step is one output block of the producer.
// The raw coordinates of an encoding (reference features 0..2 of PE(pos) / PE(sun), 0..1 of PE(time)) have no digit slot
// (-1: zero weight): they enter in fp32, see prog_table_start.
__host__ __device__ constexpr int raw_dims(int kind) { return kind == IN_PETIME ? 2 : (kind == IN_PEPOS || kind == IN_PESUN) ? 3 : 0; }
__host__ __device__ constexpr int slot_feature8(int kind, int s, int h, int j) {
    const int f = kind == IN_H ? 32 * s + acc_row(j, h) : kind == IN_PEPOS ? pepos_feature(16 * s + j, h)
                : kind == IN_PESUN ? (s == 0 ? pesun_feature(j, h) : -1) : kind == IN_PETIME ? (s == 0 ? petime_feature(j, h) : -1) : -1;
    return (kind != IN_H && f >= 0 && f < raw_dims(kind)) ? -1 : f;
}
__host__ __device__ constexpr int kind_features(int kind, int ks, int fmt = FMT_BF16) {
    return kind == IN_H ? fmt_kstep(fmt) * ks : kind == IN_PEPOS ? PE_POS_F : kind == IN_PESUN ? PE_SUN_F
         : kind == IN_PETIME ? PE_TIME_F : 0;
}

}  // namespace snerf
