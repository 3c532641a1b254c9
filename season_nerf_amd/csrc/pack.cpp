// Host-side weight packer: reference state_dict tensors -> the MFMA fragment streams of program.h.
//
// What is folded at pack time (fp64 arithmetic, then one rounding):
//   * eval-mode BatchNorm1d (misc.py:169-170,188-189):  sin(BN(30*(Wx+b)))  ==  sin(2*pi*(W''x + b''))
//       a = gamma/sqrt(running_var+eps),  W'' = a*30*W/(2*pi),  b'' = (a*(30*b - running_mean) + beta)/(2*pi)
//     (v_sin_f32 takes revolutions, so the 1/(2*pi) is free);
//   * the k-order permutation of the register-resident chain and the row maps of the head layers;
//   * the bf16 hi/lo split  w = hi + lo  (hi = RNE bf16(w), lo = RNE bf16(w - hi)).
// Pure host code: no HIP calls, usable (and unit-tested) on a machine without a GPU.
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <thread>
#include "pack.h"

#include <cmath>
#include <cstring>

namespace snerf {

// The layers of a program pack independently (disjoint regions of the stream and the tables): a few host threads take them from a counter.  A re-pack after
// a parameter change sits in front of every in-loop validation render (bench.py `repack_ms`); SNERF_PACK_THREADS=1 packs serially.
template <class F>
static bool parallel_layers(int L, std::string* err, F fn) {
    static const int cap = [] {                    // initialised once, thread-safely (two Python threads may pack at the same time)
        const char* e = getenv("SNERF_PACK_THREADS");
        const unsigned hw = std::thread::hardware_concurrency();
        const int c = e ? atoi(e) : (int)(hw ? (hw < 8 ? hw : 8) : 1);
        return c < 1 ? 1 : c;
    }();
    const int T = cap < L ? cap : L;
    std::atomic<int> next{0};
    std::atomic<bool> ok{true};
    std::mutex mu;
    auto work = [&]() {
        for (;;) {
            const int l = next.fetch_add(1);
            if (l >= L || !ok.load()) break;
            std::string e;
            bool good = false;
            try {                                  // an exception in a worker thread (bad_alloc in a layer's buffers) must become an error, not std::terminate
                good = fn(l, &e);
            } catch (const std::exception& ex) {
                e = std::string("packing layer ") + std::to_string(l) + ": " + ex.what();
            } catch (...) {
                e = "packing layer " + std::to_string(l) + ": unknown exception";
            }
            if (!good) {
                std::lock_guard<std::mutex> g(mu);
                if (ok.exchange(false)) *err = e;
            }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    return ok.load();
}

static inline uint16_t bf16_rne(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)(u >> 16);   // inf / nan: truncate
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf16_to_f32(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

const Tensor* Weights::find(const std::string& k) const {
    auto it = t.find(k);
    return it == t.end() ? nullptr : &it->second;
}

// ---- reference layer sources --------------------------------------------------------------------------
struct Src {
    const char* prefix;   // state_dict key prefix
    bool sine;            // SineLayer (weights under ".linear.") vs plain Linear
};

static const char* field_prefix(int l) {
    static const char* p[F_NUM] = {
        "G_NeRF_net.fc1", "G_NeRF_net.fc2", "G_NeRF_net.fc3", "G_NeRF_net.fc4", "G_NeRF_net.fc5", "G_NeRF_net.fc6",
        "G_NeRF_net.fc7", "G_NeRF_net.fc8", "G_NeRF_net.fc9", nullptr /*head: two sources*/,
        "G_NeRF_net.fc_solar_1", "G_NeRF_net.fc_solar_2", "G_NeRF_net.fc_solar_3", "G_NeRF_net.fc_solar_4",
        "adjust_layer_1", "adjust_layer_2", "adjust_layer_3", "adjust_col"};
    return p[l];
}
static const char* group_prefix(int l) {
    static const char* p[G_NUM] = {"time_layer_1", "time_layer_2", "get_class_layer", "G_NeRF_net.fc_sky_color_1",
                                   "G_NeRF_net.fc_sky_color_2"};
    return p[l];
}

// Dense folded layer in *reference* feature order: Wd[n_ref][k_ref], bd[n_ref]
struct Dense {
    int n = 0, k = 0;
    std::vector<double> W, b;
};

static bool fetch(const Weights& w, const std::string& key, size_t numel, const float** out, std::string* err) {
    const Tensor* t = w.find(key);
    if (!t) { *err = "missing tensor: " + key; return false; }
    if (t->data.size() != numel) {
        *err = "tensor " + key + " has " + std::to_string(t->data.size()) + " elements, expected " + std::to_string(numel);
        return false;
    }
    *out = t->data.data();
    return true;
}

static bool dense_sine(const Weights& w, const std::string& pre, int n, int k, bool fold_bn, Dense* d, std::string* err) {
    const float *W, *b;
    if (!fetch(w, pre + ".linear.weight", (size_t)n * k, &W, err)) return false;
    if (!fetch(w, pre + ".linear.bias", (size_t)n, &b, err)) return false;
    const bool has_bn = w.find(pre + ".norm.weight") != nullptr;
    const float *g = nullptr, *beta = nullptr, *mu = nullptr, *var = nullptr;
    if (has_bn) {
        if (!fold_bn) { *err = "layer " + pre + " has BatchNorm; only the folded (eval-mode) program is packed"; return false; }
        if (!fetch(w, pre + ".norm.weight", n, &g, err) || !fetch(w, pre + ".norm.bias", n, &beta, err) ||
            !fetch(w, pre + ".norm.running_mean", n, &mu, err) || !fetch(w, pre + ".norm.running_var", n, &var, err))
            return false;
    }
    const double inv2pi = 1.0 / (2.0 * M_PI);
    d->n = n; d->k = k;
    d->W.assign((size_t)n * k, 0.0); d->b.assign(n, 0.0);
    for (int r = 0; r < n; ++r) {
        double a = 1.0, shift = 0.0, off = 0.0;
        if (has_bn) { a = (double)g[r] / std::sqrt((double)var[r] + 1e-5); shift = mu[r]; off = beta[r]; }
        for (int c = 0; c < k; ++c) d->W[(size_t)r * k + c] = a * 30.0 * (double)W[(size_t)r * k + c] * inv2pi;
        d->b[r] = (a * (30.0 * (double)b[r] - shift) + off) * inv2pi;
    }
    return true;
}

static bool dense_linear(const Weights& w, const std::string& pre, int n, int k, Dense* d, std::string* err) {
    const float *W, *b;
    if (!fetch(w, pre + ".weight", (size_t)n * k, &W, err)) return false;
    if (!fetch(w, pre + ".bias", (size_t)n, &b, err)) return false;
    d->n = n; d->k = k;
    d->W.assign(W, W + (size_t)n * k);
    d->b.assign(b, b + n);
    return true;
}

// padded output row -> reference row (or -1)
static int ref_row(const LayerShape& s, int n) {
    const int b = n / 32, r = n % 32;
    switch (s.row_map) {
        case ROWS_ID: return n < s.n_ref ? n : -1;
        case ROWS_HEAD: return (b == 0 && r < 4) ? r : -1;               // col r,g,b, sigma -> acc regs 0..3 of lane-half 0
        case ROWS_SV: case ROWS_SKY: return (b == 0 && r < s.n_ref) ? r : -1;
        case ROWS_ADJ: case ROWS_CLASS:                                      // output i -> acc reg i of lane-half 0
            if (b != 0) return -1;
            for (int i = 0; i < s.n_ref && i < 16; ++i) if (acc_row(i, 0) == r) return i;
            return -1;
    }
    return -1;
}

// dense folded layer l of a program in reference feature order
static bool dense_layer(const Weights& w, int prog, int W, int C, int l, bool fold_bn, Dense* dp, std::string* err) {
    Dense& d = *dp;
    const LayerShape s = prog_layer(prog, W, C, l);
    const int k_ref = kind_features(s.kind0, s.ks0) + kind_features(s.kind1, s.ks1);
    if (prog == PROG_FIELD && l == F_HEAD) {
        Dense c3, s1;
        if (!dense_linear(w, "G_NeRF_net.fc10Col", 3, k_ref, &c3, err)) return false;
        if (!dense_linear(w, "G_NeRF_net.fc10Sigma", 1, k_ref, &s1, err)) return false;
        d.n = 4; d.k = k_ref; d.W = c3.W; d.b = c3.b;
        d.W.insert(d.W.end(), s1.W.begin(), s1.W.end()); d.b.push_back(s1.b[0]);
    } else {
        const std::string pre = prog == PROG_FIELD ? field_prefix(l) : group_prefix(l);
        // input width of the reference layer: an IN_H block narrower than its k-steps means zero-padded features
        int kr = k_ref;
        if (prog == PROG_GROUP && l == G_K2) kr = W / 4;          // fc_sky_color_2 reads W/4 features (padded to 32)
        if (s.out_kind == OUT_SIN) { if (!dense_sine(w, pre, s.n_ref, kr, fold_bn, &d, err)) return false; }
        else { if (!dense_linear(w, pre, s.n_ref, kr, &d, err)) return false; }
    }
    return true;
}

bool pack_program(const Weights& w, int prog, int W, int C, bool fold_bn, Packed* out, std::string* err) {
    if (W < 64 || W % 64 != 0) { *err = "layer width must be a multiple of 64 (got " + std::to_string(W) + ")"; return false; }
    if (C < 1 || C > kMaxClasses) { *err = "n_classes must be in [1," + std::to_string(kMaxClasses) + "]"; return false; }
    const int L = prog_layers(prog);
    out->stream.assign((size_t)prog_chunks(prog, W, C) * kChunkBytes, 0);
    out->bias.assign((size_t)prog_bias_floats(prog, W, C), 0.f);
    return parallel_layers(L, err, [&](int l, std::string* err) {
        const LayerShape s = prog_layer(prog, W, C, l);
        Dense d;
        if (!dense_layer(w, prog, W, C, l, fold_bn, &d, err)) return false;
        // ---- bias table in accumulator order: [block][lane-half][reg]
        float* bias = out->bias.data() + prog_bias_start(prog, W, C, l);
        for (int b = 0; b < s.nb(); ++b)
            for (int h = 0; h < 2; ++h)
                for (int i = 0; i < 16; ++i) {
                    const int rr = ref_row(s, 32 * b + acc_row(i, h));
                    bias[b * 32 + h * 16 + i] = rr >= 0 ? (float)d.b[rr] : 0.f;
                }
        // ---- fragments
        uint8_t* base = out->stream.data() + (size_t)prog_chunk_start(prog, W, C, l) * kChunkBytes;
        const int f0 = kind_features(s.kind0, s.ks0);
        for (int b = 0; b < s.nb(); ++b)
            for (int ks = 0; ks < s.ks(); ++ks) {
                uint16_t* hi = (uint16_t*)(base + (size_t)(b * s.ks() + ks) * kPairBytes);
                uint16_t* lo = hi + kFragBytes / 2;
                for (int lane = 0; lane < 64; ++lane) {
                    const int r = lane & 31, h = lane >> 5;
                    const int rr = ref_row(s, 32 * b + r);
                    for (int j = 0; j < 8; ++j) {
                        double v = 0.0;
                        if (rr >= 0) {
                            int col = -1;
                            if (ks < s.ks0) { const int f = slot_feature(s.kind0, 16 * ks + 8 * h + j); if (f >= 0) col = f; }
                            else { const int f = slot_feature(s.kind1, 16 * (ks - s.ks0) + 8 * h + j); if (f >= 0) col = f0 + f; }
                            if (col >= 0 && col < d.k) v = d.W[(size_t)rr * d.k + col];
                        }
                        const float vf = (float)v;
                        const uint16_t vh = bf16_rne(vf);
                        hi[lane * 8 + j] = vh;
                        lo[lane * 8 + j] = bf16_rne((float)(v - (double)bf16_to_f32(vh)));
                    }
                }
            }
        return true;
    });
}

// ---- K-split order (program.h ks_*): what the wave pairs of kernels_ks.hip consume ---------------------------------------------
bool permute_program_ks(const Packed& canon, int W, int C, Packed* out, std::string* err, int prog) {
    if (W % 256 != 0) { *err = "the K-split kernels need a width that is a multiple of 256 (got " + std::to_string(W) + ")"; return false; }
    if (canon.stream.size() != (size_t)prog_chunks(prog, W, C) * kChunkBytes) { *err = "permute_program_ks: not a packed program of this kind and width"; return false; }
    const int L = prog_layers(prog);
    auto start = [&](int l) { return prog == PROG_FIELD ? ks_chunk_start(W, C, l) : ks_chunk_start_g(W, C, l); };
    out->bias = canon.bias;
    out->stream.assign((size_t)start(L) * kChunkBytes, 0);
    for (int l = 0; l < L; ++l) {
        const LayerShape s = prog_layer(prog, W, C, l);
        const uint8_t* src = canon.stream.data() + (size_t)prog_chunk_start(prog, W, C, l) * kChunkBytes;
        uint8_t* dst = out->stream.data() + (size_t)start(l) * kChunkBytes;
        const int np = ks_layer_pairs(s);
        for (int a = 0; a < 2; ++a)
            for (int q = 0; q < np; ++q) {
                const int sp = ks_pair_source(s, a, q), b = sp / 4096, ks = sp % 4096;
                std::memcpy(dst + (size_t)(q / kKsParPairs) * kChunkBytes + (size_t)a * kKsParPairs * kPairBytes + (size_t)(q % kKsParPairs) * kPairBytes,
                            src + (size_t)(b * s.ks() + ks) * kPairBytes, kPairBytes);
            }
    }
    return true;
}

// ---- int8-digit format (program.h, FMT_I8) ----------------------------------------------------------------
// Row n of a folded layer becomes s_n * (256 T + L): wq = round(w / s_n), s_n = max_k |w| / 32512, balanced digits
// T = floor((wq + 128) / 256), L = wq - 256 T, both in [-128,127].  The activations arrive as q = 256 a + b + 128
// (q = round(32767 h)), so  sum_k w_k h_k  ~  s_n / 32767 * (256 (256 M + X) + 128 sum_k wq_k),  M = sum T a,
// X = sum (T b + L a)  (the L b term, <= 2^-16 of full scale, is dropped).  Table per row: scale 256 s_n / 32767 and
// bias b_n + 128 s_n sum_k wq_k / 32767.  The raw coordinates of an encoding stay out of the digits (program.h): their
// weights go to the layer's fp32 raw table and do not take part in the row scale.
bool pack_program_i8(const Weights& w, int prog, int W, int C, bool fold_bn, Packed* out, std::string* err) {
    if (W < 64 || W % 64 != 0) { *err = "layer width must be a multiple of 64 (got " + std::to_string(W) + ")"; return false; }
    if (C < 1 || C > kMaxClasses) { *err = "n_classes must be in [1," + std::to_string(kMaxClasses) + "]"; return false; }
    const int L = prog_layers(prog);
    out->stream.assign((size_t)prog_chunks(prog, W, C, FMT_I8) * kChunkBytes, 0);
    out->bias.assign((size_t)prog_table_floats(prog, W, C), 0.f);
    return parallel_layers(L, err, [&](int l, std::string* err) {
        const LayerShape s = prog_layer(prog, W, C, l, FMT_I8);
        Dense d;
        if (!dense_layer(w, prog, W, C, l, fold_bn, &d, err)) return false;
        const int f0 = kind_features(s.kind0, s.ks0, FMT_I8);
        // column of the dense layer behind k-slot (ks, h, j), or -1
        auto column = [&](int ks, int h, int j) {
            int col = -1;
            if (ks < s.ks0) { const int f = slot_feature8(s.kind0, ks, h, j); if (f >= 0) col = f; }
            else { const int f = slot_feature8(s.kind1, ks - s.ks0, h, j); if (f >= 0) col = f0 + f; }
            return (col >= 0 && col < d.k) ? col : -1;
        };
        // columns of the raw coordinates (fp32 path): none of the digit slots maps to them
        const int rk = raw_kind(s), rdims = raw_dims(rk), rbase = (rk != IN_NONE && rk == s.kind1 && rk != s.kind0) ? f0 : 0;
        auto is_raw_col = [&](int c) { return rk != IN_NONE && c >= rbase && c < rbase + rdims; };
        // quantise every padded output row
        const int n_pad = s.n_out;
        std::vector<double> scale(n_pad, 0.0);
        std::vector<long long> wsum(n_pad, 0);
        std::vector<int> wq((size_t)n_pad * d.k, 0);
        for (int n = 0; n < n_pad; ++n) {
            const int rr = ref_row(s, n);
            if (rr < 0) continue;
            double mx = 0.0;
            for (int c = 0; c < d.k; ++c) if (!is_raw_col(c)) mx = std::fmax(mx, std::fabs(d.W[(size_t)rr * d.k + c]));
            const double sn = mx > 0.0 ? mx / 32512.0 : 1.0;
            scale[n] = sn;
            for (int c = 0; c < d.k; ++c) wq[(size_t)n * d.k + c] = (int)std::llround(d.W[(size_t)rr * d.k + c] / sn);
        }
        uint8_t* base = out->stream.data() + (size_t)prog_chunk_start(prog, W, C, l, FMT_I8) * kChunkBytes;
        for (int b = 0; b < s.nb(); ++b)
            for (int ks = 0; ks < s.ks(); ++ks) {
                int8_t* Tf = (int8_t*)(base + (size_t)(b * s.ks() + ks) * kPairBytes);
                int8_t* Lf = Tf + kFragBytes;
                for (int lane = 0; lane < 64; ++lane) {
                    const int r = lane & 31, h = lane >> 5, n = 32 * b + r;
                    for (int j = 0; j < 16; ++j) {
                        const int col = column(ks, h, j);
                        int q = 0;
                        if (col >= 0 && ref_row(s, n) >= 0) { q = wq[(size_t)n * d.k + col]; wsum[n] += q; }
                        const int T = (q + 128) >> 8;                 // floor division (arithmetic shift), q in [-32512, 32512]
                        Tf[lane * 16 + j] = (int8_t)T;
                        Lf[lane * 16 + j] = (int8_t)(q - 256 * T);
                    }
                }
            }
        float* tab = out->bias.data() + prog_table_start(prog, W, C, l);
        for (int b = 0; b < s.nb(); ++b)
            for (int h = 0; h < 2; ++h)
                for (int i = 0; i < 16; ++i) {
                    const int n = 32 * b + acc_row(i, h), rr = ref_row(s, n);
                    float* t = tab + (b * 2 + h) * 32;
                    t[i] = rr >= 0 ? (float)(256.0 * scale[n] / 32767.0) : 0.f;
                    t[16 + i] = rr >= 0 ? (float)(d.b[rr] + 128.0 * scale[n] * (double)wsum[n] / 32767.0) : 0.f;
                }
        if (rk != IN_NONE) {      // fp32 weights of the raw coordinates: [block][lane-half][quad][dim][4 elements]
            float* rt = tab + 2 * s.n_out;
            for (int b = 0; b < s.nb(); ++b)
                for (int h = 0; h < 2; ++h)
                    for (int i = 0; i < 16; ++i) {
                        const int n = 32 * b + acc_row(i, h), rr = ref_row(s, n);
                        for (int dd = 0; dd < 3; ++dd)
                            rt[(((b * 2 + h) * 4 + i / 4) * 3 + dd) * 4 + i % 4] =
                                (rr >= 0 && dd < rdims && rbase + dd < d.k) ? (float)d.W[(size_t)rr * d.k + rbase + dd] : 0.f;
                    }
        }
        return true;
    });
}

// ---- error model of the int8-digit format ------------------------------------------------------------------
// Input of layer l of a program: index of the layer whose activations form its IN_H block, or -1 (encodings only).
static int hidden_source(int prog, int l) {
    if (prog == PROG_FIELD) {
        switch (l) {
            case F_FC1: return -1;
            case F_HEAD: case F_S1: case F_A1: return F_FC9;
            default: return l - 1;
        }
    }
    return (l == G_T1 || l == G_K1) ? -1 : l - 1;
}

static bool estimate_program(const Weights& w, int prog, int W, int C, std::vector<std::vector<double>>* out_var, double* hidden_rms,
                             long long* acc_bound, std::string* err) {
    const int L = prog_layers(prog);
    out_var->assign(L, {});
    const double qa = 1.0 / (12.0 * 32767.0 * 32767.0);       // variance of an activation's rounding, in units of [-1,1]^2
    const double lb = 5461.5 / (32767.0 * 32767.0);           // E[b^2] of a uniformly distributed low activation digit, same units
    // Pass 1, layers in parallel: everything a layer's own weights contribute (their rounding, the activations' rounding through them, the dropped L x b
    // product) and its accumulator bound.  Pass 2, in layer order and cheap: the error the INPUT activations already carry, through the weights.
    std::vector<Dense> dense(L);
    std::vector<std::vector<double>> own(L);
    std::vector<long long> bound_l(L, 0);
    std::vector<int> n_hidden_l(L, 0);
    if (!parallel_layers(L, err, [&](int l, std::string* err) {
            const LayerShape s = prog_layer(prog, W, C, l, FMT_I8);
            Dense& d = dense[l];
            if (!dense_layer(w, prog, W, C, l, /*fold_bn=*/true, &d, err)) return false;
            const int f0 = kind_features(s.kind0, s.ks0, FMT_I8);
            const int rk = raw_kind(s), rdims = raw_dims(rk), rbase = (rk != IN_NONE && rk == s.kind1 && rk != s.kind0) ? f0 : 0;
            auto is_raw_col = [&](int c) { return rk != IN_NONE && c >= rbase && c < rbase + rdims; };
            n_hidden_l[l] = s.kind0 == IN_H ? (f0 < d.k ? f0 : d.k) : 0;          // columns [0, n_hidden) are hidden activations
            own[l].assign(d.n, 0.0);
            for (int n = 0; n < d.n; ++n) {
                const double* row = d.W.data() + (size_t)n * d.k;
                double mx = 0.0;
                for (int c = 0; c < d.k; ++c) if (!is_raw_col(c)) mx = std::fmax(mx, std::fabs(row[c]));
                const double sn = mx > 0.0 ? mx / 32512.0 : 1.0;
                double v = 0.0;
                long long sT = 0, sL = 0;
                for (int c = 0; c < d.k; ++c) {
                    if (is_raw_col(c)) continue;                                       // fp32 path: no digits
                    const long long q = std::llround(row[c] / sn);
                    const long long T = (q + 128) >> 8, Lq = q - 256 * T;
                    sT += T < 0 ? -T : T; sL += Lq < 0 ? -Lq : Lq;
                    const double dw = row[c] - sn * (double)q;                         // this weight's rounding error, exactly
                    v += 0.5 * dw * dw                                                 // times an activation of mean square 1/2
                       + row[c] * row[c] * qa                                          // the activation's rounding through the weight
                       + sn * sn * (double)(Lq * Lq) * lb;                             // the dropped L x b product
                }
                // |M| <= 128 sum|T|, |X| <= 128 sum(|T| + |L|)   (digits in [-128, 127])
                const long long bound = 256LL * 128LL * sT + 128LL * (sT + sL);
                if (bound > bound_l[l]) bound_l[l] = bound;
                own[l][n] = v;
            }
            return true;
        }))
        return false;
    for (int l = 0; l < L; ++l) {
        const LayerShape s = prog_layer(prog, W, C, l, FMT_I8);
        const Dense& d = dense[l];
        const int f0 = kind_features(s.kind0, s.ks0, FMT_I8);
        const int rk = raw_kind(s), rdims = raw_dims(rk), rbase = (rk != IN_NONE && rk == s.kind1 && rk != s.kind0) ? f0 : 0;
        auto is_raw_col = [&](int c) { return rk != IN_NONE && c >= rbase && c < rbase + rdims; };
        const int src = hidden_source(prog, l);
        const std::vector<double>* ein = src >= 0 ? &(*out_var)[src] : nullptr;
        const int nh = ein ? (n_hidden_l[l] < (int)ein->size() ? n_hidden_l[l] : (int)ein->size()) : 0;
        std::vector<double>& ev = (*out_var)[l];
        ev.assign(d.n, 0.0);
        double sum = 0.0;
        if (bound_l[l] > *acc_bound) *acc_bound = bound_l[l];
        for (int n = 0; n < d.n; ++n) {
            const double* row = d.W.data() + (size_t)n * d.k;
            double v = own[l][n];
            for (int c = 0; c < nh; ++c)
                if (!is_raw_col(c)) v += row[c] * row[c] * (*ein)[c];
            // sine layer: d sin(2 pi z) = 2 pi cos(.) dz, mean square of the cosine 1/2
            ev[n] = s.out_kind == OUT_SIN ? (2.0 * M_PI) * (2.0 * M_PI) * 0.5 * v : v;
            sum += ev[n];
        }
        if (s.out_kind == OUT_SIN && d.n > 0) *hidden_rms = std::fmax(*hidden_rms, std::sqrt(sum / d.n));
    }
    return true;
}

bool estimate_i8(const Weights& w, int W, int C, I8Estimate* out, std::string* err) {
    if (W < 64 || W % 64 != 0) { *err = "layer width must be a multiple of 64 (got " + std::to_string(W) + ")"; return false; }
    if (C < 1 || C > kMaxClasses) { *err = "n_classes must be in [1," + std::to_string(kMaxClasses) + "]"; return false; }
    *out = I8Estimate();
    std::vector<std::vector<double>> ev;
    if (!estimate_program(w, PROG_FIELD, W, C, &ev, &out->hidden_rms, &out->acc_bound, err)) return false;
    auto rms = [](const std::vector<double>& v, int a, int b) {
        double s = 0.0;
        for (int i = a; i < b && i < (int)v.size(); ++i) s += v[i];
        return b > a ? std::sqrt(s / (b - a)) : 0.0;
    };
    out->head_rms[0] = rms(ev[F_HEAD], 3, 4);           // fc10Sigma
    out->head_rms[1] = rms(ev[F_HEAD], 0, 3);           // fc10Col
    out->head_rms[2] = rms(ev[F_S4], 0, 1);
    out->head_rms[3] = rms(ev[F_AC], 0, 3 * C);
    for (double v : out->head_rms) out->worst = std::fmax(out->worst, v);
    // What an error of a raw head output does to the rendered colour, relative (Eval_Tools_2.py:187-215): the density enters through
    // the transmittance weights PS (and alone sets the depth), colour and seasonal adjustment through a sigmoid (slope <= 1/4) averaged
    // over the samples of a ray, the solar visibility through the shading term.  (The per-ray networks - class softmax, sky colour -
    // never run in int8 digits: their error would not average over a ray's samples.)  Weights: round 3 fitted 0.35 / 0.15 / 0.20 / 0.10 on
    // synthetic weight families (tools/calibrate_i8_bound.py: init law, outliers, heavy tails, gains: prediction ~2x the worst observed error).
    // Round 4 measured REALLY TRAINED weights - the reference's own loop, 400-600 steps (tests/golden/trained_W*.npz, tools/trained_modes.py on
    // the GPU against the reference's eval): per unit of predicted head error they render 3-4x worse than the synthetic families (observed
    // 3.5-3.7e-5 against a prediction of 1.8-2.4e-5), so the weights are scaled x2.3: the prediction now covers every measured set
    // (trained W = 64 / 256 / 512: 5.5 / 4.0 / 3.6e-5 predicted against 3.7 / 3.5 / 3.2-3.4e-5 observed; init law 5.1e-5 / 1.4e-5; x4-outlier and
    // Laplace families 1.7-1.9e-4 / 3.1-5.8e-5 - those now go to bf16x3: the guard errs on the safe side for weights unlike anything training
    // produced here).
    // Round 5 put SURFACES into the weights - the trained fixtures with the density head scaled by g = 1 ... 256 (mean max-PS per ray 0.03 ... 0.87), rendered
    // by the reference for every g (tests/golden/sharp_sweep_W*.npz) - and measured forced int8 digits against them on 128 rays per set (tools/sharp_modes.py,
    // profiles/r5/sharp_modes_before_refit.txt): the round-4 weights UNDER-predicted the near-fog end, where the decision falls (observed / predicted 1.14 at
    // W = 64 g = 1, 1.24 at W = 256 g = 4 - observed 1.06e-4, outside the bar, predicted 8.6e-5 -> int8 digits -, 1.39 at W = 512 g = 2), and over-predict
    // hard surfaces 4-6x (the prediction is linear in g, the rendering saturates; those sets leave the bar in int8 digits anyway: observed 2-11e-4).  Scaled
    // x1.5 so that the prediction covers every set of the ladder with a margin (1.20 / 0.525 / 0.69 / 0.345; x1.4 would cover W = 64 / 256, on which it was
    // chosen, by >= 1.13x but the held-out W = 512 g = 2 rung by 1.00x): what stays on the int8 pipe is the init law and near-fog (max-PS below ~0.1);
    // tests/test_cabi_host.py::test_error_model_covers_the_gain_ladder holds every rung of all three widths.
    out->rgb_pred = 1.20 * out->head_rms[0] + 0.525 * out->head_rms[1] + 0.69 * out->head_rms[3] + 0.345 * out->head_rms[2];
    return true;
}

}  // namespace snerf
