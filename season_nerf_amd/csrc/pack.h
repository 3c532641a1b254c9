// Host-side containers for the weight packer (see pack.cpp, program.h).
#pragma once
#include <map>
#include <string>
#include <vector>

#ifndef __host__
#define __host__
#define __device__
#endif
#include "program.h"

namespace snerf {

struct Tensor {
    std::vector<float> data;
};

// state_dict of the reference T_NeRF: key -> flat fp32 data, row-major as torch stores it
struct Weights {
    std::map<std::string, Tensor> t;
    const Tensor* find(const std::string& k) const;
};

struct Packed {
    std::vector<uint8_t> stream;   // prog_chunks * 16 KiB of bf16 hi/lo fragment pairs
    std::vector<float> bias;       // prog_bias_floats, accumulator order
};

bool pack_program(const Weights& w, int prog, int W, int C, bool fold_bn, Packed* out, std::string* err);
// K-split order of a packed bf16 FIELD program (program.h ks_*; kernels_ks.hip): a permutation of `canon.stream`'s pairs, the bias table unchanged
bool permute_program_ks(const Packed& canon, int W, int C, Packed* out, std::string* err, int prog = PROG_FIELD);
// int8-digit format (FMT_I8): stream of T/L digit fragment pairs, `bias` = per-row [scale | bias] tables
bool pack_program_i8(const Weights& w, int prog, int W, int C, bool fold_bn, Packed* out, std::string* err);

// Pack-time error model of the int8-digit format (what SNERF_PREC_AUTO decides on; include/season_nerf_hip.h snerf_i8_estimate).
// For every output row of every layer: the variance of the pre-activation error the format adds - weight rounding
// (s_n / 2 per weight, s_n = row max / 32512), activation rounding (2^-16 of [-1,1]) and the dropped low x low digit product,
// all exact functions of the packed integers - plus the error of the layer's inputs carried through the row's weights;
// a sine layer turns a pre-activation error (in revolutions) into an activation error by 2 pi |cos|.  The integer bound is
// exact: the largest |(M << 8) + X| any activation digits can produce for the packed row (int32 accumulators, no wrap below 2^31).
struct I8Estimate {
    double head_rms[4] = {0, 0, 0, 0};   // predicted RMS error of the raw head outputs: density, colour, solar visibility, seasonal adjust
    double hidden_rms = 0;               // worst RMS error of a hidden layer's activations
    double worst = 0;                    // max over head_rms
    double rgb_pred = 0;                 // predicted relative error of the rendered colour (and depth): the heads weighted by what they move
    long long acc_bound = 0;             // exact bound of |(M << 8) + X| over all rows
};
bool estimate_i8(const Weights& w, int W, int C, I8Estimate* out, std::string* err);

}  // namespace snerf
