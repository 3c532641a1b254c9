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
// int8-digit format (FMT_I8): stream of T/L digit fragment pairs, `bias` = per-row [scale | bias] tables
bool pack_program_i8(const Weights& w, int prog, int W, int C, bool fold_bn, Packed* out, std::string* err);

}  // namespace snerf
