// The W = 512 (reference default width, main_lite.py:80) field instances of the int8-digit kernel: kernels_i8.hip compiled
// again with only those instantiated, under -mllvm -pragma-unroll-threshold (build.py) - see the note at launch_mlp_i8_w512.
#define SNERF_I8_W512_TU 1
#include "kernels_i8.hip"
