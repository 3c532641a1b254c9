// Kernel argument blocks and launch entry points (internal; the public surface is include/season_nerf_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace snerf {

struct snerf_field_out_dev {
    float* rho;
    float* solar_vis;
    float* col_raw;
    float* adjust;
    float* col;
    float* adjust_col;
    float* points;
    float* vis;               // VARIANT 3 only: [n_rays] exp(-sum_{j < S-1} rho_j delta_j), one value per ray
};

constexpr int kVoteBytes = 64;    // LDS behind every fused kernel's image: one float per wave for the ray-visibility early-out vote (mlp_device.h raysum_saturated)

struct MlpArgs {
    const uint8_t* stream;     // packed fragment stream (device)
    uint32_t stream_bytes;     // bytes consumed per tile = length of the cyclic DMA stream
    const float* bias;         // bias table (device)
    int bias_floats;
    int64_t n;                 // points (field program) or groups (group program)
    int n_classes;
    // field program inputs
    const float* points;       // [n,3] or NULL -> generate from rays
    const float* top;          // [R,3]
    const float* bot;          // [R,3]
    const float* tvals;        // [S]
    int n_samples;
    int64_t group_size;        // points per sun/classes row
    const float* sun;          // [G,3]
    const float* classes;      // [G,C] or NULL
    snerf_field_out_dev out;
    // group program inputs / outputs
    const float* time;         // [G,4]
    float* g_classes;          // [G,C]
    float* g_sky_raw;          // [G,3]
    float* g_sky;              // [G,3]
    uint32_t debug;            // only read by -DSNERF_ABLATE builds
    // VARIANT 3 (ray visibility: the density-only program with the sum over a ray's samples kept in registers): n = rays, every wave
    // owns one ray of a group of `waves per workgroup` rays and walks its samples 32 at a time
    int ray_flags;             // bit 1: a sample outside [-1,1]^3 contributes nothing (mg_Img_Eval.py:42,65-66); bit 2: no early-out (A/B); bit 3: passes from the sun side inwards (the order before round 6's reversal, A/B)
};

struct CompOutDev {
    float *rgb, *albedo, *pv, *pe, *ps, *delta, *shadow, *acc, *surf_loc, *surf_dist;
};

struct CompArgs {
    int64_t n_rays;
    int n_samples;
    const float *top, *bot, *tvals;
    const float *rho, *col, *solar_vis, *sky;
    int flags;
    const float* rho_prior;
    float trust;
    const float* trust_dev;       // optional: the trust factor read from device memory at run time (captured steps); overrides `trust`
    CompOutDev out;
};

struct RayGenArgs {
    double P[12];          // row-major 3x4, normalised so that P[11] = 1 is NOT assumed
    int rows, cols, ds;
    float* rows_out;       // [rows*cols, 11]: img_pt 2 | top 3 | bot 3 | view 3
    uint8_t* valid;        // [rows*cols] or NULL
};
hipError_t launch_rays_from_camera(const RayGenArgs& a, hipStream_t st);

// novel-view ray grids (float64 arithmetic as the reference's numpy, rounded to fp32 at the end)
struct RayGridArgs {
    int mode;              // 0: by direction (mg_Img_Eval.py:96-115), 1: Quick_Run (Quick_Run.py:77-109), 2: through a 3x4 camera (mg_Img_Eval.py:74-94)
    int rows, cols;        // the output grid H x W
    int64_t lo, hi;        // rays lo .. hi-1 of the row-major grid
    double q[3];           // modes 0, 1: view vector / its z component
    int has_region;        // mode 1
    double region[4];
    double P[12];          // mode 2
    int img_rows, img_cols;
    float *top, *bot;      // [hi - lo, 3]
    uint8_t* valid;        // [hi - lo] or NULL
    int32_t* pix;          // mode 2, optional: [hi - lo, 2] source pixel (row, col)
};
hipError_t launch_ray_grid(const RayGridArgs& a, hipStream_t st);

struct SweepArgs {
    int64_t n_rays;
    int n_samples, n_classes, n_times, flags;
    const float *top, *bot, *tvals;
    const float* deltas;       // optional [R,S]: explicit segment lengths (then top / bot / tvals are not read)
    const float *rho, *col_raw, *adjust, *solar_vis;
    int adjust_vec4;           // adjust is 16-byte aligned and n_classes == 4: a sample's 12 values are three 16-byte loads
    const float* sky;          // [3]
    const float* class_vecs;   // [T,C]
    float *season, *shaded;    // [T,R,3]
    float* classic;            // optional [T,R,3]: sum_s PS * sigmoid(.) * (SV + (1-SV)*sky)  (mg_Img_Eval.py:165-170)
    float *base, *shadow_adjust, *raw_shadow;   // [R,3], [R,3], [R]
};
hipError_t launch_sweep(const SweepArgs& a, hipStream_t st);
hipError_t launch_mlp(int prog, int W, int variant, bool fast, const MlpArgs& a, int n_cu, hipStream_t st);
hipError_t launch_mlp_i8(int prog, int W, int variant, const MlpArgs& a, int n_cu, hipStream_t st);      // kernels_i8.hip
int field_variant_chunks_i8(int W, int C, int variant);
hipError_t launch_mlp_i8x2(int W, int variant, const MlpArgs& a, int n_cu, hipStream_t st);                // kernels_i8x2.hip (W <= 256)
hipError_t launch_mlp_ks(int W, int variant, const MlpArgs& a, int n_cu, hipStream_t st);                // kernels_ks.hip (W = 512, bf16x3, K split over wave pairs)
hipError_t launch_mlp_ks_group(int W, const MlpArgs& a, int n_cu, hipStream_t st);                         // ... its per-ray (time / sun) networks
int field_variant_chunks_ks(int W, int C, int variant);
int group_chunks_ks(int W, int C);
int mlp_ks_lds_bytes(int bias_floats);
int mlp_ks_tile_points();
hipError_t launch_composite(const CompArgs& a, hipStream_t st);
// dsm.hip
hipError_t launch_prior_density(int64_t n, const float* pts, const float* delta, const double* hm, int hx, int hy, const float* outside,
                                float neg_log_term, float* rho, hipStream_t st);
hipError_t launch_surface_distance(int64_t n_rays, int S, const float* top, const float* bot, const float* tvals, const double* dsm, int dx,
                                   int dy, const double* levels, double* dist, hipStream_t st);
hipError_t launch_image_error(int64_t n_pix, const float* img, const float* gt, double* sums, hipStream_t st);
hipError_t launch_transmittance(int64_t n_rays, int S, const float* rho, const float* delta, float* pv, hipStream_t st);
int mlp_lds_bytes(int bias_floats);
int mlp_tile_points();
int field_variant_chunks(int W, int C, int variant);
const char* mlp_kernel_name();

}  // namespace snerf
