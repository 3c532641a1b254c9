// DSM-prior and validation kernels: the gathers and per-ray scans the reference does on the CPU
// between device passes (`.cpu()` round trips at Eval_Tools_2.py:220-221,321-326 and mg_run_NeRF.py:106-120,186-208).
//   prior_density_kernel      T_NeRF.Supervised_Sample (T_NeRF_net_v2.py:175-181)
//   surface_distance_kernel   Net_tool.get_Dist (mg_run_NeRF.py:106-120) straight from the 2-D DSM: the reference's dense
//                             volume is Dense[x,y,k] = (DSM[x,y] >= linspace(-1,1,n)[k]) + 0*DSM (mg_run_NeRF.py:55-61)
//   image_error_kernel        eval_img's colour error sums (mg_run_NeRF.py:204-208)
// All HBM-bound gathers / reductions over at most R*S elements; one thread per point (or per ray for the sequential scan).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace snerf {

__global__ __launch_bounds__(256) void prior_density_kernel(int64_t n, const float* pts, const float* delta, const double* hm, int hx,
                                                            int hy, const float* outside, float neg_log_term, float* rho) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = pts[i * 3], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
    if (outside && (x > 1.f || x < -1.f || y > 1.f || y < -1.f || z > 1.f || z < -1.f)) {      // Eval_Tools_2.py:321-325
        rho[i] = outside[i];
        return;
    }
    // xy = ((pts[:, 0:2] + 1) / 2 * (shape - 1)).long()     fp32 arithmetic, truncation
    int64_t ix = (int64_t)(__fmul_rn(__fdiv_rn(__fadd_rn(x, 1.f), 2.f), (float)(hx - 1)));
    int64_t iy = (int64_t)(__fmul_rn(__fdiv_rn(__fadd_rn(y, 1.f), 2.f), (float)(hy - 1)));
    ix = ix < 0 ? 0 : (ix > hx - 1 ? hx - 1 : ix);        // memory safety only: the reference raises outside the cube
    iy = iy < 0 ? 0 : (iy > hy - 1 ? hy - 1 : iy);
    const bool exists = hm[ix * hy + iy] >= (double)z;
    // -log(1 - min(p, .99)) / delta with p in {0, 1}: the two values of the numerator are constants
    rho[i] = __fdiv_rn(exists ? neg_log_term : -0.f, delta[i]);
}

__global__ __launch_bounds__(64) void surface_distance_kernel(int64_t n_rays, int S, const float* top, const float* bot, const float* tvals,
                                                              const double* dsm, int dx, int dy, const double* levels, double* dist) {
    const int64_t r = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (r >= n_rays) return;
    const float tx = top[r * 3], ty = top[r * 3 + 1], tz = top[r * 3 + 2];
    const float bx = bot[r * 3], by = bot[r * 3 + 1], bz = bot[r * 3 + 2];
    const float ex = tx - bx, ey = ty - by, ez = tz - bz;
    const float delta = __fdiv_rn(__fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(ez, ez))), (float)S);
    double trans = 1.0, sum_p = 0.0, sum_pd = 0.0, run = 0.0;
    for (int s = 0; s < S; ++s) {
        const float t = tvals[s], omt = __fsub_rn(1.f, t);
        const float px = __fadd_rn(__fmul_rn(tx, omt), __fmul_rn(bx, t));
        const float py = __fadd_rn(__fmul_rn(ty, omt), __fmul_rn(by, t));
        const float pz = __fadd_rn(__fmul_rn(tz, omt), __fmul_rn(bz, t));
        int64_t ix = (int64_t)(__fmul_rn(__fdiv_rn(__fadd_rn(px, 1.f), 2.f), (float)(dx - 1)));
        int64_t iy = (int64_t)(__fmul_rn(__fdiv_rn(__fadd_rn(py, 1.f), 2.f), (float)(dy - 1)));
        int64_t iz = (int64_t)(__fmul_rn(__fdiv_rn(__fadd_rn(pz, 1.f), 2.f), (float)(S - 1)));
        ix = ix < 0 ? 0 : (ix > dx - 1 ? dx - 1 : ix);
        iy = iy < 0 ? 0 : (iy > dy - 1 ? dy - 1 : iy);
        iz = iz < 0 ? 0 : (iz > S - 1 ? S - 1 : iz);
        const double h = dsm[ix * dy + iy];
        const double pe = (h != h) ? h : (h >= levels[iz] ? 1.0 : 0.0);       // NaN cells of the DSM stay NaN
        run += (double)delta;                               // torch.cumsum on the CPU: fp32 values accumulated in double,
        const double cs = (double)(float)run;               // rounded to fp32 per element
        const double p = pe * trans;
        sum_p += p;
        sum_pd += p * cs;
        trans *= (1.0 - pe);
    }
    dist[r] = sum_pd / sum_p;       // 0/0 = NaN for rays that never meet the surface, as in the reference
}

// sums[0] += sum log(1/2 (gt - img)^2 + 1), sums[1] += sum (gt - img)^2, sums[2] += 3 * #pixels with any(gt != 0)
__global__ __launch_bounds__(256) void image_error_kernel(int64_t n_pix, const float* img, const float* gt, double* sums) {
    double cauchy = 0.0, sq = 0.0, cnt = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_pix; i += (int64_t)gridDim.x * 256) {
        bool any = false;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double g = (double)gt[i * 3 + c], d = g - (double)img[i * 3 + c];
            cauchy += log(0.5 * d * d + 1.0);
            sq += d * d;
            any |= g != 0.0;
        }
        cnt += any ? 3.0 : 0.0;
    }
    for (int o = 32; o > 0; o >>= 1) {
        cauchy += __shfl_xor(cauchy, o, 64);
        sq += __shfl_xor(sq, o, 64);
        cnt += __shfl_xor(cnt, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(sums, cauchy);
        atomicAdd(sums + 1, sq);
        atomicAdd(sums + 2, cnt);
    }
}

// Eval_Tools_2.get_PV (:13-16) as a stand-alone op: PV[r,s] = exp(-sum_{j<s} rho[r,j]*delta[r,j]) for arbitrary per-sample
// deltas (the compositing kernel computes the same scan fused, with delta derived from the ray).  One wavefront per ray.
__global__ __launch_bounds__(256) void transmittance_kernel(int64_t n_rays, int S, const float* rho, const float* delta, float* pv) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rays) return;
    float carry = 0.f;
    for (int base = 0; base < S; base += 64) {
        const int s = base + lane;
        const bool in = s < S;
        const float y = in ? rho[r * S + s] * delta[r * S + s] : 0.f;
        float incl = y;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        if (in) pv[r * S + s] = expf(-(carry + incl - y));
        carry += __shfl(incl, 63, 64);
    }
}
hipError_t launch_transmittance(int64_t n_rays, int S, const float* rho, const float* delta, float* pv, hipStream_t st) {
    if (n_rays <= 0 || S <= 0) return hipSuccess;
    hipLaunchKernelGGL(transmittance_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, st, n_rays, S, rho, delta, pv);
    return hipGetLastError();
}

hipError_t launch_prior_density(int64_t n, const float* pts, const float* delta, const double* hm, int hx, int hy, const float* outside,
                                float neg_log_term, float* rho, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(prior_density_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, pts, delta, hm, hx, hy, outside,
                       neg_log_term, rho);
    return hipGetLastError();
}
hipError_t launch_surface_distance(int64_t n_rays, int S, const float* top, const float* bot, const float* tvals, const double* dsm, int dx,
                                   int dy, const double* levels, double* dist, hipStream_t st) {
    if (n_rays <= 0) return hipSuccess;
    hipLaunchKernelGGL(surface_distance_kernel, dim3((unsigned)((n_rays + 63) / 64)), dim3(64), 0, st, n_rays, S, top, bot, tvals, dsm, dx,
                       dy, levels, dist);
    return hipGetLastError();
}
hipError_t launch_image_error(int64_t n_pix, const float* img, const float* gt, double* sums, hipStream_t st) {
    if (n_pix <= 0) return hipSuccess;
    int64_t b = (n_pix + 255) / 256;
    if (b > 1024) b = 1024;
    hipLaunchKernelGGL(image_error_kernel, dim3((unsigned)b), dim3(256), 0, st, n_pix, img, gt, sums);
    return hipGetLastError();
}

}  // namespace snerf
