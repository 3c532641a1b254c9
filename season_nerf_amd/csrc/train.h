// Layer-wise training engine (internal header): kernels' argument blocks and launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace snerf {

enum GemmFlags : int { GEMM_ACCUM = 1, GEMM_ATOMIC = 2 };

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    int64_t M, N, K;
    int64_t sAm, sAk, sBk, sBn, ldc;
    float alpha;
    const float* bias;    // [N] added before alpha (only by split 0)
    float* colsum;        // [N] += column sums of the written values (BatchNorm mean)
    int flags;
    int splitk;           // >1: K split over blockIdx.z, results combined with fp32 atomics (C must be zeroed)
};
hipError_t launch_gemm(const GemmArgs& g, hipStream_t st);

// bf16x3 row-owner GEMM (forward and dgrad of the training engine): C (+)= alpha*(A Bt^T + bias); Bt pre-split into MFMA
// fragment order by launch_split_weights (n_tiles x ksteps fragments of 2 KiB: bf16 hi, bf16 lo)
struct GemmX {
    const float* A;            // [M, lda] fp32, k contiguous
    // weights: either already split (frag, 32x32x16 fragment order: [n_tiles][ksteps][2][512] bf16) or raw fp32 (W != NULL): the
    // launcher then splits them into `frag` (scratch of n_tiles * ksteps * 2 KiB, any of the two fragment orders) for the kernel it picks
    const uint16_t* frag;
    const float* W;            // optional raw weights [w_rows, w_cols]; Bt[n][k] = W[n][k] (w_transpose = 0) or W[k][n] (1)
    int w_rows, w_cols, w_transpose;
    float* C;
    int64_t M, N, K, lda, ldc;
    int n_tiles, ksteps;       // ceil(N/32), ceil(K/16)
    float alpha;
    const float* bias;
    double* stats;             // optional [2][N]: += sum_m (v - alpha*bias), += sum_m (v - alpha*bias)^2   (train-mode BatchNorm)
    int accumulate;
    int reverse;               // row tiles from the last to the first (set by the launcher, see launch_gemm_rows16)
    const float* act_tab;      // optional activation-on-load table [a | b] x act_cols for the leading columns of A: sin(2 pi (a z + b))
    int act_cols;              // multiple of 8, <= K
    int tab_lds;               // set by the launcher: the table fits in LDS behind the weights
    // activation-backward epilogue (dgrad): C = dL/dH of the SineLayer below -> written as dL/d(arg) = C * cos(2 pi (a z + b));
    // ez = that layer's pre-activation [M, eld], etab = its [a | b] table (N columns), emu/eistd = its BatchNorm statistics
    // (NULL without BatchNorm); stats (required) receives sum v and sum v*xhat per column
    const float* ez;
    int64_t eld;
    const float *etab, *emu, *eistd;
    int a_padded;              // caller's promise: columns K .. 16*ksteps-1 of A exist (lda covers them) and hold zeros
};
hipError_t launch_split_weights(const float* W, int rows, int cols, bool transpose, uint16_t* frag, int n_tiles, int ksteps, hipStream_t st);
int gemm_rows_group_tiles(int ksteps);      // 0 = K too large for the LDS-resident weight layout (caller falls back to fp32 MFMA)
hipError_t launch_gemm_bf16x3(const GemmX& g, hipStream_t st);
hipError_t launch_gemm_rows16(const GemmX& gx, int aol_mode, int act_mode, dim3 grid, size_t lds, hipStream_t st, int nt16 = 8);      // gemm16.hip
// gemm_areg.hip: K = 256 / 512 with the activations resident in AGPRs and the weights streamed through the LDS ring (the reference's default width)
bool gemm_areg_ok(const GemmX& g);
hipError_t launch_areg_split_weights(const float* W, int rows, int cols, bool transpose, uint16_t* frag, int n_tiles, int ksteps, hipStream_t st);      // k-major stream
hipError_t launch_gemm_areg(const GemmX& g, hipStream_t st);
// bf16x3 weight gradient: dW[n_out, n_in] (ld ldw) += alpha * dZ[M, n_out]^T In[M, n_in]   (fp32 atomics over M-chunks)
// optional BatchNorm backward folded into the weight-gradient kernel (dZ holds dL/dY on entry, dL/dZ on exit; needs n_in <= 256)
struct WgradBN {
    const float* z;            // pre-activation of this layer [M, ldz]
    int64_t ldz;
    const float *gamma, *mu, *istd, *sdy, *sdyx;
    float inv_m;               // 1 / rows of the (global) batch the means were taken over
    float bias_alpha;
    float* dbias;              // += bias_alpha * sum_m dZ
};
int stream_direction(int64_t rows);      // 0 / 1, alternating per large streaming launch (gemm.hip)
// per-thread launch context of a training pass: the trainer's wgrad partial-sum scratch and its launch parity (nullptr / 0: none)
void gemm_launch_context(float* wgrad_partial, size_t wgrad_floats, unsigned* parity);
size_t gemm_wgrad_partial_floats();      // floats the two-stage weight-gradient reduction needs at most (grid blocks x 8 x 8 x 1024)
hipError_t launch_wgrad_bf16x3(float* dZ, int64_t ldz, const float* In, int64_t ldi, int64_t M, int n_out, int n_in, float alpha,
                               float* dW, int64_t ldw, hipStream_t st, const float* in_tab = nullptr, int in_cols = 0,
                               const WgradBN* bn = nullptr, int in_tab_stride = 0);      // in_tab_stride: see WgradX (0 = in_cols)

// ---- elementwise / reduction kernels of the training path (train_kernels.hip)
struct PeArgs {            // positions (from rays or explicit) -> PE(pos) [N,64] (63 features + zero pad) and points [N,3]
    int64_t n;
    int n_samples;
    const float *points, *top, *bot, *tvals;
    float* pe;             // [N,64]
    float* pts;            // [N,3] or NULL
    float* pe2;            // optional second copy of PE with row stride ld2 (a concat buffer's columns), or NULL
    int64_t ld2;
};
hipError_t launch_pe_points(const PeArgs& a, hipStream_t st);
// per-row encodings of small vectors: sun [G,3] -> [G,28] (27 + pad), time[:,0:2] [G,4] -> [G,12] (10 + pad)
hipError_t launch_pe_small(const float* in, int in_stride, int n_dims, int n_freq, int64_t rows, float* out, int out_stride, hipStream_t st);

// column reductions over rows of X [M, ld] for C columns.  mode 0: sum (x - mu_c)^2 -> out0
// mode 1: dY = dH*cos(Y) with Y = g_c*(z - mu_c)*istd_c + b_c (written in place over dH), out0 += sum dY, out1 += sum dY*xhat
// mode 2: dZ = dH*cos(Z) in place, out0 += sum dZ            (layers without BatchNorm)
struct ColArgs {
    int mode;
    int64_t M;
    int C;
    int64_t ld;            // row stride of Z
    int64_t ldd;           // row stride of D (0 = same as ld)
    const float* Z;        // pre-activation [M, ld]
    float* D;              // gradient buffer [M, ldd] (in place)
    const float *mu, *istd, *gamma, *beta;
    float *out0, *out1;
    float alpha0;          // scale of the out0 accumulation (mode 2: 30 = d bias of the Linear)
    int64_t M_global;      // mode 3: rows of the global batch the BatchNorm means were taken over (>= M)
};
hipError_t launch_colreduce(const ColArgs& a, hipStream_t st);
struct ThinDgradArgs {     // input gradient of a head with K <= 4 outputs as a stream over C (train_kernels.hip: thin_dgrad_kernel)
    const float* D;        // [M, ldd]: dL/d(head output), K leading columns
    const float* W;        // [K, ldw]: the head's weights (row k = output k)
    const float* W3;       // optional: row 3 lives here instead (two heads in one launch: colour rows 0..2, density row 3)
    float* C;              // [M, ldc], N columns
    int64_t M, ldd, ldw, ldc;
    int K, N, accumulate;
    float alpha;
    // optional activation backward of the layer below (as the row GEMMs' epilogue): pre-activations, [a | b] table (N each), BatchNorm
    // mean / istd (or NULL), the double column sums [2][N]
    const float* ez;
    int64_t eld;
    const float *etab, *emu, *eistd;
    double* stats;
};
bool thin_dgrad_ok(const ThinDgradArgs& a);
hipError_t launch_thin_dgrad(const ThinDgradArgs& a, hipStream_t st);
struct ThinFwdArgs {       // forward of heads with K <= 4 outputs in all as one stream over their common input (train_kernels.hip: thin_fwd_kernel)
    const float* In;       // [M, ldi], N columns (the first tab_cols of them stored as pre-activations: activation on load)
    const float* W;        // [K, ldw] rows 0..K-1 - row 3 at W3 if set (colour head rows 0..2, density head row 3)
    const float* W3;
    const float* bias;     // [K] - element 3 at bias3 if set
    const float* bias3;
    float* Out;            // [M, ldo]: Out[m, k] = alpha * (sum_n In[m, n] W[k, n] + bias[k])
    int64_t M, ldi, ldw, ldo;
    int K, N;
    float alpha;
    const float* tab;
    int tab_cols, tab_stride;
};
bool thin_fwd_ok(const ThinFwdArgs& a);
hipError_t launch_thin_fwd(const ThinFwdArgs& a, hipStream_t st);
struct ThinWgradArgs {     // weight gradient of a head with K <= 4 outputs as a stream over its input (train_kernels.hip: thin_wgrad_kernel)
    const float* D;        // [M, ldd]: dL/d(head output), K leading columns
    const float* In;       // [M, ldi], N columns: the head's input - or, for the first tab_cols columns, its stored pre-activation (activation on load)
    float* dW;             // [K, ldw] += alpha * D^T In  (two stages: per-block sums in `partial`, then one reduction kernel - deterministic)
    float* partial;        // [blocks, K * N] scratch (set by the launcher)
    float* dW3;            // optional: row 3 of the result goes here instead (two heads in one launch)
    int64_t M, ldd, ldi, ldw;
    int K, N;
    float alpha;
    const float* tab;      // optional [a | b] table: In[:, c] = sin(2 pi (a[c] z + b[c])) for c < tab_cols, b at distance tab_stride
    int tab_cols, tab_stride;
};
float* gemm_partial_scratch(hipStream_t st, size_t floats);      // gemm.hip: the calling trainer's pre-carved partial-sum scratch (or the stream's own)
bool thin_wgrad_ok(const ThinWgradArgs& a);
hipError_t launch_thin_wgrad(const ThinWgradArgs& a, hipStream_t st);

// BatchNorm finalize: mean = sum/M, var = m2/M, istd; EMA of running stats (momentum 0.01, unbiased var)
hipError_t launch_bn_finalize(const float* colsum, const float* m2, int64_t M, int C, float* mean, float* istd,
                              float* running_mean, float* running_var, int stage, hipStream_t st);
// H = sin(gamma*(Z-mu)*istd + beta) (bn) or sin(Z)
// consumes AND clears the sums; tab (optional): the layer's activation-on-load table [a | b] from gamma / beta and the new statistics
hipError_t launch_bn_finalize_shifted(double* stats, const float* bias, float alpha, int64_t M, int C, float* mean, float* istd,
                                      float* running_mean, float* running_var, const float* gamma, const float* beta, float* tab,
                                      hipStream_t st);
hipError_t launch_sin_fwd(const float* Z, float* H, int64_t M, int C, int64_t ldz, int64_t ldh, const float* mu, const float* istd,
                          const float* gamma, const float* beta, hipStream_t st);
// BN backward second pass: dZ = gamma*istd*(dY - sdy/M - xhat*sdyx/M) in place; colsum(dZ) -> out (bias grad)
hipError_t launch_bn_bwd2(const float* Z, float* D, int64_t M, int C, int64_t ld, int64_t ldd, const float* mu, const float* istd,
                          const float* gamma, const float* beta, const float* sdy, const float* sdyx, float* dbias_sum, float alpha,
                          int64_t M_global, hipStream_t st, bool d_is_dy = false);
hipError_t launch_act_sums_finalize(double* stats, int C, float scale0, float* out0, float* out1, float* acc0, float* acc1, hipStream_t st);      // consumes and clears the sums
// activation-on-load table of a SineLayer: dst = [a | b] (n each), a = gamma*istd/(2 pi), b = (beta - gamma*mu*istd)/(2 pi);
// all-NULL statistics = a layer without BatchNorm (a = 1/(2 pi), b = 0)
hipError_t launch_act_table(const float* mu, const float* istd, const float* gamma, const float* beta, int n, float* dst, hipStream_t st);

// point outputs: rho = softplus(head[:,3]), col = sigmoid(head[:,0:3] + sum_c cls[g,c]*adj[:,c,:]), sv = sigmoid(sv_raw)
struct PointOutArgs {
    int64_t n;
    int n_samples, C;
    const float *head, *adj, *sv_raw, *cls;     // [N,4], [N,3C], [N], [R,C]
    float *rho, *col, *sv;                      // [N], [N,3], [N]
    float* adjust_col;                          // [N,3] optional: sum_c cls*adj
    // backward (all optional): d_rho [N], d_col [N,3] -> d_head [N,4], d_adj [N,3C], d_cls [R,C] (atomic), d_sv -> d_sv_raw
    const float *d_rho, *d_col, *d_sv;
    float *d_head, *d_adj, *d_cls, *d_sv_raw;
};
hipError_t launch_point_out(const PointOutArgs& a, bool backward, hipStream_t st);

// compositing backward (one wave per ray): from dL/dRendered_Col, dL/dAlbedo, dL/dPE (optional) -> d_rho, d_col, d_sky
struct CompBwdArgs {
    int64_t n_rays;
    int n_samples;
    const float *top, *bot;
    const float *rho, *col, *sv, *sky;          // forward inputs ([N], [N,3], [N], [R,3])
    const float *g_rgb, *g_albedo, *g_pe;       // [R,3] or NULL, [R,3] or NULL, [N] or NULL
    // DSM-prior phase (Eval_Tools_2.py:218-248): merged density rho*trust + rho_prior*(1-trust); gradients of
    // Rendered_Col_Merged / merged Albedo_Color.  The solar factor always comes from the un-merged PS.
    const float* rho_prior;                     // [N] or NULL
    float trust;
    const float* trust_dev;                     // optional device-resident trust (captured steps); overrides `trust`
    const float *g_rgb_m, *g_albedo_m;          // [R,3] or NULL
    float *d_rho, *d_col, *d_sky;               // [N], [N,3], [R,3] (d_sky is overwritten)
    // classic solar model (Solar_Type_2, Eval_Tools_2.py:211-212): Rendered_Col = sum PS*Col*(SV + (1-SV)*Sky) per sample;
    // the solar visibility then carries gradient: d_sv [N] (required when classic != 0)
    int classic;
    float* d_sv;
};
hipError_t launch_composite_bwd(const CompBwdArgs& a, hipStream_t st);

// softmax rows [R,C] forward / backward; sigmoid forward/backward on small tensors
hipError_t launch_softmax(const float* logits, float* p, int64_t rows, int C, hipStream_t st);
hipError_t launch_softmax_bwd(const float* p, const float* dp, float* dlogits, int64_t rows, int C, hipStream_t st);
hipError_t launch_sigmoid(const float* x, float* y, int64_t n, hipStream_t st);
hipError_t launch_sigmoid_bwd(const float* y, const float* dy, float* dx, int64_t n, hipStream_t st);
// column sum of a small-N matrix [M,C] (bias gradients of head layers): out[c] += alpha * sum_m X[m,c]
hipError_t launch_colsum(const float* X, int64_t M, int C, int64_t ld, float alpha, float* out, hipStream_t st);
// concat-free helpers
hipError_t launch_copy_cols(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int64_t M, int C, bool accumulate, hipStream_t st);
hipError_t launch_fill_zero_bytes(void* p, size_t bytes, hipStream_t st);
hipError_t launch_copy_bytes(void* d, const void* s, size_t bytes, hipStream_t st);
hipError_t launch_copy_f32(float* d, const float* s, int64_t n, hipStream_t st);            // the engine's device-to-device copy and zero-fill: kernels, so that a
hipError_t launch_fill_zero(float* p, int64_t n, hipStream_t st);                          // captured step holds no hipGraph memory-operation node (train.cpp)
hipError_t launch_round_bf16(float* p, int64_t ld, int64_t M, int C, hipStream_t st);      // experiment switch SNERF_TRAIN_DY_BF16 only
// broadcast per-group rows [G,C] to per-point rows [G*S, C] and the transposed reduction
hipError_t launch_bcast_rows(const float* src, int C, float* dst, int64_t ld_dst, int col0, int64_t n, int n_samples, hipStream_t st);
hipError_t launch_reduce_rows(const float* src, int64_t ld_src, int col0, int C, float* dst, int64_t n_groups, int n_samples, hipStream_t st);
// the scalar loss terms of get_loss (MSE colour loss, solar rays, default solar model, no prior) and their gradients (train_kernels.hip)
struct LossArgs {
    int64_t R, Rs;                             // image rays, sun rays
    int S;
    const float *rgb, *gt, *albedo, *sky;      // [R,3] each (sky: per ray, after the sigmoid)
    const float *sv, *pv, *pe;                 // sun-ray pass [Rs,S]: Solar_Vis, PV_Exact, PE
    const float* alb_min_in;                   // optional [3]: the albedo minimum over the GLOBAL batch (data parallel); NULL: the local one
    int world;                                 // ranks the global batch spans (1 without alb_min_in)
};
hipError_t launch_loss_scratch_init(void* scratch, hipStream_t st);
hipError_t launch_loss_terms(const LossArgs& a, void* scratch, float* vals5, float* minv3, hipStream_t st);
hipError_t launch_loss_terms_bwd(const LossArgs& a, const float* g5, const float* minv3, float* d_rgb, float* d_albedo, float* d_sky, float* d_sv, hipStream_t st);
hipError_t launch_adam_dev(float* params, const float* grads, float* m, float* v, int64_t n, const float* d_hyper6, hipStream_t st);
// Adam on a flat arena (torch.optim.Adam semantics, no weight decay)
hipError_t launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, int step, hipStream_t st);

}  // namespace snerf
