// Does the SHAPE of the store stream matter once the data really goes to HBM?  Copy a [393216 x 256] fp32 array (403 MB) to another,
// one 512-thread workgroup per CU, persistent over 256-row tiles as the row GEMMs are, loads always 16 B per lane / 1 KiB contiguous
// per wave; stores in three shapes:
//   0: 16 B per lane, 1 KiB contiguous per wave instruction           (a plain copy)
//   1: dword per lane in the 16x16 accumulator layout: 4 x 64 B row segments per instruction, only columns [128 c, 128 c + 128) of a
//      row by one workgroup (the other half by another workgroup, later)   (gemm_rows16_kernel)
//   2: as 1 but a workgroup writes whole rows (both column halves)     (gemm_wreg_kernel)
// build: hipcc -O3 --offload-arch=gfx950 copy_patterns.hip -o copy_patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE>
__global__ __launch_bounds__(512) void copyk(const float* __restrict__ A, float* __restrict__ C, int64_t M) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int groups = SHAPE == 1 ? 2 : 1;
    const int grp = blockIdx.x % groups, worker = blockIdx.x / groups, n_workers = gridDim.x / groups;
    const int64_t n_tiles = M / 256;
    for (int64_t t = worker; t < n_tiles; t += n_workers) {
        const int64_t row0 = t * 256 + wave * 32;                   // this wave: 32 rows x 256 columns = 32 KiB in, 32 (or 16) KiB out
        f32x4 v[SHAPE == 1 ? 16 : 32];
        // loads: 1 KiB contiguous per instruction (row r of the wave's 32, all 256 columns)
#pragma unroll
        for (int r = 0; r < (SHAPE == 1 ? 16 : 32); ++r) v[r] = *(const f32x4*)(A + (row0 + (SHAPE == 1 ? 2 * r + grp : r)) * 256 + lane * 4);
        if (SHAPE == 0) {
#pragma unroll
            for (int r = 0; r < 32; ++r) *(f32x4*)(C + (row0 + r) * 256 + lane * 4) = v[r];
        } else {
            const int jj = lane & 15, g = lane >> 4;
            const int ncol = SHAPE == 1 ? 128 : 256, c0 = SHAPE == 1 ? 128 * grp : 0;
            int q = 0;
#pragma unroll
            for (int j = 0; j < ncol / 16; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e, ++q)
                        C[(row0 + 16 * h + 4 * g + e) * 256 + c0 + 16 * j + jj] = v[q / 4][q % 4];     // values do not matter, the traffic does
        }
    }
}

// The same plain copy with only D loads (D KiB) in flight per wave and batch: how much memory-level parallelism the copy rate needs.
template <int D>
__global__ __launch_bounds__(512) void copyd(const float* __restrict__ A, float* __restrict__ C, int64_t M) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n_tiles = M / (8 * D);
    for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int64_t row0 = (t * 8 + wave) * D;
        f32x4 v[D];
#pragma unroll
        for (int r = 0; r < D; ++r) v[r] = *(const f32x4*)(A + (row0 + r) * 256 + lane * 4);
#pragma unroll
        for (int r = 0; r < D; ++r) *(f32x4*)(C + (row0 + r) * 256 + lane * 4) = v[r];
    }
}

// D KiB in flight per wave as in copyd, but the workgroup stays on a tile of 8 x TR rows (TR / D batches) before it moves on: the chip-wide
// window of rows in work is 256 x 8 x TR KiB instead of 256 x 8 x D - which of the two (in flight, window) does the rate follow?
template <int D, int TR>
__global__ __launch_bounds__(512) void copyw(const float* __restrict__ A, float* __restrict__ C, int64_t M) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n_tiles = M / (8 * TR);
    for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        for (int b = 0; b < TR / D; ++b) {
            const int64_t row0 = (t * 8 + wave) * TR + b * D;
            f32x4 v[D];
#pragma unroll
            for (int r = 0; r < D; ++r) v[r] = *(const f32x4*)(A + (row0 + r) * 256 + lane * 4);
#pragma unroll
            for (int r = 0; r < D; ++r) *(f32x4*)(C + (row0 + r) * 256 + lane * 4) = v[r];
        }
    }
}

template <int D, int TR>
static void runw(const float* A, float* C, int64_t M) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((copyw<D, TR>), dim3(256), dim3(512), 0, 0, A, C, M);
    (void)hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((copyw<D, TR>), dim3(256), dim3(512), 0, 0, A, C, M);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms / reps * 1e3, bytes = (double)M * 256 * 4 * 2;
    printf("copy, %2d KiB in flight per wave, tiles of %3d rows per wave (window %3d MiB)  %7.1f us per launch  %.2f TB/s\n", D, TR, 2 * TR, us, bytes / (us * 1e-6) / 1e12);
}

template <int D>
static void rund(const float* A, float* C, int64_t M) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(copyd<D>, dim3(256), dim3(512), 0, 0, A, C, M);
    (void)hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(copyd<D>, dim3(256), dim3(512), 0, 0, A, C, M);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms / reps * 1e3, bytes = (double)M * 256 * 4 * 2;
    printf("copy, %2d KiB in flight per wave (%3d KiB per CU)              %7.1f us per launch  %.2f TB/s (read + write)\n", D, 8 * D, us, bytes / (us * 1e-6) / 1e12);
}

template <int SHAPE>
static void run(const float* A, float* C, int64_t M, const char* name) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(copyk<SHAPE>, dim3(256), dim3(512), 0, 0, A, C, M);
    (void)hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(copyk<SHAPE>, dim3(256), dim3(512), 0, 0, A, C, M);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms / reps * 1e3, bytes = (double)M * 256 * 4 * 2;
    printf("%-58s %7.1f us per launch  %.2f TB/s (read + write)\n", name, us, bytes / (us * 1e-6) / 1e12);
}

int main() {
    const int64_t M = 393216;
    float *A, *C;
    (void)hipMalloc(&A, M * 256 * 4); (void)hipMalloc(&C, M * 256 * 4);
    (void)hipMemset(A, 0, M * 256 * 4);
    run<0>(A, C, M, "copy, 16 B per lane, 1 KiB contiguous stores");
    run<1>(A, C, M, "dword stores, 4 x 64 B, half rows per workgroup");
    run<2>(A, C, M, "dword stores, 4 x 64 B, whole rows per workgroup");
    run<0>(A, C, M, "copy, 16 B per lane, 1 KiB contiguous stores");
    rund<2>(A, C, M); rund<4>(A, C, M); rund<8>(A, C, M); rund<16>(A, C, M); rund<32>(A, C, M); rund<64>(A, C, M);
    runw<2, 32>(A, C, M); runw<4, 32>(A, C, M); runw<2, 8>(A, C, M); runw<8, 32>(A, C, M); runw<2, 128>(A, C, M); runw<32, 128>(A, C, M);
    return 0;
}
