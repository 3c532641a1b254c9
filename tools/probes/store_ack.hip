// How long does a burst of accumulator-layout stores take to be ACKNOWLEDGED while the whole chip streams (reads + writes at the rate
// of a copy)?  On gfx9 loads and stores retire through ONE in-order counter (vmcnt), so a load issued after a store burst cannot be
// waited for before the burst is acknowledged: the acknowledgement latency is what a row GEMM's first refill after its epilogue pays.
// One 512-thread workgroup per CU, persistent over 256-row tiles of a [393216 x 256] fp32 array; per tile a wave loads 16 KiB
// (16 x 1 KiB), waits, stores 16 KiB as 64 dword instructions (4 x 64 B each, the 16x16 accumulator layout, half rows), then
//   MODE 0: waits for vmcnt(0) right away and clocks the wait            (acknowledgement latency of the burst)
//   MODE 1: does GAP cycles of s_sleep-free busy work first, then waits  (how much of the latency a given distance hides)
//   MODE 2: never waits for the stores (only what the next loads' wait implies: loads are older than nothing here)
// build: hipcc -O3 --offload-arch=gfx950 store_ack.hip -o store_ack
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>
__global__ __launch_bounds__(512) void k(const float* __restrict__ A, float* __restrict__ C, int64_t M, uint64_t* out, int gap) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = blockIdx.x & 1, worker = blockIdx.x >> 1, n_workers = gridDim.x >> 1;
    const int jj = lane & 15, g = lane >> 4;
    uint64_t waited = 0, n = 0, worst = 0;
    const uint64_t k0 = __builtin_amdgcn_s_memtime();
    for (int64_t t = worker; t < M / 256; t += n_workers) {
        const int64_t row0 = t * 256 + wave * 32;
        f32x4 v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = *(const f32x4*)(A + (row0 + 2 * r + grp) * 256 + lane * 4);
        int q = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 4; ++e, ++q) C[(row0 + 16 * h + 4 * g + e) * 256 + 128 * grp + 16 * j + jj] = v[q / 4][q % 4];
        if (MODE == 1) {
            float x = v[0][0];
            for (int i = 0; i < gap; i += 8) asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0" : "+v"(x));
            if (x == 123.456f) C[0] = x;
        }
        if (MODE != 2) {
            const uint64_t t0 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint64_t t1 = __builtin_amdgcn_s_memtime();
            waited += t1 - t0; ++n;
            if (t1 - t0 > worst) worst = t1 - t0;
        }
    }
    const uint64_t k1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint64_t k2 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
        uint64_t* o = out + (blockIdx.x * 8 + wave) * 5;
        o[0] = waited; o[1] = n; o[2] = worst; o[3] = k1 - k0; o[4] = k2 - k1;
    }
}

template <int MODE>
static void run(const float* A, float* C, int64_t M, uint64_t* d_out, int gap, const char* name) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, A, C, M, d_out, gap);
    (void)hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, A, C, M, d_out, gap);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    static uint64_t h[256 * 8 * 5];
    (void)hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    double w = 0, n = 0, loop = 0, tail = 0; uint64_t worst = 0;
    for (int i = 0; i < 256 * 8; ++i) { w += h[5 * i]; n += h[5 * i + 1]; if (h[5 * i + 2] > worst) worst = h[5 * i + 2]; loop += h[5 * i + 3]; tail += h[5 * i + 4]; }
    printf("%-44s %7.1f us per launch | wait per burst: mean %7.0f cycles, worst %7llu | wave loop %8.0f cycles, final drain %7.0f cycles\n",
           name, ms / reps * 1e3, n ? w / n : 0.0, (unsigned long long)worst, loop / 2048, tail / 2048);
}

int main() {
    const int64_t M = 393216;
    float *A, *C; uint64_t* d_out;
    (void)hipMalloc(&A, M * 256 * 4); (void)hipMalloc(&C, M * 256 * 4); (void)hipMalloc(&d_out, 256 * 8 * 5 * 8);
    (void)hipMemset(A, 0, M * 256 * 4);
    run<0>(A, C, M, d_out, 0, "wait right after the burst");
    run<1>(A, C, M, d_out, 2000, "2000 cycles of other work, then wait");
    run<1>(A, C, M, d_out, 6000, "6000 cycles of other work, then wait");
    run<1>(A, C, M, d_out, 12000, "12000 cycles of other work, then wait");
    run<2>(A, C, M, d_out, 0, "no wait for the stores");
    return 0;
}
