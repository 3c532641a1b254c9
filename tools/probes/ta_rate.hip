// Vector-memory issue cost of the row GEMM's access shapes, isolated: one 512-thread workgroup per CU, every wave issues ITER
// load (or store) instructions of one shape over a buffer that stays in L2 / Infinity Cache, 8 in flight.
//   shape 0: operand layout      - lane (r, h) reads 16 B of row r (rows 1 KiB apart) at byte 32 h        (64 x 16-B pieces)
//   shape 1: coalesced           - lane l reads 16 B at byte 16 (l % 4) of row l / 4 (a quad = 64 contiguous B)
//   shape 2: fully contiguous    - lane l reads 16 B at 16 l (1 KiB contiguous)
//   shape 3: accumulator stores  - lane (c, h) stores 4 B at row 4 h, column c (two 128-B row segments per instruction)
//   shape 4: 16-byte row stores  - lane l stores 16 B at row l / 8, byte 16 (l % 8) (eight 128-B row segments)
//   shape 5: 16x16 accumulator stores           - lane (g, j) stores 4 B at row 4 g, column j (four 64-B row segments)
//   shape 6: 16x16 transposed accumulator stores - lane (g, j) stores 16 B at row j, byte 16 g (sixteen 64-B row segments)
//   shape 7: shape 6 as a pair - two such stores to the two halves of the same sixteen 128-B lines
// Prints cycles per wave-instruction per CU (all 8 waves issuing).  build: hipcc -O3 --offload-arch=gfx950 ta_rate.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE>
__global__ __launch_bounds__(512) void probe(float* buf, int64_t rows_total, uint64_t* cyc, float* sink, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t ld = 256;                                    // floats per row (1 KiB)
    int64_t row0 = ((int64_t)blockIdx.x * 8 + wave) * 32;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it0 = 0; it0 < iters; it0 += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {                          // 8 instructions in flight per wave
            const int it = it0 + u;
            const int64_t base = (row0 + (int64_t)(it / 16) * 2048 * 32) % (rows_total - 64);
            const int k = (it % 16) * 16;                     // 16 k-steps per row tile
            if (SHAPE == 0) v[u] = *(const f32x4*)(buf + (base + r) * ld + k + h * 8);
            if (SHAPE == 1) v[u] = *(const f32x4*)(buf + (base + (lane >> 2)) * ld + k + (lane & 3) * 4);
            if (SHAPE == 2) v[u] = *(const f32x4*)(buf + (base + (it % 16)) * ld + lane * 4);
            if (SHAPE == 3) buf[(base + 4 * h + (it % 4) + 8 * ((it / 4) % 4)) * ld + ((it / 16) % 8) * 32 + r] = acc[0] + it;
            if (SHAPE == 4) *(f32x4*)(buf + (base + (lane >> 3) + 8 * (it % 4)) * ld + ((it / 4) % 8) * 32 + (lane & 7) * 4) = acc + (float)it;
            if (SHAPE == 5) buf[(base + 4 * (lane >> 4) + (it % 4) + 16 * ((it / 4) % 2)) * ld + ((it / 8) % 16) * 16 + (lane & 15)] = acc[0] + it;
            if (SHAPE == 6) *(f32x4*)(buf + (base + (lane & 15) + 16 * (it % 2)) * ld + ((it / 2) % 16) * 16 + (lane >> 4) * 4) = acc + (float)it;
            if (SHAPE == 7) *(f32x4*)(buf + (base + (lane & 15) + 16 * ((it / 2) % 2)) * ld + ((it / 4) % 8) * 32 + (it % 2) * 16 + (lane >> 4) * 4) = acc + (float)it;
        }
        if (SHAPE < 3) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (blockIdx.x == 0 && lane == 0) { cyc[2 * wave] = t0; cyc[2 * wave + 1] = t1; }
}

static int g_grid = 256;
template <int SHAPE>
static void run(float* buf, int64_t rows, uint64_t* cyc, float* sink, const char* name) {
    const int iters = 4096;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe<SHAPE>, dim3(g_grid), dim3(512), 0, 0, buf, rows, cyc, sink, iters);
    (void)hipDeviceSynchronize();
    uint64_t c[16];
    (void)hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    uint64_t s0 = c[0], e1 = 0;
    for (int w = 0; w < 8; ++w) { if (c[2 * w] < s0) s0 = c[2 * w]; if (c[2 * w + 1] > e1) e1 = c[2 * w + 1]; }
    printf("%-28s %7.1f cycles per wave-instruction per CU (8 waves issuing; %.1f per instruction and wave)\n", name, (double)(e1 - s0) / iters / 8.0, (double)(e1 - s0) / iters);
}

int main(int argc, char** argv) {
    // default 128 MiB: Infinity Cache resident (the cost of the access SHAPE); `./ta_rate 2097152` = 2 GiB: every pass misses it (HBM);
    // `./ta_rate 2048` = 2 MiB: L2 resident.  Second argument: workgroups (= CUs) issuing, default 256 - fewer tell a per-CU limit from a shared one.
    const int64_t rows = argc > 1 ? atoll(argv[1]) : 131072;
    if (argc > 2) g_grid = atoi(argv[2]);
    printf("rows %lld (%.0f MiB), %d workgroups\n", (long long)rows, rows / 1024.0, g_grid);
    float* buf; uint64_t* cyc; float* sink;
    (void)hipMalloc(&buf, rows * 256 * 4); (void)hipMemset(buf, 0, rows * 256 * 4);
    (void)hipMalloc(&cyc, 256); (void)hipMalloc(&sink, 256 * 512 * 4);
    run<0>(buf, rows, cyc, sink, "load, operand layout");
    run<1>(buf, rows, cyc, sink, "load, quad-coalesced");
    run<2>(buf, rows, cyc, sink, "load, contiguous 1 KiB");
    run<3>(buf, rows, cyc, sink, "store dword, 2 x 128 B");
    run<4>(buf, rows, cyc, sink, "store dwordx4, 8 x 128 B");
    run<5>(buf, rows, cyc, sink, "store dword, 4 x 64 B");
    run<6>(buf, rows, cyc, sink, "store dwordx4, 16 x 64 B");
    run<7>(buf, rows, cyc, sink, "store dwordx4, 16 x 64 B pairs");
    return 0;
}
