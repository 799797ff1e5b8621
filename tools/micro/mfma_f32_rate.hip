// Micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 streams shaped
// like the linear kernels' inner block (NCH independent accumulator chains, operands in
// registers).  hipcc --offload-arch=gfx950 -O3 mfma_f32_rate.hip -o mfma_f32_rate
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NCH>
__global__ __launch_bounds__(256) void k16(float *out, int iters, float a0, float b0)
{
    f32x4 acc[NCH];
    for (int c = 0; c < NCH; ++c) acc[c] = f32x4{0, 0, 0, 0};
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u)
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NCH>
__global__ __launch_bounds__(256) void k32(float *out, int iters, float a0, float b0)
{
    f32x16 acc[NCH];
    for (int c = 0; c < NCH; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < NCH; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the same with a DIFFERENT A and B register for every instruction of a 32-step block - the shape of
// k_linear_rows' multiply block (A from 8 ds_read_b128, B = 96 resident W registers)
template <int NCH>
__global__ __launch_bounds__(256) void k16d(float *out, int iters, float a0, float b0)
{
    f32x4 acc[NCH];
    for (int c = 0; c < NCH; ++c) acc[c] = f32x4{0, 0, 0, 0};
    float a[32], b[NCH][32];
    for (int u = 0; u < 32; ++u) {
        a[u] = a0 + threadIdx.x + u;
        for (int c = 0; c < NCH; ++c) b[c][u] = b0 + u + 32 * c;
    }
    for (int u = 0; u < 32; ++u) {
        asm volatile("" : "+v"(a[u]));
        for (int c = 0; c < NCH; ++c) asm volatile("" : "+v"(b[c][u]));
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u)
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[c][u], acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename K> void run(const char *name, K kern, int grid, int iters, double flop_per_mfma, int mfma_per_iter)
{
    float *out;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    kern<<<grid, 256>>>(out, 10, 1.f, 1.f);
    hipEventRecord(e0);
    kern<<<grid, 256>>>(out, iters, 1.f, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)grid * 4 * iters * mfma_per_iter;
    printf("%-28s grid %5d: %8.1f us  %.1f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz, %d waves/SIMD)\n", name, grid,
           ms * 1e3, n * flop_per_mfma / (ms * 1e-3) / 1e12,
           ms * 1e-3 * 2.4e9 / (n / 1024.0), grid / 256);
    hipFree(out);
}

int main()
{
    for (int grid : {256, 512}) {
        run("16x16x4 f32, 1 chain", k16<1>, grid, 200, 2048.0, 32);
        run("16x16x4 f32, 2 chains", k16<2>, grid, 200, 2048.0, 64);
        run("16x16x4 f32, 3 chains", k16<3>, grid, 200, 2048.0, 96);
        run("16x16x4 f32, 4 chains", k16<4>, grid, 200, 2048.0, 128);
        run("16x16x4 f32, 3 chains, distinct A/B", k16d<3>, grid, 200, 2048.0, 96);
        run("16x16x4 f32, 2 chains, distinct A/B", k16d<2>, grid, 200, 2048.0, 64);
        run("32x32x2 f32, 1 chain", k32<1>, grid, 200, 4096.0, 16);
        run("32x32x2 f32, 2 chains", k32<2>, grid, 200, 4096.0, 32);
    }
    return 0;
}
