// Micro-benchmark: 256x256 tile per 4-wave workgroup (each wave 128x128 = 4x4 MFMA blocks of
// 32x32, 256 accumulator registers), upper triangle only, panels [256][32] through LDS.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/cosine_mfma256_bench.hip -o /tmp/cosb256 && /tmp/cosb256 [N] [F]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int TM = 256, TK = 32, LD = TK + 4;

__global__ __launch_bounds__(256, 1) void k_tile256(const float *__restrict__ x, int64_t N, int64_t F,
                                                    float *__restrict__ S, int nb)
{
    extern __shared__ __align__(16) float smem[];
    float *sA = smem, *sB = smem + TM * LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;          // wave's 128x128 quadrant
    int by = 0, rem = blockIdx.x;
    while (rem >= nb - by) { rem -= nb - by; ++by; }
    const int bx = by + rem;
    const int64_t row0 = (int64_t)by * TM, col0 = (int64_t)bx * TM;
    f32x16 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int kq = tid & 7, r0 = tid >> 3;            // rows r0 + 32 u, u = 0..7
    float4 va[8], vb[8];
    auto fetch = [&](int64_t k0) {
        const int64_t k = k0 + 4 * kq;
        const bool kin = k < F;
        const int64_t kc = kin ? k : 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t r_a = row0 + r0 + 32 * u, r_b = col0 + r0 + 32 * u;
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 ta = *reinterpret_cast<const float4 *>(x + min(r_a, N - 1) * F + kc);
            const float4 tb = *reinterpret_cast<const float4 *>(x + min(r_b, N - 1) * F + kc);
            va[u] = (r_a < N && kin) ? ta : z;
            vb[u] = (r_b < N && kin) ? tb : z;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            *reinterpret_cast<float4 *>(sA + (r0 + 32 * u) * LD + 4 * kq) = va[u];
            *reinterpret_cast<float4 *>(sB + (r0 + 32 * u) * LD + 4 * kq) = vb[u];
        }
    };
    fetch(0);
    const int li = lane & 31, lh = lane >> 5;
    for (int64_t k0 = 0; k0 < F; k0 += TK) {
        __syncthreads();
        stage();
        __syncthreads();
        if (k0 + TK < F) fetch(k0 + TK);
        const float *pa = sA + (wr * 128 + li) * LD + 16 * lh;
        const float *pb = sB + (wc * 128 + li) * LD + 16 * lh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 a[4], b[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                a[t] = *reinterpret_cast<const float4 *>(pa + t * 32 * LD + 4 * q);
                b[t] = *reinterpret_cast<const float4 *>(pb + t * 32 * LD + 4 * q);
            }
#define SN_STEP(E)                                                                                  \
            _Pragma("unroll") for (int ta = 0; ta < 4; ++ta)                                         \
            _Pragma("unroll") for (int tb = 0; tb < 4; ++tb)                                         \
                acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ta].E, b[tb].E, acc[ta][tb], 0, 0, 0);
            SN_STEP(x) SN_STEP(y) SN_STEP(z) SN_STEP(w)
#undef SN_STEP
        }
    }
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
            const int64_t c = col0 + wc * 128 + tb * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t rr = row0 + wr * 128 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (rr < N && c < N) S[rr * N + c] = acc[ta][tb][r];
            }
        }
}

int main(int argc, char **argv)
{
    const int64_t N = argc > 1 ? atoll(argv[1]) : 7600, F = argc > 2 ? atoll(argv[2]) : 932;
    float *x, *S;
    hipMalloc(&x, N * F * 4); hipMalloc(&S, N * N * 4);
    std::vector<float> h(N * F);
    for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(x, h.data(), N * F * 4, hipMemcpyHostToDevice);
    const int nb = (int)((N + TM - 1) / TM), tiles = nb * (nb + 1) / 2;
    const size_t lds = 2 * TM * LD * 4;
    hipFuncSetAttribute((const void *)k_tile256, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) k_tile256<<<tiles, 256, lds>>>(x, N, F, S, nb);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) k_tile256<<<tiles, 256, lds>>>(x, N, F, S, nb);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    const double gf = (double)N * N * F / 1e9;
    printf("N=%lld F=%lld tiles=%d: %.3f ms  (%.1f TF on the N^2 F flops done; %s)\n", (long long)N, (long long)F, tiles, ms,
           gf / ms, hipGetErrorString(hipGetLastError()));
    // spot check against a host dot product
    std::vector<float> s(8);
    double maxerr = 0;
    for (int t = 0; t < 8; ++t) {
        const int64_t i = (t * 977) % N, j = i + ((t * 131) % (N - i));
        float v;
        hipMemcpy(&v, S + i * N + j, 4, hipMemcpyDeviceToHost);
        double ref = 0;
        for (int64_t k = 0; k < F; ++k) ref += (double)h[i * F + k] * h[j * F + k];
        maxerr = fmax(maxerr, fabs(ref - v));
    }
    printf("spot check max abs err %.2e\n", maxerr);
    return 0;
}
