// Micro-benchmark: where the time of the symmetric fp32-MFMA cosine tile kernel goes.
// Variants of the 128x128-tile kernel with parts switched off (results are then wrong; timing only):
//   MODE 0 full   1 no global loads (panels staged once)   2 no LDS reads (constant operands)
//   3 MFMAs only (no staging, no LDS, no barriers)   4 full, panels prefetched two K-steps ahead
//   hipcc --offload-arch=gfx950 -O3 tools/micro/cosine_mfma_bench.hip -o /tmp/cosb && /tmp/cosb [N] [F]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int TB_M = 128, TB_K = 32, TB_LD = TB_K + 4;

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_tile(const float *__restrict__ x, int64_t N, int64_t F,
                                              float *__restrict__ S, int nb)
{
    __shared__ __align__(16) float sA[TB_M * TB_LD];
    __shared__ __align__(16) float sB[TB_M * TB_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    int by = 0, rem = blockIdx.x;
    while (rem >= nb - by) { rem -= nb - by; ++by; }
    const int bx = by + rem;
    const int64_t row0 = (int64_t)by * TB_M, col0 = (int64_t)bx * TB_M;
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int kq = tid & 7, r0 = tid >> 3;
    typedef float f4 __attribute__((ext_vector_type(4)));
    // loop-carried prefetch registers: NAMED vector variables and straight-line code (no
    // lambdas, no arrays, no struct float4): anything the compiler keeps in a stack slot across
    // the loop's back edge is written to scratch memory right behind its load - which waits for it
    f4 va0, va1, va2, va3, vb0, vb1, vb2, vb3;
    f4 vc0, vc1, vc2, vc3, vd0, vd1, vd2, vd3;      // MODE 4: second prefetch stage
    const float *ga0 = x + min(row0 + r0, N - 1) * F + 4 * kq, *ga1 = x + min(row0 + r0 + 32, N - 1) * F + 4 * kq;
    const float *ga2 = x + min(row0 + r0 + 64, N - 1) * F + 4 * kq, *ga3 = x + min(row0 + r0 + 96, N - 1) * F + 4 * kq;
    const float *gb0 = x + min(col0 + r0, N - 1) * F + 4 * kq, *gb1 = x + min(col0 + r0 + 32, N - 1) * F + 4 * kq;
    const float *gb2 = x + min(col0 + r0 + 64, N - 1) * F + 4 * kq, *gb3 = x + min(col0 + r0 + 96, N - 1) * F + 4 * kq;
    const bool oa0 = row0 + r0 < N, oa1 = row0 + r0 + 32 < N, oa2 = row0 + r0 + 64 < N, oa3 = row0 + r0 + 96 < N;
    const bool ob0 = col0 + r0 < N, ob1 = col0 + r0 + 32 < N, ob2 = col0 + r0 + 64 < N, ob3 = col0 + r0 + 96 < N;
    float *wa = sA + r0 * TB_LD + 4 * kq, *wb = sB + r0 * TB_LD + 4 * kq;
    const f4 z = {0.f, 0.f, 0.f, 0.f};
#define SN_FETCH(K0)                                                                         \
    {                                                                                        \
        const int64_t kk_ = (4 * kq + (K0) < F) ? (K0) : -(int64_t)(4 * kq);                 \
        va0 = *(const f4 *)(ga0 + kk_); va1 = *(const f4 *)(ga1 + kk_);                      \
        va2 = *(const f4 *)(ga2 + kk_); va3 = *(const f4 *)(ga3 + kk_);                      \
        vb0 = *(const f4 *)(gb0 + kk_); vb1 = *(const f4 *)(gb1 + kk_);                      \
        vb2 = *(const f4 *)(gb2 + kk_); vb3 = *(const f4 *)(gb3 + kk_);                      \
    }
#define SN_STAGE(K0)                                                                         \
    {                                                                                        \
        const bool kin_ = 4 * kq + (K0) < F;                                                 \
        *(f4 *)(wa) = (oa0 && kin_) ? va0 : z;              *(f4 *)(wa + 32 * TB_LD) = (oa1 && kin_) ? va1 : z; \
        *(f4 *)(wa + 64 * TB_LD) = (oa2 && kin_) ? va2 : z; *(f4 *)(wa + 96 * TB_LD) = (oa3 && kin_) ? va3 : z; \
        *(f4 *)(wb) = (ob0 && kin_) ? vb0 : z;              *(f4 *)(wb + 32 * TB_LD) = (ob1 && kin_) ? vb1 : z; \
        *(f4 *)(wb + 64 * TB_LD) = (ob2 && kin_) ? vb2 : z; *(f4 *)(wb + 96 * TB_LD) = (ob3 && kin_) ? vb3 : z; \
    }
#define SN_FETCH2(K0)                                                                        \
    {                                                                                        \
        const int64_t kk_ = (4 * kq + (K0) < F) ? (K0) : -(int64_t)(4 * kq);                 \
        vc0 = *(const f4 *)(ga0 + kk_); vc1 = *(const f4 *)(ga1 + kk_);                      \
        vc2 = *(const f4 *)(ga2 + kk_); vc3 = *(const f4 *)(ga3 + kk_);                      \
        vd0 = *(const f4 *)(gb0 + kk_); vd1 = *(const f4 *)(gb1 + kk_);                      \
        vd2 = *(const f4 *)(gb2 + kk_); vd3 = *(const f4 *)(gb3 + kk_);                      \
    }
#define SN_STAGE2(K0)                                                                        \
    {                                                                                        \
        const bool kin_ = 4 * kq + (K0) < F;                                                 \
        *(f4 *)(wa) = (oa0 && kin_) ? vc0 : z;              *(f4 *)(wa + 32 * TB_LD) = (oa1 && kin_) ? vc1 : z; \
        *(f4 *)(wa + 64 * TB_LD) = (oa2 && kin_) ? vc2 : z; *(f4 *)(wa + 96 * TB_LD) = (oa3 && kin_) ? vc3 : z; \
        *(f4 *)(wb) = (ob0 && kin_) ? vd0 : z;              *(f4 *)(wb + 32 * TB_LD) = (ob1 && kin_) ? vd1 : z; \
        *(f4 *)(wb + 64 * TB_LD) = (ob2 && kin_) ? vd2 : z; *(f4 *)(wb + 96 * TB_LD) = (ob3 && kin_) ? vd3 : z; \
    }
    vc0 = vc1 = vc2 = vc3 = vd0 = vd1 = vd2 = vd3 = z;
    SN_FETCH(0)
    if (MODE == 4) SN_FETCH2(TB_K)
    if (MODE == 1 || MODE == 2) { SN_STAGE(0) __syncthreads(); }
    const int li = lane & 31, lh = lane >> 5;
    for (int64_t k0 = 0; k0 < F; k0 += TB_K) {
        if (MODE == 0) {
            __syncthreads();
            SN_STAGE(k0)
            __syncthreads();
            if (k0 + TB_K < F) SN_FETCH(k0 + TB_K)
        }
        if (MODE == 4) {                                  // two steps ahead, register sets alternate
            __syncthreads();
            if ((k0 / TB_K) & 1) { SN_STAGE2(k0) } else { SN_STAGE(k0) }
            __syncthreads();
            if (k0 + 2 * TB_K < F) { if ((k0 / TB_K) & 1) { SN_FETCH2(k0 + 2 * TB_K) } else { SN_FETCH(k0 + 2 * TB_K) } }
        }
        const float *pa0 = sA + (wr * 64 + li) * TB_LD + 16 * lh, *pa1 = pa0 + 32 * TB_LD;
        const float *pb0 = sB + (wc * 64 + li) * TB_LD + 16 * lh, *pb1 = pb0 + 32 * TB_LD;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f4 ca0, ca1, cb0, cb1;
            if (MODE <= 1 || MODE == 4) {
                ca0 = *(const f4 *)(pa0 + 4 * q);
                ca1 = *(const f4 *)(pa1 + 4 * q);
                cb0 = *(const f4 *)(pb0 + 4 * q);
                cb1 = *(const f4 *)(pb1 + 4 * q);
            } else {
                ca0 = ca1 = cb0 = cb1 = f4{1.f + lane, 2.f, 3.f, 4.f + q};
            }
#define SN_MFMA4(E)                                                                             \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca0.E, cb0.E, acc[0][0], 0, 0, 0);    \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca0.E, cb1.E, acc[0][1], 0, 0, 0);    \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca1.E, cb0.E, acc[1][0], 0, 0, 0);    \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca1.E, cb1.E, acc[1][1], 0, 0, 0);
            SN_MFMA4(x) SN_MFMA4(y) SN_MFMA4(z) SN_MFMA4(w)
#undef SN_MFMA4
        }
    }
    // minimal epilogue: the tile itself (lanes along a row)
    for (int ta = 0; ta < 2; ++ta)
        for (int tb = 0; tb < 2; ++tb) {
            const int64_t c = col0 + wc * 64 + tb * 32 + li;
            for (int r = 0; r < 16; ++r) {
                const int64_t rr = row0 + wr * 64 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (rr < N && c < N) S[rr * N + c] = acc[ta][tb][r];
            }
        }
}

template <int MODE> float run(const float *x, int64_t N, int64_t F, float *S, int reps)
{
    const int nb = (int)((N + TB_M - 1) / TB_M), tiles = nb * (nb + 1) / 2;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) k_tile<MODE><<<tiles, 256>>>(x, N, F, S, nb);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) k_tile<MODE><<<tiles, 256>>>(x, N, F, S, nb);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main(int argc, char **argv)
{
    const int64_t N = argc > 1 ? atoll(argv[1]) : 7600, F = argc > 2 ? atoll(argv[2]) : 932;
    float *x, *S;
    hipMalloc(&x, N * F * 4); hipMalloc(&S, N * N * 4);
    std::vector<float> h(N * F);
    for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(x, h.data(), N * F * 4, hipMemcpyHostToDevice);
    const double gf = (double)N * N * F / 1e9;      // flops actually done (upper triangle): N^2 F
    const float t0 = run<0>(x, N, F, S, 10), t1 = run<1>(x, N, F, S, 10), t2 = run<2>(x, N, F, S, 10), t3 = run<3>(x, N, F, S, 10);
    const float t4 = run<4>(x, N, F, S, 10);
    printf("  prefetch two K-steps ahead: %.3f ms (%.1f TF)\n", t4, gf / t4);
    printf("N=%lld F=%lld (%.1f GF done)  full %.3f ms (%.1f TF)   no-global %.3f   no-LDS-reads %.3f   MFMA-only %.3f ms (%.1f TF)\n",
           (long long)N, (long long)F, gf, t0, gf / t0, t1, t2, t3, gf / t3);
    return 0;
}
