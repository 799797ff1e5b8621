// Micro-benchmark: the classification head's pass (log_softmax + NLL + arg-max [+ gradient]) over [N, 40] logits in the
// tree's layout (16 lanes x 4 channels per row: head_row.h) against an 8-lane layout (8 lanes x 2 x 4 channels: a
// row's 40 channels in 8 + 2 lanes' vectors, eight rows per wave instruction, three-step reductions).
//   hipcc --offload-arch=gfx950 -O3 -I../../sngnn_amd/csrc -I../../include head_g8.hip -o head_g8 && ./head_g8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "head_row.h"

using namespace sngnn;

template <bool GRAD>
__global__ __launch_bounds__(256) void k_g16(const float *__restrict__ z, const int64_t *__restrict__ y,
                                             const unsigned char *__restrict__ sel, int64_t N, int C, float scale,
                                             float *__restrict__ grad, float *__restrict__ part)
{
    constexpr int G = 16, RPW = 4, U = 2;
    const int lane = threadIdx.x & 63;
    const int gid = lane / G, lg = lane % G;
    const bool in = 4 * lg < C;
    const int c0 = in ? 4 * lg : 0;
    float loss = 0.f, corr = 0.f;
    const int64_t nw = (int64_t)gridDim.x * 4, w0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int64_t base = w0 * (RPW * U); base < N; base += nw * (RPW * U)) {
        float4 t[U]; int yi[U]; unsigned char sv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + u * RPW + gid, ic = i < N ? i : N - 1;
            sv[u] = i < N ? sel[ic] : (unsigned char)0;
            t[u] = *reinterpret_cast<const float4 *>(z + ic * C + c0);
            yi[u] = (int)y[ic];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + u * RPW + gid;
            if (i >= N) continue;
            float *gi = GRAD ? grad + i * C + c0 : nullptr;
            if (sv[u] == 0) { if (GRAD && in) *reinterpret_cast<float4 *>(gi) = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
            const HeadRow hr = head_row<G>(t[u], in, c0, yi[u]);
            if (lg == 0) { loss += hr.loss; corr += hr.corr; }
            if (GRAD && in) *reinterpret_cast<float4 *>(gi) = head_row_grad(hr, scale);
        }
    }
    for (int m = 32; m >= 1; m >>= 1) { loss += __shfl_xor(loss, m, 64); corr += __shfl_xor(corr, m, 64); }
    if (lane == 0) { const int w = blockIdx.x * 4 + (threadIdx.x >> 6); part[2 * w] = loss; part[2 * w + 1] = corr; }
}

template <int CTRL> __device__ __forceinline__ float dmax(float v) { return dpp_maxf<CTRL>(v); }

// 8 lanes per row: lane lg holds channels [4 lg, 4 lg + 4) (a) and [32 + 4 lg, 36 + 4 lg) (b)
template <bool GRAD>
__global__ __launch_bounds__(256) void k_g8(const float *__restrict__ z, const int64_t *__restrict__ y,
                                            const unsigned char *__restrict__ sel, int64_t N, int C, float scale,
                                            float *__restrict__ grad, float *__restrict__ part)
{
    constexpr int G = 8, RPW = 8, U = 2;
    const int lane = threadIdx.x & 63;
    const int gid = lane / G, lg = lane % G;
    const bool ina = 4 * lg < C, inb = 32 + 4 * lg < C;
    const int ca = ina ? 4 * lg : 0, cb = inb ? 32 + 4 * lg : 0;
    float loss = 0.f, corr = 0.f;
    const int64_t nw = (int64_t)gridDim.x * 4, w0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int64_t base = w0 * (RPW * U); base < N; base += nw * (RPW * U)) {
        float4 ta[U], tb[U]; int yi[U]; unsigned char sv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + u * RPW + gid, ic = i < N ? i : N - 1;
            sv[u] = i < N ? sel[ic] : (unsigned char)0;
            ta[u] = *reinterpret_cast<const float4 *>(z + ic * C + ca);
            tb[u] = *reinterpret_cast<const float4 *>(z + ic * C + cb);
            yi[u] = (int)y[ic];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + u * RPW + gid;
            if (i >= N) continue;
            float *gi = GRAD ? grad + i * C : nullptr;
            if (sv[u] == 0) {
                if (GRAD && ina) *reinterpret_cast<float4 *>(gi + ca) = make_float4(0.f, 0.f, 0.f, 0.f);
                if (GRAD && inb) *reinterpret_cast<float4 *>(gi + cb) = make_float4(0.f, 0.f, 0.f, 0.f);
                continue;
            }
            const float NI = -INFINITY;
            float v[8] = {ina ? ta[u].x : NI, ina ? ta[u].y : NI, ina ? ta[u].z : NI, ina ? ta[u].w : NI,
                          inb ? tb[u].x : NI, inb ? tb[u].y : NI, inb ? tb[u].z : NI, inb ? tb[u].w : NI};
            float mx = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7])));
            mx = dmax<0xB1>(mx); mx = dmax<0x4E>(mx); mx = dmax<0x141>(mx);
            int a = 1 << 30;
#pragma unroll
            for (int q = 7; q >= 4; --q) if (v[q] == mx) a = cb + q - 4;
#pragma unroll
            for (int q = 3; q >= 0; --q) if (v[q] == mx) a = ca + q;
            a = dpp_mini<0xB1>(a); a = dpp_mini<0x4E>(a); a = dpp_mini<0x141>(a);
            float e[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) e[q] = ((q < 4 ? ina : inb)) ? fast_exp_neg(v[q] - mx) : 0.f;
            float se = ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
            se = dpp_addf<0xB1>(se); se = dpp_addf<0x4E>(se); se = dpp_addf<0x141>(se);
            const int ka = yi[u] - ca, kb = yi[u] - cb;
            float mine = 0.f;
            if (ina && ka >= 0 && ka < 4) mine = ka == 0 ? v[0] : ka == 1 ? v[1] : ka == 2 ? v[2] : v[3];
            if (inb && kb >= 0 && kb < 4) mine = kb == 0 ? v[4] : kb == 1 ? v[5] : kb == 2 ? v[6] : v[7];
            float zy = mine;
            zy = dpp_addf<0xB1>(zy); zy = dpp_addf<0x4E>(zy); zy = dpp_addf<0x141>(zy);
            const float rl = -(zy - mx - __builtin_amdgcn_logf(se) * 0.6931471805599453f);
            if (lg == 0) { loss += rl; corr += (a == yi[u]) ? 1.f : 0.f; }
            if (GRAD) {
                const float inv = scale / se;
                if (ina) *reinterpret_cast<float4 *>(gi + ca) = make_float4(e[0] * inv - (ka == 0 ? scale : 0.f), e[1] * inv - (ka == 1 ? scale : 0.f),
                                                                          e[2] * inv - (ka == 2 ? scale : 0.f), e[3] * inv - (ka == 3 ? scale : 0.f));
                if (inb) *reinterpret_cast<float4 *>(gi + cb) = make_float4(e[4] * inv - (kb == 0 ? scale : 0.f), e[5] * inv - (kb == 1 ? scale : 0.f),
                                                                          e[6] * inv - (kb == 2 ? scale : 0.f), e[7] * inv - (kb == 3 ? scale : 0.f));
            }
        }
    }
    for (int m = 32; m >= 1; m >>= 1) { loss += __shfl_xor(loss, m, 64); corr += __shfl_xor(corr, m, 64); }
    if (lane == 0) { const int w = blockIdx.x * 4 + (threadIdx.x >> 6); part[2 * w] = loss; part[2 * w + 1] = corr; }
}

int sngnn::launch_head_reduce(const float *, int, float, float, int, int, float *, hipStream_t) { return 0; }

int main()
{
    const int64_t N = 169343; const int C = 40;
    std::vector<float> hz(N * C); std::vector<int64_t> hy(N); std::vector<unsigned char> hs(N);
    srand(1);
    for (auto &v : hz) v = (rand() / (float)RAND_MAX - 0.5f) * 6.f;
    for (int64_t i = 0; i < N; ++i) { hy[i] = rand() % C; hs[i] = (rand() % 10) < 6; }
    float *z, *g, *part; int64_t *y; unsigned char *s;
    hipMalloc(&z, N * C * 4); hipMalloc(&g, N * C * 4); hipMalloc(&part, 8192 * 4 * 8); hipMalloc(&y, N * 8); hipMalloc(&s, N);
    hipMemcpy(z, hz.data(), N * C * 4, hipMemcpyHostToDevice); hipMemcpy(y, hy.data(), N * 8, hipMemcpyHostToDevice);
    hipMemcpy(s, hs.data(), N, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto launch) {
        float out[2];
        for (int w = 0; w < 50; ++w) launch();
        hipMemset(part, 0, 8192 * 4 * 8); launch();
        { std::vector<float> hp(8192 * 4 * 2); hipMemcpy(hp.data(), part, 8192 * 4 * 8, hipMemcpyDeviceToHost);
          double a = 0, b = 0; for (size_t q = 0; q < hp.size(); q += 2) { a += hp[q]; b += hp[q + 1]; } out[0] = (float)a; out[1] = (float)b; }
        hipEventRecord(e0);
        for (int w = 0; w < 200; ++w) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %7.2f us   loss sum %.4f correct %.0f\n", name, ms / 200 * 1e3, out[0], out[1]);
    };
    for (int blocks : {2048, 4096, 8192}) {
        printf("blocks %d\n", blocks);
        run("16 lanes x 4, metrics", [&] { k_g16<false><<<blocks, 256>>>(z, y, s, N, C, 1e-5f, g, part); });
        run("8 lanes x 8, metrics", [&] { k_g8<false><<<blocks, 256>>>(z, y, s, N, C, 1e-5f, g, part); });
        run("16 lanes x 4, + gradient", [&] { k_g16<true><<<blocks, 256>>>(z, y, s, N, C, 1e-5f, g, part); });
        run("8 lanes x 8, + gradient", [&] { k_g8<true><<<blocks, 256>>>(z, y, s, N, C, 1e-5f, g, part); });
    }
    return 0;
}
