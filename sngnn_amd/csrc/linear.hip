// self.lin forward (models/models.py:121,237,324):  h = x W^T + b  for the skinny
// shapes of this path (N rows in the 1e5..1e6 range, C <= 64 outputs).  The product
// is HBM-bound - x is read once (4 N F bytes), h written once - and the BLAS kernel
// picked for it reaches ~2.3 TB/s; here a workgroup streams a 128-row panel of x
// through LDS in coalesced 128-byte segments with the next panel already in
// registers, and multiplies with exact-fp32 MFMA (v_mfma_f32_32x32x2_f32): 4 waves x
// (32 rows x NT*32 columns).  W^T panels come from L2.
#include "common.h"

namespace sngnn {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int LF_M = 128, LF_K = 32, LF_LD = LF_K + 1;

template <int NT>      // NT 32-column tiles of the output (C <= 32 NT)
__global__ __launch_bounds__(256) void k_linear_fwd(const float *__restrict__ x, const float *__restrict__ w,
                                                    const float *__restrict__ b, int64_t N, int F, int C,
                                                    float *__restrict__ h)
{
    __shared__ float sx[LF_M * LF_LD];
    __shared__ float sw[NT * 32 * LF_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * LF_M;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int sc = tid & 31, sr = tid >> 5;             // staging: column sc of rows sr + 8 u
    float rx[LF_M / 8], rw[NT * 4];
    auto fetch = [&](int k0) {
        const int k = k0 + sc;
#pragma unroll
        for (int u = 0; u < LF_M / 8; ++u) {
            const int64_t r = row0 + sr + 8 * u;
            rx[u] = (r < N && k < F) ? x[r * F + k] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < NT * 4; ++u) {
            const int c = sr + 8 * u;
            rw[u] = (c < C && k < F) ? w[(int64_t)c * F + k] : 0.f;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < F; k0 += LF_K) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < LF_M / 8; ++u) sx[(sr + 8 * u) * LF_LD + sc] = rx[u];
#pragma unroll
        for (int u = 0; u < NT * 4; ++u) sw[(sr + 8 * u) * LF_LD + sc] = rw[u];
        __syncthreads();
        if (k0 + LF_K < F) fetch(k0 + LF_K);
#pragma unroll
        for (int kk = 0; kk < LF_K; kk += 2) {
            // A[i = l & 31][k = l >> 5] = x row of this wave's 32-row slab, B[k][j = l & 31] = W[j][k]
            const float a = sx[(wave * 32 + (lane & 31)) * LF_LD + kk + (lane >> 5)];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float bv = sw[(t * 32 + (lane & 31)) * LF_LD + kk + (lane >> 5)];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[t], 0, 0, 0);
            }
        }
    }
    // C/D layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int c = t * 32 + (lane & 31);
        const float bias = (b != nullptr && c < C) ? b[c] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t rr = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (rr < N && c < C) h[rr * C + c] = acc[t][r] + bias;
        }
    }
}

}  // namespace sngnn

using namespace sngnn;

extern "C" int sngnn_linear_forward(const float *x, const float *weight, const float *bias, int64_t N,
                                    int F, int C, float *h, void *stream)
{
    SN_REQUIRE(N >= 0 && F >= 1 && C >= 1, SNGNN_EINVAL, "bad shape");
    SN_REQUIRE(C <= 64, SNGNN_EINVAL, "sngnn_linear_forward handles C <= 64 (use the BLAS for wider layers)");
    if (N == 0) return SNGNN_OK;
    SN_REQUIRE(x && weight && h, SNGNN_EINVAL, "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)((N + LF_M - 1) / LF_M);
    if (C <= 32) k_linear_fwd<1><<<grid, 256, 0, st>>>(x, weight, bias, N, F, C, h);
    else k_linear_fwd<2><<<grid, 256, 0, st>>>(x, weight, bias, N, F, C, h);
    SN_HIP(hipGetLastError());
    return SNGNN_OK;
}
