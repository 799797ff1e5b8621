// SNGNN++ blend (models/models.py:134): out = beta * out_0 + (1 - beta) * out_1, beta a
// learnable scalar kept on the device.  PyTorch runs this line as five elementwise
// kernels forward and about as many backward, each a full pass over [N, C]; here it is
// one pass each way, with the reference's rounding (every product and the sum rounded
// separately; the library is built with -ffp-contract=off).
#include "common.h"
#include "device_utils.h"

namespace sngnn {

constexpr int BLEND_BLOCKS = 1024;

// EPI: a hidden layer's relu + dropout behind the blend (models.py:81-84) applied in the same store -
// out = keep ? max(blend, 0) * scale : 0, the keep mask drawn from (seed, element index) like the
// aggregation's store epilogue (device_utils.h: sn_dropout_keep).
struct BlendEpi { int relu; const unsigned long long *seed; float p, scale; };

template <bool EPI>
__global__ __launch_bounds__(256) void k_blend_fwd(const float *__restrict__ o0, const float *__restrict__ o1,
                                                   const float *__restrict__ beta, int64_t n4, int64_t n,
                                                   float *__restrict__ out, const BlendEpi e)
{
    const float b = beta[0], nb = 1.0f - b;
    const int64_t stride = (int64_t)gridDim.x * 256;
    unsigned long long sd = 0ull;
    if constexpr (EPI) sd = e.seed ? *e.seed : 0ull;
    auto epi = [&](float v, int64_t idx) -> float {
        if constexpr (EPI) {
            if (e.relu) v = fmaxf(v, 0.f);
            if (e.seed) v = sn_dropout_keep(sd, (unsigned long long)idx, e.p) ? v * e.scale : 0.f;
        }
        return v;
    };
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 x = reinterpret_cast<const float4 *>(o0)[i], y = reinterpret_cast<const float4 *>(o1)[i];
        reinterpret_cast<float4 *>(out)[i] = make_float4(epi(b * x.x + nb * y.x, 4 * i), epi(b * x.y + nb * y.y, 4 * i + 1),
                                                          epi(b * x.z + nb * y.z, 4 * i + 2), epi(b * x.w + nb * y.w, 4 * i + 3));
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        out[i] = epi(b * o0[i] + nb * o1[i], i);
}

// g0 = beta * g, g1 = (1 - beta) * g, per-block partial of d beta = sum g * (out_0 - out_1)
// EPI: g is the gradient of the ACTIVATED output `act` (k_blend_fwd<true>): d = act > 0 ? g * scale : 0 first
template <bool EPI>
__global__ __launch_bounds__(256) void k_blend_bwd(const float *__restrict__ g, const float *__restrict__ o0,
                                                   const float *__restrict__ o1, const float *__restrict__ beta,
                                                   int64_t n4, int64_t n, float *__restrict__ g0,
                                                   float *__restrict__ g1, float *__restrict__ part,
                                                   const float *__restrict__ act, float scale)
{
    __shared__ float s[256];
    const float b = beta[0], nb = 1.0f - b;
    const int64_t stride = (int64_t)gridDim.x * 256;
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 x = reinterpret_cast<const float4 *>(o0)[i], y = reinterpret_cast<const float4 *>(o1)[i];
        float4 d = reinterpret_cast<const float4 *>(g)[i];
        if constexpr (EPI) {
            const float4 a = reinterpret_cast<const float4 *>(act)[i];
            d = make_float4(a.x > 0.f ? d.x * scale : 0.f, a.y > 0.f ? d.y * scale : 0.f,
                            a.z > 0.f ? d.z * scale : 0.f, a.w > 0.f ? d.w * scale : 0.f);
        }
        reinterpret_cast<float4 *>(g0)[i] = make_float4(b * d.x, b * d.y, b * d.z, b * d.w);
        reinterpret_cast<float4 *>(g1)[i] = make_float4(nb * d.x, nb * d.y, nb * d.z, nb * d.w);
        acc += (d.x * (x.x - y.x) + d.y * (x.y - y.y)) + (d.z * (x.z - y.z) + d.w * (x.w - y.w));
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        float d = g[i];
        if constexpr (EPI) d = act[i] > 0.f ? d * scale : 0.f;
        g0[i] = b * d;
        g1[i] = nb * d;
        acc += d * (o0[i] - o1[i]);
    }
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if (threadIdx.x < m) s[threadIdx.x] += s[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = s[0];
}

__global__ __launch_bounds__(256) void k_blend_reduce(const float *__restrict__ part, int nblocks,
                                                      float *__restrict__ dbeta)
{
    __shared__ double s[256];
    double a = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) a += part[i];
    s[threadIdx.x] = a;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if (threadIdx.x < m) s[threadIdx.x] += s[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) dbeta[0] = (float)s[0];
}

static int blend_grid(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>(BLEND_BLOCKS, (n / 4 + 255) / 256)); }

}  // namespace sngnn

using namespace sngnn;

extern "C" int64_t sngnn_blend_workspace_bytes(void) { return (int64_t)BLEND_BLOCKS * 4 + 256; }

extern "C" int sngnn_blend_forward(const float *out0, const float *out1, const float *beta, int64_t n,
                                   float *out, void *stream)
{
    SN_REQUIRE(n >= 0, SNGNN_EINVAL, "negative size");
    if (n == 0) return SNGNN_OK;
    SN_REQUIRE(out0 && out1 && beta && out, SNGNN_EINVAL, "NULL argument");
    const bool al = ((uintptr_t)out0 | (uintptr_t)out1 | (uintptr_t)out) % 16 == 0;
    k_blend_fwd<false><<<blend_grid(n), 256, 0, (hipStream_t)stream>>>(out0, out1, beta, al ? n / 4 : 0, n, out, BlendEpi{});
    SN_HIP(hipGetLastError());
    return SNGNN_OK;
}

extern "C" int sngnn_blend_forward_epilogue(const float *out0, const float *out1, const float *beta, int64_t n,
                                            const sngnn_epilogue_t *epi, float *out, void *stream)
{
    SN_REQUIRE(n >= 0, SNGNN_EINVAL, "negative size");
    SN_REQUIRE(epi != nullptr && epi->bias == nullptr && epi->keep == nullptr && epi->kept_bits == nullptr, SNGNN_EINVAL,
               "the blend's epilogue takes relu and a seeded dropout only");
    SN_REQUIRE(epi->seed == nullptr || (epi->p >= 0.f && epi->p < 1.f && epi->keep_scale > 0.f), SNGNN_EINVAL, "bad p / keep_scale");
    if (n == 0) return SNGNN_OK;
    SN_REQUIRE(out0 && out1 && beta && out, SNGNN_EINVAL, "NULL argument");
    const bool al = ((uintptr_t)out0 | (uintptr_t)out1 | (uintptr_t)out) % 16 == 0;
    const BlendEpi e{epi->relu, (const unsigned long long *)epi->seed, epi->p, epi->seed ? epi->keep_scale : 1.0f};
    k_blend_fwd<true><<<blend_grid(n), 256, 0, (hipStream_t)stream>>>(out0, out1, beta, al ? n / 4 : 0, n, out, e);
    SN_HIP(hipGetLastError());
    return SNGNN_OK;
}

static int blend_backward_impl(const float *grad_out, const float *out0, const float *out1, const float *beta,
                               int64_t n, const float *act, float scale, float *grad0, float *grad1,
                               float *grad_beta, void *workspace, void *stream);

extern "C" int sngnn_blend_backward(const float *grad_out, const float *out0, const float *out1,
                                    const float *beta, int64_t n, float *grad0, float *grad1,
                                    float *grad_beta, void *workspace, void *stream)
{
    return blend_backward_impl(grad_out, out0, out1, beta, n, nullptr, 1.0f, grad0, grad1, grad_beta, workspace, stream);
}

extern "C" int sngnn_blend_backward_epilogue(const float *grad_out, const float *out0, const float *out1,
                                             const float *beta, int64_t n, const float *act, float scale,
                                             float *grad0, float *grad1, float *grad_beta, void *workspace,
                                             void *stream)
{
    SN_REQUIRE(act != nullptr || n == 0, SNGNN_EINVAL, "act is NULL");
    return blend_backward_impl(grad_out, out0, out1, beta, n, act, scale, grad0, grad1, grad_beta, workspace, stream);
}

static int blend_backward_impl(const float *grad_out, const float *out0, const float *out1, const float *beta,
                               int64_t n, const float *act, float scale, float *grad0, float *grad1,
                               float *grad_beta, void *workspace, void *stream)
{
    SN_REQUIRE(n >= 0, SNGNN_EINVAL, "negative size");
    SN_REQUIRE(grad_beta && workspace, SNGNN_EINVAL, "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    int nb = 0;
    if (n > 0) {
        SN_REQUIRE(grad_out && out0 && out1 && beta && grad0 && grad1, SNGNN_EINVAL, "NULL argument");
        const bool al = ((uintptr_t)grad_out | (uintptr_t)out0 | (uintptr_t)out1 | (uintptr_t)grad0 |
                         (uintptr_t)grad1 | (uintptr_t)act) % 16 == 0;
        nb = blend_grid(n);
        if (act)
            k_blend_bwd<true><<<nb, 256, 0, st>>>(grad_out, out0, out1, beta, al ? n / 4 : 0, n, grad0, grad1,
                                                  (float *)workspace, act, scale);
        else
            k_blend_bwd<false><<<nb, 256, 0, st>>>(grad_out, out0, out1, beta, al ? n / 4 : 0, n, grad0, grad1,
                                                   (float *)workspace, nullptr, 1.0f);
    }
    k_blend_reduce<<<1, 256, 0, st>>>((const float *)workspace, nb, grad_beta);
    SN_HIP(hipGetLastError());
    return SNGNN_OK;
}

// ---------------------------------------------------------------------------
// Backward of a hidden layer's store epilogue (sngnn_agg_forward_epilogue) when the consumer of
// the activated tensor did not fold it into its own store (sngnn_linear_forward_masked):
// g_pre = act > 0 ? g * scale : 0 - autograd's threshold_backward and dropout backward
// (models.py:206-209) in one pass.
// ---------------------------------------------------------------------------
namespace sngnn {
__global__ __launch_bounds__(256) void k_mask_grad(const float *__restrict__ g, const float *__restrict__ act, float scale,
                                                   int64_t n4, int64_t n, float *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 d = reinterpret_cast<const float4 *>(g)[i], a = reinterpret_cast<const float4 *>(act)[i];
        reinterpret_cast<float4 *>(out)[i] = make_float4(a.x > 0.f ? d.x * scale : 0.f, a.y > 0.f ? d.y * scale : 0.f,
                                                          a.z > 0.f ? d.z * scale : 0.f, a.w > 0.f ? d.w * scale : 0.f);
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        out[i] = act[i] > 0.f ? g[i] * scale : 0.f;
}
}  // namespace sngnn

extern "C" int sngnn_epilogue_backward(const float *grad, const float *act, float scale, int64_t n, float *grad_pre,
                                       void *stream)
{
    SN_REQUIRE(n >= 0, SNGNN_EINVAL, "negative size");
    if (n == 0) return SNGNN_OK;
    SN_REQUIRE(grad && act && grad_pre, SNGNN_EINVAL, "NULL argument");
    const bool al = ((uintptr_t)grad % 16 == 0) && ((uintptr_t)act % 16 == 0) && ((uintptr_t)grad_pre % 16 == 0);
    const int64_t n4 = al ? n / 4 : 0;
    const int grid = (int)std::min<int64_t>((std::max<int64_t>(n4, 1) + 255) / 256, 2048);
    sngnn::k_mask_grad<<<grid, 256, 0, (hipStream_t)stream>>>(grad, act, scale, n4, n, grad_pre);
    SN_HIP(hipGetLastError());
    return SNGNN_OK;
}
