// Classification head and Linear weight-gradient kernels (gfx950) - the callers on
// either side of the aggregation inside one training epoch.
//
//   sngnn_head_nll   log_softmax (models.py:86,211,303) + nll_loss on the masked rows
//                    + the accuracy count (train.py:81-84, 98-102, 112-116) in ONE pass
//                    over the logits, optionally writing d loss / d logits, instead of
//                    log_softmax, a boolean-mask gather, nll_loss, max and two reductions.
//   sngnn_linear_wgrad  dW = g^T x and db = sum_i g_i for self.lin (models.py:98,222,308):
//                    the [C, F] result reduces over ALL N rows, a shape (K = N >> M, N) for
//                    which the BLAS heuristic runs 10-20x off the HBM roofline; here row
//                    chunks accumulate in registers and a second kernel adds the partials
//                    in fixed order (deterministic, no float atomics).
// Both are HBM-bound streams: 4 N C (+ 4 N C for the gradient) and 4 N (F + C) bytes.
#include "common.h"

namespace sngnn {

__device__ __forceinline__ float wsum(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ float wmax(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ int wmin_i(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = min(v, __shfl_xor(v, m, 64));
    return v;
}

constexpr int HEAD_ROWS_PER_BLOCK = 64;    // 4 waves x 16 rows

// One wave per row (lanes stride the C logits).  sel[i] != 0 marks the rows of the
// mask.  Per-block partial (loss sum, correct count) go to part[]; grad (optional,
// dense [N, C]) = (softmax - onehot) * scale on masked rows, 0 elsewhere.
__global__ __launch_bounds__(256) void k_head(const float *__restrict__ z, const int64_t *__restrict__ y,
                                              const unsigned char *__restrict__ sel, int64_t N, int C,
                                              float scale, float *__restrict__ grad,
                                              float *__restrict__ part)
{
    __shared__ float s_loss[4], s_corr[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float loss = 0.f, corr = 0.f;
    for (int r = 0; r < HEAD_ROWS_PER_BLOCK / 4; ++r) {
        const int64_t i = (int64_t)blockIdx.x * HEAD_ROWS_PER_BLOCK + wave * (HEAD_ROWS_PER_BLOCK / 4) + r;
        if (i >= N) break;
        const bool on = sel[i] != 0;
        if (!on) {
            if (grad) for (int c = lane; c < C; c += 64) grad[i * C + c] = 0.f;
            continue;
        }
        const float *zi = z + i * C;
        float mx = -INFINITY;
        for (int c = lane; c < C; c += 64) mx = fmaxf(mx, zi[c]);
        mx = wmax(mx);
        float se = 0.f;
        int arg = C;
        for (int c = lane; c < C; c += 64) {
            const float v = zi[c];
            se += expf(v - mx);
            if (v == mx) arg = min(arg, c);
        }
        se = wsum(se);
        arg = wmin_i(arg);                          // first maximum, like torch.max on the CPU
        const int yi = (int)y[i];
        const float lse = logf(se);
        loss += -(zi[yi] - mx - lse);
        corr += (arg == yi) ? 1.f : 0.f;
        if (grad) {
            const float inv = scale / se;
            for (int c = lane; c < C; c += 64) {
                float p = expf(zi[c] - mx) * inv;
                if (c == yi) p -= scale;
                grad[i * C + c] = p;
            }
        }
    }
    if (lane == 0) { s_loss[wave] = loss; s_corr[wave] = corr; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = (s_loss[0] + s_loss[1]) + (s_loss[2] + s_loss[3]);
        part[2 * blockIdx.x + 1] = (s_corr[0] + s_corr[1]) + (s_corr[2] + s_corr[3]);
    }
}

// fixed-order sum of the per-block partials: out[0] = loss_sum * scale, out[1] = correct
__global__ __launch_bounds__(256) void k_head_reduce(const float *__restrict__ part, int nblocks,
                                                     float scale, float *__restrict__ out)
{
    __shared__ double s[2][256];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) { a += part[2 * i]; b += part[2 * i + 1]; }
    s[0][threadIdx.x] = a;
    s[1][threadIdx.x] = b;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if (threadIdx.x < m) { s[0][threadIdx.x] += s[0][threadIdx.x + m]; s[1][threadIdx.x] += s[1][threadIdx.x + m]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = (float)(s[0][0] * (double)scale); out[1] = (float)s[1][0]; }
}

// ---------------------------------------------------------------------------
// dW partials: workgroup = (128 features) x (2 halves of a 32-channel tile) over a
// chunk of rows; thread keeps 16 accumulators.
// ---------------------------------------------------------------------------
constexpr int WG_ROWS = 512, WG_FT = 128, WG_CT = 32;

__global__ __launch_bounds__(256) void k_wgrad_partial(const float *__restrict__ g, const float *__restrict__ x,
                                                       int64_t N, int C, int F, float *__restrict__ part)
{
    const int f = blockIdx.x * WG_FT + (threadIdx.x & (WG_FT - 1));
    const int c0 = blockIdx.y * WG_CT + (threadIdx.x >> 7) * (WG_CT / 2);
    const int64_t r0 = (int64_t)blockIdx.z * WG_ROWS, r1 = min(N, r0 + WG_ROWS);
    float acc[WG_CT / 2];
#pragma unroll
    for (int k = 0; k < WG_CT / 2; ++k) acc[k] = 0.f;
    const bool fok = f < F;
    for (int64_t i = r0; i < r1; ++i) {
        const float xv = fok ? x[i * F + f] : 0.f;
        const float *gi = g + i * C + c0;
#pragma unroll
        for (int k = 0; k < WG_CT / 2; ++k) {
            const float gv = (c0 + k < C) ? gi[k] : 0.f;      // same address across the 128 lanes
            acc[k] = fmaf(gv, xv, acc[k]);
        }
    }
    if (fok) {
#pragma unroll
        for (int k = 0; k < WG_CT / 2; ++k)
            if (c0 + k < C) part[((size_t)blockIdx.z * C + (c0 + k)) * F + f] = acc[k];
    }
}

// db partials: part_b[chunk][c] = sum of g over the chunk's rows
__global__ __launch_bounds__(256) void k_bgrad_partial(const float *__restrict__ g, int64_t N, int C,
                                                       float *__restrict__ part_b)
{
    const int64_t r0 = (int64_t)blockIdx.x * WG_ROWS, r1 = min(N, r0 + WG_ROWS);
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int64_t i = r0; i < r1; ++i) s += g[i * C + c];
        part_b[(size_t)blockIdx.x * C + c] = s;
    }
}

// out[j] = sum over chunks (fixed order) of part[chunk][j]
__global__ __launch_bounds__(256) void k_sum_partials(const float *__restrict__ part, int nchunks,
                                                      int64_t len, float *__restrict__ out)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= len) return;
    float s = 0.f;
    for (int k = 0; k < nchunks; ++k) s += part[(size_t)k * len + j];
    out[j] = s;
}

}  // namespace sngnn

using namespace sngnn;

extern "C" int64_t sngnn_head_workspace_bytes(int64_t N)
{
    return (N + HEAD_ROWS_PER_BLOCK - 1) / HEAD_ROWS_PER_BLOCK * 8 + 256;
}

extern "C" int sngnn_head_nll(const float *logits, const int64_t *y, const unsigned char *row_mask,
                              int64_t N, int C, int64_t n_masked, float *grad_logits,
                              float *loss_and_correct, void *workspace, void *stream)
{
    SN_REQUIRE(N >= 0 && C >= 1, SNGNN_EINVAL, "bad shape");
    SN_REQUIRE(logits && y && row_mask && loss_and_correct && workspace, SNGNN_EINVAL, "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    const int nb = (int)((N + HEAD_ROWS_PER_BLOCK - 1) / HEAD_ROWS_PER_BLOCK);
    const float scale = 1.0f / (float)(n_masked > 0 ? n_masked : 1);     // nll_loss(reduction='mean')
    if (nb > 0)
        k_head<<<nb, 256, 0, st>>>(logits, y, row_mask, N, C, scale, grad_logits, (float *)workspace);
    k_head_reduce<<<1, 256, 0, st>>>((const float *)workspace, nb, scale, loss_and_correct);
    SN_HIP(hipGetLastError());
    return SNGNN_OK;
}

extern "C" int64_t sngnn_linear_wgrad_workspace_bytes(int64_t N, int C, int F)
{
    const int64_t chunks = (N + WG_ROWS - 1) / WG_ROWS;
    return chunks * (int64_t)C * (F + 1) * 4 + 256;
}

extern "C" int sngnn_linear_wgrad(const float *grad_out, const float *x, int64_t N, int C, int F,
                                  float *grad_weight, float *grad_bias, void *workspace, void *stream)
{
    SN_REQUIRE(N >= 0 && C >= 1 && F >= 1, SNGNN_EINVAL, "bad shape");
    SN_REQUIRE(grad_out && x && grad_weight && workspace, SNGNN_EINVAL, "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    const int chunks = (int)((N + WG_ROWS - 1) / WG_ROWS);
    float *part = (float *)workspace;
    float *part_b = part + (size_t)chunks * C * F;
    if (chunks > 0) {
        dim3 grid((F + WG_FT - 1) / WG_FT, (C + WG_CT - 1) / WG_CT, chunks);
        k_wgrad_partial<<<grid, 256, 0, st>>>(grad_out, x, N, C, F, part);
        if (grad_bias) k_bgrad_partial<<<chunks, 256, 0, st>>>(grad_out, N, C, part_b);
    }
    const int64_t len = (int64_t)C * F;
    k_sum_partials<<<(unsigned)((len + 255) / 256), 256, 0, st>>>(part, chunks, len, grad_weight);
    if (grad_bias) k_sum_partials<<<(C + 255) / 256, 256, 0, st>>>(part_b, chunks, C, grad_bias);
    SN_HIP(hipGetLastError());
    return SNGNN_OK;
}
