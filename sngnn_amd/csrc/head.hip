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
#include "head_row.h"

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

constexpr int HEAD_ROWS_PER_BLOCK = 16;    // 4 waves x 4 rows per step
constexpr int HEAD_MAX_BLOCKS = 2048;      // persistent grid: partials stay few

// Lane-per-row variant for C <= 64: a lane keeps its whole row in registers, so the
// log-softmax needs no cross-lane step at all (the wave-per-row form below spends its
// time in three 6-step shuffle reductions per row).  A row is 4 C contiguous bytes: every
// byte of every line a lane touches is used, by that lane.  Same arithmetic per row as
// the wave-per-row form (max, exp(z - max) summed in channel order, first maximum wins).
// TWO: sel[i] is a bit set - bit 0: the row belongs to split A, bit 1: to split B (the
// validation and test masks read off one eval-mode forward); part[] then holds 4 values per block.
template <bool VEC4, bool TWO = false>
__global__ __launch_bounds__(256) void k_head_rows(const float *__restrict__ z, const int64_t *__restrict__ y,
                                                   const unsigned char *__restrict__ sel, int64_t N, int C,
                                                   float scale, float *__restrict__ grad,
                                                   float *__restrict__ part)
{
    __shared__ float s_loss[4], s_corr[4], s_lossb[4], s_corrb[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float loss = 0.f, corr = 0.f, lossb = 0.f, corrb = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
        const unsigned char sv = sel[i];
        const bool on = sv != 0;
        float *gi = grad ? grad + i * C : nullptr;
        if (!on) {
            if (gi) {
                if constexpr (VEC4) {
                    for (int k = 0; 4 * k < C; ++k) reinterpret_cast<float4 *>(gi)[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                } else {
                    for (int c = 0; c < C; ++c) gi[c] = 0.f;
                }
            }
            continue;
        }
        const float *zi = z + i * C;
        const int yi = (int)y[i];
        float v[64];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (4 * k < C) {                       // C is uniform: a scalar branch
                if constexpr (VEC4) {
                    const float4 t = reinterpret_cast<const float4 *>(zi)[k];
                    v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * k + e] = 4 * k + e < C ? zi[4 * k + e] : -INFINITY;
                }
            }
        }
        float mx = -INFINITY, zy = 0.f;
        int arg = 0;
#pragma unroll
        for (int c = 0; c < 64; ++c)
            if (c < C) {
                if (v[c] > mx) { mx = v[c]; arg = c; }          // first maximum
                if (c == yi) zy = v[c];
            }
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < 64; ++c)
            if (c < C) { v[c] = expf(v[c] - mx); se += v[c]; }
        const float row_loss = -(zy - mx - logf(se)), row_corr = (arg == yi) ? 1.f : 0.f;
        if constexpr (TWO) {
            if (sv & 1) { loss += row_loss; corr += row_corr; }
            if (sv & 2) { lossb += row_loss; corrb += row_corr; }
        } else {
            loss += row_loss;
            corr += row_corr;
        }
        if (gi) {
            const float inv = scale / se;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (4 * k < C) {
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = v[4 * k + e] * inv - (4 * k + e == yi ? scale : 0.f);
                    if constexpr (VEC4) {
                        reinterpret_cast<float4 *>(gi)[k] = make_float4(o[0], o[1], o[2], o[3]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (4 * k + e < C) gi[4 * k + e] = o[e];
                    }
                }
            }
        }
    }
    loss = wsum(loss);
    corr = wsum(corr);
    if constexpr (TWO) { lossb = wsum(lossb); corrb = wsum(corrb); }
    if (lane == 0) { s_loss[wave] = loss; s_corr[wave] = corr; s_lossb[wave] = lossb; s_corrb[wave] = corrb; }
    __syncthreads();
    if (threadIdx.x == 0 && TWO) {
        part[4 * blockIdx.x] = (s_loss[0] + s_loss[1]) + (s_loss[2] + s_loss[3]);
        part[4 * blockIdx.x + 1] = (s_corr[0] + s_corr[1]) + (s_corr[2] + s_corr[3]);
        part[4 * blockIdx.x + 2] = (s_lossb[0] + s_lossb[1]) + (s_lossb[2] + s_lossb[3]);
        part[4 * blockIdx.x + 3] = (s_corrb[0] + s_corrb[1]) + (s_corrb[2] + s_corrb[3]);
    } else if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = (s_loss[0] + s_loss[1]) + (s_loss[2] + s_loss[3]);
        part[2 * blockIdx.x + 1] = (s_corr[0] + s_corr[1]) + (s_corr[2] + s_corr[3]);
    }
}

// Lane-GROUP-per-row variant for C % 4 == 0, C <= 64 (round 3): G = 8 or 16 lanes share a row,
// 16 bytes per lane, so a wave-wide load reads 64 / G whole rows as ONE contiguous segment
// (the lane-per-row form above reads 64 rows 4 C bytes apart per instruction: 64 different
// lines for 16 useful bytes each, and the lines must survive in the CU's 32 KB L1 until the
// row's other nine loads come by - 15 us for 27 MB of logits, 1.8 TB/s).  The row's maximum,
// its first arg-max, the sum of the exponentials and the label's logit are 4-step DPP
// reductions inside the group (no LDS, no shuffles).  Two rows per group in flight.
// Same per-row arithmetic up to the order of the exp sum (a fixed tree here).
// (the per-row arithmetic: head_row.h)
// BLEND (z1 != nullptr; round 5): the logits are SNGNN++'s blend of two tensors, z = beta z + (1 - beta) z1
// (models.py:134 in front of :86), formed here with sngnn_blend_forward's own rounding (two products and their
// sum, each rounded) instead of by a pass of its own; zout (optional) receives them.
template <int G, bool TWO>
__global__ __launch_bounds__(256) void k_head_groups(const float *__restrict__ z, const int64_t *__restrict__ y,
                                                     const unsigned char *__restrict__ sel, int64_t N, int C,
                                                     float scale, float *__restrict__ grad,
                                                     float *__restrict__ part, const float *__restrict__ z1 = nullptr,
                                                     const float *__restrict__ beta = nullptr,
                                                     float *__restrict__ zout = nullptr)
{
    constexpr int RPW = 64 / G, U = 2;
    __shared__ float s_loss[4], s_corr[4], s_lossb[4], s_corrb[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gid = lane / G, lg = lane % G;
    const bool in = 4 * lg < C;
    const int c0 = in ? 4 * lg : 0;
    float loss = 0.f, corr = 0.f, lossb = 0.f, corrb = 0.f;
    const int64_t nw = (int64_t)gridDim.x * 4, w0 = (int64_t)blockIdx.x * 4 + wave;
    for (int64_t base = w0 * (RPW * U); base < N; base += nw * (RPW * U)) {
        float4 t[U];
        int yi[U];
        unsigned char sv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {                         // all loads first
            const int64_t i = base + u * RPW + gid;
            const int64_t ic = i < N ? i : N - 1;
            sv[u] = i < N ? sel[ic] : (unsigned char)0;
            t[u] = *reinterpret_cast<const float4 *>(z + ic * C + c0);
            yi[u] = (int)y[ic];
        }
        if (z1 != nullptr) {                                      // (uniform) the blend, then the head on it
            const float b = beta[0], nb_ = 1.0f - b;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = base + u * RPW + gid;
                const int64_t ic = i < N ? i : N - 1;
                const float4 o1 = *reinterpret_cast<const float4 *>(z1 + ic * C + c0);
                t[u] = make_float4(b * t[u].x + nb_ * o1.x, b * t[u].y + nb_ * o1.y, b * t[u].z + nb_ * o1.z, b * t[u].w + nb_ * o1.w);
                if (zout != nullptr && i < N && in) *reinterpret_cast<float4 *>(zout + i * C + c0) = t[u];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + u * RPW + gid;
            if (i >= N) continue;                             // (group-uniform)
            float *gi = grad ? grad + i * C + c0 : nullptr;
            if (sv[u] == 0) {
                if (gi && in) *reinterpret_cast<float4 *>(gi) = make_float4(0.f, 0.f, 0.f, 0.f);
                continue;
            }
            const HeadRow hr = head_row<G>(t[u], in, c0, yi[u]);
            const float row_loss = hr.loss, row_corr = hr.corr;
            if (lg == 0) {
                if constexpr (TWO) {
                    if (sv[u] & 1) { loss += row_loss; corr += row_corr; }
                    if (sv[u] & 2) { lossb += row_loss; corrb += row_corr; }
                } else {
                    loss += row_loss;
                    corr += row_corr;
                }
            }
            if (gi && in) *reinterpret_cast<float4 *>(gi) = head_row_grad(hr, scale);
        }
    }
    loss = wsum(loss);
    corr = wsum(corr);
    if constexpr (TWO) { lossb = wsum(lossb); corrb = wsum(corrb); }
    if (lane == 0) { s_loss[wave] = loss; s_corr[wave] = corr; s_lossb[wave] = lossb; s_corrb[wave] = corrb; }
    __syncthreads();
    if (threadIdx.x == 0 && TWO) {
        part[4 * blockIdx.x] = (s_loss[0] + s_loss[1]) + (s_loss[2] + s_loss[3]);
        part[4 * blockIdx.x + 1] = (s_corr[0] + s_corr[1]) + (s_corr[2] + s_corr[3]);
        part[4 * blockIdx.x + 2] = (s_lossb[0] + s_lossb[1]) + (s_lossb[2] + s_lossb[3]);
        part[4 * blockIdx.x + 3] = (s_corrb[0] + s_corrb[1]) + (s_corrb[2] + s_corrb[3]);
    } else if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = (s_loss[0] + s_loss[1]) + (s_loss[2] + s_loss[3]);
        part[2 * blockIdx.x + 1] = (s_corr[0] + s_corr[1]) + (s_corr[2] + s_corr[3]);
    }
}

// One wave per row.  sel[i] != 0 marks the rows of the mask.  Per-block partial (loss
// sum, correct count) go to part[]; grad (optional, dense [N, C]) = (softmax - onehot)
// * scale on masked rows, 0 elsewhere.  C <= 64: the row lives in one register per lane.
__global__ __launch_bounds__(256) void k_head(const float *__restrict__ z, const int64_t *__restrict__ y,
                                              const unsigned char *__restrict__ sel, int64_t N, int C,
                                              float scale, float *__restrict__ grad,
                                              float *__restrict__ part)
{
    __shared__ float s_loss[4], s_corr[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float loss = 0.f, corr = 0.f;
    constexpr int RPW = HEAD_ROWS_PER_BLOCK / 4;
    for (int64_t blk = blockIdx.x; blk * HEAD_ROWS_PER_BLOCK < N; blk += gridDim.x) {
    const int64_t i0 = blk * HEAD_ROWS_PER_BLOCK + wave * RPW;
    if (C <= 64) {
        float v[RPW];
        int yi[RPW];
        bool on[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {             // all loads first
            const int64_t i = i0 + r;
            on[r] = i < N && sel[i] != 0;
            v[r] = (on[r] && lane < C) ? z[i * C + lane] : -INFINITY;
            yi[r] = on[r] ? (int)y[i] : 0;
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int64_t i = i0 + r;
            if (i >= N) break;
            if (!on[r]) {
                if (grad && lane < C) grad[i * C + lane] = 0.f;
                continue;
            }
            const float mx = wmax(v[r]);
            const float e = lane < C ? expf(v[r] - mx) : 0.f;
            const float se = wsum(e);
            const int arg = wmin_i((lane < C && v[r] == mx) ? lane : C);
            const float zy = __shfl(v[r], yi[r], 64);
            loss += -(zy - mx - logf(se));
            corr += (arg == yi[r]) ? 1.f : 0.f;
            if (grad && lane < C) grad[i * C + lane] = e * (scale / se) - (lane == yi[r] ? scale : 0.f);
        }
    } else {
        for (int r = 0; r < RPW; ++r) {
            const int64_t i = i0 + r;
            if (i >= N) break;
            if (sel[i] == 0) {
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
                const float t = zi[c];
                se += expf(t - mx);
                if (t == mx) arg = min(arg, c);
            }
            se = wsum(se);
            arg = wmin_i(arg);                      // first maximum, like torch.max on the CPU
            const int yy = (int)y[i];
            loss += -(zi[yy] - mx - logf(se));
            corr += (arg == yy) ? 1.f : 0.f;
            if (grad) {
                const float inv = scale / se;
                for (int c = lane; c < C; c += 64)
                    grad[i * C + c] = expf(zi[c] - mx) * inv - (c == yy ? scale : 0.f);
            }
        }
    }
    }   // persistent loop over row blocks
    if (lane == 0) { s_loss[wave] = loss; s_corr[wave] = corr; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = (s_loss[0] + s_loss[1]) + (s_loss[2] + s_loss[3]);
        part[2 * blockIdx.x + 1] = (s_corr[0] + s_corr[1]) + (s_corr[2] + s_corr[3]);
    }
}

// fixed-order sum of the per-block partials: out[0] = loss_sum * scale, out[1] = correct.
// Block b of the launch reduces the pair at offset 2 b of every `stride`-float partial
// (stride 2: one split; stride 4, two blocks: splits A and B with their own scales).
__global__ __launch_bounds__(256) void k_head_reduce(const float *__restrict__ part, int nblocks,
                                                     float scale, float scale_b, int stride,
                                                     float *__restrict__ out)
{
    __shared__ double s[2][256];
    double a = 0.0, b = 0.0;
    part += 2 * blockIdx.x;
    out += 2 * blockIdx.x;
    if (blockIdx.x == 1) scale = scale_b;
    for (int i = threadIdx.x; i < nblocks; i += 256) { a += part[(size_t)stride * i]; b += part[(size_t)stride * i + 1]; }
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
// dW / db partials.  Workgroup = 128 feature lanes x 2 channel halves over a chunk of
// WG_ROWS rows; a thread keeps KACC accumulators (channel tile = 2 KACC).  Rows are
// taken WG_STEP at a time: x values straight to registers (coalesced over f), the
// g tile through LDS (read back as broadcasts), so WG_STEP loads are in flight per
// thread and WG_STEP x KACC FMAs follow.
// ---------------------------------------------------------------------------
constexpr int WG_ROWS = 512, WG_FT = 128, WG_STEP = 16, WG_SUB = 4;   // 4 sub-blocks of 128 rows

// Thread = one feature column f and KACC channels; x[i][f] comes in coalesced vector
// loads (WG_STEP rows in flight), the g tile of the step goes through LDS and is read
// back as 16-byte broadcasts (4 channels per LDS instruction).
template <int KACC>
__global__ __launch_bounds__(256 * WG_SUB) void k_wgrad_partial(const float *__restrict__ g,
                                                                const float *__restrict__ x, int64_t N, int C,
                                                                int F, float *__restrict__ part,
                                                                float *__restrict__ part_b)
{
    constexpr int CT = 2 * KACC;
    static_assert(KACC % 4 == 0, "16-byte LDS reads");
    __shared__ __attribute__((aligned(16))) float sg[WG_SUB][WG_STEP][CT];
    __shared__ float sred[WG_SUB - 1][256];
    const int sub = threadIdx.x >> 8, tid = threadIdx.x & 255;
    const int fl = tid & (WG_FT - 1), half = tid >> 7;
    const int f = blockIdx.x * WG_FT + fl;
    const int ct0 = blockIdx.y * CT, c0 = ct0 + half * KACC;
    const int64_t rb = (int64_t)blockIdx.z * WG_ROWS + sub * (WG_ROWS / WG_SUB);
    const int64_t r1 = min(N, rb + WG_ROWS / WG_SUB);
    float acc[KACC], bacc[KACC];
#pragma unroll
    for (int k = 0; k < KACC; ++k) { acc[k] = 0.f; bacc[k] = 0.f; }
    const bool fok = f < F;
    const bool do_bias = part_b != nullptr && blockIdx.x == 0 && fl == 0;
    int64_t ib = rb;
    for (int it = 0; it < WG_ROWS / WG_SUB / WG_STEP; ib += WG_STEP, ++it) {   // uniform trip count (barriers)
        float xv[WG_STEP];
#pragma unroll
        for (int r = 0; r < WG_STEP; ++r)
            xv[r] = (fok && ib + r < r1) ? x[(ib + r) * F + f] : 0.f;
        __syncthreads();                            // previous step's sg reads are done
        for (int q = tid; q < WG_STEP * CT; q += 256) {
            const int r = q / CT, c = q % CT;
            sg[sub][r][c] = (ib + r < r1 && ct0 + c < C) ? g[(ib + r) * C + ct0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < WG_STEP; ++r)
#pragma unroll
            for (int k4 = 0; k4 < KACC / 4; ++k4) {
                const float4 gv = *reinterpret_cast<const float4 *>(&sg[sub][r][half * KACC + 4 * k4]);
                acc[4 * k4 + 0] = fmaf(gv.x, xv[r], acc[4 * k4 + 0]);
                acc[4 * k4 + 1] = fmaf(gv.y, xv[r], acc[4 * k4 + 1]);
                acc[4 * k4 + 2] = fmaf(gv.z, xv[r], acc[4 * k4 + 2]);
                acc[4 * k4 + 3] = fmaf(gv.w, xv[r], acc[4 * k4 + 3]);
                if (do_bias) {
                    bacc[4 * k4 + 0] += gv.x; bacc[4 * k4 + 1] += gv.y;
                    bacc[4 * k4 + 2] += gv.z; bacc[4 * k4 + 3] += gv.w;
                }
            }
    }
    // combine the four row quarters in fixed order, one accumulator at a time
#pragma unroll
    for (int k = 0; k < KACC; ++k) {
        __syncthreads();
        if (sub > 0) sred[sub - 1][tid] = acc[k];
        __syncthreads();
        if (sub == 0) acc[k] = ((acc[k] + sred[0][tid]) + sred[1][tid]) + sred[2][tid];
        if (part_b != nullptr && blockIdx.x == 0) {       // block-uniform (barriers inside)
            __syncthreads();
            if (sub > 0) sred[sub - 1][tid] = bacc[k];
            __syncthreads();
            if (sub == 0) bacc[k] = ((bacc[k] + sred[0][tid]) + sred[1][tid]) + sred[2][tid];
        }
    }
    if (sub != 0) return;
#pragma unroll
    for (int k = 0; k < KACC; ++k)
        if (c0 + k < C) {
            if (fok) part[((size_t)blockIdx.z * C + (c0 + k)) * F + f] = acc[k];
            if (do_bias) part_b[(size_t)blockIdx.z * C + c0 + k] = bacc[k];
        }
}

// out[j] = sum over chunks of part[chunk][j]: 16 threads per output take every 16th
// chunk, then their sums are added in fixed order (deterministic).  Two independent sums
// (weight and bias gradient) share one launch: blocks [0, nbA) do A, the rest B.
__global__ __launch_bounds__(256) void k_sum_partials(const float *__restrict__ partA, int64_t lenA,
                                                      float *__restrict__ outA, int nbA,
                                                      const float *__restrict__ partB, int64_t lenB,
                                                      float *__restrict__ outB, int nchunks)
{
    __shared__ float s[16][16];
    const bool second = (int)blockIdx.x >= nbA;
    const float *part = second ? partB : partA;
    float *out = second ? outB : outA;
    const int64_t len = second ? lenB : lenA;
    const int blk = second ? blockIdx.x - nbA : blockIdx.x;
    const int o = threadIdx.x & 15, q = threadIdx.x >> 4;
    const int64_t j = (int64_t)blk * 16 + o;
    float a = 0.f;
    if (j < len)
        for (int k = q; k < nchunks; k += 16) a += part[(size_t)k * len + j];
    s[q][o] = a;
    __syncthreads();
    if (q == 0 && j < len) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += s[w][o];
        out[j] = t;
    }
}

}  // namespace sngnn

int sngnn::launch_head_reduce(const float *part, int entries, float scale_a, float scale_b, int sets, int stride,
                              float *out, hipStream_t st)
{
    k_head_reduce<<<sets, 256, 0, st>>>(part, entries, scale_a, scale_b, stride, out);
    SN_HIP(hipGetLastError());
    return SNGNN_OK;
}

using namespace sngnn;

extern "C" int64_t sngnn_head_workspace_bytes(int64_t N)
{
    (void)N;
    return (int64_t)HEAD_MAX_BLOCKS * 16 + 256;
}

extern "C" int sngnn_head_nll2(const float *logits, const int64_t *y, const unsigned char *row_sets,
                               int64_t N, int C, int64_t n_a, int64_t n_b, float *out4, void *workspace,
                               void *stream)
{
    SN_REQUIRE(N >= 0 && C >= 1 && C <= 64, SNGNN_EINVAL, "sngnn_head_nll2 needs 1 <= C <= 64");
    SN_REQUIRE(logits && y && row_sets && out4 && workspace, SNGNN_EINVAL, "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    int nb = (int)std::min<int64_t>((N + 255) / 256, HEAD_MAX_BLOCKS);
    const float sa = 1.0f / (float)(n_a > 0 ? n_a : 1), sb = 1.0f / (float)(n_b > 0 ? n_b : 1);
    const bool vec4 = C % 4 == 0 && (uintptr_t)logits % 16 == 0;
    // (the group kernels take 32 / 64 rows per workgroup and sweep: fill the chip)
    if (vec4) nb = (int)std::min<int64_t>((N + 31) / 32, HEAD_MAX_BLOCKS);
    if (nb > 0 && vec4 && C <= 32)
        k_head_groups<8, true><<<nb, 256, 0, st>>>(logits, y, row_sets, N, C, 0.f, nullptr, (float *)workspace);
    else if (nb > 0 && vec4)
        k_head_groups<16, true><<<nb, 256, 0, st>>>(logits, y, row_sets, N, C, 0.f, nullptr, (float *)workspace);
    else if (nb > 0)
        k_head_rows<false, true><<<nb, 256, 0, st>>>(logits, y, row_sets, N, C, 0.f, nullptr, (float *)workspace);
    k_head_reduce<<<2, 256, 0, st>>>((const float *)workspace, nb, sa, sb, 4, out4);
    SN_HIP(hipGetLastError());
    return SNGNN_OK;
}

extern "C" int sngnn_head_nll_blend_supported(int C) { return (C >= 4 && C <= 64 && C % 4 == 0) ? 1 : 0; }

extern "C" int sngnn_head_nll_blend(const float *out0, const float *out1, const float *beta, const int64_t *y,
                                    const unsigned char *sel, int64_t N, int C, int sets, int64_t n_a, int64_t n_b,
                                    float *grad_logits, float *logits_out, float *metrics, void *workspace, void *stream)
{
    SN_REQUIRE(N >= 0 && sngnn_head_nll_blend_supported(C), SNGNN_EINVAL, "sngnn_head_nll_blend needs C % 4 == 0, C <= 64");
    SN_REQUIRE(sets == 1 || sets == 2, SNGNN_EINVAL, "sets must be 1 or 2");
    SN_REQUIRE(grad_logits == nullptr || sets == 1, SNGNN_EINVAL, "the gradient is one split's");
    SN_REQUIRE(out0 && out1 && beta && y && sel && metrics && workspace, SNGNN_EINVAL, "NULL argument");
    SN_REQUIRE(((uintptr_t)out0 | (uintptr_t)out1 | (uintptr_t)grad_logits | (uintptr_t)logits_out) % 16 == 0, SNGNN_EINVAL,
               "rows must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int nb = (int)std::min<int64_t>((N + 31) / 32, HEAD_MAX_BLOCKS);
    const float sa = 1.0f / (float)(n_a > 0 ? n_a : 1), sb = 1.0f / (float)(n_b > 0 ? n_b : 1);
    float *part = (float *)workspace;
    if (nb > 0 && sets == 2) {
        if (C <= 32) k_head_groups<8, true><<<nb, 256, 0, st>>>(out0, y, sel, N, C, 0.f, nullptr, part, out1, beta, logits_out);
        else k_head_groups<16, true><<<nb, 256, 0, st>>>(out0, y, sel, N, C, 0.f, nullptr, part, out1, beta, logits_out);
    } else if (nb > 0) {
        if (C <= 32) k_head_groups<8, false><<<nb, 256, 0, st>>>(out0, y, sel, N, C, sa, grad_logits, part, out1, beta, logits_out);
        else k_head_groups<16, false><<<nb, 256, 0, st>>>(out0, y, sel, N, C, sa, grad_logits, part, out1, beta, logits_out);
    }
    k_head_reduce<<<sets, 256, 0, st>>>(part, nb, sa, sb, 2 * sets, metrics);
    SN_HIP(hipGetLastError());
    return SNGNN_OK;
}

extern "C" int sngnn_head_nll(const float *logits, const int64_t *y, const unsigned char *row_mask,
                              int64_t N, int C, int64_t n_masked, float *grad_logits,
                              float *loss_and_correct, void *workspace, void *stream)
{
    SN_REQUIRE(N >= 0 && C >= 1, SNGNN_EINVAL, "bad shape");
    SN_REQUIRE(logits && y && row_mask && loss_and_correct && workspace, SNGNN_EINVAL, "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    const bool rows = C <= 64;      // lane-per-row kernel
    const int64_t per_block = rows ? 256 : HEAD_ROWS_PER_BLOCK;
    int nb = (int)std::min<int64_t>((N + per_block - 1) / per_block, HEAD_MAX_BLOCKS);
    const float scale = 1.0f / (float)(n_masked > 0 ? n_masked : 1);     // nll_loss(reduction='mean')
    const bool vec4 = C % 4 == 0 && (uintptr_t)logits % 16 == 0 && (grad_logits == nullptr || (uintptr_t)grad_logits % 16 == 0);
    if (rows && vec4) nb = (int)std::min<int64_t>((N + 31) / 32, HEAD_MAX_BLOCKS);
    if (nb > 0 && rows && vec4 && C <= 32)
        k_head_groups<8, false><<<nb, 256, 0, st>>>(logits, y, row_mask, N, C, scale, grad_logits, (float *)workspace);
    else if (nb > 0 && rows && vec4)
        k_head_groups<16, false><<<nb, 256, 0, st>>>(logits, y, row_mask, N, C, scale, grad_logits, (float *)workspace);
    else if (nb > 0 && rows)
        k_head_rows<false><<<nb, 256, 0, st>>>(logits, y, row_mask, N, C, scale, grad_logits, (float *)workspace);
    else if (nb > 0)
        k_head<<<nb, 256, 0, st>>>(logits, y, row_mask, N, C, scale, grad_logits, (float *)workspace);
    k_head_reduce<<<1, 256, 0, st>>>((const float *)workspace, nb, scale, scale, 2, loss_and_correct);
    SN_HIP(hipGetLastError());
    return SNGNN_OK;
}

extern "C" int64_t sngnn_linear_wgrad_workspace_bytes(int64_t N, int C, int F)
{
    const int64_t chunks = std::max<int64_t>((N + WG_ROWS - 1) / WG_ROWS, wgrad_mfma_partials(N, C, F));
    return chunks * (int64_t)C * (F + 1) * 4 + 256;
}

extern "C" int sngnn_linear_wgrad(const float *grad_out, const float *x, int64_t N, int C, int F,
                                  float *grad_weight, float *grad_bias, void *workspace, void *stream)
{
    SN_REQUIRE(N >= 0 && C >= 1 && F >= 1, SNGNN_EINVAL, "bad shape");
    SN_REQUIRE(grad_out && x && grad_weight && workspace, SNGNN_EINVAL, "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    const int nmfma = wgrad_mfma_partials(N, C, F);
    const bool mfma = nmfma > 0 && (uintptr_t)x % 16 == 0;
    const int chunks = mfma ? nmfma : (int)((N + WG_ROWS - 1) / WG_ROWS);
    float *part = (float *)workspace;
    float *part_b = part + (size_t)chunks * C * F;
    if (mfma) {
        if (int rc = launch_wgrad_mfma(grad_out, x, N, C, F, part, grad_bias ? part_b : nullptr, st)) return rc;
    } else if (chunks > 0) {
        // accumulators per thread: the smallest tile that covers C (or 64 channels per pass)
        const int kacc = C <= 16 ? 8 : C <= 32 ? 16 : C <= 40 ? 20 : C <= 48 ? 24 : 32;
        dim3 grid((F + WG_FT - 1) / WG_FT, (C + 2 * kacc - 1) / (2 * kacc), chunks);
        float *pb = grad_bias ? part_b : nullptr;
        switch (kacc) {
        case 8: k_wgrad_partial<8><<<grid, 256 * WG_SUB, 0, st>>>(grad_out, x, N, C, F, part, pb); break;
        case 16: k_wgrad_partial<16><<<grid, 256 * WG_SUB, 0, st>>>(grad_out, x, N, C, F, part, pb); break;
        case 20: k_wgrad_partial<20><<<grid, 256 * WG_SUB, 0, st>>>(grad_out, x, N, C, F, part, pb); break;
        case 24: k_wgrad_partial<24><<<grid, 256 * WG_SUB, 0, st>>>(grad_out, x, N, C, F, part, pb); break;
        default: k_wgrad_partial<32><<<grid, 256 * WG_SUB, 0, st>>>(grad_out, x, N, C, F, part, pb); break;
        }
    }
    const int64_t len = (int64_t)C * F;
    const int nbA = (int)((len + 15) / 16), nbB = grad_bias ? (C + 15) / 16 : 0;
    k_sum_partials<<<nbA + nbB, 256, 0, st>>>(part, len, grad_weight, nbA, part_b, C, grad_bias, chunks);
    SN_HIP(hipGetLastError());
    return SNGNN_OK;
}
