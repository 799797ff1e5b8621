// Sim-GFA toolbox kernels (gfx950): the cosine statistics of
// SimGFAToolbox/dense.py without Python row/block loops.
//
//   sngnn_cosine_dense       S = n n^T  (dense.py:138-141)  - the one dense contraction
//                            of the repo: fp32 MFMA (v_mfma_f32_32x32x2_f32, exact f32),
//                            128x128 tiles staged through LDS, epilogue scales by the
//                            two inverse norms (x is never normalised in memory).
//   sngnn_cosine_class_sums  block sums of S per class pair (dense.py:9-30, 104-130,
//                            144-149, 167-179) WITHOUT the N x N product:
//                            sum_{i in A, j in B} <n_i, n_j> = <m_A, m_B>,  m_A = sum_{i in A} n_i,
//                            so O(N F) work replaces O(N^2 F); accumulation in f64.
//   sngnn_edge_cosine        per-edge cosine of raw feature rows (dense.py:152-164),
//                            one wave per edge, norms from the same pass.
#include <hipcub/hipcub.hpp>

#include "common.h"

namespace sngnn {

using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ float wave_sum_f(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// inv[i] = 1 / max(||x_i||, eps)   (one wave per row; F.normalize semantics)
__global__ __launch_bounds__(256) void k_row_inv_norm(const float *__restrict__ x, int64_t N,
                                                      int64_t F, float *__restrict__ inv,
                                                      double *__restrict__ diag_sum)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const float *p = x + row * F;
    float ss = 0.f;
    for (int64_t c = lane; c < F; c += 64) ss = fmaf(p[c], p[c], ss);
    ss = wave_sum_f(ss);
    const float r = 1.0f / fmaxf(sqrtf(ss), EPS_NORM);
    if (lane == 0) {
        inv[row] = r;
        if (diag_sum) atomicAdd(diag_sum, (double)ss * (double)r * (double)r);   // <n_i, n_i>
    }
}

// ---------------------------------------------------------------------------
// S = diag(inv) X X^T diag(inv) with fp32 MFMA.  Workgroup = 256 threads = 2x2
// waves, 128x128 output tile, each wave 64x64 = 2x2 MFMA blocks of 32x32.
//   * S is symmetric: only tiles with row block <= column block are computed, each
//     writes its mirror image as well (half the flops of the reference's full mm);
//   * panels [128][32] of the row block and of the column block go through LDS with
//     a padded stride of 33 floats (lane -> (row l & 31, k = l >> 5) reads hit 32
//     different banks); global loads are coalesced 128-byte row segments and the
//     NEXT k-step's panels are already in registers while the current one is
//     multiplied (one LDS buffer, register double buffering).
// ---------------------------------------------------------------------------
constexpr int TB_M = 128, TB_K = 32, TB_LD = TB_K + 1, TB_LOADS = TB_M * TB_K / 256;

// Few tiles (N of a few thousand: 171 upper-triangle tiles at Chameleon's size for 256 CUs):
// the contraction is split over ks workgroups per tile, each writes its raw partial tile
// (both orientations) to part[s] and k_cosine_reduce adds the ks partials in a fixed order
// and applies the inverse norms.
__global__ __launch_bounds__(256) void k_cosine_mfma(const float *__restrict__ x, int64_t N,
                                                     int64_t F, const float *__restrict__ inv,
                                                     float *__restrict__ S, int nb, int ks, int64_t k_per,
                                                     float *__restrict__ part)
{
    __shared__ float sA[TB_M * TB_LD];
    __shared__ float sB[TB_M * TB_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;          // wave's 64x64 quadrant
    // linear workgroup id -> (by <= bx) of the upper triangle, row by row
    const int split = blockIdx.x % ks;
    const int64_t k_begin = (int64_t)split * k_per, k_end = min(F, k_begin + k_per);
    int by = 0, rem = blockIdx.x / ks;
    while (rem >= nb - by) { rem -= nb - by; ++by; }
    const int bx = by + rem;
    const int64_t row0 = (int64_t)by * TB_M, col0 = (int64_t)bx * TB_M;
    const bool diag = by == bx;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // staging: thread t owns column t & 31 of rows (t >> 5) + 8 u, u = 0..15
    const int sc = tid & 31, sr = tid >> 5;
    float ra[TB_LOADS], rb[TB_LOADS];
    auto fetch = [&](int64_t k0) {
        const int64_t k = k0 + sc;
#pragma unroll
        for (int u = 0; u < TB_LOADS; ++u) {
            const int64_t r_a = row0 + sr + 8 * u, r_b = col0 + sr + 8 * u;
            ra[u] = (r_a < N && k < k_end) ? x[r_a * F + k] : 0.f;
            rb[u] = diag ? ra[u] : ((r_b < N && k < k_end) ? x[r_b * F + k] : 0.f);
        }
    };
    fetch(k_begin);
    for (int64_t k0 = k_begin; k0 < k_end; k0 += TB_K) {
        __syncthreads();                                  // previous step's LDS reads are done
#pragma unroll
        for (int u = 0; u < TB_LOADS; ++u) {
            sA[(sr + 8 * u) * TB_LD + sc] = ra[u];
            sB[(sr + 8 * u) * TB_LD + sc] = rb[u];
        }
        __syncthreads();
        if (k0 + TB_K < k_end) fetch(k0 + TB_K);          // in flight during the MFMAs below
#pragma unroll
        for (int kk = 0; kk < TB_K; kk += 2) {
            float a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                // 32x32x2: lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]
                a[t] = sA[(wr * 64 + t * 32 + (lane & 31)) * TB_LD + kk + (lane >> 5)];
                b[t] = sB[(wc * 64 + t * 32 + (lane & 31)) * TB_LD + kk + (lane >> 5)];
            }
#pragma unroll
            for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                for (int tb = 0; tb < 2; ++tb)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
        }
    }
    // epilogue: C/D layout col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            const int64_t c = col0 + wc * 64 + tb * 32 + (lane & 31);
            const float ic = c < N ? inv[c] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t rr = row0 + wr * 64 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (rr < N && c < N) {
                    if (ks == 1) {
                        const float v = acc[ta][tb][r] * (inv[rr] * ic);
                        S[rr * N + c] = v;
                        if (!diag) S[c * N + rr] = v;     // mirror image of an off-diagonal tile
                    } else {
                        float *P = part + (size_t)split * N * N;
                        P[rr * N + c] = acc[ta][tb][r];
                        if (!diag) P[c * N + rr] = acc[ta][tb][r];
                    }
                }
            }
        }
}

// S = (sum of the ks partial products, in split order) * inv_r * inv_c; workgroup = one row segment
__global__ __launch_bounds__(256) void k_cosine_reduce(const float *__restrict__ part, int ks, int64_t N,
                                                       const float *__restrict__ inv, float *__restrict__ S)
{
    const int64_t r = blockIdx.y, total = N * N;
    const float ir = inv[r];
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < N; c += (int64_t)gridDim.x * 256) {
        const int64_t e = r * N + c;
        float v = part[e];
        for (int sp = 1; sp < ks; ++sp) v += part[(size_t)sp * total + e];
        S[e] = v * (ir * inv[c]);
    }
}

// ---------------------------------------------------------------------------
// Class sums: rows visited in class order (order[] = stable argsort of y); thread
// = one feature column, flushes its running f64 sum whenever the class changes.
// ---------------------------------------------------------------------------
constexpr int CS_ROWS = 256;      // rows per workgroup (in class order)

__global__ __launch_bounds__(256) void k_class_row_sums(const float *__restrict__ x, int64_t N,
                                                        int64_t F, const float *__restrict__ inv,
                                                        const int32_t *__restrict__ order,
                                                        const int32_t *__restrict__ y_sorted,
                                                        double *__restrict__ M)
{
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t p0 = (int64_t)blockIdx.y * CS_ROWS, p1 = min(N, p0 + CS_ROWS);
    if (f >= F || p0 >= N) return;
    int cls = y_sorted[p0];
    double run = 0.0;
    for (int64_t p = p0; p < p1; ++p) {
        const int c = y_sorted[p];
        if (c != cls) {
            atomicAdd(&M[(int64_t)cls * F + f], run);
            run = 0.0;
            cls = c;
        }
        const int64_t i = order[p];
        run += (double)(x[i * F + f] * inv[i]);
    }
    atomicAdd(&M[(int64_t)cls * F + f], run);
}

// class_sum[a][b] = <M_a, M_b>   (one workgroup per pair)
__global__ __launch_bounds__(256) void k_class_gram(const double *__restrict__ M, int n_classes,
                                                    int64_t F, double *__restrict__ out)
{
    __shared__ double part[4];
    const int a = blockIdx.x / n_classes, b = blockIdx.x % n_classes;
    double s = 0.0;
    for (int64_t f = threadIdx.x; f < F; f += 256) s += M[(int64_t)a * F + f] * M[(int64_t)b * F + f];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] += part[0] + part[1] + part[2] + part[3];
}

// ---------------------------------------------------------------------------
// Per-edge cosine of raw features: one wave per edge, three running sums.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_edge_cosine(const float *__restrict__ x, int64_t N,
                                                     int64_t F, const int64_t *__restrict__ ei,
                                                     int64_t E, float *__restrict__ sim,
                                                     int *__restrict__ bad)
{
    const int lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= E) return;
    const int64_t a = ei[e], b = ei[E + e];
    if (a < 0 || a >= N || b < 0 || b >= N) { if (lane == 0) atomicOr(bad, 1); return; }
    const float *pa = x + a * F, *pb = x + b * F;
    float d = 0.f, qa = 0.f, qb = 0.f;
    if ((F & 3) == 0 && ((uintptr_t)x & 15) == 0) {
        for (int64_t c = lane * 4; c < F; c += 256) {
            const float4 u = *reinterpret_cast<const float4 *>(pa + c);
            const float4 v = *reinterpret_cast<const float4 *>(pb + c);
            d = fmaf(u.x, v.x, fmaf(u.y, v.y, fmaf(u.z, v.z, fmaf(u.w, v.w, d))));
            qa = fmaf(u.x, u.x, fmaf(u.y, u.y, fmaf(u.z, u.z, fmaf(u.w, u.w, qa))));
            qb = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, qb))));
        }
    } else {
        for (int64_t c = lane; c < F; c += 64) {
            const float u = pa[c], v = pb[c];
            d = fmaf(u, v, d); qa = fmaf(u, u, qa); qb = fmaf(v, v, qb);
        }
    }
    d = wave_sum_f(d); qa = wave_sum_f(qa); qb = wave_sum_f(qb);
    if (lane == 0)
        sim[e] = d * ((1.0f / fmaxf(sqrtf(qa), EPS_NORM)) * (1.0f / fmaxf(sqrtf(qb), EPS_NORM)));
}

struct AsyncBuf {
    void *p = nullptr;
    hipStream_t st;
    explicit AsyncBuf(hipStream_t s) : st(s) {}
    ~AsyncBuf() { if (p) (void)hipFreeAsync(p, st); }
    int alloc(size_t bytes)
    {
        return hipMallocAsync(&p, bytes ? bytes : 4, st) == hipSuccess ? 0 : SNGNN_ENOMEM;
    }
    template <class T> T *as() { return (T *)p; }
};

__global__ void k_iota32(int32_t *a, int64_t n)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) a[t] = (int32_t)t;
}

__global__ void k_check_labels(const int32_t *y, int64_t n, int n_classes, int *bad)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n && (y[t] < 0 || y[t] >= n_classes)) atomicOr(bad, 1);
}

}  // namespace sngnn

using namespace sngnn;

extern "C" int sngnn_cosine_dense(const float *x, int64_t N, int64_t F, float *S, void *stream)
{
    SN_REQUIRE(N >= 0 && F >= 1, SNGNN_EINVAL, "bad shape");
    if (N == 0) return SNGNN_OK;
    SN_REQUIRE(x && S, SNGNN_EINVAL, "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    AsyncBuf inv(st);
    SN_REQUIRE(inv.alloc((size_t)N * 4) == 0, SNGNN_ENOMEM, "out of device memory");
    k_row_inv_norm<<<(unsigned)((N + 3) / 4), 256, 0, st>>>(x, N, F, inv.as<float>(), nullptr);
    const int nb = (int)((N + TB_M - 1) / TB_M);
    const int tiles = nb * (nb + 1) / 2;
    // tiles for at most three quarters of the CUs: split the contraction (at least two K-steps per
    // split).  Measured: Chameleon's 171 tiles 570 -> 480 us with 3 splits; at Cora's 253 tiles (one
    // per CU already) splitting only added the reduction pass (355 -> 391 us), hence the bound.
    int ks = 1;
    if (tiles <= 192) ks = (int)std::min<int64_t>(std::min<int64_t>(8, (512 + tiles - 1) / tiles), std::max<int64_t>(1, F / (2 * TB_K)));
    const int64_t k_per = ((F + ks - 1) / ks + TB_K - 1) / TB_K * TB_K;
    ks = (int)((F + k_per - 1) / k_per);
    AsyncBuf part(st);
    if (ks > 1) SN_REQUIRE(part.alloc((size_t)ks * N * N * 4) == 0, SNGNN_ENOMEM, "out of device memory");
    k_cosine_mfma<<<tiles * ks, 256, 0, st>>>(x, N, F, inv.as<float>(), S, nb, ks, k_per, part.as<float>());
    if (ks > 1) {
        SN_REQUIRE(N <= 65535, SNGNN_EINVAL, "internal: split contraction is for small N only");
        k_cosine_reduce<<<dim3((unsigned)std::min<int64_t>((N + 255) / 256, 8), (unsigned)N), 256, 0, st>>>(
            part.as<float>(), ks, N, inv.as<float>(), S);
    }
    SN_HIP(hipGetLastError());
    return SNGNN_OK;
}

extern "C" int sngnn_cosine_class_sums(const float *x, int64_t N, int64_t F, const int32_t *y,
                                       int n_classes, double *class_sum, double *diag_sum,
                                       void *stream)
{
    SN_REQUIRE(N >= 0 && F >= 1 && n_classes >= 1, SNGNN_EINVAL, "bad shape");
    SN_REQUIRE(N < ((int64_t)1 << 31), SNGNN_EINVAL, "too many rows");
    if (N == 0) return SNGNN_OK;
    SN_REQUIRE(x && y && class_sum, SNGNN_EINVAL, "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    AsyncBuf inv(st), iota(st), order(st), ys(st), M(st), bad(st), tmp(st);
    SN_REQUIRE(!inv.alloc((size_t)N * 4) && !iota.alloc((size_t)N * 4) && !order.alloc((size_t)N * 4) &&
                   !ys.alloc((size_t)N * 4) && !M.alloc((size_t)n_classes * F * 8) && !bad.alloc(4),
               SNGNN_ENOMEM, "out of device memory");
    SN_HIP(hipMemsetAsync(bad.p, 0, 4, st));
    SN_HIP(hipMemsetAsync(M.p, 0, (size_t)n_classes * F * 8, st));
    const unsigned gn = (unsigned)((N + 255) / 256);
    k_check_labels<<<gn, 256, 0, st>>>(y, N, n_classes, bad.as<int>());
    int h_bad = 0;
    SN_HIP(hipMemcpyAsync(&h_bad, bad.p, 4, hipMemcpyDeviceToHost, st));
    SN_HIP(hipStreamSynchronize(st));
    SN_REQUIRE(!h_bad, SNGNN_ERANGE, "label outside [0, n_classes)");
    k_row_inv_norm<<<(unsigned)((N + 3) / 4), 256, 0, st>>>(x, N, F, inv.as<float>(), diag_sum);
    k_iota32<<<gn, 256, 0, st>>>(iota.as<int32_t>(), N);
    size_t tb = 0;
    SN_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, y, ys.as<int32_t>(), iota.as<int32_t>(),
                                              order.as<int32_t>(), (int)N, 0, 32, st));
    SN_REQUIRE(tmp.alloc(tb) == 0, SNGNN_ENOMEM, "out of device memory");
    SN_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, y, ys.as<int32_t>(), iota.as<int32_t>(),
                                              order.as<int32_t>(), (int)N, 0, 32, st));
    dim3 grid((unsigned)((F + 255) / 256), (unsigned)((N + CS_ROWS - 1) / CS_ROWS));
    k_class_row_sums<<<grid, 256, 0, st>>>(x, N, F, inv.as<float>(), order.as<int32_t>(),
                                           ys.as<int32_t>(), M.as<double>());
    k_class_gram<<<n_classes * n_classes, 256, 0, st>>>(M.as<double>(), n_classes, F, class_sum);
    SN_HIP(hipGetLastError());
    return SNGNN_OK;
}

// ---------------------------------------------------------------------------
// Sparse columns (sparse.py:8-14): entry (a, b) of M_n^T M_n for a list of column pairs,
// M_n in CSC with ascending row ids inside a column.  One wave per pair: the lanes stride
// over the shorter column, each entry binary-searches its row id in the longer one; the
// products are summed in lane order (fixed: deterministic).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sparse_pair_dot(const int64_t *__restrict__ colptr,
                                                         const int32_t *__restrict__ rowidx,
                                                         const float *__restrict__ vals,
                                                         const int64_t *__restrict__ pa,
                                                         const int64_t *__restrict__ pb, int64_t n_pairs,
                                                         int64_t n_cols, float *__restrict__ out, int *bad)
{
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= n_pairs) return;
    int64_t a = pa[p], b = pb[p];
    if (a < 0 || a >= n_cols || b < 0 || b >= n_cols) {
        if (lane == 0) { atomicOr(bad, 1); out[p] = 0.f; }
        return;
    }
    int64_t a0 = colptr[a], a1 = colptr[a + 1], b0 = colptr[b], b1 = colptr[b + 1];
    if (a1 - a0 > b1 - b0) { int64_t t = a0; a0 = b0; b0 = t; t = a1; a1 = b1; b1 = t; }   // a = shorter
    float acc = 0.f;
    for (int64_t q = a0 + lane; q < a1; q += 64) {
        const int32_t r = rowidx[q];
        int64_t lo = b0, hi = b1;
        while (lo < hi) {
            const int64_t m = (lo + hi) >> 1;
            if (rowidx[m] < r) lo = m + 1; else hi = m;
        }
        if (lo < b1 && rowidx[lo] == r) acc = fmaf(vals[q], vals[lo], acc);
    }
    acc = wave_sum_f(acc);
    if (lane == 0) out[p] = acc;
}

extern "C" int sngnn_sparse_pair_dot(const int64_t *colptr, const int32_t *rowidx, const float *vals,
                                     int64_t n_cols, const int64_t *pair_a, const int64_t *pair_b,
                                     int64_t n_pairs, float *out, void *stream)
{
    SN_REQUIRE(n_cols >= 0 && n_pairs >= 0, SNGNN_EINVAL, "bad shape");
    if (n_pairs == 0) return SNGNN_OK;
    SN_REQUIRE(colptr && pair_a && pair_b && out, SNGNN_EINVAL, "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    AsyncBuf bad(st);
    SN_REQUIRE(bad.alloc(4) == 0, SNGNN_ENOMEM, "out of device memory");
    SN_HIP(hipMemsetAsync(bad.p, 0, 4, st));
    k_sparse_pair_dot<<<(unsigned)((n_pairs + 3) / 4), 256, 0, st>>>(colptr, rowidx, vals, pair_a, pair_b, n_pairs,
                                                                      n_cols, out, bad.as<int>());
    int h_bad = 0;
    SN_HIP(hipMemcpyAsync(&h_bad, bad.p, 4, hipMemcpyDeviceToHost, st));
    SN_HIP(hipStreamSynchronize(st));
    SN_REQUIRE(!h_bad, SNGNN_ERANGE, "a pair names a column outside [0, n_cols)");
    return SNGNN_OK;
}

extern "C" int sngnn_edge_cosine(const float *x, int64_t N, int64_t F, const int64_t *edge_index_dev,
                                 int64_t E, float *sim, void *stream)
{
    SN_REQUIRE(N >= 0 && F >= 1 && E >= 0, SNGNN_EINVAL, "bad shape");
    if (E == 0) return SNGNN_OK;
    SN_REQUIRE(x && edge_index_dev && sim, SNGNN_EINVAL, "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    AsyncBuf bad(st);
    SN_REQUIRE(bad.alloc(4) == 0, SNGNN_ENOMEM, "out of device memory");
    SN_HIP(hipMemsetAsync(bad.p, 0, 4, st));
    k_edge_cosine<<<(unsigned)((E + 3) / 4), 256, 0, st>>>(x, N, F, edge_index_dev, E, sim, bad.as<int>());
    int h_bad = 0;
    SN_HIP(hipMemcpyAsync(&h_bad, bad.p, 4, hipMemcpyDeviceToHost, st));
    SN_HIP(hipStreamSynchronize(st));
    SN_REQUIRE(!h_bad, SNGNN_ERANGE, "edge_index contains a node id outside [0, N)");
    return SNGNN_OK;
}
