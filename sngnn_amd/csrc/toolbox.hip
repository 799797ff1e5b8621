// Sim-GFA toolbox kernels (gfx950): the cosine statistics of
// SimGFAToolbox/dense.py without Python row/block loops.
//
//   sngnn_cosine_dense       S = n n^T  (dense.py:138-141)  - the one dense contraction
//                            of the repo: matrix cores at fp32 rounding (exact bf16 split of both
//                            panels, or v_mfma_f32_32x32x2_f32),
//                            128x128 tiles staged through LDS, epilogue scales by the
//                            two inverse norms (x is never normalised in memory).
//   sngnn_cosine_class_sums  block sums of S per class pair (dense.py:9-30, 104-130,
//                            144-149, 167-179) WITHOUT the N x N product:
//                            sum_{i in A, j in B} <n_i, n_j> = <m_A, m_B>,  m_A = sum_{i in A} n_i,
//                            so O(N F) work replaces O(N^2 F); accumulation in f64.
//   sngnn_edge_cosine        per-edge cosine of raw feature rows (dense.py:152-164),
//                            one wave per edge, norms from the same pass.
#include <hipcub/hipcub.hpp>

#include "device_utils.h"

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
                                                      double *__restrict__ diag)
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
        if (diag) diag[row] = (double)ss * (double)r * (double)r;                // <n_i, n_i>
    }
}

// out[0] += sum of v[0..n) in a FIXED order (one workgroup: thread t adds v[t], v[t + 1024], ...
// in sequence, then a tree over the threads) - the same bits on every run
__global__ __launch_bounds__(1024) void k_sum_fixed_d(const double *__restrict__ v, int64_t n,
                                                      double *__restrict__ out)
{
    __shared__ double sh[1024];
    double a = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) a += v[i];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int w = 512; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] += sh[0];
}

// ---------------------------------------------------------------------------
// S = diag(inv) X X^T diag(inv) on the matrix cores (fp32 rounding).  Workgroup = 256 threads = 2x2
// waves, 128x128 output tile, each wave 64x64 = 2x2 MFMA blocks of 32x32.
//   * S is symmetric: only tiles with row block <= column block are computed, each
//     writes its mirror image as well (half the flops of the reference's full mm); the
//     mirror goes through an LDS transpose so that both images leave as 128-byte rows.
//   * Panels [128][32] of the row block and of the column block go through LDS with a
//     stride of 36 floats.  The k order of a contraction is free: MFMA j of a K-step takes
//     k = j from the lanes' lower half and k = 16 + j from the upper half, so a lane's 16
//     operands of a step are 16 CONSECUTIVE floats of its row - four ds_read_b128 per block
//     row instead of sixteen ds_read_b32, bank-conflict-free at that stride.
//   * x is read as 16-byte vectors from rows of `ld` floats (ld % 4 == 0, zero-padded: the
//     launcher makes a padded copy when F is not a multiple of 4), unconditionally (clamped
//     row, value zeroed at staging time), one K-step ahead: the NEXT step's panels travel
//     while the current one is multiplied.
//   * The prefetch registers are NAMED ext-vector variables in straight-line code.  Round 2
//     found the first version (float4 structs filled by a lambda over a small array, a bounds
//     select on the loaded value) keeping them in stack slots: every load was followed by a
//     scratch store - i.e. waited for on the spot, in front of the MFMAs it was meant to
//     travel behind.  tools/micro/cosine_mfma_bench.hip: 1.40 -> 0.63 ms at Actor's size with
//     nothing else changed; the MFMAs alone take 0.44 ms.
// ---------------------------------------------------------------------------
constexpr int TB_M = 128, TB_K = 32, TB_LD = TB_K + 4;
typedef float f4 __attribute__((ext_vector_type(4)));

// Few tiles (N of a few thousand: 171 upper-triangle tiles at Chameleon's size for 256 CUs):
// the contraction is split over ks workgroups per tile, each writes its raw partial tile
// (both orientations) to part[s] and k_cosine_reduce adds the ks partials in a fixed order
// and applies the inverse norms.
// BF3 (default): the products on the BF16 matrix cores.  fp32 MFMAs run on the vector ALU's pipes
// (a 32x32x2 takes 64 cycles); a float is the exact sum of three bf16 values and a product of two
// bf16 values is exact in fp32, so every panel is split ONCE, when it is staged into LDS (three
// bf16 planes, rows of 64 + 16 bytes: conflict-free 16-byte reads), and a K-step of 32 is
// 2 x 8 `v_mfma_f32_32x32x16_bf16` per output block - the eight partial products x_i y_j with
// i + j <= 5, smallest first, fp32 accumulation - instead of 16 fp32 MFMAs: half the matrix
// cycles at an fp32 dot product's rounding (linear.hip has the argument and the measurements).
constexpr int TB_PS = 80;                     // bytes per row of a bf16 plane: 32 k-slots + 16 bytes of padding
using bf16x8 = sn_bf16x8;
using u32x4t = sn_u32x4;
using u32x2t = sn_u32x2;

template <bool BF3>
__global__ __launch_bounds__(256) void k_cosine_mfma(const float *__restrict__ x, int64_t N,
                                                     int64_t ld, const float *__restrict__ inv,
                                                     float *__restrict__ S, int nb, int ks, int64_t k_per,
                                                     float *__restrict__ part)
{
    constexpr int PANEL_BYTES = BF3 ? 3 * TB_M * TB_PS : TB_M * TB_LD * 4;
    __shared__ __align__(16) unsigned char smemA[PANEL_BYTES];
    __shared__ __align__(16) unsigned char smemB[PANEL_BYTES];
    float *sA = reinterpret_cast<float *>(smemA), *sB = reinterpret_cast<float *>(smemB);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;          // wave's 64x64 quadrant
    // linear workgroup id -> (by <= bx) of the upper triangle, row by row
    const int split = blockIdx.x % ks;
    const int64_t k_begin = (int64_t)split * k_per, k_end = min(ld, k_begin + k_per);
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

    // staging: thread t owns the 16-byte vector (t & 7) of rows (t >> 3) + 32 u, u = 0..3
    // (16 consecutive lanes stage rows r and r + 4, not r and r + 1: with the bf16 planes' 80-byte rows their 8-byte
    // stores then cover all 32 banks once - adjacent rows overlapped in four banks, 2.4 conflict cycles per LDS
    // instruction in round 4's counters, all from these stores; the fp32 panels' 16-byte stores do not care)
    const int kq = tid & 7, slot_ = tid >> 3, r0 = (slot_ & ~7) | ((slot_ & 7) >> 1) | ((slot_ & 1) << 2);
    f4 va0, va1, va2, va3, vb0, vb1, vb2, vb3;
    const float *ga0 = x + min(row0 + r0, N - 1) * ld + 4 * kq, *ga1 = x + min(row0 + r0 + 32, N - 1) * ld + 4 * kq;
    const float *ga2 = x + min(row0 + r0 + 64, N - 1) * ld + 4 * kq, *ga3 = x + min(row0 + r0 + 96, N - 1) * ld + 4 * kq;
    const float *gb0 = x + min(col0 + r0, N - 1) * ld + 4 * kq, *gb1 = x + min(col0 + r0 + 32, N - 1) * ld + 4 * kq;
    const float *gb2 = x + min(col0 + r0 + 64, N - 1) * ld + 4 * kq, *gb3 = x + min(col0 + r0 + 96, N - 1) * ld + 4 * kq;
    const bool oa0 = row0 + r0 < N, oa1 = row0 + r0 + 32 < N, oa2 = row0 + r0 + 64 < N, oa3 = row0 + r0 + 96 < N;
    const bool ob0 = col0 + r0 < N, ob1 = col0 + r0 + 32 < N, ob2 = col0 + r0 + 64 < N, ob3 = col0 + r0 + 96 < N;
    float *wa = sA + r0 * TB_LD + 4 * kq, *wb = sB + r0 * TB_LD + 4 * kq;
    const f4 z = {0.f, 0.f, 0.f, 0.f};
    // (a 16-byte vector of a row is entirely below k_end or entirely beyond: ld, k_per % 4 == 0)
#define SN_FETCH(K0)                                                                         \
    {                                                                                        \
        /* a vector beyond the split's range is never used (zeroed at staging): read the  */ \
        /* row's [k_begin, k_begin + 4) instead - always inside the row, whatever kq      */ \
        const int64_t kk_ = (4 * kq + (K0) < k_end) ? (K0) : k_begin - 4 * kq;               \
        va0 = *(const f4 *)(ga0 + kk_); va1 = *(const f4 *)(ga1 + kk_);                      \
        va2 = *(const f4 *)(ga2 + kk_); va3 = *(const f4 *)(ga3 + kk_);                      \
        if (!diag) {                                                                         \
            vb0 = *(const f4 *)(gb0 + kk_); vb1 = *(const f4 *)(gb1 + kk_);                  \
            vb2 = *(const f4 *)(gb2 + kk_); vb3 = *(const f4 *)(gb3 + kk_);                  \
        }                                                                                    \
    }
#define SN_STAGE(K0)                                                                         \
    {                                                                                        \
        const bool kin_ = 4 * kq + (K0) < k_end;                                             \
        const f4 s0 = (oa0 && kin_) ? va0 : z, s1 = (oa1 && kin_) ? va1 : z;                 \
        const f4 s2 = (oa2 && kin_) ? va2 : z, s3 = (oa3 && kin_) ? va3 : z;                 \
        *(f4 *)(wa) = s0;              *(f4 *)(wa + 32 * TB_LD) = s1;                        \
        *(f4 *)(wa + 64 * TB_LD) = s2; *(f4 *)(wa + 96 * TB_LD) = s3;                        \
        if (diag) {                                                                          \
            *(f4 *)(wb) = s0;              *(f4 *)(wb + 32 * TB_LD) = s1;                    \
            *(f4 *)(wb + 64 * TB_LD) = s2; *(f4 *)(wb + 96 * TB_LD) = s3;                    \
        } else {                                                                             \
            *(f4 *)(wb) = (ob0 && kin_) ? vb0 : z;              *(f4 *)(wb + 32 * TB_LD) = (ob1 && kin_) ? vb1 : z; \
            *(f4 *)(wb + 64 * TB_LD) = (ob2 && kin_) ? vb2 : z; *(f4 *)(wb + 96 * TB_LD) = (ob3 && kin_) ? vb3 : z; \
        }                                                                                    \
    }
    // BF3 staging: the same vectors, split, 8 bytes per plane at [plane][row][4 kq ..]
#define SN_STAGE3(K0)                                                                        \
    {                                                                                        \
        const bool kin_ = 4 * kq + (K0) < k_end;                                             \
        auto put = [&](unsigned char *base, int row, const f4 &v) {                          \
            u32x2t p1, p2, p3;                                                               \
            const float vv_[4] = {v[0], v[1], v[2], v[3]};                                   \
            split_bf16x4(vv_, p1, p2, p3);                                                   \
            unsigned char *d_ = base + row * TB_PS + 8 * kq;                                 \
            *(u32x2t *)(d_) = p1;                                                            \
            *(u32x2t *)(d_ + TB_M * TB_PS) = p2;                                             \
            *(u32x2t *)(d_ + 2 * TB_M * TB_PS) = p3;                                         \
        };                                                                                   \
        const f4 s0 = (oa0 && kin_) ? va0 : z, s1 = (oa1 && kin_) ? va1 : z;                 \
        const f4 s2 = (oa2 && kin_) ? va2 : z, s3 = (oa3 && kin_) ? va3 : z;                 \
        put(smemA, r0, s0); put(smemA, r0 + 32, s1); put(smemA, r0 + 64, s2); put(smemA, r0 + 96, s3); \
        if (diag) {                                                                          \
            put(smemB, r0, s0); put(smemB, r0 + 32, s1); put(smemB, r0 + 64, s2); put(smemB, r0 + 96, s3); \
        } else {                                                                             \
            put(smemB, r0, (ob0 && kin_) ? vb0 : z); put(smemB, r0 + 32, (ob1 && kin_) ? vb1 : z);     \
            put(smemB, r0 + 64, (ob2 && kin_) ? vb2 : z); put(smemB, r0 + 96, (ob3 && kin_) ? vb3 : z); \
        }                                                                                    \
    }
    vb0 = vb1 = vb2 = vb3 = z;
    SN_FETCH(k_begin)
    const int li = lane & 31, lh = lane >> 5;
    const float *pa0 = sA + (wr * 64 + li) * TB_LD + 16 * lh, *pa1 = pa0 + 32 * TB_LD;
    const float *pb0 = sB + (wc * 64 + li) * TB_LD + 16 * lh, *pb1 = pb0 + 32 * TB_LD;
    // BF3: lane (li, lh) of k-group g reads the 8 k-slots 16 g + 8 lh .. of its row from each plane
    const unsigned char *qa = smemA + (wr * 64 + li) * TB_PS + 16 * lh;
    const unsigned char *qb = smemB + (wc * 64 + li) * TB_PS + 16 * lh;
    for (int64_t k0 = k_begin; k0 < k_end; k0 += TB_K) {
        __syncthreads();                                  // previous step's LDS reads are done
        if constexpr (BF3) SN_STAGE3(k0) else SN_STAGE(k0)
        __syncthreads();
        if (k0 + TB_K < k_end) SN_FETCH(k0 + TB_K)        // in flight during the MFMAs below
        if constexpr (BF3) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                u32x4t A[2][3], B[2][3];
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        A[blk][pl] = *(const u32x4t *)(qa + (pl * TB_M + blk * 32) * TB_PS + 32 * g);
                        B[blk][pl] = *(const u32x4t *)(qb + (pl * TB_M + blk * 32) * TB_PS + 32 * g);
                    }
#define SN_M3(a_, b_, PA, PB)                                                                           \
                acc[a_][b_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[a_][PA]), \
                                                                      __builtin_bit_cast(bf16x8, B[b_][PB]), acc[a_][b_], 0, 0, 0);
#pragma unroll
                for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
                    for (int b_ = 0; b_ < 2; ++b_) {              // smallest partial products first
                        SN_M3(a_, b_, 2, 1) SN_M3(a_, b_, 1, 2) SN_M3(a_, b_, 2, 0) SN_M3(a_, b_, 1, 1)
                        SN_M3(a_, b_, 0, 2) SN_M3(a_, b_, 1, 0) SN_M3(a_, b_, 0, 1) SN_M3(a_, b_, 0, 0)
                    }
#undef SN_M3
            }
        } else
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f4 ca0 = *(const f4 *)(pa0 + 4 * q), ca1 = *(const f4 *)(pa1 + 4 * q);
            const f4 cb0 = *(const f4 *)(pb0 + 4 * q), cb1 = *(const f4 *)(pb1 + 4 * q);
#define SN_MFMA4(E)                                                                             \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca0.E, cb0.E, acc[0][0], 0, 0, 0);    \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca0.E, cb1.E, acc[0][1], 0, 0, 0);    \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca1.E, cb0.E, acc[1][0], 0, 0, 0);    \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca1.E, cb1.E, acc[1][1], 0, 0, 0);
            SN_MFMA4(x) SN_MFMA4(y) SN_MFMA4(z) SN_MFMA4(w)
#undef SN_MFMA4
        }
    }
#undef SN_FETCH
#undef SN_STAGE
#undef SN_STAGE3
    // epilogue: C/D layout col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
    // The tile itself leaves with the lanes along a row of S; its mirror image is transposed
    // through LDS (the panels are dead) so that it leaves the same way.
    __syncthreads();
    float *T = sA + wave * (32 * 33);                     // [32][33] per wave (4 * 1056 <= 128 * 36)
    float *dst = ks == 1 ? S : part + (size_t)split * N * N;
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            const int64_t cb = col0 + wc * 64 + tb * 32, rb0 = row0 + wr * 64 + ta * 32;
            const int64_t c = cb + li;
            const float ic = (ks == 1 && c < N) ? inv[c] : 1.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int64_t rr = rb0 + rl;
                float v = acc[ta][tb][r];
                if (ks == 1) v *= (rr < N ? inv[rr] : 0.f) * ic;
                if (rr < N && c < N) dst[rr * N + c] = v;
                if (!diag) T[li * 33 + rl] = v;           // transposed: T[column][row]
            }
            if (!diag) {                                  // (workgroup-uniform)
                __syncthreads();
#pragma unroll
                for (int cc = 0; cc < 32; cc += 2) {
                    const int64_t mc = cb + cc + lh, mr = rb0 + li;     // element (mr, mc) of S -> S[mc][mr]
                    if (mc < N && mr < N) dst[mc * N + mr] = T[(cc + lh) * 33 + li];
                }
                __syncthreads();
            }
        }
}

// rows of F floats -> rows of ld floats (ld % 4 == 0), zero padding behind column F
__global__ void k_pad_rows(const float *__restrict__ x, int64_t N, int64_t F, int64_t ld, float *__restrict__ xp)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * ld) return;
    const int64_t r = t / ld, c = t % ld;
    xp[t] = c < F ? x[r * F + c] : 0.f;
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
// = one feature column of one segment of CS_ROWS sorted positions, one running f64 sum per
// class run.  No atomics (the sums are order-dependent in f64 too): a run whose class begins
// AND ends inside the segment is that class's whole sum and is stored to M directly; the
// segment's first / last run may continue in a neighbouring segment and goes to
// P[segment][0 / 1][f]; k_class_join adds a class's pieces in ascending segment order.
// ---------------------------------------------------------------------------
constexpr int CS_ROWS = 256;      // rows per workgroup (in class order)

__global__ __launch_bounds__(256) void k_class_row_sums(const float *__restrict__ x, int64_t N,
                                                        int64_t F, const float *__restrict__ inv,
                                                        const int32_t *__restrict__ order,
                                                        const int32_t *__restrict__ y_sorted,
                                                        double *__restrict__ M, double *__restrict__ P)
{
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t seg = blockIdx.y;
    const int64_t p0 = seg * CS_ROWS, p1 = min(N, p0 + CS_ROWS);
    if (f >= F || p0 >= N) return;
    int cls = y_sorted[p0];
    int64_t a = p0;                                   // start of the current run
    double run = 0.0;
    double *Ps = P + (size_t)seg * 2 * F;
    Ps[f] = 0.0;
    Ps[F + f] = 0.0;
    auto flush = [&](int64_t b) {                     // run of class cls over [a, b)
        const bool starts = a > p0 || p0 == 0 || y_sorted[p0 - 1] != cls;
        const bool ends = b < p1 || p1 == N || y_sorted[p1] != cls;
        if (starts && ends) M[(int64_t)cls * F + f] = run;
        else Ps[(a == p0 ? 0 : F) + f] = run;
    };
    for (int64_t p = p0; p < p1; ++p) {
        const int c = y_sorted[p];
        if (c != cls) {
            flush(p);
            run = 0.0;
            cls = c;
            a = p;
        }
        const int64_t i = order[p];
        run += (double)(x[i * F + f] * inv[i]);
    }
    flush(p1);
}

// classes that span several segments: M[c][f] = their pieces in ascending segment order
__global__ __launch_bounds__(256) void k_class_join(const int32_t *__restrict__ y_sorted, int64_t N, int64_t F,
                                                    const double *__restrict__ P, double *__restrict__ M)
{
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    if (f >= F) return;
    int64_t lo = 0, hi = N;                           // first position with y_sorted >= c
    while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (y_sorted[m] < c) lo = m + 1; else hi = m; }
    const int64_t c_lo = lo;
    hi = N;                                           // first position with y_sorted > c
    while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (y_sorted[m] <= c) lo = m + 1; else hi = m; }
    const int64_t c_hi = lo;
    if (c_hi <= c_lo) return;                         // empty class: M stays 0
    const int64_t s_lo = c_lo / CS_ROWS, s_hi = (c_hi - 1) / CS_ROWS;
    if (s_lo == s_hi) return;                         // inside one segment: stored by k_class_row_sums
    double acc = 0.0;
    for (int64_t sg = s_lo; sg <= s_hi; ++sg) {
        // in its first segment the class is the last run unless it starts at the segment's
        // first position; in every later segment it is the first run
        const int slot = (sg == s_lo && c_lo > sg * CS_ROWS) ? 1 : 0;
        acc += P[((size_t)sg * 2 + slot) * F + f];
    }
    M[(int64_t)c * F + f] = acc;
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

// ---------------------------------------------------------------------------
// Mean of per-edge values grouped by an index, in the reference's arithmetic: torch_scatter's
// scatter_mean (dense.py:163) adds the values of a group in EDGE ORDER in
// fp32, counts every entry, clamps the count to 1 and divides.  The entries are brought into
// (index, edge position) order by a stable radix sort; one thread then walks one group's entries
// serially: the same additions in the same order, no atomics - bit for bit the CPU result, and
// the same bits on every run.
// ---------------------------------------------------------------------------
__global__ void k_seg_keys(const int64_t *__restrict__ index, int64_t E, int64_t M, int32_t *__restrict__ key,
                           int32_t *__restrict__ pos, int *__restrict__ bad)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= E) return;
    const int64_t i = index[t];
    if (i < 0 || i >= M) { atomicOr(bad, 1); key[t] = 0; pos[t] = (int32_t)t; return; }
    key[t] = (int32_t)i;
    pos[t] = (int32_t)t;
}

__global__ void k_seg_bounds(const int32_t *__restrict__ key_s, int64_t E, int32_t *__restrict__ first,
                             int32_t *__restrict__ last)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= E) return;
    const int32_t k = key_s[t];
    if (t == 0 || key_s[t - 1] != k) first[k] = (int32_t)t;
    if (t == E - 1 || key_s[t + 1] != k) last[k] = (int32_t)t + 1;
}

__global__ void k_seg_mean(const float *__restrict__ val, const int32_t *__restrict__ pos_s,
                           const int32_t *__restrict__ first, const int32_t *__restrict__ last, int64_t M,
                           float *__restrict__ mean, int32_t *__restrict__ count)
{
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int32_t a = first[m], b = last[m];       // both 0 for an empty group
    float s = 0.f;
    for (int32_t p = a; p < b; ++p) s += val[pos_s[p]];
    const int32_t c = b - a;
    mean[m] = s / (float)(c < 1 ? 1 : c);
    if (count) count[m] = c;
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

// sngnn_tuning_set(7, ks): force the contraction split of the dense cosine (0 = by shape)
static int g_cosine_split = 0;
namespace sngnn { int set_cosine_split(int v) { if (v < 0 || v > 16) return SNGNN_EINVAL; g_cosine_split = v; return SNGNN_OK; } }

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
    // the kernel reads 16-byte vectors from rows of ld floats: a zero-padded copy when the rows
    // of x do not allow that (F % 4 != 0 - Cora's 1433, Chameleon's 2325 - or a misaligned base)
    const int64_t ld = (F + 3) / 4 * 4;
    AsyncBuf xpad(st);
    const float *xs = x;
    if (ld != F || (uintptr_t)x % 16 != 0) {
        SN_REQUIRE(xpad.alloc((size_t)N * ld * 4) == 0, SNGNN_ENOMEM, "out of device memory");
        k_pad_rows<<<(unsigned)((N * ld + 255) / 256), 256, 0, st>>>(x, N, F, ld, xpad.as<float>());
        xs = xpad.as<float>();
    }
    // tiles for at most three quarters of the CUs: split the contraction (at least two K-steps per
    // split); the partial tiles are added in split order by k_cosine_reduce (deterministic)
    int ks = 1;
    if (tiles <= 96) ks = (int)std::min<int64_t>(std::min<int64_t>(8, (512 + tiles - 1) / tiles), std::max<int64_t>(1, ld / (2 * TB_K)));
    // fewer tiles than CUs with a long contraction (Chameleon: 171 tiles, F = 2 325): two halves per tile
    // put a second workgroup on most CUs - 0.203 -> 0.176 ms (three or four splits: the reduce pass eats it)
    else if (tiles < 220 && ld >= 16 * TB_K) ks = 2;
    if (g_cosine_split > 0) ks = (int)std::min<int64_t>(g_cosine_split, std::max<int64_t>(1, ld / (2 * TB_K)));    // (measurement)
    const int64_t k_per = ((ld + ks - 1) / ks + TB_K - 1) / TB_K * TB_K;
    ks = (int)((ld + k_per - 1) / k_per);
    AsyncBuf part(st);
    if (ks > 1) SN_REQUIRE(part.alloc((size_t)ks * N * N * 4) == 0, SNGNN_ENOMEM, "out of device memory");
    if (sngnn::fp32_mfma_only())
        k_cosine_mfma<false><<<tiles * ks, 256, 0, st>>>(xs, N, ld, inv.as<float>(), S, nb, ks, k_per, part.as<float>());
    else
        k_cosine_mfma<true><<<tiles * ks, 256, 0, st>>>(xs, N, ld, inv.as<float>(), S, nb, ks, k_per, part.as<float>());
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
    AsyncBuf inv(st), iota(st), order(st), ys(st), M(st), bad(st), tmp(st), P(st), dg(st);
    const int64_t n_seg = (N + CS_ROWS - 1) / CS_ROWS;
    SN_REQUIRE(!inv.alloc((size_t)N * 4) && !iota.alloc((size_t)N * 4) && !order.alloc((size_t)N * 4) &&
                   !ys.alloc((size_t)N * 4) && !M.alloc((size_t)n_classes * F * 8) && !bad.alloc(4) &&
                   !P.alloc((size_t)n_seg * 2 * F * 8) && !dg.alloc(diag_sum ? (size_t)N * 8 : 8),
               SNGNN_ENOMEM, "out of device memory");
    SN_HIP(hipMemsetAsync(bad.p, 0, 4, st));
    SN_HIP(hipMemsetAsync(M.p, 0, (size_t)n_classes * F * 8, st));
    const unsigned gn = (unsigned)((N + 255) / 256);
    k_check_labels<<<gn, 256, 0, st>>>(y, N, n_classes, bad.as<int>());
    int h_bad = 0;
    SN_HIP(hipMemcpyAsync(&h_bad, bad.p, 4, hipMemcpyDeviceToHost, st));
    SN_HIP(hipStreamSynchronize(st));
    SN_REQUIRE(!h_bad, SNGNN_ERANGE, "label outside [0, n_classes)");
    k_row_inv_norm<<<(unsigned)((N + 3) / 4), 256, 0, st>>>(x, N, F, inv.as<float>(),
                                                            diag_sum ? dg.as<double>() : nullptr);
    if (diag_sum) k_sum_fixed_d<<<1, 1024, 0, st>>>(dg.as<double>(), N, diag_sum);
    k_iota32<<<gn, 256, 0, st>>>(iota.as<int32_t>(), N);
    size_t tb = 0;
    SN_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, y, ys.as<int32_t>(), iota.as<int32_t>(),
                                              order.as<int32_t>(), (int)N, 0, 32, st));
    SN_REQUIRE(tmp.alloc(tb) == 0, SNGNN_ENOMEM, "out of device memory");
    SN_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, y, ys.as<int32_t>(), iota.as<int32_t>(),
                                              order.as<int32_t>(), (int)N, 0, 32, st));
    SN_REQUIRE(n_seg <= 65535, SNGNN_EINVAL, "too many rows (grid.y)");
    dim3 grid((unsigned)((F + 255) / 256), (unsigned)n_seg);
    k_class_row_sums<<<grid, 256, 0, st>>>(x, N, F, inv.as<float>(), order.as<int32_t>(),
                                           ys.as<int32_t>(), M.as<double>(), P.as<double>());
    k_class_join<<<dim3((unsigned)((F + 255) / 256), (unsigned)n_classes), 256, 0, st>>>(
        ys.as<int32_t>(), N, F, P.as<double>(), M.as<double>());
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

extern "C" int sngnn_segment_mean(const float *val, const int64_t *index, int64_t E, int64_t M, float *mean,
                                  int32_t *count, void *stream)
{
    SN_REQUIRE(E >= 0 && M >= 0, SNGNN_EINVAL, "bad shape");
    SN_REQUIRE(E < ((int64_t)1 << 31) && M < ((int64_t)1 << 31), SNGNN_EINVAL, "too many entries / groups");
    if (M == 0) { SN_REQUIRE(E == 0, SNGNN_ERANGE, "index outside [0, M)"); return SNGNN_OK; }
    SN_REQUIRE(mean && (E == 0 || (val && index)), SNGNN_EINVAL, "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    AsyncBuf key(st), pos(st), key_s(st), pos_s(st), bounds(st), bad(st), tmp(st);
    SN_REQUIRE(!key.alloc((size_t)E * 4) && !pos.alloc((size_t)E * 4) && !key_s.alloc((size_t)E * 4) &&
                   !pos_s.alloc((size_t)E * 4) && !bounds.alloc((size_t)M * 8) && !bad.alloc(4),
               SNGNN_ENOMEM, "out of device memory");
    SN_HIP(hipMemsetAsync(bad.p, 0, 4, st));
    SN_HIP(hipMemsetAsync(bounds.p, 0, (size_t)M * 8, st));
    int32_t *first = bounds.as<int32_t>(), *last = first + M;
    if (E > 0) {
        const unsigned ge = (unsigned)((E + 255) / 256);
        k_seg_keys<<<ge, 256, 0, st>>>(index, E, M, key.as<int32_t>(), pos.as<int32_t>(), bad.as<int>());
        int bits = 1;
        while (bits < 32 && ((int64_t)1 << bits) < M) ++bits;
        size_t tb = 0;
        SN_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, key.as<int32_t>(), key_s.as<int32_t>(),
                                                  pos.as<int32_t>(), pos_s.as<int32_t>(), (int)E, 0, bits, st));
        SN_REQUIRE(tmp.alloc(tb) == 0, SNGNN_ENOMEM, "out of device memory");
        SN_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, key.as<int32_t>(), key_s.as<int32_t>(),
                                                  pos.as<int32_t>(), pos_s.as<int32_t>(), (int)E, 0, bits, st));
        k_seg_bounds<<<ge, 256, 0, st>>>(key_s.as<int32_t>(), E, first, last);
    }
    k_seg_mean<<<(unsigned)((M + 255) / 256), 256, 0, st>>>(val, pos_s.as<int32_t>(), first, last, M, mean, count);
    int h_bad = 0;
    SN_HIP(hipMemcpyAsync(&h_bad, bad.p, 4, hipMemcpyDeviceToHost, st));
    SN_HIP(hipStreamSynchronize(st));
    SN_REQUIRE(!h_bad, SNGNN_ERANGE, "index outside [0, M)");
    return SNGNN_OK;
}
