// SNGNN++ adjacency-linear branch (gfx950): Linear(num_nodes, C) applied to the
// sparse adjacency, models/models.py:124-130, as a row gather-sum
//     forward   out0[i] = b + sum_{e : src_e - src_min == i} Wt[dst_e]      (CSC rows)
//     backward  dWt[d]  =     sum_{e : dst_e == d}           g0[src_e - src_min]   (CSR rows)
// Wt is the [N, C] transpose of the reference's w.weight so that every gathered
// operand is one contiguous row.  Same degree classes and fixed summation order
// as the aggregation kernels; no atomics.
#pragma once
#include "device_utils.h"

namespace sngnn {

struct AdjArgs {
    const float *table;         // rows to gather  [N, C]
    const float *w;             // per-entry weights in SEGMENT order (w[ptr[seg] + t]) or nullptr = 1 (weighted gather-sum:
                                // GGCNlayer_SP's plain propagation, models.py:1544-1549)
    const float *bias;          // [C] or nullptr
    float *out;                 // [N, C]
    float *partial;             // [n_tasks, C]
    int C, N;
    const int32_t *ptr, *idx, *perm;    // segments (CSR or CSC), gathered ids, degree order
    int seg_shift;              // output row r reads segment r + seg_shift
    int idx_shift;              // gathered row = idx[q] + idx_shift
    int n_split, n_med_end, n_tasks;
    const int32_t *task_slot, *task_chunk, *split_task0;
    int nbA, nbB;
};

template <int VEC, int G, int R>
__device__ __forceinline__ void adj_gather(const AdjArgs &a, int qs, int e0, int e1, int stride,
                                           int first, int lg, Row<VEC, G, R> &acc)
{
    using RowT = Row<VEC, G, R>;
    constexpr int U = 4 / (R >= 4 ? 4 : R);
    for (int base = e0 + first; base < e1; base += stride * U) {
        RowT x[U];
        bool act[U];
        float wv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = base + u * stride;
            act[u] = t < e1;
            const int row = act[u] ? a.idx[qs + t] + a.idx_shift : 0;
            wv[u] = (a.w && act[u]) ? a.w[qs + t] : 1.0f;         // (uniform pointer test; travels with the row)
            x[u].load(a.table + (size_t)row * a.C, a.C, lg);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (act[u]) {
                if (a.w) acc.axpy(wv[u], x[u]);                    // value * row rounded, then added (a sparse mm's order)
                else acc.add(x[u]);
            }
    }
}

template <int VEC, int G, int R>
__device__ __forceinline__ void adj_finish(const AdjArgs &a, int r, int lg, Row<VEC, G, R> &acc)
{
    if (a.bias) {
        Row<VEC, G, R> b;
        b.load(a.bias, a.C, lg);
        b.add(acc);              // bias first, then the gathered rows (sparse addmm order)
        acc = b;
    }
    acc.store(a.out + (size_t)r * a.C, a.C, lg);
}

template <int VEC, int G, int R>
__global__ __launch_bounds__(BLOCK) void k_adj(const AdjArgs a)
{
    using RowT = Row<VEC, G, R>;
    constexpr int NG = 64 / G;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int gid = lane / G, lg = lane % G;
    const int b = blockIdx.x;
    RowT acc;
    acc.zero();
    if (b < a.nbA + a.nbB) {
        // split task or one wave per segment
        const bool task = b < a.nbA;
        int seg, e0, tq = 0;
        if (task) {
            tq = b * WAVES + wave;
            if (tq >= a.n_tasks) return;
            seg = a.perm[a.task_slot[tq]];
            e0 = a.task_chunk[tq] * CHUNK;
        } else {
            const int slot = a.n_split + (b - a.nbA) * WAVES + wave;
            if (slot >= a.n_med_end) return;
            seg = a.perm[slot];
            e0 = 0;
        }
        const int qs = a.ptr[seg];
        const int deg = a.ptr[seg + 1] - qs;
        const int e1 = task ? min(deg, e0 + CHUNK) : deg;
        adj_gather<VEC, G, R>(a, qs, e0, e1, NG, gid, lg, acc);
        acc.reduce_across_groups();
        if (gid != 0) return;
        if (task) acc.store(a.partial + (size_t)tq * a.C, a.C, lg);
        else if (seg - a.seg_shift >= 0) adj_finish<VEC, G, R>(a, seg - a.seg_shift, lg, acc);
    } else {
        const int slot = a.n_med_end + ((b - a.nbA - a.nbB) * WAVES + wave) * NG + gid;
        if (slot >= a.N) return;
        const int seg = a.perm[slot];
        const int qs = a.ptr[seg];
        const int deg = a.ptr[seg + 1] - qs;
        adj_gather<VEC, G, R>(a, qs, 0, deg, 1, 0, lg, acc);
        if (seg - a.seg_shift >= 0) adj_finish<VEC, G, R>(a, seg - a.seg_shift, lg, acc);
    }
}

// split segments: the sum of the tasks' partial rows (+ bias).  Thread (c, q) adds every
// 4th task in four independent chains (the loads of a hub's ~100 tasks overlap); the sums
// are combined in fixed order (deterministic).
static __global__ __launch_bounds__(256) void k_adj_fin(const AdjArgs a)
{
    __shared__ float s[4][64];
    const int p = blockIdx.x;
    const int seg = a.perm[p];
    const int r = seg - a.seg_shift;
    if (r < 0) return;                                  // block-uniform
    const int t0 = a.split_task0[p], t1 = a.split_task0[p + 1];
    const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
    for (int c0 = 0; c0 < a.C; c0 += 64) {
        const int c = c0 + cl;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        if (c < a.C) {
            int t = t0 + q;
            for (; t + 12 < t1; t += 16) {
                v0 += a.partial[(size_t)t * a.C + c];
                v1 += a.partial[(size_t)(t + 4) * a.C + c];
                v2 += a.partial[(size_t)(t + 8) * a.C + c];
                v3 += a.partial[(size_t)(t + 12) * a.C + c];
            }
            for (; t < t1; t += 4) v0 += a.partial[(size_t)t * a.C + c];
        }
        s[q][cl] = (v0 + v1) + (v2 + v3);
        __syncthreads();
        if (q == 0 && c < a.C)
            a.out[(size_t)r * a.C + c] = ((s[0][cl] + s[1][cl]) + (s[2][cl] + s[3][cl])) + (a.bias ? a.bias[c] : 0.f);
        __syncthreads();
    }
}

// output rows whose segment lies beyond the node range (src_min > 0): bias only
static __global__ void k_adj_tail(const AdjArgs a, int first_row)
{
    const int64_t n = (int64_t)(a.N - first_row) * a.C;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n;
         t += (int64_t)gridDim.x * blockDim.x)
        a.out[(size_t)first_row * a.C + t] = a.bias ? a.bias[t % a.C] : 0.f;
}

// out[e] = <A[ia[e]], B[ib[e]]> for e in [0, E): one lane group per entry, two entries per group in flight
template <int VEC, int G, int R>
__global__ __launch_bounds__(BLOCK) void k_pair_dot(const float *__restrict__ A, const int32_t *__restrict__ ia,
                                                    const float *__restrict__ B, const int32_t *__restrict__ ib,
                                                    int64_t E, int C, float *__restrict__ out)
{
    using RowT = Row<VEC, G, R>;
    constexpr int NG = 64 / G;
    const int lane = lane_id(), gid = lane / G, lg = lane % G;
    const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6, nw = (int64_t)gridDim.x * WAVES;
    for (int64_t base = wave * 2 * NG; base < E; base += nw * 2 * NG) {
        const int64_t e0 = base + gid, e1 = base + NG + gid;
        const int64_t c0 = e0 < E ? e0 : E - 1, c1 = e1 < E ? e1 : E - 1;
        RowT a0, b0, a1, b1;
        a0.load(A + (size_t)ia[c0] * C, C, lg);
        b0.load(B + (size_t)ib[c0] * C, C, lg);
        a1.load(A + (size_t)ia[c1] * C, C, lg);
        b1.load(B + (size_t)ib[c1] * C, C, lg);
        const float d0 = group_sum<G>(a0.dot_partial(b0)), d1 = group_sum<G>(a1.dot_partial(b1));
        if (lg == 0) {
            if (e0 < E) out[e0] = d0;
            if (e1 < E) out[e1] = d1;
        }
    }
}
template <int VEC, int G, int R>
int launch_pair_dot(const float *A, const int32_t *ia, const float *B, const int32_t *ib, int64_t E, int C, float *out,
                    hipStream_t st)
{
    if (E == 0) return SNGNN_OK;
    const int grid = (int)std::min<int64_t>(ceil_div(E, (int64_t)2 * (64 / G) * WAVES), 256 * 8);
    k_pair_dot<VEC, G, R><<<grid, BLOCK, 0, st>>>(A, ia, B, ib, E, C, out);
    SN_HIP(hipGetLastError());
    return SNGNN_OK;
}

template <int VEC, int G, int R> int launch_adj(const AdjArgs &a0, hipStream_t st)
{
    constexpr int NG = 64 / G;
    AdjArgs a = a0;
    a.nbA = ceil_div(a.n_tasks, WAVES);
    a.nbB = ceil_div(a.n_med_end - a.n_split, WAVES);
    const int nbC = ceil_div(a.N - a.n_med_end, (int64_t)WAVES * NG);
    if (a.nbA + a.nbB + nbC > 0) k_adj<VEC, G, R><<<a.nbA + a.nbB + nbC, BLOCK, 0, st>>>(a);
    if (a.n_split > 0) k_adj_fin<<<a.n_split, 256, 0, st>>>(a);
    if (a.seg_shift > 0) k_adj_tail<<<std::min(1024, ceil_div((int64_t)a.seg_shift * a.C, 256)), 256, 0, st>>>(a, a.N - a.seg_shift);
    SN_HIP(hipGetLastError());
    return SNGNN_OK;
}

}  // namespace sngnn
