// Cosine-attention mode of the gather skeleton (gfx950): AGNNConv of the reference,
// models/models.py:377-405 after ``lin``.
//
//     s_e     = cos(h_i, h_j)                              per in-edge e = (j -> i)
//     alpha_e = exp(s_e) / sum_{e' into i} exp(s_e')       (softmax per target)
//     out_i   = sum_e alpha_e * h_j                        (aggr = 'add')
//
// s_e is bounded by 1 in magnitude, so exp needs no running maximum and the row sum
// l_i = sum exp(s_e) and the weighted row sum are accumulated in ONE pass over the
// source rows (partial sums of split rows simply add).  Row classes as in the
// aggregation kernels (common.h): small rows one lane group each, wave rows one wave,
// split rows CHUNK-edge tasks + a per-row finalize.
//
// Backward.  With t_e = <G_i, h_j> and dot_i = sum_e alpha_e t_e (= <G_i, out_i>):
//     ds_e  = alpha_e (t_e - dot_i)                        softmax Jacobian
//     dnT_i = sum_e ds_e n_j = A_i - dot_i B_i,   A_i = sum alpha_e t_e n_j,  B_i = sum alpha_e n_j
// so pass T is again a single gather pass (A, B and dot accumulate together); it
// leaves the records {alpha_e, ds_e} per edge and dnT per row, after which the
// source-side pass S and the F.normalize Jacobian are the aggregation's own kernels
// (agg_bwd_impl.h).  No floating-point atomics anywhere.
#pragma once
#include "agg_bwd_impl.h"

namespace sngnn {

struct AttnArgs {
    const float *h;
    int C, N, row_off;
    const int32_t *col;
    const int4 *rdesc;           // per degree-sorted slot: {row, first edge, in-degree, 0}
    float *out, *alpha;          // alpha may be NULL
    int n_split, n_med_end, n_tasks;
    const int32_t *task_slot, *task_chunk, *split_task0;
    float *partial;              // [n_tasks][C + 4]: weighted row sum | l, 0, 0, 0
    int nbA, nbB;
};

constexpr int ATTN_LDS = 2 * WAVE_T;    // words per wave: source ids | exp(s) (or t_e)

// ------------------------------ forward ------------------------------------
template <int VEC, int G, int R>
__device__ __forceinline__ void attn_small(const AttnArgs &a, int blk, int *lds_wave)
{
    using RowT = Row<VEC, G, R>;
    constexpr int RPW = 64 / G;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int gid = lane / G, lg = lane % G;
    const int slot = a.n_med_end + (blk * WAVES + wave) * RPW + gid;
    if (slot >= a.N) return;                              // group-uniform
    const int4 d = a.rdesc[slot];
    const int i = d.x, rs = d.y, deg = d.z;
    int *s_j = lds_wave + gid * 2 * SMALL_T;
    float *s_e = reinterpret_cast<float *>(s_j + SMALL_T);
    for (int t = lg; t < deg; t += G) s_j[t] = a.col[rs + t];
    RowT hi, acc;
    hi.load(a.h + (size_t)(a.row_off + i) * a.C, a.C, lg);
    const float inv_i = inv_norm_of(group_sum<G>(hi.dot_partial(hi)));
    acc.zero();
    float l = 0.f;
    wave_lds_sync();
    for (int t0 = 0; t0 < deg; t0 += 2) {
        const bool two = t0 + 1 < deg;
        RowT x0, x1;
        x0.load(a.h + (size_t)s_j[t0] * a.C, a.C, lg);
        if (two) x1.load(a.h + (size_t)s_j[t0 + 1] * a.C, a.C, lg);
        const float e0 = expf(edge_score<VEC, G, R>(hi, inv_i, x0));
        l += e0;
        fma_row<VEC, G, R>(acc, e0, x0);
        if (lg == 0) s_e[t0] = e0;
        if (two) {
            const float e1 = expf(edge_score<VEC, G, R>(hi, inv_i, x1));
            l += e1;
            fma_row<VEC, G, R>(acc, e1, x1);
            if (lg == 0) s_e[t0 + 1] = e1;
        }
    }
    if (deg > 0) acc.div(l);
    acc.store(a.out + (size_t)i * a.C, a.C, lg);
    if (a.alpha) {
        wave_lds_sync();
        for (int t = lg; t < deg; t += G) a.alpha[rs + t] = s_e[t] / l;
    }
}

template <int VEC, int G, int R>
__device__ __forceinline__ void attn_wave(const AttnArgs &a, int blk, int *lds_wave, bool task)
{
    using RowT = Row<VEC, G, R>;
    constexpr int NG = 64 / G;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int gid = lane / G, lg = lane % G;
    int slot, e0 = 0, tq = 0;
    if (task) {
        tq = blk * WAVES + wave;
        if (tq >= a.n_tasks) return;
        slot = a.task_slot[tq];
        e0 = a.task_chunk[tq] * CHUNK;
    } else {
        slot = a.n_split + blk * WAVES + wave;
        if (slot >= a.n_med_end) return;
    }
    const int4 d = a.rdesc[slot];
    const int i = d.x, rs = d.y, deg = d.z;
    const int n = task ? min(deg - e0, CHUNK) : deg;      // <= WAVE_T edges for this wave
    int *s_j = lds_wave;
    float *s_e = reinterpret_cast<float *>(lds_wave + WAVE_T);
    for (int t = lane; t < n; t += 64) s_j[t] = a.col[rs + e0 + t];
    RowT hi, acc;
    hi.load(a.h + (size_t)(a.row_off + i) * a.C, a.C, lg);
    const float inv_i = inv_norm_of(group_sum<G>(hi.dot_partial(hi)));
    acc.zero();
    float l = 0.f;
    wave_lds_sync();
    for (int q0 = 0; q0 < n; q0 += NG) {
        const int q = q0 + gid;
        if (q < n) {
            RowT x;
            x.load(a.h + (size_t)s_j[q] * a.C, a.C, lg);
            const float e = expf(edge_score<VEC, G, R>(hi, inv_i, x));
            l += e;
            fma_row<VEC, G, R>(acc, e, x);
            if (lg == 0) s_e[q] = e;
        }
    }
    l = cross_group_sum<G>(l);
    acc.reduce_across_groups();
    wave_lds_sync();
    if (task) {
        // unnormalised: the row's finalize divides by the sum over all its tasks
        float *p = a.partial + (size_t)tq * (a.C + 4);
        if (gid == 0) {
            acc.store(p, a.C, lg);
            if (lg == 0) p[a.C] = l;
        }
        if (a.alpha)
            for (int t = lane; t < n; t += 64) a.alpha[rs + e0 + t] = s_e[t];
    } else {
        acc.div(l);
        if (gid == 0) acc.store(a.out + (size_t)i * a.C, a.C, lg);
        if (a.alpha)
            for (int t = lane; t < n; t += 64) a.alpha[rs + t] = s_e[t] / l;
    }
}

template <int VEC, int G, int R>
__global__ __launch_bounds__(BLOCK) void k_attn_fwd(const AttnArgs a)
{
    __shared__ int lds[WAVES][ATTN_LDS];
    const int b = blockIdx.x;
    int *lw = lds[threadIdx.x >> 6];
    if (b < a.nbA) attn_wave<VEC, G, R>(a, b, lw, true);
    else if (b < a.nbA + a.nbB) attn_wave<VEC, G, R>(a, b - a.nbA, lw, false);
    else attn_small<VEC, G, R>(a, b - a.nbA - a.nbB, lw);
}

// split rows: l_i and the weighted sum over the row's tasks (fixed order), then the
// row's exp(s_e) become alpha_e.  One 1024-thread workgroup per row; every step is wide - the
// first version summed l with ONE thread (a hub's ~100 partials as a serial chain of loads) and
// rewrote alpha one element per thread and trip: 34.5 us per forward at arxiv size (r04 epoch
// profile), most of it the 13 k-edge hub's latency chain.
constexpr int AFIN_BLOCK = 1024, AFIN_WAVES = AFIN_BLOCK / 64;

// sum of v over the workgroup in a fixed order (lanes by shuffle, waves in wave order); every thread gets it
__device__ __forceinline__ float afin_block_sum(float v, float *s_red)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    __syncthreads();                                   // (s_red may still be read from the previous use)
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < AFIN_WAVES; ++w) tot += s_red[w];
    return tot;
}

static __global__ __launch_bounds__(AFIN_BLOCK) void k_attn_fin(const AttnArgs a)
{
    __shared__ float s[AFIN_WAVES][64];
    __shared__ float s_red[AFIN_WAVES];
    const int p = blockIdx.x;
    const int4 d = a.rdesc[p];
    const int i = d.x, rs = d.y, deg = d.z;
    const int t0 = a.split_task0[p], t1 = a.split_task0[p + 1];
    const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const size_t stride = (size_t)a.C + 4;
    float lp = 0.f;
    for (int t = t0 + (int)threadIdx.x; t < t1; t += AFIN_BLOCK) lp += a.partial[t * stride + a.C];
    const float l = afin_block_sum(lp, s_red);
    for (int c0 = 0; c0 < a.C; c0 += 64) {
        const int c = c0 + cl;
        float v = 0.f;
        if (c < a.C)
            for (int t = t0 + q; t < t1; t += AFIN_WAVES) v += a.partial[t * stride + c];
        s[q][cl] = v;
        __syncthreads();
        if (q == 0 && c < a.C) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < AFIN_WAVES; ++w) tot += s[w][cl];
            a.out[(size_t)i * a.C + c] = tot / l;
        }
        __syncthreads();
    }
    if (a.alpha) {
        // eight elements per thread in flight (loads first, then the stores: the array aliases itself)
        float *al = a.alpha + rs;
        for (int base = 0; base < deg; base += 8 * AFIN_BLOCK) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = base + u * AFIN_BLOCK + (int)threadIdx.x;
                v[u] = t < deg ? al[t] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = base + u * AFIN_BLOCK + (int)threadIdx.x;
                if (t < deg) al[t] = v[u] / l;
            }
        }
    }
}

template <int VEC, int G, int R> int launch_attn_fwd(const AttnArgs &a0, hipStream_t st)
{
    constexpr int RPW = 64 / G;
    AttnArgs a = a0;
    a.nbA = ceil_div(a.n_tasks, WAVES);
    a.nbB = ceil_div(a.n_med_end - a.n_split, WAVES);
    const int nbC = ceil_div(a.N - a.n_med_end, (int64_t)WAVES * RPW);
    if (a.nbA + a.nbB + nbC > 0) k_attn_fwd<VEC, G, R><<<a.nbA + a.nbB + nbC, BLOCK, 0, st>>>(a);
    if (a.n_split > 0) k_attn_fin<<<a.n_split, AFIN_BLOCK, 0, st>>>(a);
    SN_HIP(hipGetLastError());
    return SNGNN_OK;
}

// ------------------------------ backward, pass T ----------------------------
// BwdArgs (agg_bwd_impl.h) with wsel = alpha, records {alpha_e, ds_e} at the edge's CSC position, partT rows
// of 2C + 4 floats.

// one in-edge: accumulates A, B, dot; returns t_e
template <int VEC, int G, int R>
__device__ __forceinline__ float attn_t_edge(const Row<VEC, G, R> &x, const Row<VEC, G, R> &gp,
                                             float al, Row<VEC, G, R> &A, Row<VEC, G, R> &B,
                                             float &dot)
{
    const float invj = inv_norm_of(group_sum<G>(x.dot_partial(x)));    // same bits as the forward
    const float t = group_sum<G>(gp.dot_partial(x));
    const float at = al * t;
    dot += at;
    fma_row<VEC, G, R>(A, at * invj, x);
    fma_row<VEC, G, R>(B, al * invj, x);
    return t;
}

template <int VEC, int G, int R>
__device__ __forceinline__ void attn_t_small(const BwdArgs &a, int blk, int *lds_wave)
{
    using RowT = Row<VEC, G, R>;
    constexpr int RPW = 64 / G;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int gid = lane / G, lg = lane % G;
    const int slot = a.n_med_end + (blk * WAVES + wave) * RPW + gid;
    if (slot >= a.N) return;
    const int i = a.rperm[slot];
    const int rs = a.rowptr[i];
    const int deg = a.rowptr[i + 1] - rs;
    int *s_j = lds_wave + gid * 3 * SMALL_T;
    float *s_a = reinterpret_cast<float *>(s_j + SMALL_T);
    float *s_t = reinterpret_cast<float *>(s_j + 2 * SMALL_T);
    for (int t = lg; t < deg; t += G) {
        s_j[t] = a.col[rs + t];
        s_a[t] = a.wsel[rs + t];
    }
    RowT gp, A, B;
    gp.load(a.gout + (size_t)i * a.C, a.C, lg);
    A.zero();
    B.zero();
    float dot = 0.f;
    wave_lds_sync();
    for (int t0 = 0; t0 < deg; t0 += 2) {
        const bool two = t0 + 1 < deg;
        RowT x0, x1;
        x0.load(a.h + (size_t)s_j[t0] * a.C, a.C, lg);
        if (two) x1.load(a.h + (size_t)s_j[t0 + 1] * a.C, a.C, lg);
        const float ta = attn_t_edge<VEC, G, R>(x0, gp, s_a[t0], A, B, dot);
        if (lg == 0) s_t[t0] = ta;
        if (two) {
            const float tb = attn_t_edge<VEC, G, R>(x1, gp, s_a[t0 + 1], A, B, dot);
            if (lg == 0) s_t[t0 + 1] = tb;
        }
    }
    fma_row<VEC, G, R>(A, -dot, B);                       // dnT_i = A - dot * B
    A.store(a.dnT + (size_t)i * a.C, a.C, lg);
    wave_lds_sync();
    for (int t = lg; t < deg; t += G) a.wd[a.csc_pos[rs + t]] = make_float2(s_a[t], s_a[t] * (s_t[t] - dot));
}

template <int VEC, int G, int R>
__device__ __forceinline__ void attn_t_wave(const BwdArgs &a, int blk, int *lds_wave, bool task)
{
    using RowT = Row<VEC, G, R>;
    constexpr int NG = 64 / G;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int gid = lane / G, lg = lane % G;
    int i, e0 = 0, tq = 0;
    if (task) {
        tq = blk * WAVES + wave;
        if (tq >= a.n_tasks) return;
        i = a.rperm[a.task_slot[tq]];
        e0 = a.task_chunk[tq] * CHUNK;
    } else {
        const int slot = a.n_split + blk * WAVES + wave;
        if (slot >= a.n_med_end) return;
        i = a.rperm[slot];
    }
    const int rs = a.rowptr[i];
    const int deg = a.rowptr[i + 1] - rs;
    const int n = task ? min(deg - e0, CHUNK) : deg;
    int *s_j = lds_wave;
    float *s_a = reinterpret_cast<float *>(lds_wave + WAVE_T);
    float *s_t = reinterpret_cast<float *>(lds_wave + 2 * WAVE_T);
    for (int t = lane; t < n; t += 64) {
        s_j[t] = a.col[rs + e0 + t];
        s_a[t] = a.wsel[rs + e0 + t];
    }
    RowT gp, A, B;
    gp.load(a.gout + (size_t)i * a.C, a.C, lg);
    A.zero();
    B.zero();
    float dot = 0.f;
    wave_lds_sync();
    for (int q0 = 0; q0 < n; q0 += NG) {
        const int q = q0 + gid;
        if (q < n) {
            RowT x;
            x.load(a.h + (size_t)s_j[q] * a.C, a.C, lg);
            const float t = attn_t_edge<VEC, G, R>(x, gp, s_a[q], A, B, dot);
            if (lg == 0) s_t[q] = t;
        }
    }
    dot = cross_group_sum<G>(dot);
    A.reduce_across_groups();
    B.reduce_across_groups();
    wave_lds_sync();
    if (task) {
        // a split row's dot_i is only known after its finalize: its records stay RAW, {-alpha_e, t_e} (the
        // sign is the mark), and pass S subtracts dot_i itself (BwdArgs::rec_dot)
        float *p = a.partT + (size_t)tq * (2 * a.C + 4);
        if (gid == 0) {
            A.store(p, a.C, lg);
            B.store(p + a.C, a.C, lg);
            if (lg == 0) p[2 * a.C] = dot;
        }
        for (int t = lane; t < n; t += 64) a.wd[a.csc_pos[rs + e0 + t]] = make_float2(-s_a[t], s_t[t]);
    } else {
        fma_row<VEC, G, R>(A, -dot, B);
        if (gid == 0) A.store(a.dnT + (size_t)i * a.C, a.C, lg);
        for (int t = lane; t < n; t += 64) a.wd[a.csc_pos[rs + t]] = make_float2(s_a[t], s_a[t] * (s_t[t] - dot));
    }
}

template <int VEC, int G, int R>
__global__ __launch_bounds__(BLOCK) void k_attn_bwd_t(const BwdArgs a)
{
    __shared__ __align__(16) int lds[WAVES][3 * WAVE_T];
    const int b = blockIdx.x;
    int *lw = lds[threadIdx.x >> 6];
    if (b < a.nbA) attn_t_wave<VEC, G, R>(a, b, lw, true);
    else if (b < a.nbA + a.nbB) attn_t_wave<VEC, G, R>(a, b - a.nbA, lw, false);
    else attn_t_small<VEC, G, R>(a, b - a.nbA - a.nbB, lw);
}

// Split rows after the tasks: dnT_i and dot_i from the row's task partials, one workgroup per row, every step
// wide.  (Round 4, two steps.  The first version summed dot_i with ONE thread and rewrote the row's records
// from raw t_e to ds_e one per thread and trip behind a csc_pos -> record chain: 68 us per backward at arxiv
// size, the 13 k-edge hub's latency chain.  A wave per 128-edge task for the rewrite: 23 us.  Now no record is
// rewritten at all - the tasks mark theirs raw and pass S subtracts dot_i itself, BwdArgs::rec_dot.)
__device__ __forceinline__ float attn_row_dot(const BwdArgs &a, int t0, int t1, size_t stride)
{
    float dp = 0.f;
    for (int t = t0 + lane_id(); t < t1; t += 64) dp += a.partT[t * stride + 2 * a.C];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) dp += __shfl_xor(dp, m, 64);
    return dp;
}

static __global__ __launch_bounds__(BLOCK) void k_attn_bwd_t_fin(const BwdArgs a)
{
    __shared__ float sA[WAVES][64], sB[WAVES][64];
    const size_t stride = 2 * (size_t)a.C + 4;
    const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int p = blockIdx.x;
    const int i = a.rperm[p];
    const int t0 = a.split_task0[p], t1 = a.split_task0[p + 1];
    const float dot = attn_row_dot(a, t0, t1, stride);      // (every wave: the same fixed order)
    if (threadIdx.x == 0) a.rec_dot[i] = dot;
    for (int c0 = 0; c0 < a.C; c0 += 64) {
        const int c = c0 + cl;
        float va = 0.f, vb = 0.f;
        if (c < a.C)
            for (int t = t0 + q; t < t1; t += WAVES) {
                va += a.partT[t * stride + c];
                vb += a.partT[t * stride + a.C + c];
            }
        sA[q][cl] = va;
        sB[q][cl] = vb;
        __syncthreads();
        if (q == 0 && c < a.C) {
            float A = 0.f, B = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) { A += sA[w][cl]; B += sB[w][cl]; }
            a.dnT[(size_t)i * a.C + c] = fmaf(-dot, B, A);
        }
        __syncthreads();
    }
}

template <int VEC, int G, int R> int launch_attn_bwd(const BwdArgs &a0, hipStream_t st)
{
    constexpr int RPW = 64 / G;
    BwdArgs a = a0;
    a.nbA = ceil_div(a.n_tasks, WAVES);
    a.nbB = ceil_div(a.n_med_end - a.n_split, WAVES);
    int nbC = ceil_div(a.N - a.n_med_end, (int64_t)WAVES * RPW);
    if (a.nbA + a.nbB + nbC > 0) k_attn_bwd_t<VEC, G, R><<<a.nbA + a.nbB + nbC, BLOCK, 0, st>>>(a);
    if (a.n_split > 0) k_attn_bwd_t_fin<<<a.n_split, BLOCK, 0, st>>>(a);
    // pass S: the aggregation's kernels (every edge kept, weight alpha_e, no mean division)
    a.nbA = ceil_div(a.n_stasks, WAVES);
    a.nbB = ceil_div(a.n_smed_end - a.n_ssplit, WAVES);
    nbC = ceil_div(a.Ntot - a.n_smed_end, (int64_t)WAVES * RPW);
    if (a.nbA + a.nbB + nbC > 0) k_bwd_s<VEC, G, R, true><<<a.nbA + a.nbB + nbC, BLOCK, 0, st>>>(a);
    if (a.n_ssplit > 0) k_bwd_s_fin<VEC, G, R><<<a.n_ssplit, 64, 0, st>>>(a);
    SN_HIP(hipGetLastError());
    return SNGNN_OK;
}

int launch_attn_fwd_v1(const RowCfg &cfg, const AttnArgs &a, hipStream_t st);
int launch_attn_fwd_v2(const RowCfg &cfg, const AttnArgs &a, hipStream_t st);
int launch_attn_fwd_v4(const RowCfg &cfg, const AttnArgs &a, hipStream_t st);
int launch_attn_bwd_v1(const RowCfg &cfg, const BwdArgs &a, hipStream_t st);
int launch_attn_bwd_v2(const RowCfg &cfg, const BwdArgs &a, hipStream_t st);
int launch_attn_bwd_v4(const RowCfg &cfg, const BwdArgs &a, hipStream_t st);

}  // namespace sngnn
