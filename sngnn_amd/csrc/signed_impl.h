// Signed cosine attention of the gather skeleton (gfx950): GGCNlayer_SP of the reference,
// models/models.py:1453-1553 (use_sign branch) after ``fcn``.
//
//     s_e     = cos(Wh_i, Wh_j)                            per entry e = (i, j) of the adjacency, i != j
//     e_pos   = relu(s_e),  e_neg = -relu(-s_e)            (get_sparse_att, :1512-1519)
//     prop_pos_i = sum_e a_e e_pos Wh_j,  prop_neg_i = sum_e a_e e_neg Wh_j      (:1536-1537)
//     result  = scale (coeff_0 prop_pos + coeff_1 prop_neg + coeff_2 Wh)         (:1541)
// with a_e = adj_e * softplus(deg_coeff_0 * degree_e + deg_coeff_1) (adj_remove_diag * sc, :1529-1530).
// The two propagations share every gathered row Wh_j, so they are ONE gather here:
//     out_i = sum_e a_e kappa(s_e) s_e Wh_j,   kappa(s) = c_pos (s > 0) | c_neg (s < 0) | 0
// (c_pos = coeff_0, c_neg = coeff_1: device scalars; scale and the coeff_2 Wh term are elementwise
// and stay with the caller).  The cosines are saved (CSR order) for the backward.
//
// Backward.  With t_e = <G_i, Wh_j>:
//     d a_e    = kappa_e s_e t_e          (the caller multiplies by kappa: this kernel writes u_e = s_e t_e)
//     d c_pos  = sum_{s_e > 0} a_e u_e,   d c_neg = sum_{s_e < 0} a_e u_e        (caller, from u)
//     ds_e     = a_e kappa_e t_e,         w_e = a_e kappa_e s_e
//     dnT_i    = sum_e ds_e n_j           (pass T, below: records {w_e, ds_e} at the edge's CSC position)
//     msg_j    = sum_e w_e G_i,  dnS_j = sum_e ds_e n_i,  dWh = msg + normalize-Jacobian(dnT + dnS)
// - pass S and the Jacobian are the aggregation's own kernels (agg_bwd_impl.h, REC mode), as in
// the attention mode.  No floating-point atomics anywhere; fixed summation order.
//
// F.cosine_similarity clamps at eps = 1e-8 where F.normalize clamps at 1e-12: rows with a norm
// below 1e-8 (other than zero rows, whose cosine is 0 either way) would differ; not special-cased.
#pragma once
#include "agg_bwd_impl.h"

namespace sngnn {

struct SignedArgs {
    const float *h;              // Wh [Ntot, C]
    const float *coef;           // a_e, CSR order [E']
    const float *c2;             // dev [2]: c_pos, c_neg
    int C, N, row_off;
    const int32_t *col;
    const int4 *rdesc;           // per degree-sorted slot: {row, first edge, in-degree, 0}
    float *out, *s;              // s: cosines in CSR order (may be NULL when nothing is saved)
    int n_split, n_med_end, n_tasks;
    const int32_t *task_slot, *task_chunk, *split_task0;
    float *partial;              // [n_tasks][C]
    int nbA, nbB;
};

constexpr int SIGNED_LDS = 3 * WAVE_T;    // words per wave: source ids | a_e | s_e (or t_e)

__device__ __forceinline__ float signed_kappa(float s, float cp, float cn) { return s > 0.f ? cp : (s < 0.f ? cn : 0.f); }

// ------------------------------ forward ------------------------------------
template <int VEC, int G, int R>
__device__ __forceinline__ void signed_small(const SignedArgs &a, int blk, int *lds_wave, float cp, float cn)
{
    using RowT = Row<VEC, G, R>;
    constexpr int RPW = 64 / G;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int gid = lane / G, lg = lane % G;
    const int slot = a.n_med_end + (blk * WAVES + wave) * RPW + gid;
    if (slot >= a.N) return;                              // group-uniform
    const int4 d = a.rdesc[slot];
    const int i = d.x, rs = d.y, deg = d.z;
    int *s_j = lds_wave + gid * 3 * SMALL_T;
    float *s_a = reinterpret_cast<float *>(s_j + SMALL_T);
    float *s_e = reinterpret_cast<float *>(s_j + 2 * SMALL_T);
    for (int t = lg; t < deg; t += G) {
        s_j[t] = a.col[rs + t];
        s_a[t] = a.coef[rs + t];
    }
    RowT hi, acc;
    hi.load(a.h + (size_t)(a.row_off + i) * a.C, a.C, lg);
    const float inv_i = inv_norm_of(group_sum<G>(hi.dot_partial(hi)));
    acc.zero();
    wave_lds_sync();
    for (int t0 = 0; t0 < deg; t0 += 2) {
        const bool two = t0 + 1 < deg;
        RowT x0, x1;
        x0.load(a.h + (size_t)s_j[t0] * a.C, a.C, lg);
        x1.load(a.h + (size_t)s_j[two ? t0 + 1 : t0] * a.C, a.C, lg);
        const float e0 = edge_score<VEC, G, R>(hi, inv_i, x0);
        const float e1 = edge_score<VEC, G, R>(hi, inv_i, x1);
        fma_row<VEC, G, R>(acc, s_a[t0] * signed_kappa(e0, cp, cn) * e0, x0);
        if (lg == 0) s_e[t0] = e0;
        if (two) {
            fma_row<VEC, G, R>(acc, s_a[t0 + 1] * signed_kappa(e1, cp, cn) * e1, x1);
            if (lg == 0) s_e[t0 + 1] = e1;
        }
    }
    acc.store(a.out + (size_t)i * a.C, a.C, lg);
    if (a.s) {
        wave_lds_sync();
        for (int t = lg; t < deg; t += G) a.s[rs + t] = s_e[t];
    }
}

template <int VEC, int G, int R>
__device__ __forceinline__ void signed_wave(const SignedArgs &a, int blk, int *lds_wave, bool task, float cp, float cn)
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
    float *s_a = reinterpret_cast<float *>(lds_wave + WAVE_T);
    float *s_e = reinterpret_cast<float *>(lds_wave + 2 * WAVE_T);
    for (int t = lane; t < n; t += 64) {
        s_j[t] = a.col[rs + e0 + t];
        s_a[t] = a.coef[rs + e0 + t];
    }
    RowT hi, acc;
    hi.load(a.h + (size_t)(a.row_off + i) * a.C, a.C, lg);
    const float inv_i = inv_norm_of(group_sum<G>(hi.dot_partial(hi)));
    acc.zero();
    wave_lds_sync();
    // two rows per lane group in flight, unconditionally (a slot past the end repeats the last
    // edge with weight 0: no load behind a branch)
    for (int q0 = 0; q0 < n; q0 += 2 * NG) {
        const int qa = min(q0 + gid, n - 1), qb = min(q0 + NG + gid, n - 1);
        const bool la = q0 + gid < n, lb = q0 + NG + gid < n;
        RowT xa, xb;
        xa.load(a.h + (size_t)s_j[qa] * a.C, a.C, lg);
        xb.load(a.h + (size_t)s_j[qb] * a.C, a.C, lg);
        const float ea = edge_score<VEC, G, R>(hi, inv_i, xa);
        const float eb = edge_score<VEC, G, R>(hi, inv_i, xb);
        fma_row<VEC, G, R>(acc, la ? s_a[qa] * signed_kappa(ea, cp, cn) * ea : 0.f, xa);
        fma_row<VEC, G, R>(acc, lb ? s_a[qb] * signed_kappa(eb, cp, cn) * eb : 0.f, xb);
        if (lg == 0 && la) s_e[qa] = ea;
        if (lg == 0 && lb) s_e[qb] = eb;
    }
    acc.reduce_across_groups();
    if (gid == 0) acc.store(task ? a.partial + (size_t)tq * a.C : a.out + (size_t)i * a.C, a.C, lg);
    if (a.s) {
        wave_lds_sync();
        for (int t = lane; t < n; t += 64) a.s[rs + e0 + t] = s_e[t];
    }
    wave_lds_sync();
}

template <int VEC, int G, int R>
__global__ __launch_bounds__(BLOCK) void k_signed_fwd(const SignedArgs a)
{
    __shared__ int lds[WAVES][SIGNED_LDS];
    const int b = blockIdx.x;
    int *lw = lds[threadIdx.x >> 6];
    const float cp = a.c2[0], cn = a.c2[1];
    if (b < a.nbA) signed_wave<VEC, G, R>(a, b, lw, true, cp, cn);
    else if (b < a.nbA + a.nbB) signed_wave<VEC, G, R>(a, b - a.nbA, lw, false, cp, cn);
    else signed_small<VEC, G, R>(a, b - a.nbA - a.nbB, lw, cp, cn);
}

// split rows: the tasks' partial rows added in task order (four chains, combined in fixed order)
static __global__ __launch_bounds__(256) void k_signed_fin(const SignedArgs a)
{
    __shared__ float s[4][64];
    const int p = blockIdx.x;
    const int i = a.rdesc[p].x;
    const int t0 = a.split_task0[p], t1 = a.split_task0[p + 1];
    const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
    for (int c0 = 0; c0 < a.C; c0 += 64) {
        const int c = c0 + cl;
        float v = 0.f;
        if (c < a.C)
            for (int t = t0 + q; t < t1; t += 4) v += a.partial[(size_t)t * a.C + c];
        s[q][cl] = v;
        __syncthreads();
        if (q == 0 && c < a.C) a.out[(size_t)i * a.C + c] = (s[0][cl] + s[1][cl]) + (s[2][cl] + s[3][cl]);
        __syncthreads();
    }
}

template <int VEC, int G, int R> int launch_signed_fwd(const SignedArgs &a0, hipStream_t st)
{
    constexpr int RPW = 64 / G;
    SignedArgs a = a0;
    a.nbA = ceil_div(a.n_tasks, WAVES);
    a.nbB = ceil_div(a.n_med_end - a.n_split, WAVES);
    const int nbC = ceil_div(a.N - a.n_med_end, (int64_t)WAVES * RPW);
    if (a.nbA + a.nbB + nbC > 0) k_signed_fwd<VEC, G, R><<<a.nbA + a.nbB + nbC, BLOCK, 0, st>>>(a);
    if (a.n_split > 0) k_signed_fin<<<a.n_split, 256, 0, st>>>(a);
    SN_HIP(hipGetLastError());
    return SNGNN_OK;
}

// ------------------------------ backward, pass T ----------------------------
// BwdArgs (agg_bwd_impl.h) with wsel = the saved cosines s_e (CSR order), records {w_e, ds_e} at the
// edge's CSC position; the extra per-edge arrays ride in SignedBwdExtra.
struct SignedBwdExtra {
    const float *coef;           // a_e, CSR order
    const float *c2;             // dev [2]
    float *u;                    // [E'] CSR order: s_e t_e
};

// one in-edge: t_e = <G_i, Wh_j>; accumulates dnT_i += ds_e n_j; returns t_e
template <int VEC, int G, int R>
__device__ __forceinline__ float signed_t_edge(const Row<VEC, G, R> &x, const Row<VEC, G, R> &gp, float ak,
                                               Row<VEC, G, R> &acc)
{
    const float invj = inv_norm_of(group_sum<G>(x.dot_partial(x)));    // same bits as the forward
    const float t = group_sum<G>(gp.dot_partial(x));
    fma_row<VEC, G, R>(acc, ak * t * invj, x);                          // ak = a_e kappa_e (0 for a padding repeat)
    return t;
}

template <int VEC, int G, int R>
__device__ __forceinline__ void signed_t_small(const BwdArgs &a, const SignedBwdExtra &x, int blk, int *lds_wave,
                                               float cp, float cn)
{
    using RowT = Row<VEC, G, R>;
    constexpr int RPW = 64 / G;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int gid = lane / G, lg = lane % G;
    const int slot = a.n_med_end + (blk * WAVES + wave) * RPW + gid;
    if (slot >= a.N) return;
    const int4 d = a.rdesc[slot];
    const int i = d.x, rs = d.y, deg = d.z;
    int *s_j = lds_wave + gid * 4 * SMALL_T;
    float *s_a = reinterpret_cast<float *>(s_j + SMALL_T);      // a_e kappa_e
    float *s_s = reinterpret_cast<float *>(s_j + 2 * SMALL_T);  // s_e
    float *s_t = reinterpret_cast<float *>(s_j + 3 * SMALL_T);  // t_e
    for (int t = lg; t < deg; t += G) {
        const float se = a.wsel[rs + t];
        s_j[t] = a.col[rs + t];
        s_s[t] = se;
        s_a[t] = x.coef[rs + t] * signed_kappa(se, cp, cn);
    }
    RowT gp, acc;
    gp.load(a.gout + (size_t)i * a.C, a.C, lg);
    acc.zero();
    wave_lds_sync();
    for (int t0 = 0; t0 < deg; t0 += 2) {
        const bool two = t0 + 1 < deg;
        const int t1 = two ? t0 + 1 : t0;
        RowT x0, x1;
        x0.load(a.h + (size_t)s_j[t0] * a.C, a.C, lg);
        x1.load(a.h + (size_t)s_j[t1] * a.C, a.C, lg);
        const float ta = signed_t_edge<VEC, G, R>(x0, gp, s_a[t0], acc);
        const float tb = signed_t_edge<VEC, G, R>(x1, gp, two ? s_a[t1] : 0.f, acc);
        if (lg == 0) s_t[t0] = ta;
        if (lg == 0 && two) s_t[t1] = tb;
    }
    acc.store(a.dnT + (size_t)i * a.C, a.C, lg);
    wave_lds_sync();
    for (int t = lg; t < deg; t += G) {
        a.wd[a.csc_pos[rs + t]] = make_float2(s_a[t] * s_s[t], s_a[t] * s_t[t]);
        x.u[rs + t] = s_s[t] * s_t[t];
    }
}

template <int VEC, int G, int R>
__device__ __forceinline__ void signed_t_wave(const BwdArgs &a, const SignedBwdExtra &x, int blk, int *lds_wave,
                                              bool task, float cp, float cn)
{
    using RowT = Row<VEC, G, R>;
    constexpr int NG = 64 / G;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int gid = lane / G, lg = lane % G;
    int e0 = 0, tq = 0;
    int4 d;
    if (task) {
        tq = blk * WAVES + wave;
        if (tq >= a.n_tasks) return;
        d = a.rdesc[a.task_slot[tq]];
        e0 = a.task_chunk[tq] * CHUNK;
    } else {
        const int slot = a.n_split + blk * WAVES + wave;
        if (slot >= a.n_med_end) return;
        d = a.rdesc[slot];
    }
    const int i = d.x, rs = d.y, deg = d.z;
    const int n = task ? min(deg - e0, CHUNK) : deg;
    int *s_j = lds_wave;
    float *s_a = reinterpret_cast<float *>(lds_wave + WAVE_T);
    float *s_s = reinterpret_cast<float *>(lds_wave + 2 * WAVE_T);
    float *s_t = reinterpret_cast<float *>(lds_wave + 3 * WAVE_T);
    for (int t = lane; t < n; t += 64) {
        const float se = a.wsel[rs + e0 + t];
        s_j[t] = a.col[rs + e0 + t];
        s_s[t] = se;
        s_a[t] = x.coef[rs + e0 + t] * signed_kappa(se, cp, cn);
    }
    RowT gp, acc;
    gp.load(a.gout + (size_t)i * a.C, a.C, lg);
    acc.zero();
    wave_lds_sync();
    for (int q0 = 0; q0 < n; q0 += 2 * NG) {
        const int qa = min(q0 + gid, n - 1), qb = min(q0 + NG + gid, n - 1);
        const bool la = q0 + gid < n, lb = q0 + NG + gid < n;
        RowT xa, xb;
        xa.load(a.h + (size_t)s_j[qa] * a.C, a.C, lg);
        xb.load(a.h + (size_t)s_j[qb] * a.C, a.C, lg);
        const float ta = signed_t_edge<VEC, G, R>(xa, gp, la ? s_a[qa] : 0.f, acc);
        const float tb = signed_t_edge<VEC, G, R>(xb, gp, lb ? s_a[qb] : 0.f, acc);
        if (lg == 0 && la) s_t[qa] = ta;
        if (lg == 0 && lb) s_t[qb] = tb;
    }
    acc.reduce_across_groups();
    if (gid == 0) acc.store(task ? a.partT + (size_t)tq * a.C : a.dnT + (size_t)i * a.C, a.C, lg);
    wave_lds_sync();
    for (int t = lane; t < n; t += 64) {
        a.wd[a.csc_pos[rs + e0 + t]] = make_float2(s_a[t] * s_s[t], s_a[t] * s_t[t]);
        x.u[rs + e0 + t] = s_s[t] * s_t[t];
    }
    wave_lds_sync();
}

template <int VEC, int G, int R>
__global__ __launch_bounds__(BLOCK) void k_signed_bwd_t(const BwdArgs a, const SignedBwdExtra x)
{
    __shared__ __align__(16) int lds[WAVES][4 * WAVE_T];
    const int b = blockIdx.x;
    int *lw = lds[threadIdx.x >> 6];
    const float cp = x.c2[0], cn = x.c2[1];
    if (b < a.nbA) signed_t_wave<VEC, G, R>(a, x, b, lw, true, cp, cn);
    else if (b < a.nbA + a.nbB) signed_t_wave<VEC, G, R>(a, x, b - a.nbA, lw, false, cp, cn);
    else signed_t_small<VEC, G, R>(a, x, b - a.nbA - a.nbB, lw, cp, cn);
}

template <int VEC, int G, int R> int launch_signed_bwd(const BwdArgs &a0, const SignedBwdExtra &x, hipStream_t st)
{
    constexpr int RPW = 64 / G;
    BwdArgs a = a0;
    a.nbA = ceil_div(a.n_tasks, WAVES);
    a.nbB = ceil_div(a.n_med_end - a.n_split, WAVES);
    int nbC = ceil_div(a.N - a.n_med_end, (int64_t)WAVES * RPW);
    if (a.nbA + a.nbB + nbC > 0) k_signed_bwd_t<VEC, G, R><<<a.nbA + a.nbB + nbC, BLOCK, 0, st>>>(a, x);
    if (a.n_split > 0) k_bwd_t_fin<<<a.n_split, 256, 0, st>>>(a);      // dnT of the split rows: partT rows of C floats
    // pass S: the aggregation's kernels on the records (every edge carries one; no mean division)
    a.nbA = ceil_div(a.n_stasks, WAVES);
    a.nbB = ceil_div(a.n_smed_end - a.n_ssplit, WAVES);
    nbC = ceil_div(a.Ntot - a.n_smed_end, (int64_t)WAVES * RPW);
    if (a.nbA + a.nbB + nbC > 0) k_bwd_s<VEC, G, R, true><<<a.nbA + a.nbB + nbC, BLOCK, 0, st>>>(a);
    if (a.n_ssplit > 0) k_bwd_s_fin<VEC, G, R><<<a.n_ssplit, 64, 0, st>>>(a);
    SN_HIP(hipGetLastError());
    return SNGNN_OK;
}

int launch_signed_fwd_v1(const RowCfg &cfg, const SignedArgs &a, hipStream_t st);
int launch_signed_fwd_v2(const RowCfg &cfg, const SignedArgs &a, hipStream_t st);
int launch_signed_fwd_v4(const RowCfg &cfg, const SignedArgs &a, hipStream_t st);
int launch_signed_bwd_v1(const RowCfg &cfg, const BwdArgs &a, const SignedBwdExtra &x, hipStream_t st);
int launch_signed_bwd_v2(const RowCfg &cfg, const BwdArgs &a, const SignedBwdExtra &x, hipStream_t st);
int launch_signed_bwd_v4(const RowCfg &cfg, const BwdArgs &a, const SignedBwdExtra &x, hipStream_t st);

}  // namespace sngnn
