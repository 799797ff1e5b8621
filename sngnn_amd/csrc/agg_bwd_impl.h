// Backward of the fused similarity-navigated aggregation (gfx950).
//
// Replaces autograd through the reference's op list (models/models.py:122,132,
// 139-158 / 238-239,244-263 / 325-326,331-334; triggered at train.py:86):
// gradients reach h through the message value x_j, and through norm_i and norm_j
// of every KEPT edge (weight[idx] = norm[idx], models.py:156,261); the selection
// itself is not differentiable.
//
// With g'_i = G_i / max(deg_i,1), n_v = h_v * inv_v and, for a kept edge
// e = (j -> i) with cosine w_e:
//     ds_e  = <g'_i, h_j>
//     dnT_i = sum_{e into i}  ds_e * n_j          (pass T, CSR rows, by target)
//     msg_j = sum_{e out of j} w_e * g'_i          (pass S, CSC rows, by source)
//     dnS_j = sum_{e out of j} ds_e * n_i
//     dn_v  = dnT_v + dnS_v
//     dh_v  = msg_v + (dn_v - n_v <n_v, dn_v>) * inv_v        (F.normalize's Jacobian;
//             without the projection where the eps clamp is active)
// Both passes are gathers with a fixed summation order: no floating-point
// atomics, bitwise reproducible.  Row classes by degree as in the forward.
// What pass S needs to know about an out-edge is one bit - kept or not: w_e and ds_e are dot
// products of rows it gathers anyway (n_i, G_i) with its own row, so it recomputes them
// (ds_e = <G_i, h_j> / deg_i, w_e = <h_i, h_j> inv_i inv_j).  Pass T sets the bit of every KEPT
// edge in a bitmask in CSC order (graph.csc_pos; one fire-and-forget atomic OR per kept edge
// into a 145 KB, cache-resident mask at arxiv size) and pass S reads the bits of a source's
// out-edges as one coalesced word stream.  (Round 2 first had pass T write an 8-byte record
// {w_e / deg_i, ds_e} per EDGE at its CSC position: 1.16 M scattered sector writes, 14 us of
// pass T's 44.6; the attention mode, whose per-edge weight is not recomputable from two rows,
// still does - REC = true below.)
#pragma once
#include "device_utils.h"

#ifndef SNGNN_BWDF_STEP
#define SNGNN_BWDF_STEP 1          // kept in-edges and kept out-edges per step of a fused node (2: 74.9 us against 73.3)
#endif

namespace sngnn {

struct BwdArgs {
    const float *h, *gout, *wsel;
    int C, N, Ntot, row_off;      // N owned target rows; Ntot sources / rows of h, grad_h
    const int32_t *rowptr, *col, *rperm;
    const int4 *rdesc, *sdesc;    // per degree-sorted slot: {node, first entry, degree, 0}: one load, not a chain of three
    const int32_t *cscptr, *csc_eid, *csc_dst, *csc_pos, *sperm;
    float2 *wd;                   // [E'] per-edge record in CSC order (attention mode: attn_impl.h)
    float *rec_dot;               // attention mode: [N] dot_i = sum_e alpha_e t_e of the SPLIT targets (their finalize
                                  // writes it).  A record of an edge into a split row is {-alpha_e, t_e} - the sign
                                  // marks it raw, alpha > 0 - and pass S forms ds_e = alpha_e (t_e - dot_i) itself,
                                  // from a 4-byte read that travels with its row gathers (825 distinct addresses at
                                  // arxiv size: cache-resident; the other edges read entry 0).  Every other record,
                                  // and every record when this is nullptr (signed mode), is the final {w, ds}
    unsigned *kmask;              // [kmask_words = ceil(E'/32)] kept bits: two-pass mode in CSC order (set by pass T, read
                                  // by pass S); node-centric mode in CSR order (k_pack_kept), read by both parts
    int64_t kmask_words, Ep;
    const float *inv_deg;         // [N] 1 / max(in-degree, 1)
    float *dnT, *grad_h;
    int n_split, n_med_end, n_tasks;
    const int32_t *task_slot, *task_chunk, *split_task0;
    float *partT;
    int n_ssplit, n_smed_end, n_stasks;
    const int32_t *stask_slot, *stask_chunk, *ssplit_task0;
    float *partS;
    int nbA, nbB;
    int s_small_end;              // pass S walks the small sources sdesc[n_smed_end, s_small_end)
    // node-centric path: nodes small both as target and as source do both passes in one work item
    const int4 *fdesc;            // [n_fused] {node, first in-edge, first CSC entry, in-degree | out-degree << 8}
    const int4 *trest;            // [n_trest] small targets that are not fused (out-degree > SMALL_T): rdesc layout
    int n_fused, n_trest, nbC;
    int mode;                     // 0 = node-centric (default), 1 = the two passes for every node
    int role_mask;                // measurement aid (sngnn_tuning_set knob 4): bit 0 wave-per-node items, bit 1 fused items
    int top_k;                    // the forward's top_k when the caller gave it (at most that many kept in-edges
                                  // per row), else <= 0
    // kept bits written by the forward itself (common.h "kbits" layout; agg_fwd_impl.h): no k_pack_kept
    // launch.  Small rows: halfword at the row id; wave rows / tasks: 128-bit blocks; out-edges by csc_bit.
    const unsigned *kbits;
    const int32_t *csc_bit;
    int kb_wbase, kb_tbase;
};

__device__ __forceinline__ bool kbit(const BwdArgs &a, int b) { return (a.kbits[b >> 5] >> (b & 31)) & 1u; }

__device__ __forceinline__ bool is_kept(float w) { return w > -3.0f; }

// the G bits of a wave ballot that belong to lane group gid
template <int G> __device__ __forceinline__ unsigned long long group_bits(unsigned long long m, int gid)
{
    if constexpr (G == 64) return m;
    else return (m >> (gid * G)) & ((1ull << G) - 1ull);
}

template <int VEC, int G, int R>
__device__ __forceinline__ void fma_row(Row<VEC, G, R> &acc, float w, const Row<VEC, G, R> &x)
{
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc.x[r][v] = fmaf(w, x.x[r][v], acc.x[r][v]);
}

// ------------------------------- pass T ------------------------------------
// one kept in-edge of target i (source row x): ds_e and its contribution to dnT_i
template <int VEC, int G, int R>
__device__ __forceinline__ void t_edge_row(const Row<VEC, G, R> &x, const Row<VEC, G, R> &gp,
                                           Row<VEC, G, R> &acc, float live = 1.0f)
{
    // live = 0: a padding repeat of the previous edge (odd count) adds nothing; keeping it
    // unconditional keeps its row load in flight with its partner's instead of behind a branch
    // (rsq + one Newton step, about half an ulp: the backward only needs the value to be accurate;
    // the forward's selection uses IEEE sqrt / division on whole rows, agg_fwd_impl.h)
    const float invj = inv_norm_of(group_sum<G>(x.dot_partial(x)));
    const float d = group_sum<G>(gp.dot_partial(x));
    fma_row<VEC, G, R>(acc, d * invj * live, x);
}

// node-centric mode: kept bit of CSR edge e (the mask is 145 KB at arxiv size: cache resident)
__device__ __forceinline__ bool kept_csr(const BwdArgs &a, int e) { return (a.kmask[e >> 5] >> (e & 31)) & 1u; }

// kept bit of the out-edge at CSC position q: KB = the forward-written layout, else k_pack_kept's CSR-order mask
template <bool KB> __device__ __forceinline__ bool kept_out(const BwdArgs &a, int q)
{
    if constexpr (KB) return kbit(a, a.csc_bit[q]);
    else return kept_csr(a, a.csc_eid[q]);
}

__device__ __forceinline__ void set_kept_bit(const BwdArgs &a, int cp)
{
    atomicOr(a.kmask + (cp >> 5), 1u << (cp & 31));        // result unused: no return trip
}

// kept edges among [e0, e1) of a CSR row, compacted (ascending): their source ids into
// jlist[]; their bits in the CSC-order mask are set here.  Lanes cover the range 64 at a time.
// (MARK = false, node-centric mode: the kept bits are read from k_pack_kept's mask, nothing to set)
template <bool MARK>
__device__ __forceinline__ int kept_list(const BwdArgs &a, int rs, int e0, int e1, int *jlist)
{
    const int lane = lane_id();
    int n = 0;
    for (int base = e0; base < e1; base += 64) {
        const int t = base + lane;
        bool kept;
        int cp = 0;
        if constexpr (MARK) {
            const float w = t < e1 ? a.wsel[rs + t] : SNGNN_UNSELECTED;
            cp = t < e1 ? a.csc_pos[rs + t] : 0;
            kept = is_kept(w);
        } else {
            kept = t < e1 && kept_csr(a, rs + t);
        }
        // (with the weight, not behind `if (kept)`: a load inside the branch is a round trip of its own)
        const int j = t < e1 ? a.col[rs + t] : 0;
        const unsigned long long m = __ballot(kept);
        if (kept) {
            jlist[n + prefix_popc(m)] = j;
            if constexpr (MARK) set_kept_bit(a, cp);
        }
        n += __popcll(m);
    }
    return n;
}

// small targets (deg <= SMALL_T), one group per row.  The group's lanes first fetch the
// kept flags and source ids of all its edges in parallel (one round trip), then the
// kept source rows are gathered two at a time.
template <int VEC, int G, int R, bool MARK>
__device__ __forceinline__ void t_role_small(const BwdArgs &a, int blk, int *lds_wave, const int4 *desc, int n_desc)
{
    using RowT = Row<VEC, G, R>;
    constexpr int RPW = 64 / G;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int gid = lane / G, lg = lane % G;
    const int slot = (blk * WAVES + wave) * RPW + gid;
    if (slot >= n_desc) return;                  // group-uniform
    const int4 d = desc[slot];
    const int i = d.x, rs = d.y, deg = d.z;
    // The kept edges are compacted first (ballot within the group), so the gather loop below
    // has no `if (kept)` around its loads: a load inside a divergent branch is waited for
    // right there (s_waitcnt vmcnt(0) before the branch closes), which would serialise the
    // rows meant to be in flight together.
    int *s_j = lds_wave + gid * SMALL_T;                     // [SMALL_T] kept: source id
    const float invdeg = 1.0f / (float)max(deg, 1);
    int nk = 0;
    for (int t0 = 0; t0 < deg; t0 += G) {
        const int t = t0 + lg;
        bool kept;
        int cp = 0;
        if constexpr (MARK) {
            const float w = t < deg ? a.wsel[rs + t] : SNGNN_UNSELECTED;
            cp = t < deg ? a.csc_pos[rs + t] : 0;
            kept = is_kept(w);
        } else {
            kept = t < deg && kept_csr(a, rs + t);
        }
        const int j = t < deg ? a.col[rs + t] : 0;       // with the weight: not a round trip of its own
        const unsigned long long gm = group_bits<G>(__ballot(kept), gid);
        if (kept) {
            s_j[nk + __popcll(gm & ((1ull << lg) - 1ull))] = j;
            if constexpr (MARK) set_kept_bit(a, cp);
        }
        nk += __popcll(gm);
    }
    RowT gp, acc;
    gp.load(a.gout + (size_t)i * a.C, a.C, lg);
    gp.scale(invdeg);
    acc.zero();
    wave_lds_sync();
    for (int q0 = 0; q0 < nk; q0 += 2) {
        const bool two = q0 + 1 < nk;
        const int q1 = two ? q0 + 1 : q0;
        RowT x0, x1;
        x0.load(a.h + (size_t)s_j[q0] * a.C, a.C, lg);
        x1.load(a.h + (size_t)s_j[q1] * a.C, a.C, lg);
        t_edge_row<VEC, G, R>(x0, gp, acc);
        t_edge_row<VEC, G, R>(x1, gp, acc, two ? 1.0f : 0.0f);
    }
    acc.store(a.dnT + (size_t)i * a.C, a.C, lg);
}

template <int VEC, int G, int R, bool MARK>
__device__ __forceinline__ void t_role_wave(const BwdArgs &a, int blk, int *lds_wave, bool task)
{
    using RowT = Row<VEC, G, R>;
    constexpr int NG = 64 / G;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int gid = lane / G, lg = lane % G;
    int i, e0, e1, tq = 0;
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
        e0 = 0;
    }
    i = d.x;
    const int rs = d.y, deg = d.z;
    e1 = task ? min(deg, e0 + CHUNK) : deg;
    RowT gp, acc;
    const float invdeg = 1.0f / (float)deg;
    gp.load(a.gout + (size_t)i * a.C, a.C, lg);
    gp.scale(invdeg);
    acc.zero();
    int *jlist = lds_wave;
    const int nsel = kept_list<MARK>(a, rs, e0, e1, jlist);
    wave_lds_sync();
    // two kept rows per lane group in flight, unconditionally (a slot past the end repeats the
    // last kept edge with live = 0): no col -> row chain, no load behind a branch.  Matters on
    // graphs whose rows mostly have 17..128 in-edges (products-like), not at arxiv size.
    for (int q0 = 0; q0 < nsel; q0 += 2 * NG) {
        const int qa = min(q0 + gid, nsel - 1), qb = min(q0 + NG + gid, nsel - 1);
        RowT xa, xb;
        xa.load(a.h + (size_t)jlist[qa] * a.C, a.C, lg);
        xb.load(a.h + (size_t)jlist[qb] * a.C, a.C, lg);
        t_edge_row<VEC, G, R>(xa, gp, acc, q0 + gid < nsel ? 1.0f : 0.0f);
        t_edge_row<VEC, G, R>(xb, gp, acc, q0 + NG + gid < nsel ? 1.0f : 0.0f);
    }
    acc.reduce_across_groups();
    if (gid == 0) {
        float *dst = task ? a.partT + (size_t)tq * a.C : a.dnT + (size_t)i * a.C;
        acc.store(dst, a.C, lg);
    }
}

template <int VEC, int G, int R>
__global__ __launch_bounds__(BLOCK) void k_bwd_t(const BwdArgs a)
{
    __shared__ __align__(16) int lds[WAVES][WAVE_T];
    const int b = blockIdx.x;
    int *lw = lds[threadIdx.x >> 6];
    if (b < a.nbA) t_role_wave<VEC, G, R, true>(a, b, lw, true);
    else if (b < a.nbA + a.nbB) t_role_wave<VEC, G, R, true>(a, b - a.nbA, lw, false);
    else t_role_small<VEC, G, R, true>(a, b - a.nbA - a.nbB, lw, a.rdesc + a.n_med_end, a.N - a.n_med_end);
}

// split targets: dnT_i = sum of the tasks' partial rows.  Thread (c, q) adds every 4th
// task (four independent chains, so the loads of a hub's ~100 tasks overlap); the sums are
// combined in fixed order (deterministic).
static __global__ __launch_bounds__(256) void k_bwd_t_fin(const BwdArgs a)
{
    __shared__ float s[4][64];
    const int p = blockIdx.x;
    const int i = a.rperm[p];
    const int t0 = a.split_task0[p], t1 = a.split_task0[p + 1];
    const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
    for (int c0 = 0; c0 < a.C; c0 += 64) {
        const int c = c0 + cl;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        if (c < a.C) {
            int t = t0 + q;
            for (; t + 12 < t1; t += 16) {
                v0 += a.partT[(size_t)t * a.C + c];
                v1 += a.partT[(size_t)(t + 4) * a.C + c];
                v2 += a.partT[(size_t)(t + 8) * a.C + c];
                v3 += a.partT[(size_t)(t + 12) * a.C + c];
            }
            for (; t < t1; t += 4) v0 += a.partT[(size_t)t * a.C + c];
        }
        s[q][cl] = (v0 + v1) + (v2 + v3);
        __syncthreads();
        if (q == 0 && c < a.C) a.dnT[(size_t)i * a.C + c] = (s[0][cl] + s[1][cl]) + (s[2][cl] + s[3][cl]);
        __syncthreads();
    }
}

// ------------------------------- pass S ------------------------------------
// one kept out-edge of source v: target row i (local), record {w / deg_i, ds_e}
template <int VEC, int G, int R>
__device__ __forceinline__ void s_edge(const BwdArgs &a, int i, float2 rec, int lg,
                                       Row<VEC, G, R> &msg, Row<VEC, G, R> &dns)
{
    Row<VEC, G, R> x, gi;
    x.load(a.h + (size_t)(i + a.row_off) * a.C, a.C, lg);
    gi.load(a.gout + (size_t)i * a.C, a.C, lg);
    const float invi = inv_norm_of(group_sum<G>(x.dot_partial(x)));
    fma_row<VEC, G, R>(msg, rec.x, gi);
    fma_row<VEC, G, R>(dns, rec.y * invi, x);
}

// the rows s_finish needs, requested at the START of an item so that they travel with the
// item's first loads instead of adding a round trip at its end
template <int VEC, int G, int R> struct FinishRows {
    Row<VEC, G, R> t, hv;
    bool own;
    __device__ __forceinline__ void load(const BwdArgs &a, int v, int lg)
    {
        const int vl = v - a.row_off;                 // owned nodes also carry a target part
        own = vl >= 0 && vl < a.N;
        t.load(a.dnT + (size_t)(own ? vl : 0) * a.C, a.C, lg);
        hv.load(a.h + (size_t)v * a.C, a.C, lg);
    }
};

// dh_v from msg_v and dn_v = dnT_v + dnS_v
template <int VEC, int G, int R>
__device__ __forceinline__ void s_finish(const BwdArgs &a, int v, int lg, Row<VEC, G, R> &msg,
                                         Row<VEC, G, R> &dn, FinishRows<VEC, G, R> &f)
{
    if (f.own) dn.add(f.t);
    Row<VEC, G, R> &hv = f.hv;
    const float qv = group_sum<G>(hv.dot_partial(hv));
    const float invv = inv_norm_of(qv);
    hv.scale(invv);                               // n_v
    float proj = group_sum<G>(hv.dot_partial(dn));
    if (norm_clamped(qv)) proj = 0.f;             // eps clamp active: n = h / eps
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < VEC; ++c)
            msg.x[r][c] += (dn.x[r][c] - hv.x[r][c] * proj) * invv;
    msg.store(a.grad_h + (size_t)v * a.C, a.C, lg);
}

// one kept out-edge (v -> i) from the rows alone: x = h_i, gi = G_i, hv = h_v (own, raw),
// invv = inv_v, invdeg = 1 / deg_i;  w_e = <h_i, h_v> inv_i inv_v,  ds_e = <G_i, h_v> / deg_i
template <int VEC, int G, int R>
__device__ __forceinline__ void s_edge_recompute(const Row<VEC, G, R> &x, const Row<VEC, G, R> &gi,
                                                 const Row<VEC, G, R> &hv, float invv, float invdeg,
                                                 float live, Row<VEC, G, R> &msg, Row<VEC, G, R> &dns)
{
    const float invi = inv_norm_of(group_sum<G>(x.dot_partial(x)));
    const float w = group_sum<G>(x.dot_partial(hv)) * (invi * invv);
    const float dse = group_sum<G>(gi.dot_partial(hv)) * invdeg;
    fma_row<VEC, G, R>(msg, w * invdeg * live, gi);
    fma_row<VEC, G, R>(dns, dse * invi * live, x);
}

// the same with both rows already loaded and the per-edge scalars from a record
template <int VEC, int G, int R>
__device__ __forceinline__ void s_edge_rows(const Row<VEC, G, R> &x, const Row<VEC, G, R> &gi,
                                            float w, float dse, Row<VEC, G, R> &msg,
                                            Row<VEC, G, R> &dns)
{
    const float invi = inv_norm_of(group_sum<G>(x.dot_partial(x)));
    fma_row<VEC, G, R>(msg, w, gi);
    fma_row<VEC, G, R>(dns, dse * invi, x);
}

// small sources (out-degree <= SMALL_T), one group per source: per-edge scalars of all
// out-edges are fetched by the group's lanes in parallel, then the (h_i, G_i) row pairs
// of the kept edges are gathered two edges at a time.
//   REC = true:  per-edge scalars from the records a.wd (attention mode)
//   REC = false: kept bits from a.kmask, scalars recomputed from the rows
//   CSRM (with REC = false): the mask is in CSR order (node-centric mode): bit of csc_eid[q]
template <int VEC, int G, int R, bool REC, bool CSRM>
__device__ __forceinline__ void s_role_small(const BwdArgs &a, int blk, int *lds_wave)
{
    using RowT = Row<VEC, G, R>;
    constexpr int RPW = 64 / G;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int gid = lane / G, lg = lane % G;
    const int slot = a.n_smed_end + (blk * WAVES + wave) * RPW + gid;
    if (slot >= a.s_small_end) return;
    const int4 d = a.sdesc[slot];
    const int v = d.x, qs = d.y, od = d.z;
    FinishRows<VEC, G, R> fin;
    fin.load(a, v, lg);
    int *s_i = lds_wave + gid * 3 * SMALL_T;                          // kept out-edges: target row
    float *s_w = reinterpret_cast<float *>(s_i + SMALL_T);
    float *s_ds = reinterpret_cast<float *>(s_i + 2 * SMALL_T);
    int nk = 0;                                                       // (compacted: see t_role_small)
    for (int t0 = 0; t0 < od; t0 += G) {
        const int t = t0 + lg;
        float2 rec = make_float2(SNGNN_UNSELECTED, 0.f);
        int it = 0;
        bool kept = false;
        if (t < od) {                                                 // together: one round trip
            const int q = qs + t;
            it = a.csc_dst[q];
            if constexpr (REC) rec = a.wd[q];
            else if constexpr (CSRM) kept = a.kbits ? kbit(a, a.csc_bit[q]) : kept_csr(a, a.csc_eid[q]);
            else kept = (a.kmask[q >> 5] >> (q & 31)) & 1u;
        }
        // (REC: every out-edge carries a record - softmax terms in the attention mode, signed
        // weights of either sign and any size in the signed mode, signed_impl.h)
        if constexpr (REC) kept = t < od;
        const unsigned long long gm = group_bits<G>(__ballot(kept), gid);
        if (kept) {
            const int pos = nk + __popcll(gm & ((1ull << lg) - 1ull));
            s_i[pos] = it;
            if constexpr (REC) { s_w[pos] = rec.x; s_ds[pos] = rec.y; }
        }
        nk += __popcll(gm);
    }
    RowT msg, dns;
    msg.zero();
    dns.zero();
    float invv = 0.f;
    if constexpr (!REC) invv = inv_norm_of(group_sum<G>(fin.hv.dot_partial(fin.hv)));
    wave_lds_sync();
    for (int q0 = 0; q0 < nk; q0 += 2) {
        const bool two = q0 + 1 < nk;
        const int q1 = two ? q0 + 1 : q0;
        const int i0 = s_i[q0], i1 = s_i[q1];
        RowT x0, g0, x1, g1;
        x0.load(a.h + (size_t)(i0 + a.row_off) * a.C, a.C, lg);
        g0.load(a.gout + (size_t)i0 * a.C, a.C, lg);
        x1.load(a.h + (size_t)(i1 + a.row_off) * a.C, a.C, lg);
        g1.load(a.gout + (size_t)i1 * a.C, a.C, lg);
        // (odd count: the repeat enters with zero weights - unconditional, so that all four row
        // loads are in flight together)
        if constexpr (REC) {
            float w0 = s_w[q0], w1 = s_w[q1], ds0 = s_ds[q0], ds1 = s_ds[q1];
            if (a.rec_dot) {                                      // (uniform) raw records: BwdArgs::rec_dot
                // (the SIGN BIT is the mark: an alpha that underflowed to 0 is stored as -0.0f, which `< 0.f` misses)
                const bool r0 = (__float_as_uint(w0) >> 31) != 0, r1 = (__float_as_uint(w1) >> 31) != 0;
                const float t0_ = a.rec_dot[r0 ? i0 : 0], t1_ = a.rec_dot[r1 ? i1 : 0];   // unconditional loads
                w0 = fabsf(w0); w1 = fabsf(w1);
                ds0 = r0 ? w0 * (ds0 - t0_) : ds0;
                ds1 = r1 ? w1 * (ds1 - t1_) : ds1;
            }
            s_edge_rows<VEC, G, R>(x0, g0, w0, ds0, msg, dns);
            s_edge_rows<VEC, G, R>(x1, g1, two ? w1 : 0.f, two ? ds1 : 0.f, msg, dns);
        } else {
            const float d0 = a.inv_deg[(a.role_mask & 4) ? 0 : i0], d1 = a.inv_deg[(a.role_mask & 4) ? 0 : i1];       // travel with the rows (bit 2 of the role mask: timing experiment - one address)
            s_edge_recompute<VEC, G, R>(x0, g0, fin.hv, invv, d0, 1.0f, msg, dns);
            s_edge_recompute<VEC, G, R>(x1, g1, fin.hv, invv, d1, two ? 1.0f : 0.0f, msg, dns);
        }
    }
    s_finish<VEC, G, R>(a, v, lg, msg, dns, fin);
}

// ------------------------- node-centric work item ---------------------------
// A node that is small both as a target (in-degree <= SMALL_T) and as a source (out-degree <=
// SMALL_T) - 95 % of the nodes at arxiv size - does its pass-T part and its pass-S part in one
// work item of one lane group: its dnT row never leaves the registers (no store in pass T, no
// load in pass S), its descriptor, own rows and lists are fetched once, and the kept in-edge
// rows (h_j) and kept out-edge row pairs (h_i, G_i) are gathered in ONE loop - at ~1.2 kept
// in-edges and ~1.9 kept out-edges per node the two passes were three dependent round trips
// each, mostly for one gather step.  The sums run in the two passes' order (kept edges
// ascending, T then S), so the result equals theirs bit for bit.
// The kept bits come from k_pack_kept's mask in CSR order (below): in-edges by position, out-edges
// through csc_eid.
template <int VEC, int G, int R, bool KB = false>
__device__ __forceinline__ void f_role_node(const BwdArgs &a, int blk, int *lds_wave)
{
    using RowT = Row<VEC, G, R>;
    constexpr int RPW = 64 / G;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int gid = lane / G, lg = lane % G;
    const int slot = (blk * WAVES + wave) * RPW + gid;
    if (slot >= a.n_fused) return;               // group-uniform
    const int4 d = a.fdesc[slot];
    const int v = d.x, rs = d.y, qs = d.z, deg = d.w & 255, od = d.w >> 8;
    const int vl = v - a.row_off;
    int *s_j = lds_wave + gid * 2 * SMALL_T;     // kept in-edges: source id
    int *s_i = s_j + SMALL_T;                    // kept out-edges: target row
    FinishRows<VEC, G, R> fin;                   // .t becomes dnT_v, .hv = h_v (raw)
    RowT gp;
    gp.load(a.gout + (size_t)vl * a.C, a.C, lg);
    fin.hv.load(a.h + (size_t)v * a.C, a.C, lg);
    fin.own = true;
    int nk = 0, nko = 0;
    unsigned hw = 0u;                                        // KB: the row's own 16 kept bits, one 2-byte load
    if constexpr (KB) hw = reinterpret_cast<const unsigned short *>(a.kbits)[vl];
    for (int t0 = 0; t0 < max(deg, od); t0 += G) {           // (one trip for G >= 16)
        const int t = t0 + lg;
        bool kept_i;
        if constexpr (KB) kept_i = t < deg && ((hw >> t) & 1u);
        else kept_i = t < deg && kept_csr(a, rs + t);
        const int j = t < deg ? a.col[rs + t] : 0;
        bool kept_o = false;
        int it = 0;
        if (t < od) {
            it = a.csc_dst[qs + t];
            kept_o = kept_out<KB>(a, qs + t);                // index -> bit: the one chained load of the item
        }
        const unsigned long long gi = group_bits<G>(__ballot(kept_i), gid);
        const unsigned long long go = group_bits<G>(__ballot(kept_o), gid);
        const unsigned long long below = (1ull << lg) - 1ull;
        if (kept_i) s_j[nk + __popcll(gi & below)] = j;
        if (kept_o) s_i[nko + __popcll(go & below)] = it;
        nk += __popcll(gi);
        nko += __popcll(go);
    }
    gp.scale(1.0f / (float)max(deg, 1));
    const float invv = inv_norm_of(group_sum<G>(fin.hv.dot_partial(fin.hv)));
    RowT msg, dns;
    fin.t.zero();
    msg.zero();
    dns.zero();
    wave_lds_sync();
    // one step = FU kept in-edges and FU kept out-edges: 3 FU rows in flight, unconditionally
    // (a slot past a list's end repeats the node's own row with zero weight)
    constexpr int FU = SNGNN_BWDF_STEP;
    const int steps = (max(nk, nko) + FU - 1) / FU;
    for (int s = 0; s < steps; ++s) {
        RowT xj[FU], x[FU], gi[FU];
        float dd[FU];
        bool ti[FU], to[FU];
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int q = FU * s + u;
            ti[u] = q < nk;
            to[u] = q < nko;
            const int j = ti[u] ? s_j[q] : v;
            const int i = to[u] ? s_i[q] : vl;
            xj[u].load(a.h + (size_t)j * a.C, a.C, lg);
            x[u].load(a.h + (size_t)(i + a.row_off) * a.C, a.C, lg);
            gi[u].load(a.gout + (size_t)i * a.C, a.C, lg);
            dd[u] = a.inv_deg[i];                                        // travels with the rows
        }
#pragma unroll
        for (int u = 0; u < FU; ++u) t_edge_row<VEC, G, R>(xj[u], gp, fin.t, ti[u] ? 1.0f : 0.0f);
#pragma unroll
        for (int u = 0; u < FU; ++u)
            s_edge_recompute<VEC, G, R>(x[u], gi[u], fin.hv, invv, dd[u], to[u] ? 1.0f : 0.0f, msg, dns);
    }
    s_finish<VEC, G, R>(a, v, lg, msg, dns, fin);
}

// inclusive prefix sum over the lanes of a wave
__device__ __forceinline__ int wave_prefix_incl(int v)
{
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// ---------------------- wave-per-node work item (selective calls) ------------------------
// When the forward kept at most top_k <= WAVE_T in-edges per row (the caller says so:
// sngnn_agg_backward_topk), a target of ANY in-degree is one wave's work: the kept in-edges are
// found by scanning the row's bits in the packed mask (2048 edges per wave-wide load - 7 loads
// for a 13 000-edge hub, where the task form reads 52 KB of flags and writes ~100 partial
// rows for 16 kept edges), so there are no split-row tasks, no partial rows and no k_bwd_t_fin;
// and the wave goes on with the node's pass-S part (out-degree <= WAVE_T) with dnT in registers,
// so pass S has nothing left but split sources and, under a partition, the halo's sources.
// Robust against a wrong hint: a list that would overflow is flushed (gathered and
// accumulated) first - slower, still correct and deterministic.
template <int VEC, int G, int R, bool KB = false>
__device__ __forceinline__ void w_role_node(const BwdArgs &a, int blk, int *lds_wave)
{
    using RowT = Row<VEC, G, R>;
    constexpr int NG = 64 / G;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int gid = lane / G, lg = lane % G;
    const int slot = blk * WAVES + wave;
    if (slot >= a.n_med_end + a.n_trest) return;
    const int4 d = slot < a.n_med_end ? a.rdesc[slot] : a.trest[slot - a.n_med_end];
    const int vl = d.x, rs = d.y, deg = d.z;
    const int v = vl + a.row_off;
    const int qs = a.cscptr[v], od = a.cscptr[v + 1] - qs;
    FinishRows<VEC, G, R> fin;                   // .t becomes dnT_v
    RowT gp;
    gp.load(a.gout + (size_t)vl * a.C, a.C, lg);
    fin.hv.load(a.h + (size_t)v * a.C, a.C, lg);
    fin.own = true;
    gp.scale(1.0f / (float)max(deg, 1));
    fin.t.zero();
    int *elist = lds_wave;                       // [WAVE_T] kept in-edges: CSR index, then source id
    int *s_i = lds_wave + WAVE_T;                // [WAVE_T] kept out-edges: target row
    int n = 0;
    // gather the n listed source rows into fin.t (per-group partial sums), two per group in flight
    bool have_ids = false;                       // the list already holds source ids (rows <= WAVE_T)
    auto flush = [&]() {
        wave_lds_sync();
        if (!have_ids) {
            for (int t = lane; t < n; t += 64) elist[t] = a.col[(KB ? rs : 0) + elist[t]];   // (KB: the list holds row-local indices)
            wave_lds_sync();
        }
        for (int q0 = 0; q0 < n; q0 += 2 * NG) {
            const int qa = min(q0 + gid, n - 1), qb = min(q0 + NG + gid, n - 1);
            RowT xa, xb;
            xa.load(a.h + (size_t)elist[qa] * a.C, a.C, lg);
            xb.load(a.h + (size_t)elist[qb] * a.C, a.C, lg);
            t_edge_row<VEC, G, R>(xa, gp, fin.t, q0 + gid < n ? 1.0f : 0.0f);
            t_edge_row<VEC, G, R>(xb, gp, fin.t, q0 + NG + gid < n ? 1.0f : 0.0f);
        }
        wave_lds_sync();
        n = 0;
    };
    // append the set bits of every lane's word (ascending edge order) to the list
    auto append = [&](unsigned word, int wi, int total) {
        if (n + total > WAVE_T) flush();                              // wave-uniform
        const int cnt = __popc(word);
        int o = n + wave_prefix_incl(cnt) - cnt;
        while (word) {
            const int b = __builtin_ctz(word);
            word &= word - 1u;
            elist[o++] = wi * 32 + b;
        }
        n += total;
    };
    const int e_end = rs + deg;
    if (deg <= WAVE_T) {
        // a row one wave covers lane by edge: bits and source ids in one round trip (the word scan
        // below needs a second one for the ids of the edges it found)
        // KB: a wave row's 128 bits sit at its slot, a small target's 16 at its row id
        const int bitbase = !KB ? 0 : (slot < a.n_med_end ? 32 * (a.kb_wbase + 4 * (slot - a.n_split)) : 16 * vl);
        for (int base = 0; base < deg; base += 64) {
            const int t = base + lane;
            bool kept;
            if constexpr (KB) kept = t < deg && kbit(a, bitbase + t);
            else kept = t < deg && kept_csr(a, rs + t);
            const int j = t < deg ? a.col[rs + t] : 0;
            const unsigned long long m = __ballot(kept);
            if (kept) elist[n + prefix_popc(m)] = j;
            n += __popcll(m);
        }
        have_ids = true;
    } else if constexpr (KB) {
        // a split row's bits: 128 per task from its first task's block on (bits beyond the row are zero)
        const unsigned *rowbits = a.kbits + a.kb_tbase + 4 * a.split_task0[slot];
        const int nwords = 4 * (a.split_task0[slot + 1] - a.split_task0[slot]);
        for (int w0 = 0; w0 < nwords; w0 += 64) {
            const int wi = w0 + lane;
            const unsigned word = wi < nwords ? rowbits[wi] : 0u;
            const int total = wave_sum_i(__popc(word));
            if (total <= WAVE_T) append(word, wi, total);
            else
                for (int sb = 0; sb < 16; ++sb) {
                    const unsigned ws = (lane >> 2) == sb ? word : 0u;
                    append(ws, wi, wave_sum_i(__popc(ws)));
                }
        }
    } else {
        for (int w0 = rs >> 5; w0 * 32 < e_end; w0 += 64) {
            const int wi = w0 + lane;
            unsigned word = (int64_t)wi * 32 < e_end ? a.kmask[wi] : 0u;
            if (wi * 32 < rs) word &= ~0u << (rs & 31);                   // bits of the previous row
            if (wi * 32 + 32 > e_end) word &= wi * 32 >= e_end ? 0u : ~0u >> ((32 - (e_end & 31)) & 31);
            const int total = wave_sum_i(__popc(word));
            if (total <= WAVE_T) append(word, wi, total);
            else                                                          // (a hint that was not true)
                for (int sb = 0; sb < 16; ++sb) {
                    const unsigned ws = (lane >> 2) == sb ? word : 0u;    // 4 words = 128 edges at a time
                    append(ws, wi, wave_sum_i(__popc(ws)));
                }
        }
    }
    // kept out-edges (pass-S part, out-degree <= WAVE_T): listed while the in-edge rows travel
    int nso = 0;
    const bool s_here = od <= WAVE_T;
    if (s_here)
        for (int base = 0; base < od; base += 64) {
            const int t = base + lane;
            const int it = t < od ? a.csc_dst[qs + t] : 0;
            const bool kept = t < od && kept_out<KB>(a, qs + t);
            const unsigned long long m = __ballot(kept);
            if (kept) s_i[nso + prefix_popc(m)] = it;
            nso += __popcll(m);
        }
    if (n > 0) flush();
    fin.t.reduce_across_groups();
    if (!s_here) {                               // a split source: pass S (tasks) reads dnT from memory
        if (gid == 0) fin.t.store(a.dnT + (size_t)vl * a.C, a.C, lg);
        return;
    }
    const float invv = inv_norm_of(group_sum<G>(fin.hv.dot_partial(fin.hv)));
    RowT msg, dns;
    msg.zero();
    dns.zero();
    wave_lds_sync();
    for (int q0 = 0; q0 < nso; q0 += 2 * NG) {
        const int qa = min(q0 + gid, nso - 1), qb = min(q0 + NG + gid, nso - 1);
        const bool la = q0 + gid < nso, lb = q0 + NG + gid < nso;
        const int ia = s_i[qa], ib = s_i[qb];
        RowT xa, ga, xb, gb;
        xa.load(a.h + (size_t)(ia + a.row_off) * a.C, a.C, lg);
        ga.load(a.gout + (size_t)ia * a.C, a.C, lg);
        xb.load(a.h + (size_t)(ib + a.row_off) * a.C, a.C, lg);
        gb.load(a.gout + (size_t)ib * a.C, a.C, lg);
        const float da = a.inv_deg[ia], db = a.inv_deg[ib];
        s_edge_recompute<VEC, G, R>(xa, ga, fin.hv, invv, da, la ? 1.0f : 0.0f, msg, dns);
        s_edge_recompute<VEC, G, R>(xb, gb, fin.hv, invv, db, lb ? 1.0f : 0.0f, msg, dns);
    }
    msg.reduce_across_groups();
    dns.reduce_across_groups();
    if (gid == 0) s_finish<VEC, G, R>(a, v, lg, msg, dns, fin);
}

// kept bits of the forward's per-edge weights, in CSR (edge) order: one pass over wsel, whole
// words by ballot - no clear, no atomics.  Every part of the node-centric backward reads kept-ness
// from this mask (E'/8 bytes: cache resident) instead of 4 bytes per edge.  (First version: the
// mask in CSC order, one random 4-byte read of wsel per out-edge: 10 us - line traffic.)
static __global__ __launch_bounds__(256) void k_pack_kept(const float *__restrict__ wsel, int64_t Ep,
                                                          unsigned long long *__restrict__ kmask64)
{
    const int lane = lane_id();
    const int64_t base = (blockIdx.x * (int64_t)WAVES + (threadIdx.x >> 6)) * 256;
    if (base >= Ep) return;
    float w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t e = base + u * 64 + lane;
        w[u] = e < Ep ? wsel[e] : SNGNN_UNSELECTED;
    }
    unsigned long long m[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) m[u] = __ballot(is_kept(w[u]));
    if (lane < 4 && base + lane * 64 < Ep) kmask64[(base >> 6) + lane] = lane == 0 ? m[0] : lane == 1 ? m[1] : lane == 2 ? m[2] : m[3];
}

// Both node-centric kernels are bound by how many gathers the CU keeps in flight: with the
// register budget of 8 waves per SIMD (64 VGPRs; the compiler took 84 / 65 for the 16-byte
// one-step layouts, i.e. 5 / 7 waves) the whole backward went 68.5 -> 60.0 us.  Layouts with
// more than one step per row keep the compiler's own budget (their rows do not fit).
#define SNGNN_BWDF_ATTR __attribute__((amdgpu_waves_per_eu(R == 1 ? 8 : 1, 8)))
// selective calls: every owned node in one launch - a wave per node that is not fused (heavy rows
// first), then the fused nodes
template <int VEC, int G, int R, bool KB = false>
__global__ __launch_bounds__(BLOCK) SNGNN_BWDF_ATTR void k_bwd_w(const BwdArgs a)
{
    __shared__ __align__(16) int lds[WAVES][2 * WAVE_T];
    const int b = blockIdx.x;
    int *lw = lds[threadIdx.x >> 6];
    // The wave-per-node items are chains of dependent round trips with one node per wave, the
    // fused items move four nodes per wave: spread over the grid (every nbB-th workgroup) the
    // chains hide behind the fused items' traffic; in front of them they were 10 us of an idle
    // memory system (62 -> 5x us).
    const int stride = a.nbB;
    const int cw = min(a.nbA, (b + stride - 1) / stride);          // wave-per-node workgroups before b
    if (b % stride == 0 && b / stride < a.nbA) { if (a.role_mask & 1) w_role_node<VEC, G, R, KB>(a, b / stride, lw); }
    else if (a.role_mask & 2) f_role_node<VEC, G, R, KB>(a, b - cw, lw);
}

// pass T of the targets that are not fused (split-row tasks, wave rows, small targets with a
// long out-list) and the fused nodes, in one launch
template <int VEC, int G, int R>
__global__ __launch_bounds__(BLOCK) SNGNN_BWDF_ATTR void k_bwd_f(const BwdArgs a)
{
    __shared__ __align__(16) int lds[WAVES][256];
    const int b = blockIdx.x;
    int *lw = lds[threadIdx.x >> 6];
    if (b < a.nbA) t_role_wave<VEC, G, R, false>(a, b, lw, true);
    else if (b < a.nbA + a.nbB) t_role_wave<VEC, G, R, false>(a, b - a.nbA, lw, false);
    else if (b < a.nbA + a.nbB + a.nbC) t_role_small<VEC, G, R, false>(a, b - a.nbA - a.nbB, lw, a.trest, a.n_trest);
    else f_role_node<VEC, G, R>(a, b - a.nbA - a.nbB - a.nbC, lw);
}

template <int VEC, int G, int R, bool REC, bool CSRM>
__device__ __forceinline__ void s_role_wave(const BwdArgs &a, int blk, int *lds_wave, bool task)
{
    using RowT = Row<VEC, G, R>;
    constexpr int NG = 64 / G;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int gid = lane / G, lg = lane % G;
    int v, e0, tq = 0;
    int4 d;
    if (task) {
        tq = blk * WAVES + wave;
        if (tq >= a.n_stasks) return;
        d = a.sdesc[a.stask_slot[tq]];
        e0 = a.stask_chunk[tq] * CHUNK;
    } else {
        const int slot = a.n_ssplit + blk * WAVES + wave;
        if (slot >= a.n_smed_end) return;
        d = a.sdesc[slot];
        e0 = 0;
    }
    v = d.x;
    const int qs = d.y, od = d.z;
    const int e1 = task ? min(od, e0 + CHUNK) : od;
    RowT hv;                                                      // own row (raw): every group its copy
    float invv = 0.f;
    if constexpr (!REC) hv.load(a.h + (size_t)v * a.C, a.C, lg);
    int *s_i = lds_wave;                                         // [WAVE_T] kept target rows
    float2 *s_rec = reinterpret_cast<float2 *>(lds_wave + WAVE_T);   // their records
    int nsel = 0;
    for (int base = e0; base < e1; base += 64) {
        const int t = base + lane;
        const int q = qs + t;
        float2 rec = make_float2(SNGNN_UNSELECTED, 0.f);
        const int it = t < e1 ? a.csc_dst[q] : 0;                     // with the flags: one round trip
        bool kept;
        if constexpr (REC) {
            if (t < e1) rec = a.wd[q];
            kept = t < e1;
        } else if constexpr (CSRM) {
            kept = t < e1 && (a.kbits ? kbit(a, a.csc_bit[q]) : kept_csr(a, a.csc_eid[q]));
        } else {
            kept = t < e1 && ((a.kmask[q >> 5] >> (q & 31)) & 1u);
        }
        const unsigned long long m = __ballot(kept);
        if (kept) {
            const int o = nsel + prefix_popc(m);
            s_i[o] = it;
            if constexpr (REC) s_rec[o] = rec;
        }
        nsel += __popcll(m);
    }
    if constexpr (!REC) invv = inv_norm_of(group_sum<G>(hv.dot_partial(hv)));
    wave_lds_sync();
    RowT msg, dns;
    msg.zero();
    dns.zero();
    // two kept out-edges per lane group per step, unconditionally (a slot past the end repeats
    // the last one with zero weight): four row loads in flight per group
    for (int q0 = 0; q0 < nsel; q0 += 2 * NG) {
        const int qa = min(q0 + gid, nsel - 1), qb = min(q0 + NG + gid, nsel - 1);
        const bool la = q0 + gid < nsel, lb = q0 + NG + gid < nsel;
        const int ia = s_i[qa], ib = s_i[qb];
        RowT xa, ga, xb, gb;
        xa.load(a.h + (size_t)(ia + a.row_off) * a.C, a.C, lg);
        ga.load(a.gout + (size_t)ia * a.C, a.C, lg);
        xb.load(a.h + (size_t)(ib + a.row_off) * a.C, a.C, lg);
        gb.load(a.gout + (size_t)ib * a.C, a.C, lg);
        if constexpr (REC) {
            float2 ra = s_rec[qa], rb = s_rec[qb];
            if (a.rec_dot) {                                      // (uniform) raw records: BwdArgs::rec_dot
                const bool wa = (__float_as_uint(ra.x) >> 31) != 0, wb = (__float_as_uint(rb.x) >> 31) != 0;   // sign bit: -0.0f too
                const float da_ = a.rec_dot[wa ? ia : 0], db_ = a.rec_dot[wb ? ib : 0];   // unconditional loads
                ra.x = fabsf(ra.x); rb.x = fabsf(rb.x);
                ra.y = wa ? ra.x * (ra.y - da_) : ra.y;
                rb.y = wb ? rb.x * (rb.y - db_) : rb.y;
            }
            if (!la) ra = make_float2(0.f, 0.f);
            if (!lb) rb = make_float2(0.f, 0.f);
            s_edge_rows<VEC, G, R>(xa, ga, ra.x, ra.y, msg, dns);
            s_edge_rows<VEC, G, R>(xb, gb, rb.x, rb.y, msg, dns);
        } else {
            const float da = a.inv_deg[(a.role_mask & 4) ? 0 : ia], db = a.inv_deg[(a.role_mask & 4) ? 0 : ib];       // travel with the rows
            s_edge_recompute<VEC, G, R>(xa, ga, hv, invv, da, la ? 1.0f : 0.0f, msg, dns);
            s_edge_recompute<VEC, G, R>(xb, gb, hv, invv, db, lb ? 1.0f : 0.0f, msg, dns);
        }
    }
    msg.reduce_across_groups();
    dns.reduce_across_groups();
    if (task) {
        if (gid == 0) {
            msg.store(a.partS + (size_t)tq * 2 * a.C, a.C, lg);
            dns.store(a.partS + (size_t)tq * 2 * a.C + a.C, a.C, lg);
        }
    } else if (gid == 0) {
        FinishRows<VEC, G, R> fin;
        fin.load(a, v, lg);
        s_finish<VEC, G, R>(a, v, lg, msg, dns, fin);
    }
}

template <int VEC, int G, int R, bool REC, bool CSRM = false>
__global__ __launch_bounds__(BLOCK) void k_bwd_s(const BwdArgs a)
{
    __shared__ __align__(16) int lds[WAVES][512];
    const int b = blockIdx.x;
    int *lw = lds[threadIdx.x >> 6];
    if (b < a.nbA) s_role_wave<VEC, G, R, REC, CSRM>(a, b, lw, true);
    else if (b < a.nbA + a.nbB) s_role_wave<VEC, G, R, REC, CSRM>(a, b - a.nbA, lw, false);
    else s_role_small<VEC, G, R, REC, CSRM>(a, b - a.nbA - a.nbB, lw);
}

// split sources: one wave per source sums the tasks' partial rows, then finishes
template <int VEC, int G, int R>
__global__ __launch_bounds__(64) void k_bwd_s_fin(const BwdArgs a)
{
    using RowT = Row<VEC, G, R>;
    const int lane = lane_id();
    const int gid = lane / G, lg = lane % G;
    const int p = blockIdx.x;
    const int v = a.sperm[p];
    const int t0 = a.ssplit_task0[p], t1 = a.ssplit_task0[p + 1];
    if (gid != 0) return;
    RowT msg, dns, t;
    msg.zero();
    dns.zero();
    for (int tq = t0; tq < t1; ++tq) {
        t.load(a.partS + (size_t)tq * 2 * a.C, a.C, lg);
        msg.add(t);
        t.load(a.partS + (size_t)tq * 2 * a.C + a.C, a.C, lg);
        dns.add(t);
    }
    FinishRows<VEC, G, R> fin;
    fin.load(a, v, lg);
    s_finish<VEC, G, R>(a, v, lg, msg, dns, fin);
}

static __global__ void k_clear_words(unsigned *__restrict__ p, int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0u;
}

template <int VEC, int G, int R> int launch_agg_bwd(const BwdArgs &a0, hipStream_t st)
{
    constexpr int RPW = 64 / G;
    BwdArgs a = a0;
    if (a.mode == 0) {
        // node-centric: kept bits packed from the forward's weights, then ...
        if (a.kmask_words > 0 && a.kbits == nullptr)
            k_pack_kept<<<ceil_div(a.Ep, 256 * WAVES), 256, 0, st>>>(a.wsel, a.Ep, (unsigned long long *)a.kmask);
        if (a.kbits != nullptr && !(a.top_k >= 1 && a.top_k <= WAVE_T && a.N == a.Ntot)) {
            set_error("internal: the kept-bit backward is the hinted node-centric one");
            return SNGNN_EINVAL;
        }
        if (a.top_k >= 1 && a.top_k <= WAVE_T && a.N == a.Ntot) {
            // ... selective call on a whole graph: every node in one launch; pass S is left with
            // the split sources' tasks
            a.nbA = ceil_div(a.n_med_end + a.n_trest, WAVES);
            const int nbF = ceil_div(a.n_fused, (int64_t)WAVES * RPW);
            a.nbB = std::max(1, (a.nbA + nbF) / std::max(a.nbA, 1));          // k_bwd_w: spacing of the wave-per-node workgroups
            if (a.nbA + nbF > 0) {
                if (a.kbits) k_bwd_w<VEC, G, R, true><<<a.nbA + nbF, BLOCK, 0, st>>>(a);
                else k_bwd_w<VEC, G, R, false><<<a.nbA + nbF, BLOCK, 0, st>>>(a);
            }
            if (a.n_stasks > 0) {
                a.nbA = ceil_div(a.n_stasks, WAVES);
                a.nbB = 0;
                k_bwd_s<VEC, G, R, false, true><<<a.nbA, BLOCK, 0, st>>>(a);
                k_bwd_s_fin<VEC, G, R><<<a.n_ssplit, 64, 0, st>>>(a);
            }
            SN_HIP(hipGetLastError());
            return SNGNN_OK;
        }
        // ... the targets' pass with the fused nodes, then pass S for the sources that are left
        a.nbA = ceil_div(a.n_tasks, WAVES);
        a.nbB = ceil_div(a.n_med_end - a.n_split, WAVES);
        a.nbC = ceil_div(a.n_trest, (int64_t)WAVES * RPW);
        const int nbD = ceil_div(a.n_fused, (int64_t)WAVES * RPW);
        if (a.nbA + a.nbB + a.nbC + nbD > 0) k_bwd_f<VEC, G, R><<<a.nbA + a.nbB + a.nbC + nbD, BLOCK, 0, st>>>(a);
        if (a.n_split > 0) k_bwd_t_fin<<<a.n_split, 256, 0, st>>>(a);
        a.nbA = ceil_div(a.n_stasks, WAVES);
        a.nbB = ceil_div(a.n_smed_end - a.n_ssplit, WAVES);
        const int nbS = ceil_div(a.s_small_end - a.n_smed_end, (int64_t)WAVES * RPW);
        if (a.nbA + a.nbB + nbS > 0) k_bwd_s<VEC, G, R, false, true><<<a.nbA + a.nbB + nbS, BLOCK, 0, st>>>(a);
        if (a.n_ssplit > 0) k_bwd_s_fin<VEC, G, R><<<a.n_ssplit, 64, 0, st>>>(a);
        SN_HIP(hipGetLastError());
        return SNGNN_OK;
    }
    // kept bits in CSC order: cleared, set by pass T, read by pass S.  (A kernel, not
    // hipMemsetAsync: replayed from a captured HIP graph, the memset node did not stay ordered
    // in front of pass T on this stack - ROCm 7.0 runtime under PyTorch 2.10 - and cleared bits
    // that pass T had already set; tests/test_models_gpu.py::test_graphed_epoch_survives_... caught it.)
    if (a.kmask_words > 0)
        k_clear_words<<<(int)std::min<int64_t>(ceil_div(a.kmask_words, (int64_t)256), 1024), 256, 0, st>>>(a.kmask, a.kmask_words);
    // pass T (targets)
    a.nbA = ceil_div(a.n_tasks, WAVES);
    a.nbB = ceil_div(a.n_med_end - a.n_split, WAVES);
    int nbC = ceil_div(a.N - a.n_med_end, (int64_t)WAVES * RPW);
    if (a.nbA + a.nbB + nbC > 0) k_bwd_t<VEC, G, R><<<a.nbA + a.nbB + nbC, BLOCK, 0, st>>>(a);
    if (a.n_split > 0) k_bwd_t_fin<<<a.n_split, 256, 0, st>>>(a);
    // pass S (sources)
    a.nbA = ceil_div(a.n_stasks, WAVES);
    a.nbB = ceil_div(a.n_smed_end - a.n_ssplit, WAVES);
    nbC = ceil_div(a.s_small_end - a.n_smed_end, (int64_t)WAVES * RPW);
    if (a.nbA + a.nbB + nbC > 0) k_bwd_s<VEC, G, R, false><<<a.nbA + a.nbB + nbC, BLOCK, 0, st>>>(a);
    if (a.n_ssplit > 0) k_bwd_s_fin<VEC, G, R><<<a.n_ssplit, 64, 0, st>>>(a);
    SN_HIP(hipGetLastError());
    return SNGNN_OK;
}

int launch_agg_bwd_v1(const RowCfg &cfg, const BwdArgs &a, hipStream_t st);
int launch_agg_bwd_v2(const RowCfg &cfg, const BwdArgs &a, hipStream_t st);
int launch_agg_bwd_v4(const RowCfg &cfg, const BwdArgs &a, hipStream_t st);

}  // namespace sngnn
