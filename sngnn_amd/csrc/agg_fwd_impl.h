// Fused forward of the similarity-navigated aggregation (gfx950).
//
// One launch computes, for every target row i of the CSR-by-target graph,
//     s_e   = <h_i, h_j> / (max(|h_i|,eps) * max(|h_j|,eps))      per in-edge e = (j -> i)
//     keep  = top_k by (s desc, edge position asc) AND s >= thr    (or all, top_k < 0)
//     out_i = (1 / max(deg_i, 1)) * sum_{e kept} s_e * h_j
// i.e. models/models.py:122+132+139-158, :238-239+244-263, :325-326+331-334 of the
// reference without materialising any per-edge [E', C] tensor.
//
// Rows are processed in order of descending in-degree (graph.rperm) in three
// classes, each a block range of the same launch:
//   A  split rows  (deg > WAVE_T): one wave per CHUNK-edge task scores its edges
//      (edge-balanced); a second launch (k_agg_fin) selects and sums per row;
//   B  wave rows   (SMALL_T < deg <= WAVE_T): one wave per row, its 64/G lane
//      groups stride over the row's edges;
//   C  small rows  (deg <= SMALL_T): one G-lane group per row, 64/G rows per wave.
// A row whose degree is <= top_k needs no ranking (only the threshold), so its
// weighted sum is accumulated in the same pass that scores it ("streaming");
// otherwise scores go to LDS (or HBM scratch for split rows), the row's top-k
// is selected, and only the <= top_k kept source rows are gathered again.
//
// The kernels are HBM/Infinity-Cache bound gathers of 4*C-byte rows; each lane
// group reads one whole source row per load (VEC*4 B per lane, coalesced).
#pragma once
#include "device_utils.h"

namespace sngnn {

struct FwdArgs {
    const float *h;
    int C, N;         // N = owned target rows
    int row_off;      // row i's own feature row is h[row_off + i] (node-range partition)
    const int32_t *rowptr, *col, *rperm;
    const int4 *rdesc;     // per degree-sorted slot: {row, first edge, in-degree, 0}
    int k;            // < 0: no selection
    float thr;
    float *out, *wsel, *inv_norm;
    int32_t *sel_src;
    float *sel_w;
    int n_split, n_med_end;     // slots [0,n_split) split, [n_split,n_med_end) wave, rest small
    int n_tasks;
    const int32_t *task_slot, *task_chunk, *split_soff, *split_task0;
    const int32_t *xtask_list, *xtask_ptr;   // tasks grouped by source-range eighth (XCD affinity)
    int xcd_affinity;
    int dynamic;                // wave rows and small-row sets are handed out by atomic counters
    int32_t *dyn_ctr;           // [2 classes][DYN_SHARDS] counters, 128 bytes apart, zero at launch
    float *scores, *partial;    // workspace
    unsigned long long *cand_key;   // [n_tasks, k]  chunk-local top-k keys of split rows (k <= CAND_MAX_K)
    int32_t *cand_src;              // [n_tasks, k]  their source ids (saves the finalize a dependent load)
    int lowbits;                    // bits needed for a row-local edge index
    int32_t *split_cnt;             // [n_split] groups arrived (in-kernel finalize)
    int32_t *grp_cnt;               // [n_groups] tasks arrived per group of FIN_GT tasks
    const int32_t *split_grp0;      // [n_split+1] first group of each split row
    unsigned long long *cand2;      // [n_groups, k] champions of each group
    int nbA, nbB, nbC;          // (unused by the persistent kernel)
    int dbg_classes;            // tuning aid: bit 0 tasks, bit 1 wave rows, bit 2 small rows
    int dbg_blocks_per_cu;      // tuning aid: persistent grid size override (0 = default)
    int inkernel_fin;           // split rows finalized by their last-arriving task (experimental)
    int n_split_gt_wave;        // split rows with more than 128 / k tasks (descending order: the first ones)
    int use_dma;                // classes A/B stream source rows through LDS-DMA (C % 4 == 0, C <= 256)
};

// The switches below compile measured-but-unprofitable variants into the
// main kernel; off by default so they cost the shipped kernel no registers.
#ifndef SNGNN_EXPERIMENTAL
#define SNGNN_EXPERIMENTAL 0         // in-kernel finalize, dynamic work counters, XCD-affine tasks
#endif
#ifndef SNGNN_ENABLE_DMA
#define SNGNN_ENABLE_DMA 0           // LDS-DMA scoring path: measured no faster than register staging
#endif
constexpr int DMA_NI = 2;            // LDS-DMA instructions (1 KiB each) per batch of source rows
constexpr int LDS_DMA_OFF = 384;     // words: [0,128) scores | [128,256) kept list | [256,384) column ids
constexpr int LDS_PER_WAVE = SNGNN_ENABLE_DMA ? LDS_DMA_OFF + 2 * DMA_NI * 256 : 512;   // + two DMA buffers
constexpr int FWD_WAVES_PER_SIMD = 6;   // register budget of the main kernel (<= 80 VGPRs)

template <int R> struct Unroll { static constexpr int U = (R >= 4) ? 1 : (R == 2 ? 2 : 4); };

// ---------------------------------------------------------------------------
// Wave-level top-k of up to 128 selection keys, two per lane (0 = no key).
// Keys are unique, so "the k largest" is well defined: bitwise search for the
// k-th largest key T, then keep key >= T.  lowbits = bits of the largest
// row-local edge index (the low word of a key is 0xFFFFFFFF - index, so its
// upper 32 - lowbits bits are all ones and need no search).
// ---------------------------------------------------------------------------
constexpr int CAND_MAX_K = 32;   // split rows: chunk-local candidates are kept for k <= this

__device__ __forceinline__ void wave_topk_keys(unsigned long long key0, unsigned long long key1,
                                               int k, int lowbits, bool &kept0, bool &kept1)
{
    const bool v0 = key0 != 0ull, v1 = key1 != 0ull;
    const int cnt = __popcll(__ballot(v0)) + __popcll(__ballot(v1));
    if (cnt <= k) { kept0 = v0; kept1 = v1; return; }     // everything that passed thr fits
    unsigned long long T = 0;
    for (int b = 63; b >= 32; --b) {
        const unsigned long long cand = T | (1ull << b);
        const int c = __popcll(__ballot(key0 >= cand)) + __popcll(__ballot(key1 >= cand));
        if (c >= k) T = cand;
    }
    T |= 0xFFFFFFFFull & ~((1ull << lowbits) - 1ull);
    for (int b = lowbits - 1; b >= 0; --b) {
        const unsigned long long cand = T | (1ull << b);
        const int c = __popcll(__ballot(key0 >= cand)) + __popcll(__ballot(key1 >= cand));
        if (c >= k) T = cand;
    }
    kept0 = key0 >= T;      // T > 0 here, so empty slots (key 0) stay out
    kept1 = key1 >= T;
}

__device__ __forceinline__ float key_score(unsigned long long key)
{
    const unsigned u = (unsigned)(key >> 32);
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}
__device__ __forceinline__ int key_index(unsigned long long key)
{
    return (int)(0xFFFFFFFFu - (unsigned)key);
}

// Top-k (and >= thr) of the scores sc[0, n) of one wave's edges; edge t of the
// wave has row-local index base + t.  An edge below thr gets no key at all:
// "top-k, then drop < thr" == "drop < thr, then top-k" because the order is by score.
struct WaveSel {
    bool kept0, kept1;
    unsigned long long key0, key1;
};

__device__ __forceinline__ WaveSel wave_select(const float *sc, int n, int base, int k, float thr,
                                               int lowbits)
{
    const int lane = lane_id();
    const int i0 = lane, i1 = lane + 64;
    const float s0 = i0 < n ? sc[i0] : 0.f, s1 = i1 < n ? sc[i1] : 0.f;
    WaveSel r;
    r.key0 = (i0 < n && s0 >= thr) ? sel_key(s0, base + i0) : 0ull;
    r.key1 = (i1 < n && s1 >= thr) ? sel_key(s1, base + i1) : 0ull;
    wave_topk_keys(r.key0, r.key1, k, lowbits, r.kept0, r.kept1);
    return r;
}

// ---------------------------------------------------------------------------
// Class C: deg <= SMALL_T, one group per row.
// ---------------------------------------------------------------------------
// One set of 64/G small rows (one per lane group) whose column ids are already in
// LDS (s_col[gid][t]).  d = this group's row descriptor (deg 0 for a padding slot).
template <int VEC, int G, int R>
__device__ __forceinline__ void small_rows_set(const FwdArgs &a, const int4 d, bool valid,
                                               int *lds_wave, const int *s_col_set)
{
    using RowT = Row<VEC, G, R>;
    constexpr int U = Unroll<R>::U;
    const int lane = lane_id();
    const int gid = lane / G, lg = lane % G;
    const int i = d.x, rs = d.y, deg = d.z;
    const bool emit = a.sel_src != nullptr && a.k >= 0;
    const bool rank = a.k >= 0 && deg > a.k;
    const bool need_sc = rank || emit;

    const int *s_col = s_col_set + gid * SMALL_T;
    float *s_sc = reinterpret_cast<float *>(lds_wave + 256) + gid * SMALL_T;
    float *s_w = reinterpret_cast<float *>(lds_wave + 384) + gid * SMALL_T;

    RowT hi;
    hi.load(a.h + (size_t)(i + a.row_off) * a.C, a.C, lg);
    const int dmax = wave_max_i(deg);

    RowT acc;
    acc.zero();
    float inv_i = 0.f;
    for (int t0 = 0; t0 < dmax; t0 += U) {
        RowT x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = (t0 + u) < deg ? s_col[t0 + u] : i + a.row_off;
            x[u].load(a.h + (size_t)j * a.C, a.C, lg);
        }
        if (t0 == 0) inv_i = inv_norm_of(group_sum<G>(hi.dot_partial(hi)));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float s = edge_score<VEC, G, R>(hi, inv_i, x[u]);
            if (t0 + u < deg) {
                if (need_sc && lg == 0) s_sc[t0 + u] = s;
                if (!rank) {
                    const bool sel = (a.k < 0) || (s >= a.thr);
                    if (sel) acc.axpy(s, x[u]);
                    if (a.wsel && lg == 0) a.wsel[rs + t0 + u] = sel ? s : SNGNN_UNSELECTED;
                }
            }
        }
    }
    if (a.inv_norm) {      // isolated rows never enter the loop
        if (dmax == 0) inv_i = inv_norm_of(group_sum<G>(hi.dot_partial(hi)));
        if (valid && lg == 0) a.inv_norm[i] = inv_i;
    }

    if (need_sc) {
        wave_lds_sync();
        // rank of every edge of the row under (score desc, position asc)
        for (int e = lg; e < deg; e += G) {
            const float se = s_sc[e];
            int rk = 0;
            for (int b = 0; b < deg; ++b) {
                const float sb = s_sc[b];
                rk += (sb > se) || (sb == se && b < e);
            }
            const bool sel = rk < a.k && se >= a.thr;
            s_w[e] = sel ? se : SNGNN_UNSELECTED;
            if (rank && a.wsel) a.wsel[rs + e] = sel ? se : SNGNN_UNSELECTED;
            if (emit && sel) {
                a.sel_src[(size_t)i * a.k + rk] = s_col[e];
                a.sel_w[(size_t)i * a.k + rk] = se;
            }
        }
        if (rank) {
            wave_lds_sync();
            // second pass: gather only the kept source rows, in edge order
            // (group-divergent trip count: no cross-lane operation inside).  Batching these
            // loads (compaction + U rows in flight) was measured: +1.5 us at k = 1 where almost
            // nothing is kept, no gain at k = 16 where small rows never rank - left serial.
            for (int t = 0; t < deg; ++t) {
                const float w = s_w[t];
                if (w != SNGNN_UNSELECTED) {
                    RowT xr;
                    xr.load(a.h + (size_t)s_col[t] * a.C, a.C, lg);
                    acc.axpy(w, xr);
                }
            }
        }
        wave_lds_sync();       // s_sc / s_w are reused by the next set
    }
    if (valid) {
        acc.div((float)max(deg, 1));
        acc.store(a.out + (size_t)i * a.C, a.C, lg);
    }
}

// ---------------------------------------------------------------------------
// Class C: deg <= SMALL_T, one G-lane group per row, 64/G rows per set.  Waves
// are persistent: each walks sets wave_id, wave_id + n_waves, ... and keeps the
// NEXT set's descriptors and column ids in flight while it works on the current
// one, so a set costs one memory round trip (its feature rows) instead of a
// chain of three (descriptor -> columns -> rows).
// ---------------------------------------------------------------------------
// Hands a wave its small-row sets.  Static: set0, set0 + stride, ...  Dynamic: chunks of
// DYN_SETS consecutive sets taken from a per-shard atomic counter (shard = workgroup id
// mod #shards; chunk c of shard x covers sets (x + #shards c) * DYN_SETS ...), the next chunk is
// requested as soon as the current one is entered so the atomic's latency is hidden.
constexpr int DYN_SETS = 2;
constexpr int DYN_SHARDS = 64;          // counters per class (workgroup id mod DYN_SHARDS)
constexpr int DYN_CTR_STRIDE = 32;      // ints (128 bytes) between counters

__device__ __forceinline__ int dyn_fetch(int32_t *ctr)
{
    int v = 0;
    if (lane_id() == 0) v = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;      // lane 0 holds the value: readfirstlane at the point of use
}

struct SetSeq {
    int pos, end, stride, nsets;        // static: pos advances by stride; dynamic: [pos, end) is a chunk
    int nxt_raw;                        // dynamic: result of the pending dequeue (lane 0)
    int32_t *ctr;                       // nullptr = static
    int shard, nsh;                     // this wave's shard and the number of shards

    __device__ __forceinline__ void init_static(int set0, int st, int n)
    { pos = set0; stride = st; nsets = n; end = n; ctr = nullptr; nxt_raw = 0; shard = 0; nsh = 1; }

    __device__ __forceinline__ void init_dynamic(int32_t *c, int sh, int nshards, int n)
    {
        ctr = c; shard = sh; nsh = nshards; nsets = n; stride = 1;
        const int c0 = __builtin_amdgcn_readfirstlane(dyn_fetch(ctr));
        pos = (shard + nsh * c0) * DYN_SETS;
        end = min(pos + DYN_SETS, nsets);
        nxt_raw = pos < nsets ? dyn_fetch(ctr) : 0;
    }

    // next set id, or -1 when the wave's share is exhausted
    __device__ __forceinline__ int next()
    {
        if (pos >= nsets) return -1;
        const int s = pos;
        pos += stride;
        if (ctr != nullptr && pos >= end) {             // enter the chunk requested earlier
            const int c = __builtin_amdgcn_readfirstlane(nxt_raw);
            pos = (shard + nsh * c) * DYN_SETS;
            end = min(pos + DYN_SETS, nsets);
            if (pos < nsets) nxt_raw = dyn_fetch(ctr); else pos = nsets;
        }
        return s;
    }
};

template <int VEC, int G, int R>
__device__ __forceinline__ void role_small(const FwdArgs &a, SetSeq &seq, int *lds_wave)
{
    constexpr int RPW = 64 / G;
    constexpr int CPL = (RPW * SMALL_T + 63) / 64;      // column ids per lane per set
    const int lane = lane_id();
    const int gid = lane / G;
    const int nsets = seq.nsets;

    auto load_desc = [&](int st) -> int4 {
        const int slot = a.n_med_end + st * RPW + gid;
        return (st >= 0 && st < nsets && slot < a.N) ? a.rdesc[slot] : make_int4(0, 0, 0, -1);
    };
    // lane l fetches column ids q = l + 64 m of the set: row q / SMALL_T, edge q % SMALL_T
    auto load_cols = [&](const int4 d, int (&c)[CPL]) {
#pragma unroll
        for (int m = 0; m < CPL; ++m) {
            const int q = lane + 64 * m;
            const int r = q / SMALL_T, t = q % SMALL_T;
            // descriptor of row r of the set lives in the lanes of group r
            const int rs = __shfl(d.y, r * G, 64), dg = __shfl(d.z, r * G, 64);
            c[m] = (r < RPW && t < dg) ? a.col[rs + t] : 0;
        }
    };
    auto store_cols = [&](const int (&c)[CPL], int *dst) {
#pragma unroll
        for (int m = 0; m < CPL; ++m) {
            const int q = lane + 64 * m;
            if (q < RPW * SMALL_T) dst[q] = c[m];
        }
    };

    int s_cur = seq.next();
    if (s_cur < 0) return;
    int s_nxt = seq.next();
    // two [RPW][SMALL_T] column-id buffers at lds_wave + 128 * buf.  (Plain pointer arithmetic:
    // an array of the two pointers loses the LDS address space and every read of a column
    // id becomes a FLAT load, which must drain the whole memory pipeline - s_waitcnt
    // vmcnt(0) - in the middle of a batch of row gathers.)
    int4 d_cur = load_desc(s_cur);
    int4 d_nxt = load_desc(s_nxt);
    int cols[CPL];
    load_cols(d_cur, cols);
    store_cols(cols, lds_wave);
    int buf = 0;
    while (s_cur >= 0) {
        const int s_n2 = seq.next();
        const int4 d_n2 = load_desc(s_n2);
        load_cols(d_nxt, cols);                 // in flight during this set's work
        wave_lds_sync();
        small_rows_set<VEC, G, R>(a, d_cur, d_cur.w == 0, lds_wave, lds_wave + 128 * buf);
        store_cols(cols, lds_wave + 128 * (buf ^ 1));
        d_cur = d_nxt;
        d_nxt = d_n2;
        s_cur = s_nxt;
        s_nxt = s_n2;
        buf ^= 1;
    }
}

// ---------------------------------------------------------------------------
// Scoring pass shared by classes A and B: the wave's 64/G groups stride over
// the edges [e0, e1) of row i (row-local indices).
//   STREAM: accumulate kept rows into acc (threshold only) and write wsel
//   sc: where to put the scores (LDS or HBM scratch - every CALL SITE passes one kind only: a
//           pointer that is LDS on one path and HBM on another becomes FLAT, and FLAT accesses
//           wait for vmcnt(0), i.e. drain the gathers in flight), or nullptr; edge t goes to
//           sc[t - sc_off]  (never form an out-of-range LDS pointer: LDS pointer
//           arithmetic is 32-bit and does not survive the cast to a flat address)
// ---------------------------------------------------------------------------
template <int VEC, int G, int R>
__device__ __forceinline__ void score_edges(const FwdArgs &a, int i, int rs, int e0, int e1,
                                            const Row<VEC, G, R> &hi, float inv_i, bool stream,
                                            float *sc, int sc_off, Row<VEC, G, R> &acc,
                                            int *ids_lds = nullptr)
{
    using RowT = Row<VEC, G, R>;
    constexpr int NG = 64 / G;
    constexpr int U = Unroll<R>::U;
    const int lane = lane_id();
    const int gid = lane / G, lg = lane % G;
    // column ids one iteration ahead: the col -> row dependency of iteration n+1
    // overlaps the row loads of iteration n
    int jn[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int t = e0 + u * NG + gid;
        jn[u] = t < e1 ? a.col[rs + t] : i + a.row_off;
    }
    for (int base = e0; base < e1; base += NG * U) {
        int t[U], j[U];
        bool act[U];
        RowT x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            t[u] = base + u * NG + gid;
            act[u] = t[u] < e1;
            j[u] = jn[u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) x[u].load(a.h + (size_t)j[u] * a.C, a.C, lg);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int tn = t[u] + NG * U;
            jn[u] = tn < e1 ? a.col[rs + tn] : i + a.row_off;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float s = edge_score<VEC, G, R>(hi, inv_i, x[u]);
            if (act[u]) {
                if (sc && lg == 0) sc[t[u] - sc_off] = s;
                if (ids_lds && lg == 0) ids_lds[t[u] - sc_off] = j[u];      // for the re-gather of the kept rows
                if (stream) {
                    const bool sel = (a.k < 0) || (s >= a.thr);
                    if (sel) acc.axpy(s, x[u]);
                    if (a.wsel && lg == 0) a.wsel[rs + t[u]] = sel ? s : SNGNN_UNSELECTED;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Scoring pass of classes A and B with the source rows streamed through LDS-DMA
// (global_load_lds_dwordx4: per-lane source address, wave-contiguous 1 KiB LDS
// destination, no VGPR destination).  One DMA instruction gathers 64 / (C/4) whole
// rows; a batch is DMA_NI instructions; two batches ping-pong, so the next batch is
// in flight while the current one is scored and the number of rows in flight does
// not depend on the register budget.  The column ids were put in LDS (s_ids) with
// ordinary loads BEFORE the first DMA: beside an LDS-DMA in flight hipcc waits
// vmcnt(0) for any VGPR-destination load, which would drain the queue.
// Edges are [e0, e0 + n), n <= 128.  Same arithmetic and order as score_edges.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void wait_vmcnt_upto(int n)      // n wave-uniform, 0..DMA_NI
{
    if (n <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
}
static_assert(DMA_NI == 2, "wait_vmcnt_upto covers 0..2");

template <int VEC, int G, int R>
__device__ __forceinline__ void score_edges_dma(const FwdArgs &a, int rs, int e0, int n,
                                                const Row<VEC, G, R> &hi, float inv_i, bool stream,
                                                float *s_sc, const int *s_ids, float *dma,
                                                Row<VEC, G, R> &acc)
{
    using RowT = Row<VEC, G, R>;
    constexpr int NG = 64 / G;
    const int lane = lane_id();
    const int gid = lane / G, lg = lane % G;
    const int CH = a.C >> 2;                      // 16-byte chunks per row (<= G)
    const int RPI = 64 / CH;                      // rows per DMA instruction
    const int BR = DMA_NI * RPI;                  // rows per batch
    const int r_in = lane / CH, c_in = lane - r_in * CH;
    const bool dlane = r_in < RPI;
    const int nb = (n + BR - 1) / BR;
    auto issue = [&](int b, int buf) {
#pragma unroll
        for (int q = 0; q < DMA_NI; ++q) {
            const int e = b * BR + q * RPI + r_in;
            if (dlane && e < n) {
                const int j = s_ids[e];
                __builtin_amdgcn_global_load_lds(
                    a.h + (size_t)j * a.C + c_in * 4,
                    (__attribute__((address_space(3))) void *)(dma + (buf * DMA_NI + q) * 256), 16, 0, 0);
            }
        }
    };
    auto n_instr = [&](int b) {                   // DMA instructions of batch b with an active lane
        const int rows = min(BR, n - b * BR);
        return rows <= 0 ? 0 : (rows + RPI - 1) / RPI;
    };
    issue(0, 0);
    if (nb > 1) issue(1, 1);
    for (int b = 0; b < nb; ++b) {
        wait_vmcnt_upto(b + 1 < nb ? n_instr(b + 1) : 0);     // batch b has landed
        const float *buf = dma + (b & 1) * DMA_NI * 256;
        const int rows = min(BR, n - b * BR);
        for (int r0 = 0; r0 < rows; r0 += NG) {
            const int r = r0 + gid;
            const bool act = r < rows;
            const int rr = act ? r : 0;
            RowT x;
            x.load(buf + (rr / RPI) * 256 + (rr % RPI) * CH * 4, a.C, lg);
            const float s = edge_score<VEC, G, R>(hi, inv_i, x);
            if (act) {
                const int t = b * BR + r;             // chunk-local edge index
                if (s_sc && lg == 0) s_sc[t] = s;
                if (stream) {
                    const bool sel = (a.k < 0) || (s >= a.thr);
                    if (sel) acc.axpy(s, x);
                    if (a.wsel && lg == 0) a.wsel[rs + e0 + t] = sel ? s : SNGNN_UNSELECTED;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this buffer's reads are done
        if (b + 2 < nb) issue(b + 2, b & 1);
    }
}

// ---------------------------------------------------------------------------
// Class B: SMALL_T < deg <= WAVE_T, one wave per row.
// ---------------------------------------------------------------------------
template <int VEC, int G, int R>
__device__ __forceinline__ void role_wave(const FwdArgs &a, int item, int *lds_wave)
{
    using RowT = Row<VEC, G, R>;
    constexpr int NG = 64 / G;
    const int lane = lane_id();
    const int gid = lane / G, lg = lane % G;
    const int slot = a.n_split + item;
    const int4 d = a.rdesc[slot];
    const int i = d.x, rs = d.y, deg = d.z;
    const bool emit = a.sel_src != nullptr && a.k >= 0;
    const bool rank = a.k >= 0 && deg > a.k;
    const bool need_sc = rank || emit;

    float *s_sc = reinterpret_cast<float *>(lds_wave);     // [WAVE_T]
    int *s_list = lds_wave + WAVE_T;                         // [WAVE_T]

    RowT hi;
    hi.load(a.h + (size_t)(i + a.row_off) * a.C, a.C, lg);
    const float inv_i = inv_norm_of(group_sum<G>(hi.dot_partial(hi)));
    // (inv_norm[i] is stored at the END of the row: a store here would pin the scoring pass's
    // first column-id loads behind it - the pointers are not restrict - and with them behind
    // the wait for hi)

    RowT acc;
    acc.zero();
    bool dma_done = false;
    if constexpr (SNGNN_ENABLE_DMA && VEC == 4 && R == 1) {
        if (a.use_dma) {
            int *s_ids = lds_wave + 2 * WAVE_T;
            for (int t = lane; t < deg; t += 64) s_ids[t] = a.col[rs + t];
            wave_lds_sync();
            score_edges_dma<VEC, G, R>(a, rs, 0, deg, hi, inv_i, !rank, need_sc ? s_sc : nullptr, s_ids,
                                       reinterpret_cast<float *>(lds_wave + LDS_DMA_OFF), acc);
            dma_done = true;
        }
    }
    if (!dma_done)
        score_edges<VEC, G, R>(a, i, rs, 0, deg, hi, inv_i, !rank, need_sc ? s_sc : nullptr, 0, acc,
                               rank ? lds_wave + 2 * WAVE_T : nullptr);

    if (need_sc) {
        wave_lds_sync();
        const WaveSel ws = wave_select(s_sc, deg, 0, a.k, a.thr, 7);
        const int i0 = lane, i1 = lane + 64;
        // kept list in ascending position order
        const unsigned long long m0 = __ballot(ws.kept0), m1 = __ballot(ws.kept1);
        const int n0 = __popcll(m0);
        if (ws.kept0) s_list[prefix_popc(m0)] = i0;
        if (ws.kept1) s_list[n0 + prefix_popc(m1)] = i1;
        const int nsel = n0 + __popcll(m1);
        if (rank && a.wsel) {
            if (i0 < deg) a.wsel[rs + i0] = ws.kept0 ? s_sc[i0] : SNGNN_UNSELECTED;
            if (i1 < deg) a.wsel[rs + i1] = ws.kept1 ? s_sc[i1] : SNGNN_UNSELECTED;
        }
        wave_lds_sync();
        if (emit) {
            // rank of a kept edge = number of keys above it
            for (int q = 0; q < nsel; ++q) {
                const int idx = s_list[q];
                const unsigned long long kq = sel_key(s_sc[idx], idx);
                const int rk = __popcll(__ballot(ws.key0 > kq)) + __popcll(__ballot(ws.key1 > kq));
                if (lane == 0) {
                    a.sel_src[(size_t)i * a.k + rk] = a.col[rs + idx];
                    a.sel_w[(size_t)i * a.k + rk] = s_sc[idx];
                }
            }
        }
        if (rank && nsel > 0) {
            // the kept rows again, U per lane group in flight, unconditionally (a slot past the
            // end repeats the last kept edge with weight 0): their column ids wait in LDS, so
            // this is ONE memory round trip instead of a col -> row chain per kept edge
            constexpr int U = Unroll<R>::U;
            const int *s_ids = dma_done ? nullptr : lds_wave + 2 * WAVE_T;
            for (int q0 = 0; q0 < nsel; q0 += U * NG) {
                RowT x[U];
                float w[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int q = q0 + u * NG + gid;
                    const int idx = s_list[min(q, nsel - 1)];
                    const int j = s_ids ? s_ids[idx] : a.col[rs + idx];
                    w[u] = q < nsel ? s_sc[idx] : 0.f;
                    x[u].load(a.h + (size_t)j * a.C, a.C, lg);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) acc.axpy(w[u], x[u]);
            }
        }
    }
    acc.reduce_across_groups();
    if (gid == 0) {
        acc.div((float)deg);
        acc.store(a.out + (size_t)i * a.C, a.C, lg);
    }
    if (lane == 0 && a.inv_norm) a.inv_norm[i] = inv_i;
    wave_lds_sync();            // the wave's LDS scratch is reused by its next item
}

// ---------------------------------------------------------------------------
// In-kernel finalize of split rows (top_k <= CAND_MAX_K), two levels, no waiting:
//   * the tasks of a row form groups of FIN_GT; the wave whose task is the LAST of its
//     group to arrive merges the group's candidate keys into <= k champions;
//   * the wave whose group is the last of the ROW to arrive merges the groups'
//     champions, gathers the <= k winners and writes the row.
// Nobody spins: a wave only ever waits for its own stores.  Hand-off (guide G16):
// producers store keys write-through (sc1) and drain vmcnt before their relaxed
// agent-scope counter add; a last arriver takes an agent-scope acquire and reads the
// keys with sc1 loads.  Counters are reset by the last arriver (0 between launches).
// ---------------------------------------------------------------------------
constexpr int FIN_GT = 6;      // tasks per group: 6 * 16 keys fill one merge step beside 32 champions
static_assert(FIN_GT == FIN_GT_HOST, "keep graph.hip in sync");

// merge n keys at ck (sc1 loads) into champ[] (LDS, <= k keys); returns their number
__device__ __forceinline__ int merge_keys(const unsigned long long *ck, int n, int k, int lowbits,
                                          unsigned long long *champ)
{
    const int lane = lane_id();
    int nchamp = 0;
    for (int base = 0; base < n; base += 96) {
        // lanes 0..31 of slot 0: champions so far; the other 96 slots: new keys
        const int q0 = base + lane - 32, q1 = base + 32 + lane;
        unsigned long long key0, key1;
        if (lane < 32) key0 = lane < nchamp ? champ[lane] : 0ull;
        else key0 = q0 < n ? __hip_atomic_load(ck + q0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        key1 = q1 < n ? __hip_atomic_load(ck + q1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        bool k0, k1;
        wave_topk_keys(key0, key1, k, lowbits, k0, k1);
        const unsigned long long m0 = __ballot(k0), m1 = __ballot(k1);
        const int n0 = __popcll(m0);
        wave_lds_sync();                    // every lane has read its champion
        if (k0) champ[prefix_popc(m0)] = key0;
        if (k1) champ[n0 + prefix_popc(m1)] = key1;
        nchamp = n0 + __popcll(m1);
        wave_lds_sync();
    }
    return nchamp;
}

// kept edges of a split row from its champion keys: weights for backward, rank-ordered
// selection, weighted sum
template <int VEC, int G, int R>
__device__ __forceinline__ void finalize_split_row(const FwdArgs &a, const int4 d,
                                                   const unsigned long long *champ, int nchamp)
{
    using RowT = Row<VEC, G, R>;
    constexpr int NG = 64 / G;
    const int lane = lane_id();
    const int gid = lane / G, lg = lane % G;
    const int i = d.x, rs = d.y, deg = d.z;
    const bool emit = a.sel_src != nullptr;
    if (lane < nchamp) {
        const unsigned long long kq = champ[lane];
        const int idx = key_index(kq);
        const float sq = key_score(kq);
        if (a.wsel) a.wsel[rs + idx] = sq;
        if (emit) {
            int rk = 0;
            for (int r = 0; r < nchamp; ++r) rk += champ[r] > kq;
            a.sel_src[(size_t)i * a.k + rk] = a.col[rs + idx];
            a.sel_w[(size_t)i * a.k + rk] = sq;
        }
    }
    RowT acc;
    acc.zero();
    for (int q0 = 0; q0 < nchamp; q0 += NG) {
        const int q = q0 + gid;
        if (q < nchamp) {
            const unsigned long long kq = champ[q];
            RowT x;
            x.load(a.h + (size_t)a.col[rs + key_index(kq)] * a.C, a.C, lg);
            acc.axpy(key_score(kq), x);
        }
    }
    acc.reduce_across_groups();
    if (gid == 0) {
        acc.div((float)deg);
        acc.store(a.out + (size_t)i * a.C, a.C, lg);
    }
}

// my stores are out -> count me in; true for the last of `total` arrivers, which
// then also holds an acquire and has reset the counter
__device__ __forceinline__ bool arrive_last(int32_t *cnt, int total)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int arrived = 0;
    if (lane_id() == 0)
        arrived = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    arrived = __builtin_amdgcn_readfirstlane(arrived);
    if (arrived != total - 1) return false;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane_id() == 0) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

// ---------------------------------------------------------------------------
// Class A: one CHUNK-edge task of a split row.
// ---------------------------------------------------------------------------
template <int VEC, int G, int R>
__device__ __forceinline__ void role_task(const FwdArgs &a, int tq, int *lds_wave)
{
    using RowT = Row<VEC, G, R>;
    const int lane = lane_id();
    const int gid = lane / G, lg = lane % G;
    const int p = a.task_slot[tq], c = a.task_chunk[tq];
    const int4 d = a.rdesc[p];
    const int i = d.x, rs = d.y, deg = d.z;
    const int e0 = c * CHUNK, e1 = min(deg, e0 + CHUNK);
    const bool emit = a.sel_src != nullptr && a.k >= 0;
    const bool rank = a.k >= 0 && deg > a.k;
    const bool cand = rank && a.k <= CAND_MAX_K;        // chunk-local top-k -> candidates

    RowT hi;
    hi.load(a.h + (size_t)(i + a.row_off) * a.C, a.C, lg);
    const float inv_i = inv_norm_of(group_sum<G>(hi.dot_partial(hi)));
    if (c == 0 && lane == 0 && a.inv_norm) a.inv_norm[i] = inv_i;

    RowT acc;
    acc.zero();
    float *s_sc = reinterpret_cast<float *>(lds_wave);          // [CHUNK], chunk-local
    float *sc_glb = (!cand && (rank || emit)) ? a.scores + a.split_soff[p] : nullptr;   // HBM scratch
    bool dma_done = false;
    if constexpr (SNGNN_ENABLE_DMA && VEC == 4 && R == 1) {
        if (a.use_dma && (cand || (!rank && !emit))) {      // scores (if any) go to LDS in these modes
            int *s_ids = lds_wave + 2 * WAVE_T;
            for (int t = lane; t < e1 - e0; t += 64) s_ids[t] = a.col[rs + e0 + t];
            wave_lds_sync();
            score_edges_dma<VEC, G, R>(a, rs, e0, e1 - e0, hi, inv_i, !rank, cand ? s_sc : nullptr, s_ids,
                                       reinterpret_cast<float *>(lds_wave + LDS_DMA_OFF), acc);
            dma_done = true;
        }
    }
    if (!dma_done) {
        if (cand) score_edges<VEC, G, R>(a, i, rs, e0, e1, hi, inv_i, !rank, s_sc, e0, acc);          // LDS
        else score_edges<VEC, G, R>(a, i, rs, e0, e1, hi, inv_i, !rank, sc_glb, 0, acc);             // HBM / none
    }
    if (cand) {
        wave_lds_sync();
        const WaveSel ws = wave_select(s_sc, e1 - e0, e0, a.k, a.thr, a.lowbits);
        unsigned long long *ck = a.cand_key + (size_t)tq * a.k;
        const unsigned long long m0 = __ballot(ws.kept0), m1 = __ballot(ws.kept1);
        const int n0 = __popcll(m0), nsel = n0 + __popcll(m1);
        // write-through (sc1) stores: the row's last-arriving wave reads these keys
        if (ws.kept0) __hip_atomic_store(ck + prefix_popc(m0), ws.key0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ws.kept1) __hip_atomic_store(ck + n0 + prefix_popc(m1), ws.key1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int32_t *cs = a.cand_src + (size_t)tq * a.k;
        if (ws.kept0) cs[prefix_popc(m0)] = a.col[rs + e0 + lane];
        if (ws.kept1) cs[n0 + prefix_popc(m1)] = a.col[rs + e0 + 64 + lane];
        if (lane >= nsel && lane < a.k)           // empty slots (k <= 32 < 64)
            __hip_atomic_store(ck + lane, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a.wsel) {     // the finalize overwrites the kept edges of the row
            float *w = a.wsel + rs + e0;
            if (lane < e1 - e0) __hip_atomic_store(w + lane, SNGNN_UNSELECTED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane + 64 < e1 - e0) __hip_atomic_store(w + lane + 64, SNGNN_UNSELECTED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (SNGNN_EXPERIMENTAL && a.inkernel_fin) {
            unsigned long long *champ = reinterpret_cast<unsigned long long *>(lds_wave);   // [32]
            const int t0 = a.split_task0[p], nt = a.split_task0[p + 1] - t0;
            const int gl = c / FIN_GT;                                  // my group inside the row
            const int g0 = a.split_grp0[p], ng = a.split_grp0[p + 1] - g0;
            const int gsize = min(FIN_GT, nt - gl * FIN_GT);
            if (arrive_last(a.grp_cnt + g0 + gl, gsize)) {              // wave-uniform
                wave_lds_sync();
                int nchamp = merge_keys(a.cand_key + (size_t)(t0 + gl * FIN_GT) * a.k, gsize * a.k, a.k,
                                        a.lowbits, champ);
                bool mine = ng == 1;
                if (!mine) {
                    unsigned long long *c2 = a.cand2 + (size_t)(g0 + gl) * a.k;
                    if (lane < a.k)
                        __hip_atomic_store(c2 + lane, lane < nchamp ? champ[lane] : 0ull, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                    if (arrive_last(a.split_cnt + p, ng)) {
                        wave_lds_sync();
                        nchamp = merge_keys(a.cand2 + (size_t)g0 * a.k, ng * a.k, a.k, a.lowbits, champ);
                        mine = true;
                    }
                }
                if (mine) finalize_split_row<VEC, G, R>(a, d, champ, nchamp);
            }
        }
    } else if (!rank) {
        acc.reduce_across_groups();
        if (gid == 0) acc.store(a.partial + (size_t)tq * a.C, a.C, lg);
    }
    wave_lds_sync();            // the wave's LDS scratch is reused by its next item
}

// Persistent waves: wave w of the grid takes work items w, w + n_waves, ... of the
// list [split-row tasks | wave rows | small-row sets], each class in order of
// descending degree, so every wave gets a similar mix and the grid drains evenly.
template <int VEC, int G, int R>
__global__ __launch_bounds__(BLOCK, FWD_WAVES_PER_SIMD) void k_agg_fwd(const FwdArgs a)
{
    __shared__ int lds[WAVES][LDS_PER_WAVE];
    const int wave = threadIdx.x >> 6;
    int *lw = lds[wave];
    const int nw = gridDim.x * WAVES;
    const int n_wave_rows = a.n_med_end - a.n_split;
    int it = blockIdx.x * WAVES + wave;
    if (SNGNN_EXPERIMENTAL && a.xcd_affinity && (gridDim.x & 7) == 0) {
        // Workgroups b and b + 8 share an XCD (observed round-robin placement; speed
        // only, never correctness).  XCD group x takes the tasks whose sources lie in
        // the x-th eighth of the node range: ~1/8 of the feature table per L2.
        const int x = blockIdx.x & 7;
        const int j = (blockIdx.x >> 3) * WAVES + wave, nwx = (gridDim.x >> 3) * WAVES;
        const int q1 = a.xtask_ptr[x + 1];
        if (a.dbg_classes & 1)
            for (int q = a.xtask_ptr[x] + j; q < q1; q += nwx) role_task<VEC, G, R>(a, a.xtask_list[q], lw);
        // waves without a task (high j) start with the biggest wave rows
        it = (nwx - 1 - j) * 8 + x;
    } else {
        for (; it < a.n_tasks; it += nw)
            if (a.dbg_classes & 1) role_task<VEC, G, R>(a, it, lw);
        it -= a.n_tasks;
    }
    constexpr int RPW = 64 / G;
    const int nsets = (a.N - a.n_med_end + RPW - 1) / RPW;
    SetSeq seq;
    if (SNGNN_EXPERIMENTAL && a.dynamic) {
        // after its (static) tasks a wave asks for work: waves that also finalized a
        // split row simply come back later and take less
        const int nsh = min(DYN_SHARDS, (int)gridDim.x);     // every shard must have a workgroup
        const int shard = blockIdx.x % nsh;
        int32_t *ctr_b = a.dyn_ctr + shard * DYN_CTR_STRIDE;
        int raw = dyn_fetch(ctr_b);
        for (;;) {
            const int row = shard + nsh * __builtin_amdgcn_readfirstlane(raw);
            if (row >= n_wave_rows) break;
            raw = dyn_fetch(ctr_b);             // requested before the row is processed
            if (a.dbg_classes & 2) role_wave<VEC, G, R>(a, row, lw);
        }
        seq.init_dynamic(a.dyn_ctr + (DYN_SHARDS + shard) * DYN_CTR_STRIDE, shard, nsh, nsets);
    } else {
        for (; it < n_wave_rows; it += nw)
            if (a.dbg_classes & 2) role_wave<VEC, G, R>(a, it, lw);
        it -= n_wave_rows;
        seq.init_static(it, nw, nsets);
    }
    if (a.dbg_classes & 4) role_small<VEC, G, R>(a, seq, lw);
}

// ---------------------------------------------------------------------------
// Finalize of split rows: one FIN_BLOCK-thread workgroup per row.
// ---------------------------------------------------------------------------
struct __align__(16) FinShared {
    int red[2][FIN_BLOCK / 64];
    int wave_off[FIN_BLOCK / 64];
    int nsel;
    int pad[3];
};
static_assert(sizeof(FinShared) % 16 == 0, "keep the dynamic LDS base 16-byte aligned");
static_assert(WAVE_T == 128 && SMALL_T == 16, "wave_select / role_small LDS layouts assume these");

__device__ __forceinline__ int block_count(int c, FinShared &sh, int &parity)
{
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    c = wave_sum_i(c);
    if (lane == 0) sh.red[parity][wave] = c;
    __syncthreads();
    int tot = 0;
#pragma unroll
    for (int w = 0; w < FIN_BLOCK / 64; ++w) tot += sh.red[parity][w];
    parity ^= 1;
    return tot;
}

template <int VEC, int G, int R>
__global__ __launch_bounds__(FIN_BLOCK) void k_agg_fin(const FwdArgs a, int lds_scores)
{
    using RowT = Row<VEC, G, R>;
    constexpr int NG = 64 / G;
    constexpr int NW = FIN_BLOCK / 64;
    extern __shared__ __align__(16) unsigned char dyn[];
    __shared__ FinShared sh;
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const int gid = lane / G, lg = lane % G;
    const int p = blockIdx.x;
    const int i = a.rperm[p];
    const int rs = a.rowptr[i];
    const int deg = a.rowptr[i + 1] - rs;
    const bool emit = a.sel_src != nullptr && a.k >= 0;
    const bool rank = a.k >= 0 && deg > a.k;
    const int kk = a.k < 0 ? 0 : min(a.k, deg);

    // dynamic LDS: [C * NW] partial rows | [kk] kept list | [lds_scores] scores
    float *s_part = reinterpret_cast<float *>(dyn);
    int *s_list = reinterpret_cast<int *>(s_part + (size_t)a.C * NW);
    float *s_scl = reinterpret_cast<float *>(s_list + (a.k < 0 ? 0 : a.k));
    const int t0 = a.split_task0[p], t1 = a.split_task0[p + 1];

    if (!rank && !emit) {
        // streaming row: add the tasks' partial rows in task order
        for (int c = tid; c < a.C; c += FIN_BLOCK) {
            float s = 0.f;
            for (int t = t0; t < t1; ++t) s += a.partial[(size_t)t * a.C + c];
            a.out[(size_t)i * a.C + c] = s / (float)deg;
        }
        return;
    }

    const float *g_sc = a.scores + a.split_soff[p];
    const float *sc = g_sc;
    if (deg <= lds_scores) {
        for (int e = tid; e < deg; e += FIN_BLOCK) s_scl[e] = g_sc[e];
        sc = s_scl;
    }
    __syncthreads();

    int parity = 0;
    int c = 0;
    for (int e = tid; e < deg; e += FIN_BLOCK) c += (sc[e] >= a.thr);
    const int cnt_thr = block_count(c, sh, parity);
    unsigned long long T = 0;
    const bool thr_only = cnt_thr <= a.k;
    if (!thr_only) {
        int lowbits = 1;
        while ((1 << lowbits) < deg) ++lowbits;
        for (int b = 63; b >= 0; --b) {
            if (b == 31) { T |= ~((1ull << lowbits) - 1ull) & 0xFFFFFFFFull; b = lowbits - 1; }
            const unsigned long long cand = T | (1ull << b);
            c = 0;
            for (int e = tid; e < deg; e += FIN_BLOCK) c += (sel_key(sc[e], e) >= cand);
            if (block_count(c, sh, parity) >= a.k) T = cand;
        }
    }
    // ordered compaction of the kept edges (+ wsel)
    if (tid == 0) sh.nsel = 0;
    __syncthreads();
    for (int base = 0; base < deg; base += FIN_BLOCK) {
        const int e = base + tid;
        const float s = e < deg ? sc[e] : 0.f;
        const bool kept = e < deg && (thr_only ? (s >= a.thr) : (sel_key(s, e) >= T));
        const unsigned long long m = __ballot(kept);
        if (lane == 0) sh.wave_off[wave] = __popcll(m);
        __syncthreads();
        int off = sh.nsel;
        for (int w = 0; w < wave; ++w) off += sh.wave_off[w];
        if (kept) s_list[off + prefix_popc(m)] = e;
        if (rank && a.wsel && e < deg) a.wsel[rs + e] = kept ? s : SNGNN_UNSELECTED;
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < NW; ++w) tot += sh.wave_off[w];
            sh.nsel += tot;
        }
        __syncthreads();
    }
    const int nsel = sh.nsel;

    if (emit) {
        for (int q = tid; q < nsel; q += FIN_BLOCK) {
            const int idx = s_list[q];
            const unsigned long long kq = sel_key(sc[idx], idx);
            int rk = 0;
            for (int r = 0; r < nsel; ++r) {
                const int ir = s_list[r];
                rk += sel_key(sc[ir], ir) > kq;
            }
            a.sel_src[(size_t)i * a.k + rk] = a.col[rs + idx];
            a.sel_w[(size_t)i * a.k + rk] = sc[idx];
        }
    }

    RowT acc;
    acc.zero();
    if (rank) {
        for (int q0 = 0; q0 < nsel; q0 += NW * NG) {
            const int q = q0 + wave * NG + gid;
            if (q < nsel) {
                const int idx = s_list[q];
                RowT x;
                x.load(a.h + (size_t)a.col[rs + idx] * a.C, a.C, lg);
                acc.axpy(sc[idx], x);
            }
        }
        acc.reduce_across_groups();
        if (gid == 0) acc.store(s_part + (size_t)wave * a.C, a.C, lg);
        __syncthreads();
        for (int ch = tid; ch < a.C; ch += FIN_BLOCK) {
            float s = 0.f;
            for (int w = 0; w < NW; ++w) s += s_part[(size_t)w * a.C + ch];
            a.out[(size_t)i * a.C + ch] = s / (float)deg;
        }
    } else {
        // emit on a streaming row: the sum itself still comes from the partials
        for (int ch = tid; ch < a.C; ch += FIN_BLOCK) {
            float s = 0.f;
            for (int t = t0; t < t1; ++t) s += a.partial[(size_t)t * a.C + ch];
            a.out[(size_t)i * a.C + ch] = s / (float)deg;
        }
    }
    (void)kk;
}

// ---------------------------------------------------------------------------
// Finalize of split rows from chunk-local candidates (top_k <= CAND_MAX_K):
// one 256-thread workgroup per row merges the tasks' candidate keys 128 at a
// time (tournament of wave-level top-k) and gathers the <= k winners.
// ---------------------------------------------------------------------------
constexpr int FIN_WAVE_MIN_ROWS = 2048;   // fewer moderate split rows than this: one launch (k_agg_fin_cand) for all
constexpr int FINC_BLOCK = 1024, FINC_WAVES = FINC_BLOCK / 64;   // 8 waves: the tournament's first level runs wide

template <int VEC, int G, int R>
__global__ __launch_bounds__(FINC_BLOCK) void k_agg_fin_cand(const FwdArgs a, int max_slots)
{
    using RowT = Row<VEC, G, R>;
    constexpr int NG = 64 / G;
    extern __shared__ __align__(16) unsigned char dyn[];   // no static LDS in front of it
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const int gid = lane / G, lg = lane % G;
    const int p = blockIdx.x;
    const int4 d = a.rdesc[p];
    const int i = d.x, rs = d.y, deg = d.z;
    const int t0 = a.split_task0[p], t1 = a.split_task0[p + 1];
    const bool emit = a.sel_src != nullptr && a.k >= 0;
    const bool rank = a.k >= 0 && deg > a.k;

    // dynamic LDS: [FINC_WAVES * C] partial rows | keysA | keysB [max_slots] | srcA | srcB | count
    float *s_part = reinterpret_cast<float *>(dyn);
    unsigned long long *kA = reinterpret_cast<unsigned long long *>(s_part + (size_t)FINC_WAVES * a.C + ((FINC_WAVES * a.C) & 1));
    unsigned long long *kB = kA + max_slots;
    int *sA = reinterpret_cast<int *>(kB + max_slots);
    int *sB = sA + max_slots;
    int &s_n = sB[max_slots];

    if (!rank) {
        // streaming row (deg <= top_k or no selection): add the tasks' partial rows
        for (int c = tid; c < a.C; c += FINC_BLOCK) {
            float s = 0.f;
            for (int t = t0; t < t1; ++t) s += a.partial[(size_t)t * a.C + c];
            a.out[(size_t)i * a.C + c] = s / (float)deg;
        }
        if (emit) {
            // selection of a streaming split row: every edge >= thr, ranked.  Rare
            // (needs top_k >= deg > WAVE_T); done by plain counting from the scratch scores.
            const float *sc = a.scores + a.split_soff[p];
            for (int e = tid; e < deg; e += FINC_BLOCK) {
                const float se = sc[e];
                if (!(se >= a.thr)) continue;
                int rk = 0;
                for (int b = 0; b < deg; ++b) rk += (sc[b] > se) || (sc[b] == se && b < e);
                a.sel_src[(size_t)i * a.k + rk] = a.col[rs + e];
                a.sel_w[(size_t)i * a.k + rk] = se;
            }
        }
        return;
    }

    int n = (t1 - t0) * a.k;
    for (int q = tid; q < n; q += FINC_BLOCK) {
        kA[q] = a.cand_key[(size_t)t0 * a.k + q];
        sA[q] = a.cand_src[(size_t)t0 * a.k + q];
    }
    __syncthreads();
    while (n > 128) {
        const int groups = (n + 127) / 128;
        for (int g = wave; g < groups; g += FINC_WAVES) {
            const int q0 = g * 128 + lane, q1 = q0 + 64;
            const unsigned long long key0 = q0 < n ? kA[q0] : 0ull, key1 = q1 < n ? kA[q1] : 0ull;
            bool k0, k1;
            wave_topk_keys(key0, key1, a.k, a.lowbits, k0, k1);
            const unsigned long long m0 = __ballot(k0), m1 = __ballot(k1);
            const int n0 = __popcll(m0), ns = n0 + __popcll(m1);
            if (k0) { const int o = g * a.k + prefix_popc(m0); kB[o] = key0; sB[o] = sA[q0]; }
            if (k1) { const int o = g * a.k + n0 + prefix_popc(m1); kB[o] = key1; sB[o] = sA[q1]; }
            if (lane >= ns && lane < a.k) kB[g * a.k + lane] = 0ull;
        }
        __syncthreads();
        n = groups * a.k;
        unsigned long long *t = kA; kA = kB; kB = t;
        int *ts = sA; sA = sB; sB = ts;
    }
    if (wave == 0) {
        const unsigned long long key0 = lane < n ? kA[lane] : 0ull;
        const unsigned long long key1 = lane + 64 < n ? kA[lane + 64] : 0ull;
        bool k0, k1;
        wave_topk_keys(key0, key1, a.k, a.lowbits, k0, k1);
        const unsigned long long m0 = __ballot(k0), m1 = __ballot(k1);
        const int n0 = __popcll(m0);
        if (k0) { const int o = prefix_popc(m0); kB[o] = key0; sB[o] = sA[lane]; }
        if (k1) { const int o = n0 + prefix_popc(m1); kB[o] = key1; sB[o] = sA[lane + 64]; }
        if (lane == 0) s_n = n0 + __popcll(m1);
    }
    __syncthreads();
    const int nsel = s_n;
    const unsigned long long *win = kB;
    const int *wsrc = sB;

    // gather first (the long latency), bookkeeping stores behind it
    RowT acc;
    acc.zero();
    for (int q0 = 0; q0 < nsel; q0 += FINC_WAVES * NG) {
        const int q = q0 + wave * NG + gid;
        if (q < nsel) {
            RowT x;
            x.load(a.h + (size_t)wsrc[q] * a.C, a.C, lg);
            acc.axpy(key_score(win[q]), x);
        }
    }
    for (int q = tid; q < nsel; q += FINC_BLOCK) {
        const unsigned long long kq = win[q];
        const float sq = key_score(kq);
        if (a.wsel) a.wsel[rs + key_index(kq)] = sq;
        if (emit) {
            int rk = 0;
            for (int r = 0; r < nsel; ++r) rk += win[r] > kq;
            a.sel_src[(size_t)i * a.k + rk] = wsrc[q];
            a.sel_w[(size_t)i * a.k + rk] = sq;
        }
    }
    acc.reduce_across_groups();
    if (gid == 0) acc.store(s_part + (size_t)wave * a.C, a.C, lg);
    __syncthreads();
    for (int ch = tid; ch < a.C; ch += FINC_BLOCK) {
        float s = 0.f;
        for (int w = 0; w < FINC_WAVES; ++w) s += s_part[(size_t)w * a.C + ch];
        a.out[(size_t)i * a.C + ch] = s / (float)deg;
    }
}

// The same finalize for split rows whose candidates fit one wave-level selection
// ((tasks) * k <= 128, i.e. deg <= 8 * CHUNK at k = 16): one WAVE per row, no workgroup
// barrier, 4 rows per workgroup.  On graphs with many moderately large rows
// (products-like: ~10^5 split rows) the 1024-thread tournament above is mostly idle.
template <int VEC, int G, int R>
__global__ __launch_bounds__(BLOCK) void k_agg_fin_wave(const FwdArgs a, int first, int count)
{
    using RowT = Row<VEC, G, R>;
    constexpr int NG = 64 / G;
    __shared__ unsigned long long s_key[WAVES][CAND_MAX_K];
    __shared__ int s_src[WAVES][CAND_MAX_K];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int gid = lane / G, lg = lane % G;
    const int q = blockIdx.x * WAVES + wave;
    if (q >= count) return;                                   // wave-uniform
    const int p = first + q;
    const int4 d = a.rdesc[p];
    const int i = d.x, rs = d.y, deg = d.z;
    const int t0 = a.split_task0[p], t1 = a.split_task0[p + 1];
    if (a.k < 0) {
        for (int c = lane; c < a.C; c += 64) {
            float s = 0.f;
            for (int t = t0; t < t1; ++t) s += a.partial[(size_t)t * a.C + c];
            a.out[(size_t)i * a.C + c] = s / (float)deg;
        }
        return;
    }
    const bool emit = a.sel_src != nullptr;
    const int n = (t1 - t0) * a.k;                            // <= 128 by the launch split
    const size_t c0 = (size_t)t0 * a.k;
    const unsigned long long key0 = lane < n ? a.cand_key[c0 + lane] : 0ull;
    const unsigned long long key1 = lane + 64 < n ? a.cand_key[c0 + lane + 64] : 0ull;
    const int src0 = lane < n ? a.cand_src[c0 + lane] : 0;
    const int src1 = lane + 64 < n ? a.cand_src[c0 + lane + 64] : 0;
    bool k0, k1;
    wave_topk_keys(key0, key1, a.k, a.lowbits, k0, k1);
    const unsigned long long m0 = __ballot(k0), m1 = __ballot(k1);
    const int n0 = __popcll(m0), nsel = n0 + __popcll(m1);
    if (k0) { const int o = prefix_popc(m0); s_key[wave][o] = key0; s_src[wave][o] = src0; }
    if (k1) { const int o = n0 + prefix_popc(m1); s_key[wave][o] = key1; s_src[wave][o] = src1; }
    wave_lds_sync();
    RowT acc;
    acc.zero();
    constexpr int GU = R >= 4 ? 2 : 4;                        // winner rows in flight per lane group
    for (int w0 = 0; w0 < nsel; w0 += GU * NG) {
        RowT x[GU];
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int w = w0 + u * NG + gid;
            if (w < nsel) x[u].load(a.h + (size_t)s_src[wave][w] * a.C, a.C, lg);
        }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int w = w0 + u * NG + gid;
            if (w < nsel) acc.axpy(key_score(s_key[wave][w]), x[u]);
        }
    }
    if (lane < nsel) {
        const unsigned long long kq = s_key[wave][lane];
        const float sq = key_score(kq);
        if (a.wsel) a.wsel[rs + key_index(kq)] = sq;
        if (emit) {
            int rk = 0;
            for (int r = 0; r < nsel; ++r) rk += s_key[wave][r] > kq;
            a.sel_src[(size_t)i * a.k + rk] = s_src[wave][lane];
            a.sel_w[(size_t)i * a.k + rk] = sq;
        }
    }
    acc.reduce_across_groups();
    acc.div((float)deg);
    if (gid == 0) acc.store(a.out + (size_t)i * a.C, a.C, lg);
}

template <int VEC, int G, int R>
int launch_agg_fwd(const FwdArgs &a0, int max_split_deg, hipEvent_t *ev, hipStream_t st)
{
    constexpr int RPW = 64 / G;
    FwdArgs a = a0;
    const int n_small = a.N - a.n_med_end;
    const int64_t items = (int64_t)a.n_tasks + (a.n_med_end - a.n_split) + ceil_div(n_small, RPW);
    // persistent grid: what the chip holds at the kernel's occupancy, or less
    const int bpc = a.dbg_blocks_per_cu > 0 ? a.dbg_blocks_per_cu : FWD_WAVES_PER_SIMD;
    const int grid = (int)std::min<int64_t>(ceil_div(items, WAVES), 256 * bpc);
    if (SNGNN_EXPERIMENTAL && a.dynamic) SN_HIP(hipMemsetAsync(a.dyn_ctr, 0, 2 * DYN_SHARDS * DYN_CTR_STRIDE * sizeof(int32_t), st));
    if (ev) SN_HIP(hipEventRecord(ev[0], st));
    if (grid > 0) k_agg_fwd<VEC, G, R><<<grid, BLOCK, 0, st>>>(a);
    if (ev) SN_HIP(hipEventRecord(ev[1], st));
    const bool any_streaming_split = a.k < 0 || a.k > WAVE_T;   // split rows have deg > WAVE_T
    if (SNGNN_EXPERIMENTAL && a.n_split > 0 && a.k <= CAND_MAX_K && !any_streaming_split && a.inkernel_fin) {
        // every split row is finalized inside k_agg_fwd by its last-arriving task
    } else if (a.n_split > 0 && (a.k < 0 || a.k <= CAND_MAX_K)) {
        // streaming rows and candidate tournament
        const int max_tasks = ceil_div(max_split_deg, CHUNK);
        const int max_slots = std::max(1, max_tasks * std::max(a.k, 0));
        const size_t dyn = ((size_t)FINC_WAVES * a.C + 1) * 4 + (size_t)max_slots * 24 + 16;
        if (dyn > 150 * 1024) { set_error("in-degree too large for the split-row finalize"); return SNGNN_EINVAL; }
        if (dyn > 48 * 1024)
            SN_HIP(hipFuncSetAttribute((const void *)k_agg_fin_cand<VEC, G, R>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        // rows are in descending degree order: the first n_big need the workgroup tournament,
        // the rest fit one wave-level selection
        // (a second launch only pays when there are many such rows: arxiv-like graphs have a
        // few hundred split rows in all, products-like ones 10^5)
        const int n_wave = a.n_split - std::min(a.n_split, a.n_split_gt_wave);
        const int n_big = (a.k > 0 && n_wave >= FIN_WAVE_MIN_ROWS) ? a.n_split - n_wave : (a.k > 0 ? a.n_split : 0);
        if (n_big > 0) k_agg_fin_cand<VEC, G, R><<<n_big, FINC_BLOCK, dyn, st>>>(a, max_slots);
        if (a.n_split > n_big)
            k_agg_fin_wave<VEC, G, R><<<ceil_div(a.n_split - n_big, WAVES), BLOCK, 0, st>>>(a, n_big, a.n_split - n_big);
    } else if (a.n_split > 0) {
        const size_t fixed = (size_t)a.C * (FIN_BLOCK / 64) * 4 + (size_t)(a.k < 0 ? 0 : a.k) * 4;
        const size_t budget = 120 * 1024;
        int lds_scores = 0;
        if (fixed < budget) lds_scores = (int)std::min<size_t>((budget - fixed) / 4, (size_t)max_split_deg);
        const size_t dyn = fixed + (size_t)lds_scores * 4;
        if (dyn > 150 * 1024) { set_error("top_k too large for the split-row finalize"); return SNGNN_EINVAL; }
        if (dyn > 48 * 1024)
            SN_HIP(hipFuncSetAttribute((const void *)k_agg_fin<VEC, G, R>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        k_agg_fin<VEC, G, R><<<a.n_split, FIN_BLOCK, dyn, st>>>(a, lds_scores);
    }
    if (ev) SN_HIP(hipEventRecord(ev[2], st));
    SN_HIP(hipGetLastError());
    return SNGNN_OK;
}

// one translation unit per VEC instantiates these
int launch_agg_fwd_v1(const RowCfg &cfg, const FwdArgs &a, int max_split_deg, hipEvent_t *ev,
                      hipStream_t st);
int launch_agg_fwd_v2(const RowCfg &cfg, const FwdArgs &a, int max_split_deg, hipEvent_t *ev,
                      hipStream_t st);
int launch_agg_fwd_v4(const RowCfg &cfg, const FwdArgs &a, int max_split_deg, hipEvent_t *ev,
                      hipStream_t st);

#define SNGNN_DISPATCH_GR(FN, VEC, cfg, ...)                                   \
    switch ((cfg).g * 100 + (cfg).r) {                                         \
    case 801: return FN<VEC, 8, 1>(__VA_ARGS__);                               \
    case 1601: return FN<VEC, 16, 1>(__VA_ARGS__);                             \
    case 3201: return FN<VEC, 32, 1>(__VA_ARGS__);                             \
    case 6401: return FN<VEC, 64, 1>(__VA_ARGS__);                             \
    case 6402: return FN<VEC, 64, 2>(__VA_ARGS__);                             \
    case 6404: return FN<VEC, 64, 4>(__VA_ARGS__);                             \
    case 6408: return FN<VEC, 64, 8>(__VA_ARGS__);                             \
    default: sngnn::set_error("unsupported channel layout"); return SNGNN_EINVAL; \
    }

}  // namespace sngnn
