// Graph structure build for libsngnn_hip (gfx950).
//
// Replaces the per-forward self-loop handling of the reference's conv layers
// (models/models.py:117-120, 234-236, 323: PyG add_self_loops then optionally
// remove_self_loops) and the per-target grouping that PyG propagate /
// torch_scatter do implicitly on every call, by a one-time device build of
//   * CSR by target, edges of a row in their original relative order (stable
//     radix sort on the target id) - "edge position" is the reference's
//     tie-breaker (torch_scatter scatter_max CPU: first occurrence wins),
//   * CSC by source (for the atomics-free backward),
//   * rows / sources sorted by degree, descending (work balancing),
//   * the split-row task lists for rows longer than one wave handles.
// Sorting and scanning use rocPRIM through hipCUB (setup, not the hot path).
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstring>
#include <numeric>

#include "common.h"

namespace sngnn {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
int hip_fail(hipError_t e, const char *what, const char *file, int line)
{
    g_err = std::string("HIP error: ") + hipGetErrorString(e) + " in " + what + " at " + file +
            ":" + std::to_string(line);
    return SNGNN_EHIP;
}

// candidate t in [0, E + n_loops): original edge or appended self-loop
__global__ void k_mark(const int64_t *__restrict__ ei, int64_t E, int64_t T, int64_t Ntot,
                       int64_t row0, int64_t row1, int remove_loops, int32_t *__restrict__ keep,
                       int32_t *__restrict__ bad)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    int64_t s, d;
    if (t < E) { s = ei[t]; d = ei[E + t]; } else { s = d = row0 + (t - E); }
    if (s < 0 || s >= Ntot || d < 0 || d >= Ntot) { atomicOr(bad, 1); keep[t] = 0; return; }
    // a partition keeps only the edges that point into its own node range
    // remove_loops 1: every loop goes; 2 (SNGNN_LOOPS_REPLACE): only the original ones
    const bool drop_loop = s == d && (remove_loops == 1 || (remove_loops == 2 && t < E));
    keep[t] = (drop_loop || d < row0 || d >= row1) ? 0 : 1;
}

__global__ void k_compact(const int64_t *__restrict__ ei, int64_t E, int64_t T, int64_t row0,
                          const int32_t *__restrict__ keep, const int32_t *__restrict__ pos,
                          int32_t *__restrict__ src32, int32_t *__restrict__ dst32)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T || !keep[t]) return;
    int64_t s, d;
    if (t < E) { s = ei[t]; d = ei[E + t]; } else { s = d = row0 + (t - E); }
    src32[pos[t]] = (int32_t)s;              // global source id
    dst32[pos[t]] = (int32_t)(d - row0);     // local target row
}

__global__ void k_iota(int32_t *a, int64_t n)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) a[t] = (int32_t)t;
}

// inv[perm[q]] = q
__global__ void k_invert(const int32_t *__restrict__ perm, int32_t *__restrict__ inv, int64_t n)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) inv[perm[t]] = (int32_t)t;
}

__global__ void k_gather(const int32_t *__restrict__ idx, const int32_t *__restrict__ table,
                         int32_t *__restrict__ out, int64_t n)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = table[idx[t]];
}

// ptr[i] = first position p with sorted_keys[p] >= i   (i in [0, N])
__global__ void k_lower_bound(const int32_t *__restrict__ keys, int64_t n, int64_t N,
                              int32_t *__restrict__ ptr)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > N) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t m = (lo + hi) >> 1;
        if (keys[m] < (int32_t)i) lo = m + 1; else hi = m;
    }
    ptr[i] = (int32_t)lo;
}

__global__ void k_inv_deg(const int32_t *__restrict__ rowptr, int64_t N, float *__restrict__ out)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < N) out[i] = 1.0f / (float)max(rowptr[i + 1] - rowptr[i], 1);
}

__global__ void k_row_desc(const int32_t *__restrict__ rperm, const int32_t *__restrict__ rowptr,
                           int64_t N, int4 *__restrict__ desc)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    const int i = rperm[p];
    const int rs = rowptr[i];
    desc[p] = make_int4(i, rs, rowptr[i + 1] - rs, 0);
}

// source slice (of N_SLICE equal node ranges) of the middle edge of every split-row task
constexpr int N_SLICE = 8;      // = XCDs: one slice of a gathered table per L2
__global__ void k_task_slice(const int4 *__restrict__ rdesc, const int32_t *__restrict__ col,
                             const int32_t *__restrict__ task_slot, const int32_t *__restrict__ task_chunk,
                             int n_tasks, int64_t Ntot, int32_t *__restrict__ slice)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tasks) return;
    const int4 d = rdesc[task_slot[t]];
    const int e0 = task_chunk[t] * CHUNK, e1 = min(d.z, e0 + CHUNK);
    const int64_t j = col[d.y + (e0 + e1) / 2];
    slice[t] = (int32_t)min<int64_t>(j * N_SLICE / max<int64_t>(Ntot, 1), N_SLICE - 1);
}

// sort key of a source: its out-degree, but every small source (<= SMALL_T) the same - the
// stable sort then leaves the small sources in their natural order: the passes that walk the
// sources (backward pass S, the ++ branch) touch one or a few rows of h / dnT / grad_h per
// work item and gather little, so consecutive node ids = full cache lines matter more than
// equal degrees inside a wave (k_bwd_s 42.2 -> 36.9 us at arxiv size).  The TARGET rows stay
// sorted by degree: the forward issues loads up to the longest row of a set, and mixed
// degrees cost it more (main kernel 54.9 -> 60.1 us) than the line-aligned rows save.
// Among the small sources the stable sort makes two runs: first those the backward's pass S
// walks (key SMALL_T), then the FUSED nodes (key SMALL_T - 1): owned nodes small both as source
// and as target, whose two backward passes run as one work item (agg_bwd_impl.h: f_role_node).
__global__ void k_class_key(int32_t *__restrict__ deg, int64_t Ntot, const int32_t *__restrict__ rowptr,
                            int64_t row0, int64_t N)
{
    int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= Ntot) return;
    const int od = deg[v];
    if (od > SMALL_T) return;
    const int64_t vl = v - row0;
    const bool fused = vl >= 0 && vl < N && rowptr[vl + 1] - rowptr[vl] <= SMALL_T;
    deg[v] = fused ? SMALL_T - 1 : SMALL_T;
}

// 1 for an owned small target that is not fused (its out-list is long), else 0
__global__ void k_trest_key(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ cscptr, int64_t row0,
                            int64_t N, int32_t *__restrict__ key)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int64_t v = i + row0;
    key[i] = (rowptr[i + 1] - rowptr[i] <= SMALL_T && cscptr[v + 1] - cscptr[v] > SMALL_T) ? 1 : 0;
}

// descriptors of the fused nodes from their source descriptors {node, first CSC entry, out-degree}
__global__ void k_fused_desc(const int4 *__restrict__ sdesc, int64_t n, const int32_t *__restrict__ rowptr,
                             int64_t row0, int4 *__restrict__ fdesc)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int4 s = sdesc[p];
    const int64_t vl = s.x - row0;
    const int rs = rowptr[vl];
    fdesc[p] = make_int4(s.x, rs, s.y, (rowptr[vl + 1] - rs) | (s.z << 8));
}

// col with the rows in SLOT order: the in-edges of slot p follow those of slot p - 1, so a
// kernel that walks the slots reads the column ids of consecutive (short) rows as one
// stream instead of one 128-byte line per row.  off[p] = first entry of slot p (also left in
// rdesc[p].w); one thread per entry finds its slot by bisection.
__global__ void k_slot_col(int4 *__restrict__ desc, const int32_t *__restrict__ off, int64_t N, int64_t Ep,
                           const int32_t *__restrict__ col, int32_t *__restrict__ col_s)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < N) desc[q].w = off[q];
    if (q >= Ep) return;
    int64_t lo = 0, hi = N;                    // last slot with off[slot] <= q
    while (hi - lo > 1) {
        const int64_t m = (lo + hi) / 2;
        if (off[m] <= q) lo = m; else hi = m;
    }
    col_s[q] = col[desc[lo].y + (int)(q - off[lo])];
}

// sort key of the forward's STREAMING row order: a small row (<= SMALL_T) by its bucket of 4
// degrees - the small-row loop takes 4 edges per step, so the rows of a bucket cost the same
// number of steps - and, the sort being stable, in natural order inside the bucket
__global__ void k_bucket_key(const int32_t *__restrict__ deg, int64_t N, int32_t *__restrict__ key)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) key[i] = deg[i] > SMALL_T ? deg[i] : (deg[i] + 3) / 4 * 4;
}

// bit index of every CSC entry's edge in the forward-written kept-bit layout (common.h: kbits)
__global__ void k_csc_bit(const int32_t *__restrict__ csc_eid, const int32_t *__restrict__ csc_dst,
                          const int32_t *__restrict__ rowptr, const int32_t *__restrict__ rslot,
                          const int32_t *__restrict__ split_task0, int64_t Ep, int n_split, int n_med_end,
                          int64_t wbase, int64_t tbase, int32_t *__restrict__ csc_bit)
{
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Ep) return;
    const int i = csc_dst[q];
    const int t = csc_eid[q] - rowptr[i];
    const int slot = rslot[i];
    int64_t b;
    if (slot >= n_med_end) b = 16 * (int64_t)i + t;
    else if (slot >= n_split) b = 32 * wbase + 128 * (int64_t)(slot - n_split) + t;
    else b = 32 * tbase + 128 * (int64_t)split_task0[slot] + t;
    csc_bit[q] = (int32_t)b;
}

__global__ void k_degree(const int32_t *__restrict__ ptr, int64_t N, int32_t *__restrict__ deg)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) deg[i] = ptr[i + 1] - ptr[i];
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 4) == hipSuccess ? 0 : SNGNN_ENOMEM; }
    template <class T> T *as() { return (T *)p; }
};

static inline dim3 grid1(int64_t n, int b = 256) { return dim3((unsigned)((n + b - 1) / b)); }

static int bits_for(int64_t n)
{
    int b = 1;
    while (((int64_t)1 << b) < n && b < 31) ++b;
    return b;
}

// stable sort of (key, value) pairs by key ascending / descending
static int sort_pairs(const int32_t *kin, int32_t *kout, const int32_t *vin, int32_t *vout,
                      int64_t n, int end_bit, bool descending, hipStream_t st)
{
    size_t tmp_bytes = 0;
    if (descending)
        SN_HIP(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tmp_bytes, kin, kout, vin, vout,
                                                           (int)n, 0, end_bit, st));
    else
        SN_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, kin, kout, vin, vout, (int)n,
                                                  0, end_bit, st));
    DevBuf tmp;
    if (tmp.alloc(tmp_bytes)) { set_error("out of device memory (sort scratch)"); return SNGNN_ENOMEM; }
    if (descending)
        SN_HIP(hipcub::DeviceRadixSort::SortPairsDescending(tmp.p, tmp_bytes, kin, kout, vin, vout,
                                                           (int)n, 0, end_bit, st));
    else
        SN_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, kin, kout, vin, vout, (int)n, 0,
                                                  end_bit, st));
    SN_HIP(hipStreamSynchronize(st));
    return 0;
}

template <class T> static int dev_alloc(T **p, int64_t n)
{
    if (hipMalloc((void **)p, (size_t)std::max<int64_t>(n, 1) * sizeof(T)) != hipSuccess) {
        set_error("out of device memory (graph arrays)");
        return SNGNN_ENOMEM;
    }
    return 0;
}

// Task lists for rows (or sources) whose degree exceeds WAVE_T: the first
// n_split slots of the degree-sorted permutation, CHUNK edges per task.
static int build_tasks(const std::vector<int32_t> &deg_desc, int n_split, int32_t **task_slot,
                       int32_t **task_chunk, int32_t **task0, int32_t **soff, int *n_tasks,
                       int64_t *split_edges)
{
    std::vector<int32_t> slot, chunk, t0(n_split + 1, 0), so(n_split + 1, 0);
    int64_t off = 0;
    for (int p = 0; p < n_split; ++p) {
        int nch = (deg_desc[p] + CHUNK - 1) / CHUNK;
        t0[p] = (int32_t)slot.size();
        so[p] = (int32_t)off;
        for (int c = 0; c < nch; ++c) { slot.push_back(p); chunk.push_back(c); }
        off += deg_desc[p];
    }
    t0[n_split] = (int32_t)slot.size();
    so[n_split] = (int32_t)off;
    *n_tasks = (int)slot.size();
    if (split_edges) *split_edges = off;
    int rc;
    if ((rc = dev_alloc(task_slot, (int64_t)slot.size()))) return rc;
    if ((rc = dev_alloc(task_chunk, (int64_t)slot.size()))) return rc;
    if ((rc = dev_alloc(task0, n_split + 1))) return rc;
    if (soff && (rc = dev_alloc(soff, n_split + 1))) return rc;
    if (!slot.empty()) {
        SN_HIP(hipMemcpy(*task_slot, slot.data(), slot.size() * 4, hipMemcpyHostToDevice));
        SN_HIP(hipMemcpy(*task_chunk, chunk.data(), chunk.size() * 4, hipMemcpyHostToDevice));
    }
    SN_HIP(hipMemcpy(*task0, t0.data(), t0.size() * 4, hipMemcpyHostToDevice));
    if (soff) SN_HIP(hipMemcpy(*soff, so.data(), so.size() * 4, hipMemcpyHostToDevice));
    return 0;
}

static int build(sngnn_graph *g, const int64_t *ei, int64_t E, int64_t Ntot, int64_t row0,
                 int64_t row1, int add_loops, int remove_loops, hipStream_t st)
{
    const int64_t N = row1 - row0;                                   // owned target rows
    const int64_t n_loops = (add_loops && remove_loops != 1) ? N : 0;   // add+remove == remove
    const int64_t T = E + n_loops;
    SN_REQUIRE(T < ((int64_t)1 << 31) - 1 && Ntot < ((int64_t)1 << 31) - 1, SNGNN_EINVAL,
               "graph too large for 32-bit indices");
    SN_HIP(hipGetDevice(&g->device));
    g->N = N; g->Ntot = Ntot; g->row_off = row0;
    g->E_in = E; g->add_loops = add_loops; g->remove_loops = remove_loops;
    int rc;

    // 1. mark + validate, 2. scan, 3. compact to int32 (src, dst) in list order
    DevBuf keep, pos, bad, src32, dst32;
    if (keep.alloc((size_t)T * 4) || pos.alloc((size_t)T * 4) || bad.alloc(4)) {
        set_error("out of device memory (graph build)");
        return SNGNN_ENOMEM;
    }
    SN_HIP(hipMemsetAsync(bad.p, 0, 4, st));
    int64_t Ep = 0;
    if (T > 0) {
        k_mark<<<grid1(T), 256, 0, st>>>(ei, E, T, Ntot, row0, row1, remove_loops,
                                         keep.as<int32_t>(), bad.as<int32_t>());
        size_t tmp_bytes = 0;
        SN_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, keep.as<int32_t>(),
                                                pos.as<int32_t>(), (int)T, st));
        DevBuf tmp;
        if (tmp.alloc(tmp_bytes)) { set_error("out of device memory (scan scratch)"); return SNGNN_ENOMEM; }
        SN_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tmp_bytes, keep.as<int32_t>(),
                                                pos.as<int32_t>(), (int)T, st));
        int32_t last_pos = 0, last_keep = 0, h_bad = 0;
        SN_HIP(hipMemcpyAsync(&last_pos, pos.as<int32_t>() + (T - 1), 4, hipMemcpyDeviceToHost, st));
        SN_HIP(hipMemcpyAsync(&last_keep, keep.as<int32_t>() + (T - 1), 4, hipMemcpyDeviceToHost, st));
        SN_HIP(hipMemcpyAsync(&h_bad, bad.p, 4, hipMemcpyDeviceToHost, st));
        SN_HIP(hipStreamSynchronize(st));
        SN_REQUIRE(!h_bad, SNGNN_ERANGE, "edge_index contains a node id outside [0, N)");
        Ep = (int64_t)last_pos + last_keep;
    }
    g->Ep = Ep;
    if (src32.alloc((size_t)Ep * 4) || dst32.alloc((size_t)Ep * 4)) {
        set_error("out of device memory (graph build)");
        return SNGNN_ENOMEM;
    }
    if (T > 0)
        k_compact<<<grid1(T), 256, 0, st>>>(ei, E, T, row0, keep.as<int32_t>(), pos.as<int32_t>(),
                                            src32.as<int32_t>(), dst32.as<int32_t>());

    if ((rc = dev_alloc(&g->rowptr, N + 1)) || (rc = dev_alloc(&g->col, Ep)) ||
        (rc = dev_alloc(&g->eid, Ep)) || (rc = dev_alloc(&g->cscptr, Ntot + 1)) ||
        (rc = dev_alloc(&g->csc_eid, Ep)) || (rc = dev_alloc(&g->csc_dst, Ep)) || (rc = dev_alloc(&g->csc_pos, Ep)) ||
        (rc = dev_alloc(&g->rperm, N)) || (rc = dev_alloc(&g->sperm, Ntot)))
        return rc;

    const int nbits = bits_for(std::max<int64_t>(Ntot, 2));
    DevBuf iota, keys_sorted, deg, deg_sorted;
    if (iota.alloc((size_t)std::max(Ep, Ntot) * 4) || keys_sorted.alloc((size_t)std::max(Ep, Ntot) * 4) ||
        deg.alloc((size_t)Ntot * 4) || deg_sorted.alloc((size_t)Ntot * 4)) {
        set_error("out of device memory (graph build)");
        return SNGNN_ENOMEM;
    }
    int32_t *d_iota = iota.as<int32_t>(), *d_keys = keys_sorted.as<int32_t>();
    if (std::max(Ep, Ntot) > 0) k_iota<<<grid1(std::max(Ep, Ntot)), 256, 0, st>>>(d_iota, std::max(Ep, Ntot));

    // 4. CSR by target: stable sort of list positions by dst
    if (Ep > 0) {
        if ((rc = sort_pairs(dst32.as<int32_t>(), d_keys, d_iota, g->eid, Ep, nbits, false, st))) return rc;
        k_gather<<<grid1(Ep), 256, 0, st>>>(g->eid, src32.as<int32_t>(), g->col, Ep);
    }
    k_lower_bound<<<grid1(N + 1), 256, 0, st>>>(d_keys, Ep, N, g->rowptr);
    DevBuf dst_csr;   // target of each CSR edge (== sorted keys); keep a copy for CSC
    if (dst_csr.alloc((size_t)Ep * 4)) { set_error("out of device memory (graph build)"); return SNGNN_ENOMEM; }
    if (Ep > 0) SN_HIP(hipMemcpyAsync(dst_csr.p, d_keys, (size_t)Ep * 4, hipMemcpyDeviceToDevice, st));

    // 5. rows by in-degree, descending (stable => ascending row id inside a degree)
    g->rdeg.assign((size_t)N, 0);
    if (N > 0) {
        k_degree<<<grid1(N), 256, 0, st>>>(g->rowptr, N, deg.as<int32_t>());
        if ((rc = sort_pairs(deg.as<int32_t>(), deg_sorted.as<int32_t>(), d_iota, g->rperm, N, 31, true, st)))
            return rc;
        SN_HIP(hipMemcpy(g->rdeg.data(), deg_sorted.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    }
    g->max_in_deg = N ? g->rdeg[0] : 0;
    {   // prefix sums over the rows above the small class (host; the filter's rule reads them per call)
        const int m = g->rows_gt(SMALL_T);
        g->rdeg_wave_psum.assign((size_t)m + 1, 0);
        for (int p = 0; p < m; ++p) g->rdeg_wave_psum[(size_t)p + 1] = g->rdeg_wave_psum[(size_t)p] + g->rdeg[(size_t)p];
    }
    // descriptors + slot-ordered column ids of a row order (perm, its degrees in slot order)
    auto describe = [&](const int32_t *perm, const std::vector<int32_t> &deg_slot, int4 **desc, int32_t **col_s) -> int {
        int rc2;
        if ((rc2 = dev_alloc(desc, N)) || (rc2 = dev_alloc(col_s, Ep))) return rc2;
        if (N > 0) k_row_desc<<<grid1(N), 256, 0, st>>>(perm, g->rowptr, N, *desc);
        std::vector<int32_t> off((size_t)N + 1, 0);
        for (int64_t p = 0; p < N; ++p) off[p + 1] = off[p] + deg_slot[p];
        DevBuf d_off;
        if (d_off.alloc(((size_t)N + 1) * 4)) { set_error("out of device memory (graph build)"); return SNGNN_ENOMEM; }
        SN_HIP(hipMemcpyAsync(d_off.p, off.data(), off.size() * 4, hipMemcpyHostToDevice, st));
        const int64_t nthreads = std::max<int64_t>(N, Ep);
        if (nthreads > 0)
            k_slot_col<<<grid1(nthreads), 256, 0, st>>>(*desc, d_off.as<int32_t>(), N, Ep, g->col, *col_s);
        SN_HIP(hipStreamSynchronize(st));      // off[] (host) and d_off go out of scope
        return 0;
    };
    if ((rc = describe(g->rperm, g->rdeg, &g->rdesc, &g->col_s))) return rc;
    // 5b. the forward's order for calls that STREAM the small rows (no ranking there: top_k >=
    //     SMALL_T or no top_k): split and wave rows as above, small rows by 4-degree bucket and
    //     natural order inside - rows 4x denser in memory per set at the same number of steps
    //     (main kernel 53.4 -> 52.3 us at config 4).  Calls that rank inside the small rows
    //     (top_k < SMALL_T) keep the exact order: mixed degrees cost their rank loops more
    //     (top_k = 1: 49.9 -> 51.3 us with this order).
    if (N > 0) {
        k_bucket_key<<<grid1(N), 256, 0, st>>>(deg.as<int32_t>(), N, d_keys);
        if ((rc = dev_alloc(&g->rperm_b, N))) return rc;
        if ((rc = sort_pairs(d_keys, deg_sorted.as<int32_t>(), d_iota, g->rperm_b, N, 31, true, st))) return rc;
        k_gather<<<grid1(N), 256, 0, st>>>(g->rperm_b, deg.as<int32_t>(), deg_sorted.as<int32_t>(), N);   // true degrees, slot order
        std::vector<int32_t> rdeg_b((size_t)N);
        SN_HIP(hipMemcpy(rdeg_b.data(), deg_sorted.p, (size_t)N * 4, hipMemcpyDeviceToHost));
        if ((rc = describe(g->rperm_b, rdeg_b, &g->rdesc_b, &g->col_s_b))) return rc;
    }
    if ((rc = dev_alloc(&g->inv_deg, N))) return rc;
    if (N > 0) k_inv_deg<<<grid1(N), 256, 0, st>>>(g->rowptr, N, g->inv_deg);

    // 6. CSC by source: stable sort of CSR positions by col
    if (Ep > 0) {
        if ((rc = sort_pairs(g->col, d_keys, d_iota, g->csc_eid, Ep, nbits, false, st))) return rc;
        k_gather<<<grid1(Ep), 256, 0, st>>>(g->csc_eid, dst_csr.as<int32_t>(), g->csc_dst, Ep);
        k_invert<<<grid1(Ep), 256, 0, st>>>(g->csc_eid, g->csc_pos, Ep);
        int32_t mn = 0;
        SN_HIP(hipMemcpy(&mn, d_keys, 4, hipMemcpyDeviceToHost));
        g->src_min = mn;
    }
    k_lower_bound<<<grid1(Ntot + 1), 256, 0, st>>>(d_keys, Ep, Ntot, g->cscptr);
    g->sdeg.assign((size_t)Ntot, 0);
    if (Ntot > 0) {
        k_degree<<<grid1(Ntot), 256, 0, st>>>(g->cscptr, Ntot, deg.as<int32_t>());
        k_class_key<<<grid1(Ntot), 256, 0, st>>>(deg.as<int32_t>(), Ntot, g->rowptr, row0, N);
        if ((rc = sort_pairs(deg.as<int32_t>(), deg_sorted.as<int32_t>(), d_iota, g->sperm, Ntot, 31, true, st)))
            return rc;
        SN_HIP(hipMemcpy(g->sdeg.data(), deg_sorted.p, (size_t)Ntot * 4, hipMemcpyDeviceToHost));
    }
    g->max_out_deg = Ntot ? g->sdeg[0] : 0;
    if ((rc = dev_alloc(&g->sdesc, Ntot))) return rc;
    if (Ntot > 0) k_row_desc<<<grid1(Ntot), 256, 0, st>>>(g->sperm, g->cscptr, Ntot, g->sdesc);
    // 6b. the backward's node-centric lists: fused nodes (the tail of sdesc) and the small
    //     targets left to pass T
    {
        const int n_srest_end = g->srcs_gt(SMALL_T - 1);
        g->n_fused = (int)Ntot - n_srest_end;
        if ((rc = dev_alloc(&g->fdesc, g->n_fused))) return rc;
        if (g->n_fused > 0)
            k_fused_desc<<<grid1(g->n_fused), 256, 0, st>>>(g->sdesc + n_srest_end, g->n_fused, g->rowptr, row0, g->fdesc);
        g->n_trest = 0;
        if (N > 0) {
            DevBuf perm_t;
            if (perm_t.alloc((size_t)N * 4)) { set_error("out of device memory (graph build)"); return SNGNN_ENOMEM; }
            k_trest_key<<<grid1(N), 256, 0, st>>>(g->rowptr, g->cscptr, row0, N, d_keys);
            if ((rc = sort_pairs(d_keys, deg_sorted.as<int32_t>(), d_iota, perm_t.as<int32_t>(), N, 1, true, st))) return rc;
            std::vector<int32_t> key_sorted((size_t)N);
            SN_HIP(hipMemcpy(key_sorted.data(), deg_sorted.p, (size_t)N * 4, hipMemcpyDeviceToHost));
            while (g->n_trest < N && key_sorted[g->n_trest]) ++g->n_trest;
            if ((rc = dev_alloc(&g->trest, g->n_trest))) return rc;
            if (g->n_trest > 0) k_row_desc<<<grid1(g->n_trest), 256, 0, st>>>(perm_t.as<int32_t>(), g->rowptr, g->n_trest, g->trest);
            SN_HIP(hipStreamSynchronize(st));      // perm_t goes out of scope
        } else if ((rc = dev_alloc(&g->trest, 0))) return rc;
    }

    // 7. split-row / split-source task lists
    g->n_split = g->rows_gt(WAVE_T);
    if ((rc = build_tasks(g->rdeg, g->n_split, &g->task_slot, &g->task_chunk, &g->split_task0,
                          &g->split_soff, &g->n_tasks, &g->split_edges)))
        return rc;
    // 7b. the order in which the forward deals the split rows' tasks to its persistent waves.
    //     Position q goes to wave q mod n_waves, i.e. (grids being multiples of 8 workgroups of
    //     WAVES waves, dealt round-robin over the 8 XCDs) to XCD (q / WAVES) % 8.  The edges of a
    //     row keep the edge list's order - ascending sources for a coalesced list - so a 128-edge
    //     task reads a narrow band of source rows: tasks are dealt so that those of source slice
    //     x land on XCD x, whose 4 MB L2 then holds its slice of the filter table (2.7 MB of
    //     21.7 MB at arxiv size) instead of a random eighth of all of it.  Speed only: any
    //     order is correct, and nothing relies on the placement.
    if ((rc = dev_alloc(&g->task_order, g->n_tasks))) return rc;
    if (g->n_tasks > 0) {
        DevBuf d_slice;
        if (d_slice.alloc((size_t)g->n_tasks * 4)) { set_error("out of device memory (graph build)"); return SNGNN_ENOMEM; }
        k_task_slice<<<grid1(g->n_tasks), 256, 0, st>>>(g->rdesc, g->col, g->task_slot, g->task_chunk, g->n_tasks,
                                                       Ntot, d_slice.as<int32_t>());
        std::vector<int32_t> slice((size_t)g->n_tasks), order((size_t)g->n_tasks);
        SN_HIP(hipMemcpyAsync(slice.data(), d_slice.p, slice.size() * 4, hipMemcpyDeviceToHost, st));
        SN_HIP(hipStreamSynchronize(st));
        std::vector<std::vector<int32_t>> bucket(N_SLICE);
        for (int t = 0; t < g->n_tasks; ++t) bucket[slice[t]].push_back(t);      // ascending t = descending degree
        size_t head[N_SLICE] = {0};
        for (int q = 0; q < g->n_tasks; ++q) {
            int x = (q / WAVES) % N_SLICE;
            if (head[x] >= bucket[x].size()) {          // slice exhausted: take from the fullest one
                size_t best = 0;
                for (int y = 0; y < N_SLICE; ++y)
                    if (bucket[y].size() - head[y] > best) { best = bucket[y].size() - head[y]; x = y; }
            }
            order[q] = bucket[x][head[x]++];
        }
        SN_HIP(hipMemcpy(g->task_order, order.data(), order.size() * 4, hipMemcpyHostToDevice));
    }
    g->n_ssplit = g->srcs_gt(WAVE_T);
    if ((rc = build_tasks(g->sdeg, g->n_ssplit, &g->stask_slot, &g->stask_chunk, &g->ssplit_task0,
                          nullptr, &g->n_stasks, nullptr)))
        return rc;
    // 8. the kept-bit layout a training forward writes itself (common.h): geometry + the static bit
    //    index of every CSC entry.  (Whole graphs only: the node-centric backward that reads it is.)
    {
        const int n_med_end = g->rows_gt(SMALL_T);
        g->kb_wbase = ((N + 1) / 2 + 3) / 4 * 4;
        g->kb_tbase = g->kb_wbase + 4 * (int64_t)(n_med_end - g->n_split);
        g->kb_words = g->kb_tbase + 4 * (int64_t)g->n_tasks;
        if (N == Ntot && Ep > 0 && g->kb_words * 32 < ((int64_t)1 << 31)) {
            DevBuf rslot;
            if (rslot.alloc((size_t)N * 4)) { set_error("out of device memory (graph build)"); return SNGNN_ENOMEM; }
            if ((rc = dev_alloc(&g->csc_bit, Ep))) return rc;
            k_invert<<<grid1(N), 256, 0, st>>>(g->rperm, rslot.as<int32_t>(), N);
            k_csc_bit<<<grid1(Ep), 256, 0, st>>>(g->csc_eid, g->csc_dst, g->rowptr, rslot.as<int32_t>(), g->split_task0,
                                                 Ep, g->n_split, n_med_end, g->kb_wbase, g->kb_tbase, g->csc_bit);
            SN_HIP(hipStreamSynchronize(st));      // rslot goes out of scope
        }
    }
    SN_HIP(hipStreamSynchronize(st));
    return 0;
}

}  // namespace sngnn

#define SN_STR2(x) #x
#define SN_STR(x) SN_STR2(x)
using namespace sngnn;

extern "C" {

const char *sngnn_last_error(void) { return g_err.c_str(); }

const char *sngnn_build_info(void)
{
    return "libsngnn_hip;arch=gfx950;hip=" SN_STR(HIP_VERSION_MAJOR) "." SN_STR(HIP_VERSION_MINOR);
}

int sngnn_graph_create_partition(const int64_t *edge_index_dev, int64_t E, int64_t N_total,
                                 int64_t row_begin, int64_t row_end, int add_loops,
                                 int remove_loops, void *stream, sngnn_graph_t **out_graph)
{
    SN_REQUIRE(out_graph != nullptr, SNGNN_EINVAL, "out_graph is NULL");
    *out_graph = nullptr;
    SN_REQUIRE(E >= 0 && N_total >= 0, SNGNN_EINVAL, "negative size");
    SN_REQUIRE(0 <= row_begin && row_begin <= row_end && row_end <= N_total, SNGNN_EINVAL,
               "row range must satisfy 0 <= row_begin <= row_end <= N_total");
    SN_REQUIRE(E == 0 || edge_index_dev != nullptr, SNGNN_EINVAL, "edge_index is NULL");
    SN_REQUIRE(remove_loops >= 0 && remove_loops <= SNGNN_LOOPS_REPLACE, SNGNN_EINVAL,
               "remove_loops must be 0, 1 or SNGNN_LOOPS_REPLACE");
    sngnn_graph *g = new (std::nothrow) sngnn_graph();
    SN_REQUIRE(g != nullptr, SNGNN_ENOMEM, "out of host memory");
    int rc = build(g, edge_index_dev, E, N_total, row_begin, row_end, add_loops != 0,
                   remove_loops, (hipStream_t)stream);
    if (rc != 0) { sngnn_graph_destroy(g); return rc; }
    *out_graph = g;
    return SNGNN_OK;
}

int sngnn_graph_create(const int64_t *edge_index_dev, int64_t E, int64_t N, int add_loops,
                       int remove_loops, void *stream, sngnn_graph_t **out_graph)
{
    return sngnn_graph_create_partition(edge_index_dev, E, N, 0, N, add_loops, remove_loops,
                                        stream, out_graph);
}

void sngnn_graph_destroy(sngnn_graph_t *g)
{
    if (!g) return;
    void *ptrs[] = {g->rowptr, g->col, g->eid, g->cscptr, g->csc_eid, g->csc_dst, g->csc_pos, g->col_s, g->rperm_b, g->rdesc_b, g->col_s_b, g->rperm, g->fdesc, g->trest,
                    g->sperm, g->rdesc, g->sdesc, g->inv_deg, g->task_slot, g->task_chunk, g->split_soff, g->split_task0, g->task_order,
                    g->stask_slot, g->stask_chunk, g->ssplit_task0, g->csc_bit};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    delete g;
}

int64_t sngnn_graph_num_nodes(const sngnn_graph_t *g) { return g ? g->N : -1; }
int64_t sngnn_graph_kept_bits_bytes(const sngnn_graph_t *g) { return (g && g->csc_bit) ? g->kb_words * 4 : 0; }
int64_t sngnn_graph_num_total_nodes(const sngnn_graph_t *g) { return g ? g->Ntot : -1; }
int64_t sngnn_graph_row_offset(const sngnn_graph_t *g) { return g ? g->row_off : -1; }
int64_t sngnn_graph_num_edges(const sngnn_graph_t *g) { return g ? g->Ep : -1; }
int64_t sngnn_graph_max_in_degree(const sngnn_graph_t *g) { return g ? g->max_in_deg : -1; }
int64_t sngnn_graph_src_min(const sngnn_graph_t *g) { return g ? g->src_min : -1; }
int64_t sngnn_graph_num_fused_nodes(const sngnn_graph_t *g) { return g ? g->n_fused : -1; }

int64_t sngnn_graph_workspace_bytes(const sngnn_graph_t *g, int C)
{
    if (!g || C < 1) return -1;
    // forward: unit rows | norms | scores of split rows | one partial row per split task |
    //          CAND_MAX_K candidate keys and source ids per task | one done word per task (the finalize role: agg_fwd_impl.h)
    int64_t fwd = sngnn::fwd_table_bytes(g->Ntot, C) +
                  (g->split_edges + 3) / 4 * 4 * 4 + ((int64_t)g->n_tasks * C + 3) / 4 * 4 * 4 +
                  (int64_t)g->n_tasks * 32 * 8 + (int64_t)g->n_tasks * 32 * 4 + (int64_t)g->n_tasks * 8;
    // backward: {w, ds} record per edge | dnT per node | partT per split task | partS (2 rows) per
    //           split-source task
    //           (attention mode: 2 rows + 4 scalars) | partS (2 rows) per split-source task
    int64_t bwd = (2 * g->Ep + 3) / 4 * 4 * 4 + g->N * (int64_t)C * 4 +
                  (int64_t)g->n_tasks * (2 * C + 4) * 4 + (int64_t)g->n_stasks * C * 4 * 2 +
                  (g->N + 3) / 4 * 4 * 4;            // (attention mode: dot_i per target, BwdArgs::rec_dot)
    int64_t b = std::max(fwd, bwd);
    return (b + 255) / 256 * 256;
}

static const void *graph_array(const sngnn_graph_t *g, int which, int64_t *n)
{
    switch (which) {
    case 0: *n = g->N + 1; return g->rowptr;
    case 1: *n = g->Ep; return g->col;
    case 2: *n = g->Ep; return g->eid;
    case 3: *n = g->Ntot + 1; return g->cscptr;
    case 4: *n = g->Ep; return g->csc_eid;
    case 5: *n = g->N; return g->rperm;
    default: *n = 0; return nullptr;
    }
}

int sngnn_graph_copy_array(const sngnn_graph_t *g, int which, void *host_dst)
{
    SN_REQUIRE(g && host_dst, SNGNN_EINVAL, "NULL argument");
    int64_t n = 0;
    const void *p = graph_array(g, which, &n);
    SN_REQUIRE(p != nullptr, SNGNN_EINVAL, "unknown array id");
    if (n > 0) SN_HIP(hipMemcpy(host_dst, p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return SNGNN_OK;
}

const void *sngnn_graph_array_dev(const sngnn_graph_t *g, int which)
{
    if (!g) return nullptr;
    int64_t n = 0;
    return graph_array(g, which, &n);
}

}  // extern "C"
