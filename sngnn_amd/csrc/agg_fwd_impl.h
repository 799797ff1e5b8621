// Fused forward of the similarity-navigated aggregation (gfx950).
//
// A normalisation pass and the gather kernels compute, for every target row i of the
// CSR-by-target graph,
//     n_j   = h_j / nrm_j,  nrm_j = max(|h_j|_2, eps)      (F.normalize; k_normalize_rows)
//     s_e   = <n_i, n_j>                               per in-edge e = (j -> i)
//     keep  = top_k by (s desc, edge position asc) AND s >= thr    (or all, top_k < 0)
//     out_i = (1 / max(deg_i, 1)) * sum_{e kept} s_e * h_j
// i.e. models/models.py:122+132+139-158, :238-239+244-263, :325-326+331-334 of the
// reference without materialising any per-edge [E', C] tensor.
//
// Normalise-then-dot, in the reference's order (F.normalize first, then
// (norm_i * norm_j).sum(-1)), with IEEE square root and division per ROW: whenever two
// source rows normalise to the same bits in the reference (duplicates, power-of-two
// multiples, rows with one non-zero channel: exactly +-1 there) they do here, their
// cosines are the same bits, and the tie falls to the edge position, as in the reference.
// The gather reads the unit row n_j (4C bytes) and, for a kept edge, its 4-byte norm
// (h_j = n_j * nrm_j to half an ulp).  Measured alternatives: dividing each
// gathered h_j by its norm inside the gather (no unit-row table) costs the kernel 17 us of
// vector arithmetic; it is at its balance point between the vector pipes and the line rate
// of the memory system.
//
// Rows are processed in order of descending in-degree (graph.rperm) in three
// classes, all work items of one persistent launch:
//   A  split rows  (deg > WAVE_T): one wave per CHUNK-edge task scores its edges
//      (edge-balanced) and keeps its chunk-local top-k as candidates; a finalize launch
//      merges the candidates per row and gathers the winners;
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
#include "agg_fwd_filter.h"
#include "device_utils.h"
#include "head_row.h"

namespace sngnn {

struct FwdArgs {
    // TABLE mode: n = unit rows [Ntot, C] (k_normalize_rows), nrm = max(|h|_2, eps) [Ntot].
    // OTF mode (on the fly; nrm == nullptr): n = the RAW rows h themselves.  Every edge gets the
    // fast cosine <h_i, h_j> inv_i inv_j (device_utils.h: edge_score), which differs from the
    // reference-order value s = <h_i / d_i, h_j / d_j> by at most `delta`; wherever a decision
    // cannot be taken from the fast value - within delta of thr, within 2 delta of the top_k-th or
    // of another edge whose rank is asked for - the edge is scored again exactly (IEEE
    // normalisation of the two rows in registers, the table path's dot), so selections, ties
    // included, are those of the table path, bit for bit, without the normalisation pass.
    const float *n;
    const float *nrm;
    float delta;          // OTF: bound on |fast - exact| (launcher: (4 C + 32) 2^-24)
    const uint4 *filt;    // [Ntot, filter_row_halfs(C)] fp16 filter rows (agg_fwd_filter.h) or nullptr
    int filt_small;       // the filter also in front of the SMALL rows (small_rows_set_filt): pays when a threshold
                          // prunes - then most rows fetch no fp32 row at all (arxiv size, top_k 16 / thr 0.9: main
                          // kernel 40.4 -> 36.2 us); with thr 0 it is a second dependent round trip per set for
                          // rows that keep something anyway (top_k 1: 49.3 -> 61.3 us), so the launcher sets it
                          // for thr >= 0.25 (and top_k >= 4) only
    int filt_min_deg;     // wave rows below this in-degree score their fp32 rows directly (two dependent round trips
                          // for a short row cost more than the second lines they save): launcher, agg_fwd.hip
    int C, N;             // N = owned target rows
    int row_off;          // row i's own feature row is n[row_off + i] (node-range partition)
    const int32_t *rowptr, *col, *rperm;
    const int32_t *col_s;       // col with the rows in slot order (rdesc.w = first entry)
    const int4 *rdesc;     // per degree-sorted slot: {row, first edge, in-degree, first entry in col_s}
    int k;            // < 0: no selection
    float thr;
    float *out, *wsel, *inv_norm;
    int32_t *sel_src;
    float *sel_w;
    int n_split, n_med_end;     // slots [0,n_split) split, [n_split,n_med_end) wave, rest small
    int n_tasks;
    const int32_t *task_slot, *task_chunk, *split_soff, *split_task0;
    const int32_t *task_order;      // position in the dealing order -> task (graph.hip 7b)
    float *scores, *partial;    // workspace
    unsigned long long *cand_key;   // [n_tasks, k]  chunk-local top-k keys of split rows
    int32_t *cand_src;              // [n_tasks, k]  their source ids (saves the finalize a dependent load)
    int use_cand;                   // split rows keep chunk-local candidates (k <= CAND_MAX_K and they fit LDS)
    int lowbits;                    // bits needed for a row-local edge index
    int n_split_gt_wave;        // split rows with more than 128 / k tasks (descending order: the first ones)
    // The split rows' finalize INSIDE the main launch (round 5; fin_block_pair / fin_group_batch / fin_stream_row below): workgroups [main_blocks, gridDim.x)
    // take no work items - their waves finalize split rows, each as soon as the row's tasks have published their
    // candidates (fin_done[task] == fin_nonce).  main_blocks == gridDim.x: the finalize is the next launch.
    int main_blocks;
    unsigned long long *fin_done;   // [n_tasks] workspace; left zero by the wave that consumed them
    unsigned long long fin_nonce;   // this call's own value (never 0, never an earlier call's)
    unsigned k_magic;               // ceil(2^32 / k) for k >= 2: q / k = __umulhi(q, k_magic) for q < 2^32 / k (fin_slot)
    long long n_edges;              // E' of the call's graph (host side: the launcher's estimate of the launch's time)
    int role_mask;              // measurement aid (sngnn_tuning_set): bit 0 tasks, 1 wave rows, 2 small rows
    // row filter (sngnn_agg_forward_rows): only the target rows i with row_flag[i] == row_want are
    // computed and written; nullptr = all rows.  (A rank's interior rows - all sources local - run
    // while the halo rows of its boundary rows are still in flight, sngnn_amd/dist.py.)
    const uint8_t *row_flag;
    int row_want;
    __device__ __forceinline__ bool skip_row(int i) const { return row_flag != nullptr && row_flag[i] != row_want; }
    // store epilogue of a hidden layer (models.py:204-209: conv -> [+ bias] -> relu_ -> dropout), applied to
    // the finished mean row on its way out: out = keep ? max(mean + bias, 0) * scale : 0.
    // epi_flags: bit 0 relu, bit 1 bias, bit 2 keep mask; 0 = none (sngnn_agg_forward_epilogue).
    // Kept bits written by the forward itself (training calls; common.h "kbits" layout; nullptr = off, the
    // per-edge weights wsel are written instead): a small row stores its 16 bits as one halfword
    // at its row id, a wave row its 128 bits at its slot, a split-row task zeroes its 128 bits and
    // the finalize sets the winners' - plain stores of whole units each row owns: no atomics between
    // rows, no clearing pass, and the backward needs no k_pack_kept launch.  Host guarantees
    // 0 <= top_k <= SMALL_T (wave rows and tasks always rank) and the candidate finalize.
    unsigned *kbits;
    int kb_wbase, kb_tbase;
    // epi_flags: bit 0 relu, bit 1 bias, bit 2 keep mask given, bit 3 keep mask drawn here; 0 = none.
    int epi_flags;
    const float *epi_bias;          // [C]
    const uint8_t *epi_keep;        // [N, C] 1 = kept (the caller's own Bernoulli(1 - p) draw), or
    const unsigned long long *epi_seed;   // dev [1]: the mask is drawn here, keep = u(seed, i C + c) >= p
    float epi_p;
    float epi_scale;                // 1 / (1 - p) when something is dropped, else 1
    // The classification head of the LAST layer inside the forward's second launch (head_row.h;
    // sngnn_epilogue_t.head_*): the finished mean row (+ epi_bias) is the row of logits - its
    // log-softmax NLL term and arg-max hit are added to an entry of head_part ([entries][4] floats:
    // loss A, correct A, loss B, correct B) and, training, d loss / d logits replaces the row in `out`.
    // Split rows go through it in their finalize, while they are in registers (entry head_nmain + p);
    // all other rows are read back from `out` by extra workgroups of the SAME launch (head_rows_role;
    // entry = their wave id): the finalize is a latency chain of a few hundred workgroups on an
    // otherwise idle chip, the head a bandwidth-bound pass - one launch, the longer of the two.
    // (In the main kernel's stores instead: its 80-register budget spilled 15-39 registers, at 96
    // the kernel took 62 against 51 us - more than the head pass costs.)
    // head_flags: bit 0 two splits (head_sel is a bit set), bit 1 out = the gradient, else out = the logits.
    const int64_t *head_y;
    const uint8_t *head_sel;
    float *head_part;
    int head_flags, head_nmain;
    float head_scale;               // 1 / rows of split A (the gradient's factor, the mean's)
    float head_scale_b;             // 1 / rows of split B
    float *head_out;                // dev [2] or [4]: (mean NLL, correct count) per split, by launch_head_reduce
    __device__ __forceinline__ bool drawn_keep(int i, int c) const       // (device_utils.h: sn_dropout_keep)
    {
        return sn_dropout_keep(*epi_seed, (unsigned long long)i * (unsigned)C + (unsigned)c, epi_p);
    }
    __device__ __forceinline__ float epilogue(float v, int i, int c) const
    {
        if (epi_flags & 2) v += epi_bias[c];
        if (epi_flags & 1) v = fmaxf(v, 0.f);
        if (epi_flags & 4) v = epi_keep[(size_t)i * C + c] ? v * epi_scale : 0.f;
        if (epi_flags & 8) v = drawn_keep(i, c) ? v * epi_scale : 0.f;
        return v;
    }
};

// the same on a row held by a lane group (channels beyond C: untouched, never stored)
template <int VEC, int G, int R>
__device__ __forceinline__ void row_epilogue(const FwdArgs &a, Row<VEC, G, R> &acc, int i, int lg)
{
    if (a.epi_flags == 0) return;                         // (uniform)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int c0 = (r * G + lg) * VEC;
        if (c0 < a.C) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc.x[r][v] = a.epilogue(acc.x[r][v], i, c0 + v);
        }
    }
}

// what a wave has added up over its rows (lanes lg == 0 of every lane group hold terms)
struct HeadAcc { float loss = 0.f, corr = 0.f, lossb = 0.f, corrb = 0.f; };

// the finished mean row `acc` of target i, held by a G-lane group, goes through the head; yy / sv = the
// row's label and split byte (loaded early by the caller: not a round trip at the end of the row)
// (in_memory: `out` already holds the row as it is in acc - stored again only if the bias changes it)
template <int VEC, int G, int R>
__device__ __forceinline__ void head_store_row(const FwdArgs &a, const Row<VEC, G, R> &acc, int i, int lg, int yy,
                                               unsigned sv, HeadAcc &ha, bool in_memory = false)
{
    if constexpr (VEC == 4 && R == 1 && (G == 8 || G == 16)) {
        const bool in = 4 * lg < a.C;
        const int c0 = in ? 4 * lg : 0;
        float4 t = make_float4(acc.x[0][0], acc.x[0][1], acc.x[0][2], acc.x[0][3]);
        if (a.epi_flags & 2) {
            const float4 b = *reinterpret_cast<const float4 *>(a.epi_bias + c0);
            t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w;
        }
        float4 *o = reinterpret_cast<float4 *>(a.out + (size_t)i * a.C + c0);
        const bool grad = (a.head_flags & 2) != 0;
        if (sv == 0) {                                        // (group-uniform) the row is in no split
            if (in && grad) *o = make_float4(0.f, 0.f, 0.f, 0.f);
            else if (in && (!in_memory || (a.epi_flags & 2))) *o = t;
            return;
        }
        if (in && !grad && (!in_memory || (a.epi_flags & 2))) *o = t;
        const HeadRow hr = head_row<G>(t, in, c0, yy);
        if (lg == 0) {
            if (a.head_flags & 1) {
                if (sv & 1) { ha.loss += hr.loss; ha.corr += hr.corr; }
                if (sv & 2) { ha.lossb += hr.loss; ha.corrb += hr.corr; }
            } else {
                ha.loss += hr.loss;
                ha.corr += hr.corr;
            }
        }
        if (in && grad) *o = head_row_grad(hr, a.head_scale);
    }
}

// one entry of head_part from a wave's accumulators (fixed order: a shuffle tree over the lanes)
__device__ __forceinline__ void head_write_entry(const FwdArgs &a, int entry, const HeadAcc &ha)
{
    float v0 = ha.loss, v1 = ha.corr, v2 = ha.lossb, v3 = ha.corrb;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        v0 += __shfl_xor(v0, m, 64); v1 += __shfl_xor(v1, m, 64);
        v2 += __shfl_xor(v2, m, 64); v3 += __shfl_xor(v3, m, 64);
    }
    if (lane_id() == 0) *reinterpret_cast<float4 *>(a.head_part + 4 * (size_t)entry) = make_float4(v0, v1, v2, v3);
}

// Every row that is NOT a split row (in-degree <= WAVE_T), in natural order, read back from `out`: wave w of
// nw walks 2 x 64 / G rows per step and returns what it has added up.
template <int VEC, int G, int R>
__device__ __forceinline__ HeadAcc head_rows_role(const FwdArgs &a, int w, int nw)
{
    HeadAcc ha;
    if constexpr (VEC == 4 && R == 1 && (G == 8 || G == 16)) {
        constexpr int RPW = 64 / G, U = 2;
        const int lane = lane_id();
        const int gid = lane / G, lg = lane % G;
        const int c0 = 4 * lg < a.C ? 4 * lg : 0;
        for (int64_t base = (int64_t)w * (RPW * U); base < a.N; base += (int64_t)nw * (RPW * U)) {
            Row<VEC, G, R> row[U];
            int yy[U], deg[U];
            unsigned sv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {                       // all loads first
                const int64_t i = base + u * RPW + gid;
                const int64_t ic = i < a.N ? i : a.N - 1;
                deg[u] = a.rowptr[ic + 1] - a.rowptr[ic];
                sv[u] = a.head_sel[ic];
                yy[u] = (int)a.head_y[ic];
                const float4 t = *reinterpret_cast<const float4 *>(a.out + ic * a.C + c0);
                row[u].x[0][0] = t.x; row[u].x[0][1] = t.y; row[u].x[0][2] = t.z; row[u].x[0][3] = t.w;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = base + u * RPW + gid;
                if (i >= a.N || deg[u] > WAVE_T) continue;      // (group-uniform; split rows: their finalize)
                head_store_row<VEC, G, R>(a, row[u], (int)i, lg, yy[u], sv[u], ha, true);
            }
        }
    }
    return ha;
}

// one entry of head_part per WORKGROUP of the head role: the waves' sums through LDS, added in wave order
// (s_mem: 4 floats per wave).  Called by every thread of the workgroup.
__device__ __forceinline__ void head_block_entry(const FwdArgs &a, int entry, const HeadAcc &ha, float *s_mem)
{
    float v0 = ha.loss, v1 = ha.corr, v2 = ha.lossb, v3 = ha.corrb;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        v0 += __shfl_xor(v0, m, 64); v1 += __shfl_xor(v1, m, 64);
        v2 += __shfl_xor(v2, m, 64); v3 += __shfl_xor(v3, m, 64);
    }
    const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    if (lane_id() == 0) *reinterpret_cast<float4 *>(s_mem + 4 * wave) = make_float4(v0, v1, v2, v3);
    __syncthreads();
    if (threadIdx.x == 0) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int w = 0; w < nwaves; ++w) {
            const float4 q = *reinterpret_cast<const float4 *>(s_mem + 4 * w);
            t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
        }
        *reinterpret_cast<float4 *>(a.head_part + 4 * (size_t)entry) = t;
    }
}

// workgroups of the head role in a launch of `block` threads (head_part entry = the workgroup's index):
// enough waves to fill the chip, never more than the rows need
inline int head_role_blocks(int64_t N, int G, int block)
{
    // (inside the finalize launch the workgroups hold 75 registers - six waves per SIMD, three 512-thread
    // workgroups per CU - and the finalize's own workgroups sit beside them: ONE round of 2 per CU)
    const int64_t cap = block == BLOCK ? 256 * 32 : (int64_t)256 * 2 * (block / 64);
    const int64_t waves = std::min<int64_t>(ceil_div(N, 2 * (64 / G)), cap);
    return ceil_div(waves, block / 64);
}

template <int VEC, int G, int R>
__global__ __launch_bounds__(BLOCK) void k_agg_head_rows(const FwdArgs a)
{
    __shared__ __align__(16) float s_mem[4 * WAVES];
    const int nw = gridDim.x * WAVES;
    const HeadAcc ha = head_rows_role<VEC, G, R>(a, blockIdx.x * WAVES + (threadIdx.x >> 6), nw);
    head_block_entry(a, blockIdx.x, ha, s_mem);
}

#ifndef SNGNN_FWD_WAVES
#define SNGNN_FWD_WAVES 6
#endif
constexpr int FWD_WAVES_PER_SIMD = SNGNN_FWD_WAVES;   // register budget of the main kernel (6: <= 80 VGPRs)

// source rows in flight per lane group (x 64/G groups per wave): ~16 registers of row data
template <int R> struct Unroll { static constexpr int U = (R == 1) ? 4 : (R == 2 ? 2 : 1); };

// words of LDS per wave of the main kernel: 4 regions of SETW words
//   small rows: column ids (two buffers) | scores | kept weights      [64/G rows][SMALL_T] each
//   wave rows / tasks: scores [128] | kept list [128] | column ids [128]
template <int G> struct WaveLds {
    static constexpr int RPW = 64 / G;
    static constexpr int SETW = (RPW * SMALL_T > 128) ? RPW * SMALL_T : 128;
    //   finalize role (fin_group_batch): 96 words per row, one row per lane group
    static constexpr int WORDS = 4 * SETW > RPW * 96 ? 4 * SETW : RPW * 96;
};

// ---------------------------------------------------------------------------
// F.normalize (models.py:122,238,325): n = h / max(|h|_2, eps), one lane group per row.
// IEEE sqrt and division: a row with a single non-zero channel becomes exactly +-1 there.
// ---------------------------------------------------------------------------
template <int VEC, int G, int R>
__global__ __launch_bounds__(BLOCK) void k_normalize_rows(const float *__restrict__ h, int64_t rows, int C,
                                                          float *__restrict__ n, float *__restrict__ nrm,
                                                          uint2 *__restrict__ filt)
{
    using RowT = Row<VEC, G, R>;
    constexpr int RPW = 64 / G;
    constexpr int U = Unroll<R>::U >= 2 ? 2 : 1;      // rows per lane group per step
    const int lane = lane_id();
    const int gid = lane / G, lg = lane % G;
    const int64_t nw = (int64_t)gridDim.x * WAVES;
    const int64_t w0 = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    for (int64_t base = w0 * RPW * U; base < rows; base += nw * RPW * U) {
        RowT x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = base + u * RPW + gid;
            x[u].load(h + (r < rows ? r : rows - 1) * C, C, lg);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = base + u * RPW + gid;
            const float q = group_sum<G>(x[u].dot_partial(x[u]));
            const float d = fmaxf(ieee_sqrt(q), EPS_NORM);
            x[u].div_rn(d);
            if (r < rows) {
                x[u].store(n + r * C, C, lg);
                if (lg == 0) nrm[r] = d;
                if constexpr (VEC == 4) {
                    // fp16 filter row (agg_fwd_filter.h): the lanes' 4-channel vectors cover exactly
                    // filter_row_halfs(C) = 4 G R channels, the ones beyond C hold zeros
                    if (filt) {
#pragma unroll
                        for (int q = 0; q < R; ++q)
                            filt[(size_t)r * (G * R) + q * G + lg] =
                                make_uint2(pack_half2(x[u].x[q][0] * FILT_SCALE, x[u].x[q][1] * FILT_SCALE),
                                           pack_half2(x[u].x[q][2] * FILT_SCALE, x[u].x[q][3] * FILT_SCALE));
                    }
                }
            }
        }
    }
}

template <int VEC, int G, int R>
int launch_normalize_rows(const float *h, int64_t rows, int C, float *n, float *nrm, void *filt, hipStream_t st)
{
    constexpr int RPW = 64 / G;
    constexpr int U = Unroll<R>::U >= 2 ? 2 : 1;
    if (rows <= 0) return SNGNN_OK;
    // a streaming pass: short work items (one step per wave up to 8 resident workgroups per CU,
    // grid-stride beyond), so the launch drains evenly
    const int64_t steps = (rows + RPW * U - 1) / (RPW * U);
    const int grid = (int)std::min<int64_t>(ceil_div(steps, WAVES), 256 * 8 * 4);
    if (filt && !(VEC == 4 && filter_row_bytes(C) == 8 * G * R)) {
        set_error("internal: filter rows need 16-byte row vectors");
        return SNGNN_EINVAL;
    }
    k_normalize_rows<VEC, G, R><<<grid, BLOCK, 0, st>>>(h, rows, C, n, nrm, (uint2 *)filt);
    SN_HIP(hipGetLastError());
    return SNGNN_OK;
}

// fp16 filter rows from unit rows the caller already holds (sngnn_agg_forward_normalized)
template <int G, int R>
__global__ __launch_bounds__(BLOCK) void k_filter_rows(const float *__restrict__ n, int64_t rows, int C,
                                                       uint2 *__restrict__ filt)
{
    using RowT = Row<4, G, R>;
    constexpr int RPW = 64 / G;
    const int lane = lane_id();
    const int gid = lane / G, lg = lane % G;
    const int64_t nw = (int64_t)gridDim.x * WAVES;
    for (int64_t r = ((int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6)) * RPW + gid; r < rows; r += nw * RPW) {
        RowT x;
        x.load(n + r * C, C, lg);
#pragma unroll
        for (int q = 0; q < R; ++q)
            filt[(size_t)r * (G * R) + q * G + lg] =
                make_uint2(pack_half2(x.x[q][0] * FILT_SCALE, x.x[q][1] * FILT_SCALE),
                           pack_half2(x.x[q][2] * FILT_SCALE, x.x[q][3] * FILT_SCALE));
    }
}

template <int VEC, int G, int R>
int launch_filter_rows(const float *n, int64_t rows, int C, void *filt, hipStream_t st)
{
    if constexpr (VEC == 4) {
        if (rows <= 0) return SNGNN_OK;
        if (filter_row_bytes(C) != 8 * G * R) { set_error("internal: filter row layout"); return SNGNN_EINVAL; }
        const int grid = (int)std::min<int64_t>(ceil_div(rows, (64 / G) * WAVES), 256 * 8 * 4);
        k_filter_rows<G, R><<<grid, BLOCK, 0, st>>>(n, rows, C, (uint2 *)filt);
        SN_HIP(hipGetLastError());
        return SNGNN_OK;
    } else {
        set_error("internal: filter rows need 16-byte row vectors");
        return SNGNN_EINVAL;
    }
}

// per-edge cosine of two unit rows (fma chain over the lane's channels, fixed-order
// group sum); -0.0 -> +0.0: the reference orders floats, not bit patterns
template <int VEC, int G, int R>
__device__ __forceinline__ float unit_dot(const Row<VEC, G, R> &a, const Row<VEC, G, R> &x)
{
    return group_sum<G>(a.dot_partial(x)) + 0.0f;
}

// F.normalize of a row held in registers - the instruction sequence of k_normalize_rows (same
// fma chain, DPP tree, IEEE square root and division: the same bits); returns the clamped norm
template <int VEC, int G, int R>
__device__ __forceinline__ float normalize_in_place(Row<VEC, G, R> &x)
{
    const float q = group_sum<G>(x.dot_partial(x));
    const float d = fmaxf(ieee_sqrt(q), EPS_NORM);
    x.div_rn(d);
    return d;
}

// the reference-order cosine of a UNIT target row and a RAW source row (OTF mode's exact score:
// what the table path computes from its two table rows)
template <int VEC, int G, int R>
__device__ __forceinline__ float exact_score_raw(const Row<VEC, G, R> &nu, Row<VEC, G, R> x)
{
    normalize_in_place<VEC, G, R>(x);
    return unit_dot<VEC, G, R>(nu, x);
}

// the G bits of a wave ballot that belong to lane group gid
template <int G> __device__ __forceinline__ unsigned long long fwd_group_bits(unsigned long long m, int gid)
{
    if constexpr (G == 64) return m;
    else return (m >> (gid * G)) & ((1ull << G) - 1ull);
}

// ---------------------------------------------------------------------------
// Wave-level top-k of up to 128 selection keys, two per lane (0 = no key).
// Keys are unique, so "the k largest" is well defined: bitwise search for the
// k-th largest key T, then keep key >= T.  lowbits = bits of the largest
// row-local edge index (the low word of a key is 0xFFFFFFFF - index, so its
// upper 32 - lowbits bits are all ones and need no search).
// ---------------------------------------------------------------------------
constexpr int CAND_MAX_K = 32;   // split rows: chunk-local candidates are kept for k <= this

// (wave_topk_keys_n: device_utils.h)
__device__ __forceinline__ void wave_topk_keys(unsigned long long key0, unsigned long long key1,
                                               int k, int lowbits, bool &kept0, bool &kept1)
{
    const unsigned long long key[2] = {key0, key1};
    bool kept[2];
    wave_topk_keys_n<2>(key, k, lowbits, kept);
    kept0 = kept[0];
    kept1 = kept[1];
}

// the same among CANDIDATES (keys that already won a selection: wave_topk_keys_n's PREFIX note) - the finalize's calls
__device__ __forceinline__ void wave_topk_cands(unsigned long long key0, unsigned long long key1,
                                                int k, int lowbits, bool &kept0, bool &kept1)
{
    const unsigned long long key[2] = {key0, key1};
    bool kept[2];
    wave_topk_keys_n<2, true>(key, k, lowbits, kept);
    kept0 = kept[0];
    kept1 = kept[1];
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
template <int VEC, int G, int R, bool OTF, int EPI>
__device__ __forceinline__ void small_rows_set(const FwdArgs &a, const int4 d, bool valid,
                                               int *lds_wave, const int *s_col_set)
{
    using RowT = Row<VEC, G, R>;
    constexpr int U = Unroll<R>::U;
    constexpr int SETW = WaveLds<G>::SETW;
    const int lane = lane_id();
    const int gid = lane / G, lg = lane % G;
    const int i = d.x, rs = d.y, deg = d.z;
    const bool emit = a.sel_src != nullptr && a.k >= 0;
    const bool rank = a.k >= 0 && deg > a.k;
    const bool need_sc = rank || emit;
    const int self = i + a.row_off;

    const int *s_col = s_col_set + gid * SMALL_T;
    float *s_sc = reinterpret_cast<float *>(lds_wave + 2 * SETW) + gid * SMALL_T;
    float *s_w = reinterpret_cast<float *>(lds_wave + 3 * SETW) + gid * SMALL_T;

    RowT ni;
    ni.load(a.n + (size_t)self * a.C, a.C, lg);
    const int dmax = wave_max_i(deg);
    float inv_i = 0.f, q_i = 0.f;
    if constexpr (OTF) {
        q_i = group_sum<G>(ni.dot_partial(ni));
        inv_i = inv_norm_of(q_i);
    }

    RowT acc;
    acc.zero();
    unsigned kb = 0u;                 // the row's kept bits (a.kbits)
    for (int t0 = 0; t0 < dmax; t0 += U) {
        RowT x[U];
        float nj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = (t0 + u) < deg ? s_col[t0 + u] : self;
            x[u].load(a.n + (size_t)j * a.C, a.C, lg);
            if constexpr (OTF) nj[u] = 1.f;
            else nj[u] = rank ? 0.f : a.nrm[j];   // a streaming row weighs the row it has just scored
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float s;
            if constexpr (OTF) {
                s = edge_score<VEC, G, R>(ni, inv_i, x[u]);
                if (a.k >= 0 && !rank) {
                    // the threshold decides this edge: exact where the fast value cannot tell
                    const bool amb = (t0 + u < deg) && fabsf(s - a.thr) <= a.delta;
                    if (__ballot(amb) != 0ull) {                 // (wave-uniform; rare)
                        RowT nu = ni;
                        normalize_in_place<VEC, G, R>(nu);
                        const float se = exact_score_raw<VEC, G, R>(nu, x[u]);
                        if (amb) s = se;
                    }
                }
            } else {
                s = unit_dot<VEC, G, R>(ni, x[u]);
            }
            if (t0 + u < deg) {
                if (need_sc && lg == 0) s_sc[t0 + u] = s;
                if (!rank) {
                    const bool sel = (a.k < 0) || (s >= a.thr);
                    if (sel) acc.axpy(s * nj[u], x[u]);
                    if (a.wsel && lg == 0) a.wsel[rs + t0 + u] = sel ? s : SNGNN_UNSELECTED;
                    if constexpr (!OTF) kb |= sel ? (1u << (t0 + u)) : 0u;     // (group-uniform; table mode only)
                }
            }
        }
    }
    if (a.inv_norm && valid && lg == 0) {
        if constexpr (OTF) a.inv_norm[i] = ieee_div(1.0f, fmaxf(ieee_sqrt(q_i), EPS_NORM));
        else a.inv_norm[i] = ieee_div(1.0f, a.nrm[self]);
    }

    if constexpr (OTF) {
        if (need_sc) {
            // ranks are asked for: wherever two fast scores of a row (or one and thr) are too close
            // to order, the whole row (<= 16 edges, cache-hot) is scored again exactly
            wave_lds_sync();
            bool amb = false;
            for (int e = lg; e < deg; e += G) {
                const float se = s_sc[e];
                // edges surely above / possibly above this one: its side of the top_k cut is in
                // doubt iff the cut falls between the two counts (ranks emitted: any near pair)
                int above = 0, maybe = 0;
                for (int b = 0; b < deg; ++b) {
                    const float sb = s_sc[b];
                    above += sb > se + 2.0f * a.delta;
                    maybe += (b != e) && sb >= se - 2.0f * a.delta;
                }
                amb |= emit ? (maybe > above) : (above < a.k && maybe >= a.k);
                amb |= above < a.k && fabsf(se - a.thr) <= a.delta;
            }
            const bool row_amb = fwd_group_bits<G>(__ballot(amb), gid) != 0ull;
            if (__ballot(row_amb) != 0ull) {                         // (wave-uniform)
                RowT nu = ni;
                normalize_in_place<VEC, G, R>(nu);
                const int dm = wave_max_i(row_amb ? deg : 0);
                for (int t = 0; t < dm; ++t) {
                    const bool live = row_amb && t < deg;
                    RowT xr;
                    xr.load(a.n + (size_t)(live ? s_col[t] : self) * a.C, a.C, lg);
                    const float se = exact_score_raw<VEC, G, R>(nu, xr);
                    if (live && lg == 0) s_sc[t] = se;
                }
            }
        }
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
            // (lane lg == 0 runs every iteration any lane of its group runs: it ends with all the bits)
            if constexpr (!OTF)
                if (rank && a.kbits) kb |= (unsigned)fwd_group_bits<G>(__ballot(sel), gid) << (e - lg);
            if (emit && sel) {
                a.sel_src[(size_t)i * a.k + rk] = s_col[e];
                a.sel_w[(size_t)i * a.k + rk] = se;
            }
        }
        if (rank) {
            wave_lds_sync();
            // second pass: gather only the kept source rows, in edge order
            // (group-divergent trip count: no cross-lane operation inside)
            for (int t = 0; t < deg; ++t) {
                const float w = s_w[t];
                if (w != SNGNN_UNSELECTED) {
                    const int j = s_col[t];
                    RowT xr;
                    xr.load(a.n + (size_t)j * a.C, a.C, lg);
                    if constexpr (OTF) acc.axpy(w, xr);
                    else acc.axpy(w * a.nrm[j], xr);
                }
            }
        }
        wave_lds_sync();       // s_sc / s_w are reused by the next set
    }
    if constexpr (!OTF)
        if (a.kbits && valid && lg == 0) reinterpret_cast<unsigned short *>(a.kbits)[i] = (unsigned short)kb;
    if (valid) {
        acc.div((float)max(deg, 1));
        if constexpr (EPI == 1) row_epilogue<VEC, G, R>(a, acc, i, lg);
        acc.store(a.out + (size_t)i * a.C, a.C, lg);
    }
}

// The same set with the fp16 FILTER in front (agg_fwd_filter.h; the FILT kernel, table mode, G >= 16 >=
// SMALL_T: lane lg of a group owns edge lg of its row).  The unfiltered set gathers the fp32 unit row of
// EVERY in-edge to score it - two lines at C = 40 or 64 - although a ranking row keeps top_k of them and
// a threshold may drop most: 95 % of an arxiv-like graph's rows are small rows.  Here every edge is scored
// from its one-line filter row first (|s~ - s| <= FILT_EPS), and only the CANDIDATES
//        s~ >= thr - eps   and   fewer than top_k edges with s~' > s~ + 2 eps
// are fetched and scored in fp32: every edge of the exact selection is a candidate (the filter header's
// argument), the exact rule then runs on the candidates' exact scores, so indices, weights, ranks and kept
// bits are those of the unfiltered set, bit for bit.
template <int VEC, int G, int R, int EPI>
__device__ __forceinline__ void small_rows_set_filt(const FwdArgs &a, const int4 d, bool valid,
                                                    int *lds_wave, const int *s_col_set)
{
    static_assert(G >= SMALL_T, "one lane per edge of a small row");
    using RowT = Row<VEC, G, R>;
    constexpr int U = Unroll<R>::U;
    constexpr int UF = (R == 1) ? 4 : 2;                 // filter rows in flight per lane group
    constexpr int SETW = WaveLds<G>::SETW;
    constexpr int FR = G * R;                            // 8-byte lane entries per filter row
    const int lane = lane_id();
    const int gid = lane / G, lg = lane % G;
    const int i = d.x, rs = d.y, deg = d.z;
    const bool emit = a.sel_src != nullptr;
    const bool rank = deg > a.k;                         // (a.k >= 0: the filter belongs to selecting calls)
    const bool need_sc = rank || emit;
    const int self = i + a.row_off;
    const uint2 *f2 = reinterpret_cast<const uint2 *>(a.filt);

    const int *s_col = s_col_set + gid * SMALL_T;
    float *s_sc = reinterpret_cast<float *>(lds_wave + 2 * SETW) + gid * SMALL_T;   // approximate, then exact scores
    float *s_w = reinterpret_cast<float *>(lds_wave + 3 * SETW) + gid * SMALL_T;    // kept weights
    int *s_c = reinterpret_cast<int *>(s_w);                                        // (before that: candidate edges)

    uint2 fi[R];
#pragma unroll
    for (int q = 0; q < R; ++q) fi[q] = f2[(size_t)self * FR + q * G + lg];
    const int dmax = wave_max_i(deg);

    // phase 1: approximate scores of all edges from the filter rows
    for (int t0 = 0; t0 < dmax; t0 += UF) {
        uint2 x[UF][R];
#pragma unroll
        for (int u = 0; u < UF; ++u) {
            const int j = (t0 + u) < deg ? s_col[t0 + u] : self;
#pragma unroll
            for (int q = 0; q < R; ++q) x[u][q] = f2[(size_t)j * FR + q * G + lg];
        }
#pragma unroll
        for (int u = 0; u < UF; ++u) {
            float pr = 0.f;
#pragma unroll
            for (int q = 0; q < R; ++q) {
                pr = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, fi[q].x), __builtin_bit_cast(half2_t, x[u][q].x), pr, false);
                pr = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, fi[q].y), __builtin_bit_cast(half2_t, x[u][q].y), pr, false);
            }
            const float sa = group_sum<G>(pr) * FILT_UNSCALE;
            if (t0 + u < deg && lg == 0) s_sc[t0 + u] = sa;
        }
    }
    wave_lds_sync();

    // phase 2: lane lg decides edge lg
    const bool has = lg < deg;
    const float sa = has ? s_sc[lg] + 0.0f : -INFINITY;
    const float lo = a.thr - FILT_EPS;
    bool cand = has && sa >= lo;
    if (cand && rank) {
        int above = 0;
        for (int b = 0; b < deg; ++b) {
            const float sb = s_sc[b];
            above += (sb >= lo) && (sb > sa + 2.0f * FILT_EPS);
        }
        cand = above < a.k;
    }
    const unsigned long long gm = fwd_group_bits<G>(__ballot(cand), gid);
    const int nc = __popcll(gm);
    wave_lds_sync();                                          // (every s_sc read above is done)
    if (cand) s_c[__popcll(gm & ((1ull << lg) - 1ull))] = lg;
    if (has && !cand) {
        s_sc[lg] = -INFINITY;                                 // never selected, ranks below every candidate
        if (!need_sc && a.wsel) a.wsel[rs + lg] = SNGNN_UNSELECTED;
    }
    wave_lds_sync();

    // phase 3: the candidates' fp32 rows and exact scores
    RowT acc;
    acc.zero();
    unsigned kb = 0u;
    const int ncmax = wave_max_i(nc);
    // (the row's own fp32 unit row only now, with the first candidates' rows: a set without candidates - most
    // sets behind a threshold that prunes - touches one-line filter rows only)
    RowT ni;
    if (ncmax > 0) ni.load(a.n + (size_t)self * a.C, a.C, lg);
    else ni.zero();
    for (int q0 = 0; q0 < ncmax; q0 += U) {
        RowT x[U];
        float nj[U];
        int e[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            e[u] = (q0 + u) < nc ? s_c[q0 + u] : -1;
            const int j = e[u] >= 0 ? s_col[e[u]] : self;
            x[u].load(a.n + (size_t)j * a.C, a.C, lg);
            nj[u] = need_sc ? 0.f : a.nrm[j];                 // a streaming row weighs the row it has just scored
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float sx = unit_dot<VEC, G, R>(ni, x[u]);
            if (e[u] >= 0) {
                if (need_sc) {
                    if (lg == 0) s_sc[e[u]] = sx;
                } else {
                    const bool sel = sx >= a.thr;
                    if (sel) acc.axpy(sx * nj[u], x[u]);
                    if (a.wsel && lg == 0) a.wsel[rs + e[u]] = sel ? sx : SNGNN_UNSELECTED;
                    kb |= sel ? (1u << e[u]) : 0u;
                }
            }
        }
    }
    if (a.inv_norm && valid && lg == 0) a.inv_norm[i] = ieee_div(1.0f, a.nrm[self]);

    if (need_sc) {
        wave_lds_sync();
        // phase 4: the exact rule on the exact scores (non-candidates hold -inf) - the unfiltered set's code
        for (int e = lg; e < deg; e += G) {
            const float se = s_sc[e];
            int rk = 0;
            for (int b = 0; b < deg; ++b) {
                const float sb = s_sc[b];
                rk += (sb > se) || (sb == se && b < e);
            }
            const bool sel = rk < a.k && se >= a.thr;
            s_w[e] = sel ? se : SNGNN_UNSELECTED;
            if (a.wsel) a.wsel[rs + e] = sel ? se : SNGNN_UNSELECTED;
            if (a.kbits) kb |= (unsigned)fwd_group_bits<G>(__ballot(sel), gid) << (e - lg);
            if (emit && sel) {
                a.sel_src[(size_t)i * a.k + rk] = s_col[e];
                a.sel_w[(size_t)i * a.k + rk] = se;
            }
        }
        wave_lds_sync();
        // the kept rows again (they were candidates a moment ago: cache-hot), in edge order
        for (int t = 0; t < deg; ++t) {
            const float w = s_w[t];
            if (w != SNGNN_UNSELECTED) {
                const int j = s_col[t];
                RowT xr;
                xr.load(a.n + (size_t)j * a.C, a.C, lg);
                acc.axpy(w * a.nrm[j], xr);
            }
        }
        wave_lds_sync();       // s_sc / s_w are reused by the next set
    }
    if (a.kbits && valid && lg == 0) reinterpret_cast<unsigned short *>(a.kbits)[i] = (unsigned short)kb;
    if (valid) {
        acc.div((float)max(deg, 1));
        if constexpr (EPI == 1) row_epilogue<VEC, G, R>(a, acc, i, lg);
        acc.store(a.out + (size_t)i * a.C, a.C, lg);
    }
}

// ---------------------------------------------------------------------------
// Class C driver: waves are persistent, each walks sets wave_id, wave_id + n_waves, ...
// and keeps the NEXT set's descriptors and column ids in flight while it works on the
// current one, so a set costs one memory round trip (its feature rows) instead of a
// chain of three (descriptor -> columns -> rows).
// ---------------------------------------------------------------------------
template <int VEC, int G, int R, bool OTF, int EPI, bool FS = false>
__device__ __forceinline__ void role_small(const FwdArgs &a, int set0, int stride, int nsets, int *lds_wave)
{
    constexpr int RPW = 64 / G;
    constexpr int CPL = (RPW * SMALL_T + 63) / 64;      // column ids per lane per set
    constexpr int SETW = WaveLds<G>::SETW;
    const int lane = lane_id();
    const int gid = lane / G;

    auto load_desc = [&](int st) -> int4 {
        const int slot = a.n_med_end + st * RPW + gid;
        int4 d = (st < nsets && slot < a.N) ? a.rdesc[slot] : make_int4(0, 0, 0, -1);
        if (a.row_flag != nullptr && d.w >= 0 && a.skip_row(d.x)) d = make_int4(0, 0, 0, -1);   // not this pass's row
        return d;
    };
    // lane l fetches column ids q = l + 64 m of the set: row q / SMALL_T, edge q % SMALL_T
    auto load_cols = [&](const int4 d, int (&c)[CPL]) {
#pragma unroll
        for (int m = 0; m < CPL; ++m) {
            const int q = lane + 64 * m;
            const int r = q / SMALL_T, t = q % SMALL_T;
            // descriptor of row r of the set lives in the lanes of group r
            const int rss = __shfl(d.w, r * G, 64), dg = __shfl(d.z, r * G, 64);
            c[m] = (r < RPW && t < dg) ? a.col_s[rss + t] : 0;      // consecutive slots: one stream
        }
    };
    auto store_cols = [&](const int (&c)[CPL], int *dst) {
#pragma unroll
        for (int m = 0; m < CPL; ++m) {
            const int q = lane + 64 * m;
            if (q < RPW * SMALL_T) dst[q] = c[m];
        }
    };

    int s_cur = set0;
    if (s_cur >= nsets) return;
    int s_nxt = s_cur + stride;
    // two [RPW][SMALL_T] column-id buffers at lds_wave + SETW * buf.  (Plain pointer arithmetic:
    // an array of the two pointers loses the LDS address space and every read of a column
    // id becomes a FLAT load, which must drain the whole memory pipeline - s_waitcnt
    // vmcnt(0) - in the middle of a batch of row gathers.)
    int4 d_cur = load_desc(s_cur);
    int4 d_nxt = load_desc(s_nxt);
    int cols[CPL];
    load_cols(d_cur, cols);
    store_cols(cols, lds_wave);
    int buf = 0;
    while (s_cur < nsets) {
        const int s_n2 = s_nxt + stride;
        const int4 d_n2 = load_desc(s_n2);
        load_cols(d_nxt, cols);                 // in flight during this set's work
        wave_lds_sync();
        if (a.row_flag == nullptr || __ballot(d_cur.w >= 0) != 0ull) {
            bool done = false;
            if constexpr (FS && G >= SMALL_T && !OTF) {
                if (a.filt_small) {                               // (uniform)
                    small_rows_set_filt<VEC, G, R, EPI>(a, d_cur, d_cur.w >= 0, lds_wave, lds_wave + SETW * buf);
                    done = true;
                }
            }
            if (!done) small_rows_set<VEC, G, R, OTF, EPI>(a, d_cur, d_cur.w >= 0, lds_wave, lds_wave + SETW * buf);
        }
        store_cols(cols, lds_wave + SETW * (buf ^ 1));
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
//   OTF: ni is the RAW target row, inv_i its fast inverse norm.  Scores are the fast cosine,
//           except: a streaming edge within delta of thr is scored exactly; with exact_all EVERY
//           edge is (the paths that rank from scores in HBM scratch: top_k > CAND_MAX_K, hubs
//           beyond the candidate finalize, ranks of a streaming split row).
template <int VEC, int G, int R, bool OTF>
__device__ __forceinline__ void score_edges(const FwdArgs &a, int self, int rs, int e0, int e1,
                                            const Row<VEC, G, R> &ni, float inv_i, bool stream,
                                            float *sc, int sc_off, Row<VEC, G, R> &acc,
                                            int *ids_lds = nullptr, bool exact_all = false)
{
    using RowT = Row<VEC, G, R>;
    constexpr int NG = 64 / G;
    constexpr int U = Unroll<R>::U;
    const int lane = lane_id();
    const int gid = lane / G, lg = lane % G;
    RowT nu;                                   // OTF + exact_all: the unit target row
    if constexpr (OTF) {
        if (exact_all) { nu = ni; normalize_in_place<VEC, G, R>(nu); }
    }
    // column ids one iteration ahead: the col -> row dependency of iteration n+1
    // overlaps the row loads of iteration n
    int jn[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int t = e0 + u * NG + gid;
        jn[u] = t < e1 ? a.col[rs + t] : self;
    }
    for (int base = e0; base < e1; base += NG * U) {
        int t[U], j[U];
        bool act[U];
        RowT x[U];
        float nj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            t[u] = base + u * NG + gid;
            act[u] = t[u] < e1;
            j[u] = jn[u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            x[u].load(a.n + (size_t)j[u] * a.C, a.C, lg);
            if constexpr (OTF) nj[u] = 1.f;
            else nj[u] = stream ? a.nrm[j[u]] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int tn = t[u] + NG * U;
            jn[u] = tn < e1 ? a.col[rs + tn] : self;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float s;
            if constexpr (OTF) {
                if (exact_all) {
                    s = exact_score_raw<VEC, G, R>(nu, x[u]);
                } else {
                    s = edge_score<VEC, G, R>(ni, inv_i, x[u]);
                    if (stream && a.k >= 0) {
                        const bool amb = act[u] && fabsf(s - a.thr) <= a.delta;
                        if (__ballot(amb) != 0ull) {             // (wave-uniform; rare)
                            RowT nt = ni;
                            normalize_in_place<VEC, G, R>(nt);
                            const float se = exact_score_raw<VEC, G, R>(nt, x[u]);
                            if (amb) s = se;
                        }
                    }
                }
            } else {
                s = unit_dot<VEC, G, R>(ni, x[u]);
            }
            if (act[u]) {
                if (sc && lg == 0) sc[t[u] - sc_off] = s;
                if (ids_lds && lg == 0) ids_lds[t[u] - sc_off] = j[u];      // for the re-gather of the kept rows
                if (stream) {
                    const bool sel = (a.k < 0) || (s >= a.thr);
                    if (sel) acc.axpy(s * nj[u], x[u]);
                    if (a.wsel && lg == 0) a.wsel[rs + t[u]] = sel ? s : SNGNN_UNSELECTED;
                }
            }
        }
    }
}

// Exact scores of the listed edges (list[q] = chunk-local edge index, ids[idx] = its source):
// sc[idx] = <n_i, n_j>.  U rows per lane group in flight; a slot past the end repeats the
// last listed edge and stores nothing.
// (OTF: ni is the UNIT target row, the listed rows are raw and are normalised in registers)
template <int VEC, int G, int R, bool OTF>
__device__ __forceinline__ void score_list(const FwdArgs &a, const Row<VEC, G, R> &ni, const int *list, int ncand,
                                           const int *ids, float *sc)
{
    using RowT = Row<VEC, G, R>;
    constexpr int NG = 64 / G;
#ifndef SNGNN_LIST_U
#define SNGNN_LIST_U 1
#endif
    constexpr int U = Unroll<R>::U * (R == 1 ? SNGNN_LIST_U : 1);     // the usual k + a few candidates: one trip
    const int lane = lane_id();
    const int gid = lane / G, lg = lane % G;
    for (int q0 = 0; q0 < ncand; q0 += NG * U) {
        RowT x[U];
        int idx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = q0 + u * NG + gid;
            idx[u] = list[min(q, ncand - 1)];
            x[u].load(a.n + (size_t)ids[idx[u]] * a.C, a.C, lg);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float s;
            if constexpr (OTF) s = exact_score_raw<VEC, G, R>(ni, x[u]);
            else s = unit_dot<VEC, G, R>(ni, x[u]);
            if (q0 + u * NG + gid < ncand && lg == 0) sc[idx[u]] = s;
        }
    }
}

// "Score the edges [e0, e1) of a ranking row and select" in two precisions: approximate scores
// of all n = e1 - e0 <= 128 edges, exact scores of the CANDIDATES only - the edges within
// 2 eps of the top_k-th approximate score and not below thr - eps, where eps bounds
// |approximate - exact| (agg_fwd_filter.h has the argument) - and the exact selection among them.
//   FILT: approximate = fp16 filter rows (eps = FILT_EPS), exact = table rows;
//   OTF:  approximate = the fast cosine of the raw rows (eps = a.delta), exact = rows normalised
//         in registers; ni = raw target row, inv_i its fast inverse norm.
// On return sc[t] holds the exact score of every candidate (in particular of every kept edge),
// ids[t] the source of every edge, and the returned keys / flags are what wave_select would have
// given on exact scores of all edges.  list[] is scratch.
//   OTF, optimistic = true: the selection is first made on the fast scores alone and kept when it
//         is beyond doubt - no edge within delta of thr, and the weakest kept edge more than
//         2 delta above the strongest valid edge left out (then the exact scores would select
//         the same set; the kept edges' weights are the fast values, within delta of the exact
//         ones).  Only a row that fails this test - ties and near ties at the cut - takes the
//         candidate path.  (Not for split-row tasks: their keys are merged with other tasks'
//         keys by the finalize, which needs them exact; not when ranks are emitted.)
template <int VEC, int G, int R, bool OTF>
__device__ __forceinline__ WaveSel banded_select(const FwdArgs &a, const Row<VEC, G, R> &ni, float inv_i, int self,
                                                 int rs, int e0, int e1, float *sc, int *list, int *ids, int lowbits,
                                                 bool optimistic = false)
{
    using RowT = Row<VEC, G, R>;
    const int lane = lane_id();
    const int n = e1 - e0;
    float eps;
    if constexpr (OTF) {
        RowT unused;
        unused.zero();
        score_edges<VEC, G, R, true>(a, self, rs, e0, e1, ni, inv_i, false, sc, e0, unused, ids);
        eps = a.delta;
    } else {
        constexpr int GF = G * R / 2;                // 16-byte lanes per filter row (VEC == 4)
        filter_scores<GF>(a.filt, a.col, self, rs, e0, e1, sc, ids);
        eps = FILT_EPS;
    }
    wave_lds_sync();
    if constexpr (OTF) {
        if (optimistic) {
            const int i0 = lane, i1 = lane + 64;
            const float f0 = i0 < n ? sc[i0] + 0.0f : 0.f, f1 = i1 < n ? sc[i1] + 0.0f : 0.f;
            WaveSel r;
            r.key0 = (i0 < n && f0 >= a.thr) ? sel_key(f0, e0 + i0) : 0ull;
            r.key1 = (i1 < n && f1 >= a.thr) ? sel_key(f1, e0 + i1) : 0ull;
            wave_topk_keys(r.key0, r.key1, a.k, lowbits, r.kept0, r.kept1);
            bool doubt = (i0 < n && fabsf(f0 - a.thr) <= eps) || (i1 < n && fabsf(f1 - a.thr) <= eps);
            // weakest kept against strongest valid edge left out
            float kmin = fminf(r.kept0 ? f0 : INFINITY, r.kept1 ? f1 : INFINITY);
            float umax = fmaxf((r.key0 != 0ull && !r.kept0) ? f0 : -INFINITY, (r.key1 != 0ull && !r.kept1) ? f1 : -INFINITY);
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                kmin = fminf(kmin, __shfl_xor(kmin, m, 64));
                umax = fmaxf(umax, __shfl_xor(umax, m, 64));
            }
            doubt = doubt || (kmin - umax <= 2.0f * eps);      // (umax = -inf: nothing was left out)
            if (__ballot(doubt) == 0ull) return r;
        }
    }
    bool c0, c1;
    approx_candidates(sc, n, a.k, a.thr, eps, c0, c1);
    const unsigned long long m0 = __ballot(c0), m1 = __ballot(c1);
    const int n0 = __popcll(m0), ncand = n0 + __popcll(m1);
    if (c0) list[prefix_popc(m0)] = lane;
    if (c1) list[n0 + prefix_popc(m1)] = lane + 64;
    wave_lds_sync();
    if (ncand > 0) {
        if constexpr (OTF) {
            RowT nu = ni;
            normalize_in_place<VEC, G, R>(nu);
            score_list<VEC, G, R, true>(a, nu, list, ncand, ids, sc);
        } else {
            score_list<VEC, G, R, false>(a, ni, list, ncand, ids, sc);
        }
    }
    wave_lds_sync();
    WaveSel r;
    const float s0 = c0 ? sc[lane] : 0.f, s1 = c1 ? sc[lane + 64] : 0.f;
    r.key0 = (c0 && s0 >= a.thr) ? sel_key(s0, e0 + lane) : 0ull;
    r.key1 = (c1 && s1 >= a.thr) ? sel_key(s1, e0 + lane + 64) : 0ull;
    wave_topk_keys(r.key0, r.key1, a.k, lowbits, r.kept0, r.kept1);
    return r;
}

// ---------------------------------------------------------------------------
// Class B: SMALL_T < deg <= WAVE_T, one wave per row.
// ---------------------------------------------------------------------------
template <int VEC, int G, int R, bool FILT, bool OTF, int EPI>
__device__ __forceinline__ void role_wave(const FwdArgs &a, int item, int *lds_wave)
{
    using RowT = Row<VEC, G, R>;
    constexpr int NG = 64 / G;
    const int lane = lane_id();
    const int gid = lane / G, lg = lane % G;
    const int slot = a.n_split + item;
    const int4 d = a.rdesc[slot];
    const int i = d.x, rs = d.y, deg = d.z;
    if (a.skip_row(i)) return;                              // (wave-uniform)
    const int self = i + a.row_off;
    const bool emit = a.sel_src != nullptr && a.k >= 0;
    const bool rank = a.k >= 0 && deg > a.k;
    const bool need_sc = rank || emit;

    float *s_sc = reinterpret_cast<float *>(lds_wave);     // [WAVE_T]
    int *s_list = lds_wave + WAVE_T;                         // [WAVE_T]
    int *s_ids = lds_wave + 2 * WAVE_T;                      // [WAVE_T]

    RowT ni;
    ni.load(a.n + (size_t)self * a.C, a.C, lg);
    float inv_i = 0.f, q_i = 0.f;
    if constexpr (OTF) {
        q_i = group_sum<G>(ni.dot_partial(ni));
        inv_i = inv_norm_of(q_i);
    }

    RowT acc;
    acc.zero();
    // two-precision selection: the filter for ranking rows; OTF whenever scores are kept (ranks
    // asked for on a streaming row included: their order needs exact scores too)
    const bool banded = OTF ? need_sc : (FILT && rank && deg >= a.filt_min_deg);
    if (!banded)
        score_edges<VEC, G, R, OTF>(a, self, rs, 0, deg, ni, inv_i, !rank, need_sc ? s_sc : nullptr, 0, acc,
                                    rank ? s_ids : nullptr);

    if (need_sc) {
        WaveSel ws;
        if constexpr (FILT || OTF) {
            if (banded) {
                ws = banded_select<VEC, G, R, OTF>(a, ni, inv_i, self, rs, 0, deg, s_sc, s_list, s_ids, 7,
                                                   /*optimistic=*/OTF && !emit);
            } else {
                wave_lds_sync();
                ws = wave_select(s_sc, deg, 0, a.k, a.thr, 7);
            }
        } else {
            wave_lds_sync();
            ws = wave_select(s_sc, deg, 0, a.k, a.thr, 7);
        }
        const int i0 = lane, i1 = lane + 64;
        // kept list in ascending position order
        const unsigned long long m0 = __ballot(ws.kept0), m1 = __ballot(ws.kept1);
        const int n0 = __popcll(m0);
        if (ws.kept0) s_list[prefix_popc(m0)] = i0;
        if (ws.kept1) s_list[n0 + prefix_popc(m1)] = i1;
        const int nsel = n0 + __popcll(m1);
        if ((rank || banded) && a.wsel) {
            if (i0 < deg) a.wsel[rs + i0] = ws.kept0 ? s_sc[i0] : SNGNN_UNSELECTED;
            if (i1 < deg) a.wsel[rs + i1] = ws.kept1 ? s_sc[i1] : SNGNN_UNSELECTED;
        }
        if constexpr (!OTF)
            if (a.kbits && lane < 4)          // the row's 128 kept bits at its slot
                a.kbits[a.kb_wbase + 4 * item + lane] = (unsigned)((lane < 2 ? m0 : m1) >> (32 * (lane & 1)));
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
        if ((rank || banded) && nsel > 0) {
            // the kept rows again, U per lane group in flight, unconditionally (a slot past the
            // end repeats the last kept edge with weight 0): their column ids wait in LDS, so
            // this is ONE memory round trip instead of a col -> row chain per kept edge
            constexpr int U = Unroll<R>::U;
            for (int q0 = 0; q0 < nsel; q0 += U * NG) {
                RowT x[U];
                float w[U], nj[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int q = q0 + u * NG + gid;
                    const int idx = s_list[min(q, nsel - 1)];
                    const int j = s_ids[idx];
                    w[u] = q < nsel ? s_sc[idx] : 0.f;
                    x[u].load(a.n + (size_t)j * a.C, a.C, lg);
                    if constexpr (OTF) nj[u] = 1.f;
                    else nj[u] = a.nrm[j];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) acc.axpy(w[u] * nj[u], x[u]);
            }
        }
    }
    acc.reduce_across_groups();
    if (gid == 0) {
        acc.div((float)deg);
        if constexpr (EPI == 1) row_epilogue<VEC, G, R>(a, acc, i, lg);
        acc.store(a.out + (size_t)i * a.C, a.C, lg);
    }
    // (stored at the END of the row: a store up front would pin the scoring pass's first
    // column-id loads behind it - the pointers are not restrict)
    if (lane == 0 && a.inv_norm) {
        if constexpr (OTF) a.inv_norm[i] = ieee_div(1.0f, fmaxf(ieee_sqrt(q_i), EPS_NORM));
        else a.inv_norm[i] = ieee_div(1.0f, a.nrm[self]);
    }
    wave_lds_sync();            // the wave's LDS scratch is reused by its next item
}

// ---------------------------------------------------------------------------
// Class A: one CHUNK-edge task of a split row.
// ---------------------------------------------------------------------------
// FIN: the row's finalize runs in THIS launch (fin_block_pair / fin_group_batch / fin_stream_row) - what it reads
// or overwrites is stored at agent scope, then the task says "done".  The launches whose finalize is the next
// launch run the FIN = false instantiation: plain stores, the code of round 4.
template <int VEC, int G, int R, bool FILT, bool OTF, bool FIN>
__device__ __forceinline__ void role_task(const FwdArgs &a, int tq, int *lds_wave)
{
    using RowT = Row<VEC, G, R>;
    const int lane = lane_id();
    const int gid = lane / G, lg = lane % G;
    const int p = a.task_slot[tq], c = a.task_chunk[tq];
    const int4 d = a.rdesc[p];
    const int i = d.x, rs = d.y, deg = d.z;
    if (a.skip_row(i)) return;                              // (wave-uniform)
    const int self = i + a.row_off;
    const int e0 = c * CHUNK, e1 = min(deg, e0 + CHUNK);
    const bool emit = a.sel_src != nullptr && a.k >= 0;
    const bool rank = a.k >= 0 && deg > a.k;
    const bool cand = rank && a.use_cand;        // chunk-local top-k -> candidates

    RowT ni;
    ni.load(a.n + (size_t)self * a.C, a.C, lg);
    float inv_i = 0.f;
    if constexpr (OTF) {
        const float q_i = group_sum<G>(ni.dot_partial(ni));
        inv_i = inv_norm_of(q_i);
        if (c == 0 && lane == 0 && a.inv_norm) a.inv_norm[i] = ieee_div(1.0f, fmaxf(ieee_sqrt(q_i), EPS_NORM));
    } else {
        if (c == 0 && lane == 0 && a.inv_norm) a.inv_norm[i] = ieee_div(1.0f, a.nrm[self]);
    }

    RowT acc;
    acc.zero();
    float *s_sc = reinterpret_cast<float *>(lds_wave);          // [CHUNK], chunk-local
    float *sc_glb = (!cand && (rank || emit)) ? a.scores + a.split_soff[p] : nullptr;   // HBM scratch
    if ((FILT || OTF) && cand) { /* scored below, in two precisions */ }
    else if (cand) score_edges<VEC, G, R, OTF>(a, self, rs, e0, e1, ni, inv_i, !rank, s_sc, e0, acc);   // LDS
    // HBM scratch (ranked later from those scores: OTF writes exact ones) / none (pure streaming)
    else score_edges<VEC, G, R, OTF>(a, self, rs, e0, e1, ni, inv_i, !rank, sc_glb, 0, acc, nullptr,
                                     /*exact_all=*/sc_glb != nullptr);
    if (cand) {
        WaveSel ws;
        if constexpr (FILT || OTF) {
            ws = banded_select<VEC, G, R, OTF>(a, ni, inv_i, self, rs, e0, e1, s_sc, lds_wave + CHUNK,
                                               lds_wave + 2 * CHUNK, a.lowbits);
        } else {
            wave_lds_sync();
            ws = wave_select(s_sc, e1 - e0, e0, a.k, a.thr, a.lowbits);
        }
        // (FIN: agent-scope stores - the row's finalize runs in another workgroup of THIS launch, on another XCD;
        // what it reads, and what it overwrites, must be in memory before the task says "done")
        auto put = [](auto *q, auto v) {
            if constexpr (FIN) st_agent(q, v);
            else *q = v;
        };
        unsigned long long *ck = a.cand_key + (size_t)tq * CAND_MAX_K;
        const unsigned long long m0 = __ballot(ws.kept0), m1 = __ballot(ws.kept1);
        const int n0 = __popcll(m0), nsel = n0 + __popcll(m1);
        if (ws.kept0) put(ck + prefix_popc(m0), ws.key0);
        if (ws.kept1) put(ck + n0 + prefix_popc(m1), ws.key1);
        if (lane >= nsel && lane < a.k) put(ck + lane, 0ull);          // empty slots (k <= 32 < 64)
        int32_t *cs = a.cand_src + (size_t)tq * CAND_MAX_K;       // saves the finalize a dependent load
        if (ws.kept0) put(cs + prefix_popc(m0), a.col[rs + e0 + lane]);
        if (ws.kept1) put(cs + n0 + prefix_popc(m1), a.col[rs + e0 + 64 + lane]);
        if (a.wsel) {     // the finalize overwrites the kept edges of the row
            float *w = a.wsel + rs + e0;
            if (lane < e1 - e0) put(w + lane, SNGNN_UNSELECTED);
            if (lane + 64 < e1 - e0) put(w + lane + 64, SNGNN_UNSELECTED);
        }
        if constexpr (!OTF)
            if (a.kbits && lane < 4) put(a.kbits + a.kb_tbase + 4 * tq + lane, 0u);      // the finalize sets the winners' bits
    } else if (!rank) {
        acc.reduce_across_groups();
        if constexpr (FIN) {
            if (gid == 0) {
                float *pr = a.partial + (size_t)tq * a.C;
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if ((r * G + lg) * VEC < a.C) st_agent_vec<VEC>(pr + (r * G + lg) * VEC, acc.x[r]);
            }
        } else {
            if (gid == 0) acc.store(a.partial + (size_t)tq * a.C, a.C, lg);
        }
    }
    if constexpr (FIN) {
        if (cand || !rank) {                                      // (uniform) publish
            wave_vmem_drain();
            if (lane == 0) st_agent(a.fin_done + tq, a.fin_nonce);
        }
    }
    wave_lds_sync();            // the wave's LDS scratch is reused by its next item
}

// ---------------------------------------------------------------------------
// The split rows' finalize inside the main launch (round 5).  As a launch of its own it was 7.4 us of a 70 us
// step at arxiv size - a dependency chain (candidates -> selection -> 16 rows -> store) of 825 rows on an
// otherwise idle chip, behind a launch boundary - although its inputs exist ~5 us into the main kernel: the tasks
// are every wave's FIRST work item.  So the LAST workgroups of the launch (FwdArgs::main_blocks ..) take no work
// items: they finalize the split rows, each as soon as the row's tasks have published their candidates, while the
// other workgroups stream the wave rows and small rows.  What bounds the role is round trips, not work (an
// agent-scope load comes from memory: 2-3 us under the main roles' traffic), so both forms below keep as many
// rows' loads in flight as the registers hold:
//   * rows of more than 128 candidates (arxiv size: 94, up to 1 584 candidates): two WAVES per row, two rows per
//     workgroup - a wave selects among half of the candidates (512 keys per round, eight per lane, all loads of a
//     round in flight together), the even wave among the pair's winners.  (One wave per row, 96 new candidates per
//     round: 17 dependent round trips for the biggest row - the launch's critical path, 66 us against 48.)
//   * the others: one lane GROUP per row, 64 / G rows per wave at once (fin_group_batch) - the accumulation runs in
//     the order of the one-wave-per-row finalize launch (fin_wave_row), so the rows' bits are the same in both.
//   Progress: a task never waits, and workgroups are dispatched in index order, so whenever a finalize workgroup is
// resident every task's workgroup has been dispatched.  Should that ever not hold the wait is BOUNDED: a row whose
// tasks have not shown up after FIN_SPIN_MAX polls (~1 s) is written as NaN and its waves stop waiting for the
// rows behind it - a loud wrong answer, never a hung GPU.
//   Visibility: tasks store what this role reads (candidate keys / sources) or overwrites (unselected marks, zeroed
// kept-bit words) at agent scope and drain them before the done word; this role loads them at agent scope.
// ---------------------------------------------------------------------------
constexpr int FIN_SPIN_MAX = 1 << 19;
constexpr int FIN_BLOCKS_MAX = 256;
constexpr int FIN_GROUP_WORDS = 3 * CAND_MAX_K;      // LDS words of one row's winners: 32 keys (64 words) | 32 source ids

template <int VEC, int G, int R, bool HEAD>
__device__ __forceinline__ void fin_winners_row(const FwdArgs &a, int p, int i, int rs, int deg, int t0, int nsel,
                                                const unsigned long long *s_key_w, const int *s_src_w, int head_yy,
                                                unsigned head_sv);

// candidate q of a row whose first task is t0 (task t0 + q / k, entry q % k of its CAND_MAX_K-slot record) without
// the twenty-instruction division per candidate (eight per lane in flight: their temporaries were the role's spills)
__device__ __forceinline__ unsigned fin_slot(const FwdArgs &a, int t0, int q)
{
    const int t = a.k == 1 ? q : (int)__umulhi((unsigned)q, a.k_magic);
    return (unsigned)(t0 + t) * CAND_MAX_K + (unsigned)(q - t * a.k);       // (n_tasks * 32 < 2^31: int32 edge ids, CHUNK = 128)
}

// Top-k of the keys of ONE LANE GROUP (NK per lane, 0 = no key): wave_topk_keys_n's search with every count taken
// over the group (the groups of a wave search different rows: no uniform early return - a group that is done
// idles until the last one is).  The kept set is the k largest keys: the same set whichever routine picks it.
template <int G, int NK>
__device__ __forceinline__ void group_topk_keys_n(const unsigned long long (&key)[NK], int k, int lowbits,
                                                  bool (&kept)[NK])
{
    const float fk = (float)k;                                // (counts <= 64 NK / (64 / G): exact in fp32)
    float c0 = 0.f;
#pragma unroll
    for (int u = 0; u < NK; ++u) c0 += key[u] != 0ull ? 1.f : 0.f;
    const bool fits = group_sum<G>(c0) <= fk;                 // everything that passed thr fits
    unsigned hi[NK];
#pragma unroll
    for (int u = 0; u < NK; ++u) hi[u] = (unsigned)(key[u] >> 32);
    bool done = fits, exact = false;
    // the score bits all of the group's keys share need no search (wave_topk_keys_n's PREFIX note); the wave starts at
    // the highest bit in which any of its groups' keys differ
    unsigned o = 0u, an = 0xFFFFFFFFu;
#pragma unroll
    for (int u = 0; u < NK; ++u)
        if (key[u] != 0ull) { o |= hi[u]; an &= hi[u]; }
#pragma unroll
    for (int m = 1; m < G; m <<= 1) { o |= __shfl_xor(o, m, 64); an &= __shfl_xor(an, m, 64); }
    const unsigned diff = o ^ an;                             // (a group without keys: all ones - and `fits`)
    const int top = diff ? 31 - __clz(diff) : -1;
    unsigned Th = top >= 0 ? (an & ~((2u << top) - 1u)) : an;
    for (int b = wave_max_i(done ? -1 : top); b >= 0; --b) {
        if (__all(done)) break;
        const unsigned cand = Th | (1u << b);
        float c = 0.f;
#pragma unroll
        for (int u = 0; u < NK; ++u) c += hi[u] >= cand ? 1.f : 0.f;
        c = group_sum<G>(c);
        if (!done && c >= fk) {
            Th = cand;
            if (c == fk) { exact = true; done = true; }       // no further bit changes the set
        }
    }
    unsigned long long T = ((unsigned long long)Th << 32) | (0xFFFFFFFFull & ~((1ull << lowbits) - 1ull));
    for (int b = lowbits - 1; b >= 0; --b) {
        if (__all(done)) break;
        const unsigned long long cand = T | (1ull << b);
        float c = 0.f;
#pragma unroll
        for (int u = 0; u < NK; ++u) c += key[u] >= cand ? 1.f : 0.f;
        c = group_sum<G>(c);
        if (!done && c >= fk) {
            T = cand;
            if (c == fk) done = true;
        }
    }
#pragma unroll
    for (int u = 0; u < NK; ++u) kept[u] = fits ? key[u] != 0ull : (exact ? hi[u] >= Th : key[u] >= T);
}

// Rows [pb, pb + 64 / G) of at most 128 candidates each, one per lane group.  seen (per group): the row's done
// words were all read as done by the batch before (the look-ahead below); on return: whether row pb_next + gid's were.
template <int VEC, int G, int R>
__device__ __forceinline__ void fin_group_batch(const FwdArgs &a, int pb, int pb_next, int *lds_wave, bool &dead,
                                                bool &seen)
{
    using RowT = Row<VEC, G, R>;
    constexpr int NG = 64 / G, NKG = 128 / G;
    constexpr unsigned long long FULL = G == 64 ? ~0ull : ((1ull << (G & 63)) - 1ull);
    const int lane = lane_id();
    int gid = lane / G, lg = lane % G;
    // (opaque per call: what depends on the lane only - eight slot offsets, masks, shuffle addresses - was hoisted
    // out of the caller's loop over batches and held in ~50 registers across it: 131 against the kernel's 80)
    asm volatile("" : "+v"(gid), "+v"(lg));
    const int p = pb + gid;
    const bool valid = p < a.n_split;
    const int pc = valid ? p : a.n_split - 1;
    const int4 d = a.rdesc[pc];
    const int i = d.x, rs = d.y, deg = d.z;
    const bool act = valid && !a.skip_row(i);                 // (group-uniform; a skipped row's tasks skipped it too)
    const int t0 = a.split_task0[pc], t1 = a.split_task0[pc + 1];
    unsigned long long *s_key = reinterpret_cast<unsigned long long *>(lds_wave + gid * FIN_GROUP_WORDS);
    int *s_src = lds_wave + gid * FIN_GROUP_WORDS + 2 * CAND_MAX_K;
    // wait for the rows' tasks
    bool ready = seen || !act;
    for (int spin = 0; !dead && spin < FIN_SPIN_MAX && !__all(ready); ++spin) {
        bool mine = true;
        for (int t = t0 + lg; t < t1; t += G) mine = mine && ld_agent(a.fin_done + t) == a.fin_nonce;
        ready = ready || fwd_group_bits<G>(__ballot(mine), gid) == FULL;
        if (!__all(ready)) __builtin_amdgcn_s_sleep(16);
    }
    if (!__all(ready)) dead = true;
    const bool run = act && ready;
    if (act && !ready)
        for (int c = lg; c < a.C; c += G) a.out[(size_t)i * a.C + c] = __uint_as_float(0x7FC00000u);
    if (run)
        for (int t = t0 + lg; t < t1; t += G) st_agent(a.fin_done + t, 0ull);     // consumed (a replayed launch finds zeros)
    // look ahead: the next batch's done words, in flight under this batch's round trips (by now its tasks are
    // long done - one round trip less per batch; a "done" read early stays true: only this wave clears the words)
    bool mine_next = pb_next + gid < a.n_split;
    if (mine_next) {
        const int u0 = a.split_task0[pb_next + gid], u1 = a.split_task0[pb_next + gid + 1];
        for (int t = u0 + lg; t < u1; t += G) mine_next = mine_next && ld_agent(a.fin_done + t) == a.fin_nonce;
    }
    // candidates: slot q = lg + G u of the row
    const int n = (t1 - t0) * a.k;                            // <= 128: the rows behind FwdArgs::n_split_gt_wave
    auto slot = [&](int q) { return fin_slot(a, t0, q); };
    unsigned long long key[NKG];
    int src[NKG];                                             // (with the keys: one round trip, not two)
#pragma unroll
    for (int u = 0; u < NKG; ++u) {
        const int q = lg + G * u;
        const bool in = run && q < n;
        key[u] = in ? ld_agent(a.cand_key + slot(q)) : 0ull;
        src[u] = in ? ld_agent(a.cand_src + slot(q)) : 0;
    }
    bool kp[NKG];
    group_topk_keys_n<G, NKG>(key, a.k, a.lowbits, kp);
    int nsel = 0;                                             // winners in ascending slot order (fin_wave_row's order)
#pragma unroll
    for (int u = 0; u < NKG; ++u) {
        const unsigned long long m = fwd_group_bits<G>(__ballot(kp[u]), gid);
        if (kp[u]) {
            const int o = nsel + __popcll(m & ((1ull << lg) - 1ull));
            s_key[o] = key[u];
            s_src[o] = src[u];
        }
        nsel += __popcll(m);
    }
    wave_lds_sync();
    // the weighted sum, in the order of fin_winners_row: there lane group g adds the winners w = g (mod 64 / G) in
    // ascending order and the groups' sums meet in a butterfly - here ONE group holds those 64 / G partial sums
    RowT part[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) part[g].zero();
    // rows in flight per group: a multiple of 64 / G.  (Eight at R == 1: 84 registers - three spilled, 12-16 bytes of
    // scratch per lane for EVERY wave of the launch - and no faster: 64.6-64.9 against 64.9-65.2 us per forward,
    // while the launch without the role lost 0.4 us to the scratch set-up.)
    constexpr int STEP = R == 1 ? (NG > 4 ? NG : 4) : (R == 2 ? 4 : 2);
    static_assert(STEP % NG == 0, "the partial sum of a winner must be a compile-time index");
    const int nsel_max = wave_max_i(nsel);
    for (int w0 = 0; w0 < nsel_max; w0 += STEP) {
        RowT x[STEP];
        float nj[STEP];
#pragma unroll
        for (int u = 0; u < STEP; ++u) {
            const int j = nsel > 0 ? s_src[min(w0 + u, nsel - 1)] : 0;
            x[u].load(a.n + (size_t)j * a.C, a.C, lg);
            nj[u] = a.nrm ? a.nrm[j] : 1.0f;
        }
#pragma unroll
        for (int u = 0; u < STEP; ++u)
            if (w0 + u < nsel) part[u % NG].axpy(key_score(s_key[w0 + u]) * nj[u], x[u]);
    }
    const bool emit = a.sel_src != nullptr;
    for (int w = lg; w < nsel; w += G) {
        const unsigned long long kq = s_key[w];
        const float sq = key_score(kq);
        if (a.wsel) a.wsel[rs + key_index(kq)] = sq;
        if (a.kbits) atomicOr(a.kbits + a.kb_tbase + 4 * t0 + (key_index(kq) >> 5), 1u << (key_index(kq) & 31));
        if (emit) {
            int rk = 0;
            for (int r = 0; r < nsel; ++r) rk += s_key[r] > kq;
            a.sel_src[(size_t)i * a.k + rk] = s_src[w];
            a.sel_w[(size_t)i * a.k + rk] = sq;
        }
    }
#pragma unroll
    for (int m = 1; m < NG; m <<= 1)
#pragma unroll
        for (int g = 0; g < NG; g += 2 * m) part[g].add(part[g + m]);
    part[0].div((float)deg);
    row_epilogue<VEC, G, R>(a, part[0], i, lg);
    if (run) part[0].store(a.out + (size_t)i * a.C, a.C, lg);
    seen = fwd_group_bits<G>(__ballot(mine_next), gid) == FULL;
    wave_lds_sync();            // the wave's LDS scratch is reused by its next batch
}

// A split row of a call that selects nothing (top_k < 0: SNConv): the sum of its tasks' partial rows, one wave per
// row, a lane per channel - in the finalize launch's own order (sixteen slices, slice s = tasks s, s + 16, .. added
// in turn, then the slices in order: fin_cand_row / fin_wave_row), sixteen loads in flight.  seen / p_next: the
// look-ahead of fin_group_batch.
template <int VEC, int G, int R>
__device__ __forceinline__ void fin_stream_row(const FwdArgs &a, int p, int p_next, bool &dead, bool &seen)
{
    int lane = lane_id();
    asm volatile("" : "+v"(lane));                          // (opaque per call: fin_group_batch's note)
    const int4 d = a.rdesc[p];
    const int i = d.x, deg = d.z;
    const bool was_seen = seen;
    seen = false;
    if (a.skip_row(i)) return;                              // (wave-uniform; its tasks skipped it too)
    const int t0 = a.split_task0[p], t1 = a.split_task0[p + 1];
    bool ready = was_seen;
    for (int spin = 0; !ready && !dead && spin < FIN_SPIN_MAX; ++spin) {
        bool mine = true;
        for (int t = t0 + lane; t < t1; t += 64) mine = mine && ld_agent(a.fin_done + t) == a.fin_nonce;
        if (__all(mine)) ready = true;
        else __builtin_amdgcn_s_sleep(16);
    }
    if (!ready) {
        dead = true;
        for (int c = lane; c < a.C; c += 64) a.out[(size_t)i * a.C + c] = __uint_as_float(0x7FC00000u);
        return;
    }
    for (int t = t0 + lane; t < t1; t += 64) st_agent(a.fin_done + t, 0ull);      // consumed (a replayed launch finds zeros)
    bool mine_next = p_next < a.n_split;
    if (mine_next) {
        const int u0 = a.split_task0[p_next], u1 = a.split_task0[p_next + 1];
        for (int t = u0 + lane; t < u1; t += 64) mine_next = mine_next && ld_agent(a.fin_done + t) == a.fin_nonce;
    }
    for (int c0 = 0; c0 < a.C; c0 += 64) {
        const int c = c0 + lane;
        const bool in = c < a.C;
        float sl[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) sl[u] = 0.f;
        for (int tb = t0; tb < t1; tb += 16) {
#pragma unroll
            for (int h = 0; h < 16; h += 8) {                  // (eight loads in flight: sixteen spilled registers)
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    v[u] = (in && tb + h + u < t1) ? ld_agent(a.partial + (size_t)(tb + h + u) * a.C + c) : 0.f;
#pragma unroll
                for (int u = 0; u < 8; ++u) sl[h + u] += v[u];
            }
        }
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) sum += sl[u];
        if (in) a.out[(size_t)i * a.C + c] = a.epilogue(sum / (float)deg, i, c);
    }
    seen = __all(mine_next) != 0;
}

// Two rows of more than 128 candidates each, by the workgroup: waves 0, 1 take the halves of row p0's candidates,
// waves 2, 3 those of row p0 + 1 (none: they only keep the barriers company); the even wave of a pair merges.
// Called by all waves of the workgroup - every wave passes the same three barriers whatever its row's state.
// LDS per wave: its winners' keys [0, 64) words | source ids [64, 96) | count [96] | ready flag [97].
template <int VEC, int G, int R>
__device__ __forceinline__ void fin_block_pair(const FwdArgs &a, int p0, int n_big, int (*lds)[WaveLds<G>::WORDS],
                                               bool &dead)
{
    int lane = lane_id();
    asm volatile("" : "+v"(lane));                          // (opaque per call: fin_group_batch's note)
    const int wave = threadIdx.x >> 6, half = wave & 1;
    int *lw = lds[wave];
    const int p = p0 + (wave >> 1);
    const bool valid = p < n_big;
    const int pc = valid ? p : p0;
    const int4 d = a.rdesc[pc];
    const int i = d.x, rs = d.y, deg = d.z;
    const bool act = valid && !a.skip_row(i);               // (wave-uniform; a skipped row's tasks skipped it too)
    const int t0 = a.split_task0[pc], t1 = a.split_task0[pc + 1];
    unsigned long long *s_key_w = reinterpret_cast<unsigned long long *>(lw);
    int *s_src_w = lw + 2 * CAND_MAX_K;
    // both waves of a pair wait for all of the row's tasks, then they agree
    bool ready = !act;
    for (int spin = 0; !ready && !dead && spin < FIN_SPIN_MAX; ++spin) {
        bool mine = true;
        for (int t = t0 + lane; t < t1; t += 64) mine = mine && ld_agent(a.fin_done + t) == a.fin_nonce;
        if (__all(mine)) ready = true;
        else __builtin_amdgcn_s_sleep(16);
    }
    if (lane == 0) lw[97] = ready ? 1 : 0;
    __syncthreads();
    const bool run = act && ready && lds[wave ^ 1][97] != 0;
    if (act && !run) {
        dead = true;
        if (half == 0)
            for (int c = lane; c < a.C; c += 64) a.out[(size_t)i * a.C + c] = __uint_as_float(0x7FC00000u);
    }
    if (run && half == 0)
        for (int t = t0 + lane; t < t1; t += 64) st_agent(a.fin_done + t, 0ull);      // consumed (both waves have seen them)
    const int n = (t1 - t0) * a.k;
    auto slot = [&](int q) { return fin_slot(a, t0, q); };
    // this wave's half, in rounds of 512 keys: lanes [0, 32) of the first hold the running winners (k <=
    // CAND_MAX_K = 32), the rest 480 new candidates - all loads of a round in flight together; source ids: the
    // winners' only, behind the selection
    const int per = (n + 1) / 2;
    const int qlo = half * per, qhi = run ? min(n, qlo + per) : qlo;
    constexpr int NK = 8, NEW = 64 * NK - CAND_MAX_K;
    int nsel = 0;
    for (int q0 = qlo; q0 < qhi; q0 += NEW) {
        unsigned long long key[NK];
#pragma unroll
        for (int u = 0; u < NK; ++u) {
            const int q = q0 + u * 64 + lane - CAND_MAX_K;
            const bool in = (u > 0 || lane >= CAND_MAX_K) && q < qhi;
            key[u] = in ? ld_agent(a.cand_key + slot(q)) : 0ull;
        }
        int src_run = 0;
        const bool had = lane < nsel;                      // key[0] is a running winner (its source id is known)
        if (had) { key[0] = s_key_w[lane]; src_run = s_src_w[lane]; }
        wave_lds_sync();                                   // the winners are in registers before their slots are reused
        bool kp[NK];
        wave_topk_keys_n<NK, true>(key, a.k, a.lowbits, kp);
        int off = 0;
#pragma unroll
        for (int u = 0; u < NK; ++u) {
            const unsigned long long m = __ballot(kp[u]);
            if (kp[u]) {
                const int o = off + prefix_popc(m);
                s_key_w[o] = key[u];
                s_src_w[o] = (u == 0 && had) ? src_run : ld_agent(a.cand_src + slot(q0 + u * 64 + lane - CAND_MAX_K));
            }
            off += __popcll(m);
        }
        nsel = off;
        wave_lds_sync();
    }
    if (lane == 0) lw[96] = nsel;
    __syncthreads();
    // the even wave: the pair's winners (<= 32 each) -> one selection
    unsigned long long key0 = 0ull;
    int src0 = 0;
    if (half == 0) {
        const int wa = wave + (lane >> 5), idx = lane & 31;
        if (idx < lds[wa][96]) {
            key0 = reinterpret_cast<const unsigned long long *>(lds[wa])[idx];
            src0 = lds[wa][2 * CAND_MAX_K + idx];
        }
    }
    __syncthreads();                                        // (read before the odd waves reuse their regions)
    if (half != 0 || !run) return;
    bool k0, k1;
    wave_topk_cands(key0, 0ull, a.k, a.lowbits, k0, k1);
    const unsigned long long m0 = __ballot(k0);
    const int nfin = __popcll(m0);
    if (k0) { const int o = prefix_popc(m0); s_key_w[o] = key0; s_src_w[o] = src0; }
    wave_lds_sync();
    fin_winners_row<VEC, G, R, false>(a, p, i, rs, deg, t0, nfin, s_key_w, s_src_w, 0, 0u);
    wave_lds_sync();
}

// Persistent waves: wave w of the grid takes work items w, w + n_waves, ... of the
// list [split-row tasks | wave rows | small-row sets], each class in order of
// descending degree, so every wave gets a similar mix and the grid drains evenly.
// EPI: the hidden-layer store epilogue (FwdArgs::epilogue) compiled into the row stores - its own
// instantiation, so the plain forward keeps its register allocation (a runtime flag cost the
// filter variant two spilled registers and 2 us)
// FS: the filter also in front of the small rows (small_rows_set_filt) - its own instantiation: compiled into
// the plain FILT kernel it cost that kernel's other calls 2.5 us (41.8 against 39.3 us at top_k 1 / thr 0.99
// without the small-row form: six spilled registers)
// FIN: the launch's last workgroups finalize the split rows (role_task's note); its own instantiation, so the
// launches without the role run the kernel they ran before it existed
template <int VEC, int G, int R, bool FILT, bool OTF, int EPI = 0, bool FS = false, bool FIN = false>
__global__ __launch_bounds__(BLOCK, FWD_WAVES_PER_SIMD) void k_agg_fwd(const FwdArgs a)
{
    static_assert(!(FILT && OTF), "the filter belongs to the table mode");
    static_assert(FILT || !FS, "the small-row filter belongs to the FILT kernel");
    __shared__ __align__(16) int lds[WAVES][WaveLds<G>::WORDS];
    const int wave = threadIdx.x >> 6;
    int *lw = lds[wave];
    constexpr int RPW = 64 / G;
    const int nsets = (a.N - a.n_med_end + RPW - 1) / RPW;
    if constexpr (FIN) if ((int)blockIdx.x >= a.main_blocks) {                   // (workgroup-uniform) the finalize role
        // Its arguments through the kernel-argument segment itself, behind an opaque pointer: read through `a`,
        // the role's two dozen fields were loaded at the kernel's entry and held in scalar registers across the
        // OTHER roles (55-72 spilled scalars, 6-8 spilled vector registers, 60 bytes of scratch per lane in
        // every instantiation); this way they are loaded here, and the work-item roles compile as before.
        auto kp = __builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        const FwdArgs &fa = *(const FwdArgs *)kp;             // (the kernel's only explicit argument: offset 0)
        const int nfb = (int)gridDim.x - fa.main_blocks, fb = (int)blockIdx.x - fa.main_blocks;
        bool dead = false, seen = false;
        if (fa.k < 0) {                                       // (uniform) nothing selected: sums of partial rows
            for (int p = fb * WAVES + wave; p < fa.n_split; p += nfb * WAVES)
                fin_stream_row<VEC, G, R>(fa, p, p + nfb * WAVES, dead, seen);
            return;
        }
        const int n_big = min(fa.n_split, fa.n_split_gt_wave);
        for (int p0 = 2 * fb; p0 < n_big; p0 += 2 * nfb) fin_block_pair<VEC, G, R>(fa, p0, n_big, lds, dead);
        for (int pb = n_big + (fb * WAVES + wave) * RPW; pb < fa.n_split; pb += nfb * WAVES * RPW)
            fin_group_batch<VEC, G, R>(fa, pb, pb + nfb * WAVES * RPW, lw, dead, seen);
        // (Small-row sets kept back for these waves to take behind their rows - 400 .. 1 600 of 40 383 - moved
        // nothing: 50.9-51.6 us against 51.0; the launch is 6 % slower than without the role's 96 workgroups, the
        // share of the chip's wave slots they hold.)
        return;
    }
    const int nw = a.main_blocks * WAVES;
    const int n_wave_rows = a.n_med_end - a.n_split;
    int it = blockIdx.x * WAVES + wave;
    for (; it < a.n_tasks; it += nw)
        if (a.role_mask & 1) role_task<VEC, G, R, FILT, OTF, FIN>(a, a.task_order[it], lw);
    it -= a.n_tasks;
    for (; it < n_wave_rows; it += nw)
        if (a.role_mask & 2) role_wave<VEC, G, R, FILT, OTF, EPI>(a, it, lw);
    it -= n_wave_rows;
    if (a.role_mask & 4) role_small<VEC, G, R, OTF, EPI, FS>(a, it, nw, nsets, lw);
}

// ---------------------------------------------------------------------------
// Finalize of split rows from scores in HBM scratch (top_k > CAND_MAX_K, or a hub whose
// candidates do not fit LDS): one FIN_BLOCK-thread workgroup per row, any degree.
// ---------------------------------------------------------------------------
struct __align__(16) FinShared {
    int red[2][FIN_BLOCK / 64];
    int wave_off[FIN_BLOCK / 64];
    int nsel;
    int pad[3];
};
static_assert(sizeof(FinShared) % 16 == 0, "keep the dynamic LDS base 16-byte aligned");
static_assert(WAVE_T == 128 && SMALL_T == 16 && CHUNK == 128, "LDS layouts of the main kernel assume these");

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
    if (a.skip_row(i)) return;                              // (workgroup-uniform)
    const int rs = a.rowptr[i];
    const int deg = a.rowptr[i + 1] - rs;
    const bool emit = a.sel_src != nullptr && a.k >= 0;
    const bool rank = a.k >= 0 && deg > a.k;

    // dynamic LDS: [C * NW] partial rows | [k] kept list | [lds_scores] scores
    float *s_part = reinterpret_cast<float *>(dyn);
    int *s_list = reinterpret_cast<int *>(s_part + (size_t)a.C * NW);
    float *s_scl = reinterpret_cast<float *>(s_list + (a.k < 0 ? 0 : min(a.k, deg)));
    const int t0 = a.split_task0[p], t1 = a.split_task0[p + 1];

    if (!rank && !emit) {
        // streaming row: add the tasks' partial rows in task order
        for (int c = tid; c < a.C; c += FIN_BLOCK) {
            float s = 0.f;
            for (int t = t0; t < t1; ++t) s += a.partial[(size_t)t * a.C + c];
            a.out[(size_t)i * a.C + c] = a.epilogue(s / (float)deg, i, c);
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
                const int j = a.col[rs + idx];
                RowT x;
                x.load(a.n + (size_t)j * a.C, a.C, lg);
                acc.axpy(sc[idx] * (a.nrm ? a.nrm[j] : 1.0f), x);        // (OTF: the rows are h itself)
            }
        }
        acc.reduce_across_groups();
        if (gid == 0) acc.store(s_part + (size_t)wave * a.C, a.C, lg);
        __syncthreads();
        for (int ch = tid; ch < a.C; ch += FIN_BLOCK) {
            float s = 0.f;
            for (int w = 0; w < NW; ++w) s += s_part[(size_t)w * a.C + ch];
            a.out[(size_t)i * a.C + ch] = a.epilogue(s / (float)deg, i, ch);
        }
    } else {
        // emit on a streaming row: the sum itself still comes from the partials
        for (int ch = tid; ch < a.C; ch += FIN_BLOCK) {
            float s = 0.f;
            for (int t = t0; t < t1; ++t) s += a.partial[(size_t)t * a.C + ch];
            a.out[(size_t)i * a.C + ch] = a.epilogue(s / (float)deg, i, ch);
        }
    }
}

// ---------------------------------------------------------------------------
// Finalize of split rows from chunk-local candidates (top_k <= CAND_MAX_K):
// one 1024-thread workgroup per row merges the tasks' candidate keys 128 at a
// time (tournament of wave-level top-k) and gathers the <= k winners.
// ---------------------------------------------------------------------------
constexpr int FIN_WAVE_MIN_ROWS = 2048;   // fewer moderate split rows than this: one launch (k_agg_fin_cand) for all
// Waves per workgroup of the candidate finalize (template NW): 16 where the biggest row's tournament has more
// than 1 024 candidates to get through (its first level runs wide: arxiv size at top_k 16, 1 632 keys:
// 7.6 us against 8.5 with 8 waves), else 8 - a smaller workgroup is less launch to pay (top_k 1: 1.9
// against 2.7 us) - and 8 whenever the head role rides in the launch (75 registers: three 512-thread
// workgroups fit a CU, one of 1 024 threads).
constexpr int FINC_WAVES_MAX = 16;
constexpr size_t FINC_LDS_BUDGET = 150 * 1024;
inline size_t finc_lds_bytes(int C, int max_slots, int /*nw*/ = FINC_WAVES_MAX) { return ((size_t)FINC_WAVES_MAX * C + 1) * 4 + (size_t)max_slots * 24 + 16; }

// a split row's head (FwdArgs::head_sel): the finished mean row sits in LDS (s_row[0, C)); the first lane
// group of wave 0 takes it through head_store_row and the row's entry of head_part is written.
// Called by the whole workgroup, behind the barrier that made s_row complete.
template <int VEC, int G, int R>
__device__ __forceinline__ void fin_row_head(const FwdArgs &a, int i, int p, const float *s_row)
{
    if constexpr (VEC == 4 && R == 1 && (G == 8 || G == 16)) {
        if (threadIdx.x >= 64) return;                         // (wave-uniform)
        const int lane = lane_id();
        const int gid = lane / G, lg = lane % G;
        HeadAcc ha;
        if (gid == 0) {
            Row<VEC, G, R> row;
            const float4 t = *reinterpret_cast<const float4 *>(s_row + (4 * lg < a.C ? 4 * lg : 0));
            row.x[0][0] = t.x; row.x[0][1] = t.y; row.x[0][2] = t.z; row.x[0][3] = t.w;
            head_store_row<VEC, G, R>(a, row, i, lg, (int)a.head_y[i], a.head_sel[i], ha);
        }
        head_write_entry(a, a.head_nmain + p, ha);
    }
}

template <int VEC, int G, int R, int NW, bool HEAD>
__device__ __forceinline__ void fin_cand_row(const FwdArgs &a, int p, int max_slots, unsigned char *dyn)
{
    using RowT = Row<VEC, G, R>;
    constexpr int NG = 64 / G;
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const int gid = lane / G, lg = lane % G;
    const int4 d = a.rdesc[p];
    const int i = d.x, rs = d.y, deg = d.z;
    if (a.skip_row(i)) return;                              // (workgroup-uniform)
    const int t0 = a.split_task0[p], t1 = a.split_task0[p + 1];
    const bool emit = a.sel_src != nullptr && a.k >= 0;
    const bool rank = a.k >= 0 && deg > a.k;

    // dynamic LDS: [FINC_WAVES_MAX * C] partial rows | keysA | keysB [max_slots] | srcA | srcB | count
    float *s_part = reinterpret_cast<float *>(dyn);
    unsigned long long *kA = reinterpret_cast<unsigned long long *>(s_part + (size_t)FINC_WAVES_MAX * a.C + ((FINC_WAVES_MAX * a.C) & 1));
    unsigned long long *kB = kA + max_slots;
    int *sA = reinterpret_cast<int *>(kB + max_slots);
    int *sB = sA + max_slots;
    int &s_n = sB[max_slots];

    if (!rank) {
        // streaming row (deg <= top_k or no selection): add the tasks' partial rows.  The
        // FINC_WAVES_MAX slices, slice s = the sum of every FINC_WAVES_MAX-th task from s on (a wave takes
        // the slices wave, wave + NW, ..: their loads in flight together), then the slices are added in
        // slice order: a fixed order WHATEVER the workgroup size, and a hub's ~100 partial
        // rows cost a handful of memory round trips instead of one each.
        for (int c0 = 0; c0 < a.C; c0 += 64) {
            const int c = c0 + lane;
            for (int sl = wave; sl < FINC_WAVES_MAX; sl += NW) {
                float s = 0.f;
                if (c < a.C)
                    for (int t = t0 + sl; t < t1; t += FINC_WAVES_MAX) s += a.partial[(size_t)t * a.C + c];
                if (c < a.C) s_part[(size_t)sl * a.C + c] = s;
            }
        }
        __syncthreads();
        for (int c = tid; c < a.C; c += (NW * 64)) {
            float s = 0.f;
            for (int w = 0; w < FINC_WAVES_MAX; ++w) s += s_part[(size_t)w * a.C + c];
            if constexpr (HEAD) s_part[c] = s / (float)deg;     // (C <= 64 <= (NW * 64): the thread's own channel only)
            else a.out[(size_t)i * a.C + c] = a.epilogue(s / (float)deg, i, c);
        }
        if constexpr (HEAD) {
            __syncthreads();
            fin_row_head<VEC, G, R>(a, i, p, s_part);
        }
        if (emit) {
            // selection of a streaming split row: every edge >= thr, ranked.  Rare
            // (needs top_k >= deg > WAVE_T); done by plain counting from the scratch scores.
            const float *sc = a.scores + a.split_soff[p];
            for (int e = tid; e < deg; e += (NW * 64)) {
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
    for (int q = tid; q < n; q += (NW * 64)) {
        kA[q] = a.cand_key[(size_t)(t0 + q / a.k) * CAND_MAX_K + q % a.k];
        sA[q] = a.cand_src[(size_t)(t0 + q / a.k) * CAND_MAX_K + q % a.k];
    }
    __syncthreads();
    // tournament level: 256 keys per wave (4 per lane), so the biggest hub of an arxiv-like graph
    // (13 k in-edges: 102 tasks x 16 = 1 632 keys) is down to 7 x 16 = 112 keys - one wave-level
    // selection - after ONE level (128 per wave needed two levels and a third selection)
    while (n > 128) {
        const int groups = (n + 255) / 256;
        for (int g = wave; g < groups; g += NW) {
            unsigned long long key[4];
            int q[4];
            bool kp[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                q[u] = g * 256 + u * 64 + lane;
                key[u] = q[u] < n ? kA[q[u]] : 0ull;
            }
            wave_topk_keys_n<4, true>(key, a.k, a.lowbits, kp);
            int off = g * a.k;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned long long m = __ballot(kp[u]);
                if (kp[u]) { const int o = off + prefix_popc(m); kB[o] = key[u]; sB[o] = sA[q[u]]; }
                off += __popcll(m);
            }
            if (g * a.k + lane >= off && lane < a.k) kB[g * a.k + lane] = 0ull;      // empty slots (k <= 32 < 64)
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
        wave_topk_cands(key0, key1, a.k, a.lowbits, k0, k1);
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
    for (int q0 = 0; q0 < nsel; q0 += NW * NG) {
        const int q = q0 + wave * NG + gid;
        if (q < nsel) {
            const int j = wsrc[q];
            RowT x;
            x.load(a.n + (size_t)j * a.C, a.C, lg);
            acc.axpy(key_score(win[q]) * (a.nrm ? a.nrm[j] : 1.0f), x);
        }
    }
    for (int q = tid; q < nsel; q += (NW * 64)) {
        const unsigned long long kq = win[q];
        const float sq = key_score(kq);
        if (a.wsel) a.wsel[rs + key_index(kq)] = sq;
        if (a.kbits) atomicOr(a.kbits + a.kb_tbase + 4 * t0 + (key_index(kq) >> 5), 1u << (key_index(kq) & 31));
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
    for (int ch = tid; ch < a.C; ch += (NW * 64)) {
        float s = 0.f;
        for (int w = 0; w < NW; ++w) s += s_part[(size_t)w * a.C + ch];
        if constexpr (HEAD) s_part[ch] = s / (float)deg;
        else a.out[(size_t)i * a.C + ch] = a.epilogue(s / (float)deg, i, ch);
    }
    if constexpr (HEAD) {
        __syncthreads();
        fin_row_head<VEC, G, R>(a, i, p, s_part);
    }
}

template <int VEC, int G, int R, int NW, bool HEAD>
__global__ __launch_bounds__(NW * 64) void k_agg_fin_cand(const FwdArgs a, int max_slots)
{
    extern __shared__ __align__(16) unsigned char dyn[];   // no static LDS in front of it
    fin_cand_row<VEC, G, R, NW, HEAD>(a, blockIdx.x, max_slots, dyn);
}

// The winners of a split row (s_key_w / s_src_w [0, nsel): keys and source ids, in the selection's own order) ->
// the row's weighted sum, its mean and stores, the bookkeeping of the kept edges.  One wave.
template <int VEC, int G, int R, bool HEAD>
__device__ __forceinline__ void fin_winners_row(const FwdArgs &a, int p, int i, int rs, int deg, int t0, int nsel,
                                                const unsigned long long *s_key_w, const int *s_src_w, int head_yy,
                                                unsigned head_sv)
{
    using RowT = Row<VEC, G, R>;
    constexpr int NG = 64 / G;
    const int lane = lane_id();
    const int gid = lane / G, lg = lane % G;
    const bool emit = a.sel_src != nullptr;
    RowT acc;
    acc.zero();
    constexpr int GU = R >= 3 ? 2 : 4;                        // winner rows in flight per lane group
    for (int w0 = 0; w0 < nsel; w0 += GU * NG) {
        RowT x[GU];
        float nj[GU];
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int w = min(w0 + u * NG + gid, nsel - 1);
            const int j = s_src_w[w];
            x[u].load(a.n + (size_t)j * a.C, a.C, lg);
            nj[u] = a.nrm ? a.nrm[j] : 1.0f;
        }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int w = w0 + u * NG + gid;
            if (w < nsel) acc.axpy(key_score(s_key_w[w]) * nj[u], x[u]);
        }
    }
    if (lane < nsel) {
        const unsigned long long kq = s_key_w[lane];
        const float sq = key_score(kq);
        if (a.wsel) a.wsel[rs + key_index(kq)] = sq;
        if (a.kbits) atomicOr(a.kbits + a.kb_tbase + 4 * t0 + (key_index(kq) >> 5), 1u << (key_index(kq) & 31));
        if (emit) {
            int rk = 0;
            for (int r = 0; r < nsel; ++r) rk += s_key_w[r] > kq;
            a.sel_src[(size_t)i * a.k + rk] = s_src_w[lane];
            a.sel_w[(size_t)i * a.k + rk] = sq;
        }
    }
    acc.reduce_across_groups();
    acc.div((float)deg);
    if constexpr (HEAD) {
        HeadAcc ha;
        if (gid == 0) head_store_row<VEC, G, R>(a, acc, i, lg, head_yy, head_sv, ha);
        head_write_entry(a, a.head_nmain + p, ha);
        return;
    }
    row_epilogue<VEC, G, R>(a, acc, i, lg);
    if (gid == 0) acc.store(a.out + (size_t)i * a.C, a.C, lg);
}

// The same finalize for split rows whose candidates fit one wave-level selection
// ((tasks) * k <= 128, i.e. deg <= 8 * CHUNK at k = 16): one WAVE per row, no workgroup
// barrier.  On graphs with many moderately large rows
// (products-like: ~10^5 split rows) the 1024-thread tournament above is mostly idle.
// s_key_w / s_src_w: the wave's own CAND_MAX_K LDS slots.
template <int VEC, int G, int R, bool HEAD>
__device__ __forceinline__ void fin_wave_row(const FwdArgs &a, int p, unsigned long long *s_key_w, int *s_src_w)
{
    using RowT = Row<VEC, G, R>;
    const int lane = lane_id();
    const int gid = lane / G, lg = lane % G;
    const int4 d = a.rdesc[p];
    const int i = d.x, rs = d.y, deg = d.z;
    if (a.skip_row(i)) return;                              // (wave-uniform)
    const int t0 = a.split_task0[p], t1 = a.split_task0[p + 1];
    int head_yy = 0;
    unsigned head_sv = 0u;
    if constexpr (HEAD) { head_yy = (int)a.head_y[i]; head_sv = a.head_sel[i]; }
    if (a.k < 0) {
        // (rows of at most 16 tasks come here: their partial rows are loaded together)
        float *s_row = reinterpret_cast<float *>(s_key_w);      // head: the row through LDS (C <= 64 floats fit)
        for (int c = lane; c < a.C; c += 64) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = t0 + u < t1 ? a.partial[(size_t)(t0 + u) * a.C + c] : 0.f;
            float s = 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u];
            for (int t = t0 + 16; t < t1; ++t) s += a.partial[(size_t)t * a.C + c];
            if constexpr (HEAD) s_row[c] = s / (float)deg;
            else a.out[(size_t)i * a.C + c] = a.epilogue(s / (float)deg, i, c);
        }
        if constexpr (HEAD) {
            if constexpr (VEC == 4 && R == 1 && (G == 8 || G == 16)) {
                wave_lds_sync();
                HeadAcc ha;
                if (gid == 0) {
                    RowT row;
                    const float4 t = *reinterpret_cast<const float4 *>(s_row + (4 * lg < a.C ? 4 * lg : 0));
                    row.x[0][0] = t.x; row.x[0][1] = t.y; row.x[0][2] = t.z; row.x[0][3] = t.w;
                    head_store_row<VEC, G, R>(a, row, i, lg, head_yy, head_sv, ha);
                }
                head_write_entry(a, a.head_nmain + p, ha);
            }
        }
        return;
    }
    const int n = (t1 - t0) * a.k;                            // <= 128 by the launch split
    auto slot = [&](int q) { return (size_t)(t0 + q / a.k) * CAND_MAX_K + q % a.k; };
    const unsigned long long key0 = lane < n ? a.cand_key[slot(lane)] : 0ull;
    const unsigned long long key1 = lane + 64 < n ? a.cand_key[slot(lane + 64)] : 0ull;
    const int src0 = lane < n ? a.cand_src[slot(lane)] : 0;
    const int src1 = lane + 64 < n ? a.cand_src[slot(lane + 64)] : 0;
    bool k0, k1;
    wave_topk_cands(key0, key1, a.k, a.lowbits, k0, k1);
    const unsigned long long m0 = __ballot(k0), m1 = __ballot(k1);
    const int n0 = __popcll(m0), nsel = n0 + __popcll(m1);
    if (k0) { const int o = prefix_popc(m0); s_key_w[o] = key0; s_src_w[o] = src0; }
    if (k1) { const int o = n0 + prefix_popc(m1); s_key_w[o] = key1; s_src_w[o] = src1; }
    wave_lds_sync();
    fin_winners_row<VEC, G, R, HEAD>(a, p, i, rs, deg, t0, nsel, s_key_w, s_src_w, head_yy, head_sv);
}

// 4 rows per 256-thread workgroup
template <int VEC, int G, int R, bool HEAD>
__global__ __launch_bounds__(BLOCK) void k_agg_fin_wave(const FwdArgs a, int first, int count)
{
    __shared__ unsigned long long s_key[WAVES][CAND_MAX_K];
    __shared__ int s_src[WAVES][CAND_MAX_K];
    const int wave = threadIdx.x >> 6;
    const int q = blockIdx.x * WAVES + wave;
    if (q >= count) return;                                   // wave-uniform
    fin_wave_row<VEC, G, R, HEAD>(a, first + q, s_key[wave], s_src[wave]);
}
// Both in ONE launch when the moderate split rows are few (arxiv-like graphs: a few hundred
// split rows in all): workgroups [0, n_big) run the tournament of one big row each, the
// others one moderate row per wave - 140 workgroups that all start at once instead of 825
// tournaments of which 512 fit the chip (finalize 8.8 -> see DESIGN.md 4.1).
// HEAD: its own instantiation (the head's code inside the plain one cost it 9 registers, 12 bytes of
// scratch per lane and ~0.5 us of the headline step)
template <int VEC, int G, int R, int NW, bool HEAD>
__global__ __launch_bounds__(NW * 64) void k_agg_fin_mixed(const FwdArgs a, int max_slots, int n_big, int n_split,
                                                              int n_fin_blocks)
{
    extern __shared__ __align__(16) unsigned char dyn[];   // no static LDS in front of it
    if constexpr (HEAD) {
        if ((int)blockIdx.x >= n_fin_blocks) {             // (workgroup-uniform) the head role of the launch
            const int hb = (int)blockIdx.x - n_fin_blocks;
            const HeadAcc ha = head_rows_role<VEC, G, R>(a, hb * NW + (threadIdx.x >> 6), a.head_nmain * NW);
            head_block_entry(a, hb, ha, reinterpret_cast<float *>(dyn));
            return;
        }
    }
    if ((int)blockIdx.x < n_big) {                         // (workgroup-uniform)
        fin_cand_row<VEC, G, R, NW, HEAD>(a, blockIdx.x, max_slots, dyn);
        return;
    }
    const int wave = threadIdx.x >> 6;
    const int p = n_big + ((int)blockIdx.x - n_big) * NW + wave;
    if (p >= n_split) return;                              // wave-uniform
    unsigned long long *s_key = reinterpret_cast<unsigned long long *>(dyn) + wave * CAND_MAX_K;
    int *s_src = reinterpret_cast<int *>(dyn + NW * CAND_MAX_K * 8) + wave * CAND_MAX_K;
    fin_wave_row<VEC, G, R, HEAD>(a, p, s_key, s_src);
}

// whether split rows keep chunk-local candidates (else: scores to HBM scratch + k_agg_fin)
inline bool fwd_use_candidates(int top_k, int C, int max_split_deg)
{
    if (top_k < 0) return true;                 // no selection: partial rows, candidate launch shape
    if (top_k > CAND_MAX_K) return false;
    const int max_slots = std::max(1, ceil_div(max_split_deg, CHUNK) * std::max(top_k, 0));
    return finc_lds_bytes(C, max_slots) <= FINC_LDS_BUDGET;
}

// how the split rows of a call are finalized (rows are in descending degree order: the first n_big_true
// need the workgroup tournament, the rest fit one wave-level selection)
struct FinShape { int n_wave, n_big, n_big_true; bool mixed; };
inline FinShape finalize_shape(const FwdArgs &a)
{
    FinShape f;
    f.n_wave = a.n_split - std::min(a.n_split, a.n_split_gt_wave);
    // (no selection, k < 0: the split is by the number of partial rows to add, > 16 tasks -> workgroup)
    f.n_big = a.k < 0 ? a.n_split - f.n_wave
                      : ((a.k > 0 && f.n_wave >= FIN_WAVE_MIN_ROWS) ? a.n_split - f.n_wave : (a.k > 0 ? a.n_split : 0));
    f.n_big_true = a.n_split - f.n_wave;         // rows that need the tournament
    // few moderate rows (arxiv-like graphs: a few hundred split rows in all; products-like ones have 10^5): ONE launch
    f.mixed = a.n_split > 0 && a.use_cand && a.k > 0 && f.n_big == a.n_split && f.n_wave > 0 && f.n_big_true < a.n_split;
    return f;
}
inline bool head_role_in_finalize(const FwdArgs &a) { return finalize_shape(a).mixed; }

// the head role as a launch of its own (finalize shapes that do not carry it; graphs without split rows)
template <int VEC, int G, int R>
int launch_head_rows(const FwdArgs &a, hipStream_t st)
{
    if (a.head_sel && a.head_nmain > 0) k_agg_head_rows<VEC, G, R><<<a.head_nmain, BLOCK, 0, st>>>(a);
    return SNGNN_OK;
}

// waves per workgroup of the candidate finalize for a call (FINC_WAVES_MAX's comment)
inline int finc_waves(const FwdArgs &a, int max_split_deg)
{
    const int64_t max_slots = (int64_t)ceil_div(max_split_deg, CHUNK) * std::max(a.k, 0);
    return (a.head_sel == nullptr && (a.k < 0 || max_slots > 1024)) ? 16 : 8;
}

// the candidate finalize's launches at NW waves per workgroup
template <int VEC, int G, int R, int NW, bool HEAD>
int launch_cand_finalize(const FwdArgs &a, int max_split_deg, hipStream_t st)
{
    const int max_tasks = ceil_div(max_split_deg, CHUNK);
    const int max_slots = std::max(1, max_tasks * std::max(a.k, 0));
    const size_t dyn = finc_lds_bytes(a.C, max_slots, NW);
    if (dyn > FINC_LDS_BUDGET) { set_error("internal: candidate finalize does not fit LDS"); return SNGNN_EINVAL; }
    if (dyn > 48 * 1024)
        SN_HIP(hipFuncSetAttribute((const void *)k_agg_fin_cand<VEC, G, R, NW, HEAD>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    const FinShape fs = finalize_shape(a);
    const int n_wave = fs.n_wave, n_big = fs.n_big, n_big_true = fs.n_big_true;
    if (fs.mixed) {
        // few moderate rows: one mixed launch
        const size_t dyn_mixed = std::max(dyn, (size_t)NW * CAND_MAX_K * 12);
        if (dyn_mixed > 48 * 1024)
            SN_HIP(hipFuncSetAttribute((const void *)k_agg_fin_mixed<VEC, G, R, NW, HEAD>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_mixed));
        const int n_fin_blocks = n_big_true + ceil_div(n_wave, NW);
        const int n_head_blocks = HEAD ? a.head_nmain : 0;
        k_agg_fin_mixed<VEC, G, R, NW, HEAD><<<n_fin_blocks + n_head_blocks, NW * 64, dyn_mixed, st>>>(
            a, max_slots, n_big_true, a.n_split, n_fin_blocks);
    } else {
        if constexpr (HEAD) launch_head_rows<VEC, G, R>(a, st);
        if (n_big > 0) k_agg_fin_cand<VEC, G, R, NW, HEAD><<<n_big, NW * 64, dyn, st>>>(a, max_slots);
        if (a.n_split > n_big)
            k_agg_fin_wave<VEC, G, R, HEAD><<<ceil_div(a.n_split - n_big, WAVES), BLOCK, 0, st>>>(a, n_big, a.n_split - n_big);
    }
    return SNGNN_OK;
}

// launches of the split rows' finalize (after their tasks, same stream)
template <int VEC, int G, int R>
int launch_split_finalize(const FwdArgs &a, int max_split_deg, hipStream_t st)
{
    if (a.n_split > 0 && a.use_cand) {
        // streaming rows and candidate tournament
        if constexpr (VEC == 4 && R == 1 && (G == 8 || G == 16)) {
            if (a.head_sel) return launch_cand_finalize<VEC, G, R, 8, true>(a, max_split_deg, st);
        }
        if (finc_waves(a, max_split_deg) == 16) return launch_cand_finalize<VEC, G, R, 16, false>(a, max_split_deg, st);
        return launch_cand_finalize<VEC, G, R, 8, false>(a, max_split_deg, st);
    } else if (a.n_split > 0) {
        const size_t fixed = (size_t)a.C * (FIN_BLOCK / 64) * 4 + (size_t)std::min(std::max(a.k, 0), max_split_deg) * 4;
        const size_t budget = 120 * 1024;
        if (fixed > 150 * 1024) { set_error("top_k too large for the split-row finalize"); return SNGNN_EINVAL; }
        int lds_scores = 0;
        if (fixed < budget) lds_scores = (int)std::min<size_t>((budget - fixed) / 4, (size_t)max_split_deg);
        const size_t dyn = fixed + (size_t)lds_scores * 4;
        if (dyn > 48 * 1024)
            SN_HIP(hipFuncSetAttribute((const void *)k_agg_fin<VEC, G, R>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        k_agg_fin<VEC, G, R><<<a.n_split, FIN_BLOCK, dyn, st>>>(a, lds_scores);      // (no head there: the host refuses)
    } else {
        launch_head_rows<VEC, G, R>(a, st);
    }
    return SNGNN_OK;
}

// profiling (sngnn_profile_enable(reps)): every launch is issued `reps` times back to back
// between its two events, so the event pair's own cost is spread over reps launches
extern int g_prof_reps;
// sngnn_tuning_set(9, v): 0 = the split rows' finalize as a launch of its own (round 4's form), 1 = inside the main
// launch when it fits (launch_agg_fwd_impl), v > 1 = inside it on v workgroups
extern int g_fin_inline, g_last_fin_blocks;
unsigned long long next_fin_nonce();

template <int VEC, int G, int R, int EPI>
int launch_agg_fwd_impl(const FwdArgs &a0, int max_split_deg, hipEvent_t *ev, hipStream_t st)
{
    const int reps = ev ? std::max(g_prof_reps, 1) : 1;
    constexpr int RPW = 64 / G;
    const int n_small = a0.N - a0.n_med_end;
    const int64_t items = (int64_t)a0.n_tasks + (a0.n_med_end - a0.n_split) + ceil_div(n_small, RPW);
    FwdArgs a = a0;
    // the split rows' finalize inside this launch (fin_block_pair / fin_group_batch): rows that rank from candidates,
    // no head behind them - when it pays.  All figures in us, measured on MI355X (tools/sweep_fwd.py FIN=..,
    // arxiv and products size): a wave takes ~5.8 us per work item; the role starts ~6 us in (the tasks are the
    // first items), a pair of big rows takes a workgroup ~17 us, a batch of 64 / G moderate rows a wave ~12.5.
    //   * the role must end by 3/4 of the launch: the smallest number nb of workgroups that does;
    //   * each of them is six waves' worth of slots the work items lose: the launch grows by nb / (slots - nb);
    //   * against that, the launch it replaces: ~4.6 us + its rows over the whole chip, + the boundary (1.1).
    // Arxiv size, C 40: 48 workgroups, +1.6 us against 7.1 saved (measured: 68.1 -> 62.0 us per forward, hot device).
    // Products size: 64 workgroups, 4.65 -> 4.55 ms per forward.
    // Small graphs, narrow rows (C 32: a 35 us launch): nothing ends the role in time - a launch.
    //   Calls that select nothing (top_k < 0: fin_stream_row, one wave per row): ~4 us per row + 2.5 per sixteen tasks
    // of the biggest row.
    const bool fin_ok = g_fin_inline != 0 && a.n_split > 0 && a.use_cand && a.k != 0 && a.head_sel == nullptr &&
                        a.fin_done != nullptr && ((a.role_mask & 7) == 7 || g_fin_inline > 1);
    // (a forced role with the tasks masked out - knobs 9 and 0 - is the test of the bounded wait: the split rows come
    // back as NaN after FIN_SPIN_MAX polls, tests/test_fin_inline_gpu.py)
    const int fin_big = std::min(a.n_split, a.n_split_gt_wave);
    const int fin_batches = ceil_div(a.n_split - fin_big, RPW);
    constexpr int SLOTS = 256 * FWD_WAVES_PER_SIMD;
    // (eight rows per batch, C <= 32: sixteen keys per lane and eight partial sums - twice the time; measured at
    // arxiv size, C 32, top_k 1: the role on 32 workgroups 51.8 against 50.7 us as a launch)
    constexpr double t_batch = RPW >= 8 ? 25.0 : 12.5;
    int fin_blocks = 0;
    if (fin_ok && g_fin_inline > 1) fin_blocks = std::min(g_fin_inline, FIN_BLOCKS_MAX);      // (forced: tuning knob 9)
    else if (fin_ok) {
        for (int nb = 8; nb <= FIN_BLOCKS_MAX && fin_blocks == 0; nb += 8) {
            const int64_t main_waves = (int64_t)std::min<int64_t>(ceil_div(items, WAVES), SLOTS - nb) * WAVES;
            // the launch's time: work items at their latency-bound pace, or - big graphs, whose items are mostly
            // 128-edge tasks - the forward's bytes at the rate this kernel reaches (0.8 x B_fwd at 4.8 TB/s; products
            // size: 4.3 ms estimated, 4.1-4.2 measured; there 64 workgroups are the best count: 4.55 ms per forward
            // against 4.65 with the finalize as a launch, 4.64 with 48 or 128)
            const double b_fwd = (double)a.n_edges * (4.0 * a.C + 8.0) + (double)a.N * (8.0 * a.C + 8.0);
            const double t_main = std::max(5.8 * (double)ceil_div(items, main_waves), 0.8 * b_fwd / 4.8e6);
            const double t_fin = a.k < 0
                ? 6.0 + 4.0 * ceil_div(a.n_split, nb * WAVES) + 2.5 * ceil_div(ceil_div(max_split_deg, CHUNK), 16)
                : 6.0 + 17.0 * ceil_div(ceil_div(fin_big, 2), nb) + t_batch * ceil_div(fin_batches, nb * WAVES);
            if (t_fin > 0.75 * t_main) continue;
            const double t_sep = a.k < 0 ? 4.6 + 1.1 + 4.0 * a.n_split / (256.0 * 24)
                                         : 4.6 + 1.1 + 12.0 * fin_big / (256.0 * 3) + 12.5 * fin_batches / (256.0 * 24);
            const double loss = 5.8 * (double)ceil_div(items, (int64_t)SLOTS * WAVES) * nb / (double)(SLOTS - nb);
            if (loss + 1.0 < t_sep) fin_blocks = nb;
            break;                                            // (more workgroups only cost more)
        }
    }
    const bool fin_inline = fin_blocks > 0;
    // persistent grid: what the chip holds at the kernel's occupancy, or less (the finalize role's workgroups
    // are resident from the start: they come out of the same budget)
    const int grid_main = (int)std::min<int64_t>(ceil_div(items, WAVES), 256 * FWD_WAVES_PER_SIMD - fin_blocks);
    const int grid = grid_main + fin_blocks;
    a.main_blocks = grid_main;
    g_last_fin_blocks = fin_blocks;
    if (fin_inline) a.fin_nonce = next_fin_nonce();
    // head_part: one entry per workgroup of the head role, then one per split row.  (The role rides in the
    // mixed finalize launch - 512 threads then - where there is one, else in a launch of its own.)
    a.head_nmain = a.head_sel ? head_role_blocks(a.N, G, head_role_in_finalize(a) ? 512 : BLOCK) : 0;
    if (ev) SN_HIP(hipEventRecord(ev[0], st));
#define SNGNN_FWD_LAUNCH(FILTV, OTFV, FSV)                                                              \
    do {                                                                                              \
        if (fin_inline) k_agg_fwd<VEC, G, R, FILTV, OTFV, EPI, FSV, true><<<grid, BLOCK, 0, st>>>(a);  \
        else k_agg_fwd<VEC, G, R, FILTV, OTFV, EPI, FSV, false><<<grid, BLOCK, 0, st>>>(a);            \
    } while (0)
    for (int rep = 0; rep < reps && grid > 0; ++rep) {
        if (a.nrm == nullptr) {                                  // OTF: a.n holds the raw rows
            SNGNN_FWD_LAUNCH(false, true, false);
        } else if constexpr (VEC == 4 && G >= 16 && G * R <= 128) {     // the (G, R) that C in 36 .. 512 maps to
            if (a.filt && a.k >= 0 && a.filt_small) SNGNN_FWD_LAUNCH(true, false, true);
            else if (a.filt && a.k >= 0) SNGNN_FWD_LAUNCH(true, false, false);
            else SNGNN_FWD_LAUNCH(false, false, false);
        } else {
            SNGNN_FWD_LAUNCH(false, false, false);
        }
    }
#undef SNGNN_FWD_LAUNCH
    if (ev) SN_HIP(hipEventRecord(ev[1], st));
    for (int rep = 0; rep < reps && !fin_inline; ++rep)
        if (int rc = launch_split_finalize<VEC, G, R>(a, max_split_deg, st)) return rc;
    if (ev) {
        SN_HIP(hipEventRecord(ev[2], st));
        SN_HIP(hipEventRecord(ev[3], st));       // empty interval: the cost of an event pair
    }
    SN_HIP(hipGetLastError());
    if (a.head_sel)
        return launch_head_reduce(a.head_part, a.head_nmain + a.n_split, a.head_scale, a.head_scale_b,
                                  (a.head_flags & 1) ? 2 : 1, 4, a.head_out, st);
    return SNGNN_OK;
}

template <int VEC, int G, int R>
int launch_agg_fwd(const FwdArgs &a, int max_split_deg, hipEvent_t *ev, hipStream_t st)
{
    return launch_agg_fwd_impl<VEC, G, R, 0>(a, max_split_deg, ev, st);
}
template <int VEC, int G, int R>
int launch_agg_fwd_epi(const FwdArgs &a, int max_split_deg, hipEvent_t *ev, hipStream_t st)
{
    return launch_agg_fwd_impl<VEC, G, R, 1>(a, max_split_deg, ev, st);
}


// one translation unit per VEC instantiates these
int launch_agg_fwd_epi_v4(const RowCfg &cfg, const FwdArgs &a, int max_split_deg, hipEvent_t *ev,
                          hipStream_t st);      // (the store epilogue: 16-byte rows only)
int launch_agg_fwd_v1(const RowCfg &cfg, const FwdArgs &a, int max_split_deg, hipEvent_t *ev,
                      hipStream_t st);
int launch_agg_fwd_v2(const RowCfg &cfg, const FwdArgs &a, int max_split_deg, hipEvent_t *ev,
                      hipStream_t st);
int launch_agg_fwd_v4(const RowCfg &cfg, const FwdArgs &a, int max_split_deg, hipEvent_t *ev,
                      hipStream_t st);
int launch_normalize_v1(const RowCfg &cfg, const float *h, int64_t rows, int C, float *n, float *nrm, void *filt,
                        hipStream_t st);
int launch_normalize_v2(const RowCfg &cfg, const float *h, int64_t rows, int C, float *n, float *nrm, void *filt,
                        hipStream_t st);
int launch_normalize_v4(const RowCfg &cfg, const float *h, int64_t rows, int C, float *n, float *nrm, void *filt,
                        hipStream_t st);
int launch_filter_v4(const RowCfg &cfg, const float *n, int64_t rows, int C, void *filt, hipStream_t st);

}  // namespace sngnn
