// C-ABI entries of the fused aggregation forward (see agg_fwd_impl.h for the kernels).
#include <atomic>
#include "agg_fwd_impl.h"

using namespace sngnn;

// Optional in-library timing of the forward's launches (bench.py's roofline leg):
// HIP events recorded on the caller's stream around each launch.
//   ev[0] .. normalisation pass .. ev[1] .. main kernel .. ev[2] .. split-row finalize .. ev[3] .. (nothing) .. ev[4]
// The empty last interval measures what an event pair itself adds to an interval (a few
// microseconds on this stack); sngnn_profile_last_forward reports it so that the caller can
// take it off the three kernel figures.
static bool g_prof_on = false;
namespace sngnn { int g_prof_reps = 1; }
static hipEvent_t g_prof_ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};

extern "C" int sngnn_profile_enable(int on)
{
    if (on && !g_prof_ev[0])
        for (auto &e : g_prof_ev) SN_HIP(hipEventCreate(&e));
    g_prof_on = on != 0;
    g_prof_reps = on > 1 ? on : 1;
    return SNGNN_OK;
}

extern "C" int sngnn_profile_last_forward(float *norm_ms, float *main_ms, float *fin_ms, float *empty_ms)
{
    SN_REQUIRE(g_prof_ev[0] != nullptr && norm_ms && main_ms && fin_ms && empty_ms, SNGNN_EINVAL,
               "profiling is not enabled");
    SN_HIP(hipEventSynchronize(g_prof_ev[4]));
    SN_HIP(hipEventElapsedTime(norm_ms, g_prof_ev[0], g_prof_ev[1]));
    SN_HIP(hipEventElapsedTime(main_ms, g_prof_ev[1], g_prof_ev[2]));
    SN_HIP(hipEventElapsedTime(fin_ms, g_prof_ev[2], g_prof_ev[3]));
    SN_HIP(hipEventElapsedTime(empty_ms, g_prof_ev[3], g_prof_ev[4]));
    *norm_ms /= (float)g_prof_reps;
    *main_ms /= (float)g_prof_reps;
    *fin_ms /= (float)g_prof_reps;
    return SNGNN_OK;
}

static __global__ void k_fill_sel(int32_t *__restrict__ src, float *__restrict__ w, int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        src[i] = -1;
        w[i] = 0.f;
    }
}

static int normalize_dispatch(const RowCfg &cfg, const float *h, int64_t rows, int C, float *n, float *nrm,
                              void *filt, hipStream_t st)
{
    switch (cfg.vec) {
    case 1: return launch_normalize_v1(cfg, h, rows, C, n, nrm, filt, st);
    case 2: return launch_normalize_v2(cfg, h, rows, C, n, nrm, filt, st);
    default: return launch_normalize_v4(cfg, h, rows, C, n, nrm, filt, st);
    }
}

// wave rows shorter than this score their fp32 rows directly: two dependent round trips for a short row cost more than
// the second lines they save (uniform in-degree 24, top_k 4: 213 us unfiltered, 273 filtered, 223 with this cut)
constexpr int FILT_MIN_DEG = 32;
constexpr double FILT_PRUNABLE_SHARE = 0.62;
// fp16 filter: 0 = never, 1 = when it is expected to pay (default), 2 = whenever it applies (small rows too),
// 3 = the rows above the small class (wave rows, split-row tasks) whenever it applies, at any threshold
static int g_filter_mode = 1;
// measurement aids: the fp16 filter can be switched off (the selections are the same either
// way); sngnn_tuning_set(0, mask) runs only some row classes of the main kernel (bit 0 split-row
// tasks, 1 wave rows, 2 small rows; results are then incomplete - timing only)
static int g_role_mask = 7;
// how sngnn_agg_forward scores: 0 = auto (on the fly from h when nothing is selected - top_k < 0,
// SNConv: 68 -> 58 us per forward at arxiv size, no normalisation pass, no table; the unit-row
// table otherwise: for ranking rows one cheap pass over table rows beats a fast pass plus the
// exact re-scoring of the candidates, 72.2 against 73.5 us at top_k 16 / thr 0), 1 = table
// always, 2 = on the fly always (same selections, bit for bit)
static int g_table_mode = 0;
// (sngnn_tuning_set(8, d): wave rows with fewer than d in-edges skip the fp16 filter; -1 = the library's rule)
static int g_filt_min_deg = -1;
int sngnn::g_fin_inline = 1;
int sngnn::g_last_fin_blocks = 0;
extern "C" int sngnn_last_forward_finalize_workgroups(void) { return sngnn::g_last_fin_blocks; }
// a value no earlier call of this process has used (the finalize role: the tasks' done words)
unsigned long long sngnn::next_fin_nonce()
{
    static std::atomic<unsigned long long> counter{0};
    return 0xF1A0000000000000ull + counter.fetch_add(1) + 1;
}
bool sngnn::fwd_scores_on_the_fly_forced() { return g_table_mode == 2; }
extern "C" int sngnn_tuning_set(int which, int value)
{
    SN_REQUIRE(which == 0 || (which >= 2 && which <= 9), SNGNN_EINVAL, "unknown tuning knob");
    if (which == 9) { sngnn::g_fin_inline = value < 0 ? 0 : value; return SNGNN_OK; }
    if (which == 8) { g_filt_min_deg = value; return SNGNN_OK; }
    if (which == 7) return sngnn::set_cosine_split(value);
    if (which == 6) return sngnn::set_knn_route(value);
    if (which == 5) return sngnn::set_lin_mode(value);
    if (which == 3) return sngnn::set_bwd_mode(value);
    if (which == 4) return sngnn::set_bwd_roles(value);
    if (which == 0) g_role_mask = value & 7;
    else g_table_mode = value;
    return SNGNN_OK;
}

extern "C" int sngnn_filter_enable(int mode)
{
    SN_REQUIRE(mode >= 0 && mode <= 3, SNGNN_EINVAL,
               "filter mode must be 0 (off), 1 (auto), 2 (always) or 3 (wave rows and tasks always)");
    g_filter_mode = mode;
    return SNGNN_OK;
}

extern "C" int64_t sngnn_filter_row_bytes(int C) { return filter_row_bytes(C); }

extern "C" int sngnn_normalize_rows_filter(const float *h, int64_t rows, int C, float *n, float *nrm, void *filt,
                                           void *stream)
{
    SN_REQUIRE(rows >= 0, SNGNN_EINVAL, "negative row count");
    SN_REQUIRE(rows == 0 || (h && n && nrm), SNGNN_EINVAL, "h/n/nrm is NULL");
    RowCfg cfg;
    SN_REQUIRE(row_cfg(C, cfg), SNGNN_EINVAL, "C must be in [1, " + std::to_string(SNGNN_MAX_CHANNELS) + "]");
    SN_REQUIRE(((uintptr_t)h % (cfg.vec * 4)) == 0 && ((uintptr_t)n % (cfg.vec * 4)) == 0, SNGNN_EINVAL,
               "h/n must be aligned to the row vector width");
    SN_REQUIRE(filt == nullptr || filter_row_bytes(C) > 0, SNGNN_EINVAL,
               "no filter rows for this C (sngnn_filter_row_bytes(C) == 0)");
    SN_REQUIRE(((uintptr_t)filt % 16) == 0, SNGNN_EINVAL, "filt must be 16-byte aligned");
    return normalize_dispatch(cfg, h, rows, C, n, nrm, filt, (hipStream_t)stream);
}

extern "C" int sngnn_normalize_rows(const float *h, int64_t rows, int C, float *n, float *nrm, void *stream)
{
    return sngnn_normalize_rows_filter(h, rows, C, n, nrm, nullptr, stream);
}

// approximate cosine of node pairs straight from filter rows (the instruction sequence of the
// forward's filter pass) - lets a test check the error bound FILT_EPS on the hardware
template <int GF>
static __global__ __launch_bounds__(BLOCK) void k_filter_pair_scores(const uint4 *__restrict__ filt,
                                                                     const int64_t *__restrict__ pa,
                                                                     const int64_t *__restrict__ pb, int64_t n_pairs,
                                                                     float *__restrict__ out)
{
    constexpr int NGF = 64 / GF;
    const int lane = lane_id();
    const int gid = lane / GF, lf = lane % GF;
    const int64_t w = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    const int64_t p = w * NGF + gid;
    const int64_t q = p < n_pairs ? p : n_pairs - 1;
    const uint4 fa = filt[(size_t)pa[q] * GF + lf], fb = filt[(size_t)pb[q] * GF + lf];
    const float s = group_sum<GF>(fdot8(fa, fb)) * FILT_UNSCALE;
    if (p < n_pairs && lf == 0) out[p] = s;
}

extern "C" int sngnn_filter_pair_scores(const void *filt, int C, const int64_t *pair_a, const int64_t *pair_b,
                                        int64_t n_pairs, float *out, void *stream)
{
    SN_REQUIRE(filter_row_bytes(C) > 0, SNGNN_EINVAL, "no filter rows for this C");
    SN_REQUIRE(n_pairs >= 0, SNGNN_EINVAL, "negative pair count");
    if (n_pairs == 0) return SNGNN_OK;
    SN_REQUIRE(filt && pair_a && pair_b && out, SNGNN_EINVAL, "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    const int gf = (int)(filter_row_bytes(C) / 16);
    const int grid = ceil_div(n_pairs, (64 / gf) * WAVES);
    const uint4 *f = (const uint4 *)filt;
    switch (gf) {
    case 8: k_filter_pair_scores<8><<<grid, BLOCK, 0, st>>>(f, pair_a, pair_b, n_pairs, out); break;
    case 16: k_filter_pair_scores<16><<<grid, BLOCK, 0, st>>>(f, pair_a, pair_b, n_pairs, out); break;
    case 32: k_filter_pair_scores<32><<<grid, BLOCK, 0, st>>>(f, pair_a, pair_b, n_pairs, out); break;
    default: k_filter_pair_scores<64><<<grid, BLOCK, 0, st>>>(f, pair_a, pair_b, n_pairs, out); break;
    }
    SN_HIP(hipGetLastError());
    return SNGNN_OK;
}

// everything after the unit rows exist
static int forward_normalized(const sngnn_graph_t *g, const RowCfg &cfg, const float *n, const float *nrm,
                              const void *filt, int C, int top_k, float thr, float *out, float *wsel, float *inv_norm,
                              int32_t *sel_src, float *sel_w, void *scratch, hipEvent_t *ev, hipStream_t st,
                              const uint8_t *row_flag = nullptr, int row_want = 0, const sngnn_epilogue_t *epi = nullptr)
{
    if (top_k > (1 << 20)) top_k = 1 << 20;     // more than any row can use

    if (sel_src && top_k > 0 && row_flag == nullptr) {
        // (a kernel, not hipMemsetAsync: inside a captured HIP graph a memset node was seen to
        // race with the kernel nodes behind it on this stack - see agg_bwd_impl.h)
        const int64_t nw = g->N * (int64_t)top_k;
        k_fill_sel<<<(int)std::min<int64_t>((nw + 255) / 256, 2048), 256, 0, st>>>(sel_src, sel_w, nw);
    }

    FwdArgs a;
    a.n = n; a.nrm = nrm; a.C = C; a.N = (int)g->N; a.row_off = (int)g->row_off;
    a.filt = nrm ? (const uint4 *)filt : nullptr;
    // (sngnn_filter_enable(2) forces it at any threshold: the tests' and the fuzz runs' way in)
    // (top_k >= 4: at top_k 1 the small-row form gains 1.5 us where it prunes - 39.3 -> 37.6-38.8 us at thr
    // 0.99 - and costs 13.6 us per call where it does not: the reference scripts' own knobs)
    // (a pruning threshold: every ranking row takes the filter, as before; by the graph's prunable share: long rows only)
    a.filt_min_deg = g_filt_min_deg >= 0 ? g_filt_min_deg : (thr >= 0.25f ? 0 : FILT_MIN_DEG);
    a.filt_small = (a.filt != nullptr && top_k >= 0 && g_filter_mode != 0 &&
                    ((thr >= 0.25f && top_k >= 4) || g_filter_mode == 2)) ? 1 : 0;
    // OTF (nrm == NULL, n = raw rows): bound on |fast cosine - reference-order cosine|.  Either
    // value is within (2 C + 8) u of the real cosine (u = 2^-24: C products and sums of the dot,
    // C / 2 + 3 for each norm, the scalings), so they differ by less than (4 C + 16) u; twice
    // the margin on the constant term.
    a.delta = (float)(4 * C + 32) * 5.9604644775390625e-8f;
    a.role_mask = g_role_mask;
    a.row_flag = row_flag; a.row_want = row_want;
    a.epi_flags = 0; a.epi_bias = nullptr; a.epi_keep = nullptr; a.epi_seed = nullptr; a.epi_p = 0.f; a.epi_scale = 1.0f;
    a.head_y = nullptr; a.head_sel = nullptr; a.head_part = nullptr; a.head_flags = 0; a.head_nmain = 0;
    a.head_scale = a.head_scale_b = 0.f; a.head_out = nullptr;
    a.kbits = nullptr; a.kb_wbase = (int)g->kb_wbase; a.kb_tbase = (int)g->kb_tbase;
    if (epi && epi->kept_bits) {
        SN_REQUIRE(sngnn::kept_bits_path(g, top_k) && row_flag == nullptr && nrm != nullptr, SNGNN_EINVAL,
                   "no kept-bit path for this graph / top_k (sngnn_agg_kept_bits_supported)");
        a.kbits = (unsigned *)epi->kept_bits;
    }
    const bool head = epi != nullptr && epi->head_y != nullptr;
    if (head) {
        SN_REQUIRE(cfg.vec == 4 && cfg.r == 1 && cfg.g <= 16, SNGNN_EINVAL,
                   "the head epilogue needs C % 4 == 0 and C <= 64 (sngnn_agg_head_supported)");
        SN_REQUIRE(epi->head_sel && epi->head_metrics && epi->head_workspace, SNGNN_EINVAL,
                   "head_sel / head_metrics / head_workspace is NULL");
        SN_REQUIRE(epi->head_sets == 1 || epi->head_sets == 2, SNGNN_EINVAL, "head_sets must be 1 or 2");
        SN_REQUIRE((epi->head_out_mode == 1 || epi->head_out_mode == 2) && (epi->head_out_mode != 2 || epi->head_sets == 1),
                   SNGNN_EINVAL, "head_out_mode: 1 logits, 2 gradient (one split only)");
        SN_REQUIRE(out != nullptr, SNGNN_EINVAL, "out is NULL");
        SN_REQUIRE(!epi->relu && epi->keep == nullptr && epi->seed == nullptr, SNGNN_EINVAL,
                   "the head follows the LAST layer: no relu / dropout with it");
        SN_REQUIRE(row_flag == nullptr && sel_src == nullptr, SNGNN_EINVAL,
                   "the head epilogue covers all rows of a call and emits no selection lists");
        a.head_y = epi->head_y; a.head_sel = epi->head_sel;
        a.head_part = (float *)epi->head_workspace; a.head_out = epi->head_metrics;
        a.head_flags = (epi->head_sets == 2 ? 1 : 0) | (epi->head_out_mode == 2 ? 2 : 0);
        a.head_scale = 1.0f / (float)(epi->head_n_a > 0 ? epi->head_n_a : 1);
        a.head_scale_b = 1.0f / (float)(epi->head_n_b > 0 ? epi->head_n_b : 1);
    }
    if (epi) {
        a.epi_bias = epi->bias; a.epi_keep = epi->keep;
        a.epi_seed = (const unsigned long long *)epi->seed; a.epi_p = epi->p;
        const bool drops = epi->keep != nullptr || epi->seed != nullptr;
        a.epi_scale = drops ? epi->keep_scale : 1.0f;
        a.epi_flags = (epi->relu ? 1 : 0) | (epi->bias ? 2 : 0) | (epi->keep ? 4 : (epi->seed ? 8 : 0));
    }
    // row order: calls that stream the small rows (deg <= SMALL_T <= top_k, or no top_k) take the
    // bucket order, calls that rank inside them the exact degree order (graph.hip 5b)
    const bool stream_small = top_k < 0 || top_k >= SMALL_T;
    a.rowptr = g->rowptr; a.col = g->col;
    a.col_s = stream_small ? g->col_s_b : g->col_s;
    a.rperm = stream_small ? g->rperm_b : g->rperm;
    a.rdesc = stream_small ? g->rdesc_b : g->rdesc;
    a.k = top_k < 0 ? -1 : top_k; a.thr = thr;
    a.out = out; a.wsel = wsel; a.inv_norm = inv_norm;
    a.sel_src = top_k > 0 ? sel_src : nullptr; a.sel_w = top_k > 0 ? sel_w : nullptr;
    a.n_split = g->n_split;
    a.n_med_end = g->rows_gt(SMALL_T);
    a.n_tasks = g->n_tasks;
    a.task_slot = g->task_slot; a.task_chunk = g->task_chunk;
    a.task_order = g->task_order;
    a.split_soff = g->split_soff; a.split_task0 = g->split_task0;
    a.scores = (float *)scratch;
    a.partial = a.scores ? a.scores + (g->split_edges + 3) / 4 * 4 : nullptr;   // 16-B aligned rows
    a.cand_key = a.partial ? (unsigned long long *)(a.partial + ((size_t)g->n_tasks * C + 3) / 4 * 4)
                           : nullptr;
    a.cand_src = a.cand_key ? (int32_t *)(a.cand_key + (size_t)g->n_tasks * CAND_MAX_K) : nullptr;
    a.fin_done = a.cand_src ? (unsigned long long *)(a.cand_src + (size_t)g->n_tasks * CAND_MAX_K) : nullptr;   // 8-byte aligned
    a.fin_nonce = 0ull; a.main_blocks = 0; a.n_edges = (long long)g->Ep;
    a.k_magic = top_k >= 2 ? (unsigned)(0xFFFFFFFFu / (unsigned)top_k + 1u) : 0u;
    const int max_split = g->n_split ? g->rdeg[0] : 0;
    a.use_cand = fwd_use_candidates(a.k, C, max_split) ? 1 : 0;
    SN_REQUIRE(a.kbits == nullptr || a.use_cand, SNGNN_EINVAL,
               "no kept-bit path: this graph's biggest row takes the scratch-score finalize");
    // split rows whose (tasks * top_k) candidates exceed one 128-key wave selection
    // (no selection: split rows with more than 16 partial rows to add)
    a.n_split_gt_wave = top_k > 0 ? g->rows_gt((int64_t)(128 / std::min(top_k, 128)) * CHUNK)
                                  : (top_k < 0 ? g->rows_gt((int64_t)16 * CHUNK) : 0);
    a.lowbits = 1;
    while ((1ll << a.lowbits) < g->max_in_deg && a.lowbits < 31) ++a.lowbits;
    if (head) {
        SN_REQUIRE(g->n_split == 0 || a.use_cand, SNGNN_EINVAL,
                   "no head epilogue: this graph's biggest row takes the scratch-score finalize (sngnn_agg_head_supported)");
        return launch_agg_fwd_v4(cfg, a, max_split, ev, st);      // (the plain kernels: the head rides in the second launch)
    }
    if (a.epi_flags != 0) {
        SN_REQUIRE(cfg.vec == 4, SNGNN_EINVAL, "the store epilogue needs C % 4 == 0 (16-byte rows)");
        return launch_agg_fwd_epi_v4(cfg, a, max_split, ev, st);
    }
    switch (cfg.vec) {
    case 1: return launch_agg_fwd_v1(cfg, a, max_split, ev, st);
    case 2: return launch_agg_fwd_v2(cfg, a, max_split, ev, st);
    default: return launch_agg_fwd_v4(cfg, a, max_split, ev, st);
    }
}

static int check_forward_args(const sngnn_graph_t *g, const float *rows, int C, int top_k, const float *out,
                              const int32_t *sel_src, const float *sel_w, RowCfg &cfg)
{
    SN_REQUIRE(g != nullptr, SNGNN_EINVAL, "graph is NULL");
    SN_REQUIRE(g->N == 0 || (rows != nullptr && out != nullptr), SNGNN_EINVAL, "h/out is NULL");
    SN_REQUIRE(row_cfg(C, cfg), SNGNN_EINVAL,
               "C must be in [1, " + std::to_string(SNGNN_MAX_CHANNELS) + "]");
    SN_REQUIRE(((uintptr_t)rows % (cfg.vec * 4)) == 0 && ((uintptr_t)out % (cfg.vec * 4)) == 0,
               SNGNN_EINVAL, "h/out must be aligned to the row vector width");
    SN_REQUIRE((sel_src == nullptr) == (sel_w == nullptr), SNGNN_EINVAL,
               "sel_src and sel_w go together");
    SN_REQUIRE(sel_src == nullptr || top_k >= 0, SNGNN_EINVAL, "sel_src needs top_k >= 0");
    return SNGNN_OK;
}

// the filter region of the workspace, and whether a call uses it: rows that rank (in-degree >
// top_k) above the small class must exist, otherwise nothing would read the table
static void *ws_filter(void *workspace, int64_t Ntot, int C)
{
    return (char *)workspace + (Ntot * (int64_t)C * 4 + 255) / 256 * 256 + (Ntot * 4 + 255) / 256 * 256;
}
static bool use_filter(const sngnn_graph_t *g, int C, int top_k, float thr)
{
    if (g_filter_mode == 0 || top_k < 0 || filter_row_bytes(C) == 0) return false;
    // a pruning threshold: the small rows read the table too (FwdArgs::filt_small) - any graph
    if ((thr >= 0.25f && top_k >= 4) || g_filter_mode == 2) return true;
    if (thr >= 0.25f || g_filter_mode == 3) return g->rows_gt(std::max(top_k, SMALL_T)) != 0;
    // No threshold to prune with, but top_k itself prunes: a row of `deg` in-edges keeps at most top_k of them, yet
    // the unfiltered pass fetches the fp32 unit row (two lines for C in 36 .. 60) of every in-edge.  Measured in round
    // 5 (tools/filter_gate_sweep.py, profiles/r05_filter_gate.txt): with the filter on the rows above the small
    // class at thr 0 the forward takes 0.73-0.92 of its time on products' degree law at every size tried (unit-row
    // table 23 MiB .. 470 MB: 5.54 -> 4.16 ms main kernel at full size), 0.79-0.84 at uniform in-degree 50, 0.89-0.93 on
    // arxiv's law at 4x its edges - and 1.01-1.03 at arxiv's own size (the extra table costs the normalisation pass
    // 3 us, the main kernel gains 1).  The rule is the graph's PRUNABLE share: edges beyond top_k in rows of at
    // least FILT_MIN_DEG in-edges (shorter rows skip the filter: FwdArgs::filt_min_deg), over E', at top_k 16 / 4:
    // products 0.72 / 0.77, uniform 50 0.68 / 0.92, arxiv x4 0.64 / 0.69, arxiv x2 0.58 / 0.62 (0.98 / 0.97 of its time)
    // | arxiv 0.53 / 0.57.
    if (g->Ep > 0 && (double)g->prunable_edges(top_k, FILT_MIN_DEG) >= FILT_PRUNABLE_SHARE * (double)g->Ep) return true;
    // No threshold to prune with: the filter could still skip, in rows much longer than top_k, the edges
    // far below the k-th - round 3's rule was "on for top_k <= 8" (-7 us per forward at top_k 1 then).
    // Re-measured in round 4 (same box, arxiv size, C = 40, thr 0; filter on / off): top_k 1: 70.7 / 69.5
    // us per call, top_k 4: 73.6 / 70.9, top_k 8: 75.0 / 72.6 - the main kernel gains ~1 us, the extra table
    // costs the normalisation pass 3; and a model's later layers (nearly parallel rows: every edge a
    // candidate) lose more: 2- / 3-layer hidden-64 top_k-1 epochs 0.711 / 1.099 ms with, 0.703 / 1.052
    // without.  Hence: a selective threshold only.
    return false;
}

static int agg_forward_impl(const sngnn_graph_t *g, const float *h, int C, int top_k,
                            float thr, float *out, float *wsel, float *inv_norm,
                            int32_t *sel_src, float *sel_w, void *workspace, void *stream, const sngnn_epilogue_t *epi)
{
    RowCfg cfg;
    if (int rc = check_forward_args(g, h, C, top_k, out, sel_src, sel_w, cfg)) return rc;
    if (g->N == 0 && epi != nullptr && epi->head_y != nullptr && epi->head_metrics != nullptr)      // no row: zero metrics
        return launch_head_reduce(nullptr, 0, 0.f, 0.f, epi->head_sets == 2 ? 2 : 1, 4, epi->head_metrics, (hipStream_t)stream);
    if (g->N == 0) return SNGNN_OK;
    SN_REQUIRE(workspace != nullptr, SNGNN_EINVAL, "workspace is NULL");
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t *ev0 = g_prof_on ? g_prof_ev : nullptr;
    void *scratch0 = (char *)workspace + fwd_table_bytes(g->Ntot, C);
    if (g_table_mode == 2 || (g_table_mode == 0 && top_k < 0)) {
        // on the fly: no normalisation pass, no table - the kernels gather h itself (FwdArgs)
        if (ev0) {
            SN_HIP(hipEventRecord(ev0[0], st));
        }
        return forward_normalized(g, cfg, h, nullptr, nullptr, C, top_k, thr, out, wsel, inv_norm, sel_src, sel_w,
                                  scratch0, ev0 ? ev0 + 1 : nullptr, st, nullptr, 0, epi);
    }
    // table mode (sngnn_tuning_set(2, 1)): the normalisation pass first.
    // workspace: unit rows [Ntot, C] | norms [Ntot] | fp16 filter rows | scratch of the split rows
    float *n = (float *)workspace;
    float *nrm = (float *)((char *)workspace + (g->Ntot * (int64_t)C * 4 + 255) / 256 * 256);
    void *scratch = (char *)workspace + fwd_table_bytes(g->Ntot, C);
    void *filt = (use_filter(g, C, top_k, thr) && !(epi && epi->no_filter)) ? ws_filter(workspace, g->Ntot, C) : nullptr;
    hipEvent_t *ev = g_prof_on ? g_prof_ev : nullptr;
    if (ev) SN_HIP(hipEventRecord(ev[0], st));
    for (int rep = 0; rep < (ev ? g_prof_reps : 1); ++rep)
        if (int rc = normalize_dispatch(cfg, h, g->Ntot, C, n, nrm, filt, st)) return rc;
    return forward_normalized(g, cfg, n, nrm, filt, C, top_k, thr, out, wsel, inv_norm, sel_src, sel_w, scratch,
                              ev ? ev + 1 : nullptr, st, nullptr, 0, epi);
}

extern "C" int sngnn_agg_forward(const sngnn_graph_t *g, const float *h, int C, int top_k,
                                 float thr, float *out, float *wsel, float *inv_norm,
                                 int32_t *sel_src, float *sel_w, void *workspace, void *stream)
{
    return agg_forward_impl(g, h, C, top_k, thr, out, wsel, inv_norm, sel_src, sel_w, workspace, stream, nullptr);
}

extern "C" int64_t sngnn_agg_head_workspace_bytes(const sngnn_graph_t *g)
{
    // one 16-byte entry per wave of the head role and per split row
    return g ? ((int64_t)256 * 32 + g->n_split) * 16 + 256 : 0;
}

extern "C" int sngnn_agg_head_supported(const sngnn_graph_t *g, int C, int top_k)
{
    RowCfg cfg;
    if (g == nullptr || !row_cfg(C, cfg) || cfg.vec != 4 || cfg.r != 1 || cfg.g > 16) return 0;
    const int max_split = g->n_split ? g->rdeg[0] : 0;
    return (g->n_split == 0 || fwd_use_candidates(top_k < 0 ? -1 : top_k, C, max_split)) ? 1 : 0;
}

static int check_epilogue(const sngnn_epilogue_t *epi)
{
    SN_REQUIRE(epi != nullptr, SNGNN_EINVAL, "epilogue is NULL");
    SN_REQUIRE((epi->keep == nullptr && epi->seed == nullptr) || (epi->keep_scale > 0.f && epi->keep_scale < 1e30f),
               SNGNN_EINVAL, "keep_scale must be a positive finite number (1 / (1 - p))");
    SN_REQUIRE(epi->seed == nullptr || (epi->p >= 0.f && epi->p < 1.f), SNGNN_EINVAL, "p must be in [0, 1)");
    return SNGNN_OK;
}

extern "C" int sngnn_agg_forward_epilogue(const sngnn_graph_t *g, const float *h, int C, int top_k, float thr,
                                          const sngnn_epilogue_t *epi, float *out, float *wsel, float *inv_norm,
                                          void *workspace, void *stream)
{
    if (int rc = check_epilogue(epi)) return rc;
    return agg_forward_impl(g, h, C, top_k, thr, out, wsel, inv_norm, nullptr, nullptr, workspace, stream, epi);
}

static int agg_forward_prepared_impl(const sngnn_graph_t *g, const float *n, const float *nrm,
                                     const void *filt, int C, int top_k, float thr, float *out, float *wsel,
                                     float *inv_norm, int32_t *sel_src, float *sel_w, void *workspace,
                                     void *stream, const sngnn_epilogue_t *epi)
{
    RowCfg cfg;
    if (int rc = check_forward_args(g, n, C, top_k, out, sel_src, sel_w, cfg)) return rc;
    if (g->N == 0 && epi != nullptr && epi->head_y != nullptr && epi->head_metrics != nullptr)
        return launch_head_reduce(nullptr, 0, 0.f, 0.f, epi->head_sets == 2 ? 2 : 1, 4, epi->head_metrics, (hipStream_t)stream);
    if (g->N == 0) return SNGNN_OK;
    SN_REQUIRE(nrm != nullptr, SNGNN_EINVAL, "nrm is NULL");
    SN_REQUIRE(workspace != nullptr || (g->n_tasks == 0 && (filt != nullptr || !use_filter(g, C, top_k, thr))),
               SNGNN_EINVAL, "workspace is NULL");
    SN_REQUIRE(filt == nullptr || filter_row_bytes(C) > 0, SNGNN_EINVAL,
               "no filter rows for this C (sngnn_filter_row_bytes(C) == 0)");
    SN_REQUIRE(((uintptr_t)filt % 16) == 0, SNGNN_EINVAL, "filt must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    // same workspace layout as sngnn_agg_forward; the unit-row region stays unused
    void *scratch = workspace ? (char *)workspace + fwd_table_bytes(g->Ntot, C) : nullptr;
    hipEvent_t *ev = g_prof_on ? g_prof_ev : nullptr;
    if (ev) { SN_HIP(hipEventRecord(ev[0], st)); }
    const void *f = nullptr;
    if (use_filter(g, C, top_k, thr) && !(epi && epi->no_filter)) {
        f = filt;
        if (!f) {       // the caller holds no filter rows: one more pass over its unit rows
            void *wf = ws_filter(workspace, g->Ntot, C);
            if (int rc = launch_filter_v4(cfg, n, g->Ntot, C, wf, st)) return rc;
            f = wf;
        }
    }
    return forward_normalized(g, cfg, n, nrm, f, C, top_k, thr, out, wsel, inv_norm, sel_src, sel_w, scratch,
                              ev ? ev + 1 : nullptr, st, nullptr, 0, epi);
}

extern "C" int sngnn_agg_forward_prepared(const sngnn_graph_t *g, const float *n, const float *nrm,
                                          const void *filt, int C, int top_k, float thr, float *out, float *wsel,
                                          float *inv_norm, int32_t *sel_src, float *sel_w, void *workspace,
                                          void *stream)
{
    return agg_forward_prepared_impl(g, n, nrm, filt, C, top_k, thr, out, wsel, inv_norm, sel_src, sel_w, workspace,
                                     stream, nullptr);
}

extern "C" int sngnn_agg_forward_prepared_epilogue(const sngnn_graph_t *g, const float *n, const float *nrm,
                                                   const void *filt, int C, int top_k, float thr,
                                                   const sngnn_epilogue_t *epi, float *out, float *wsel,
                                                   float *inv_norm, void *workspace, void *stream)
{
    if (int rc = check_epilogue(epi)) return rc;
    return agg_forward_prepared_impl(g, n, nrm, filt, C, top_k, thr, out, wsel, inv_norm, nullptr, nullptr, workspace,
                                     stream, epi);
}

extern "C" int sngnn_agg_forward_rows(const sngnn_graph_t *g, const float *n, const float *nrm, const void *filt,
                                      int C, int top_k, float thr, const uint8_t *row_flag, int row_want,
                                      float *out, float *wsel, float *inv_norm, void *workspace, void *stream)
{
    RowCfg cfg;
    if (int rc = check_forward_args(g, n, C, top_k, out, nullptr, nullptr, cfg)) return rc;
    if (g->N == 0) return SNGNN_OK;
    // (nrm == NULL: n holds the RAW rows h and the call scores on the fly, like sngnn_agg_forward)
    SN_REQUIRE(row_flag != nullptr && workspace != nullptr, SNGNN_EINVAL, "row_flag/workspace is NULL");
    SN_REQUIRE(nrm != nullptr || filt == nullptr, SNGNN_EINVAL, "filter rows go with unit rows + norms");
    SN_REQUIRE(filt == nullptr || filter_row_bytes(C) > 0, SNGNN_EINVAL,
               "no filter rows for this C (sngnn_filter_row_bytes(C) == 0)");
    SN_REQUIRE(((uintptr_t)filt % 16) == 0, SNGNN_EINVAL, "filt must be 16-byte aligned");
    void *scratch = (char *)workspace + fwd_table_bytes(g->Ntot, C);
    // (the caller decides whether filter rows exist - sngnn_filter_wanted - and passes them or NULL)
    return forward_normalized(g, cfg, n, nrm, filt, C, top_k, thr, out, wsel, inv_norm, nullptr, nullptr, scratch,
                              nullptr, (hipStream_t)stream, row_flag, row_want);
}

// (the forward side of the question: split rows must take the candidate finalize, whose winners set their bits)
extern "C" int sngnn_agg_kept_bits_supported(const sngnn_graph_t *g, int C, int top_k)
{
    if (!sngnn::kept_bits_path(g, top_k)) return 0;
    return fwd_use_candidates(top_k, C, g->n_split ? g->rdeg[0] : 0) ? 1 : 0;
}

extern "C" int sngnn_filter_wanted(const sngnn_graph_t *g, int C, int top_k, float thr)
{
    return g != nullptr && use_filter(g, C, top_k, thr) ? 1 : 0;
}

extern "C" int sngnn_agg_forward_normalized(const sngnn_graph_t *g, const float *n, const float *nrm, int C,
                                            int top_k, float thr, float *out, float *wsel, float *inv_norm,
                                            int32_t *sel_src, float *sel_w, void *workspace, void *stream)
{
    return sngnn_agg_forward_prepared(g, n, nrm, nullptr, C, top_k, thr, out, wsel, inv_norm, sel_src, sel_w,
                                      workspace, stream);
}
