// C-ABI entries of the SNGNN++ adjacency-linear branch (kernels: adj_linear_impl.h).
#include "adj_linear_impl.h"

using namespace sngnn;

namespace sngnn {
static int dispatch_adj(const RowCfg &cfg, const AdjArgs &a, hipStream_t st)
{
    switch (cfg.vec) {
    case 1: SNGNN_DISPATCH_GR(launch_adj, 1, cfg, a, st)
    case 2: SNGNN_DISPATCH_GR(launch_adj, 2, cfg, a, st)
    default: SNGNN_DISPATCH_GR(launch_adj, 4, cfg, a, st)
    }
}
}  // namespace sngnn

extern "C" int sngnn_adj_linear_forward(const sngnn_graph_t *g, const float *wt, const float *bias,
                                        int C, float *out0, void *workspace, void *stream)
{
    SN_REQUIRE(g != nullptr, SNGNN_EINVAL, "graph is NULL");
    SN_REQUIRE(g->N == g->Ntot, SNGNN_EINVAL, "the adjacency branch needs an unpartitioned graph");
    if (g->N == 0) return SNGNN_OK;
    SN_REQUIRE(wt && out0, SNGNN_EINVAL, "NULL argument");
    SN_REQUIRE(g->n_stasks == 0 || workspace, SNGNN_EINVAL, "workspace is NULL");
    RowCfg cfg;
    SN_REQUIRE(row_cfg(C, cfg), SNGNN_EINVAL,
               "C must be in [1, " + std::to_string(SNGNN_MAX_CHANNELS) + "]");
    AdjArgs a;
    a.table = wt; a.w = nullptr; a.bias = bias; a.out = out0; a.partial = (float *)workspace;
    a.C = C; a.N = (int)g->N;
    a.ptr = g->cscptr; a.idx = g->csc_dst; a.perm = g->sperm;
    a.seg_shift = (int)g->src_min;        // out row i <- CSC row i + src_min (models.py:125)
    a.idx_shift = 0;
    a.n_split = g->n_ssplit; a.n_med_end = g->srcs_gt(SMALL_T); a.n_tasks = g->n_stasks;
    a.task_slot = g->stask_slot; a.task_chunk = g->stask_chunk; a.split_task0 = g->ssplit_task0;
    a.nbA = a.nbB = 0;
    return dispatch_adj(cfg, a, (hipStream_t)stream);
}

extern "C" int sngnn_adj_linear_backward(const sngnn_graph_t *g, const float *g0, int C, float *dwt,
                                         void *workspace, void *stream)
{
    SN_REQUIRE(g != nullptr, SNGNN_EINVAL, "graph is NULL");
    SN_REQUIRE(g->N == g->Ntot, SNGNN_EINVAL, "the adjacency branch needs an unpartitioned graph");
    if (g->N == 0) return SNGNN_OK;
    SN_REQUIRE(g0 && dwt, SNGNN_EINVAL, "NULL argument");
    SN_REQUIRE(g->n_tasks == 0 || workspace, SNGNN_EINVAL, "workspace is NULL");
    RowCfg cfg;
    SN_REQUIRE(row_cfg(C, cfg), SNGNN_EINVAL,
               "C must be in [1, " + std::to_string(SNGNN_MAX_CHANNELS) + "]");
    AdjArgs a;
    a.table = g0; a.w = nullptr; a.bias = nullptr; a.out = dwt; a.partial = (float *)workspace;
    a.C = C; a.N = (int)g->N;
    a.ptr = g->rowptr; a.idx = g->col; a.perm = g->rperm;
    a.seg_shift = 0;
    a.idx_shift = -(int)g->src_min;       // g0 row of source s is s - src_min
    a.n_split = g->n_split; a.n_med_end = g->rows_gt(SMALL_T); a.n_tasks = g->n_tasks;
    a.task_slot = g->task_slot; a.task_chunk = g->task_chunk; a.split_task0 = g->split_task0;
    a.nbA = a.nbB = 0;
    return dispatch_adj(cfg, a, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// The same two gather-sums on ANY graph, partitions included (multi-GPU SNGNN++:
// the adjacency branch of a rank runs on the partition of the FLIPPED edge list,
// whose "targets" are the rank's own source nodes).
// ---------------------------------------------------------------------------
static int gather_sum_impl(const sngnn_graph_t *g, const float *table, const float *w, const float *bias, int C,
                           float *out, void *workspace, void *stream);
extern "C" int sngnn_gather_sum_rows(const sngnn_graph_t *g, const float *table, const float *bias,
                                     int C, float *out, void *workspace, void *stream)
{
    return gather_sum_impl(g, table, nullptr, bias, C, out, workspace, stream);
}
extern "C" int sngnn_weighted_gather_sum_rows(const sngnn_graph_t *g, const float *table, const float *w_csr, int C,
                                              float *out, void *workspace, void *stream)
{
    SN_REQUIRE(g == nullptr || g->Ep == 0 || w_csr != nullptr, SNGNN_EINVAL, "w_csr is NULL");
    return gather_sum_impl(g, table, w_csr, nullptr, C, out, workspace, stream);
}
static int gather_sum_impl(const sngnn_graph_t *g, const float *table, const float *w, const float *bias, int C,
                           float *out, void *workspace, void *stream)
{
    SN_REQUIRE(g != nullptr, SNGNN_EINVAL, "graph is NULL");
    if (g->N == 0) return SNGNN_OK;
    SN_REQUIRE(table && out, SNGNN_EINVAL, "NULL argument");
    SN_REQUIRE(g->n_tasks == 0 || workspace, SNGNN_EINVAL, "workspace is NULL");
    RowCfg cfg;
    SN_REQUIRE(row_cfg(C, cfg), SNGNN_EINVAL,
               "C must be in [1, " + std::to_string(SNGNN_MAX_CHANNELS) + "]");
    AdjArgs a;
    a.table = table; a.w = w; a.bias = bias; a.out = out; a.partial = (float *)workspace;
    a.C = C; a.N = (int)g->N;                    // one segment per owned CSR row
    a.ptr = g->rowptr; a.idx = g->col; a.perm = g->rperm;
    a.seg_shift = 0; a.idx_shift = 0;
    a.n_split = g->n_split; a.n_med_end = g->rows_gt(SMALL_T); a.n_tasks = g->n_tasks;
    a.task_slot = g->task_slot; a.task_chunk = g->task_chunk; a.split_task0 = g->split_task0;
    a.nbA = a.nbB = 0;
    return dispatch_adj(cfg, a, (hipStream_t)stream);
}

static int scatter_sum_impl(const sngnn_graph_t *g, const float *vals, const float *w, int C, float *out,
                            void *workspace, void *stream);
extern "C" int sngnn_scatter_sum_rows(const sngnn_graph_t *g, const float *vals, int C, float *out,
                                      void *workspace, void *stream)
{
    return scatter_sum_impl(g, vals, nullptr, C, out, workspace, stream);
}
extern "C" int sngnn_weighted_scatter_sum_rows(const sngnn_graph_t *g, const float *vals, const float *w_csc, int C,
                                               float *out, void *workspace, void *stream)
{
    SN_REQUIRE(g == nullptr || g->Ep == 0 || w_csc != nullptr, SNGNN_EINVAL, "w_csc is NULL");
    return scatter_sum_impl(g, vals, w_csc, C, out, workspace, stream);
}
// out[p] = <a_rows[idx_a[p]], b_rows[idx_b[p]]> for p in [0, n_pairs): the gradient of a weighted gather-sum with
// respect to its per-entry weights (a_rows = the output's gradient by target, b_rows = the gathered table by source)
namespace sngnn {
static int dispatch_pair_dot(const RowCfg &cfg, const float *A, const int32_t *ia, const float *B, const int32_t *ib,
                             int64_t E, int C, float *out, hipStream_t st)
{
    switch (cfg.vec) {
    case 1: SNGNN_DISPATCH_GR(launch_pair_dot, 1, cfg, A, ia, B, ib, E, C, out, st)
    case 2: SNGNN_DISPATCH_GR(launch_pair_dot, 2, cfg, A, ia, B, ib, E, C, out, st)
    default: SNGNN_DISPATCH_GR(launch_pair_dot, 4, cfg, A, ia, B, ib, E, C, out, st)
    }
}
}  // namespace sngnn
extern "C" int sngnn_pair_dot_rows(const float *a_rows, const int32_t *idx_a, const float *b_rows, const int32_t *idx_b,
                                   int64_t n_pairs, int C, float *out, void *stream)
{
    SN_REQUIRE(n_pairs >= 0, SNGNN_EINVAL, "negative pair count");
    if (n_pairs == 0) return SNGNN_OK;
    SN_REQUIRE(a_rows && b_rows && idx_a && idx_b && out, SNGNN_EINVAL, "NULL argument");
    RowCfg cfg;
    SN_REQUIRE(row_cfg(C, cfg), SNGNN_EINVAL, "C must be in [1, " + std::to_string(SNGNN_MAX_CHANNELS) + "]");
    const uintptr_t al = (uintptr_t)cfg.vec * 4;
    SN_REQUIRE((uintptr_t)a_rows % al == 0 && (uintptr_t)b_rows % al == 0, SNGNN_EINVAL,
               "rows must be aligned to the row vector width");
    return dispatch_pair_dot(cfg, a_rows, idx_a, b_rows, idx_b, n_pairs, C, out, (hipStream_t)stream);
}
static int scatter_sum_impl(const sngnn_graph_t *g, const float *vals, const float *w, int C, float *out,
                            void *workspace, void *stream)
{
    SN_REQUIRE(g != nullptr, SNGNN_EINVAL, "graph is NULL");
    if (g->Ntot == 0) return SNGNN_OK;
    SN_REQUIRE(out && (vals || g->N == 0), SNGNN_EINVAL, "NULL argument");
    SN_REQUIRE(g->n_stasks == 0 || workspace, SNGNN_EINVAL, "workspace is NULL");
    RowCfg cfg;
    SN_REQUIRE(row_cfg(C, cfg), SNGNN_EINVAL,
               "C must be in [1, " + std::to_string(SNGNN_MAX_CHANNELS) + "]");
    AdjArgs a;
    a.table = vals; a.w = w; a.bias = nullptr; a.out = out; a.partial = (float *)workspace;
    a.C = C; a.N = (int)g->Ntot;                 // one segment per source node (CSC row)
    a.ptr = g->cscptr; a.idx = g->csc_dst; a.perm = g->sperm;
    a.seg_shift = 0; a.idx_shift = 0;
    a.n_split = g->n_ssplit; a.n_med_end = g->srcs_gt(SMALL_T); a.n_tasks = g->n_stasks;
    a.task_slot = g->stask_slot; a.task_chunk = g->stask_chunk; a.split_task0 = g->ssplit_task0;
    a.nbA = a.nbB = 0;
    return dispatch_adj(cfg, a, (hipStream_t)stream);
}
