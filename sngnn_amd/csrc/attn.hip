// C-ABI entries of the cosine-attention mode (kernels: attn_impl.h).
#include "attn_impl.h"

using namespace sngnn;

static int check_rows(int C, RowCfg &cfg, const void *p0, const void *p1, const void *p2)
{
    SN_REQUIRE(row_cfg(C, cfg), SNGNN_EINVAL,
               "C must be in [1, " + std::to_string(SNGNN_MAX_CHANNELS) + "]");
    const uintptr_t al = (uintptr_t)cfg.vec * 4;
    SN_REQUIRE((uintptr_t)p0 % al == 0 && (uintptr_t)p1 % al == 0 && (uintptr_t)p2 % al == 0,
               SNGNN_EINVAL, "feature tables must be aligned to the row vector width");
    return SNGNN_OK;
}

extern "C" int sngnn_attn_forward(const sngnn_graph_t *g, const float *h, int C, float *out,
                                  float *alpha, void *workspace, void *stream)
{
    SN_REQUIRE(g != nullptr, SNGNN_EINVAL, "graph is NULL");
    if (g->N == 0) return SNGNN_OK;
    SN_REQUIRE(h && out, SNGNN_EINVAL, "NULL argument");
    SN_REQUIRE(workspace != nullptr || g->n_tasks == 0, SNGNN_EINVAL, "workspace is NULL");
    RowCfg cfg;
    if (int rc = check_rows(C, cfg, h, out, nullptr)) return rc;
    AttnArgs a;
    a.h = h; a.C = C; a.N = (int)g->N; a.row_off = (int)g->row_off;
    a.col = g->col; a.rdesc = g->rdesc;
    a.out = out; a.alpha = alpha;
    a.n_split = g->n_split; a.n_med_end = g->rows_gt(SMALL_T); a.n_tasks = g->n_tasks;
    a.task_slot = g->task_slot; a.task_chunk = g->task_chunk; a.split_task0 = g->split_task0;
    a.partial = (float *)workspace;     // [n_tasks][C + 4] <= the forward workspace of the graph
    a.nbA = a.nbB = 0;
    hipStream_t st = (hipStream_t)stream;
    switch (cfg.vec) {
    case 1: return launch_attn_fwd_v1(cfg, a, st);
    case 2: return launch_attn_fwd_v2(cfg, a, st);
    default: return launch_attn_fwd_v4(cfg, a, st);
    }
}

extern "C" int sngnn_attn_backward(const sngnn_graph_t *g, const float *h, int C,
                                   const float *grad_out, const float *alpha, float *grad_h,
                                   void *workspace, void *stream)
{
    SN_REQUIRE(g != nullptr, SNGNN_EINVAL, "graph is NULL");
    if (g->Ntot == 0) return SNGNN_OK;
    SN_REQUIRE(h && grad_h && workspace && (grad_out || g->N == 0), SNGNN_EINVAL, "NULL argument");
    SN_REQUIRE(alpha != nullptr || g->Ep == 0, SNGNN_EINVAL, "alpha is NULL");
    RowCfg cfg;
    if (int rc = check_rows(C, cfg, h, grad_out, grad_h)) return rc;
    BwdArgs a;
    a.h = h; a.gout = grad_out; a.wsel = alpha;
    a.C = C; a.N = (int)g->N; a.Ntot = (int)g->Ntot; a.row_off = (int)g->row_off;
    a.rowptr = g->rowptr; a.col = g->col; a.rperm = g->rperm; a.rdesc = g->rdesc; a.sdesc = g->sdesc;
    a.cscptr = g->cscptr; a.csc_eid = g->csc_eid; a.csc_dst = g->csc_dst; a.csc_pos = g->csc_pos;
    a.sperm = g->sperm;
    // workspace layout (sngnn_graph_workspace_bytes): wd (2 floats per edge) | dnT | partT (2C+4 per task) | partS
    float *ws = (float *)workspace;
    a.wd = (float2 *)ws;
    a.kmask = nullptr; a.kmask_words = 0; a.inv_deg = nullptr;
    const size_t ds_len = (2 * (size_t)g->Ep + 3) / 4 * 4;
    a.dnT = ws + ds_len;
    a.partT = a.dnT + (size_t)g->N * C;
    a.partS = a.partT + (size_t)g->n_tasks * (2 * C + 4);
    a.rec_dot = a.partS + (size_t)g->n_stasks * 2 * C;
    a.grad_h = grad_h;
    a.n_split = g->n_split; a.n_med_end = g->rows_gt(SMALL_T); a.n_tasks = g->n_tasks;
    a.task_slot = g->task_slot; a.task_chunk = g->task_chunk; a.split_task0 = g->split_task0;
    a.n_ssplit = g->n_ssplit; a.n_smed_end = g->srcs_gt(SMALL_T); a.n_stasks = g->n_stasks;
    a.stask_slot = g->stask_slot; a.stask_chunk = g->stask_chunk; a.ssplit_task0 = g->ssplit_task0;
    a.nbA = a.nbB = a.nbC = 0;
    a.mode = 1; a.top_k = -1; a.role_mask = 3; a.s_small_end = (int)g->Ntot; a.Ep = g->Ep;
    a.fdesc = nullptr; a.trest = nullptr; a.n_fused = a.n_trest = 0;
    a.kbits = nullptr; a.csc_bit = nullptr; a.kb_wbase = a.kb_tbase = 0;
    hipStream_t st = (hipStream_t)stream;
    switch (cfg.vec) {
    case 1: return launch_attn_bwd_v1(cfg, a, st);
    case 2: return launch_attn_bwd_v2(cfg, a, st);
    default: return launch_attn_bwd_v4(cfg, a, st);
    }
}
