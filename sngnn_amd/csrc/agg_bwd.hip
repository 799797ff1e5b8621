// C-ABI entry of the aggregation backward (kernels: agg_bwd_impl.h).
#include "agg_bwd_impl.h"

using namespace sngnn;

// sngnn_tuning_set(3, v): 0 = node-centric backward when most nodes are fused work items (default),
// 1 = the two passes for every node, 2 = node-centric whatever the graph (the same bits;
// measurement / test aid)
static int g_bwd_mode = 0, g_bwd_roles = 3;
int sngnn::set_bwd_mode(int v) { g_bwd_mode = (v == 1 || v == 2) ? v : 0; return SNGNN_OK; }
int sngnn::set_bwd_roles(int v) { g_bwd_roles = v & 7; return SNGNN_OK; }

extern "C" int sngnn_agg_backward(const sngnn_graph_t *g, const float *h, int C,
                                  const float *grad_out, const float *wsel, float *grad_h,
                                  void *workspace, void *stream)
{
    return sngnn_agg_backward_topk(g, h, C, grad_out, wsel, -1, grad_h, workspace, stream);
}

// whether a forward with this top_k can write the kept bits itself AND the backward that would read
// them is the hinted node-centric one (agg_fwd.hip asks the same question)
static bool node_centric(const sngnn_graph_t *g)
{
    const bool mostly_fused = (int64_t)g->n_fused * 2 >= g->N;
    return g_bwd_mode == 2 || (g_bwd_mode == 0 && mostly_fused);
}
bool sngnn::kept_bits_path(const sngnn_graph_t *g, int top_k)
{
    return g != nullptr && g->csc_bit != nullptr && g->N == g->Ntot && top_k >= 1 && top_k <= SMALL_T && node_centric(g) &&
           !fwd_scores_on_the_fly_forced();       // (the forward writes the bits in table mode only)
}

static int backward_impl(const sngnn_graph_t *g, const float *h, int C, const float *grad_out, const float *wsel,
                         const unsigned *kbits, int top_k, float *grad_h, void *workspace, void *stream);

extern "C" int sngnn_agg_backward_topk(const sngnn_graph_t *g, const float *h, int C,
                                       const float *grad_out, const float *wsel, int top_k,
                                       float *grad_h, void *workspace, void *stream)
{
    SN_REQUIRE(g == nullptr || wsel != nullptr || g->Ep == 0, SNGNN_EINVAL, "wsel is NULL");
    return backward_impl(g, h, C, grad_out, wsel, nullptr, top_k, grad_h, workspace, stream);
}

extern "C" int sngnn_agg_backward_bits(const sngnn_graph_t *g, const float *h, int C, const float *grad_out,
                                       const void *kept_bits, int top_k, float *grad_h, void *workspace,
                                       void *stream)
{
    SN_REQUIRE(g != nullptr && kept_bits != nullptr, SNGNN_EINVAL, "graph / kept_bits is NULL");
    SN_REQUIRE(kept_bits_path(g, top_k), SNGNN_EINVAL,
               "no kept-bit path for this graph / top_k (sngnn_agg_kept_bits_supported)");
    return backward_impl(g, h, C, grad_out, nullptr, (const unsigned *)kept_bits, top_k, grad_h, workspace, stream);
}

static int backward_impl(const sngnn_graph_t *g, const float *h, int C, const float *grad_out, const float *wsel,
                         const unsigned *kbits, int top_k, float *grad_h, void *workspace, void *stream)
{
    SN_REQUIRE(g != nullptr, SNGNN_EINVAL, "graph is NULL");
    if (g->Ntot == 0) return SNGNN_OK;
    SN_REQUIRE(h && grad_h && workspace && (grad_out || g->N == 0), SNGNN_EINVAL, "NULL argument");
    RowCfg cfg;
    SN_REQUIRE(row_cfg(C, cfg), SNGNN_EINVAL,
               "C must be in [1, " + std::to_string(SNGNN_MAX_CHANNELS) + "]");
    const uintptr_t al = (uintptr_t)cfg.vec * 4;
    SN_REQUIRE((uintptr_t)h % al == 0 && (uintptr_t)grad_out % al == 0 && (uintptr_t)grad_h % al == 0,
               SNGNN_EINVAL, "h/grad_out/grad_h must be aligned to the row vector width");
    BwdArgs a;
    a.h = h; a.gout = grad_out; a.wsel = wsel;
    a.C = C; a.N = (int)g->N; a.Ntot = (int)g->Ntot; a.row_off = (int)g->row_off;
    a.rowptr = g->rowptr; a.col = g->col; a.rperm = g->rperm; a.rdesc = g->rdesc; a.sdesc = g->sdesc;
    a.cscptr = g->cscptr; a.csc_eid = g->csc_eid; a.csc_dst = g->csc_dst; a.csc_pos = g->csc_pos;
    a.sperm = g->sperm;
    // workspace layout (sngnn_graph_workspace_bytes): [2 floats per edge: the kept-bit mask
    // lives in its first words; the attention mode keeps records there] | dnT | partT | partS
    float *ws = (float *)workspace;
    a.wd = nullptr; a.rec_dot = nullptr;
    a.kmask = (unsigned *)ws;
    a.kmask_words = (g->Ep + 31) / 32;
    a.Ep = g->Ep;
    a.inv_deg = g->inv_deg;
    const size_t ds_len = (2 * (size_t)g->Ep + 3) / 4 * 4;  // keep rows 16-byte aligned
    a.dnT = ws + ds_len;
    a.partT = a.dnT + (size_t)g->N * C;
    a.partS = a.partT + (size_t)g->n_tasks * C;
    a.grad_h = grad_h;
    a.n_split = g->n_split; a.n_med_end = g->rows_gt(SMALL_T); a.n_tasks = g->n_tasks;
    a.task_slot = g->task_slot; a.task_chunk = g->task_chunk; a.split_task0 = g->split_task0;
    a.n_ssplit = g->n_ssplit; a.n_smed_end = g->srcs_gt(SMALL_T); a.n_stasks = g->n_stasks;
    a.stask_slot = g->stask_slot; a.stask_chunk = g->stask_chunk; a.ssplit_task0 = g->ssplit_task0;
    a.nbA = a.nbB = a.nbC = 0;
    // The node-centric path pays when most nodes are FUSED work items (small both as target and as
    // source: four nodes per wave, both passes in one go) - 95 % at arxiv's degree law, where it wins
    // at every size tried (x1 .. x32: 1.2 M .. 37 M edges, 60 vs 77 us .. 2.26 vs 2.72 ms).  Where
    // nodes mostly have 17..128 out-edges (products' degree law: no fused node at all) every node
    // would be a wave-per-node item, one node per wave with a longer dependent chain than either
    // pass has: 4.9 ms against the two passes' 4.2 ms at products size.  So: node-centric when at
    // least half of the owned nodes are fused.
    a.mode = node_centric(g) ? 0 : 1;   // (knob 3 = 2: node-centric whatever the graph - measurement)
    a.top_k = top_k;
    a.role_mask = g_bwd_roles;
    a.fdesc = g->fdesc; a.trest = g->trest;
    a.n_fused = g->n_fused; a.n_trest = g->n_trest;
    a.kbits = kbits; a.csc_bit = g->csc_bit; a.kb_wbase = (int)g->kb_wbase; a.kb_tbase = (int)g->kb_tbase;
    a.s_small_end = a.mode == 0 ? g->srcs_gt(SMALL_T - 1) : (int)g->Ntot;
    hipStream_t st = (hipStream_t)stream;
    switch (cfg.vec) {
    case 1: return launch_agg_bwd_v1(cfg, a, st);
    case 2: return launch_agg_bwd_v2(cfg, a, st);
    default: return launch_agg_bwd_v4(cfg, a, st);
    }
}
