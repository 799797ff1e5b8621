// C-ABI entry of the fused aggregation forward (see agg_fwd_impl.h for the kernels).
#include <cstdlib>

#include "agg_fwd_impl.h"

using namespace sngnn;

// Optional in-library timing of the forward's launches (bench.py's roofline leg):
// HIP events recorded on the caller's stream around each launch.
static bool g_prof_on = false;
static hipEvent_t g_prof_ev[3] = {nullptr, nullptr, nullptr};

extern "C" int sngnn_profile_enable(int on)
{
    if (on && !g_prof_ev[0])
        for (auto &e : g_prof_ev) SN_HIP(hipEventCreate(&e));
    g_prof_on = on != 0;
    return SNGNN_OK;
}

extern "C" int sngnn_profile_last_forward(float *main_ms, float *fin_ms)
{
    SN_REQUIRE(g_prof_ev[0] != nullptr && main_ms && fin_ms, SNGNN_EINVAL, "profiling is not enabled");
    SN_HIP(hipEventSynchronize(g_prof_ev[2]));
    SN_HIP(hipEventElapsedTime(main_ms, g_prof_ev[0], g_prof_ev[1]));
    SN_HIP(hipEventElapsedTime(fin_ms, g_prof_ev[1], g_prof_ev[2]));
    return SNGNN_OK;
}

extern "C" int sngnn_agg_forward(const sngnn_graph_t *g, const float *h, int C, int top_k,
                                 float thr, float *out, float *wsel, float *inv_norm,
                                 int32_t *sel_src, float *sel_w, void *workspace, void *stream)
{
    SN_REQUIRE(g != nullptr, SNGNN_EINVAL, "graph is NULL");
    SN_REQUIRE(g->N == 0 || (h != nullptr && out != nullptr), SNGNN_EINVAL, "h/out is NULL");
    RowCfg cfg;
    SN_REQUIRE(row_cfg(C, cfg), SNGNN_EINVAL,
               "C must be in [1, " + std::to_string(SNGNN_MAX_CHANNELS) + "]");
    SN_REQUIRE(((uintptr_t)h % (cfg.vec * 4)) == 0 && ((uintptr_t)out % (cfg.vec * 4)) == 0,
               SNGNN_EINVAL, "h/out must be aligned to the row vector width");
    SN_REQUIRE((sel_src == nullptr) == (sel_w == nullptr), SNGNN_EINVAL,
               "sel_src and sel_w go together");
    SN_REQUIRE(sel_src == nullptr || top_k >= 0, SNGNN_EINVAL, "sel_src needs top_k >= 0");
    SN_REQUIRE(sngnn_graph_workspace_bytes(g, C) == 0 || workspace != nullptr || g->n_tasks == 0,
               SNGNN_EINVAL, "workspace is NULL");
    hipStream_t st = (hipStream_t)stream;
    if (g->N == 0) return SNGNN_OK;
    if (top_k > (1 << 20)) top_k = 1 << 20;     // more than any row can use

    if (sel_src && top_k > 0) {
        SN_HIP(hipMemsetAsync(sel_src, 0xFF, (size_t)g->N * top_k * 4, st));
        SN_HIP(hipMemsetAsync(sel_w, 0, (size_t)g->N * top_k * 4, st));
    }

    FwdArgs a;
    a.h = h; a.C = C; a.N = (int)g->N; a.row_off = (int)g->row_off;
    a.rowptr = g->rowptr; a.col = g->col; a.rperm = g->rperm; a.rdesc = g->rdesc;
    a.nbC = 0;
    {   // tuning aids (unset in production)
        static const char *e_cls = getenv("SNGNN_DEBUG_CLASSES");
        static const char *e_bpc = getenv("SNGNN_DEBUG_BLOCKS_PER_CU");
        const char *c1 = getenv("SNGNN_DEBUG_LIVE") ? getenv("SNGNN_DEBUG_CLASSES") : e_cls;
        const char *c2 = getenv("SNGNN_DEBUG_LIVE") ? getenv("SNGNN_DEBUG_BLOCKS_PER_CU") : e_bpc;
        a.dbg_classes = c1 ? atoi(c1) : 7;
        a.dbg_blocks_per_cu = c2 ? atoi(c2) : 0;
        { const char *dy = getenv("SNGNN_DEBUG_LIVE") ? getenv("SNGNN_DYNAMIC") : nullptr;
          static const char *dy0 = getenv("SNGNN_DYNAMIC");
          if (!dy) dy = dy0;
          a.dynamic = dy ? atoi(dy) : 0; }
        static const char *e_fin = getenv("SNGNN_INKERNEL_FIN");
        a.inkernel_fin = e_fin ? atoi(e_fin) : 0;
        { const char *xa = getenv("SNGNN_DEBUG_LIVE") ? getenv("SNGNN_XCD_AFFINITY") : nullptr;
          static const char *xa0 = getenv("SNGNN_XCD_AFFINITY");
          if (!xa) xa = xa0;
          a.xcd_affinity = xa ? atoi(xa) : 0; }
        { const char *lf = getenv("SNGNN_DEBUG_LIVE") ? getenv("SNGNN_INKERNEL_FIN") : nullptr; if (lf) a.inkernel_fin = atoi(lf); }
        const char *e_dma = getenv("SNGNN_DEBUG_LIVE") ? getenv("SNGNN_FWD_DMA") : nullptr;
        static const char *e_dma0 = getenv("SNGNN_FWD_DMA");
        if (!e_dma) e_dma = e_dma0;
        a.use_dma = (e_dma ? atoi(e_dma) : 1) && (C % 4 == 0) && (C <= 256);
    }
    a.k = top_k < 0 ? -1 : top_k; a.thr = thr;
    a.out = out; a.wsel = wsel; a.inv_norm = inv_norm;
    a.sel_src = top_k > 0 ? sel_src : nullptr; a.sel_w = top_k > 0 ? sel_w : nullptr;
    a.n_split = g->n_split;
    a.n_med_end = g->rows_gt(SMALL_T);
    a.n_tasks = g->n_tasks;
    a.task_slot = g->task_slot; a.task_chunk = g->task_chunk;
    a.split_soff = g->split_soff; a.split_task0 = g->split_task0;
    a.xtask_list = g->xtask_list; a.xtask_ptr = g->xtask_ptr;
    a.dyn_ctr = g->dyn_ctr;
    a.scores = (float *)workspace;
    a.partial = a.scores ? a.scores + (g->split_edges + 3) / 4 * 4 : nullptr;   // 16-B aligned rows
    a.cand_key = a.partial ? (unsigned long long *)(a.partial + ((size_t)g->n_tasks * C + 3) / 4 * 4)
                           : nullptr;
    a.split_cnt = g->split_cnt; a.grp_cnt = g->grp_cnt; a.split_grp0 = g->split_grp0;
    a.cand2 = a.cand_key ? a.cand_key + (size_t)g->n_tasks * 32 : nullptr;
    a.cand_src = a.cand2 ? (int32_t *)(a.cand2 + (size_t)g->n_groups * 32) : nullptr;
    // split rows whose (tasks * top_k) candidates exceed one 128-key wave selection
    a.n_split_gt_wave = top_k > 0 ? g->rows_gt((int64_t)(128 / std::min(top_k, 128)) * CHUNK) : 0;
    a.lowbits = 1;
    while ((1ll << a.lowbits) < g->max_in_deg && a.lowbits < 31) ++a.lowbits;
    a.nbA = ceil_div(g->n_tasks, WAVES);
    a.nbB = ceil_div(a.n_med_end - a.n_split, WAVES);
    const int max_split = g->n_split ? g->rdeg[0] : 0;
    hipEvent_t *ev = g_prof_on ? g_prof_ev : nullptr;
    switch (cfg.vec) {
    case 1: return launch_agg_fwd_v1(cfg, a, max_split, ev, st);
    case 2: return launch_agg_fwd_v2(cfg, a, max_split, ev, st);
    default: return launch_agg_fwd_v4(cfg, a, max_split, ev, st);
    }
}
