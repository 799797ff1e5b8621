// Shared host/device definitions of libsngnn_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <string>
#include <vector>

#include "sngnn_hip.h"

namespace sngnn {

// ---- row classes of the aggregation kernels (by in-degree) -----------------
constexpr int SMALL_T = 16;    // deg <= SMALL_T : one G-lane group per row
constexpr int WAVE_T = 128;    // deg <= WAVE_T  : one wave per row
constexpr int CHUNK = 128;     // deg >  WAVE_T  : split into CHUNK-edge wave tasks
constexpr int BLOCK = 256;     // threads per workgroup of the main kernels
constexpr int WAVES = BLOCK / 64;
constexpr int FIN_BLOCK = 1024;   // threads per workgroup of the split-row finalize
constexpr float EPS_NORM = 1e-12f;   // F.normalize eps (models.py:122,238,325)

void set_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define SN_HIP(expr)                                                       \
    do {                                                                   \
        hipError_t _e = (expr);                                            \
        if (_e != hipSuccess) return sngnn::hip_fail(_e, #expr, __FILE__, __LINE__); \
    } while (0)

#define SN_REQUIRE(cond, code, msg)                                        \
    do {                                                                   \
        if (!(cond)) { sngnn::set_error(msg); return (code); }             \
    } while (0)

// MFMA weight gradient of self.lin (linear.hip), used by sngnn_linear_wgrad (head.hip)
constexpr int WGRAD_MFMA_WGS = 256;           // persistent workgroups = partial results
int wgrad_mfma_partials(int64_t N, int C, int F);
int launch_wgrad_mfma(const float *g, const float *x, int64_t N, int C, int F, float *part, float *part_b,
                      hipStream_t st);

bool kept_bits_path(const sngnn_graph_t *g, int top_k);
bool fwd_scores_on_the_fly_forced();          // agg_fwd.hip (knob 2 == 2)     // agg_bwd.hip: forward-written kept bits usable
int set_bwd_mode(int v);      // agg_bwd.hip (sngnn_tuning_set knobs 3, 4)
int set_bwd_roles(int v);
int set_lin_mode(int v);       // linear.hip (knob 5)
int set_knn_route(int v);      // knn.hip (knob 6)
int set_cosine_split(int v);   // toolbox.hip (knob 7)
bool fp32_mfma_only();         // knob 5 == 1: the fp32 contractions stay on fp32 MFMAs

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Lane layout of one node row for C channels:
//   VEC floats per lane per step, G lanes per group, R steps  (C <= VEC*G*R).
struct RowCfg { int vec, g, r; };
// (the forward, backward, attention and adjacency kernels are instantiated for the layouts
// this function can return: SNGNN_DISPATCH_GR below)
inline bool row_cfg(int C, RowCfg &cfg)
{
    if (C < 1 || C > SNGNN_MAX_CHANNELS) return false;
    int vec = (C % 4 == 0) ? 4 : (C % 2 == 0) ? 2 : 1;
    int lanes = C / vec;
    int g = 8;
    while (g < lanes && g < 64) g <<= 1;
    int r = (lanes + g - 1) / g;
    int rr = 1;
    while (rr < r) rr <<= 1;
    if (rr > 8) return false;
    cfg = {vec, g, rr};
    return true;
}

// bytes of one fp16 filter row (agg_fwd_filter.h): whole 128-byte lines; 0 = no filter for this C
// (it pays when a unit row is longer than one line and the rows are 16-byte vectors)
inline int64_t filter_row_bytes(int C)
{
    if (C % 4 != 0 || C <= 32 || C > SNGNN_MAX_CHANNELS) return 0;
    int64_t b = 128;                 // 2 bytes x (4 G R) channels of the row layout: 64, 128, 256 or 512
    while (b < 2 * (int64_t)C) b <<= 1;
    return b;
}

// bytes of the forward's unit-row table + norms + filter rows in front of its scratch inside
// the workspace (256-byte aligned regions)
inline int64_t fwd_table_bytes(int64_t Ntot, int C)
{
    return (Ntot * (int64_t)C * 4 + 255) / 256 * 256 + (Ntot * 4 + 255) / 256 * 256 +
           (Ntot * filter_row_bytes(C) + 255) / 256 * 256;
}

}  // namespace sngnn

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

// The graph object behind the opaque handle.
struct sngnn_graph {
    int64_t N = 0, E_in = 0, Ep = 0;   // N = target rows owned by this graph
    int64_t Ntot = 0, row_off = 0;       // sources / feature-table rows; first owned node id
    int add_loops = 0, remove_loops = 0;
    // device arrays
    int32_t *rowptr = nullptr, *col = nullptr, *eid = nullptr;
    int32_t *cscptr = nullptr, *csc_eid = nullptr, *csc_dst = nullptr;
    int32_t *csc_pos = nullptr;       // [E'] CSR edge index -> its position in the CSC order (inverse of csc_eid)
    int32_t *rperm = nullptr, *sperm = nullptr;
    int4 *rdesc = nullptr;     // [N] per slot of rperm: {row, first edge, in-degree, first entry in col_s}
    int32_t *col_s = nullptr;  // [E'] col with the rows in slot order (forward, small rows)
    // the same three for the forward's streaming order (small rows by 4-degree bucket, natural inside)
    int32_t *rperm_b = nullptr, *col_s_b = nullptr;
    int4 *rdesc_b = nullptr;
    int4 *sdesc = nullptr;     // [Ntot] per slot of sperm: {source, first CSC entry, out-degree, 0}
    float *inv_deg = nullptr;  // [N] 1 / max(in-degree, 1) by row (backward pass S)
    // backward, node-centric path: the small sources of sdesc / sperm are ordered [not fused |
    // fused], natural order inside each; fused = owned, in-degree and out-degree <= SMALL_T
    // kept-bit layout the forward can write without atomics ("kbits": agg_fwd_impl.h, agg_bwd_impl.h):
    //   words [0, kb_wbase): one HALFWORD per row id (bit t = edge t of a small row);
    //   [kb_wbase, kb_tbase): 4 words per wave row, by slot;  [kb_tbase, kb_words): 4 words per split-row task.
    // csc_bit[q] = the bit index of CSC entry q's edge in that layout (static); nullptr = layout unavailable
    int32_t *csc_bit = nullptr;
    int64_t kb_wbase = 0, kb_tbase = 0, kb_words = 0;
    int4 *fdesc = nullptr;     // [n_fused] {node, first in-edge, first CSC entry, in-degree | out-degree << 8}
    int4 *trest = nullptr;     // [n_trest] {row, first edge, in-degree, 0}: small targets with out-degree > SMALL_T
    int n_fused = 0, n_trest = 0;
    // split rows (in-degree > WAVE_T) = the first n_split slots of rperm
    int32_t *task_slot = nullptr, *task_chunk = nullptr;   // [n_tasks]
    int32_t *task_order = nullptr;    // [n_tasks] dealing order of the forward's tasks (XCD-affine by source slice)
    int32_t *split_soff = nullptr;    // [n_split+1] offset of the row's scores in scratch
    int32_t *split_task0 = nullptr;   // [n_split+1] first task of the row
    // split sources (out-degree > WAVE_T) = the first n_ssplit slots of sperm
    int32_t *stask_slot = nullptr, *stask_chunk = nullptr;  // [n_stasks]
    int32_t *ssplit_task0 = nullptr;  // [n_ssplit+1]
    // host copies
    std::vector<int32_t> rdeg;   // in-degrees, descending  (rdeg[p] = deg(rperm[p]))
    std::vector<int64_t> rdeg_wave_psum;   // prefix sums of rdeg over the rows ABOVE the small class (rows_gt(SMALL_T) + 1 entries)
    std::vector<int32_t> sdeg;   // out-degrees, descending
    int64_t max_in_deg = 0, max_out_deg = 0, src_min = 0;
    int n_split = 0, n_tasks = 0;
    int64_t split_edges = 0;
    int n_ssplit = 0, n_stasks = 0;
    int device = 0;

    // number of rows with in-degree > t (rdeg is descending)
    int rows_gt(int64_t t) const
    {
        int lo = 0, hi = (int)rdeg.size();
        while (lo < hi) { int m = (lo + hi) / 2; if (rdeg[m] > t) lo = m + 1; else hi = m; }
        return lo;
    }
    // edges a top_k selection can PRUNE in rows of at least min_deg in-edges: sum of (deg - top_k) over the rows with
    // deg >= min_deg and deg > top_k (those rows rank) - the fp16 filter's rule (agg_fwd.hip:use_filter)
    int64_t prunable_edges(int top_k, int min_deg) const
    {
        const int64_t t = std::max<int64_t>(std::max<int64_t>(top_k, (int64_t)min_deg - 1), (int64_t)sngnn::SMALL_T);
        const int m = rows_gt(t);
        if (m <= 0 || (size_t)m >= rdeg_wave_psum.size()) return 0;
        return rdeg_wave_psum[(size_t)m] - (int64_t)std::max(top_k, 0) * m;
    }
    int srcs_gt(int64_t t) const
    {
        int lo = 0, hi = (int)sdeg.size();
        while (lo < hi) { int m = (lo + hi) / 2; if (sdeg[m] > t) lo = m + 1; else hi = m; }
        return lo;
    }
};
