/*
 * sngnn_hip.h - C ABI of the MI355X (gfx950) implementation of SNGNN's
 * similarity-navigated aggregation path.
 *
 * The reference (MinhZou/SNGNN) is pure Python and has no FFI of its own; the
 * seam this library replaces is the body of the three conv layers'
 * ``forward(x, edge_index)`` after ``self.lin`` (models/models.py:116-137,
 * 233-242, 322-329) together with their ``message`` hooks (:139-158, :244-263,
 * :331-334) and the third-party calls underneath them (PyG propagate /
 * add_self_loops / remove_self_loops, torch_scatter.scatter_max / scatter(mean),
 * torch_sparse.SparseTensor) - plus the Sim-GFA toolbox's cosine statistics
 * (SimGFAToolbox/dense.py:138-179).  Each entry point cites the reference lines
 * it stands in for.  INTEGRATION.md shows the ctypes binding a maintainer of the
 * reference would add.
 *
 * Conventions
 *   - every pointer marked "dev" is a device (HBM) pointer; nothing here takes
 *     or returns a torch type;
 *   - fp32 values, row-major, dense (leading dimension == number of columns);
 *   - ``stream`` is a hipStream_t passed as void* (NULL = the null stream);
 *     forward/backward entry points only enqueue work: no allocation, no host
 *     synchronisation, safe to capture in a hipGraph;
 *   - every function returns 0 on success or a negative SNGNN_E* code;
 *     sngnn_last_error() gives the message of the calling thread's last failure.
 */
#ifndef SNGNN_HIP_H
#define SNGNN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNGNN_OK            0
#define SNGNN_EINVAL       -1   /* bad argument (NULL, negative size, C unsupported) */
#define SNGNN_ERANGE       -2   /* node id outside [0, N) in edge_index            */
#define SNGNN_EHIP         -3   /* a HIP runtime call failed                       */
#define SNGNN_ENOMEM       -4   /* allocation failed                               */

/* weight value stored for an edge that was NOT selected (training-mode output
 * of sngnn_agg_forward); a selected edge stores its cosine, which is > -1.1 */
#define SNGNN_UNSELECTED   (-4.0f)

/* largest supported number of channels per node row */
#define SNGNN_MAX_CHANNELS  512

const char *sngnn_last_error(void);
/* "gfx950;rocm-x.y;..." build identification */
const char *sngnn_build_info(void);

/* ------------------------------------------------------------------------
 * Graph structure (built once per edge_index / self-loop mode and cached by
 * the caller; the reference redoes this work on every forward).
 * ------------------------------------------------------------------------ */
typedef struct sngnn_graph sngnn_graph_t;

/*
 * Replaces: add_self_loops + remove_self_loops (models.py:117-120, 234-236,
 * 323) and the COO->per-target grouping that PyG's propagate / torch_scatter
 * perform implicitly on every call (models.py:132, 239, 326).
 *
 *   edge_index_dev  int64 [2, E] row-major (row 0 = source, row 1 = target),
 *                   exactly what the reference passes as ``edge_index``
 *   add_loops       append (v, v) for every v at the END of the list
 *   remove_loops    1: then drop EVERY edge with src == dst (so add+remove == remove)
 *                   2 (SNGNN_LOOPS_REPLACE): drop the ORIGINAL loops first, then
 *                   append - AGNNConv's order (models.py:393-395)
 *
 * The resulting list of E' edges keeps the reference's order; the library
 * stores it as CSR by target with the original relative order inside each row
 * (stable), which is what makes "first occurrence wins a tie" reproducible.
 * Synchronises the stream (one-time setup).
 */
#define SNGNN_LOOPS_REPLACE 2
int sngnn_graph_create(const int64_t *edge_index_dev, int64_t E, int64_t N,
                       int add_loops, int remove_loops, void *stream,
                       sngnn_graph_t **out_graph);
/*
 * Node-range partition of a larger graph (one process per GPU): this graph owns
 * the target rows [row_begin, row_end) of an N_total-node graph.  edge_index
 * holds GLOBAL node ids; edges whose target lies outside the owned range are
 * dropped, self-loops are appended for the owned nodes only.  Feature tables
 * handed to forward/backward then have N_total rows (the all-gathered h), the
 * outputs row_end - row_begin rows.  The reference has no counterpart (it is
 * single-device); with row range [0, N) this is sngnn_graph_create.
 */
int sngnn_graph_create_partition(const int64_t *edge_index_dev, int64_t E,
                                 int64_t N_total, int64_t row_begin,
                                 int64_t row_end, int add_loops, int remove_loops,
                                 void *stream, sngnn_graph_t **out_graph);
void sngnn_graph_destroy(sngnn_graph_t *g);

int64_t sngnn_graph_num_nodes(const sngnn_graph_t *g);       /* owned target rows */
int64_t sngnn_graph_num_total_nodes(const sngnn_graph_t *g); /* N_total            */
int64_t sngnn_graph_row_offset(const sngnn_graph_t *g);      /* row_begin          */
int64_t sngnn_graph_num_edges(const sngnn_graph_t *g);      /* E' */
int64_t sngnn_graph_max_in_degree(const sngnn_graph_t *g);
int64_t sngnn_graph_src_min(const sngnn_graph_t *g);         /* models.py:125 */
/* owned nodes with in-degree and out-degree <= 16: the one-work-item nodes of the node-centric
 * backward, which sngnn_agg_backward* takes when they are at least half of the owned nodes */
int64_t sngnn_graph_num_fused_nodes(const sngnn_graph_t *g);
/* bytes of device workspace sngnn_agg_forward/backward need for C channels */
int64_t sngnn_graph_workspace_bytes(const sngnn_graph_t *g, int C);

/* Copy one of the graph's arrays to host memory (tests / inspection).
 *   which: 0 rowptr   int32 [N+1]   CSR by target
 *          1 col      int32 [E']    source id of each CSR edge
 *          2 eid      int32 [E']    position of the CSR edge in the E' edge list
 *          3 cscptr   int32 [N_total+1]  CSC by source
 *          4 csc_eid  int32 [E']    CSR edge index of each CSC entry
 *          5 rperm    int32 [N]     rows sorted by in-degree, descending (stable)
 */
int sngnn_graph_copy_array(const sngnn_graph_t *g, int which, void *host_dst);
/* device pointer to the same arrays (owned by the graph) */
const void *sngnn_graph_array_dev(const sngnn_graph_t *g, int which);

/* ------------------------------------------------------------------------
 * Fused aggregation:  F.normalize + per-edge cosine + top-k/threshold
 * selection + similarity-weighted scatter-mean.
 * ------------------------------------------------------------------------ */
/*
 * Replaces: models.py:122+132+139-158 (SNConv_plus_plus, the out_1 branch),
 * :238-239+244-263 (SNConv_plus), :325-326+331-334 (SNConv) and, underneath,
 * PyG propagate's four index_select gathers, torch_scatter.scatter_max (x top_k
 * rounds) and torch_scatter.scatter(reduce='mean').
 *
 *   h        dev f32 [N_total, C]  output of self.lin (NOT normalised); N_total == N
 *                             unless the graph is a partition
 *   top_k    < 0: SNConv - every edge weighted by its cosine, no selection
 *            >= 0: keep, per target, the top_k in-edges by (cosine descending,
 *            edge position ascending) and of those only the ones with
 *            cosine >= (float)thr; all other edges get weight 0
 *   out      dev f32 [N, C]   (1 / max(indeg', 1)) * sum_e weight_e * h[src_e]
 *                             (the mean divides by the FULL in-degree)
 * Optional outputs (NULL to skip):
 *   wsel     dev f32 [E']     per CSR edge: its cosine if selected, else
 *                             SNGNN_UNSELECTED (saved for backward)
 *   inv_norm dev f32 [N]      1 / max(||h_i||_2, 1e-12)
 *   sel_src  dev i32 [N, top_k]  selected source ids in rank order, -1 padded
 *   sel_w    dev f32 [N, top_k]  their cosines (0 padded)
 *   workspace dev, sngnn_graph_workspace_bytes(g, C) bytes (may be NULL when
 *            that is 0)
 */
int sngnn_agg_forward(const sngnn_graph_t *g, const float *h, int C, int top_k,
                      float thr, float *out, float *wsel, float *inv_norm,
                      int32_t *sel_src, float *sel_w, void *workspace,
                      void *stream);

/*
 * The two halves of sngnn_agg_forward, for a caller that already holds the unit rows
 * (e.g. produced by the epilogue of its own ``lin``) or wants them.
 *
 * sngnn_normalize_rows replaces F.normalize(x, p=2, dim=-1) (models.py:122,238,325):
 *   n[r, :] = h[r, :] / max(||h[r, :]||_2, 1e-12),  nrm[r] = that clamped norm,
 * IEEE square root and division, so rows that the reference normalises to the same
 * bits (duplicates, power-of-two multiples, rows with a single non-zero channel) get the
 * same bits here and their cosines tie exactly as in the reference.
 *   h, n  dev f32 [rows, C];  nrm dev f32 [rows]
 *
 * sngnn_agg_forward_normalized is sngnn_agg_forward on (n, nrm) of the N_total feature
 * rows: the per-edge cosine is <n_i, n_j>, a kept message is cosine * nrm_j * n_j
 * (= cosine * h_j to half an ulp).  Same workspace size and optional outputs.
 */
int sngnn_normalize_rows(const float *h, int64_t rows, int C, float *n, float *nrm,
                         void *stream);
/*
 * Filter rows (no reference counterpart; csrc/agg_fwd_filter.h).  For C % 4 == 0, C > 32 the
 * selecting forward first scores the edges of a row with more than top_k in-edges against an
 * fp16 copy of the unit rows, one 128-byte-aligned entry of sngnn_filter_row_bytes(C) bytes per
 * node, and fetches the fp32 unit row only of the edges that can still be among the top_k
 * (approximate cosine within a proven bound of the k-th); the selection itself is made on exact
 * fp32 cosines, so indices and weights are those of the unfiltered path, bit for bit.
 * sngnn_filter_row_bytes returns 0 when the filter is not used for this C.
 * sngnn_normalize_rows_filter is sngnn_normalize_rows that also writes the filter rows
 * (filt: dev, rows * sngnn_filter_row_bytes(C) bytes, 16-byte aligned; NULL to skip).
 * sngnn_agg_forward_prepared is sngnn_agg_forward_normalized for a caller that holds them
 * (filt == NULL: they are built inside the workspace by one more pass over n).
 * sngnn_filter_enable(mode): 0 = never, 1 = when it is expected to pay (default: a selective
 * threshold, thr >= 0.25; round 3 also switched it on for top_k <= 8: re-measured, it no longer pays there),
 * 2 = whenever it applies (the small rows' two-phase form too), 3 = for the rows above the small class
 * (wave rows, split-row tasks) at any threshold (measurement: the regime where the unit-row table does not
 * fit the Infinity Cache).  Results do not depend on it.
 */
int64_t sngnn_filter_row_bytes(int C);
int sngnn_normalize_rows_filter(const float *h, int64_t rows, int C, float *n, float *nrm,
                                void *filt, void *stream);
int sngnn_agg_forward_prepared(const sngnn_graph_t *g, const float *n, const float *nrm,
                               const void *filt, int C, int top_k, float thr, float *out,
                               float *wsel, float *inv_norm, int32_t *sel_src, float *sel_w,
                               void *workspace, void *stream);

/*
 * The forward with a hidden layer's store epilogue.  Replaces, on top of sngnn_agg_forward /
 * sngnn_agg_forward_prepared: the elementwise passes the reference runs between two conv layers
 * (models.py:204-209 / :79-84 / :296-301): `out + self.bias` (:135-136, 240-241, 327-328),
 * `F.relu(x, inplace=True)` and, in training, `self.dropout(x)` - applied to each finished mean row
 * on its way out instead of three more passes over [N, C]:
 *     out[i, c] = keep[i, c] ? max(mean[i, c] + bias[c], 0) * keep_scale : 0
 * bias NULL: none; relu 0: none; keep NULL and seed NULL: no dropout (evaluation, or p == 0).  keep
 * is the caller's own Bernoulli(1 - p) draw, uint8 [N, C]; or seed (a device counter) lets the
 * kernel draw it; keep_scale = 1 / (1 - p) either way.  C % 4 == 0.  The mask the backward
 * needs is out > 0 (times keep_scale): nothing else has to be saved (sngnn_linear_forward_masked).
 * (No batch norm in between: models.py:207-208's bn stays a layer of its own.)
 */
typedef struct sngnn_epilogue {
    const float *bias;
    const unsigned char *keep;
    float keep_scale;
    int relu;
    const void *seed;       /* dev uint64 [1] or NULL (with keep == NULL): draw the keep mask in the kernel, */
    float p;                /* keep[i, c] = u(seed, i * C + c) >= p, u a counter-based uniform: the same     */
                            /* seed gives the same mask; the caller advances the seed between forwards       */
    void *kept_bits;        /* dev, sngnn_graph_kept_bits_bytes(g) bytes, or NULL: training calls - the      */
                            /* forward writes WHICH edges it kept as packed bits (its own layout) for        */
                            /* sngnn_agg_backward_bits, instead of wsel + a packing launch in the backward;  */
                            /* only where sngnn_agg_kept_bits_supported(g, top_k); an all-zero struct with   */
                            /* this field set is a plain forward that saves the bits                         */
    /* The classification head of the LAST layer inside the forward's own launches (round 4; NULL
     * head_y = off): what the wrappers do to the last conv's output - log_softmax (models.py:86,211,
     * 303) - and what the harness does to that - nll_loss on a mask and the accuracy count
     * (train.py:81-84, 98-102, 112-116).  The rows go through it in the forward's SECOND launch: the
     * split rows in their finalize, all others read back by extra workgroups of that launch, which is
     * otherwise a latency chain on an idle chip - instead of two more launches behind it
     * (sngnn_head_nll / _nll2, which compute the same per-row bits).  `bias` (the conv's) is added
     * first; relu / keep / seed must be off.  Needs sngnn_agg_head_supported.                        */
    const int64_t *head_y;          /* dev int64 [N] labels                                                      */
    const unsigned char *head_sel;  /* dev u8 [N]: head_sets == 1: != 0 marks the split's rows; == 2: bit 0 = in */
                                    /* split A, bit 1 = in split B (validation and test off ONE forward)          */
    int head_sets;                  /* 1 or 2                                                                    */
    int head_out_mode;              /* what `out` holds afterwards: 1 the logits (mean rows + bias), 2 d(mean    */
                                    /* NLL of the split) / d logits, zero rows outside it (head_sets == 1):     */
                                    /* the tensor the backward starts from                                     */
    int64_t head_n_a, head_n_b;     /* rows in split A / B: the means' denominators                              */
    float *head_metrics;            /* dev f32 [2 * head_sets]: (mean NLL, correct count) per split              */
    void *head_workspace;           /* dev, sngnn_agg_head_workspace_bytes(g) bytes                              */
    int no_filter;                  /* != 0: this call does not use the fp16 filter whatever sngnn_filter_enable */
                                    /* would decide - the caller knows its rows (nearly parallel ones all pass   */
                                    /* the threshold: nothing to prune, the filter is a pass for nothing)       */
} sngnn_epilogue_t;
int64_t sngnn_agg_head_workspace_bytes(const sngnn_graph_t *g);
/* 1 if a forward on this graph at this width / top_k can take the head epilogue (C % 4 == 0, C <= 64, the
 * split rows on the candidate finalize), else 0 */
int sngnn_agg_head_supported(const sngnn_graph_t *g, int C, int top_k);
int sngnn_agg_forward_epilogue(const sngnn_graph_t *g, const float *h, int C, int top_k, float thr,
                               const sngnn_epilogue_t *epi, float *out, float *wsel, float *inv_norm,
                               void *workspace, void *stream);
int sngnn_agg_forward_prepared_epilogue(const sngnn_graph_t *g, const float *n, const float *nrm,
                                        const void *filt, int C, int top_k, float thr,
                                        const sngnn_epilogue_t *epi, float *out, float *wsel,
                                        float *inv_norm, void *workspace, void *stream);
int sngnn_filter_enable(int mode);
/* whether sngnn_agg_forward would use filter rows for this call (a caller that prepares the
 * rows itself writes them only then) */
int sngnn_filter_wanted(const sngnn_graph_t *g, int C, int top_k, float thr);
/*
 * sngnn_agg_forward_prepared restricted to the target rows i with row_flag[i] == row_want
 * (row_flag: dev u8 [N]); the other rows of out / wsel / inv_norm are not touched.  Two calls
 * with complementary flags equal one unrestricted call, bit for bit.  One rank of a node-range
 * partition aggregates its INTERIOR rows (all sources local) while the halo rows its other
 * rows need are still in flight (sngnn_amd/dist.py).  No reference counterpart.
 */
int sngnn_agg_forward_rows(const sngnn_graph_t *g, const float *n, const float *nrm,
                           const void *filt, int C, int top_k, float thr,
                           const uint8_t *row_flag, int row_want, float *out, float *wsel,
                           float *inv_norm, void *workspace, void *stream);
/* measurement aid: knob 0 = row classes the main forward kernel runs (bit 0 split-row tasks,
 * 1 wave rows, 2 small rows; default 7 - anything else leaves the output incomplete);
 * knob 2 = how sngnn_agg_forward scores: 0 (default) = on the fly from h when nothing is selected
 * (top_k < 0) and through the normalisation pass + unit-row table otherwise; 1 = table always;
 * 2 = on the fly always (fast cosine, exact normalise-then-dot wherever a decision is in doubt:
 * the same selections, bit for bit);
 * knob 3 = sngnn_agg_backward: 0 (default) = node-centric (a node small both as target and as
 * source does both passes in one work item; with a top_k hint every node) on graphs where such
 * nodes are at least half of the owned nodes, the two passes otherwise; 1 = the two passes for
 * every node (same bits without the hint; equal to rounding on split rows with it); 2 =
 * node-centric whatever the graph;
 * knob 4 = which items the hinted node-centric backward runs (bit 0 wave-per-node, bit 1 fused;
 * default 3 - anything else leaves grad_h incomplete: timing only);
 * knob 5 = sngnn_linear_forward*, sngnn_cosine_dense, sngnn_knn_graph: 0 (default) = products on the bf16 matrix
 * cores after an exact three-way split of both operands (fp32 accumulation, an fp32 contraction's
 * rounding), 1 = fp32 MFMAs;
 * knobs 6, 7, 8 = the kNN builder's route, the dense cosine's contraction split, the in-degree below which a wave row
 * skips the fp16 filter (csrc/knn.hip, toolbox.hip, agg_fwd.hip);
 * knob 9 = where the split rows (in-degree > 128) of sngnn_agg_forward* are finalized: 1 (default) = inside the main
 * launch - by its last workgroups, each row as soon as its tasks have published their candidates - when that
 * chain of round trips fits under the launch's own time (arxiv-sized graphs and up; calls that rank from
 * candidates, no head epilogue), else in a launch of its own; 0 = always a launch of its own; v > 1 = inside the
 * main launch on v workgroups.  Same selections and kept weights either way; the rows' sums agree bit for bit for
 * rows of at most 128 candidates and to rounding (another summation order) for the bigger ones. */
int sngnn_tuning_set(int which, int value);
/* how many workgroups of the last sngnn_agg_forward* launch on this process finalized split rows (0 = the
 * finalize was a launch of its own, or there was nothing to finalize) - measurement aid, see knob 9 */
int sngnn_last_forward_finalize_workgroups(void);
/* test aid: out[p] = the filter pass's approximate cosine of nodes pair_a[p], pair_b[p] (dev i64) */
int sngnn_filter_pair_scores(const void *filt, int C, const int64_t *pair_a, const int64_t *pair_b,
                             int64_t n_pairs, float *out, void *stream);
int sngnn_agg_forward_normalized(const sngnn_graph_t *g, const float *n, const float *nrm,
                                 int C, int top_k, float thr, float *out, float *wsel,
                                 float *inv_norm, int32_t *sel_src, float *sel_w,
                                 void *workspace, void *stream);

/*
 * Replaces: autograd through the op list above (triggered at train.py:86).
 * Deterministic: no floating-point atomics; every sum has a fixed order.
 *
 *   grad_out dev f32 [N, C]        dL/d out
 *   wsel                           as written by sngnn_agg_forward on the same h (only WHICH
 *                             edges were kept is read from it; the cosines are recomputed from
 *                             the rows the passes gather anyway)
 *   grad_h   dev f32 [N_total, C]  dL/d h through all three routes (message
 *                             value, norm_i, norm_j) and F.normalize's Jacobian.
 *                             For a partition this is the rank's PARTIAL gradient
 *                             (sum over ranks = reduce-scatter gives the total).
 */
int sngnn_agg_backward(const sngnn_graph_t *g, const float *h, int C,
                       const float *grad_out, const float *wsel, float *grad_h,
                       void *workspace, void *stream);
/* The same with the forward's top_k passed along (<= 0: unknown, == sngnn_agg_backward).  A
 * top_k in [1, 128] promises at most that many kept in-edges per row, which lets one wave take a
 * whole target row of any in-degree (kept edges found by scanning the packed kept bits): one
 * launch for all nodes, no split-row partial sums.  The result does not depend on the promise
 * being true (an over-full row is processed in pieces), only the speed does. */
int sngnn_agg_backward_topk(const sngnn_graph_t *g, const float *h, int C,
                            const float *grad_out, const float *wsel, int top_k, float *grad_h,
                            void *workspace, void *stream);
/* The same from the kept bits a forward wrote itself (sngnn_epilogue_t.kept_bits): no packing
 * launch in front.  sngnn_agg_kept_bits_supported: whole graph, 1 <= top_k <= 16, a graph on which
 * the backward is the node-centric one (most nodes small as target and as source) and whose
 * biggest row's candidates fit the finalize at this C.  The gradient
 * equals sngnn_agg_backward_topk's on the same forward, bit for bit. */
int sngnn_agg_kept_bits_supported(const sngnn_graph_t *g, int C, int top_k);
int64_t sngnn_graph_kept_bits_bytes(const sngnn_graph_t *g);
int sngnn_agg_backward_bits(const sngnn_graph_t *g, const float *h, int C, const float *grad_out,
                            const void *kept_bits, int top_k, float *grad_h, void *workspace, void *stream);

/*
 * Cosine-attention mode of the same gather skeleton.
 * Replaces: AGNNConv.forward after ``lin`` + message + aggr='add'
 * (models.py:396-405): alpha_e = softmax over the in-edges of target i of
 * cos(h_i, h_j), out_i = sum_e alpha_e * h_j.  Build the graph with
 * add_loops = 1, remove_loops = SNGNN_LOOPS_REPLACE for AGNNConv's edge list.
 * A cosine lies in [-1, 1], so exp() is applied without the running maximum PyG's
 * softmax subtracts (identical up to rounding; the 1e-16 added to a sum >= e^-1 is
 * below fp32 resolution).
 *
 *   out    dev f32 [N, C]
 *   alpha  dev f32 [E'] or NULL  attention coefficients in CSR order (what
 *                                sngnn_attn_backward needs)
 */
int sngnn_attn_forward(const sngnn_graph_t *g, const float *h, int C, float *out,
                       float *alpha, void *workspace, void *stream);
/* autograd of the lines above; same conventions as sngnn_agg_backward */
int sngnn_attn_backward(const sngnn_graph_t *g, const float *h, int C,
                        const float *grad_out, const float *alpha, float *grad_h,
                        void *workspace, void *stream);

/*
 * Signed cosine attention of the same gather skeleton.
 * Replaces: GGCNlayer_SP.forward's use_sign branch after ``fcn`` (models.py:1512-1519 get_sparse_att,
 * :1529-1537 the two sparse products): with s_e = cos(Wh_i, Wh_j) for every entry e = (i, j),
 * i != j, of the adjacency and a_e = adj_e * softplus(deg_coeff_0 * degree_e + deg_coeff_1),
 *     out_i = sum_e a_e * (c_pos * relu(s_e) - c_neg * relu(-s_e)) * Wh_j
 *           = c_pos * prop_pos_i + c_neg * prop_neg_i            (coeff_0 / coeff_1 of :1541)
 * - the two propagations share every gathered row, so they are one gather.  Build the graph from
 * edge_index = [adj column; adj row] (source = column, target = row) with remove_loops = 1
 * (adj_remove_diag); per-edge arrays are in the graph's CSR order (array "eid" maps a CSR position
 * to its position in the loop-free entry list).
 *
 *   wh     dev f32 [N_total, C]
 *   coef   dev f32 [E']   a_e, CSR order
 *   c2     dev f32 [2]    c_pos, c_neg (device scalars: they are functions of a parameter)
 *   out    dev f32 [N, C]
 *   s      dev f32 [E'] or NULL   the cosines, CSR order (what sngnn_signed_backward needs)
 */
int sngnn_signed_forward(const sngnn_graph_t *g, const float *wh, int C, const float *coef,
                         const float *c2, float *out, float *s, void *workspace, void *stream);
/*
 * autograd of the lines above: grad_wh [N_total, C] through the message values, both rows of
 * every cosine and the normalisation's Jacobian (no floating-point atomics: two gather passes in
 * fixed order, like sngnn_attn_backward), and u [E'] (CSR order) = s_e * <grad_out_i, Wh_j>, from
 * which the caller finishes the small gradients: d a_e = kappa_e u_e with kappa_e = c_pos (s_e > 0)
 * | c_neg (s_e < 0) | 0, d c_pos = sum_{s_e > 0} a_e u_e, d c_neg = sum_{s_e < 0} a_e u_e.
 */
int sngnn_signed_backward(const sngnn_graph_t *g, const float *wh, int C, const float *grad_out,
                          const float *coef, const float *s, const float *c2, float *grad_wh, float *u,
                          void *workspace, void *stream);

/*
 * Replaces: the SNGNN++ blend  out = beta * out_0 + (1 - beta) * out_1  (models.py:134)
 * and its autograd, one pass over the n = N * C elements each way instead of five
 * elementwise launches.  beta: dev f32 [1] (the layer's Parameter).  Backward writes
 * grad0 = beta * grad_out, grad1 = (1 - beta) * grad_out and grad_beta [1] =
 * sum grad_out * (out_0 - out_1) (fixed-order sum).  workspace:
 * sngnn_blend_workspace_bytes().
 */
int64_t sngnn_blend_workspace_bytes(void);
int sngnn_blend_forward(const float *out0, const float *out1, const float *beta, int64_t n,
                        float *out, void *stream);
int sngnn_blend_backward(const float *grad_out, const float *out0, const float *out1,
                         const float *beta, int64_t n, float *grad0, float *grad1,
                         float *grad_beta, void *workspace, void *stream);
/* The blend with a hidden layer's relu + dropout behind it (models.py:81-84) in the same store:
 * sngnn_epilogue_t with relu and / or a seeded dropout (bias, keep and kept_bits must be NULL;
 * the keep draw is the aggregation epilogue's: seed, flat element index); and its backward from
 * the activated output `act`: d = act > 0 ? grad_out * scale : 0, then as above. */
int sngnn_blend_forward_epilogue(const float *out0, const float *out1, const float *beta, int64_t n,
                                 const sngnn_epilogue_t *epi, float *out, void *stream);
int sngnn_blend_backward_epilogue(const float *grad_out, const float *out0, const float *out1,
                                  const float *beta, int64_t n, const float *act, float scale,
                                  float *grad0, float *grad1, float *grad_beta, void *workspace,
                                  void *stream);

/*
 * The same two gather-sums on any graph, node-range partitions included (multi-GPU
 * SNGNN++; new - the reference is single-device).  A rank builds the partition of the
 * FLIPPED edge list (row 0 and row 1 of edge_index swapped), whose owned "targets" are
 * its own source nodes; then
 *   out0_local = sngnn_gather_sum_rows(g_flipped, wt_full, bias)      out[i] = bias + sum_{e in CSR row i} table[col_e]
 *   dwt_partial = sngnn_scatter_sum_rows(g_flipped, g0_local)         out[v] = sum_{q in CSC row v} vals[dst_q]
 * table / out of scatter have N_total rows, out of gather / vals have the owned rows.
 * Requires src_min == 0 (models.py:125's shift would cross partitions otherwise).
 */
int sngnn_gather_sum_rows(const sngnn_graph_t *g, const float *table, const float *bias,
                          int C, float *out, void *workspace, void *stream);
int sngnn_scatter_sum_rows(const sngnn_graph_t *g, const float *vals, int C, float *out,
                           void *workspace, void *stream);
/*
 * Replaces: GGCNlayer_SP's plain propagation `torch.sparse.mm(adj * sc, Wh)` (models.py:1544-1549,
 * use_sign=False) and its autograd - the same two gather-sums with one WEIGHT per entry:
 *   out[i]  = sum_{q in CSR row i} w_csr[q] * table[col_q]          (weights in the graph's CSR order)
 *   out[j]  = sum_{q in CSC row j} w_csc[q] * vals[dst_q]           (the transpose; weights in CSC order)
 * (value * row rounded, then added: a sparse mm's arithmetic; fixed order, no atomics), and the weights'
 * gradient out[p] = <a_rows[idx_a[p]], b_rows[idx_b[p]]> per entry (idx_*: dev i32 [n_pairs]).
 */
int sngnn_weighted_gather_sum_rows(const sngnn_graph_t *g, const float *table, const float *w_csr, int C,
                                   float *out, void *workspace, void *stream);
int sngnn_weighted_scatter_sum_rows(const sngnn_graph_t *g, const float *vals, const float *w_csc, int C,
                                    float *out, void *workspace, void *stream);
int sngnn_pair_dot_rows(const float *a_rows, const int32_t *idx_a, const float *b_rows, const int32_t *idx_b,
                        int64_t n_pairs, int C, float *out, void *stream);

/*
 * Measurement aid (no reference counterpart): while enabled, sngnn_agg_forward
 * records HIP events on the caller's stream around its launches;
 * sngnn_profile_last_forward waits for the last call and returns the device time (ms)
 * of the normalisation pass, of the main kernel and of what follows it on the caller's
 * stream (the split-row finalize launches), and of an EMPTY interval between two events -
 * what an event pair itself adds to each of the three figures on this stack.
 * sngnn_profile_enable(reps) with reps > 1: every launch of the forward is issued reps times
 * back to back between its two events (the launches are idempotent) and the three figures are
 * the intervals divided by reps - the average launch duration with the event pair's own cost
 * spread over reps launches.
 */
int sngnn_profile_enable(int on);
int sngnn_profile_last_forward(float *norm_ms, float *main_ms, float *fin_ms, float *empty_ms);
/*
 * Measurement aid (no reference counterpart): the memory work of the aggregation forward
 * (models.py:239,244-263: the gathers of PyG's propagate + the scatter-mean's output) with NONE of
 * its arithmetic, over the caller's own graph and feature table - the floor bench.py prints beside
 * the main kernel's time (roofline.gather_floor_ms / kernel_over_floor).
 *   mode 0: one coalesced 4C-byte row read per entry of the graph's own column list (CSR order);
 *   mode 1: the same + per owned node its own row read and an output row written to `out`
 *           (dev f32 [N, C]; NULL in mode 0) - every byte of SURVEY.md 8d's B_fwd.
 * table: dev f32 [N_total, C], C % 4 == 0, C <= 256; workspace: sngnn_gather_floor_workspace_bytes().
 */
int64_t sngnn_gather_floor_workspace_bytes(void);
int sngnn_gather_floor(const sngnn_graph_t *g, const float *table, int C, int mode, float *out,
                       void *workspace, void *stream);

/* ------------------------------------------------------------------------
 * SNGNN++ adjacency-linear branch and blend.
 * ------------------------------------------------------------------------ */
/*
 * Replaces: models.py:124-130 - SparseTensor(row - row.min(), col)
 * .to_torch_sparse_coo_tensor() followed by Linear(num_nodes, C) on it:
 *   out0[i] = b + sum over edges e with src_e - src_min == i of W[:, dst_e].
 *   wt   dev f32 [N, C]  the TRANSPOSE of the reference's w.weight ([C, N]);
 *                        row d is W[:, d] so every gather is one contiguous row
 *   bias dev f32 [C] or NULL
 *   workspace: sngnn_graph_workspace_bytes(g, C) bytes, as for the aggregation
 */
int sngnn_adj_linear_forward(const sngnn_graph_t *g, const float *wt,
                             const float *bias, int C, float *out0,
                             void *workspace, void *stream);
/*
 * Gradient of the above w.r.t. wt:  dwt[d] = sum over in-edges e of d of
 * g0[src_e - src_min]  (dense [N, C], like the reference's dense w.weight.grad).
 */
int sngnn_adj_linear_backward(const sngnn_graph_t *g, const float *g0, int C,
                              float *dwt, void *workspace, void *stream);

/* ------------------------------------------------------------------------
 * The callers on either side of the aggregation inside one training epoch
 * (SURVEY.md 8f rank 1): classification head and self.lin's weight gradient.
 * ------------------------------------------------------------------------ */
/*
 * Replaces: F.log_softmax (models.py:86,211,303) + F.nll_loss on the masked rows +
 * the accuracy count (train.py:81-84, 98-102, 112-116) by one pass over the logits.
 *   logits   dev f32 [N, C]     the last conv's output (before log_softmax)
 *   y        dev i64 [N]        labels
 *   row_mask dev u8  [N]        1 for the rows of the split (train/val/test mask)
 *   n_masked                    number of ones in row_mask (the mean's denominator)
 *   grad_logits dev f32 [N, C] or NULL: d loss / d logits (zero rows outside the mask)
 *   loss_and_correct dev f32 [2]: mean NLL over the masked rows, number of correct rows
 *   workspace: sngnn_head_workspace_bytes(N) bytes.  Deterministic (fixed-order sums).
 */
int64_t sngnn_head_workspace_bytes(int64_t N);
int sngnn_head_nll(const float *logits, const int64_t *y, const unsigned char *row_mask,
                   int64_t N, int C, int64_t n_masked, float *grad_logits,
                   float *loss_and_correct, void *workspace, void *stream);
/*
 * The same for TWO splits read off one forward: validate_step and test_step (train.py:92-117)
 * run the same eval-mode forward and differ in their mask.  row_sets dev u8 [N]: bit 0 = the row
 * is in split A, bit 1 = in split B; out4 dev f32 [4] = {mean NLL A, correct A, mean NLL B,
 * correct B}.  C <= 64.  Same per-row arithmetic and summation order as sngnn_head_nll.
 */
int sngnn_head_nll2(const float *logits, const int64_t *y, const unsigned char *row_sets,
                    int64_t N, int C, int64_t n_a, int64_t n_b, float *out4, void *workspace,
                    void *stream);
/*
 * Replaces: SNGNN++'s last line `beta * out_0 + (1 - beta) * out_1` (models.py:134) followed by the wrapper's
 * log_softmax and the harness' nll_loss / accuracy (models.py:86; train.py:81-84, 98-102, 112-116) in ONE pass:
 * sngnn_head_nll / _nll2 on the blend of two tensors, which is formed in registers with sngnn_blend_forward's
 * rounding and never stored unless logits_out is given.  sets = 1: sel != 0 marks the split's rows, metrics[2];
 * sets = 2: bit 0 / bit 1, metrics[4].  grad_logits (sets == 1, or NULL): d (mean NLL) / d logits, zero rows
 * outside the split - the tensor sngnn_blend_backward starts from.  C % 4 == 0, C <= 64
 * (sngnn_head_nll_blend_supported); workspace: sngnn_head_workspace_bytes(N).
 */
int sngnn_head_nll_blend_supported(int C);
int sngnn_head_nll_blend(const float *out0, const float *out1, const float *beta, const int64_t *y,
                         const unsigned char *sel, int64_t N, int C, int sets, int64_t n_a, int64_t n_b,
                         float *grad_logits, float *logits_out, float *metrics, void *workspace, void *stream);
/*
 * Replaces: autograd of self.lin w.r.t. its parameters (models.py:121,237,324):
 *   grad_weight [C, F] = grad_out^T [C, N] . x [N, F],  grad_bias [C] = sum_i grad_out[i]
 * (grad_bias may be NULL).  workspace: sngnn_linear_wgrad_workspace_bytes(N, C, F).
 */
/*
 * Replaces: self.lin's forward for narrow layers, h = x W^T + b with C <= 64
 * (models.py:121,237,324).  x dev f32 [N, F], weight dev f32 [C, F], bias dev f32 [C]
 * or NULL, h dev f32 [N, C].  fp32 in, fp32 out, an fp32 dot product's rounding: for F in
 * {32, 64, 128} the products run on the bf16 matrix cores after an EXACT three-way split of both
 * operands, accumulated in fp32 (knob 5 of sngnn_tuning_set: fp32 MFMAs instead); HBM-bound (x
 * read once).  An infinity or NaN in a row of x makes that output row NaN.
 */
int sngnn_linear_forward(const float *x, const float *weight, const float *bias,
                         int64_t N, int F, int C, float *h, void *stream);
/*
 * h = act > 0 ? (x W^T + b) * act_scale : 0, act dev f32 [N, C]: the input gradient g W of a layer
 * (x = g, weight = W^T) whose input `act` was the epilogue'd output of the layer before
 * (sngnn_agg_forward_epilogue): autograd's threshold_backward and dropout backward
 * (models.py:206-209) folded into the store.  Any F, C <= 64.
 */
int sngnn_linear_forward_masked(const float *x, const float *weight, const float *bias, int64_t N,
                                int F, int C, const float *act, float act_scale, float *h, void *stream);
/* The same mask as a pass of its own (for a consumer that did not fold it into its store):
 * grad_pre[q] = act[q] > 0 ? grad[q] * scale : 0 over n = N * C elements (may alias grad). */
int sngnn_epilogue_backward(const float *grad, const float *act, float scale, int64_t n, float *grad_pre,
                            void *stream);
/*
 * Replaces: self.lin followed by F.normalize (models.py:237-238, :121-122, :324-325 - adjacent
 * lines of every conv forward): h as above AND, from the same launch's epilogue, the unit rows
 * n = h / max(||h||, 1e-12), the clamped norms and (filt != NULL, C > 32) the fp16 filter rows -
 * bit for bit what sngnn_normalize_rows_filter computes from h (same summation tree, IEEE
 * square root and division), so sngnn_agg_forward_prepared can follow without a normalisation
 * pass.  C % 4 == 0, C <= 64, F in {16, 32, 64, 128} (sngnn_linear_normalized_supported).
 */
int sngnn_linear_normalized_supported(int64_t N, int F, int C);
int sngnn_linear_forward_normalized(const float *x, const float *weight, const float *bias,
                                    int64_t N, int F, int C, float *h, float *n, float *nrm,
                                    void *filt, void *stream);
int64_t sngnn_linear_wgrad_workspace_bytes(int64_t N, int C, int F);
int sngnn_linear_wgrad(const float *grad_out, const float *x, int64_t N, int C, int F,
                       float *grad_weight, float *grad_bias, void *workspace, void *stream);

/* ------------------------------------------------------------------------
 * Sim-GFA toolbox (SimGFAToolbox/dense.py).
 * ------------------------------------------------------------------------ */
/* dense.py:138-141: S = normalize(x) normalize(x)^T, dev f32 [N, N]. */
int sngnn_cosine_dense(const float *x, int64_t N, int64_t F, float *S, void *stream);
/*
 * Fused statistics of S without materialising it (dense.py:9-30, 104-130,
 * 144-149, 167-179): class_sum dev f64 [n_classes, n_classes] receives the sum
 * of S over every (class_i, class_j) block INCLUDING the diagonal, diag_sum dev
 * f64 [1] the sum of S's diagonal.  y dev int32 [N] (all zeros for the global
 * mean).  Both outputs must be zeroed by the caller.
 */
int sngnn_cosine_class_sums(const float *x, int64_t N, int64_t F, const int32_t *y,
                            int n_classes, double *class_sum, double *diag_sum,
                            void *stream);
/*
 * dense.py:152-164: per-edge cosine of raw features for an arbitrary COO edge
 * list: sim[e] = <n[a_e], n[b_e]>, a = edge_index[0], b = edge_index[1].
 */
int sngnn_edge_cosine(const float *x, int64_t N, int64_t F,
                      const int64_t *edge_index_dev, int64_t E, float *sim,
                      void *stream);
/*
 * dense.py:163 `scatter_mean(sim, edge_index[0], dim=0)` (torch_scatter) and
 * the per-node means of dense.py:85-96 / sparse.py:98-115: mean[m] = (sum of val[e] over the
 * entries with index[e] == m, added in ENTRY ORDER in fp32) / max(count, 1); an empty group
 * reads 0.  No atomics: the same additions in the same order as the CPU's serial scatter, on
 * every run.  val dev f32 [E], index dev int64 [E] with values in [0, M), mean dev f32 [M],
 * count dev int32 [M] or NULL.
 */
int sngnn_segment_mean(const float *val, const int64_t *index, int64_t E, int64_t M, float *mean,
                       int32_t *count, void *stream);
/*
 * sparse.py:8-14 without forming the product: entries (pair_a[p], pair_b[p]) of
 * M_n^T M_n, M_n = the column-normalised sparse input in CSC form (colptr int64
 * [n_cols + 1], rowidx int32 ascending inside a column, vals = the NORMALISED values).
 * What sparse.py's linked / neighbourhood statistics read out of the scipy product
 * row by row (sparse.py:45-120).  Synchronises the stream (range check of the pairs).
 */
int sngnn_sparse_pair_dot(const int64_t *colptr, const int32_t *rowidx, const float *vals,
                          int64_t n_cols, const int64_t *pair_a, const int64_t *pair_b,
                          int64_t n_pairs, float *out, void *stream);

/*
 * kNN similarity graph (the north star's "Node-Similarity build"):
 * for every node the k most cosine-similar nodes, without storing the N x N similarity
 * the reference materialises (dense.py:138-141) - tiled matrix-core products (fp32 rounding:
 * exact bf16 split or fp32 MFMAs, knob 5) with a fused per-row top-k.  Order (cosine desc, node id asc), k <= 32.
 *   nbr_idx dev i32 [N, k]   neighbour ids in rank order, -1 padded when N - 1 < k
 *   nbr_sim dev f32 [N, k]   their cosines (0 padded)
 *   exclude_self             a node is not its own neighbour
 *   workspace: sngnn_knn_workspace_bytes(N, k)
 */
int64_t sngnn_knn_workspace_bytes(int64_t N, int k);
int sngnn_knn_graph(const float *x, int64_t N, int64_t F, int k, int exclude_self,
                    int32_t *nbr_idx, float *nbr_sim, void *workspace, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SNGNN_HIP_H */
