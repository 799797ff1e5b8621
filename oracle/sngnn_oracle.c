/*
 * CPU oracle (plain C) for SNGNN's similarity-navigated aggregation path.
 *
 * TEST INFRASTRUCTURE ONLY: linked/loaded by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg, never by the product (sngnn_amd/).
 *
 * PARITY UNPINNED at the third-party boundary (see oracle/sngnn_oracle.py's
 * header): the reference ships no golden vectors and its third-party
 * dependencies (torch-geometric 2.0.4, torch-scatter 2.0.9, torch-sparse
 * 0.6.13 - requirements.txt:67-69) are absent, so this file restates
 *   - the reference's own op sequence (models/models.py:116-158, 233-263,
 *     322-334) and
 *   - the published CPU algorithms of the third-party calls at those call
 *     sites, as literal serial loops,
 * and is pinned by the hand-derived KATs of SURVEY.md Appendix B plus
 * agreement with the independent core-torch restatement.
 *
 * All loops are serial and in edge order, like the reference's CPU path
 * (torch_scatter's CPU kernels are single serial loops over the edges).
 * fp32 arithmetic throughout; compile WITHOUT -ffast-math and without FMA
 * contraction (-ffp-contract=off) so that a*b+c is two roundings as in ATen.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* PyG add_self_loops + remove_self_loops (models.py:117-120, 234-236, 323;
 * Appendix A-1/A-2).  out_ei must hold 2*(E+N) int64; returns E'. */
int64_t sno_edge_list(const int64_t *ei, int64_t E, int64_t N, int add_loops,
                      int remove_loops, int64_t *out_src, int64_t *out_dst)
{
    int64_t n = 0;
    for (int64_t e = 0; e < E; ++e) {
        int64_t s = ei[e], d = ei[E + e];
        if (remove_loops && s == d) continue;
        out_src[n] = s; out_dst[n] = d; ++n;
    }
    if (add_loops && !remove_loops)
        for (int64_t v = 0; v < N; ++v) { out_src[n] = v; out_dst[n] = v; ++n; }
    return n;
}

/* F.normalize(x, p=2, dim=-1): x / max(||x||_2, 1e-12)  (models.py:122,238,325;
 * Appendix A-7). */
void sno_normalize(const float *h, int64_t N, int64_t C, float *norm)
{
    for (int64_t i = 0; i < N; ++i) {
        float ss = 0.f;
        for (int64_t c = 0; c < C; ++c) ss = ss + h[i * C + c] * h[i * C + c];
        float d = sqrtf(ss);
        if (d < 1e-12f) d = 1e-12f;
        for (int64_t c = 0; c < C; ++c) norm[i * C + c] = h[i * C + c] / d;
    }
}

/* (norm_i * norm_j).sum(-1)  (models.py:140,245,332). */
void sno_edge_cosine(const float *norm, int64_t C, const int64_t *src,
                     const int64_t *dst, int64_t E, float *s)
{
    for (int64_t e = 0; e < E; ++e) {
        const float *a = norm + dst[e] * C, *b = norm + src[e] * C;
        float acc = 0.f;
        for (int64_t c = 0; c < C; ++c) acc = acc + a[c] * b[c];
        s[e] = acc;
    }
}

/* torch-scatter 2.0.9 CPU scatter_max(src, index, dim=0), Appendix A-5:
 * strict '>' in one forward serial loop, arg initialised to E, untouched
 * outputs zeroed afterwards. */
void sno_scatter_max(const float *src, const int64_t *index, int64_t E,
                     int64_t M, float *out, int64_t *arg)
{
    for (int64_t i = 0; i < M; ++i) { out[i] = -FLT_MAX; arg[i] = E; }
    for (int64_t e = 0; e < E; ++e) {
        int64_t i = index[e];
        if (src[e] > out[i]) { out[i] = src[e]; arg[i] = e; }
    }
    for (int64_t i = 0; i < M; ++i) if (arg[i] == E) out[i] = 0.f;
}

/* models.py:141-156 / 246-261: top_k rounds of scatter_max, -2 for empty
 * groups, fp32 '>= thr', -1.1 knock-out; weight[e] = s[e] for chosen edges.
 * sel_pos (optional, [M*top_k], -1 padded) receives the chosen edge positions
 * per target in rank order (first selection of an edge only). */
void sno_topk_weights(const float *s, const int64_t *index, int64_t E, int64_t M,
                      int top_k, double thr_d, float *weight, int64_t *sel_pos)
{
    const float thr = (float)thr_d;     /* torch compares in fp32 */
    float *tmp = (float *)malloc(sizeof(float) * (size_t)(E ? E : 1));
    float *mx = (float *)malloc(sizeof(float) * (size_t)(M ? M : 1));
    int64_t *arg = (int64_t *)malloc(sizeof(int64_t) * (size_t)(M ? M : 1));
    int64_t *fill = (int64_t *)calloc((size_t)(M ? M : 1), sizeof(int64_t));
    unsigned char *seen = (unsigned char *)calloc((size_t)(E ? E : 1), 1);
    memcpy(tmp, s, sizeof(float) * (size_t)E);
    for (int64_t e = 0; e < E; ++e) weight[e] = 0.f;
    if (sel_pos) for (int64_t i = 0; i < M * top_k; ++i) sel_pos[i] = -1;
    for (int r = 0; r < top_k; ++r) {
        sno_scatter_max(tmp, index, E, M, mx, arg);
        for (int64_t i = 0; i < M; ++i) {
            float v = (arg[i] == E) ? -2.f : mx[i];
            if (v >= thr) {
                int64_t e = arg[i];     /* (arg==E with thr<=-2 would be an
                                           out-of-range scatter in the reference) */
                if (e == E) continue;
                tmp[e] = -1.1f;
                weight[e] = s[e];
                if (sel_pos && !seen[e]) { sel_pos[i * top_k + fill[i]++] = e; seen[e] = 1; }
            }
        }
    }
    free(tmp); free(mx); free(arg); free(fill); free(seen);
}

/* torch-scatter scatter(..., reduce='mean') on weight[e] * x[src[e]]
 * (models.py:157,262,333 + Appendix A-4): scatter_add in edge order, count of
 * ALL edges, clamp(min=1), divide. */
void sno_scatter_mean(const float *x, int64_t N, int64_t C, const int64_t *src,
                      const int64_t *dst, const float *weight, int64_t E, float *out)
{
    float *cnt = (float *)calloc((size_t)(N ? N : 1), sizeof(float));
    memset(out, 0, sizeof(float) * (size_t)(N * C));
    for (int64_t e = 0; e < E; ++e) {
        const float *xj = x + src[e] * C;
        float *o = out + dst[e] * C;
        float w = weight[e];
        for (int64_t c = 0; c < C; ++c) { float m = w * xj[c]; o[c] = o[c] + m; }
        cnt[dst[e]] = cnt[dst[e]] + 1.f;
    }
    for (int64_t i = 0; i < N; ++i) {
        float d = cnt[i] < 1.f ? 1.f : cnt[i];
        for (int64_t c = 0; c < C; ++c) out[i * C + c] = out[i * C + c] / d;
    }
    free(cnt);
}

/* Whole operator after self.lin: returns E'.  top_k < 0 selects SNConv (no
 * selection).  Optional outputs may be NULL: s_out/weight_out [E+N],
 * sel_src [N*top_k] (source ids in rank order, -1 padded), ei_out [2*(E+N)]. */
int64_t sno_aggregate(const float *h, int64_t N, int64_t C, const int64_t *ei,
                      int64_t E, int add_loops, int remove_loops, int top_k,
                      double thr, float *out, float *s_out, float *weight_out,
                      int64_t *sel_src, int64_t *ei_out)
{
    int64_t cap = E + N + 1;
    int64_t *src = (int64_t *)malloc(sizeof(int64_t) * (size_t)cap);
    int64_t *dst = (int64_t *)malloc(sizeof(int64_t) * (size_t)cap);
    int64_t Ep = sno_edge_list(ei, E, N, add_loops, remove_loops, src, dst);
    float *norm = (float *)malloc(sizeof(float) * (size_t)(N * C + 1));
    float *s = (float *)malloc(sizeof(float) * (size_t)cap);
    float *w = (float *)malloc(sizeof(float) * (size_t)cap);
    sno_normalize(h, N, C, norm);
    sno_edge_cosine(norm, C, src, dst, Ep, s);
    if (top_k < 0) {
        memcpy(w, s, sizeof(float) * (size_t)Ep);
    } else {
        int64_t M = 0;
        for (int64_t e = 0; e < Ep; ++e) if (dst[e] + 1 > M) M = dst[e] + 1;
        int64_t *sel_pos = NULL;
        if (sel_src) {
            sel_pos = (int64_t *)malloc(sizeof(int64_t) * (size_t)(N * top_k + 1));
        }
        sno_topk_weights(s, dst, Ep, M, top_k, thr, w, sel_pos);
        if (sel_src) {
            for (int64_t i = 0; i < N * top_k; ++i) sel_src[i] = -1;
            for (int64_t i = 0; i < M * top_k; ++i)
                if (sel_pos[i] >= 0) sel_src[i] = src[sel_pos[i]];
            free(sel_pos);
        }
    }
    sno_scatter_mean(h, N, C, src, dst, w, Ep, out);
    if (s_out) memcpy(s_out, s, sizeof(float) * (size_t)Ep);
    if (weight_out) memcpy(weight_out, w, sizeof(float) * (size_t)Ep);
    if (ei_out) {
        memcpy(ei_out, src, sizeof(int64_t) * (size_t)Ep);
        memcpy(ei_out + Ep, dst, sizeof(int64_t) * (size_t)Ep);
    }
    free(src); free(dst); free(norm); free(s); free(w);
    return Ep;
}

/* SNConv_plus_plus adjacency branch (models.py:124-130; Appendix A-6):
 * out_0 = Linear(N, C)(A), A[src - min(src), dst] += 1 for every edge of the
 * post-self-loop list; entries are visited sorted by (row, col) as the COO
 * tensor stores them.  W is [C, N] row-major, b [C]. */
void sno_adj_linear(const float *W, const float *b, int64_t N, int64_t C,
                    const int64_t *src, const int64_t *dst, int64_t E, float *out0)
{
    int64_t mn = E ? src[0] : 0;
    for (int64_t e = 1; e < E; ++e) if (src[e] < mn) mn = src[e];
    /* counting sort by (row, col): stable sort by col then by row */
    int64_t *p1 = (int64_t *)malloc(sizeof(int64_t) * (size_t)(E + 1));
    int64_t *p2 = (int64_t *)malloc(sizeof(int64_t) * (size_t)(E + 1));
    int64_t *cnt = (int64_t *)calloc((size_t)(N + 1), sizeof(int64_t));
    for (int64_t e = 0; e < E; ++e) cnt[dst[e] + 1]++;
    for (int64_t i = 0; i < N; ++i) cnt[i + 1] += cnt[i];
    for (int64_t e = 0; e < E; ++e) p1[cnt[dst[e]]++] = e;
    memset(cnt, 0, sizeof(int64_t) * (size_t)(N + 1));
    for (int64_t e = 0; e < E; ++e) cnt[src[e] - mn + 1]++;
    for (int64_t i = 0; i < N; ++i) cnt[i + 1] += cnt[i];
    for (int64_t k = 0; k < E; ++k) { int64_t e = p1[k]; p2[cnt[src[e] - mn]++] = e; }
    /* sparse addmm: r = bias (broadcast), then r[row] += 1 * W^T[col] per nnz */
    for (int64_t i = 0; i < N; ++i)
        for (int64_t c = 0; c < C; ++c) out0[i * C + c] = b[c];
    for (int64_t k = 0; k < E; ++k) {
        int64_t e = p2[k], r = src[e] - mn, d = dst[e];
        for (int64_t c = 0; c < C; ++c) out0[r * C + c] = out0[r * C + c] + W[c * N + d];
    }
    free(p1); free(p2); free(cnt);
}

/* SimGFAToolbox/dense.py:138-141: S = n n^T (N x N, row-major). */
void sno_cosine_dense(const float *x, int64_t N, int64_t F, float *S)
{
    float *n = (float *)malloc(sizeof(float) * (size_t)(N * F + 1));
    sno_normalize(x, N, F, n);
    for (int64_t i = 0; i < N; ++i)
        for (int64_t j = 0; j < N; ++j) {
            float acc = 0.f;
            for (int64_t c = 0; c < F; ++c) acc = acc + n[i * F + c] * n[j * F + c];
            S[i * N + j] = acc;
        }
    free(n);
}
