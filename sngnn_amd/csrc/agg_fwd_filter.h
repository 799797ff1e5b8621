// Half-precision FILTER rows for the selecting forward (gfx950).
//
// A row with more than top_k in-edges keeps at most top_k of them, yet the scoring pass
// gathers the 4C-byte unit row of EVERY in-edge - at C = 40 a 160-byte row, which always
// straddles two 128-byte lines (the line rate of the memory system is the
// kernel's bound).  Two thirds of ogbn-arxiv's edges sit in rows that keep 16 of 17..13 000.
//
// The normalisation pass therefore also writes, per node, its unit row rounded to fp16 (scaled
// by 2^10 so that nothing underflows) into a table of 128-byte-aligned entries: ONE line per
// source row for C <= 64 (a power-of-two number of lines in general: the lanes' 4-channel
// vectors of the row layout (VEC 4, G, R) cover 4 G R channels).  A wave row / split-row task first scores all its edges against that
// table (approximate cosine s~, |s~ - s| <= FILT_EPS, see below), finds the top_k-th largest
// s~ = t~, and only the CANDIDATES
//        s~ >= t~ - 2 FILT_EPS   and   s~ >= thr - FILT_EPS
// are scored exactly from the fp32 unit rows.  Every edge of the exact top_k is a candidate:
// at least top_k edges have s >= t~ - FILT_EPS, and an edge below the cut has
// s <= s~ + FILT_EPS < t~ - FILT_EPS.  The exact selection (score descending, edge position
// ascending, >= thr) then runs on the candidates' exact fp32 scores, so the selected indices
// and weights are bit-identical to the unfiltered path: the filter only decides which rows
// are never fetched in full precision.  With many (near-)ties at the cut - duplicate rows,
// nearly parallel rows - the candidate list simply grows, up to the whole row.
//
// Error bound.  f = fl16(1024 a) = 1024 a (1 + d), |d| <= 2^-11 for |a| >= 2^-24 (fp16 normal
// range after the scaling; smaller components contribute at most 2^-24 each in absolute
// terms, flushed or not).  For unit rows a, b (|a|_2, |b|_2 <= 1):
//   |sum f_a f_b 2^-20 - <a, b>| <= (2^-10 + 2^-22) sum |a_c b_c| + 2 sqrt(C) 2^-24
//                                <= 9.78e-4 + 2.7e-6          (C <= 512)
// plus the fp32 accumulation error of either dot product (<= C 2^-24 each, 3.1e-5 at C = 512).
// FILT_EPS = 1.1e-3 leaves 5e-5 of slack on top of that.
#pragma once
#include "device_utils.h"

namespace sngnn {

constexpr float FILT_SCALE = 1024.0f;
constexpr float FILT_UNSCALE = 1.0f / (1024.0f * 1024.0f);
constexpr float FILT_EPS = 1.1e-3f;

// (row size: filter_row_bytes(C), common.h)

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_half2(float a, float b)
{
    const half2_t p = {(_Float16)a, (_Float16)b};     // v_cvt_pk_f16_f32, round to nearest even
    return __builtin_bit_cast(unsigned, p);
}

// <a, b> of 8 fp16 pairs, fp32 accumulation (v_dot2c_f32_f16)
__device__ __forceinline__ float fdot8(const uint4 &a, const uint4 &b)
{
    float s = 0.f;
    s = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, a.x), __builtin_bit_cast(half2_t, b.x), s, false);
    s = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, a.y), __builtin_bit_cast(half2_t, b.y), s, false);
    s = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, a.z), __builtin_bit_cast(half2_t, b.z), s, false);
    s = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, a.w), __builtin_bit_cast(half2_t, b.w), s, false);
    return s;
}

// Approximate scores of the edges [e0, e1) of a row (row-local indices) into LDS:
// sc[t - e0] = s~, ids[t - e0] = source id.  GF lanes x 16 bytes = one filter row; the wave's
// 64 / GF groups stride over the edges, UF rows in flight per group, column ids one
// iteration ahead (as in score_edges).
template <int GF>
__device__ __forceinline__ void filter_scores(const uint4 *__restrict__ filt, const int32_t *__restrict__ col,
                                              int self, int rs, int e0, int e1, float *sc, int *ids)
{
    constexpr int NGF = 64 / GF;
#ifndef SNGNN_FILT_UF
#define SNGNN_FILT_UF 4
#endif
    constexpr int UF = GF >= 32 ? 2 : (GF == 16 ? 4 : SNGNN_FILT_UF);
    const int lane = lane_id();
    const int gid = lane / GF, lf = lane % GF;
    const uint4 fi = filt[(size_t)self * GF + lf];
    int jn[UF];
#pragma unroll
    for (int u = 0; u < UF; ++u) {
        const int t = e0 + u * NGF + gid;
        jn[u] = t < e1 ? col[rs + t] : self;
    }
    for (int base = e0; base < e1; base += NGF * UF) {
        int t[UF], j[UF];
        uint4 x[UF];
#pragma unroll
        for (int u = 0; u < UF; ++u) {
            t[u] = base + u * NGF + gid;
            j[u] = jn[u];
        }
#pragma unroll
        for (int u = 0; u < UF; ++u) x[u] = filt[(size_t)j[u] * GF + lf];
#pragma unroll
        for (int u = 0; u < UF; ++u) {
            const int tn = t[u] + NGF * UF;
            jn[u] = tn < e1 ? col[rs + tn] : self;
        }
#pragma unroll
        for (int u = 0; u < UF; ++u) {
            const float s = group_sum<GF>(fdot8(fi, x[u])) * FILT_UNSCALE;
            if (t[u] < e1 && lf == 0) {
                sc[t[u] - e0] = s;
                ids[t[u] - e0] = j[u];
            }
        }
    }
}

// Which of the wave's n <= 128 approximately scored edges (two per lane: lane, lane + 64) must
// be scored exactly.
// eps: the bound on |approximate - exact| of the scores in sc[]
__device__ __forceinline__ void approx_candidates(const float *sc, int n, int k, float thr, float eps, bool &c0,
                                                  bool &c1)
{
    const int lane = lane_id();
    const int i0 = lane, i1 = lane + 64;
    const float s0 = i0 < n ? sc[i0] + 0.0f : 0.f, s1 = i1 < n ? sc[i1] + 0.0f : 0.f;
    const float lo = thr - eps;
    const bool v0 = i0 < n && s0 >= lo, v1 = i1 < n && s1 >= lo;
    const int cnt = __popcll(__ballot(v0)) + __popcll(__ballot(v1));
    if (cnt <= k) { c0 = v0; c1 = v1; return; }
    // order key of the k-th largest approximate score (ties do not matter here)
    const unsigned k0 = v0 ? f2key(s0) : 0u, k1 = v1 ? f2key(s1) : 0u;
    unsigned T = 0;
    for (int b = 31; b >= 0; --b) {
        const unsigned cand = T | (1u << b);
        const int c = __popcll(__ballot(k0 >= cand)) + __popcll(__ballot(k1 >= cand));
        if (c >= k) T = cand;
    }
    const float tk = __uint_as_float((T & 0x80000000u) ? (T & 0x7FFFFFFFu) : ~T);
    const float cut = tk - 2.0f * eps;
    c0 = v0 && s0 >= cut;
    c1 = v1 && s1 >= cut;
}

}  // namespace sngnn
