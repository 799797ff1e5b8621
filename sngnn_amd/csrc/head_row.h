// The classification head on ONE row held by a lane group (C % 4 == 0, C <= 64: G = 8 or 16 lanes,
// four channels per lane) - log_softmax (models.py:86,211,303), the row's NLL term, its arg-max
// against the label (train.py:81-84) and d loss / d logits.  One instruction sequence for the
// stand-alone head kernel (head.hip: k_head_groups) and for the head inside the aggregation's second
// launch (agg_fwd_impl.h: head_store_row): the same bits either way.
#pragma once
#include "common.h"

namespace sngnn {

template <int CTRL> __device__ __forceinline__ float dpp_maxf(float v)
{
    return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false)));
}
template <int CTRL> __device__ __forceinline__ int dpp_mini(int v)
{
    return min(v, __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false));
}
template <int CTRL> __device__ __forceinline__ float dpp_addf(float v)
{
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int G> __device__ __forceinline__ float gmaxf(float v)
{
    static_assert(G == 8 || G == 16, "rows of 8 or 16 lanes");
    v = dpp_maxf<0xB1>(v); v = dpp_maxf<0x4E>(v); v = dpp_maxf<0x141>(v);
    if constexpr (G == 16) v = dpp_maxf<0x140>(v);
    return v;
}
template <int G> __device__ __forceinline__ int gmini(int v)
{
    v = dpp_mini<0xB1>(v); v = dpp_mini<0x4E>(v); v = dpp_mini<0x141>(v);
    if constexpr (G == 16) v = dpp_mini<0x140>(v);
    return v;
}
template <int G> __device__ __forceinline__ float gsumf(float v)
{
    v = dpp_addf<0xB1>(v); v = dpp_addf<0x4E>(v); v = dpp_addf<0x141>(v);
    if constexpr (G == 16) v = dpp_addf<0x140>(v);
    return v;
}

__device__ __forceinline__ float fast_exp_neg(float t) { return __builtin_amdgcn_exp2f(t * 1.44269504088896341f); }

struct HeadRow {
    float loss, corr;       // the row's NLL term and 1 / 0 for a correct arg-max (the same in every lane of the group)
    float e0, e1, e2, e3;   // exp(z - max) of the lane's four channels (0 beyond C)
    float se;               // their sum over the row
    int k;                  // label - first channel of the lane: the lane holds the label's logit iff 0 <= k < 4
};

// t = the lane's four logits at channels c0 .. c0 + 3 (in == false: the lane is beyond C), yy the label.
// Every lane of the group must be active.
template <int G>
__device__ __forceinline__ HeadRow head_row(const float4 t, bool in, int c0, int yy)
{
    HeadRow r;
    const float v0 = in ? t.x : -INFINITY, v1 = in ? t.y : -INFINITY;
    const float v2 = in ? t.z : -INFINITY, v3 = in ? t.w : -INFINITY;
    const float mx = gmaxf<G>(fmaxf(fmaxf(v0, v1), fmaxf(v2, v3)));
    // first channel that attains the maximum (torch.max on the CPU keeps the first)
    int a = 1 << 30;
    if (v3 == mx) a = c0 + 3;
    if (v2 == mx) a = c0 + 2;
    if (v1 == mx) a = c0 + 1;
    if (v0 == mx) a = c0;
    const int arg = gmini<G>(a);
    // exp(t) for t <= 0 as 2^(t log2 e): v_exp_f32 (1 ulp) on a product rounded once - a
    // relative error of at most |t| 2^-24 + 2^-23 per term, i.e. below 3e-6 even for the
    // terms 40 below the maximum (which weigh e^-40); the library expf spends ~15
    // instructions per value on the last bit and made the head kernel VALU-bound
    r.e0 = in ? fast_exp_neg(v0 - mx) : 0.f; r.e1 = in ? fast_exp_neg(v1 - mx) : 0.f;
    r.e2 = in ? fast_exp_neg(v2 - mx) : 0.f; r.e3 = in ? fast_exp_neg(v3 - mx) : 0.f;
    r.se = gsumf<G>((r.e0 + r.e1) + (r.e2 + r.e3));
    r.k = yy - c0;                                    // the label's logit sits in exactly one lane
    const float mine = (in && r.k >= 0 && r.k < 4) ? (r.k == 0 ? v0 : r.k == 1 ? v1 : r.k == 2 ? v2 : v3) : 0.f;
    const float zy = gsumf<G>(mine);
    // ln(se), se in [1, C]: v_log_f32 (1 ulp of log2) times ln 2 - an absolute error below 4e-7 on a term
    // of size ln C; the library logf spends ~25 instructions on denormals and the last bit (this
    // arithmetic is issue-bound: ~130 vector instructions per row)
    r.loss = -(zy - mx - __builtin_amdgcn_logf(r.se) * 0.6931471805599453f);
    r.corr = (arg == yy) ? 1.f : 0.f;
    return r;
}

// (softmax - onehot) * scale of the lane's four channels
__device__ __forceinline__ float4 head_row_grad(const HeadRow &r, float scale)
{
    const float inv = scale / r.se;
    return make_float4(r.e0 * inv - (r.k == 0 ? scale : 0.f), r.e1 * inv - (r.k == 1 ? scale : 0.f),
                       r.e2 * inv - (r.k == 2 ? scale : 0.f), r.e3 * inv - (r.k == 3 ? scale : 0.f));
}

// fixed-order sum of partial entries into the metrics (head.hip; entries of `stride` floats:
// loss A, correct A [, loss B, correct B]; out[2 * sets])
int launch_head_reduce(const float *part, int entries, float scale_a, float scale_b, int sets, int stride,
                       float *out, hipStream_t st);

}  // namespace sngnn
