// Device-side helpers shared by the aggregation kernels (gfx950, wave = 64).
#pragma once
#include "common.h"

namespace sngnn {

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// v + (v of the lane selected by a DPP control): one VALU instruction, no LDS.
template <int CTRL> __device__ __forceinline__ float dpp_add(float v)
{
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false);
    return v + __int_as_float(moved);
}

// Sum over the G lanes of a group; every lane of the group ends with the same
// bits (each step adds a pair (a, b) as a + b in one lane and b + a in the other).
// Within a 16-lane row the partner lanes come from DPP: quad_perm [1,0,3,2],
// quad_perm [2,3,0,1], row_half_mirror (i <-> 7-i: the other quad, whose lanes
// already agree), row_mirror (i <-> 15-i: the other half).  Wider groups finish
// with ds_swizzle / bpermute.  The order is fixed and identical in every kernel.
template <int G> __device__ __forceinline__ float group_sum(float v)
{
    static_assert(G == 4 || G == 8 || G == 16 || G == 32 || G == 64, "group width");
    v = dpp_add<0xB1>(v);      // xor 1
    v = dpp_add<0x4E>(v);      // xor 2
    if constexpr (G >= 8) v = dpp_add<0x141>(v);      // other quad of the 8
    if constexpr (G >= 16) v = dpp_add<0x140>(v);     // other half of the 16
    if constexpr (G >= 32) v += __shfl_xor(v, 16, 64);
    if constexpr (G >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}

// Sum of the per-group values over the 64/G groups of a wave (same lane offset).
template <int G> __device__ __forceinline__ float cross_group_sum(float v)
{
#pragma unroll
    for (int m = G; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}

__device__ __forceinline__ int wave_max_i(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m, 64));
    return v;
}

__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// Orders LDS traffic between the lanes of ONE wave (DS ops of a wave execute in
// issue order; the fences stop the compiler from moving LDS accesses across).  The
// fences name the LDS address space only: an all-address-space fence would also drain
// the wave's outstanding global loads and stores (s_waitcnt vmcnt(0)) at every call.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Stores and loads other workgroups of the SAME launch see (agent scope: written through / read past the XCD's
// own L2, which is not coherent with the other seven).  Relaxed: the caller orders them (st: s_waitcnt vmcnt(0)
// in front of the word that publishes them; ld: issued behind the load that saw that word).
template <typename T> __device__ __forceinline__ void st_agent(T *p, T v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T> __device__ __forceinline__ T ld_agent(const T *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// VEC consecutive floats at agent scope in ONE store instruction (the compiler's own atomic stores are one dword
// each: four address computations where a row's store has one - six spilled registers in the task role)
template <int VEC> __device__ __forceinline__ void st_agent_vec(float *p, const float (&x)[VEC])
{
    if constexpr (VEC == 4) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 v = {x[0], x[1], x[2], x[3]};
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    } else if constexpr (VEC == 2) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 v = {x[0], x[1]};
        asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    } else {
        asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(x[0]) : "memory");
    }
}
// every vector memory operation this wave has issued is complete at the scope it named
__device__ __forceinline__ void wave_vmem_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// number of set bits of `mask` below this lane
__device__ __forceinline__ int prefix_popc(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// fp32 -> uint32 whose unsigned order equals the float order (-0.0 must have
// been canonicalised to +0.0 by the caller: the reference compares floats).
__device__ __forceinline__ unsigned f2key(float s)
{
    unsigned u = __float_as_uint(s);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Selection key: larger == better.  (cosine descending, position ascending.)
__device__ __forceinline__ unsigned long long sel_key(float s, unsigned idx)
{
    return ((unsigned long long)f2key(s) << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
}

// ---------------------------------------------------------------------------
// Wave-level top-k of selection keys (sel_key above; 0 = no key), NK per lane: bitwise search for
// a threshold that leaves exactly the k largest.  Keys are unique, so "the k largest" is well
// defined.  lowbits = bits of the largest index (the low word of a key is 0xFFFFFFFF - index, so
// its upper 32 - lowbits bits are all ones and need no search).
// ---------------------------------------------------------------------------
// NK keys per lane (64 NK slots).  The search runs over the SCORE word first (32-bit compares) and
// stops as soon as a prefix separates exactly k keys from the rest - for unrelated scores that is
// the first bit in which the k-th and the (k+1)-th differ, ~15 of the 32 + lowbits steps of a full
// search; the position bits are searched only when equal scores straddle the cut.  The kept set
// is the full search's, bit for bit.
// PREFIX: the search starts below the score bits that ALL keys share (their AND and OR agree there: a step on such a
// bit keeps or drops every key at once, so the threshold's bits are the keys' own).  Twelve cross-lane steps to find
// them - not worth it where the keys are a row's raw scores (both signs: they differ in the first bit), worth it
// where they are candidates that already won a selection (the split rows' finalize: all close to the row's best;
// on nearly parallel rows - a deep layer's input - they share 20 bits and more, and the full search was the launch's
// critical path: 60 us against 50).
template <int NK, bool PREFIX = false>
__device__ __forceinline__ void wave_topk_keys_n(const unsigned long long (&key)[NK], int k, int lowbits,
                                                 bool (&kept)[NK])
{
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < NK; ++q) cnt += __popcll(__ballot(key[q] != 0ull));
    if (cnt <= k) {                                           // everything that passed thr fits
#pragma unroll
        for (int q = 0; q < NK; ++q) kept[q] = key[q] != 0ull;
        return;
    }
    unsigned hi[NK];
#pragma unroll
    for (int q = 0; q < NK; ++q) hi[q] = (unsigned)(key[q] >> 32);
    unsigned Th = 0;
    int b_first = 31;
    if constexpr (PREFIX) {
        unsigned o = 0u, an = 0xFFFFFFFFu;
#pragma unroll
        for (int q = 0; q < NK; ++q)
            if (key[q] != 0ull) { o |= hi[q]; an &= hi[q]; }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) { o |= __shfl_xor(o, m, 64); an &= __shfl_xor(an, m, 64); }
        const unsigned diff = o ^ an;                         // (cnt > k >= 0: there are keys)
        b_first = diff ? 31 - __clz(diff) : -1;               // the highest bit in which two keys differ
        Th = b_first >= 0 ? (an & ~((2u << b_first) - 1u)) : an;
    }
    for (int b = b_first; b >= 0; --b) {
        const unsigned cand = Th | (1u << b);
        int c = 0;
#pragma unroll
        for (int q = 0; q < NK; ++q) c += __popcll(__ballot(hi[q] >= cand));
        if (c >= k) {
            Th = cand;
            if (c == k) {                                     // (wave-uniform) no further bit changes the set
#pragma unroll
                for (int q = 0; q < NK; ++q) kept[q] = hi[q] >= Th;
                return;
            }
        }
    }
    // equal scores on both sides of the cut: the edge position decides
    unsigned long long T = ((unsigned long long)Th << 32) | (0xFFFFFFFFull & ~((1ull << lowbits) - 1ull));
    for (int b = lowbits - 1; b >= 0; --b) {
        const unsigned long long cand = T | (1ull << b);
        int c = 0;
#pragma unroll
        for (int q = 0; q < NK; ++q) c += __popcll(__ballot(key[q] >= cand));
        if (c >= k) {
            T = cand;
            if (c == k) break;
        }
    }
#pragma unroll
    for (int q = 0; q < NK; ++q) kept[q] = key[q] >= T;      // T > 0 here, so empty slots (key 0) stay out
}

// ---------------------------------------------------------------------------
// Dropout drawn in the kernel: keep(seed, element index) = u >= p with u a counter-based uniform
// in [0, 1) - two rounds of a 32-bit avalanche hash over the (hashed) seed and the flat element
// index.  The same element of the same call gets the same draw in every kernel that asks (the
// aggregation's main and finalize kernels, the blend); the seed is hashed first so that
// consecutive seeds do not give masks that are each other's pairwise swaps.
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned sn_mix32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ bool sn_dropout_keep(unsigned long long seed, unsigned long long idx, float p)
{
    const unsigned k1 = sn_mix32((unsigned)seed * 0x9E3779B9u + (unsigned)(seed >> 32)), k2 = sn_mix32(k1 ^ 0x85EBCA6Bu);
    unsigned hsh = sn_mix32((unsigned)idx + k1);
    hsh = sn_mix32((hsh ^ k2) + 0x9E3779B9u * (unsigned)(idx >> 32));
    return (float)(hsh >> 8) * 5.9604644775390625e-8f >= p;
}

// ---- fp32 operands for the bf16 matrix cores --------------------------------------------------
// A float is the EXACT sum of three bf16 values (truncate to the top 16 bits, subtract, repeat:
// 8 + 8 + 8 significant bits; every step exact), and a product of two bf16 values is exact in
// fp32.  The fp32 contractions of the library (lin, dense cosine, kNN builder) therefore run as
// eight bf16 partial products x_i y_j, i + j <= 5, accumulated in fp32 by `v_mfma_f32_*_bf16`
// (XDL pipe, 8 k-slots per lane and instruction) instead of fp32 MFMAs, which execute on the
// vector ALU's own pipes at a quarter of the k-slots per cycle (linear.hip has the measurements).
using sn_bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using sn_u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using sn_u32x2 = __attribute__((ext_vector_type(2))) unsigned;

// one float -> its three planes' bit patterns (the bf16 value of plane p is the top half of u[p])
__device__ __forceinline__ void split_bf16_bits(float v, unsigned &u1, unsigned &u2, unsigned &u3)
{
    u1 = __float_as_uint(v);
    const float r1 = v - __uint_as_float(u1 & 0xFFFF0000u);     // exact; <= 16 significant bits (inf, NaN -> NaN)
    u2 = __float_as_uint(r1);
    u3 = __float_as_uint(r1 - __uint_as_float(u2 & 0xFFFF0000u));   // exact; <= 8 significant bits
}
// top halves of two floats' bits -> one register (even element in the low half)
__device__ __forceinline__ unsigned pack_bf16_hi(unsigned even, unsigned odd)
{
    return __builtin_amdgcn_perm(odd, even, 0x07060302u);
}
__device__ __forceinline__ void split_bf16x8(const float (&v)[8], sn_u32x4 &p1, sn_u32x4 &p2, sn_u32x4 &p3)
{
    unsigned u1[8], u2[8], u3[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) split_bf16_bits(v[i], u1[i], u2[i], u3[i]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        p1[q] = pack_bf16_hi(u1[2 * q], u1[2 * q + 1]);
        p2[q] = pack_bf16_hi(u2[2 * q], u2[2 * q + 1]);
        p3[q] = pack_bf16_hi(u3[2 * q], u3[2 * q + 1]);
    }
}
__device__ __forceinline__ void split_bf16x4(const float (&v)[4], sn_u32x2 &p1, sn_u32x2 &p2, sn_u32x2 &p3)
{
    unsigned u1[4], u2[4], u3[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split_bf16_bits(v[i], u1[i], u2[i], u3[i]);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        p1[q] = pack_bf16_hi(u1[2 * q], u1[2 * q + 1]);
        p2[q] = pack_bf16_hi(u2[2 * q], u2[2 * q + 1]);
        p3[q] = pack_bf16_hi(u3[2 * q], u3[2 * q + 1]);
    }
}

// Correctly rounded fp32 square root and division.  Plain sqrtf() and `/` ARE that under
// hipcc's defaults (-fhip-fp32-correctly-rounded-divide-sqrt; the build passes no fast-math
// flag) - unlike HIP's __fsqrt_rn(), which ROCm 7.2's headers map to the approximate
// __ocml_native_sqrt_f32 unless OCML_BASIC_ROUNDED_OPERATIONS is defined.
__device__ __forceinline__ float ieee_sqrt(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ float ieee_div(float a, float b) { return a / b; }

// One node row spread over the G lanes of a group: lane lg holds VEC consecutive
// channels per step, R steps.  Channels beyond C read as zero.
template <int VEC, int G, int R> struct Row {
    float x[R][VEC];

    __device__ __forceinline__ void zero()
    {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int v = 0; v < VEC; ++v) x[r][v] = 0.f;
    }

    // Lanes whose channels lie beyond C read channel 0 and discard it: the load itself is
    // unconditional.  (A load under `if (c0 < C)` makes the compiler drain the memory
    // pipeline - s_waitcnt vmcnt(0) - at every such branch, i.e. between the loads of a
    // batch of rows that are meant to be in flight together.)
    __device__ __forceinline__ void load(const float *__restrict__ row, int C, int lg)
    {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int c0 = (r * G + lg) * VEC;
            const bool in = c0 < C;
            const int cc = in ? c0 : 0;
            if constexpr (VEC == 4) {
                const float4 t = *reinterpret_cast<const float4 *>(row + cc);
                x[r][0] = in ? t.x : 0.f; x[r][1] = in ? t.y : 0.f; x[r][2] = in ? t.z : 0.f; x[r][3] = in ? t.w : 0.f;
            } else if constexpr (VEC == 2) {
                const float2 t = *reinterpret_cast<const float2 *>(row + cc);
                x[r][0] = in ? t.x : 0.f; x[r][1] = in ? t.y : 0.f;
            } else {
                const float t = row[cc];
                x[r][0] = in ? t : 0.f;
            }
        }
    }

    __device__ __forceinline__ void store(float *__restrict__ row, int C, int lg) const
    {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int c0 = (r * G + lg) * VEC;
            if (c0 < C) {
                if constexpr (VEC == 4) {
                    *reinterpret_cast<float4 *>(row + c0) = make_float4(x[r][0], x[r][1], x[r][2], x[r][3]);
                } else if constexpr (VEC == 2) {
                    *reinterpret_cast<float2 *>(row + c0) = make_float2(x[r][0], x[r][1]);
                } else {
                    row[c0] = x[r][0];
                }
            }
        }
    }

    // per-lane partial of <this, o> (fma chain in channel order)
    __device__ __forceinline__ float dot_partial(const Row &o) const
    {
        float d = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int v = 0; v < VEC; ++v) d = fmaf(x[r][v], o.x[r][v], d);
        return d;
    }

    // this += w * o   with the product rounded before the add, like ATen's
    // (weight * x_j) followed by scatter_add
    __device__ __forceinline__ void axpy(float w, const Row &o)
    {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int v = 0; v < VEC; ++v) x[r][v] = __fadd_rn(x[r][v], __fmul_rn(w, o.x[r][v]));
    }

    __device__ __forceinline__ void add(const Row &o)
    {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int v = 0; v < VEC; ++v) x[r][v] += o.x[r][v];
    }

    __device__ __forceinline__ void scale(float w)
    {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int v = 0; v < VEC; ++v) x[r][v] *= w;
    }

    __device__ __forceinline__ void div(float d)
    {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int v = 0; v < VEC; ++v) x[r][v] = x[r][v] / d;
    }

    // x / d, correctly rounded (F.normalize's division)
    __device__ __forceinline__ void div_rn(float d)
    {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int v = 0; v < VEC; ++v) x[r][v] = ieee_div(x[r][v], d);
    }

    // sum the per-group rows of a wave (all lanes end with the total)
    __device__ __forceinline__ void reduce_across_groups()
    {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int v = 0; v < VEC; ++v) x[r][v] = cross_group_sum<G>(x[r][v]);
    }
};

// 1 / max(||row||_2, eps) from the group-summed sum of squares: hardware
// reciprocal square root of max(sumsq, eps^2) plus one Newton step (about half an
// ulp; the same bits wherever a row's norm is needed, forward and backward).
__device__ __forceinline__ float inv_norm_of(float sumsq)
{
    const float q = fmaxf(sumsq, EPS_NORM * EPS_NORM);
    const float r = __builtin_amdgcn_rsqf(q);
    return r * fmaf(-0.5f * q * r, r, 1.5f);
}

// F.normalize clamps the norm at eps; below it the normalisation is a plain scale
__device__ __forceinline__ bool norm_clamped(float sumsq) { return sumsq < EPS_NORM * EPS_NORM; }

// cosine of target row a (inverse norm inv_i) and source row x from the UN-normalised rows:
// <a, x> * (inv_i * inv_j).  Used by the backward and the attention mode, where the value
// only has to be accurate; the selecting forward scores unit rows instead (agg_fwd_impl.h:
// normalise-then-dot, so that equal unit rows give equal bits).
template <int VEC, int G, int R>
__device__ __forceinline__ float edge_score(const Row<VEC, G, R> &a, float inv_i,
                                            const Row<VEC, G, R> &x)
{
    float d = group_sum<G>(a.dot_partial(x));
    float q = group_sum<G>(x.dot_partial(x));
    const float s = d * (inv_i * inv_norm_of(q));
    return s + 0.0f;     // -0.0 -> +0.0: the reference orders floats, not bit patterns
}

}  // namespace sngnn
