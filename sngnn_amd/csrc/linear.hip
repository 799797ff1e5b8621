// self.lin forward (models/models.py:121,237,324):  h = x W^T + b  for the skinny
// shapes of this path (N rows in the 1e5..1e6 range, C <= 64 outputs).  The product
// is HBM-bound - x is read once (4 N F bytes), h written once - and the BLAS kernel
// picked for it reaches ~2.3 TB/s; here a workgroup streams a 128-row panel of x
// through LDS in coalesced 128-byte segments with the next panel already in
// registers, and multiplies with fp32 MFMAs (v_mfma_f32_32x32x2_f32): 4 waves x
// (32 rows x NT*32 columns).  W^T panels come from L2.  (This panel kernel serves the shapes the
// row-tile kernel below does not: F outside {16, 32, 64, 128}.  The row-tile kernel - every
// shape of the benchmarked models - multiplies on the bf16 matrix cores after an exact split.)
#include <algorithm>
#include <cstdlib>

#include "agg_fwd_filter.h"
#include "common.h"
#include "device_utils.h"

namespace sngnn {

// Orders LDS traffic between the lanes of ONE wave: LDS-only fences (the wave's global
// loads and stores stay in flight - an all-address-space fence would wait for them with
// s_waitcnt vmcnt(0) and serialise the pipeline below).
__device__ __forceinline__ void wave_barrier_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int LF_M = 128, LF_K = 32, LF_LD = LF_K + 1;

// act != NULL (sngnn_linear_forward_masked): the result leaves through the mask of an activated
// tensor of its own shape, h = act > 0 ? (x W^T + b) * act_scale : 0 - the input gradient of a
// layer whose input was relu (+ inverted dropout) of something: the two elementwise backward
// passes of models.py:206-209 folded into the store that produces their operand.
template <int NT, bool MASK = false>      // NT 32-column tiles of the output (C <= 32 NT)
__global__ __launch_bounds__(256) void k_linear_fwd(const float *__restrict__ x, const float *__restrict__ w,
                                                    const float *__restrict__ b, int64_t N, int F, int C,
                                                    float *__restrict__ h, const float *__restrict__ act = nullptr,
                                                    float act_scale = 1.0f)
{
    __shared__ float sx[LF_M * LF_LD];
    __shared__ float sw[NT * 32 * LF_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * LF_M;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // MASK: the mask values of this lane's outputs, requested before anything else (unconditional,
    // clamped addresses): they travel while the panels stream.  (Loaded in the epilogue they were a
    // burst behind the last multiply of every workgroup at once: 55.6 us against 32.9 us unmasked.)
    float av[MASK ? NT : 1][16];
    if constexpr (MASK) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t rr = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                av[t][r] = act[min(rr, N - 1) * C + min(t * 32 + (lane & 31), C - 1)];
            }
    }
    const int sc = tid & 31, sr = tid >> 5;             // staging: column sc of rows sr + 8 u
    float rx[LF_M / 8], rw[NT * 4];
    auto fetch = [&](int k0) {
        const int k = k0 + sc;
#pragma unroll
        for (int u = 0; u < LF_M / 8; ++u) {
            const int64_t r = row0 + sr + 8 * u;
            rx[u] = (r < N && k < F) ? x[r * F + k] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < NT * 4; ++u) {
            const int c = sr + 8 * u;
            rw[u] = (c < C && k < F) ? w[(int64_t)c * F + k] : 0.f;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < F; k0 += LF_K) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < LF_M / 8; ++u) sx[(sr + 8 * u) * LF_LD + sc] = rx[u];
#pragma unroll
        for (int u = 0; u < NT * 4; ++u) sw[(sr + 8 * u) * LF_LD + sc] = rw[u];
        __syncthreads();
        if (k0 + LF_K < F) fetch(k0 + LF_K);
#pragma unroll
        for (int kk = 0; kk < LF_K; kk += 2) {
            // A[i = l & 31][k = l >> 5] = x row of this wave's 32-row slab, B[k][j = l & 31] = W[j][k]
            const float a = sx[(wave * 32 + (lane & 31)) * LF_LD + kk + (lane >> 5)];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float bv = sw[(t * 32 + (lane & 31)) * LF_LD + kk + (lane >> 5)];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[t], 0, 0, 0);
            }
        }
    }
    // C/D layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int c = t * 32 + (lane & 31);
        const float bias = (b != nullptr && c < C) ? b[c] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t rr = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            float v = acc[t][r] + bias;
            if constexpr (MASK) v = av[t][r] > 0.f ? v * act_scale : 0.f;
            if (rr < N && c < C) h[rr * C + c] = v;
        }
    }
}


// ---------------------------------------------------------------------------
// Row-tile variant for F in {16, 32, 64, 128} (F = 16 FQ): persistent waves, each
// owning 16-row tiles.  (fp32-MFMA form; the default bf16-split form, BF3 below, keeps this
// layout with 8 k's per lane and step.)  v_mfma_f32_16x16x4_f32 contracts 4 k's per step and the order
// of the k's is free, so lane (row r = l & 15, quarter q = l >> 4) takes the
// CONTIGUOUS quarter k in [q F/4, (q+1) F/4) of its row (A) / of its output column's
// weight row (B): the whole W^T slice a lane ever needs is NT * F/4 registers, loaded
// once per wave, and x is the only stream.  A tile of x is fetched with fully
// coalesced 16-byte lane loads (whole rows), transposed through a wave-private LDS
// tile (row stride F + 4 floats: conflict-free 16-byte reads); two tiles are staged in
// registers, so a tile's loads are issued two multiply blocks before they are needed.
// x and h are addressed through buffer resources: the hardware bounds check replaces
// the row / column predicates (no branches, no 64-bit address arithmetic in the loop).
// No workgroup barriers.
// ---------------------------------------------------------------------------
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr unsigned BUF_OOB = 0x80000000u;       // an offset no table of < 2 GiB contains

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}

// ---- fp32 products on the bf16 matrix cores (BF3 = true) --------------------------------------
// The fp32 MFMAs (`v_mfma_f32_16x16x4_f32`) are not XDL operations: they run on the vector ALU's
// pipes, 32 cycles each, and nothing else of the wave executes beside them - 13.5 us of the
// 128 -> 40 kernel's 30, in front of which the LDS transpose and (NORM) the epilogue's divisions
// queue up.  A float is the exact sum of three bf16 values (8 + 8 + 8 significant bits: truncate
// to the top 16 bits, subtract, repeat - every step exact), a product of two bf16 values is
// exact in fp32, and `v_mfma_f32_16x16x32_bf16` multiplies 8 k-slots per lane in 16 cycles on
// the XDL pipe while the vector ALU stays free.  x W^T is therefore computed as the eight partial
// products x_i w_j, i + j <= 5 (all but x_3 w_3, which is below 2^-32 of the result), accumulated
// in fp32 by the matrix core, smallest terms first: the rounding is that of an fp32 dot product
// (tests/test_head_linear_gpu.py checks it against float64 beside the fp32-MFMA form), the
// matrix work drops from 3 072 to 1 536 cycles per tile and hides the split, the transpose and
// the epilogue behind it.  Non-finite inputs: an infinity or a NaN in x (in W) makes its whole
// output row (column) NaN - the second plane of inf is inf - inf - where fp32 arithmetic gives
// +-inf for an infinity times a non-zero weight: the one deviation (such rows are NaN one line
// later in the reference too: F.normalize of a row with an infinity, models.py:238).
using u32x4v = sn_u32x4;

__device__ __forceinline__ f32x4 mfma_bf16(const u32x4v &a, const u32x4v &b, const f32x4 &c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sn_bf16x8, a), __builtin_bit_cast(sn_bf16x8, b), c, 0, 0, 0);
}

// (one wave per SIMD is what is launched; the widest configuration - 64 output columns of 128
// inputs: W^T alone is 128 registers - is allowed the whole register file, the others keep the
// two-wave budget they were tuned with: at 256 registers the wide one spilled and took 65 us
// instead of 36 us; rocBLAS: 83 us)
// NORM: the epilogue also writes F.normalize of the finished rows (models.py:237-238 are adjacent
// lines) - unit rows, clamped norms and, when asked for, the fp16 filter rows - with the very
// instruction sequence of k_normalize_rows (agg_fwd_impl.h): a finished 16-row tile goes through a
// wave-private LDS tile into the aggregation's row layout (VEC 4, G = 8 or 16 lanes per row),
// there the same fma chain, the same DPP tree, the same IEEE square root and division.  Needs
// C % 4 == 0.  h itself then leaves from that layout too (16-byte stores).
// BF3: the products on the bf16 matrix cores (above; FQ even); W^T is then three bf16 planes -
// 1.5x the registers - and the kernel gets the whole register file of its one wave per SIMD.
template <int NT, int FQ, bool NORM, bool BF3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(((NT == 4 && FQ == 8) || BF3) ? 1 : 2, ((NT == 4 && FQ == 8) || BF3) ? 1 : 2)))
void k_linear_rows(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ b,
                   int N, int C, float *__restrict__ h, int ntiles, float *__restrict__ un,
                   float *__restrict__ unrm, void *__restrict__ filt)
{
    constexpr int F = 16 * FQ, KS = 4 * FQ, LD = F + 4;
    constexpr int RPI = 16 / FQ;                     // rows covered by one wave-wide 16-byte load
    constexpr int HLD = 16 * NT + 4;                 // row stride of the finished tile in LDS
    __shared__ __align__(16) float lds[4][16 * LD];
    __shared__ __align__(16) float hts[NORM ? 4 : 1][NORM ? 16 * HLD : 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    float *tile = lds[wave];
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(x, (unsigned)N * F * 4u);
    const __amdgpu_buffer_rsrc_t hr = make_rsrc(h, (unsigned)N * (unsigned)C * 4u);

    const int lrow = lane / KS, lc4 = lane % KS;     // staging: row within a load, float4 within the row
    const unsigned xoff_lane = (unsigned)(lrow * F + 4 * lc4) * 4u;
    // load q of a tile covers rows [q RPI, (q+1) RPI): RPI * F * 4 = 1024 bytes further on
    auto xload = [&](unsigned tile_off, int q) {
        return __builtin_amdgcn_raw_buffer_load_b128(xr, tile_off + xoff_lane + (unsigned)q * 1024u, 0, 0);
    };
    // Two staged tiles in named registers (a loop-carried private array gets demoted to
    // LDS/scratch): set a holds tiles 0, 2, 4.. of this wave, set b tiles 1, 3, 5..; a set is
    // refilled right after it has been copied to LDS, i.e. two multiply blocks ahead of use.
    // Tiles past the end read out of bounds: zeros, never used.
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    u32x4 a0, a1, a2, a3, a4, a5, a6, a7, b0, b1, b2, b3, b4, b5, b6, b7;
    a0 = a1 = a2 = a3 = a4 = a5 = a6 = a7 = b0 = b1 = b2 = b3 = b4 = b5 = b6 = b7 = u32x4{0u, 0u, 0u, 0u};
#define SNGNN_FETCH_TILE(P, TL)                                                            \
    {                                                                                      \
        const unsigned to_ = (unsigned)(TL) * (16u * F * 4u);                              \
        P##0 = xload(to_, 0);                                                              \
        if constexpr (FQ > 1) P##1 = xload(to_, 1);                                        \
        if constexpr (FQ > 2) { P##2 = xload(to_, 2); P##3 = xload(to_, 3); }              \
        if constexpr (FQ > 4) { P##4 = xload(to_, 4); P##5 = xload(to_, 5); P##6 = xload(to_, 6); P##7 = xload(to_, 7); } \
    }
#define SNGNN_PUT_TILE(P)                                                                  \
    {                                                                                      \
        auto put = [&](int q, const u32x4 &v) {                                            \
            *reinterpret_cast<u32x4 *>(tile + (q * RPI + lrow) * LD + 4 * lc4) = v;        \
        };                                                                                 \
        put(0, P##0);                                                                      \
        if constexpr (FQ > 1) put(1, P##1);                                                \
        if constexpr (FQ > 2) { put(2, P##2); put(3, P##3); }                              \
        if constexpr (FQ > 4) { put(4, P##4); put(5, P##5); put(6, P##6); put(7, P##7); }  \
    }
    const int nwaves = gridDim.x * 4;
    int tl = blockIdx.x * 4 + wave;
    SNGNN_FETCH_TILE(a, tl)
    SNGNN_FETCH_TILE(b, tl + nwaves)

    // B operands: W[c = 16 t + r16][kq KS + s].  (Columns >= C are clamped, not masked: they
    // only feed outputs that are never stored.)
    constexpr int NS = BF3 ? FQ / 2 : 1;              // bf16 steps of 8 k-slots per lane
    float bw[BF3 ? 1 : NT][BF3 ? 1 : KS];
    u32x4v bw1[BF3 ? NT : 1][NS], bw2[BF3 ? NT : 1][NS], bw3[BF3 ? NT : 1][NS];
    float bias[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int c = min(16 * t + r16, C - 1);
        bias[t] = b != nullptr ? b[c] : 0.f;
        if constexpr (!BF3) {
#pragma unroll
            for (int j = 0; j < FQ; ++j) {
                const float4 v = *reinterpret_cast<const float4 *>(w + (size_t)c * F + kq * KS + 4 * j);
                bw[t][4 * j + 0] = v.x; bw[t][4 * j + 1] = v.y; bw[t][4 * j + 2] = v.z; bw[t][4 * j + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int sb = 0; sb < NS; ++sb) {
                const float4 v0 = *reinterpret_cast<const float4 *>(w + (size_t)c * F + kq * KS + 8 * sb);
                const float4 v1 = *reinterpret_cast<const float4 *>(w + (size_t)c * F + kq * KS + 8 * sb + 4);
                const float wv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                split_bf16x8(wv, bw1[t][sb], bw2[t][sb], bw3[t][sb]);
            }
        }
    }
    // D: col = lane & 15, row = 4 (lane >> 4) + reg.  Byte offset of this lane's element (t, r)
    // inside a tile of h; a column >= C gets an out-of-bounds offset (the store is dropped),
    // and so does, by itself, a row >= N.
    unsigned hoff[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            hoff[t][r] = (16 * t + r16 < C) ? (unsigned)((4 * kq + r) * C + 16 * t + r16) * 4u : BUF_OOB;
    // NORM: buffer resources of the three extra outputs; lane roles in the aggregation's row layout
    constexpr int GN = NT <= 2 ? 8 : 16;             // lanes per row: C <= 32 -> 8, C <= 64 -> 16 (row_cfg)
    constexpr int RPWN = 64 / GN;
    const int gidn = lane / GN, lgn = lane % GN;
    __amdgpu_buffer_rsrc_t ur = hr, nr = hr, fr = hr;
    if constexpr (NORM) {
        ur = make_rsrc(un, (unsigned)N * (unsigned)C * 4u);
        nr = make_rsrc(unrm, (unsigned)N * 4u);
        fr = make_rsrc(filt, filt != nullptr ? (unsigned)N * 128u : 0u);
    }
    // NORM epilogue: part 1 writes the finished tile into the wave's LDS tile, part 2 reads it back
    // in the aggregation's row layout.  (Carried INSIDE the next tile's multiply block - sliced
    // between the MFMAs, fp32 or bf16 - it was no faster: 38.3 against 36.9 us.)
    auto epi_write = [&](const f32x4 (&v)[NT]) {
        float *ht = hts[wave];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) ht[(4 * kq + r) * HLD + 16 * t + r16] = v[t][r];
    };
    auto epi_rest = [&](int tl_) {
        // Stage by stage over ALL the tile's passes (4 rows of 16 lanes, or 8 of 8, per pass): one
        // wave per SIMD has nobody to hide a dependent instruction's latency behind, so the
        // independent rows are laid side by side (38.9 -> 36.9 us at 128 -> 40).
        using RowT = Row<4, GN, 1>;
        using u32x4n = __attribute__((ext_vector_type(4))) unsigned;
        constexpr int NP = 16 / RPWN;
        float *ht = hts[wave];
        const bool in = 4 * lgn < C;
        RowT xr[NP];
        unsigned roff[NP], grow[NP];
        float q[NP], d[NP];
#pragma unroll
        for (int sidx = 0; sidx < NP; ++sidx) {
            const int rt = sidx * RPWN + gidn;
            grow[sidx] = (unsigned)tl_ * 16u + (unsigned)rt;
            const float4 tv = *reinterpret_cast<const float4 *>(ht + rt * HLD + (in ? 4 * lgn : 0));
            xr[sidx].x[0][0] = in ? tv.x : 0.f; xr[sidx].x[0][1] = in ? tv.y : 0.f;
            xr[sidx].x[0][2] = in ? tv.z : 0.f; xr[sidx].x[0][3] = in ? tv.w : 0.f;
            roff[sidx] = in ? (grow[sidx] * (unsigned)C + 4u * lgn) * 4u : BUF_OOB;
            __builtin_amdgcn_raw_buffer_store_b128(
                u32x4n{__float_as_uint(xr[sidx].x[0][0]), __float_as_uint(xr[sidx].x[0][1]),
                       __float_as_uint(xr[sidx].x[0][2]), __float_as_uint(xr[sidx].x[0][3])}, hr, roff[sidx], 0, 0);
        }
        // F.normalize, exactly as k_normalize_rows does it
#pragma unroll
        for (int sidx = 0; sidx < NP; ++sidx) q[sidx] = xr[sidx].dot_partial(xr[sidx]);
#pragma unroll
        for (int sidx = 0; sidx < NP; ++sidx) q[sidx] = group_sum<GN>(q[sidx]);
#pragma unroll
        for (int sidx = 0; sidx < NP; ++sidx) d[sidx] = fmaxf(ieee_sqrt(q[sidx]), EPS_NORM);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
#pragma unroll
            for (int sidx = 0; sidx < NP; ++sidx)
                xr[sidx].x[0][cc] = ieee_div(xr[sidx].x[0][cc], d[sidx]);        // == Row::div_rn
            }
#pragma unroll
        for (int sidx = 0; sidx < NP; ++sidx) {
            __builtin_amdgcn_raw_buffer_store_b128(
                u32x4n{__float_as_uint(xr[sidx].x[0][0]), __float_as_uint(xr[sidx].x[0][1]),
                       __float_as_uint(xr[sidx].x[0][2]), __float_as_uint(xr[sidx].x[0][3])}, ur, roff[sidx], 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(d[sidx]), nr, lgn == 0 ? grow[sidx] * 4u : BUF_OOB, 0, 0);
            if constexpr (GN == 16) {
                using u32x2n = __attribute__((ext_vector_type(2))) unsigned;
                __builtin_amdgcn_raw_buffer_store_b64(
                    u32x2n{pack_half2(xr[sidx].x[0][0] * FILT_SCALE, xr[sidx].x[0][1] * FILT_SCALE),
                           pack_half2(xr[sidx].x[0][2] * FILT_SCALE, xr[sidx].x[0][3] * FILT_SCALE)},
                    fr, grow[sidx] * 128u + 8u * lgn, 0, 0);
            }
        }
    };
    auto store_tile = [&](int tl_, const f32x4 (&v)[NT]) {
        if constexpr (!NORM) {
            const unsigned to = (unsigned)tl_ * 16u * (unsigned)C * 4u;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[t][r]), hr, hoff[t][r] + to, 0, 0);
        } else {
            epi_write(v);
            wave_barrier_lds();
            epi_rest(tl_);
        }
    };
    // The finished values of tile i (bias added) stay in their own registers and are stored
    // in iteration i + 1, before that iteration's loads are issued.  A store reads its data
    // registers late, so registers that were just handed to a store cannot be rewritten
    // until it completes (s_waitcnt vmcnt): with a whole multiply block between the stores
    // and the next write of prev[] that wait is free, and the loads behind the stores in the
    // in-order vmcnt queue are not waited for.
    f32x4 prev[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) prev[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    int prev_tl = ntiles;            // "no tile yet": its stores fall out of bounds
    // The W registers are re-defined by empty asm statements before the loop, so their loads
    // are waited for HERE: otherwise the loop body keeps the waits its first trip needs for
    // them (s_waitcnt vmcnt(n) in front of the MFMAs), and on every later trip those same
    // waits drain the stores issued just before them.
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        asm volatile("" : "+v"(bias[t]));
        if constexpr (!BF3) {
#pragma unroll
            for (int k = 0; k < KS; ++k) asm volatile("" : "+v"(bw[t][k]));
        } else {
#pragma unroll
            for (int sb = 0; sb < NS; ++sb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    asm volatile("" : "+v"(bw1[t][sb][q]));
                    asm volatile("" : "+v"(bw2[t][sb][q]));
                    asm volatile("" : "+v"(bw3[t][sb][q]));
                }
        }
    }
    auto multiply = [&](int tcur) {
        float4 a[FQ];
#pragma unroll
        for (int j = 0; j < FQ; ++j)
            a[j] = *reinterpret_cast<const float4 *>(tile + r16 * LD + kq * KS + 4 * j);
        // keep the loads above the multiply block (the scheduler would sink them below it to
        // save registers, which exposes the whole memory latency every iteration)
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (!BF3) {
#pragma unroll
            for (int j = 0; j < FQ; ++j) {
                const float av[4] = {a[j].x, a[j].y, a[j].z, a[j].w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bw[t][4 * j + e], acc[t], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int sb = 0; sb < NS; ++sb) {
                const float av[8] = {a[2 * sb].x, a[2 * sb].y, a[2 * sb].z, a[2 * sb].w,
                                     a[2 * sb + 1].x, a[2 * sb + 1].y, a[2 * sb + 1].z, a[2 * sb + 1].w};
                u32x4v a1, a2, a3;
                split_bf16x8(av, a1, a2, a3);
#pragma unroll
                for (int t = 0; t < NT; ++t) {                 // smallest partial products first
                    acc[t] = mfma_bf16(a3, bw2[t][sb], acc[t]);
                    acc[t] = mfma_bf16(a2, bw3[t][sb], acc[t]);
                    acc[t] = mfma_bf16(a3, bw1[t][sb], acc[t]);
                    acc[t] = mfma_bf16(a2, bw2[t][sb], acc[t]);
                    acc[t] = mfma_bf16(a1, bw3[t][sb], acc[t]);
                    acc[t] = mfma_bf16(a2, bw1[t][sb], acc[t]);
                    acc[t] = mfma_bf16(a1, bw2[t][sb], acc[t]);
                    acc[t] = mfma_bf16(a1, bw1[t][sb], acc[t]);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) prev[t][r] = acc[t][r] + bias[t];
        prev_tl = tcur;
    };
#define SNGNN_TILE_STEP(P, TCUR)                                                           \
    {                                                                                      \
        wave_barrier_lds();          /* the previous tile's LDS reads are done */          \
        SNGNN_PUT_TILE(P)                                                                  \
        wave_barrier_lds();                                                                \
        store_tile(prev_tl, prev);                                                         \
        SNGNN_FETCH_TILE(P, (TCUR) + 2 * nwaves)                                           \
        multiply(TCUR);                                                                    \
    }
    for (; tl < ntiles; tl += 2 * nwaves) {
        SNGNN_TILE_STEP(a, tl)
        if (tl + nwaves < ntiles) SNGNN_TILE_STEP(b, tl + nwaves)
    }
    store_tile(prev_tl, prev);
}

#undef SNGNN_TILE_STEP
#undef SNGNN_PUT_TILE
#undef SNGNN_FETCH_TILE

// ---------------------------------------------------------------------------
// Weight gradient of the same layer on the matrix cores:
//     dW[C, F] = g^T [C, N] . x [N, F],   db[C] = sum_rows g          (train.py:86)
// The contraction runs over the N rows, so every wave streams its own 16-row blocks and
// keeps a full C x F partial in accumulators (NT x UF tiles of 16 x 16).  Both the row order
// inside a block and the assignment of output rows / columns to lanes are free, which
// lets every operand come straight from a coalesced global load, no LDS:
//   A (16 channels x 4 rows): lane (i = l & 15, kq = l >> 4) holds g[row + kq][NT i + t]
//       - NT consecutive floats per lane,
//   B (4 rows x 16 features): lane (j, kq) holds x[row + kq][f(j, u)] with
//       f(j, u) = 64 (u / 4) + 4 j + u % 4  (UF >= 4)  or  UF j + u  - one or two 16-byte
//       loads per lane, a whole row per 16 lanes.
// Rows past N and channels past C read out of bounds (zero) through buffer resources.
// A workgroup's four partials are added in LDS in wave order and written out; the
// existing k_sum_partials adds the workgroups (fixed order: deterministic).
// ---------------------------------------------------------------------------
template <int NT, int UF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2)))
void k_wgrad_mfma(const float *__restrict__ g, const float *__restrict__ x, int N, int C,
                  float *__restrict__ part, float *__restrict__ part_b, int nblocks)
{
    constexpr int F = 16 * UF;
    constexpr int XV = UF >= 4 ? 4 : UF;            // floats per x load
    constexpr int XL = UF / XV;                     // x loads per row (1 or 2)
    __shared__ float red[4][NT * UF * 4][64];      // (NT = 4, UF = 8: 128 KiB of the CU's 160)
    __shared__ float redb[4][NT][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    const __amdgpu_buffer_rsrc_t gr = make_rsrc(g, (unsigned)N * (unsigned)C * 4u);
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(x, (unsigned)N * F * 4u);

    // per-lane byte offsets inside a 16-row block: row kq of step 0 (step s adds 4 rows)
    unsigned goff[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
        goff[t] = (NT * i16 + t < C) ? (unsigned)(kq * C + NT * i16 + t) * 4u : BUF_OOB;
    const unsigned xoff = (unsigned)(kq * F + XV * i16) * 4u;
    const unsigned gstep = 4u * (unsigned)C * 4u;   // 4 rows of g

    f32x4 acc[NT][UF];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int u = 0; u < UF; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bsum[t] = 0.f;

    struct Blk { float a[4][NT]; float b[4][UF]; };
    auto fetch = [&](Blk &k, int blk) {
        const unsigned go = (unsigned)blk * 16u * (unsigned)C * 4u;
        const unsigned xo = (unsigned)blk * 16u * F * 4u;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
                k.a[s][t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(gr, goff[t] + go + s * gstep, 0, 0));
#pragma unroll
            for (int q = 0; q < XL; ++q) {
                const unsigned o = xo + xoff + (unsigned)(s * 4 * F * 4 + q * 256);
                if constexpr (XV == 4) {
                    const auto v = __builtin_amdgcn_raw_buffer_load_b128(xr, o, 0, 0);
                    k.b[s][4 * q + 0] = __uint_as_float(v[0]); k.b[s][4 * q + 1] = __uint_as_float(v[1]);
                    k.b[s][4 * q + 2] = __uint_as_float(v[2]); k.b[s][4 * q + 3] = __uint_as_float(v[3]);
                } else if constexpr (XV == 2) {
                    const auto v = __builtin_amdgcn_raw_buffer_load_b64(xr, o, 0, 0);
                    k.b[s][0] = __uint_as_float(v[0]); k.b[s][1] = __uint_as_float(v[1]);
                } else {
                    k.b[s][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, o, 0, 0));
                }
            }
        }
    };
    // one 4-row step of a block: loads of step s of block `blk` into set k
    auto fetch_step = [&](Blk &k, int blk, int s) {
        const unsigned go = (unsigned)blk * 16u * (unsigned)C * 4u;
        const unsigned xo = (unsigned)blk * 16u * F * 4u;
#pragma unroll
        for (int t = 0; t < NT; ++t)
            k.a[s][t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(gr, goff[t] + go + s * gstep, 0, 0));
#pragma unroll
        for (int q = 0; q < XL; ++q) {
            const unsigned o = xo + xoff + (unsigned)(s * 4 * F * 4 + q * 256);
            if constexpr (XV == 4) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(xr, o, 0, 0);
                k.b[s][4 * q + 0] = __uint_as_float(v[0]); k.b[s][4 * q + 1] = __uint_as_float(v[1]);
                k.b[s][4 * q + 2] = __uint_as_float(v[2]); k.b[s][4 * q + 3] = __uint_as_float(v[3]);
            } else if constexpr (XV == 2) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b64(xr, o, 0, 0);
                k.b[s][0] = __uint_as_float(v[0]); k.b[s][1] = __uint_as_float(v[1]);
            } else {
                k.b[s][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, o, 0, 0));
            }
        }
    };
    // Multiply a block and refill its set, step by step: the loads of step s of the block
    // three ahead are issued right behind the MFMAs that consumed step s, so with one wave
    // per SIMD the load issue hides in the MFMA pipeline instead of following it.
    auto multiply_refill = [&](Blk &k, int next_blk) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                bsum[t] += k.a[s][t];
#pragma unroll
                for (int u = 0; u < UF; ++u)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(k.a[s][t], k.b[s][u], acc[t][u], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            fetch_step(k, next_blk, s);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // Three operand sets in rotation: a set is needed again two multiply blocks (~3 us)
    // after its refill - one block is less than the memory latency.  Blocks past the end
    // read zeros.
    const int nwaves = gridDim.x * 4;
    int blk = blockIdx.x * 4 + wave;
    Blk k0, k1, k2;
    fetch(k0, blk);
    fetch(k1, blk + nwaves);
    fetch(k2, blk + 2 * nwaves);
    for (; blk < nblocks; blk += 3 * nwaves) {
        multiply_refill(k0, blk + 3 * nwaves);
        if (blk + nwaves >= nblocks) break;
        multiply_refill(k1, blk + 4 * nwaves);
        if (blk + 2 * nwaves >= nblocks) break;
        multiply_refill(k2, blk + 5 * nwaves);
    }

    // workgroup reduction: every wave parks its partial in LDS, then each thread adds the
    // four copies of its outputs in wave order (deterministic) and writes them out.
    // slot (t, u, r) of lane l is dW[c = NT (4 (l >> 4) + r) + t][f = f(l & 15, u)]
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int u = 0; u < UF; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][(t * UF + u) * 4 + r][lane] = acc[t][u][r];
        float bs = bsum[t];
        bs += __shfl_xor(bs, 16, 64);
        bs += __shfl_xor(bs, 32, 64);
        if (kq == 0) redb[wave][t][i16] = bs;
    }
    __syncthreads();
    float *pw = part + (size_t)blockIdx.x * C * F;
    for (int q = threadIdx.x; q < NT * UF * 4 * 64; q += 256) {
        const int l = q & 63, slot = q >> 6;
        const int r = slot & 3, u = (slot >> 2) % UF, t = (slot >> 2) / UF;
        const int c = NT * (4 * (l >> 4) + r) + t;
        const int j = l & 15;
        const int f = UF >= 4 ? 64 * (u / 4) + 4 * j + u % 4 : UF * j + u;
        if (c < C) pw[(size_t)c * F + f] = ((red[0][slot][l] + red[1][slot][l]) + red[2][slot][l]) + red[3][slot][l];
    }
    if (part_b != nullptr && threadIdx.x < NT * 16) {
        const int t = threadIdx.x / 16, i = threadIdx.x % 16;
        const int c = NT * i + t;
        if (c < C)
            part_b[(size_t)blockIdx.x * C + c] = ((redb[0][t][i] + redb[1][t][i]) + redb[2][t][i]) + redb[3][t][i];
    }
}

template <int UF>
static int launch_wgrad_uf(const float *g, const float *x, int N, int C, float *part, float *part_b,
                           int nwg, hipStream_t st)
{
    const int nblocks = (N + 15) / 16;
    switch ((C + 15) / 16) {
    case 1: k_wgrad_mfma<1, UF><<<nwg, 256, 0, st>>>(g, x, N, C, part, part_b, nblocks); break;
    case 2: k_wgrad_mfma<2, UF><<<nwg, 256, 0, st>>>(g, x, N, C, part, part_b, nblocks); break;
    case 3: k_wgrad_mfma<3, UF><<<nwg, 256, 0, st>>>(g, x, N, C, part, part_b, nblocks); break;
    default: k_wgrad_mfma<4, UF><<<nwg, 256, 0, st>>>(g, x, N, C, part, part_b, nblocks); break;
    }
    SN_HIP(hipGetLastError());
    return SNGNN_OK;
}

// number of per-workgroup partials the MFMA weight gradient writes for N rows (0: shape
// not handled, the caller falls back to k_wgrad_partial)
int wgrad_mfma_partials(int64_t N, int C, int F)
{
    const int64_t lim = ((int64_t)1 << 31) - ((int64_t)64 << 20);
    if (C > 64 || (F != 16 && F != 32 && F != 64 && F != 128)) return 0;
    if (N < 1024 || N * F * 4 >= lim || N * (int64_t)C * 4 >= lim) return 0;
    return (int)std::min<int64_t>(WGRAD_MFMA_WGS, (N + 63) / 64);
}

int launch_wgrad_mfma(const float *g, const float *x, int64_t N, int C, int F, float *part, float *part_b,
                      hipStream_t st)
{
    const int nwg = wgrad_mfma_partials(N, C, F);
    SN_REQUIRE(nwg > 0, SNGNN_EINVAL, "shape not handled by the MFMA weight gradient");
    SN_REQUIRE((uintptr_t)x % 16 == 0, SNGNN_EINVAL, "x must be 16-byte aligned");
    switch (F) {
    case 16: return launch_wgrad_uf<1>(g, x, (int)N, C, part, part_b, nwg, st);
    case 32: return launch_wgrad_uf<2>(g, x, (int)N, C, part, part_b, nwg, st);
    case 64: return launch_wgrad_uf<4>(g, x, (int)N, C, part, part_b, nwg, st);
    default: return launch_wgrad_uf<8>(g, x, (int)N, C, part, part_b, nwg, st);
    }
}

// sngnn_tuning_set(5, v): 0 (default) = the products of k_linear_rows on the bf16 matrix cores
// (exact split, fp32 accumulation), 1 = fp32 MFMAs
static int g_lin_fp32_mfma = 0;
int set_lin_mode(int v) { g_lin_fp32_mfma = v ? 1 : 0; return SNGNN_OK; }
bool fp32_mfma_only() { return g_lin_fp32_mfma != 0; }

template <int FQ>
static int launch_linear_rows(const float *x, const float *w, const float *b, int N, int C, float *h,
                              float *un, float *unrm, void *filt, hipStream_t st)
{
    const int ntiles = (int)((N + 15) / 16);
    // one workgroup per CU (one wave per SIMD): measured faster than two (29 vs 36 us at
    // arxiv size) - the kernel is bound by its MFMA stream, more waves only add W traffic
#ifndef SNGNN_LIN_NORM_WGS
#define SNGNN_LIN_NORM_WGS 1
#endif
    int grid = std::min(256, (ntiles + 3) / 4);
    if (un != nullptr) grid = std::min(256 * SNGNN_LIN_NORM_WGS, (ntiles + 3) / 4);
    const int nt = std::min((C + 15) / 16, 4);
    // products on the bf16 matrix cores (exact three-way split of both operands) unless switched
    // off (sngnn_tuning_set(5, 1)) or the lane holds fewer than 8 k-slots (F = 16)
    const bool bf3 = g_lin_fp32_mfma == 0 && FQ % 2 == 0;
#define SNGNN_LIN_LAUNCH(NTV, NORMV, BF3V)                                                              \
    k_linear_rows<NTV, FQ, NORMV, BF3V><<<grid, 256, 0, st>>>(x, w, b, N, C, h, ntiles, un, unrm, filt)
#define SNGNN_LIN_NT(NORMV, BF3V)                                                                       \
    switch (nt) {                                                                                      \
    case 1: SNGNN_LIN_LAUNCH(1, NORMV, BF3V); break;                                                    \
    case 2: SNGNN_LIN_LAUNCH(2, NORMV, BF3V); break;                                                    \
    case 3: SNGNN_LIN_LAUNCH(3, NORMV, BF3V); break;                                                    \
    default: SNGNN_LIN_LAUNCH(4, NORMV, BF3V); break;                                                   \
    }
    if constexpr (FQ % 2 == 0) {
        if (bf3) {
            if (un != nullptr) { SNGNN_LIN_NT(true, true) } else { SNGNN_LIN_NT(false, true) }
            SN_HIP(hipGetLastError());
            return SNGNN_OK;
        }
    }
    if (un != nullptr) { SNGNN_LIN_NT(true, false) } else { SNGNN_LIN_NT(false, false) }
#undef SNGNN_LIN_NT
#undef SNGNN_LIN_LAUNCH
    SN_HIP(hipGetLastError());
    return SNGNN_OK;
}

}  // namespace sngnn

using namespace sngnn;

// whether sngnn_linear_forward_normalized handles a shape (else: sngnn_linear_forward + a
// normalisation pass)
extern "C" int sngnn_linear_normalized_supported(int64_t N, int F, int C)
{
    const int64_t lim = ((int64_t)1 << 31) - ((int64_t)64 << 20);
    return (C % 4 == 0 && C >= 4 && C <= 64 && (F == 16 || F == 32 || F == 64 || F == 128) && N >= 1 &&
            N * F * 4 < lim && N * (int64_t)128 < lim)
               ? 1 : 0;
}

extern "C" int sngnn_linear_forward_normalized(const float *x, const float *weight, const float *bias, int64_t N,
                                               int F, int C, float *h, float *n, float *nrm, void *filt,
                                               void *stream)
{
    SN_REQUIRE(sngnn_linear_normalized_supported(N, F, C), SNGNN_EINVAL,
               "shape not handled (sngnn_linear_normalized_supported)");
    SN_REQUIRE(x && weight && h && n && nrm, SNGNN_EINVAL, "NULL argument");
    SN_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)weight % 16 == 0 && (uintptr_t)h % 16 == 0 &&
                   (uintptr_t)n % 16 == 0 && (uintptr_t)filt % 16 == 0,
               SNGNN_EINVAL, "x / weight / h / n / filt must be 16-byte aligned");
    SN_REQUIRE(filt == nullptr || filter_row_bytes(C) == 128, SNGNN_EINVAL, "no 128-byte filter rows for this C");
    hipStream_t st = (hipStream_t)stream;
    switch (F) {
    case 16: return launch_linear_rows<1>(x, weight, bias, (int)N, C, h, n, nrm, filt, st);
    case 32: return launch_linear_rows<2>(x, weight, bias, (int)N, C, h, n, nrm, filt, st);
    case 64: return launch_linear_rows<4>(x, weight, bias, (int)N, C, h, n, nrm, filt, st);
    default: return launch_linear_rows<8>(x, weight, bias, (int)N, C, h, n, nrm, filt, st);
    }
}

extern "C" int sngnn_linear_forward(const float *x, const float *weight, const float *bias, int64_t N,
                                    int F, int C, float *h, void *stream)
{
    SN_REQUIRE(N >= 0 && F >= 1 && C >= 1, SNGNN_EINVAL, "bad shape");
    SN_REQUIRE(C <= 64, SNGNN_EINVAL, "sngnn_linear_forward handles C <= 64 (use the BLAS for wider layers)");
    if (N == 0) return SNGNN_OK;
    SN_REQUIRE(x && weight && h, SNGNN_EINVAL, "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    const bool al16 = ((uintptr_t)x % 16 == 0) && ((uintptr_t)weight % 16 == 0);
    // buffer addressing: both tables below 2 GiB (plus two tiles of slack for the prefetch)
    const int64_t lim = ((int64_t)1 << 31) - ((int64_t)64 << 20);
    if (al16 && N * F * 4 < lim && N * (int64_t)C * 4 < lim) {
        switch (F) {
        case 16: return launch_linear_rows<1>(x, weight, bias, (int)N, C, h, nullptr, nullptr, nullptr, st);
        case 32: return launch_linear_rows<2>(x, weight, bias, (int)N, C, h, nullptr, nullptr, nullptr, st);
        case 64: return launch_linear_rows<4>(x, weight, bias, (int)N, C, h, nullptr, nullptr, nullptr, st);
        case 128: return launch_linear_rows<8>(x, weight, bias, (int)N, C, h, nullptr, nullptr, nullptr, st);
        default: break;
        }
    }
    const unsigned grid = (unsigned)((N + LF_M - 1) / LF_M);
    if (C <= 32) k_linear_fwd<1><<<grid, 256, 0, st>>>(x, weight, bias, N, F, C, h);
    else k_linear_fwd<2><<<grid, 256, 0, st>>>(x, weight, bias, N, F, C, h);
    SN_HIP(hipGetLastError());
    return SNGNN_OK;
}

extern "C" int sngnn_linear_forward_masked(const float *x, const float *weight, const float *bias, int64_t N,
                                           int F, int C, const float *act, float act_scale, float *h, void *stream)
{
    SN_REQUIRE(N >= 0 && F >= 1 && C >= 1, SNGNN_EINVAL, "bad shape");
    SN_REQUIRE(C <= 64, SNGNN_EINVAL, "sngnn_linear_forward_masked handles C <= 64");
    if (N == 0) return SNGNN_OK;
    SN_REQUIRE(x && weight && h && act, SNGNN_EINVAL, "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)((N + LF_M - 1) / LF_M);
    if (C <= 32) k_linear_fwd<1, true><<<grid, 256, 0, st>>>(x, weight, bias, N, F, C, h, act, act_scale);
    else k_linear_fwd<2, true><<<grid, 256, 0, st>>>(x, weight, bias, N, F, C, h, act, act_scale);
    SN_HIP(hipGetLastError());
    return SNGNN_OK;
}
