// kNN similarity-graph builder (gfx950): for every node the k most cosine-similar
// nodes, straight from the feature table - the N x N similarity is never stored.
// the producer of edges for the aggregation kernel (the
// "Node-Similarity build" of the north star; the reference only ever materialises S,
// SimGFAToolbox/dense.py:138-141, and has no graph builder).
//
// Workgroup = 4 waves = 128 rows; it walks ALL column tiles of 128 nodes.  A tile
// S[128 x 128] = diag(inv) X_rows X_cols^T diag(inv) comes from the matrix cores at fp32 rounding
// (exact bf16 split or v_mfma_f32_32x32x2_f32) like sngnn_cosine_dense; each WAVE owns 32 whole rows of it
// (1 x 4 blocks of 32 x 32), so a row's 128 new similarities sit in one half-wave and
// its running top-k list (LDS, k 64-bit keys) is private to the wave: no atomics, no
// workgroup barrier in the selection.  A similarity only enters the selection when it
// beats the row's current k-th key (one ballot per accumulator register); after the
// first few tiles that is rare (~k ln(N/k) times per row in total), so the epilogue
// costs a fraction of the MFMA time.
//
// Order: (cosine descending, node id ascending) - the aggregation's tie rule; a node is
// never its own neighbour when exclude_self is set.
#include <algorithm>

#include "device_utils.h"

namespace sngnn {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int KN_M = 128, KN_K = 32, KN_LD = KN_K + 4, KN_LOADS = KN_M * KN_K / 256;   // (stride 36: 16-byte rows, conflict-free b128 reads)
typedef float knn_f4 __attribute__((ext_vector_type(4)));
constexpr int KNN_MAX_K = 32;
#if defined(SNGNN_KNN_EXP) && SNGNN_KNN_EXP == 4            // measurement build: event counters of the selection
__device__ unsigned long long g_knn_dbg[8];
#define SN_KNN_COUNT(i, v) do { if (lane_id() == 0) atomicAdd(&g_knn_dbg[i], (unsigned long long)(v)); } while (0)
#else
#define SN_KNN_COUNT(i, v) do { } while (0)
#endif

// k-th largest of the (unique, non-zero) keys held two per lane; at least k keys are set
__device__ __forceinline__ unsigned long long kth_largest(unsigned long long key0, unsigned long long key1, int k)
{
    unsigned long long T = 0;
    for (int b = 63; b >= 0; --b) {
        const unsigned long long cand = T | (1ull << b);
        const int c = __popcll(__ballot(key0 >= cand)) + __popcll(__ballot(key1 >= cand));
        if (c >= k) T = cand;
    }
    return T;
}

// smallest of the wave's 64-bit values (all lanes get it)
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)v, m, 64), hi = __shfl_xor((unsigned)(v >> 32), m, 64);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        v = o < v ? o : v;
    }
    return v;
}

// Merge the row's list (k slots, 0 = empty) with cnt candidates (cnt + k <= 128); returns
// the row's new threshold key (k-th largest, or 0 while the list is not full).
// Round 5: the k largest by the aggregation's early-exit search (wave_topk_keys_n: it walks the cosine word
// and stops at the first prefix that separates exactly k keys - ~15 ballot steps for unrelated cosines) instead
// of a 64-step search for the k-th key; the threshold is then the smallest kept key.  A scan at arxiv size
// makes 1.65 M merges (9.7 per row), and a merge in one wave holds the workgroup's other three at the step's
// barrier.
__device__ __forceinline__ unsigned long long merge_row(unsigned long long *list, const unsigned long long *cand,
                                                        int cnt, int k)
{
    const int lane = lane_id();
    const int q0 = lane, q1 = lane + 64;
    auto pick = [&](int q) -> unsigned long long {
        return q < k ? list[q] : (q - k < cnt ? cand[q - k] : 0ull);
    };
    const unsigned long long key[2] = {pick(q0), pick(q1)};
    const int total = __popcll(__ballot(key[0] != 0ull)) + __popcll(__ballot(key[1] != 0ull));
    bool kept[2];
    wave_topk_keys_n<2>(key, k, 31, kept);               // (low word = ~node id: 31 bits can differ)
    const unsigned long long m0 = __ballot(kept[0]), m1 = __ballot(kept[1]);
    const int n0 = __popcll(m0), ns = n0 + __popcll(m1);
    unsigned long long T = 0ull;
    if (total >= k) T = wave_min_u64(min(kept[0] ? key[0] : ~0ull, kept[1] ? key[1] : ~0ull));
    wave_lds_sync();                       // every lane has read the old list
    if (kept[0]) list[prefix_popc(m0)] = key[0];
    if (kept[1]) list[n0 + prefix_popc(m1)] = key[1];
    if (lane >= ns && lane < k) list[lane] = 0ull;
    wave_lds_sync();
    return T;
}

// blockIdx.y = column split: this workgroup covers column tiles [y * tiles_per_split, ...) and
// writes its lists as KEYS to part[y][N][k] (nsplit > 1) or the final idx / sim (nsplit == 1).
//
// FH > 0 (F == 2 FH, FH a multiple of 16, F <= 128): the wave's 32 rows never change, so their
// MFMA operands stay in REGISTERS for the whole scan - the k order of a contraction is free,
// lane (row, half h) keeps the contiguous half k in [h FH, (h + 1) FH) of its row - and only
// the column panel streams through LDS (half the staging, 4 LDS reads per 4 MFMAs).
// FH == 0: any F, both panels through LDS.
// BF3 (with FH > 0; default): the products on the bf16 matrix cores (device_utils.h: exact
// three-way split of both operands, eight partial products, fp32 accumulation) - the rows'
// operands are split once, into registers; the column panel is split when it is staged (three
// bf16 planes, rows of 64 + 16 bytes); a step of 16 k-slots per half is 2 x 8
// `v_mfma_f32_32x32x16_bf16` per column block instead of 16 fp32 MFMAs.
constexpr int KN_PS = 80;                      // bytes per row of a bf16 plane of the column panel
// NWV waves per workgroup (32 rows each), CB column blocks of 32 per tile.  <4, 4>: 128 x 128 tiles, one wave per
// SIMD (64 accumulator registers).  <8, 2> (round 5, F = 128): 256 rows x 64-column tiles - 32 accumulator registers,
// the kernel fits the 256-register budget of TWO waves per SIMD, so one wave's selection, staging and barrier waits
// run under the other's products (at one wave per SIMD nothing overlapped them: products ~24-35 ms, everything else
// ~45 of the 76 ms at arxiv size).
template <int FH, bool BF3, int NWV = 4, int CB = 4>
__global__ __launch_bounds__(64 * NWV) void k_knn_mfma(const float *__restrict__ x, int64_t N, int64_t F,
                                                  const float *__restrict__ inv, int k, int exclude_self,
                                                  int tiles_per_split, unsigned long long *__restrict__ part,
                                                  int32_t *__restrict__ out_idx, float *__restrict__ out_sim)
{
    static_assert(!BF3 || FH > 0, "the bf16 form is the register-operand path's");
    static_assert((NWV == 4 && CB == 4) || (BF3 && FH > 0), "other tile shapes: the bf16 register-operand path only");
    static_assert((4 * CB) % NWV == 0 && 4 * CB / NWV >= 1 && 4 * CB / NWV <= 4, "staging: 1..4 vectors per thread and step");
    constexpr int KR = 32 * NWV, KC = 32 * CB;                  // rows per workgroup, columns per tile
    __shared__ float sA[FH > 0 ? 1 : KN_M * KN_LD];
    __shared__ __align__(16) float sB[(FH > 0 ? 2 : 1) * (BF3 ? 3 * KC * KN_PS / 4 : KC * KN_LD)];   // FH > 0: two buffers
    __shared__ unsigned long long s_list[NWV][32][KNN_MAX_K];   // running top-k keys per row
    __shared__ unsigned long long s_thr[NWV][32];               // k-th key per row (0: list not full)
    __shared__ float s_thrf[NWV][32];                           // its cosine (-inf: list not full)
    __shared__ float s_ts[NWV][32];                             // the fast reject's threshold on acc * icol (see the selection)
    __shared__ float s_irow[NWV][32];                           // the rows' inverse norms (a hot register reads LDS, not memory)
    __shared__ int s_pend[NWV][32];                             // candidates parked in list[k .. KNN_MAX_K)
    __shared__ unsigned long long s_cand[NWV][2][128];          // candidates of the two rows of a register
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l32 = lane & 31;
    const int64_t row0 = (int64_t)blockIdx.x * KR;
    for (int q = lane; q < 32 * KNN_MAX_K; q += 64) s_list[wave][q / KNN_MAX_K][q % KNN_MAX_K] = 0ull;
    if (lane < 32) {
        s_thr[wave][lane] = 0ull; s_thrf[wave][lane] = -INFINITY; s_pend[wave][lane] = 0;
        s_ts[wave][lane] = (row0 + wave * 32 + lane < N) ? -INFINITY : INFINITY;      // (rows beyond N: never a candidate)
        s_irow[wave][lane] = (row0 + wave * 32 + lane < N) ? inv[row0 + wave * 32 + lane] : 0.f;
    }
    const int cap = KNN_MAX_K - k;                               // spare slots behind a row's list
    const int sc = tid & 31, sr = tid >> 5;
    const int64_t ncol_tiles = (N + KC - 1) / KC;
    const int64_t ct_begin = (int64_t)blockIdx.y * tiles_per_split;
    const int64_t ct_end = min(ncol_tiles, ct_begin + tiles_per_split);

    float areg[(FH > 0 && !BF3) ? FH : 1];
    sn_u32x4 ap1[BF3 ? FH / 8 : 1], ap2[BF3 ? FH / 8 : 1], ap3[BF3 ? FH / 8 : 1];     // the rows' three bf16 planes
    if constexpr (FH > 0) {
        const int64_t ar = min(row0 + wave * 32 + l32, N - 1);          // (rows >= N are never used)
        if constexpr (!BF3) {
#pragma unroll
            for (int q = 0; q < FH / 4; ++q) {
                const float4 v = *reinterpret_cast<const float4 *>(x + ar * F + half * FH + 4 * q);
                areg[4 * q] = v.x; areg[4 * q + 1] = v.y; areg[4 * q + 2] = v.z; areg[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < FH / 8; ++q) {
                const float4 v0 = *reinterpret_cast<const float4 *>(x + ar * F + half * FH + 8 * q);
                const float4 v1 = *reinterpret_cast<const float4 *>(x + ar * F + half * FH + 8 * q + 4);
                const float av[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                split_bf16x8(av, ap1[q], ap2[q], ap3[q]);
            }
        }
    }

    knn_f4 rb0 = {0.f, 0.f, 0.f, 0.f}, rb1 = rb0, rb2 = rb0, rb3 = rb0;      // the column panel in flight (FH > 0)
    for (int64_t ct = ct_begin; ct < ct_end; ++ct) {
        const int64_t col0 = ct * KC;
        f32x16 acc[CB];
#pragma unroll
        for (int b = 0; b < CB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
        // the tile's column ids and inverse norms (for the selection): requested here, they travel under the products
        float icol[CB];
        int64_t jcol[CB];
#pragma unroll
        for (int b = 0; b < CB; ++b) {
            jcol[b] = col0 + b * 32 + l32;
            icol[b] = inv[min(jcol[b], N - 1)];
        }
#pragma unroll
        for (int b = 0; b < CB; ++b) icol[b] = jcol[b] < N ? icol[b] : 0.f;
        if constexpr (FH > 0) {
            // k-step = 16 steps of each half: columns [s0, s0 + 16) and [FH + s0, FH + s0 + 16) of
            // the column block's rows, as 16-byte vectors: thread t owns vector (t & 7) - four
            // of each half - of rows (t >> 3) + 32 u.  The panel of the NEXT step travels while
            // this one is multiplied, and the first panel of the NEXT TILE while this tile's
            // cosines go through the selection below.  (Named registers: see toolbox.hip,
            // k_cosine_mfma - a private array that lives across the tile loop's back edge goes
            // to scratch memory and its loads are waited for at once.)
            // Round 5: the column panel is DOUBLE-BUFFERED - step L's products read buffer L & 1 while the
            // panel of step L + 1 (requested a step earlier, split here) is written into the other one and the
            // panel of step L + 2 is requested: ONE workgroup barrier per step instead of two, and the staging's
            // vector work (the bf16 split: ~100 instructions per step and thread) sits in the same block as the
            // step's 64 matrix instructions, where the wave issues it beside them (one wave per SIMD: nobody
            // else would).  The steps run on across tile boundaries (the next tile's first panel is staged
            // under this tile's last products and waits in LDS during the selection).
            // Staging map: 16 consecutive lanes write rows r and r + 4 (not r and r + 1): with the 80-byte
            // plane rows their 8-byte stores cover all 32 banks once - adjacent rows overlapped in four
            // (SQ_LDS_BANK_CONFLICT 777 M cycles at arxiv size, 1.24 per LDS instruction, all from these stores).
            const int seg = tid & 7, slot = tid >> 3;
            const int pr = (slot & ~7) | ((slot & 7) >> 1) | ((slot & 1) << 2);
            const int kseg = seg < 4 ? 4 * seg : FH + 4 * (seg - 4);
            constexpr int NS = FH / 16;                                   // steps per tile
            constexpr int PANEL = BF3 ? 3 * KC * KN_PS : KC * KN_LD * 4;     // bytes of one buffer
            constexpr int SROWS = 8 * NWV, UV = KC / SROWS;               // rows staged per vector slot, vectors per thread
#define SN_KNN_FETCH(COL0, S0)                                                                            \
            {                                                                                             \
                const float *g_ = x + (S0) + kseg;                                                        \
                rb0 = *(const knn_f4 *)(g_ + min((COL0) + pr, N - 1) * F);                                \
                if constexpr (UV > 1) rb1 = *(const knn_f4 *)(g_ + min((COL0) + pr + SROWS, N - 1) * F);  \
                if constexpr (UV > 2) rb2 = *(const knn_f4 *)(g_ + min((COL0) + pr + 2 * SROWS, N - 1) * F); \
                if constexpr (UV > 3) rb3 = *(const knn_f4 *)(g_ + min((COL0) + pr + 3 * SROWS, N - 1) * F); \
            }
            unsigned char *sBb = reinterpret_cast<unsigned char *>(sB);
            // BF3: plane p of column row r at byte (p * KN_M + r) * KN_PS; its 32 k-slots are the
            // step's 16 of half 0 followed by the 16 of half 1 (seg 0..3 | 4..7, 8 bytes each)
            auto stage = [&](int boff) {
                if constexpr (!BF3) {
                    float *wb = reinterpret_cast<float *>(sBb + boff) + pr * KN_LD + 4 * seg;
                    *(knn_f4 *)(wb) = rb0;
                    if constexpr (UV > 1) *(knn_f4 *)(wb + SROWS * KN_LD) = rb1;
                    if constexpr (UV > 2) *(knn_f4 *)(wb + 2 * SROWS * KN_LD) = rb2;
                    if constexpr (UV > 3) *(knn_f4 *)(wb + 3 * SROWS * KN_LD) = rb3;
                } else {
                    unsigned char *wb3 = sBb + boff + pr * KN_PS + 8 * seg;
                    auto put = [&](int u, const knn_f4 &v) {
                        const float vv[4] = {v[0], v[1], v[2], v[3]};
                        sn_u32x2 p1, p2, p3;
                        split_bf16x4(vv, p1, p2, p3);
                        unsigned char *d_ = wb3 + SROWS * u * KN_PS;
                        *(sn_u32x2 *)(d_) = p1;
                        *(sn_u32x2 *)(d_ + KC * KN_PS) = p2;
                        *(sn_u32x2 *)(d_ + 2 * KC * KN_PS) = p3;
                    };
                    put(0, rb0);
                    if constexpr (UV > 1) put(1, rb1);
                    if constexpr (UV > 2) put(2, rb2);
                    if constexpr (UV > 3) put(3, rb3);
                }
            };
            // linear step L = (ct - ct_begin) NS + s0 / 16 lives in buffer L & 1
            auto fetch_step = [&](int64_t L) {
                const int64_t t_ = ct_begin + L / NS;
                const int st_ = (int)(L % NS) * 16;
                if (t_ < ct_end) SN_KNN_FETCH(t_ * KC, st_)
            };
            if (ct == ct_begin) {
                fetch_step(0);
                stage(0);
                fetch_step(1);
                __syncthreads();
            }
            const int64_t L0 = (ct - ct_begin) * NS;
#pragma unroll
            for (int s0 = 0; s0 < FH; s0 += 16) {
                const int64_t L = L0 + s0 / 16;
                const int cur = (int)(L & 1) * PANEL;
                if (s0 + 16 < FH || ct + 1 < ct_end) stage(cur ^ PANEL);      // step L + 1's panel (in the registers)
                fetch_step(L + 2);
                const float *pb = reinterpret_cast<const float *>(sBb + cur) + l32 * KN_LD + half * 16;
                const unsigned char *pb3 = sBb + cur + l32 * KN_PS + 32 * half;
                if constexpr (BF3) {
#pragma unroll
                    for (int g = 0; g < 2; ++g) {          // 8 k-slots of each half per MFMA
                        const int aq = s0 / 8 + g;
#pragma unroll
                        for (int b = 0; b < CB; ++b) {
                            const unsigned char *r_ = pb3 + 32 * b * KN_PS + 16 * g;
                            const sn_u32x4 b1 = *(const sn_u32x4 *)(r_);
                            const sn_u32x4 b2 = *(const sn_u32x4 *)(r_ + KC * KN_PS);
                            const sn_u32x4 b3 = *(const sn_u32x4 *)(r_ + 2 * KC * KN_PS);
#if defined(SNGNN_KNN_EXP) && SNGNN_KNN_EXP == 2        // timing experiment: one product of eight
#define SN_KNN_M3(PA, PB) asm volatile("" ::"v"(PA), "v"(PB));
                            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sn_bf16x8, ap1[aq]),
                                                                             __builtin_bit_cast(sn_bf16x8, b1), acc[b], 0, 0, 0);
#else
#define SN_KNN_M3(PA, PB)                                                                                   \
                            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sn_bf16x8, PA), \
                                                                             __builtin_bit_cast(sn_bf16x8, PB), acc[b], 0, 0, 0);
#endif
                            SN_KNN_M3(ap3[aq], b2) SN_KNN_M3(ap2[aq], b3) SN_KNN_M3(ap3[aq], b1) SN_KNN_M3(ap2[aq], b2)
                            SN_KNN_M3(ap1[aq], b3) SN_KNN_M3(ap2[aq], b1) SN_KNN_M3(ap1[aq], b2) SN_KNN_M3(ap1[aq], b1)
#undef SN_KNN_M3
                        }
                    }
                } else
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const knn_f4 b0 = *(const knn_f4 *)(pb + 4 * q), b1 = *(const knn_f4 *)(pb + 32 * KN_LD + 4 * q);
                    const knn_f4 b2 = *(const knn_f4 *)(pb + 64 * KN_LD + 4 * q), b3 = *(const knn_f4 *)(pb + 96 * KN_LD + 4 * q);
#define SN_KNN_MFMA(E, SS)                                                                                \
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[s0 + 4 * q + SS], b0.E, acc[0], 0, 0, 0);  \
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[s0 + 4 * q + SS], b1.E, acc[1], 0, 0, 0);  \
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[s0 + 4 * q + SS], b2.E, acc[2], 0, 0, 0);  \
                    acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[s0 + 4 * q + SS], b3.E, acc[3], 0, 0, 0);
                    SN_KNN_MFMA(x, 0) SN_KNN_MFMA(y, 1) SN_KNN_MFMA(z, 2) SN_KNN_MFMA(w, 3)
#undef SN_KNN_MFMA
                }
                __syncthreads();       // step L's reads are done, step L + 1's panel is complete
            }
#undef SN_KNN_FETCH
        } else {
        float ra[KN_LOADS], rb[KN_LOADS];
        auto fetch = [&](int64_t k0) {
            const int64_t kk = k0 + sc;
#pragma unroll
            for (int u = 0; u < KN_LOADS; ++u) {
                const int64_t r_a = row0 + sr + 8 * u, r_b = col0 + sr + 8 * u;
                ra[u] = (r_a < N && kk < F) ? x[r_a * F + kk] : 0.f;
                rb[u] = (r_b < N && kk < F) ? x[r_b * F + kk] : 0.f;
            }
        };
        fetch(0);
        for (int64_t k0 = 0; k0 < F; k0 += KN_K) {
            __syncthreads();
#pragma unroll
            for (int u = 0; u < KN_LOADS; ++u) {
                sA[(sr + 8 * u) * KN_LD + sc] = ra[u];
                sB[(sr + 8 * u) * KN_LD + sc] = rb[u];
            }
            __syncthreads();
            if (k0 + KN_K < F) fetch(k0 + KN_K);
#pragma unroll
            for (int kk = 0; kk < KN_K; kk += 2) {
                // 32x32x2: lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]
                const float a = sA[(wave * 32 + l32) * KN_LD + kk + half];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float bv = sB[(b * 32 + l32) * KN_LD + kk + half];
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[b], 0, 0, 0);
                }
            }
        }
        }   // FH == 0
#if defined(SNGNN_KNN_EXP) && SNGNN_KNN_EXP == 1        // timing experiment: no selection at all
#pragma unroll
        for (int b = 0; b < CB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[b][r]));
        continue;
#endif
        // ---- selection: C/D layout col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
        // Round 5: the fast reject of ALL sixteen registers first, from thresholds fetched together.  Register r
        // holds two rows of the wave's 32 (one per half-wave), each row sits in exactly one register, so the
        // sixteen thresholds of a tile can be read up front: one LDS round trip instead of sixteen serial ones,
        // and no read of the rows' inverse norms at all - s_ts[row] is the row's k-th cosine DIVIDED by its
        // inverse norm, lowered by a relative 1e-6 (a superset test: `acc * icol >= ts` holds whenever the exact
        // `acc * (irow * icol) >= thrf` below does; the exact rule still decides).  Before, every register cost a
        // dependent LDS read and a global read of inv[i]: the selection was 34 of the kernel's 88 ms at arxiv
        // size (a build without any selection: 54 ms), 26 of 80 now.  What is left are the ~3.6 registers per tile
        // and wave that do hold a candidate (k ln(N / k) insertions per row over the scan) and their LDS round
        // trips, which nothing hides at one wave per SIMD.  Measured and dropped: the candidates appended by their
        // lanes to a wave-private LDS queue (LDS atomic) and worked off one by one - 85.6 ms against 80.0 (64
        // divergent tests per tile cost more than 16 ballots; an entry still costs three dependent LDS reads).
        float tsr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) tsr[r] = s_ts[wave][(r & 3) + 8 * (r >> 2) + 4 * half];
        unsigned hot16 = 0u;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            bool m_ = false;
#pragma unroll
            for (int b = 0; b < CB; ++b) m_ |= acc[b][r] * icol[b] >= tsr[r];
            if (__ballot(m_) != 0ull) hot16 |= 1u << r;
        }
        SN_KNN_COUNT(0, 1);                                             // tiles x waves
        SN_KNN_COUNT(1, __popc(hot16));                                 // registers that pass the superset test
        // ONE copy of the hot path's code, looped over the set bits: unrolled per register it was sixteen copies of
        // ~700 instructions each (the merges inlined) - far more than the instruction cache holds, and the ~2 hot
        // registers of a tile jump to different copies every time.  Only the four accumulator values are fetched
        // per register (a switch of sixteen four-move cases: accumulator indices must be literals).
        unsigned hm_ = hot16;
#if defined(SNGNN_KNN_EXP) && SNGNN_KNN_EXP == 5        // timing experiment: the fast test of every register, no hot path
        asm volatile("" ::"s"(hm_));
        hm_ = 0u;
#endif
        while (hm_ != 0u) {
            const int r = __builtin_ctz(hm_);                            // (wave-uniform)
            hm_ &= hm_ - 1u;
            float ar[CB];
            switch (r) {
#define SN_KNN_PICK(R) case R: _Pragma("unroll") for (int b_ = 0; b_ < CB; ++b_) ar[b_] = acc[b_][R]; break;
            SN_KNN_PICK(0) SN_KNN_PICK(1) SN_KNN_PICK(2) SN_KNN_PICK(3) SN_KNN_PICK(4) SN_KNN_PICK(5) SN_KNN_PICK(6) SN_KNN_PICK(7)
            SN_KNN_PICK(8) SN_KNN_PICK(9) SN_KNN_PICK(10) SN_KNN_PICK(11) SN_KNN_PICK(12) SN_KNN_PICK(13) SN_KNN_PICK(14)
            default: _Pragma("unroll") for (int b_ = 0; b_ < CB; ++b_) ar[b_] = acc[b_][15]; break;
#undef SN_KNN_PICK
            }
            const int lr = (r & 3) + 8 * (r >> 2) + 4 * half;          // this lane's row within the wave's 32
            const int64_t i = row0 + wave * 32 + lr;
            const float irow = s_irow[wave][lr];
            // fast reject on the cosine alone: nothing of this register reaches its row's
            // k-th value - the common case once the lists have warmed up
            const float thrf = s_thrf[wave][lr];
            float sv[CB];
            bool maybe = false;
#pragma unroll
            for (int b = 0; b < CB; ++b) {
                sv[b] = ar[b] * (irow * icol[b]) + 0.0f;
                maybe |= sv[b] >= thrf;
            }
            if (__ballot(maybe) == 0ull) continue;
            const unsigned long long thr = s_thr[wave][lr];
            unsigned long long key[CB];
            bool any = false;
#pragma unroll
            for (int b = 0; b < CB; ++b) {
                const bool ok = i < N && jcol[b] < N && !(exclude_self && jcol[b] == i);
                key[b] = ok ? sel_key(sv[b], (unsigned)jcol[b]) : 0ull;
                if (key[b] <= thr) key[b] = 0ull;                        // cannot enter the list
                any |= key[b] != 0ull;
            }
            const unsigned long long hot = __ballot(any);
            if (hot == 0ull) continue;
            SN_KNN_COUNT(2, 1);                                         // registers with a real candidate
            // The common hot register after warm-up holds ONE candidate in the whole wave: its lane parks it behind
            // its row's list by itself - no ballots, no compaction through LDS, no per-half hand-off (a full
            // spare area falls through to the general path below, which merges).
            if (__popcll(hot) == 1) {
                int nk_ = 0;
                unsigned long long kor_ = 0ull;
#pragma unroll
                for (int b = 0; b < CB; ++b) { nk_ += key[b] != 0ull; kor_ |= key[b]; }
                if (__builtin_amdgcn_readlane(nk_, __ffsll((long long)hot) - 1) == 1) {
                    bool parked = false;
                    if (any) {
                        const int pc_ = s_pend[wave][lr];
                        if (pc_ < cap) {
                            s_list[wave][lr][k + pc_] = kor_;
                            s_pend[wave][lr] = pc_ + 1;
                            parked = true;
                        }
                    }
                    if (__ballot(parked) != 0ull) {
                        SN_KNN_COUNT(3, 1);                             // ... parked by their one lane
                        wave_lds_sync();
                        continue;
                    }
                }
            }
            // compact the candidates of the two rows (one per half-wave) into LDS
            int cnt = 0;                                                 // per half
#pragma unroll
            for (int b = 0; b < CB; ++b) {
                const unsigned long long m = __ballot(key[b] != 0ull);
                const unsigned mh = half ? (unsigned)(m >> 32) : (unsigned)m;
                if (key[b] != 0ull) s_cand[wave][half][cnt + __popc(mh & ((1u << l32) - 1u))] = key[b];
                cnt += __popc(mh);
            }
            wave_lds_sync();
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int c = __builtin_amdgcn_readlane(cnt, h * 32);
                if (c == 0) continue;
                const int rowh = (r & 3) + 8 * (r >> 2) + 4 * h;
                unsigned long long *cand = s_cand[wave][h];
                unsigned long long *list = s_list[wave][rowh];
                const int pc = s_pend[wave][rowh];
                if (pc + c <= cap) {
                    // park them behind the list: the merge waits until the spare slots are full
                    // (the row's threshold goes stale meanwhile - it only lets more through)
                    if (lane < c) list[k + pc + lane] = cand[lane];
                    if (lane == 0) s_pend[wave][rowh] = pc + c;
                    continue;
                }
                if (c + pc + k > 128) {
                    // too many for one merge (first tiles only): keep the best k candidates
                    const unsigned long long c0 = lane < c ? cand[lane] : 0ull, c1 = lane + 64 < c ? cand[lane + 64] : 0ull;
                    const unsigned long long T = kth_largest(c0, c1, k);
                    const bool k0 = c0 != 0ull && c0 >= T, k1 = c1 != 0ull && c1 >= T;
                    const unsigned long long m0 = __ballot(k0), m1 = __ballot(k1);
                    wave_lds_sync();
                    if (k0) cand[prefix_popc(m0)] = c0;
                    if (k1) cand[__popcll(m0) + prefix_popc(m1)] = c1;
                    wave_lds_sync();
                    c = k;
                }
                if (lane < pc) cand[c + lane] = list[k + lane];        // the parked ones join
                wave_lds_sync();
                const unsigned long long T = merge_row(list, cand, c + pc, k);
                SN_KNN_COUNT(4, 1);                                     // merges
                if (lane == 0) {
                    s_pend[wave][rowh] = 0;
                    s_thr[wave][rowh] = T;
                    const unsigned u = (unsigned)(T >> 32);
                    const float tf = T ? __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u) : -INFINITY;
                    s_thrf[wave][rowh] = tf;
                    // (rows beyond N never get here: their lists take no key)
                    const float tq = tf / s_irow[wave][rowh];
                    s_ts[wave][rowh] = tq - (fabsf(tq) * 1e-6f + 1e-30f);
                }
            }
            wave_lds_sync();
        }
    }
    // parked candidates of every row
    wave_lds_sync();
    for (int lr = 0; lr < 32; ++lr) {
        const int pc = s_pend[wave][lr];
        if (pc == 0) continue;                                           // wave-uniform (LDS value)
        unsigned long long *cand = s_cand[wave][0];
        if (lane < pc) cand[lane] = s_list[wave][lr][k + lane];
        wave_lds_sync();
        merge_row(s_list[wave][lr], cand, pc, k);
    }
    // ---- output: the lists in rank order, or as keys for the merge of the column splits
    wave_lds_sync();
    for (int lr = 0; lr < 32; ++lr) {
        const int64_t i = row0 + wave * 32 + lr;
        if (i >= N) break;
        const unsigned long long *list = s_list[wave][lr];
        if (part != nullptr) {
            if (lane < k) part[((size_t)blockIdx.y * N + i) * k + lane] = list[lane];
            continue;
        }
        if (lane < k) {
            const unsigned long long kq = list[lane];
            int rk = 0, n = 0;
            for (int q = 0; q < k; ++q) {
                const unsigned long long o = list[q];
                rk += o > kq;
                n += o != 0ull;
            }
            if (kq != 0ull) {
                const unsigned u = (unsigned)(kq >> 32);
                out_idx[i * k + rk] = (int32_t)(0xFFFFFFFFu - (unsigned)kq);
                out_sim[i * k + rk] = __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
            }
            if (lane >= n) {               // fewer than k eligible nodes: pad
                out_idx[i * k + lane] = -1;
                out_sim[i * k + lane] = 0.f;
            }
        }
    }
}

// top-k of the column splits' lists of one row (nsplit * k keys, 128 at a time)
__global__ __launch_bounds__(256) void k_knn_merge(const unsigned long long *__restrict__ part, int64_t N,
                                                   int k, int nsplit, int32_t *__restrict__ out_idx,
                                                   float *__restrict__ out_sim)
{
    __shared__ unsigned long long s_list[4][KNN_MAX_K];
    __shared__ unsigned long long s_cand[4][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    if (i >= N) return;                                   // wave-uniform
    if (lane < KNN_MAX_K) s_list[wave][lane] = 0ull;
    wave_lds_sync();
    const int per = (128 - k) / k;                        // splits merged per round
    for (int s0 = 0; s0 < nsplit; s0 += per) {
        const int ns = min(per, nsplit - s0), cnt = ns * k;
        for (int q = lane; q < cnt; q += 64)
            s_cand[wave][q] = part[((size_t)(s0 + q / k) * N + i) * k + q % k];
        wave_lds_sync();
        merge_row(s_list[wave], s_cand[wave], cnt, k);    // (zero keys of short lists are ignored)
    }
    const unsigned long long *list = s_list[wave];
    if (lane < k) {
        const unsigned long long kq = list[lane];
        int rk = 0, n = 0;
        for (int q = 0; q < k; ++q) {
            const unsigned long long o = list[q];
            rk += o > kq;
            n += o != 0ull;
        }
        if (kq != 0ull) {
            const unsigned u = (unsigned)(kq >> 32);
            out_idx[i * k + rk] = (int32_t)(0xFFFFFFFFu - (unsigned)kq);
            out_sim[i * k + rk] = __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
        }
        if (lane >= n) {
            out_idx[i * k + lane] = -1;
            out_sim[i * k + lane] = 0.f;
        }
    }
}

// ---------------------------------------------------------------------------
// Small graphs: materialise + select.  When S fits comfortably (N <= KNN_DENSE_MAX_N) the dense cosine
// kernel (toolbox.hip: upper triangle on the matrix cores, mirrored) followed by this row selection
// beats the fused scan - whose both-panels-through-LDS path (wide features) pays eight barriers per
// tile and whose column splits (few row blocks) each warm their lists up from empty: Chameleon
// (2 277 x 2 325) and Actor (7 600 x 932) - BASELINE configs 2 and 3 - are this case.
// One wave per row of S: 256 entries per step (16 bytes per lane), an entry enters the selection
// only when its key beats the row's current k-th key; survivors are parked behind the list in
// LDS and merged when the buffer is full (the aggregation's early-exit k-th-key search).
// Same keys, same order and padding as the fused kernel.
// ---------------------------------------------------------------------------
constexpr int64_t KNN_DENSE_MAX_N = 32768;       // S = 4.3 GB at most (a stream-ordered allocation)

__global__ __launch_bounds__(256) void k_row_topk(const float *__restrict__ S, int64_t N, int k, int exclude_self,
                                                  int lowbits, int32_t *__restrict__ out_idx,
                                                  float *__restrict__ out_sim)
{
    __shared__ unsigned long long s_key[4][128];          // [0, k): the list (0 = empty); [k, k + pend): parked
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    if (i >= N) return;                                   // wave-uniform
    unsigned long long *keys = s_key[wave];
    keys[lane] = 0ull;
    keys[lane + 64] = 0ull;
    const int cap = 128 - k;
    int pend = 0;
    unsigned long long T = 0ull;                          // the list's k-th key (0: not full yet)
    auto merge = [&]() {
        wave_lds_sync();
        const unsigned long long key[2] = {keys[lane], keys[lane + 64]};
        bool kept[2];
        wave_topk_keys_n<2>(key, k, lowbits, kept);
        const unsigned long long m0 = __ballot(kept[0]), m1 = __ballot(kept[1]);
        const int n0 = __popcll(m0), ns = n0 + __popcll(m1);
        // the new threshold: the smallest kept key once the list is full
        unsigned long long mn = ~0ull;
        if (kept[0]) mn = key[0];
        if (kept[1] && key[1] < mn) mn = key[1];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const unsigned long long o = __shfl_xor(mn, m, 64);
            mn = o < mn ? o : mn;
        }
        wave_lds_sync();                                  // every lane has read the old contents
        keys[lane] = 0ull;
        keys[lane + 64] = 0ull;
        wave_lds_sync();
        if (kept[0]) keys[prefix_popc(m0)] = key[0];
        if (kept[1]) keys[n0 + prefix_popc(m1)] = key[1];
        wave_lds_sync();
        T = ns >= k ? mn : 0ull;
        pend = 0;
    };
    const float *row = S + i * N;
    const bool vec = (N % 4 == 0) && ((uintptr_t)S % 16 == 0);
    for (int64_t base = 0; base < N; base += 256) {
        const int64_t c0 = base + lane * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (vec && c0 + 3 < N) {
            const float4 t = *reinterpret_cast<const float4 *>(row + c0);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (c0 + u < N) v[u] = row[c0 + u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t c = c0 + u;
            const bool valid = c < N && !(exclude_self && c == i);
            const unsigned long long key = valid ? sel_key(v[u] + 0.0f, (unsigned)c) : 0ull;
            const bool pass = key > T;                    // (an invalid slot's key 0 never passes)
            unsigned long long m = __ballot(pass);
            if (m == 0ull) continue;                      // wave-uniform: the common case after the first steps
            if (pend + __popcll(m) > cap) merge();        // (at most 64 new ones, cap >= 96)
            const bool still = pass && key > T;           // the merge may have raised the threshold
            m = __ballot(still);
            if (still) keys[k + pend + prefix_popc(m)] = key;
            pend += __popcll(m);
        }
    }
    if (pend > 0) merge();
    wave_lds_sync();
    if (lane < k) {
        const unsigned long long kq = keys[lane];
        int rk = 0, n = 0;
        for (int q = 0; q < k; ++q) {
            const unsigned long long o = keys[q];
            rk += o > kq;
            n += o != 0ull;
        }
        if (kq != 0ull) {
            const unsigned u = (unsigned)(kq >> 32);
            out_idx[i * k + rk] = (int32_t)(0xFFFFFFFFu - (unsigned)kq);
            out_sim[i * k + rk] = __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
        }
        if (lane >= n) {               // fewer than k eligible nodes: pad
            out_idx[i * k + lane] = -1;
            out_sim[i * k + lane] = 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void k_knn_inv_norm(const float *__restrict__ x, int64_t N, int64_t F,
                                                      float *__restrict__ inv)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const float *p = x + row * F;
    float ss = 0.f;
    for (int64_t c = lane; c < F; c += 64) ss = fmaf(p[c], p[c], ss);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) ss += __shfl_xor(ss, m, 64);
    if (lane == 0) inv[row] = 1.0f / fmaxf(sqrtf(ss), EPS_NORM);
}

}  // namespace sngnn

using namespace sngnn;

// 0 = by shape (default), 1 = the fused scan always, 2 = materialise + select always (measurement: sngnn_tuning_set(6, v))
// (value 3 / 4 - measurement: the fused scan with the 128 x 128 / the 256 x 64 tile shape whatever the width allows)
static int g_knn_route = 0, g_knn_shape = 0;
namespace sngnn {
int set_knn_route(int v)
{
    if (v < 0 || v > 4) return SNGNN_EINVAL;
    g_knn_shape = v == 3 ? 1 : 0;            // 1: never the <8, 2> shape
    g_knn_route = v >= 3 ? 1 : v;
    return SNGNN_OK;
}
}

// column splits: enough workgroups to fill the chip when there are few row blocks
// (F in {32, 64, 96, 128} with 16-byte rows, bf16 products: the <8, 2> shape - 256 rows per workgroup, 64-column tiles,
// two waves per SIMD: 76 -> 58 ms at arxiv size)
static bool knn_wide(int64_t F, const float *x)
{
    return (F == 128 || F == 96 || F == 64 || F == 32) && (uintptr_t)x % 16 == 0 && !sngnn::fp32_mfma_only() && g_knn_shape != 1;
}
static int knn_splits(int64_t N, int rows_per_wg = KN_M)
{
    const int64_t nrb = (N + rows_per_wg - 1) / rows_per_wg;
    // (every split warms its lists up from empty: only as many as it takes to occupy the CUs)
    return (int)std::max<int64_t>(1, std::min<int64_t>(nrb, (320 + nrb - 1) / nrb));
}

extern "C" int64_t sngnn_knn_workspace_bytes(int64_t N, int k)
{
    const int ns = std::max(knn_splits(N), knn_splits(N, 256));       // (either tile shape)
    return (N + 63) / 64 * 256 + (ns > 1 ? (int64_t)ns * N * k * 8 : 0) + 256;
}

extern "C" int sngnn_knn_graph(const float *x, int64_t N, int64_t F, int k, int exclude_self,
                               int32_t *nbr_idx, float *nbr_sim, void *workspace, void *stream)
{
    SN_REQUIRE(N >= 0 && F >= 1, SNGNN_EINVAL, "bad shape");
    SN_REQUIRE(k >= 1 && k <= KNN_MAX_K, SNGNN_EINVAL, "k must be in [1, " + std::to_string(KNN_MAX_K) + "]");
    SN_REQUIRE(N < ((int64_t)1 << 31), SNGNN_EINVAL, "too many rows");
    if (N == 0) return SNGNN_OK;
    SN_REQUIRE(x && nbr_idx && nbr_sim && workspace, SNGNN_EINVAL, "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    // (narrow features too: at 20 000 x 128 the fused scan's register-operand path takes 3.1 ms - few row
    // blocks, lists warmed up per column split - against ~1.2 ms for S + selection)
    if (g_knn_route == 2 || (g_knn_route == 0 && N <= KNN_DENSE_MAX_N && N >= 2)) {
        // materialise + select (see k_row_topk); S lives in a stream-ordered allocation
        SN_REQUIRE(N <= 65535, SNGNN_EINVAL, "the dense route is for small graphs");
        // (outside the caller's allocator: up to 4.3 GB.  When the device cannot give it - the caller's pool holds
        // most of HBM - the fused scan below does the same job from the few MB of the workspace, unless the
        // dense route was forced)
        void *S = nullptr;
        const hipError_t got = hipMallocAsync(&S, (size_t)N * N * 4, st);
        if (got == hipSuccess) {
            int rc = sngnn_cosine_dense(x, N, F, (float *)S, stream);
            if (rc == SNGNN_OK) {
                int lowbits = 1;
                while (((int64_t)1 << lowbits) < N && lowbits < 31) ++lowbits;
                k_row_topk<<<(unsigned)((N + 3) / 4), 256, 0, st>>>((const float *)S, N, k, exclude_self, lowbits, nbr_idx, nbr_sim);
                if (hipGetLastError() != hipSuccess) rc = SNGNN_EHIP;
            }
            (void)hipFreeAsync(S, st);
            return rc;
        }
        (void)hipGetLastError();          // clear the allocation failure
        SN_REQUIRE(g_knn_route != 2, SNGNN_ENOMEM, "out of device memory for the dense route's N x N matrix");
    }
    float *inv = (float *)workspace;
    unsigned long long *part = (unsigned long long *)((char *)workspace + (N + 63) / 64 * 256);
    const bool wide = knn_wide(F, x);
    const int rows_wg = wide ? 256 : KN_M, cols_tile = wide ? 64 : KN_M;
    const int ns = knn_splits(N, rows_wg);
    const int64_t nrb = (N + rows_wg - 1) / rows_wg, nct = (N + cols_tile - 1) / cols_tile;
    const int tps = (int)((nct + ns - 1) / ns);          // column tiles per split
    const int ns_used = (int)((nct + tps - 1) / tps);
    k_knn_inv_norm<<<(unsigned)((N + 3) / 4), 256, 0, st>>>(x, N, F, inv);
    dim3 grid((unsigned)nrb, (unsigned)ns_used);
    unsigned long long *pp = ns_used > 1 ? part : nullptr;
    const bool areg = (uintptr_t)x % 16 == 0;
    const bool bf3 = !sngnn::fp32_mfma_only();          // sngnn_tuning_set(5, 1): fp32 MFMAs
    if (wide && F == 128) k_knn_mfma<64, true, 8, 2><<<grid, 512, 0, st>>>(x, N, F, inv, k, exclude_self, tps, pp, nbr_idx, nbr_sim);
    else if (wide && F == 96) k_knn_mfma<48, true, 8, 2><<<grid, 512, 0, st>>>(x, N, F, inv, k, exclude_self, tps, pp, nbr_idx, nbr_sim);
    else if (wide && F == 64) k_knn_mfma<32, true, 8, 2><<<grid, 512, 0, st>>>(x, N, F, inv, k, exclude_self, tps, pp, nbr_idx, nbr_sim);
    else if (wide) k_knn_mfma<16, true, 8, 2><<<grid, 512, 0, st>>>(x, N, F, inv, k, exclude_self, tps, pp, nbr_idx, nbr_sim);
    else if (areg && F == 128) { if (bf3) k_knn_mfma<64, true><<<grid, 256, 0, st>>>(x, N, F, inv, k, exclude_self, tps, pp, nbr_idx, nbr_sim); else k_knn_mfma<64, false><<<grid, 256, 0, st>>>(x, N, F, inv, k, exclude_self, tps, pp, nbr_idx, nbr_sim); }
    else if (areg && F == 96) { if (bf3) k_knn_mfma<48, true><<<grid, 256, 0, st>>>(x, N, F, inv, k, exclude_self, tps, pp, nbr_idx, nbr_sim); else k_knn_mfma<48, false><<<grid, 256, 0, st>>>(x, N, F, inv, k, exclude_self, tps, pp, nbr_idx, nbr_sim); }
    else if (areg && F == 64) { if (bf3) k_knn_mfma<32, true><<<grid, 256, 0, st>>>(x, N, F, inv, k, exclude_self, tps, pp, nbr_idx, nbr_sim); else k_knn_mfma<32, false><<<grid, 256, 0, st>>>(x, N, F, inv, k, exclude_self, tps, pp, nbr_idx, nbr_sim); }
    else if (areg && F == 32) { if (bf3) k_knn_mfma<16, true><<<grid, 256, 0, st>>>(x, N, F, inv, k, exclude_self, tps, pp, nbr_idx, nbr_sim); else k_knn_mfma<16, false><<<grid, 256, 0, st>>>(x, N, F, inv, k, exclude_self, tps, pp, nbr_idx, nbr_sim); }
    else k_knn_mfma<0, false><<<grid, 256, 0, st>>>(x, N, F, inv, k, exclude_self, tps, pp, nbr_idx, nbr_sim);
    if (ns_used > 1) k_knn_merge<<<(unsigned)((N + 3) / 4), 256, 0, st>>>(part, N, k, ns_used, nbr_idx, nbr_sim);
    SN_HIP(hipGetLastError());
    return SNGNN_OK;
}

#if defined(SNGNN_KNN_EXP) && SNGNN_KNN_EXP == 4
extern "C" int sngnn_knn_debug_counters(unsigned long long *out8, int reset)
{
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (out8) SN_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(sngnn::g_knn_dbg), sizeof(z)));
    if (reset) SN_HIP(hipMemcpyToSymbol(HIP_SYMBOL(sngnn::g_knn_dbg), z, sizeof(z)));
    return SNGNN_OK;
}
#endif
