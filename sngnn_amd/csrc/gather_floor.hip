// Measurement aid (no reference counterpart): the memory work of the aggregation forward's main
// kernel WITHOUT its arithmetic - the floor bench.py prints beside the kernel's time
// (`roofline.gather_floor_ms`, `kernel_over_floor`).
//
// sngnn_gather_floor runs, over the CALL's own graph and feature table,
//   mode 0: the bare gather - one coalesced 4C-byte row read per entry of the graph's own `col`
//           (CSR order: what the forward must read whatever it computes), the rows folded into one
//           checksum word per wave so that nothing is dead code;
//   mode 1: the same plus, per owned node, its own row read and an output row written (the
//           N (8C + 8) term of SURVEY.md 8d's byte model): all of B_fwd's traffic, none of its work.
// No selection, no ranking, no norms, no second fetch of kept rows: the time is what the memory
// system needs for this access pattern on this chip, measured on the box that runs the bench.
#include "common.h"

using namespace sngnn;

namespace sngnn {

constexpr int GF_U = 4;          // rows in flight per lane group (the forward's own depth)

template <int G>
__global__ __launch_bounds__(BLOCK) void k_gather_floor(const float *__restrict__ table, const int32_t *__restrict__ col,
                                                        int64_t E, int C, int64_t N, int64_t row_off, int mode,
                                                        float *__restrict__ out, float *__restrict__ sink)
{
    constexpr int RPI = 64 / G;                                   // rows per wave-load
    const int lane = threadIdx.x & 63, gid = lane / G, lg = lane % G;
    const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6, nw = (int64_t)gridDim.x * WAVES;
    const bool act = lg * 4 < C;
    const int off = act ? lg * 4 : 0;                              // (unconditional loads: a clamped address)
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t base = wave * RPI * GF_U; base < E; base += nw * RPI * GF_U) {
        float4 x[GF_U];
#pragma unroll
        for (int u = 0; u < GF_U; ++u) {
            const int64_t e = base + u * RPI + gid;
            const int32_t j = col[e < E ? e : E - 1];
            x[u] = *reinterpret_cast<const float4 *>(table + (size_t)j * C + off);
        }
#pragma unroll
        for (int u = 0; u < GF_U; ++u) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
    }
    if (mode == 1) {
        for (int64_t r = wave * RPI + gid; r < N; r += nw * RPI) {
            const float4 v = *reinterpret_cast<const float4 *>(table + (size_t)(r + row_off) * C + off);
            if (act) *reinterpret_cast<float4 *>(out + (size_t)r * C + off) = v;
        }
    }
    float s = act ? (acc.x + acc.y) + (acc.z + acc.w) : 0.f;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) sink[wave] = s;
}

}  // namespace sngnn

extern "C" int64_t sngnn_gather_floor_workspace_bytes(void)
{
    return (int64_t)256 * 8 * WAVES * 4;                          // one checksum word per wave of the grid
}

extern "C" int sngnn_gather_floor(const sngnn_graph_t *g, const float *table, int C, int mode, float *out,
                                  void *workspace, void *stream)
{
    SN_REQUIRE(g != nullptr, SNGNN_EINVAL, "graph is NULL");
    SN_REQUIRE(mode == 0 || mode == 1, SNGNN_EINVAL, "mode must be 0 (edges only) or 1 (edges + own rows + stores)");
    SN_REQUIRE(C >= 4 && C <= 256 && C % 4 == 0, SNGNN_EINVAL, "the gather floor is measured on 16-byte rows: C % 4 == 0, C <= 256");
    if (g->Ep == 0 && (mode == 0 || g->N == 0)) return SNGNN_OK;
    SN_REQUIRE(table != nullptr && workspace != nullptr && (mode == 0 || out != nullptr), SNGNN_EINVAL, "NULL argument");
    SN_REQUIRE(((uintptr_t)table % 16) == 0 && ((uintptr_t)out % 16) == 0, SNGNN_EINVAL, "table / out must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int grid = 256 * 8;                                      // 8 workgroups per CU: 8 waves per SIMD
    float *sink = (float *)workspace;
    const int lanes = C / 4;
    if (lanes <= 8) k_gather_floor<8><<<grid, BLOCK, 0, st>>>(table, g->col, g->Ep, C, g->N, g->row_off, mode, out, sink);
    else if (lanes <= 16) k_gather_floor<16><<<grid, BLOCK, 0, st>>>(table, g->col, g->Ep, C, g->N, g->row_off, mode, out, sink);
    else if (lanes <= 32) k_gather_floor<32><<<grid, BLOCK, 0, st>>>(table, g->col, g->Ep, C, g->N, g->row_off, mode, out, sink);
    else k_gather_floor<64><<<grid, BLOCK, 0, st>>>(table, g->col, g->Ep, C, g->N, g->row_off, mode, out, sink);
    SN_HIP(hipGetLastError());
    return SNGNN_OK;
}
