// Micro-benchmark (tuning aid, not product code): gather 160-byte rows (C = 40 fp32)
// by index from a 27 MB table, (a) straight into registers, (b) with LDS-DMA
// (global_load_lds_dwordx4) into a per-wave double buffer.  Prints GB/s of useful
// bytes and a checksum per variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_nt(const float4 *p)
{
    const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int C = 40, CH = C / 4, RPI = 64 / CH;   // 10 chunks per row, 6 rows per DMA instruction

template <int U>
__global__ __launch_bounds__(256) void k_reg(const float *__restrict__ h, const int *__restrict__ ids,
                                             int E, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63, gid = lane >> 4, lg = lane & 15;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = gridDim.x * 4;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int base = wave * 4 * U; base < E; base += nw * 4 * U) {
        float4 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = base + u * 4 + gid;
            const int j = e < E ? ids[e] : 0;
            x[u] = lg < CH ? *reinterpret_cast<const float4 *>(h + (size_t)j * C + lg * 4) : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
    }
    float s = acc.x + acc.y + acc.z + acc.w;
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) out[wave] = s;
}

// split layout: table A [N][32] (one 128-B line per row) + table B [N][8] (32 B per row,
// four rows per line, 5.4 MB: mostly L2-resident)
template <int U>
__global__ __launch_bounds__(256) void k_split(const float *__restrict__ ha, const float *__restrict__ hb,
                                               const int *__restrict__ ids, int E, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63, gid = lane >> 4, lg = lane & 15;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = gridDim.x * 4;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int base = wave * 4 * U; base < E; base += nw * 4 * U) {
        float4 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = base + u * 4 + gid;
            const int j = e < E ? ids[e] : 0;
            x[u] = lg < 8 ? *reinterpret_cast<const float4 *>(ha + (size_t)j * 32 + lg * 4)
                 : lg < 10 ? *reinterpret_cast<const float4 *>(hb + (size_t)j * 8 + (lg - 8) * 4)
                           : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
    }
    float s = acc.x + acc.y + acc.z + acc.w;
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) out[wave] = s;
}

// 128-byte rows only (C = 32): what a one-line-per-row table can deliver
template <int U>
__global__ __launch_bounds__(256) void k_c32(const float *__restrict__ ha, const int *__restrict__ ids,
                                             int E, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63, gid = lane >> 3, lg = lane & 7;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = gridDim.x * 4;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int base = wave * 8 * U; base < E; base += nw * 8 * U) {
        float4 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = base + u * 8 + gid;
            const int j = e < E ? ids[e] : 0;
            x[u] = *reinterpret_cast<const float4 *>(ha + (size_t)j * 32 + lg * 4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
    }
    float s = acc.x + acc.y + acc.z + acc.w;
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) out[wave] = s;
}


// rows of C floats at a padded stride of STRIDE floats (alignment / sector experiments)
template <int U, int STRIDE, bool NT>
__global__ __launch_bounds__(256) void k_stride(const float *__restrict__ h, const int *__restrict__ ids,
                                                int E, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63, gid = lane >> 4, lg = lane & 15;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = gridDim.x * 4;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int base = wave * 4 * U; base < E; base += nw * 4 * U) {
        float4 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = base + u * 4 + gid;
            const int j = e < E ? ids[e] : 0;
            const float4 *p = reinterpret_cast<const float4 *>(h + (size_t)j * STRIDE + (lg < CH ? lg : 0) * 4);
            x[u] = NT ? ld_nt(p) : *p;
            if (lg >= CH) x[u] = make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
    }
    float s = acc.x + acc.y + acc.z + acc.w;
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) out[wave] = s;
}

// W-float rows (W = 16: 64 B, W = 32: 128 B, W = 8: 32 B), W/4 lanes per row, 256/W rows per instruction
template <int U, int W, bool NT>
__global__ __launch_bounds__(256) void k_narrow(const float *__restrict__ ha, const int *__restrict__ ids,
                                                int E, float *__restrict__ out)
{
    constexpr int G = W / 4, RPI_ = 64 / G;
    const int lane = threadIdx.x & 63, gid = lane / G, lg = lane % G;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = gridDim.x * 4;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int base = wave * RPI_ * U; base < E; base += nw * RPI_ * U) {
        float4 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = base + u * RPI_ + gid;
            const int j = e < E ? ids[e] : 0;
            const float4 *p = reinterpret_cast<const float4 *>(ha + (size_t)j * W + lg * 4);
            x[u] = NT ? ld_nt(p) : *p;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
    }
    float s = acc.x + acc.y + acc.z + acc.w;
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) out[wave] = s;
}

// split layout with the A stream non-temporal (so that it does not evict the small B table from L2)
template <int U, bool NTA>
__global__ __launch_bounds__(256) void k_split2(const float *__restrict__ ha, const float *__restrict__ hb,
                                                const int *__restrict__ ids, int E, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63, gid = lane >> 4, lg = lane & 15;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = gridDim.x * 4;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int base = wave * 4 * U; base < E; base += nw * 4 * U) {
        float4 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = base + u * 4 + gid;
            const int j = e < E ? ids[e] : 0;
            const float4 *pa = reinterpret_cast<const float4 *>(ha + (size_t)j * 32 + (lg & 7) * 4);
            const float4 *pb = reinterpret_cast<const float4 *>(hb + (size_t)j * 8 + (lg & 1) * 4);
            if (lg < 8) x[u] = NTA ? ld_nt(pa) : *pa;
            else if (lg < 10) x[u] = *pb;
            else x[u] = make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
    }
    float s = acc.x + acc.y + acc.z + acc.w;
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) out[wave] = s;
}

// NI DMA instructions (RPI rows each) per batch, two LDS buffers per wave
template <int NI>
__global__ __launch_bounds__(256) void k_dma(const float *__restrict__ h, const int *__restrict__ ids,
                                             int E, float *__restrict__ out)
{
    constexpr int BR = NI * RPI;                  // rows per batch
    __shared__ __attribute__((aligned(16))) float lds[4][2][NI * 256];   // 1 KiB per DMA instruction
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gid = lane >> 4, lg = lane & 15;
    const int wave = blockIdx.x * 4 + wv, nw = gridDim.x * 4;
    const int r_in = lane / CH, c_in = lane % CH;     // this lane's row / chunk inside one DMA instruction
    const bool dma_lane = lane < RPI * CH;
    float4 acc = make_float4(0, 0, 0, 0);
    int base = wave * BR;
    auto issue = [&](int b0, int buf) {
        // ids through SCALAR loads (lgkmcnt): a VGPR-destination load next to LDS-DMA makes
        // hipcc wait vmcnt(0) before every use and drains the DMA queue
        const int *idp = ids + __builtin_amdgcn_readfirstlane(b0);
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            int j = 0;
#pragma unroll
            for (int r = 0; r < RPI; ++r) {
                const int e = q * RPI + r;
                const int jr = idp[min(e, E - 1 - b0)];
                j = (r_in == r) ? jr : j;
            }
            const int e = b0 + q * RPI + r_in;
            if (dma_lane && e < E)
                __builtin_amdgcn_global_load_lds(h + (size_t)j * C + c_in * 4,
                                                 (__attribute__((address_space(3))) void *)&lds[wv][buf][q * 256],
                                                 16, 0, 0);
        }
    };
    int buf = 0;
    if (base < E) issue(base, 0);
    for (; base < E; base += nw * BR) {
        const int nxt = base + nw * BR;
        if (nxt < E) issue(nxt, buf ^ 1);
        // wait for the CURRENT batch only: the NI instructions of the next batch stay in flight
        if (nxt < E) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int nrow = min(BR, E - base);
        for (int r = gid; r < nrow; r += 4) {
            if (lg < CH) {
                const float4 x = *reinterpret_cast<const float4 *>(&lds[wv][buf][(r / RPI) * 256 + ((r % RPI) * CH + lg) * 4]);
                acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // reads done before the buffer is refilled
        buf ^= 1;
    }
    float s = acc.x + acc.y + acc.z + acc.w;
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) out[wave] = s;
}

int main()
{
    const int N = 169343, E = 1163820;
    std::vector<float> hh((size_t)N * C);
    std::vector<int> hi(E);
    srand(1);
    for (auto &v : hh) v = (rand() % 1000) * 0.001f;
    for (auto &v : hi) v = (int)(((long long)rand() * 32768 + rand()) % N);
    float *dh, *dout, *dha, *dhb;
    std::vector<float> ha((size_t)N * 32), hb((size_t)N * 8);
    for (int i = 0; i < N; ++i) { for (int c = 0; c < 32; ++c) ha[(size_t)i * 32 + c] = hh[(size_t)i * C + c]; for (int c = 0; c < 8; ++c) hb[(size_t)i * 8 + c] = hh[(size_t)i * C + 32 + c]; }
    int *di;
    CK(hipMalloc(&dh, hh.size() * 4));
    CK(hipMalloc(&di, hi.size() * 4));
    CK(hipMalloc(&dha, ha.size() * 4)); CK(hipMalloc(&dhb, hb.size() * 4));
    CK(hipMemcpy(dha, ha.data(), ha.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dhb, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dout, 4 * 65536));
    CK(hipMemcpy(dh, hh.data(), hh.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(di, hi.data(), hi.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch) {
        float best = 1e9f, sum = 0;
        for (int it = 0; it < 30; ++it) {
            CK(hipMemset(dout, 0, 4 * 65536));
            CK(hipEventRecord(e0));
            launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (it >= 5 && ms < best) best = ms;
            { std::vector<float> ho(65536); CK(hipMemcpy(ho.data(), dout, 4 * 65536, hipMemcpyDeviceToHost)); double t = 0; for (float v : ho) t += v; sum = (float)t; }
        }
        printf("%-28s best %7.1f us  %7.1f GB/s useful   checksum %.3f\n", name, best * 1e3,
               (double)E * C * 4 / (best * 1e-3) / 1e9, sum);
        fflush(stdout);
    };
    // padded tables for the stride experiments (same row contents)
    float *dh192, *dh256;
    {
        std::vector<float> p192((size_t)N * 48, 0.f), p256((size_t)N * 64, 0.f);
        for (int i = 0; i < N; ++i)
            for (int c = 0; c < C; ++c) { p192[(size_t)i * 48 + c] = hh[(size_t)i * C + c]; p256[(size_t)i * 64 + c] = hh[(size_t)i * C + c]; }
        CK(hipMalloc(&dh192, p192.size() * 4)); CK(hipMalloc(&dh256, p256.size() * 4));
        CK(hipMemcpy(dh192, p192.data(), p192.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dh256, p256.data(), p256.size() * 4, hipMemcpyHostToDevice));
    }
    for (int bpc : {6, 8}) {
        const int grid = 256 * bpc;
        printf("-- %d blocks/CU\n", bpc);
        run("reg U=4 (stride 160)", [&] { k_reg<4><<<grid, 256>>>(dh, di, E, dout); });
        run("stride 160 nt", [&] { k_stride<4, 40, true><<<grid, 256>>>(dh, di, E, dout); });
        run("stride 192", [&] { k_stride<4, 48, false><<<grid, 256>>>(dh192, di, E, dout); });
        run("stride 256", [&] { k_stride<4, 64, false><<<grid, 256>>>(dh256, di, E, dout); });
        run("128-B rows (8 rows/instr)", [&] { k_narrow<4, 32, false><<<grid, 256>>>(dha, di, E, dout); });
        run("128-B rows nt", [&] { k_narrow<4, 32, true><<<grid, 256>>>(dha, di, E, dout); });
        run("64-B rows (16 rows/instr)", [&] { k_narrow<4, 16, false><<<grid, 256>>>(dha, di, E, dout); });
        run("64-B rows U=8", [&] { k_narrow<8, 16, false><<<grid, 256>>>(dha, di, E, dout); });
        run("32-B rows (32 rows/instr)", [&] { k_narrow<4, 8, false><<<grid, 256>>>(dhb, di, E, dout); });
        run("split 128+32", [&] { k_split2<4, false><<<grid, 256>>>(dha, dhb, di, E, dout); });
        run("split 128(nt)+32", [&] { k_split2<4, true><<<grid, 256>>>(dha, dhb, di, E, dout); });
    }
    return 0;
}
