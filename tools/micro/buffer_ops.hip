// Micro-check of raw buffer load/store semantics on gfx950 (bounds check drops stores / zeroes loads).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *p, float *q, int n, unsigned flags)
{
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, n * 4, flags);
    __amdgpu_buffer_rsrc_t w = __builtin_amdgcn_make_buffer_rsrc((void *)q, 0, n * 4, flags);
    u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16, 0, 0);
    float s = __uint_as_float(v.x) + __uint_as_float(v.y) + __uint_as_float(v.z) + __uint_as_float(v.w);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(s), w, threadIdx.x * 4, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(123.f), w, 0x80000000u + threadIdx.x * 4, 0, 0);
}
int main()
{
    const int n = 200;   // 64 lanes read 256 floats: lanes >= 50 are out of bounds
    float *p, *q, hp[256], hq[256];
    hipMalloc(&p, 1024); hipMalloc(&q, 1024);
    for (int i = 0; i < 256; ++i) hp[i] = 1.f;
    for (unsigned flags : {0x00020000u, 0x00027000u}) {
        hipMemcpy(p, hp, 1024, hipMemcpyHostToDevice);
        hipMemset(q, 0, 1024);
        k<<<1, 64>>>(p, q, n, flags);
        hipMemcpy(hq, q, 1024, hipMemcpyDeviceToHost);
        printf("flags %08x: q[0]=%g q[49]=%g q[50]=%g q[63]=%g q[199]=%g q[200]=%g\n", flags, hq[0], hq[49], hq[50], hq[63], hq[199], hq[200]);
    }
    return 0;
}
