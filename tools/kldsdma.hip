// tools/kldsdma.hip — does `buffer_load_dword ... lds` (LDS-DMA) put lane i's dword at M0 base + 4 i, and do two of them 256 bytes apart give every lane
// its two consecutive rows back through one ds_read_b64?  (the layout step_march.hpp's prefetch of a general column's own populations relies on)
//   hipcc --offload-arch=gfx950 -O3 -o tools/kldsdma tools/kldsdma.hip && ./tools/kldsdma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float *src, float *dst, int n)
{
    __shared__ float buf[4][128];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, n * 4, 0x00020000);
    const unsigned base = (blockIdx.x * 4 + w) * 512;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)&buf[w][0], 4, lane * 4, base, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)&buf[w][64], 4, lane * 4, base + 256, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float2 v = *reinterpret_cast<float2 *>(&buf[w][2 * lane]);
    dst[(blockIdx.x * 256 + threadIdx.x) * 2] = v.x;
    dst[(blockIdx.x * 256 + threadIdx.x) * 2 + 1] = v.y;
}
int main()
{
    const int nb = 64, n = nb * 512;
    std::vector<float> h(n), o(n);
    for (int i = 0; i < n; i++) h[i] = (float)i * 0.5f + 1.0f;
    float *s, *d;
    hipMalloc(&s, n * 4); hipMalloc(&d, n * 4);
    hipMemcpy(s, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, s, d, n);
    hipMemcpy(o.data(), d, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) bad += o[i] != h[i];
    printf("LDS-DMA layout check: %d of %d elements differ\n", bad, n);
    return bad != 0;
}
