// kbench2.hip — developer micro-benchmark: what bounds an 18-stream (9 in / 9 out) float4 copy?
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>
typedef float f4n __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

// classic single-stream copy, grid-stride
__global__ __launch_bounds__(256) void k_copy1(const f4n *__restrict__ s, f4n *__restrict__ d, long n)
{
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long)gridDim.x * 256) d[t] = s[t];
}
// one 1-KiB tile per wave, no loop
__global__ __launch_bounds__(256) void k_copy1_tile(const f4n *__restrict__ s, f4n *__restrict__ d, long n)
{
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t < n) d[t] = s[t];
}

// 18-stream copy. NTL/NTS: nontemporal loads/stores. REV: traverse tiles backwards.
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_copy18(const float *__restrict__ fs, float *__restrict__ fd, long ntiles, int tpc, long pitch, long plane, int rev)
{
    const int lane = threadIdx.x & 63;
    long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (rev) tile = ntiles - 1 - tile;
    const int i = (int)(tile / tpc), jt = (int)(tile % tpc);
    const long c = (long)(i + 1) * pitch + jt * 256 + lane * 4;
    f4n v[9];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const f4n *p = reinterpret_cast<const f4n *>(fs + k * plane + c);
        v[k] = NTL ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int k = 0; k < 9; k++) {
        f4n *p = reinterpret_cast<f4n *>(fd + k * plane + c);
        if (NTS) __builtin_nontemporal_store(v[k], p); else *p = v[k];
    }
}

// same, but each wave walks NT consecutive tiles (NT KiB contiguous per stream per wave)
template <int NT>
__global__ __launch_bounds__(256) void k_copy18_multi(const float *__restrict__ fs, float *__restrict__ fd, long ntiles, int tpc, long pitch, long plane, int rev)
{
    const int lane = threadIdx.x & 63;
    long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (rev) w = ntiles / NT - 1 - w;
#pragma unroll 1
    for (int q = 0; q < NT; q++) {
        const long tile = w * NT + q;
        const int i = (int)(tile / tpc), jt = (int)(tile % tpc);
        const long c = (long)(i + 1) * pitch + jt * 256 + lane * 4;
        f4n v[9];
#pragma unroll
        for (int k = 0; k < 9; k++) v[k] = *reinterpret_cast<const f4n *>(fs + k * plane + c);
#pragma unroll
        for (int k = 0; k < 9; k++) *reinterpret_cast<f4n *>(fd + k * plane + c) = v[k];
    }
}

// occupancy-limited variant (dynamic LDS steals capacity)
__global__ __launch_bounds__(256) void k_copy18_lds(const float *__restrict__ fs, float *__restrict__ fd, long ntiles, int tpc, long pitch, long plane, int rev)
{
    extern __shared__ float dummy[];
    const int lane = threadIdx.x & 63;
    long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (rev) tile = ntiles - 1 - tile;
    const int i = (int)(tile / tpc), jt = (int)(tile % tpc);
    const long c = (long)(i + 1) * pitch + jt * 256 + lane * 4;
    f4n v[9];
#pragma unroll
    for (int k = 0; k < 9; k++) v[k] = *reinterpret_cast<const f4n *>(fs + k * plane + c);
    if (pitch < 0) dummy[threadIdx.x] = v[0].x;   // never true; keeps the LDS allocation alive
#pragma unroll
    for (int k = 0; k < 9; k++) *reinterpret_cast<f4n *>(fd + k * plane + c) = v[k];
}

struct Variant { std::string name; std::function<void(float *, float *, int)> launch; std::vector<float> ms; };

int main(int argc, char **argv)
{
    const int nx = 4096, ny = 4096, rounds = argc > 1 ? atoi(argv[1]) : 12;
    const long pitch = ny, tpc = pitch / 256;
    const long plane0 = (long)(nx + 2) * pitch;
    const long pad_max = 1 << 20;
    const size_t lat = (size_t)9 * (plane0 + pad_max) * 4;
    float *f0, *f1;
    CK(hipMalloc(&f0, lat)); CK(hipMalloc(&f1, lat));
    CK(hipMemset(f0, 0, lat)); CK(hipMemset(f1, 0, lat));
    hipStream_t st; CK(hipStreamCreate(&st));
    const long ntiles = (long)nx * tpc;
    const dim3 grid((unsigned)(ntiles / 4)), block(256);
    const long n4 = (long)9 * nx * ny / 4;   // float4 elements of one lattice copy
    std::vector<Variant> vs;
    vs.push_back({"copy1 one tile per wave", [&](float *a, float *b, int) { hipLaunchKernelGGL(k_copy1_tile, dim3((unsigned)(n4 / 256)), block, 0, st, (const f4n *)a, (f4n *)b, n4); }, {}});
    vs.push_back({"copy18 fwd", [&](float *a, float *b, int) { hipLaunchKernelGGL((k_copy18<false, false>), grid, block, 0, st, a, b, ntiles, (int)tpc, pitch, plane0, 0); }, {}});
    for (long pad : {0L, 1088L, 4352L, 4352L + 64, 17408L + 1088 + 64}) {
        vs.push_back({"alt ntload pad " + std::to_string(pad * 4) + " B", [&, pad](float *a, float *b, int r) { hipLaunchKernelGGL((k_copy18<true, false>), grid, block, 0, st, a, b, ntiles, (int)tpc, pitch, plane0 + pad, r); }, {}});
        vs.push_back({"alt ntload+ntstore pad " + std::to_string(pad * 4) + " B", [&, pad](float *a, float *b, int r) { hipLaunchKernelGGL((k_copy18<true, true>), grid, block, 0, st, a, b, ntiles, (int)tpc, pitch, plane0 + pad, r); }, {}});
        vs.push_back({"alt plain pad " + std::to_string(pad * 4) + " B", [&, pad](float *a, float *b, int r) { hipLaunchKernelGGL((k_copy18<false, false>), grid, block, 0, st, a, b, ntiles, (int)tpc, pitch, plane0 + pad, r); }, {}});
        vs.push_back({"fwd ntload pad " + std::to_string(pad * 4) + " B", [&, pad](float *a, float *b, int) { hipLaunchKernelGGL((k_copy18<true, false>), grid, block, 0, st, a, b, ntiles, (int)tpc, pitch, plane0 + pad, 0); }, {}});
    }
    CK(hipFuncSetAttribute((const void *)k_copy18_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));

    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 6;   // even: ping-pong returns to the start
    for (int r = 0; r < rounds + 2; r++) {
        for (auto &v : vs) {
            CK(hipEventRecord(e0, st));
            for (int q = 0; q < reps; q++) { if (q & 1) v.launch(f1, f0, 1); else v.launch(f0, f1, 0); }
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) v.ms.push_back(ms / reps);
        }
    }
    const double bytes = 72.0 * nx * ny;
    printf("%-40s %10s %10s %10s\n", "variant (ping-pong f0<->f1)", "med us", "min us", "GB/s(med)");
    for (auto &v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        const double med = v.ms[v.ms.size() / 2], mn = v.ms[0];
        printf("%-40s %10.1f %10.1f %10.0f\n", v.name.c_str(), med * 1e3, mn * 1e3, bytes / (med * 1e-3) / 1e9);
    }
    return 0;
}
