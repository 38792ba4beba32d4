// klds.hip — developer experiment: the FAST path of k_step with the +-1 shifted populations staged
// through LDS (what BASELINE.json's north star sketches) against the production forms (element-aligned
// 16-B loads; aligned loads + lane shuffles).  4096x4096 fp32, body-free lattice, ping-pong, alternating
// sweep, nontemporal loads everywhere.  Bit-equality with the production kernel is checked.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>
#include "../airfoil-cfd-tool_amd/csrc/kernels.hpp"
#include "../airfoil-cfd-tool_amd/csrc/step_fast.hpp"
using namespace wt;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

// One block = 4 waves = 4 consecutive tiles (1024 rows) of one column.  The six populations that need a
// shift along j are loaded with ALIGNED 16-B loads, written to LDS (6 x 1024 floats + 2 halo floats each),
// and read back shifted by one element; only the block's two edge rows come from global memory.
__global__ __launch_bounds__(256) void k_step_lds(const float *__restrict__ fs, float *__restrict__ fd, Geom g, int tpc, float tau, float U0, int rev)
{
    __shared__ float sh[6][1024 + 8];
    const int t = threadIdx.x;                 // 0..255, row offset 4*t inside the block
    long blk = blockIdx.x;
    const long nblk = (long)g.nxl * (tpc / 4);
    if (rev) blk = nblk - 1 - blk;
    const int i = (int)(blk / (tpc / 4)), jb = (int)(blk % (tpc / 4)) * 1024;
    if (i == 0 || i == g.nxl - 1) return;      // experiment: interior columns only
    const int j0 = jb + 4 * t;
    const long c = (long)i * g.pitch + j0;
    const float *s = fs + g.pitch;
    float *d = fd + g.pitch;
    const long P = g.plane;
    Vec<float> fin[9];
    fin[0] = vload<float, true>(s + 0 * P + c);
    fin[1] = vload<float, true>(s + 1 * P + c - g.pitch);
    fin[3] = vload<float, true>(s + 3 * P + c + g.pitch);
    const float *src[6] = {s + 2 * P + c, s + 5 * P + c - g.pitch, s + 6 * P + c + g.pitch,      // need j-1
                           s + 4 * P + c, s + 7 * P + c + g.pitch, s + 8 * P + c - g.pitch};     // need j+1
#pragma unroll
    for (int q = 0; q < 6; q++) {
        const Vec<float> r = vload<float, true>(src[q]);
        *reinterpret_cast<float4 *>(&sh[q][4 + 4 * t]) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
        if (t == 0) sh[q][3] = src[q][-1];                  // row jb-1
        if (t == 255) sh[q][4 + 1024] = src[q][4];          // row jb+1024
    }
    __syncthreads();
    const int kmap[6] = {2, 5, 6, 4, 7, 8};
#pragma unroll
    for (int q = 0; q < 6; q++) {
        const int off = (q < 3) ? 3 : 5;                    // j-1 -> element 4*t+3 ; j+1 -> 4*t+5
#pragma unroll
        for (int v = 0; v < 4; v++) fin[kmap[q]].v[v] = sh[q][off + 4 * t + v];
    }
    float feq0[9];
    feq_all<float>(1.0f, U0, 0.0f, feq0);
    Vec<float> out[9];
#pragma unroll
    for (int v = 0; v < 4; v++) {
        float a[9], o[9], rho, ux, uy;
#pragma unroll
        for (int k = 0; k < 9; k++) a[k] = fin[k].v[v];
        collide<float>(a, tau, o, rho, ux, uy);
        const int j = j0 + v;
        const bool far = (j == 0) || (j == g.ny - 1);
#pragma unroll
        for (int k = 0; k < 9; k++) out[k].v[v] = far ? feq0[k] : o[k];
    }
#pragma unroll
    for (int k = 0; k < 9; k++) vstore<float>(d + k * P + c, out[k]);
}

int main(int argc, char **argv)
{
    const int nx = 4096, ny = 4096, rounds = argc > 1 ? atoi(argv[1]) : 10;
    Geom g; g.nxl = nx; g.ny = ny; g.gi0 = 0; g.nx_g = nx; g.pitch = ny;
    g.plane = (((long)(nx + 2) * g.pitch * 4 + 4095) / 4096 * 4096 + 17408) / 4;
    const int tpc = (int)(g.pitch / 256);
    const size_t lat = (size_t)9 * g.plane * 4;
    float *f0, *f1, *f2, *macro; uint8_t *mask, *tiles;
    CK(hipMalloc(&f0, lat)); CK(hipMalloc(&f1, lat)); CK(hipMalloc(&f2, lat)); CK(hipMalloc(&macro, (size_t)3 * nx * g.pitch * 4));
    CK(hipMalloc(&mask, (size_t)(nx + 2) * g.pitch)); CK(hipMalloc(&tiles, (size_t)nx * tpc));
    CK(hipMemset(mask, 0, (size_t)(nx + 2) * g.pitch));
    hipStream_t st; CK(hipStreamCreate(&st));
    classify_tiles(mask, tiles, g, tpc, st);
    {
        std::vector<float> h((size_t)9 * g.plane);
        unsigned long long x = 88172645463325252ULL;
        for (int k = 0; k < 9; k++) { const double w = k == 0 ? 4.0 / 9 : (k <= 4 ? 1.0 / 9 : 1.0 / 36);
            for (long t = 0; t < g.plane; t++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[(size_t)k * g.plane + t] = (float)w * (1.0f + 0.02f * ((x >> 40) / 16777216.0f - 0.5f)); } }
        CK(hipMemcpy(f0, h.data(), lat, hipMemcpyHostToDevice));
    }
    const float tau = 0.58f, U0 = 0.06f;
    step_columns<float, 3>(f0, f1, macro, mask, tiles, tpc, g, 0, nx, tau, U0, false, 0, st);
    CK(hipMemset(f2, 0, lat)); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_step_lds, dim3((unsigned)((long)nx * (tpc / 4))), dim3(256), 0, st, f0, f2, g, tpc, tau, U0, 0);
    CK(hipStreamSynchronize(st)); CK(hipGetLastError());
    {
        std::vector<float> a((size_t)g.plane), b((size_t)g.plane); long bad = 0;
        for (int k = 0; k < 9; k++) {
            CK(hipMemcpy(a.data(), f1 + (size_t)k * g.plane, g.plane * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(b.data(), f2 + (size_t)k * g.plane, g.plane * 4, hipMemcpyDeviceToHost));
            for (int i = 1; i < nx - 1; i++) for (int j = 0; j < ny; j++) { const size_t o = (size_t)(i + 1) * g.pitch + j; if (memcmp(&a[o], &b[o], 4)) bad++; }
        }
        printf("LDS-staged kernel vs production: %ld values differ\n", bad);
    }
    struct Var { std::string name; std::function<void(const float *, float *, int)> fn; std::vector<float> ms; };
    std::vector<Var> vs;
    vs.push_back({"production: element-aligned 16-B loads", [&](const float *a, float *b, int r) { step_columns<float, 3>(a, b, macro, mask, tiles, tpc, g, 0, nx, tau, U0, false, r, st); }, {}});
    vs.push_back({"aligned loads + lane shuffles", [&](const float *a, float *b, int r) { step_columns<float, 1>(a, b, macro, mask, tiles, tpc, g, 0, nx, tau, U0, false, r, st); }, {}});
    vs.push_back({"aligned loads staged through LDS", [&](const float *a, float *b, int r) { hipLaunchKernelGGL(k_step_lds, dim3((unsigned)((long)nx * (tpc / 4))), dim3(256), 0, st, a, b, g, tpc, tau, U0, r); }, {}});
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < rounds + 2; r++) for (auto &v : vs) {
        CK(hipEventRecord(e0, st));
        for (int q = 0; q < 6; q++) { if (q & 1) v.fn(f1, f0, 1); else v.fn(f0, f1, 0); }
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 2) v.ms.push_back(ms / 6);
    }
    printf("%-42s %10s %10s %10s\n", "variant (4096^2 fp32, no body)", "med us", "min us", "GB/s");
    for (auto &v : vs) { std::sort(v.ms.begin(), v.ms.end()); const double med = v.ms[v.ms.size() / 2];
        printf("%-42s %10.1f %10.1f %10.0f\n", v.name.c_str(), med * 1e3, v.ms[0] * 1e3, 72.0 * nx * ny / (med * 1e-3) / 1e9); }
    return 0;
}
