// kbench.hip — developer micro-benchmark for the step kernel (not part of the product or tests).
// Runs kernel variants interleaved in ONE process on a 4096x4096 fp32 lattice and prints the
// median launch time / algorithmic GB/s of each (cdna_hip_programming.md §5.4 rule 24).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o tools/kbench tools/kbench.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <functional>
#include <string>
#include <vector>
#include "../airfoil-cfd-tool_amd/csrc/kernels.hpp"
#include "../airfoil-cfd-tool_amd/csrc/step_fast.hpp"

using namespace wt;
typedef float f4n __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

// ---------------------------------------------------------------------------------------------
// variant A: pure 9-in / 9-out float4 copy with the production indexing
// ---------------------------------------------------------------------------------------------
template <bool NT>
__global__ __launch_bounds__(256) void k_copy18(const float *__restrict__ fs, float *__restrict__ fd, int tiles_per_col, Geom g)
{
    const int lane = threadIdx.x & 63;
    const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int i = (int)(tile / tiles_per_col), jt = (int)(tile % tiles_per_col);
    const long c = (long)i * g.pitch + jt * 256 + lane * 4;
    const float *s = fs + g.pitch;
    float *d = fd + g.pitch;
    float4 v[9];
#pragma unroll
    for (int k = 0; k < 9; k++) v[k] = *reinterpret_cast<const float4 *>(s + k * g.plane + c);
#pragma unroll
    for (int k = 0; k < 9; k++) {
        if (NT) { f4n y = {v[k].x, v[k].y, v[k].z, v[k].w}; __builtin_nontemporal_store(y, reinterpret_cast<f4n *>(d + k * g.plane + c)); }
        else *reinterpret_cast<float4 *>(d + k * g.plane + c) = v[k];
    }
}

// ---------------------------------------------------------------------------------------------
// variant B: experimental fast path with knobs
// ---------------------------------------------------------------------------------------------
// DIV: 0 = IEEE '/', 1 = reciprocal + one Newton/FMA correction (exact for normal results), 2 = multiply by 1/tau (NOT exact)
template <int DIV>
__device__ __forceinline__ float div_tau(float x, float tau, float rtau)
{
    if (DIV == 0) return x / tau;
    if (DIV == 2) return x * rtau;
    // q0 = x*r ; r = fma(-q0,tau,x) ; q = fma(r, rtau, q0)
    const float q0 = x * rtau;
    const float rem = __builtin_fmaf(-q0, tau, x);
    return __builtin_fmaf(rem, rtau, q0);
}

template <int DIV, bool NTS, bool NTL, int SHUF>
__global__ __launch_bounds__(256) void k_fast(const float *__restrict__ fs, float *__restrict__ fd, int tiles_per_col, Geom g, float tau, float U0)
{
    const int lane = threadIdx.x & 63;
    const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int i = 1 + (int)(tile / tiles_per_col), jt = (int)(tile % tiles_per_col);
    if (i >= g.nxl - 1) return;
    const int j0 = jt * 256 + lane * 4;
    const long c = (long)i * g.pitch + j0;
    const float *s = fs + g.pitch;
    float *d = fd + g.pitch;
    const long P = g.plane;
    auto ld = [&](const float *p) -> Vec<float> {
        if (NTL) { f4n x = __builtin_nontemporal_load(reinterpret_cast<const f4n *>(p)); Vec<float> r; r.v[0] = x.x; r.v[1] = x.y; r.v[2] = x.z; r.v[3] = x.w; return r; }
        return vload<float>(p);
    };
    Vec<float> fin[9];
    fin[0] = ld(s + 0 * P + c);
    fin[1] = ld(s + 1 * P + c - g.pitch);
    fin[3] = ld(s + 3 * P + c + g.pitch);
    const float *p2 = s + 2 * P + c, *p5 = s + 5 * P + c - g.pitch, *p6 = s + 6 * P + c + g.pitch;
    const float *p4 = s + 4 * P + c, *p7 = s + 7 * P + c + g.pitch, *p8 = s + 8 * P + c - g.pitch;
    if (SHUF == 0) {   // production: shuffles + edge loads
        const Vec<float> r2 = ld(p2), r5 = ld(p5), r6 = ld(p6), r4 = ld(p4), r7 = ld(p7), r8 = ld(p8);
        fin[2] = shift_from_below<float>(r2, p2, lane); fin[5] = shift_from_below<float>(r5, p5, lane); fin[6] = shift_from_below<float>(r6, p6, lane);
        fin[4] = shift_from_above<float>(r4, p4, lane); fin[7] = shift_from_above<float>(r7, p7, lane); fin[8] = shift_from_above<float>(r8, p8, lane);
    } else {           // unaligned 16-B loads straight from the shifted address
        auto ldu = [&](const float *p) -> Vec<float> { Vec<float> r; typedef float f4u __attribute__((ext_vector_type(4), aligned(4))); f4u x = *reinterpret_cast<const f4u *>(p); r.v[0] = x.x; r.v[1] = x.y; r.v[2] = x.z; r.v[3] = x.w; return r; };
        fin[2] = ldu(p2 - 1); fin[5] = ldu(p5 - 1); fin[6] = ldu(p6 - 1);
        fin[4] = ldu(p4 + 1); fin[7] = ldu(p7 + 1); fin[8] = ldu(p8 + 1);
    }
    const float rtau = 1.0f / tau;
    Vec<float> out[9];
#pragma unroll
    for (int v = 0; v < 4; v++) {
        float a[9], eq[9], r, ux, uy;
#pragma unroll
        for (int k = 0; k < 9; k++) a[k] = fin[k].v[v];
        moments(a, r, ux, uy);
        r = (r < 0.5f) ? 0.5f : r; r = (2.0f < r) ? 2.0f : r;
        const float spd2 = ux * ux + uy * uy;
        if (spd2 > 0.35f * 0.35f) { const float k = 0.35f / sqrtf(spd2); ux *= k; uy *= k; }
        feq_all(r, ux, uy, eq);
#pragma unroll
        for (int k = 0; k < 9; k++) out[k].v[v] = a[k] - div_tau<DIV>(a[k] - eq[k], tau, rtau);
    }
#pragma unroll
    for (int k = 0; k < 9; k++) {
        float4 x; x.x = out[k].v[0]; x.y = out[k].v[1]; x.z = out[k].v[2]; x.w = out[k].v[3];
        if (NTS) { f4n y = {x.x, x.y, x.z, x.w}; __builtin_nontemporal_store(y, reinterpret_cast<f4n *>(d + k * P + c)); }
        else *reinterpret_cast<float4 *>(d + k * P + c) = x;
    }
}

struct Variant { std::string name; std::function<void(float *, float *, int)> launch; std::vector<float> ms; };

int main(int argc, char **argv)
{
    const int nx = argc > 1 ? atoi(argv[1]) : 4096, ny = argc > 2 ? atoi(argv[2]) : 4096, rounds = argc > 3 ? atoi(argv[3]) : 15;
    Geom g; g.nxl = nx; g.ny = ny; g.gi0 = 0; g.nx_g = nx; g.pitch = ((long)ny + 255) / 256 * 256; g.plane = (((long)(nx + 2) * g.pitch * 4 + 4095) / 4096 * 4096 + (argc > 4 ? atol(argv[4]) : 17408)) / 4;
    const int tpc = (int)(g.pitch / 256);
    const size_t lat = (size_t)9 * g.plane * 4;
    float *f0, *f1, *macro; uint8_t *mask, *mask_empty, *tiles, *tiles_empty;
    CK(hipMalloc(&f0, lat)); CK(hipMalloc(&f1, lat)); CK(hipMalloc(&macro, (size_t)3 * nx * g.pitch * 4));
    CK(hipMalloc(&mask, (size_t)(nx + 2) * g.pitch)); CK(hipMalloc(&mask_empty, (size_t)(nx + 2) * g.pitch));
    CK(hipMalloc(&tiles, (size_t)nx * tpc)); CK(hipMalloc(&tiles_empty, (size_t)nx * tpc));
    // body: ellipse, chord nx/1.84, 12 % thick, 10 deg
    std::vector<uint8_t> hm((size_t)(nx + 2) * g.pitch, 0);
    const double chord = nx / 1.84, cx = 0.42 * chord + 0.5 * chord, cy = ny / 2.0, a = 0.5 * chord, b = 0.06 * chord, th = -10.0 * M_PI / 180;
    long nsolid = 0;
    for (int i = 0; i < nx; i++) for (int j = 0; j < ny; j++) {
        const double dx = i - cx, dy = j - cy, xr = dx * cos(th) - dy * sin(th), yr = dx * sin(th) + dy * cos(th);
        if (xr * xr / (a * a) + yr * yr / (b * b) <= 1.0) { hm[(size_t)(i + 1) * g.pitch + j] = 1; nsolid++; }
    }
    CK(hipMemcpy(mask, hm.data(), hm.size(), hipMemcpyHostToDevice));
    CK(hipMemset(mask_empty, 0, hm.size()));
    hipStream_t st; CK(hipStreamCreate(&st));
    classify_tiles(mask, tiles, g, tpc, st); classify_tiles(mask_empty, tiles_empty, g, tpc, st);
    CK(hipStreamSynchronize(st));
    std::vector<uint8_t> ht((size_t)nx * tpc); CK(hipMemcpy(ht.data(), tiles, ht.size(), hipMemcpyDeviceToHost));
    long cnt[5] = {0, 0, 0, 0, 0}; for (auto t : ht) cnt[t]++;
    printf("lattice %dx%d, solid %ld, tiles: general %ld fast %ld solid %ld inlet %ld outlet %ld\n", nx, ny, nsolid, cnt[0], cnt[1], cnt[2], cnt[3], cnt[4]);
    Init9<float> iv; const double u0 = 0.06;
    for (int k = 0; k < 9; k++) { const double w = k == 0 ? 4.0 / 9 : (k <= 4 ? 1.0 / 9 : 1.0 / 36); const double eu = ex_of(k) * u0; iv.v[k] = (float)(w * (1 + 3 * eu + 4.5 * eu * eu - 1.5 * u0 * u0)); }
    iv.u0 = (float)u0;
    hipLaunchKernelGGL(k_fill_equilibrium<float>, dim3(2048), dim3(256), 0, st, f0, f1, macro, g, iv);
    CK(hipStreamSynchronize(st));

    const long ntiles = (long)nx * tpc;
    const dim3 grid((unsigned)((ntiles + 3) / 4)), block(256);
    const float tau = 0.58f, U0 = 0.06f;
    std::vector<Variant> vs;
    vs.push_back({"prod step, body mask", [&](float *a, float *b, int r) { step_columns<float, 3>(a, b, macro, mask, tiles, tpc, g, 0, nx, tau, U0, false, r, st); }, {}});
    vs.push_back({"prod step, empty mask", [&](float *a, float *b, int r) { step_columns<float, 3>(a, b, macro, mask_empty, tiles_empty, tpc, g, 0, nx, tau, U0, false, r, st); }, {}});
    vs.push_back({"prod step, body mask (again)", [&](float *a, float *b, int r) { step_columns<float, 3>(a, b, macro, mask, tiles, tpc, g, 0, nx, tau, U0, false, r, st); }, {}});

    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 6;
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
    printf("%-36s %10s %10s %10s\n", "variant", "med us", "min us", "GB/s(med)");
    for (auto &v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        const double med = v.ms[v.ms.size() / 2], mn = v.ms[0];
        printf("%-36s %10.1f %10.1f %10.0f\n", v.name.c_str(), med * 1e3, mn * 1e3, bytes / (med * 1e-3) / 1e9);
    }
    return 0;
}
