// kifetch.hip — developer experiment: does the LENGTH of a straight-line loop body limit the VALU issue rate?  The marching
// kernel's loop body is ~5000 instructions (~35 KB of code) executed once per iteration by two waves per SIMD; if the
// instruction cache (shared by CUs) cannot stream that fast, the vector pipe starves.  Same work (independent v_fma_f32 on
// 8 / 24 accumulators), loop bodies of 128 ... 16384 instructions, 1 / 2 / 4 waves per SIMD, all CUs busy.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/kifetch.hip -o tools/kifetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

// VOP3 encoding (8 bytes per instruction) with three distinct sources, like most of the collision code
template <int BODY8>      // loop body = BODY8 * 8 instructions
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < BODY8; u++) {
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(a), "v"(b));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(a), "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(a), "v"(b));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x4) : "v"(a), "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x5) : "v"(a), "v"(b));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x6) : "v"(a), "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x7) : "v"(a), "v"(b));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int BODY8> static void run(float *out, hipStream_t st)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long total = 1L << 21;                       // instructions per wave
    const int iters = (int)(total / (BODY8 * 8));
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * wps;
        hipLaunchKernelGGL(k<BODY8>, dim3(blocks), dim3(256), 0, st, out, 2, 1.0001f, 0.5f);
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(k<BODY8>, dim3(blocks), dim3(256), 0, st, out, iters, 1.0001f, 0.5f);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double per_simd = (double)iters * BODY8 * 8 * wps;
        printf("loop body %6d instr (%4d KB)  waves/SIMD=%d: %.3f ms, %.2f ns per wave-instr per SIMD\n", BODY8 * 8, BODY8 * 8 * 8 / 1024, wps, ms, ms * 1e6 / per_simd);
    }
}

int main()
{
    hipStream_t st; CK(hipStreamCreate(&st));
    float *out; CK(hipMalloc(&out, 256 * 1024 * 4 * 4));
    run<16>(out, st); run<128>(out, st); run<512>(out, st); run<1024>(out, st); run<2048>(out, st);
    return 0;
}
