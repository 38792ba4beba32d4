// kvalu.hip — developer experiment: VALU issue rate of gfx950 per SIMD for plain f32 FMA / MUL / ADD / v_pk_fma_f32 at 1, 2, 4
// waves per SIMD (answers: is one wave enough to saturate a SIMD's vector pipe? do packed f32 ops double the rate?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    const f2 pa = {a, a}, pb = {b, b};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (MODE == 0) { x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                             x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b); }
            if (MODE == 1) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(pa), "v"(pb)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(pa), "v"(pb));
                             asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(pa), "v"(pb)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(pa), "v"(pb)); }
            if (MODE == 2) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x0) : "v"(a)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(x1) : "v"(b)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x2) : "v"(a)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(x3) : "v"(b));
                             asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x4) : "v"(a)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(x5) : "v"(b)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x6) : "v"(a)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(x7) : "v"(b)); }
            if (MODE == 3) { asm volatile("v_rcp_f32 %0, %0" : "+v"(x0)); asm volatile("v_rcp_f32 %0, %0" : "+v"(x1)); asm volatile("v_rcp_f32 %0, %0" : "+v"(x2)); asm volatile("v_rcp_f32 %0, %0" : "+v"(x3));
                             asm volatile("v_rcp_f32 %0, %0" : "+v"(x4)); asm volatile("v_rcp_f32 %0, %0" : "+v"(x5)); asm volatile("v_rcp_f32 %0, %0" : "+v"(x6)); asm volatile("v_rcp_f32 %0, %0" : "+v"(x7)); }
            if (MODE == 4) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x0) : "v"(a)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x1) : "v"(a)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x2) : "v"(a)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x3) : "v"(a));
                             asm volatile("v_mov_b32 %0, %1" : "=v"(x4) : "v"(x5)); asm volatile("v_mov_b32 %0, %1" : "=v"(x5) : "v"(x6)); asm volatile("v_mov_b32 %0, %1" : "=v"(x6) : "v"(x7)); asm volatile("v_mov_b32 %0, %1" : "=v"(x7) : "v"(x4)); }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE> static void run(const char *name, int ninstr_per_u, float *out, hipStream_t st)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000;
    for (int wps : {1, 2, 4}) {           // waves per SIMD: blocks of 256 threads = one wave per SIMD; wps blocks per CU
        const int blocks = 256 * wps;
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, st, out, 10, 1.0001f, 0.5f);
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, st, out, iters, 1.0001f, 0.5f);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double instr = (double)iters * 16 * ninstr_per_u;        // per wave
        const double per_simd = instr * wps;                            // wave-instructions per SIMD
        printf("%-28s waves/SIMD=%d: %.3f ms, %.2f ns per wave-instr per SIMD -> %.2f cycles at 2.4 GHz (%.2f at 2.1)\n", name, wps, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4, ms * 1e6 / per_simd * 2.1);
    }
}

int main()
{
    hipStream_t st; CK(hipStreamCreate(&st));
    float *out; CK(hipMalloc(&out, 256 * 1024 * 4 * 4));
    run<0>("v_fma_f32 x8 independent", 8, out, st);
    run<1>("v_pk_fma_f32 x4 independent", 4, out, st);
    run<2>("v_mul/v_add x8", 8, out, st);
    run<3>("v_rcp_f32 x8", 8, out, st);
    run<4>("v_cndmask/v_mov x8", 8, out, st);
    return 0;
}
