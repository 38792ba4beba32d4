// kstorehazard.hip — is there a store-data hazard behind buffer_store_dwordx4 with an SGPR soffset on gfx950?
//
// Round 2 saw wrong macro values in 4-lane groups at the outlet column of a 16384 x 4096 lattice, not on every run, and attributed them to a
// buffer_store_dwordx4 whose data VGPRs were overwritten by the next VALU instruction (store_data_fence() in csrc/step_march.hpp; LLVM pads
// that overwrite only for stores WITHOUT a register soffset).  This micro-kernel isolates the claim: every wave stores {a,a,a,a} tuples with
//     buffer_store_dwordx4 v[10:13], voff, rsrc, SOFF offen        (SOFF: an SGPR, or the immediate 0)
// and overwrites v11 and v13 with b in the very next instructions (PAD wait states in between); thousands of waves do that back to back so that
// the memory pipe is backed up.  Afterwards the host counts 16-byte slots that do not hold {a,a,a,a}.
//     hipcc --offload-arch=gfx950 -O3 -o tools/kstorehazard tools/kstorehazard.hip && ./tools/kstorehazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int PAD, bool SGPR_SOFF>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned bytes, int iters, unsigned a, unsigned b)
{
    const u4 rsrc = {(unsigned)(unsigned long long)out, (unsigned)((unsigned long long)out >> 32), bytes, 0x00020000u};
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    for (int it = 0; it < iters; it++) {
        const unsigned voff = lane * 16u;
        const unsigned soff = __builtin_amdgcn_readfirstlane((wave * (unsigned)iters + (unsigned)it) * 1024u);
        if (SGPR_SOFF) {
            asm volatile("v_mov_b32 v10, %[a]\n\tv_mov_b32 v11, %[a]\n\tv_mov_b32 v12, %[a]\n\tv_mov_b32 v13, %[a]\n\ts_nop 4\n\t"
                         "buffer_store_dwordx4 v[10:13], %[voff], %[rsrc], %[soff] offen\n\t"
                         "s_nop %[pad]\n\t"
                         "v_mov_b32 v11, %[b]\n\tv_mov_b32 v13, %[b]\n\tv_mov_b32 v10, %[b]\n\tv_mov_b32 v12, %[b]\n\t"
                         :: [a] "v"(a), [b] "v"(b), [voff] "v"(voff), [rsrc] "s"(rsrc), [soff] "s"(soff), [pad] "n"(PAD > 0 ? PAD - 1 : 0)
                         : "v10", "v11", "v12", "v13", "memory");
        } else {
            const unsigned v2 = voff + soff;
            asm volatile("v_mov_b32 v10, %[a]\n\tv_mov_b32 v11, %[a]\n\tv_mov_b32 v12, %[a]\n\tv_mov_b32 v13, %[a]\n\ts_nop 4\n\t"
                         "buffer_store_dwordx4 v[10:13], %[voff], %[rsrc], 0 offen\n\t"
                         "s_nop %[pad]\n\t"
                         "v_mov_b32 v11, %[b]\n\tv_mov_b32 v13, %[b]\n\tv_mov_b32 v10, %[b]\n\tv_mov_b32 v12, %[b]\n\t"
                         :: [a] "v"(a), [b] "v"(b), [voff] "v"(v2), [rsrc] "s"(rsrc), [pad] "n"(PAD > 0 ? PAD - 1 : 0)
                         : "v10", "v11", "v12", "v13", "memory");
        }
    }
}
// PAD = 0 must issue NO s_nop at all: a separate instantiation without the instruction
template <bool SGPR_SOFF>
__global__ __launch_bounds__(256) void k0(unsigned *out, unsigned bytes, int iters, unsigned a, unsigned b)
{
    const u4 rsrc = {(unsigned)(unsigned long long)out, (unsigned)((unsigned long long)out >> 32), bytes, 0x00020000u};
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    for (int it = 0; it < iters; it++) {
        const unsigned voff = lane * 16u;
        const unsigned soff = __builtin_amdgcn_readfirstlane((wave * (unsigned)iters + (unsigned)it) * 1024u);
        if (SGPR_SOFF) {
            asm volatile("v_mov_b32 v10, %[a]\n\tv_mov_b32 v11, %[a]\n\tv_mov_b32 v12, %[a]\n\tv_mov_b32 v13, %[a]\n\ts_nop 4\n\t"
                         "buffer_store_dwordx4 v[10:13], %[voff], %[rsrc], %[soff] offen\n\t"
                         "v_mov_b32 v11, %[b]\n\tv_mov_b32 v13, %[b]\n\tv_mov_b32 v10, %[b]\n\tv_mov_b32 v12, %[b]\n\t"
                         :: [a] "v"(a), [b] "v"(b), [voff] "v"(voff), [rsrc] "s"(rsrc), [soff] "s"(soff) : "v10", "v11", "v12", "v13", "memory");
        } else {
            const unsigned v2 = voff + soff;
            asm volatile("v_mov_b32 v10, %[a]\n\tv_mov_b32 v11, %[a]\n\tv_mov_b32 v12, %[a]\n\tv_mov_b32 v13, %[a]\n\ts_nop 4\n\t"
                         "buffer_store_dwordx4 v[10:13], %[voff], %[rsrc], 0 offen\n\t"
                         "v_mov_b32 v11, %[b]\n\tv_mov_b32 v13, %[b]\n\tv_mov_b32 v10, %[b]\n\tv_mov_b32 v12, %[b]\n\t"
                         :: [a] "v"(a), [b] "v"(b), [voff] "v"(v2), [rsrc] "s"(rsrc) : "v10", "v11", "v12", "v13", "memory");
        }
    }
}

int main()
{
    const int blocks = 2048, iters = 64;                                  // 8192 waves x 64 stores x 1 KiB = 512 MiB
    const size_t words = (size_t)blocks * 4 * iters * 256;
    unsigned *d = nullptr;
    if (hipMalloc(&d, words * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    std::vector<unsigned> h(words);
    const unsigned a = 0x11111111u, b = 0xEEEEEEEEu;
    auto run = [&](const char *name, auto launch) {
        long bad_total = 0, bad_runs = 0;
        for (int rep = 0; rep < 6; rep++) {
            hipMemset(d, 0, words * 4);
            launch();
            hipDeviceSynchronize();
            hipMemcpy(h.data(), d, words * 4, hipMemcpyDeviceToHost);
            long bad = 0, firstbad = -1;
            for (size_t i = 0; i < words; i++)
                if (h[i] != a) { bad++; if (firstbad < 0) firstbad = (long)i; }
            bad_total += bad; bad_runs += bad != 0;
            if (bad && rep < 2) printf("    rep %d: %ld wrong words, first at word %ld (lane %ld of its wave, component %ld): 0x%08x\n", rep, bad, firstbad, (firstbad / 4) % 64, firstbad % 4, h[firstbad]);
        }
        printf("%-64s wrong words in 6 runs: %ld (%ld runs affected)\n", name, bad_total, bad_runs);
    };
    const unsigned bytes = (unsigned)(words * 4 > 0xffffffffull ? 0xffffffffu : words * 4);
    run("SGPR soffset, overwrite in the next instruction (no pad)", [&] { hipLaunchKernelGGL((k0<true>), dim3(blocks), dim3(256), 0, 0, d, bytes, iters, a, b); });
    run("SGPR soffset, s_nop 0 (one wait state)", [&] { hipLaunchKernelGGL((k<1, true>), dim3(blocks), dim3(256), 0, 0, d, bytes, iters, a, b); });
    run("SGPR soffset, s_nop 1 (two wait states: store_data_fence)", [&] { hipLaunchKernelGGL((k<2, true>), dim3(blocks), dim3(256), 0, 0, d, bytes, iters, a, b); });
    run("immediate soffset 0, overwrite in the next instruction (no pad)", [&] { hipLaunchKernelGGL((k0<false>), dim3(blocks), dim3(256), 0, 0, d, bytes, iters, a, b); });
    run("immediate soffset 0, s_nop 1", [&] { hipLaunchKernelGGL((k<2, false>), dim3(blocks), dim3(256), 0, 0, d, bytes, iters, a, b); });
    hipFree(d);
    return 0;
}
