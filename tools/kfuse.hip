// kfuse.hip — prototype: TWO lattice steps per launch, register-resident, wave-private windows.
// Each wave owns a window of 256 rows (j) and marches along a chunk of L columns (i): it computes
// step 1 for column c+1 from HBM (9 sixteen-byte loads), keeps the post-collision populations of the
// last three columns in registers, and computes step 2 for column c from them (the +-1 shifts along j
// are lane shuffles; the two window-edge sites are not produced — windows overlap by 4 rows).
// Developer experiment: measures the speed and checks bit-equality with two production steps.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
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

typedef Vec<float> V4;

// step 1 of one column `col` for this lane's 4 rows starting at j0 (FAST semantics + top/bottom far field)
__device__ __forceinline__ void step1_column(const float *__restrict__ s, const Geom &g, long P, int col, int j0, float tau, const float (&feq0)[9], V4 (&G)[9])
{
    const long c = (long)col * g.pitch + j0;
    V4 fin[9];
    fin[0] = vload<float, true>(s + 0 * P + c);
    fin[1] = vload<float, true>(s + 1 * P + c - g.pitch);
    fin[3] = vload<float, true>(s + 3 * P + c + g.pitch);
    fin[2] = vload<float, true, true>(s + 2 * P + c - 1);
    fin[5] = vload<float, true, true>(s + 5 * P + c - g.pitch - 1);
    fin[6] = vload<float, true, true>(s + 6 * P + c + g.pitch - 1);
    fin[4] = vload<float, true, true>(s + 4 * P + c + 1);
    fin[7] = vload<float, true, true>(s + 7 * P + c + g.pitch + 1);
    fin[8] = vload<float, true, true>(s + 8 * P + c - g.pitch + 1);
#pragma unroll
    for (int v = 0; v < 4; v++) {
        float a[9], o[9], rho, ux, uy;
#pragma unroll
        for (int k = 0; k < 9; k++) a[k] = fin[k].v[v];
        collide<float>(a, tau, o, rho, ux, uy);
        const int j = j0 + v;
        const bool far = (j == 0) || (j == g.ny - 1);
#pragma unroll
        for (int k = 0; k < 9; k++) G[k].v[v] = far ? feq0[k] : o[k];
    }
}

__device__ __forceinline__ V4 from_below(const V4 &r)   // value at j-1 (lane 0 / v 0 is garbage)
{
    V4 o;
    o.v[0] = lane_up(r.v[3]);
    o.v[1] = r.v[0]; o.v[2] = r.v[1]; o.v[3] = r.v[2];
    return o;
}
__device__ __forceinline__ V4 from_above(const V4 &r)   // value at j+1 (lane 63 / v 3 is garbage)
{
    V4 o;
    o.v[0] = r.v[1]; o.v[1] = r.v[2]; o.v[2] = r.v[3];
    o.v[3] = lane_down(r.v[0]);
    return o;
}

template <int L>
__global__ __launch_bounds__(256) void k_step2(const float *__restrict__ fs, float *__restrict__ fd, Geom g, int ca, int cb, int nwin, float tau, float U0, int rev)
{
    const int lane = threadIdx.x & 63;
    const int nchunk = (cb - ca + L - 1) / L;
    long unit = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nunits = (long)nchunk * nwin;
    if (unit >= nunits) return;
    if (rev) unit = nunits - 1 - unit;
    const int q = (int)(unit / nwin), w = (int)(unit % nwin);
    const int ia = ca + q * L, ib = min(ia + L, cb);
    const int j0 = w * 252 + lane * 4;
    const float *s = fs + g.pitch;
    float *d = fd + g.pitch;
    const long P = g.plane;
    float feq0[9];
    feq_all<float>(1.0f, U0, 0.0f, feq0);

    V4 Gm[9], Gc[9], Gp[9];          // step-1 results of columns c-1, c, c+1
    step1_column(s, g, P, ia - 1, j0, tau, feq0, Gm);
    step1_column(s, g, P, ia, j0, tau, feq0, Gc);
    const bool first_win = (w == 0);
#pragma unroll 1
    for (int c = ia; c < ib; c++) {
        step1_column(s, g, P, c + 1, j0, tau, feq0, Gp);
        V4 fin[9];
        fin[0] = Gc[0]; fin[1] = Gm[1]; fin[3] = Gp[3];
        fin[2] = from_below(Gc[2]); fin[5] = from_below(Gm[5]); fin[6] = from_below(Gp[6]);
        fin[4] = from_above(Gc[4]); fin[8] = from_above(Gm[8]); fin[7] = from_above(Gp[7]);
        V4 out[9];
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
        const long cc = (long)c * g.pitch + j0;
        if (j0 + 3 < g.ny) {
            if (lane == 0 && !first_win) {
#pragma unroll
                for (int k = 0; k < 9; k++) *reinterpret_cast<float2 *>(d + k * P + cc + 2) = make_float2(out[k].v[2], out[k].v[3]);
            } else if (lane == 63) {
#pragma unroll
                for (int k = 0; k < 9; k++) *reinterpret_cast<float2 *>(d + k * P + cc) = make_float2(out[k].v[0], out[k].v[1]);
            } else {
#pragma unroll
                for (int k = 0; k < 9; k++) vstore<float>(d + k * P + cc, out[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < 9; k++) { Gm[k] = Gc[k]; Gc[k] = Gp[k]; }
    }
}

// ---- variant with software prefetch: the 9 input vectors of column c+2 are requested before the
// ---- arithmetic of column c+1 / c starts, so HBM latency overlaps a whole iteration of VALU work.
__device__ __forceinline__ void load_inputs(const float *__restrict__ s, const Geom &g, long P, int col, int j0, V4 (&fin)[9])
{
    const long c = (long)col * g.pitch + j0;
    fin[0] = vload<float, true>(s + 0 * P + c);
    fin[1] = vload<float, true>(s + 1 * P + c - g.pitch);
    fin[3] = vload<float, true>(s + 3 * P + c + g.pitch);
    fin[2] = vload<float, true, true>(s + 2 * P + c - 1);
    fin[5] = vload<float, true, true>(s + 5 * P + c - g.pitch - 1);
    fin[6] = vload<float, true, true>(s + 6 * P + c + g.pitch - 1);
    fin[4] = vload<float, true, true>(s + 4 * P + c + 1);
    fin[7] = vload<float, true, true>(s + 7 * P + c + g.pitch + 1);
    fin[8] = vload<float, true, true>(s + 8 * P + c - g.pitch + 1);
}
__device__ __forceinline__ void collide_column(const V4 (&fin)[9], const Geom &g, int j0, float tau, const float (&feq0)[9], V4 (&G)[9])
{
#pragma unroll
    for (int v = 0; v < 4; v++) {
        float a[9], o[9], rho, ux, uy;
#pragma unroll
        for (int k = 0; k < 9; k++) a[k] = fin[k].v[v];
        collide<float>(a, tau, o, rho, ux, uy);
        const int j = j0 + v;
        const bool far = (j == 0) || (j == g.ny - 1);
#pragma unroll
        for (int k = 0; k < 9; k++) G[k].v[v] = far ? feq0[k] : o[k];
    }
}

template <int L>
__global__ __launch_bounds__(256) void k_step2p(const float *__restrict__ fs, float *__restrict__ fd, Geom g, int ca, int cb, int nwin, float tau, float U0, int rev)
{
    const int lane = threadIdx.x & 63;
    const int nchunk = (cb - ca + L - 1) / L;
    long unit = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nunits = (long)nchunk * nwin;
    if (unit >= nunits) return;
    if (rev) unit = nunits - 1 - unit;
    const int q = (int)(unit / nwin), w = (int)(unit % nwin);
    const int ia = ca + q * L, ib = min(ia + L, cb);
    const int j0 = w * 252 + lane * 4;
    const float *s = fs + g.pitch;
    float *d = fd + g.pitch;
    const long P = g.plane;
    float feq0[9];
    feq_all<float>(1.0f, U0, 0.0f, feq0);
    const bool first_win = (w == 0);

    V4 G158m[3], G024c[3], G158c[3];      // what step 2 still needs from columns c-1 and c
    V4 in[9], G[9];
    load_inputs(s, g, P, ia - 1, j0, in);
    collide_column(in, g, j0, tau, feq0, G);
    G158m[0] = G[1]; G158m[1] = G[5]; G158m[2] = G[8];
    load_inputs(s, g, P, ia, j0, in);
    collide_column(in, g, j0, tau, feq0, G);
    G024c[0] = G[0]; G024c[1] = G[2]; G024c[2] = G[4];
    G158c[0] = G[1]; G158c[1] = G[5]; G158c[2] = G[8];
    load_inputs(s, g, P, ia + 1, j0, in);                 // in flight while nothing else to do yet
#pragma unroll 1
    for (int c = ia; c < ib; c++) {
        V4 nxt[9];
        const int cn = (c + 2 <= ib) ? c + 2 : c + 1;      // last iteration: harmless re-load
        load_inputs(s, g, P, cn, j0, nxt);                 // prefetch for the NEXT iteration
        collide_column(in, g, j0, tau, feq0, G);           // step 1 of column c+1
        V4 fin[9];
        fin[0] = G024c[0]; fin[1] = G158m[0]; fin[3] = G[3];
        fin[2] = from_below(G024c[1]); fin[5] = from_below(G158m[1]); fin[6] = from_below(G[6]);
        fin[4] = from_above(G024c[2]); fin[8] = from_above(G158m[2]); fin[7] = from_above(G[7]);
        V4 out[9];
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
        const long cc = (long)c * g.pitch + j0;
        if (j0 + 3 < g.ny) {
            if (lane == 0 && !first_win) {
#pragma unroll
                for (int k = 0; k < 9; k++) *reinterpret_cast<float2 *>(d + k * P + cc + 2) = make_float2(out[k].v[2], out[k].v[3]);
            } else if (lane == 63) {
#pragma unroll
                for (int k = 0; k < 9; k++) *reinterpret_cast<float2 *>(d + k * P + cc) = make_float2(out[k].v[0], out[k].v[1]);
            } else {
#pragma unroll
                for (int k = 0; k < 9; k++) vstore<float>(d + k * P + cc, out[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < 3; k++) G158m[k] = G158c[k];
        G024c[0] = G[0]; G024c[1] = G[2]; G024c[2] = G[4];
        G158c[0] = G[1]; G158c[1] = G[5]; G158c[2] = G[8];
#pragma unroll
        for (int k = 0; k < 9; k++) in[k] = nxt[k];
    }
}

// ---- generic variant: NS sites per lane (4 or 2), optional dynamic unit queue ---------------------
template <int NS> struct VecN { float v[NS]; };
template <int NS> __device__ __forceinline__ VecN<NS> ldv(const float *p, bool unaligned)
{
    typedef float VA __attribute__((ext_vector_type(NS)));
    typedef float VU __attribute__((ext_vector_type(NS), aligned(4)));
    VecN<NS> r;
    if (unaligned) { const VU x = __builtin_nontemporal_load(reinterpret_cast<const VU *>(p));
#pragma unroll
        for (int v = 0; v < NS; v++) r.v[v] = x[v]; }
    else { const VA x = __builtin_nontemporal_load(reinterpret_cast<const VA *>(p));
#pragma unroll
        for (int v = 0; v < NS; v++) r.v[v] = x[v]; }
    return r;
}
static __device__ int g_nt_store = 0;   // experiment switch (set from the host)
template <int NS> __device__ __forceinline__ void stv(float *p, const VecN<NS> &r)
{
    typedef float VA __attribute__((ext_vector_type(NS)));
    VA x;
#pragma unroll
    for (int v = 0; v < NS; v++) x[v] = r.v[v];
    if (g_nt_store) __builtin_nontemporal_store(x, reinterpret_cast<VA *>(p));
    else *reinterpret_cast<VA *>(p) = x;
}
template <int NS> __device__ __forceinline__ void load9(const float *__restrict__ s, const Geom &g, long P, int col, int j0, VecN<NS> (&fin)[9])
{
    const long c = (long)col * g.pitch + j0;
    fin[0] = ldv<NS>(s + 0 * P + c, false);
    fin[1] = ldv<NS>(s + 1 * P + c - g.pitch, false);
    fin[3] = ldv<NS>(s + 3 * P + c + g.pitch, false);
    fin[2] = ldv<NS>(s + 2 * P + c - 1, true);
    fin[5] = ldv<NS>(s + 5 * P + c - g.pitch - 1, true);
    fin[6] = ldv<NS>(s + 6 * P + c + g.pitch - 1, true);
    fin[4] = ldv<NS>(s + 4 * P + c + 1, true);
    fin[7] = ldv<NS>(s + 7 * P + c + g.pitch + 1, true);
    fin[8] = ldv<NS>(s + 8 * P + c - g.pitch + 1, true);
}
// aligned 16-B loads only; the +-1 shifts come from the neighbouring lanes (window-edge rows become
// invalid one step earlier, which costs nothing: those rows are not stored anyway)
template <int NS> __device__ __forceinline__ void load9a(const float *__restrict__ s, const Geom &g, long P, int col, int j0, VecN<NS> (&fin)[9])
{
    const long c = (long)col * g.pitch + j0;
    fin[0] = ldv<NS>(s + 0 * P + c, false);
    fin[1] = ldv<NS>(s + 1 * P + c - g.pitch, false);
    fin[3] = ldv<NS>(s + 3 * P + c + g.pitch, false);
    fin[2] = ldv<NS>(s + 2 * P + c, false);
    fin[5] = ldv<NS>(s + 5 * P + c - g.pitch, false);
    fin[6] = ldv<NS>(s + 6 * P + c + g.pitch, false);
    fin[4] = ldv<NS>(s + 4 * P + c, false);
    fin[7] = ldv<NS>(s + 7 * P + c + g.pitch, false);
    fin[8] = ldv<NS>(s + 8 * P + c - g.pitch, false);
}
template <int NS> __device__ __forceinline__ void collideN(const VecN<NS> (&fin)[9], const Geom &g, int j0, float tau, const float (&feq0)[9], VecN<NS> (&G)[9])
{
#pragma unroll
    for (int v = 0; v < NS; v++) {
        float a[9], o[9], rho, ux, uy;
#pragma unroll
        for (int k = 0; k < 9; k++) a[k] = fin[k].v[v];
        collide<float>(a, tau, o, rho, ux, uy);
        const int j = j0 + v;
        const bool far = (j == 0) || (j == g.ny - 1);
#pragma unroll
        for (int k = 0; k < 9; k++) G[k].v[v] = far ? feq0[k] : o[k];
    }
}
template <int NS> __device__ __forceinline__ VecN<NS> belowN(const VecN<NS> &r)
{
    VecN<NS> o; o.v[0] = lane_up(r.v[NS - 1]);
#pragma unroll
    for (int v = 1; v < NS; v++) o.v[v] = r.v[v - 1];
    return o;
}
template <int NS> __device__ __forceinline__ VecN<NS> aboveN(const VecN<NS> &r)
{
    VecN<NS> o;
#pragma unroll
    for (int v = 0; v < NS - 1; v++) o.v[v] = r.v[v + 1];
    o.v[NS - 1] = lane_down(r.v[0]);
    return o;
}

// QUEUE: units are taken from a global atomic counter by persistent waves (grid = resident capacity)
template <int NS, bool QUEUE, int MINW, bool ALIGNED = false, int ORDER = 0>
__global__ __launch_bounds__(256, MINW) void k_step2g(const float *__restrict__ fs, float *__restrict__ fd, Geom g, int ca, int cb, int L, int nwin,
                                                      float tau, float U0, int rev, unsigned int *__restrict__ counter)
{
    constexpr int WS = 64 * NS - 4;             // rows a window advances by
    const int lane = threadIdx.x & 63;
    const int nchunk = (cb - ca + L - 1) / L;
    const long nunits = (long)nchunk * nwin;
    const float *s = fs + g.pitch;
    float *d = fd + g.pitch;
    const long P = g.plane;
    float feq0[9];
    feq_all<float>(1.0f, U0, 0.0f, feq0);
#define FIXUP(v) do { v[2] = belowN<NS>(v[2]); v[5] = belowN<NS>(v[5]); v[6] = belowN<NS>(v[6]); v[4] = aboveN<NS>(v[4]); v[7] = aboveN<NS>(v[7]); v[8] = aboveN<NS>(v[8]); } while (0)
#define LOADRAW(col, v) do { if (ALIGNED) load9a<NS>(s, g, P, (col), j0, v); else load9<NS>(s, g, P, (col), j0, v); } while (0)
#define LOADIN(col, v) do { LOADRAW(col, v); } while (0)
    long unit = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (;;) {
        if (QUEUE) {
            unsigned int u = 0;
            if (lane == 0) u = atomicAdd(counter, 1u);
            unit = __builtin_amdgcn_readfirstlane(u);
        }
        if (unit >= nunits) return;
        const int xflags = rev >> 1;
        const long uu = (rev & 1) ? nunits - 1 - unit : unit;
        const int q = (int)(uu / nwin), w = (int)(uu % nwin);
        const int ia = ca + q * L, ib = min(ia + L, cb);
        const int j0 = w * WS + lane * NS;
        const bool first_win = (w == 0);
        VecN<NS> G158m[3], G024c[3], G158c[3], in[9], G[9];
        LOADIN(ia - 1, in);
        if (ALIGNED) { FIXUP(in); }
        collideN<NS>(in, g, j0, tau, feq0, G);
        G158m[0] = G[1]; G158m[1] = G[5]; G158m[2] = G[8];
        LOADIN(ia, in);
        if (ALIGNED) { FIXUP(in); }
        collideN<NS>(in, g, j0, tau, feq0, G);
        G024c[0] = G[0]; G024c[1] = G[2]; G024c[2] = G[4];
        G158c[0] = G[1]; G158c[1] = G[5]; G158c[2] = G[8];
        LOADIN(ia + 1, in);
#pragma unroll 1
        for (int c = ia; c < ib; c++) {
            VecN<NS> nxt[9];
            if (ORDER == 0 || ORDER == 2) { if (!(xflags & 2)) LOADRAW((c + 2 <= ib) ? c + 2 : c + 1, nxt); else { for (int k = 0; k < 9; k++) nxt[k] = in[k]; } }
            if (ALIGNED) { FIXUP(in); }
            collideN<NS>(in, g, j0, tau, feq0, G);
            if (ORDER == 1) { __builtin_amdgcn_sched_barrier(0); LOADRAW((c + 2 <= ib) ? c + 2 : c + 1, nxt); __builtin_amdgcn_sched_barrier(0); }
            VecN<NS> fin[9], out[9];
            fin[0] = G024c[0]; fin[1] = G158m[0]; fin[3] = G[3];
            fin[2] = belowN<NS>(G024c[1]); fin[5] = belowN<NS>(G158m[1]); fin[6] = belowN<NS>(G[6]);
            fin[4] = aboveN<NS>(G024c[2]); fin[8] = aboveN<NS>(G158m[2]); fin[7] = aboveN<NS>(G[7]);
            collideN<NS>(fin, g, j0, tau, feq0, out);
            const long cc = (long)c * g.pitch + j0;
            if (j0 + NS - 1 < g.ny && !(xflags & 1)) {
                if (NS == 4) {
                    if (lane == 0 && !first_win) {
#pragma unroll
                        for (int k = 0; k < 9; k++) *reinterpret_cast<float2 *>(d + k * P + cc + 2) = make_float2(out[k].v[NS - 2], out[k].v[NS - 1]);
                    } else if (lane == 63) {
#pragma unroll
                        for (int k = 0; k < 9; k++) *reinterpret_cast<float2 *>(d + k * P + cc) = make_float2(out[k].v[0], out[k].v[1]);
                    } else {
#pragma unroll
                        for (int k = 0; k < 9; k++) stv<NS>(d + k * P + cc, out[k]);
                    }
                } else {   // NS == 2: lane 0 (unless first window) and lane 63 produce nothing
                    if (!((lane == 0 && !first_win) || lane == 63)) {
#pragma unroll
                        for (int k = 0; k < 9; k++) stv<NS>(d + k * P + cc, out[k]);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 3; k++) G158m[k] = G158c[k];
            G024c[0] = G[0]; G024c[1] = G[2]; G024c[2] = G[4];
            G158c[0] = G[1]; G158c[1] = G[5]; G158c[2] = G[8];
            if (ORDER == 2) {
                // materialise the copies HERE: the compiler then waits for the prefetched loads with a COUNTED
                // vmcnt (the 9 stores issued after them stay in flight) instead of vmcnt(0) at the loop top
#pragma unroll
                for (int k = 0; k < 9; k++)
#pragma unroll
                    for (int v = 0; v < NS; v++) asm volatile("v_mov_b32 %0, %1" : "=v"(in[k].v[v]) : "v"(nxt[k].v[v]));
            } else {
#pragma unroll
                for (int k = 0; k < 9; k++) in[k] = nxt[k];
            }
        }
        if (!QUEUE) return;
    }
}

// ---- variant with hand-counted waits: the 9 prefetch loads are inline asm (invisible to hipcc's waitcnt
// ---- pass, which otherwise drains the 9..27 stores of the iteration too), waited for with vmcnt(9) at the
// ---- END of the iteration: the loads are older than at least 9 stores, so "all but the 9 youngest" covers
// ---- them, and the main group of stores (issued last) stays in flight across the loop boundary.
typedef float f4v __attribute__((ext_vector_type(4)));
struct In9 { f4v a[9]; };
__device__ __forceinline__ void asm_load9(const float *__restrict__ s, const Geom &g, long P, int col, int j0, In9 &x)
{
    const long c = (long)col * g.pitch + j0;
    const float *p0 = s + 0 * P + c, *p1 = s + 1 * P + c - g.pitch, *p3 = s + 3 * P + c + g.pitch;
    const float *p2 = s + 2 * P + c - 1, *p5 = s + 5 * P + c - g.pitch - 1, *p6 = s + 6 * P + c + g.pitch - 1;
    const float *p4 = s + 4 * P + c + 1, *p7 = s + 7 * P + c + g.pitch + 1, *p8 = s + 8 * P + c - g.pitch + 1;
#define ALD(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(ptr) : "memory")
    ALD(x.a[0], p0); ALD(x.a[1], p1); ALD(x.a[3], p3); ALD(x.a[2], p2); ALD(x.a[5], p5); ALD(x.a[6], p6); ALD(x.a[4], p4); ALD(x.a[7], p7); ALD(x.a[8], p8);
#undef ALD
}
#define WAIT_IN9(N, x) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(x.a[0]), "+v"(x.a[1]), "+v"(x.a[2]), "+v"(x.a[3]), "+v"(x.a[4]), "+v"(x.a[5]), "+v"(x.a[6]), "+v"(x.a[7]), "+v"(x.a[8]) :: "memory")
__device__ __forceinline__ void in9_to_vec(const In9 &x, V4 (&fin)[9])
{
#pragma unroll
    for (int k = 0; k < 9; k++) { fin[k].v[0] = x.a[k][0]; fin[k].v[1] = x.a[k][1]; fin[k].v[2] = x.a[k][2]; fin[k].v[3] = x.a[k][3]; }
}

__device__ __forceinline__ void fused_iteration(const In9 &cur, const float *__restrict__ s, float *__restrict__ d, const Geom &g, long P, int c, int j0, int lane,
                                                bool first_win, float tau, const float (&feq0)[9], V4 (&G158m)[3], V4 (&G024c)[3], V4 (&G158c)[3])
{
    V4 in[9], G[9];
    in9_to_vec(cur, in);
    collide_column(in, g, j0, tau, feq0, G);                       // step 1 of column c+1
    V4 fin[9];
    fin[0] = G024c[0]; fin[1] = G158m[0]; fin[3] = G[3];
    fin[2] = from_below(G024c[1]); fin[5] = from_below(G158m[1]); fin[6] = from_below(G[6]);
    fin[4] = from_above(G024c[2]); fin[8] = from_above(G158m[2]); fin[7] = from_above(G[7]);
    V4 out[9];
    collide_column(fin, g, j0, tau, feq0, out);                    // step 2 of column c
    const long cc = (long)c * g.pitch + j0;
    if (j0 + 3 < g.ny) {
        // edge lanes first, the main group LAST (it is the one left in flight by the counted wait)
        if (lane == 0 && !first_win) {
#pragma unroll
            for (int k = 0; k < 9; k++) *reinterpret_cast<float2 *>(d + k * P + cc + 2) = make_float2(out[k].v[2], out[k].v[3]);
        }
        if (lane == 63) {
#pragma unroll
            for (int k = 0; k < 9; k++) *reinterpret_cast<float2 *>(d + k * P + cc) = make_float2(out[k].v[0], out[k].v[1]);
        }
        if (lane != 63 && !(lane == 0 && !first_win)) {
#pragma unroll
            for (int k = 0; k < 9; k++) vstore<float>(d + k * P + cc, out[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) G158m[k] = G158c[k];
    G024c[0] = G[0]; G024c[1] = G[2]; G024c[2] = G[4];
    G158c[0] = G[1]; G158c[1] = G[5]; G158c[2] = G[8];
}

template <int L>
__global__ __launch_bounds__(256) void k_step2a(const float *__restrict__ fs, float *__restrict__ fd, Geom g, int ca, int cb, int nwin, float tau, float U0, int rev)
{
    const int lane = threadIdx.x & 63;
    const int nchunk = (cb - ca + L - 1) / L;
    long unit = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nunits = (long)nchunk * nwin;
    if (unit >= nunits) return;
    if (rev) unit = nunits - 1 - unit;
    const int q = (int)(unit / nwin), w = (int)(unit % nwin);
    const int ia = ca + q * L, ib = min(ia + L, cb);
    const int j0 = w * 252 + lane * 4;
    const float *s = fs + g.pitch;
    float *d = fd + g.pitch;
    const long P = g.plane;
    float feq0[9];
    feq_all<float>(1.0f, U0, 0.0f, feq0);
    const bool first_win = (w == 0);
    V4 G158m[3], G024c[3], G158c[3];
    {
        V4 in[9], G[9];
        load_inputs(s, g, P, ia - 1, j0, in);
        collide_column(in, g, j0, tau, feq0, G);
        G158m[0] = G[1]; G158m[1] = G[5]; G158m[2] = G[8];
        load_inputs(s, g, P, ia, j0, in);
        collide_column(in, g, j0, tau, feq0, G);
        G024c[0] = G[0]; G024c[1] = G[2]; G024c[2] = G[4];
        G158c[0] = G[1]; G158c[1] = G[5]; G158c[2] = G[8];
    }
    In9 X, Y;
    asm_load9(s, g, P, ia + 1, j0, X);
    WAIT_IN9(0, X);                                  // nothing younger yet: plain drain before the loop
    int c = ia;
#pragma unroll 1
    for (; c + 1 < ib; c += 2) {
        asm_load9(s, g, P, c + 2, j0, Y);            // for iteration c+1
        fused_iteration(X, s, d, g, P, c, j0, lane, first_win, tau, feq0, G158m, G024c, G158c);
        WAIT_IN9(9, Y);                              // Y is older than the >= 9 stores just issued
        asm_load9(s, g, P, (c + 3 <= ib) ? c + 3 : c + 2, j0, X);   // for iteration c+2 (harmless re-load at the end)
        fused_iteration(Y, s, d, g, P, c + 1, j0, lane, first_win, tau, feq0, G158m, G024c, G158c);
        WAIT_IN9(9, X);
    }
    if (c < ib) fused_iteration(X, s, d, g, P, c, j0, lane, first_win, tau, feq0, G158m, G024c, G158c);
}

int main(int argc, char **argv)
{
    const int nx = argc > 1 ? atoi(argv[1]) : 4096, ny = argc > 2 ? atoi(argv[2]) : 4096, rounds = argc > 3 ? atoi(argv[3]) : 10;
    Geom g; g.nxl = nx; g.ny = ny; g.gi0 = 0; g.nx_g = nx; g.pitch = ((long)ny + 255) / 256 * 256;
    g.plane = (((long)(nx + 2) * g.pitch * 4 + 4095) / 4096 * 4096 + 17408) / 4;
    const int tpc = (int)(g.pitch / 256);
    const size_t lat = (size_t)9 * g.plane * 4;
    float *f0, *f1, *f2, *f3, *macro; uint8_t *mask, *tiles;
    CK(hipMalloc(&f0, lat)); CK(hipMalloc(&f1, lat)); CK(hipMalloc(&f2, lat)); CK(hipMalloc(&f3, lat));
    CK(hipMalloc(&macro, (size_t)3 * nx * g.pitch * 4));
    CK(hipMalloc(&mask, (size_t)(nx + 2) * g.pitch)); CK(hipMalloc(&tiles, (size_t)nx * tpc));
    CK(hipMemset(mask, 0, (size_t)(nx + 2) * g.pitch));
    hipStream_t st; CK(hipStreamCreate(&st));
    classify_tiles(mask, tiles, g, tpc, st);
    // non-trivial initial state: pseudo-random perturbation of the equilibrium
    {
        std::vector<float> h((size_t)9 * g.plane);
        const double u0 = 0.06;
        unsigned long long x = 88172645463325252ULL;
        for (int k = 0; k < 9; k++) {
            const double wgt = k == 0 ? 4.0 / 9 : (k <= 4 ? 1.0 / 9 : 1.0 / 36); const double eu = ex_of(k) * u0;
            const float base = (float)(wgt * (1 + 3 * eu + 4.5 * eu * eu - 1.5 * u0 * u0));
            for (long t = 0; t < g.plane; t++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[(size_t)k * g.plane + t] = base * (1.0f + 0.02f * ((x >> 40) / 16777216.0f - 0.5f)); }
        }
        CK(hipMemcpy(f0, h.data(), lat, hipMemcpyHostToDevice));
    }
    const float tau = 0.58f, U0 = 0.06f;
    // reference: two production steps f0 -> f1 -> f2
    step_columns<float, 3>(f0, f1, macro, mask, tiles, tpc, g, 0, nx, tau, U0, false, 0, st);
    step_columns<float, 3>(f1, f2, macro, mask, tiles, tpc, g, 0, nx, tau, U0, false, 1, st);
    CK(hipStreamSynchronize(st));
    const int ca = 2, cb = nx - 2, nwin = (ny - 2 + 251) / 252;
    auto launch2 = [&](auto kern, int L, const float *a, float *b, int rev) {
        const int nchunk = (cb - ca + L - 1) / L;
        const long nunits = (long)nchunk * nwin;
        hipLaunchKernelGGL(kern, dim3((unsigned)((nunits + 3) / 4)), dim3(256), 0, st, a, b, g, ca, cb, nwin, tau, U0, rev);
    };
    CK(hipMemset(f3, 0, lat));
    launch2(k_step2p<16>, 16, f0, f3, 0);
    CK(hipStreamSynchronize(st)); CK(hipGetLastError());
    {
        std::vector<float> a((size_t)g.plane), b((size_t)g.plane);
        long bad = 0, checked = 0;
        for (int k = 0; k < 9; k++) {
            CK(hipMemcpy(a.data(), f2 + (size_t)k * g.plane, g.plane * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(b.data(), f3 + (size_t)k * g.plane, g.plane * 4, hipMemcpyDeviceToHost));
            for (int i = ca; i < cb; i++) for (int j = 0; j < ny; j++) {
                const size_t o = (size_t)(i + 1) * g.pitch + j;
                checked++;
                if (memcmp(&a[o], &b[o], 4) != 0) { if (bad < 5) printf("mismatch k=%d i=%d j=%d: %g vs %g\n", k, i, j, a[o], b[o]); bad++; }
            }
        }
        printf("fused vs 2x production: %ld / %ld values differ\n", bad, checked);
    }
    unsigned int *counter; CK(hipMalloc(&counter, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto check = [&](const char *name) {
        std::vector<float> a((size_t)g.plane), b((size_t)g.plane);
        long bad = 0;
        for (int k = 0; k < 9; k++) {
            CK(hipMemcpy(a.data(), f2 + (size_t)k * g.plane, g.plane * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(b.data(), f3 + (size_t)k * g.plane, g.plane * 4, hipMemcpyDeviceToHost));
            for (int i = ca; i < cb; i++) for (int j = 0; j < ny; j++) { const size_t o = (size_t)(i + 1) * g.pitch + j; if (memcmp(&a[o], &b[o], 4) != 0) bad++; }
        }
        printf("check %-22s: %ld values differ\n", name, bad);
    };
    {
        const int L = 12;
        for (int variant = 0; variant < 2; variant++) {
            CK(hipMemset(f3, 0, lat)); CK(hipDeviceSynchronize());
            const int NS = variant == 0 ? 2 : 4; const int WS = 64 * NS - 4; const int nw = (ny - 2 + WS - 1) / WS;
            const long nunits = (long)((cb - ca + L - 1) / L) * nw;
            CK(hipMemsetAsync(counter, 0, 4, st));
            if (variant == 0) hipLaunchKernelGGL((k_step2g<2, true, 1>), dim3(1024), dim3(256), 0, st, f0, f3, g, ca, cb, L, nw, tau, U0, 0, counter);
            else hipLaunchKernelGGL((k_step2g<4, false, 1, false, 2>), dim3((unsigned)((nunits + 3) / 4)), dim3(256), 0, st, f0, f3, g, ca, cb, L, nw, tau, U0, 1, counter);
            CK(hipStreamSynchronize(st)); CK(hipGetLastError());
            check(variant == 0 ? "NS=2 queue L=12" : "NS=4 counted-wait L=12 rev");
        }
    }
    { CK(hipMemset(f3, 0, lat)); CK(hipDeviceSynchronize()); launch2(k_step2a<24>, 24, f0, f3, 0); CK(hipStreamSynchronize(st)); CK(hipGetLastError()); check("counted waits L=24");
      CK(hipMemset(f3, 0, lat)); CK(hipDeviceSynchronize()); launch2(k_step2a<35>, 35, f0, f3, 1); CK(hipStreamSynchronize(st)); check("counted waits L=35 rev"); }
    struct Var { std::string name; std::function<void(const float *, float *, int)> fn; std::vector<float> ms; int steps; };
    std::vector<Var> vs;
    vs.push_back({"production, 2 launches", [&](const float *a, float *b, int r) { step_columns<float, 3>(a, f1, macro, mask, tiles, tpc, g, 0, nx, tau, U0, false, 0, st); step_columns<float, 3>(f1, b, macro, mask, tiles, tpc, g, 0, nx, tau, U0, false, 1, st); }, {}, 2});
    vs.push_back({"fused+prefetch L=16", [&](const float *a, float *b, int r) { launch2(k_step2p<16>, 16, a, b, r); }, {}, 2});
    vs.push_back({"fused+prefetch L=24", [&](const float *a, float *b, int r) { launch2(k_step2p<24>, 24, a, b, r); }, {}, 2});
    auto launchg = [&](auto kern, int NS, bool queue, int L, int wavesPerSimd, const float *a, float *b, int rev) {
        const int WS = 64 * NS - 4;
        const int nw = (ny - 2 + WS - 1) / WS;
        const int nchunk = (cb - ca + L - 1) / L;
        const long nunits = (long)nchunk * nw;
        long blocks = (nunits + 3) / 4;
        if (queue) { CK(hipMemsetAsync(counter, 0, 4, st)); blocks = 256L * wavesPerSimd; if (blocks * 4 > nunits) blocks = (nunits + 3) / 4; }
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, st, a, b, g, ca, cb, L, nw, tau, U0, rev, counter);
    };
    vs.push_back({"counted waits (asm) L=12", [&](const float *a, float *b, int r) { launch2(k_step2a<12>, 12, a, b, r); }, {}, 2});
    vs.push_back({"counted waits (asm) L=24", [&](const float *a, float *b, int r) { launch2(k_step2a<24>, 24, a, b, r); }, {}, 2});
    vs.push_back({"counted waits (asm) L=35", [&](const float *a, float *b, int r) { launch2(k_step2a<35>, 35, a, b, r); }, {}, 2});
    const int reps = 4;
    for (int r = 0; r < rounds + 2; r++)
        for (auto &v : vs) {
            CK(hipEventRecord(e0, st));
            for (int q = 0; q < reps; q++) { if (q & 1) v.fn(f3, f0, 1); else v.fn(f0, f3, 0); }
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) v.ms.push_back(ms / reps / v.steps);
        }
    printf("%-28s %12s %12s %10s\n", "variant", "us per STEP", "min", "GLUPS");
    for (auto &v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        const double med = v.ms[v.ms.size() / 2];
        printf("%-28s %12.1f %12.1f %10.1f\n", v.name.c_str(), med * 1e3, v.ms[0] * 1e3, (double)nx * ny / (med * 1e-3) / 1e9);
    }
    return 0;
}
