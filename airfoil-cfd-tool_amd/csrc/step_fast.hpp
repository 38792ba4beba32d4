// step_fast.hpp — the fused D2Q9 step kernel (STEP_FS main(), html:283-360) for gfx950.
//
// One WAVE (64 lanes) owns one TILE: VEC consecutive sites per lane along the fast axis j
// (VEC = 16 B / sizeof(T): 4 for fp32, 2 for fp64), i.e. 64*VEC sites of one column i.  Every
// tile has a class, computed once per mask upload by k_classify with wave ballots:
//   TILE_FAST    no solid site in the tile's 3 x (64*VEC+2) neighbourhood, not an inlet/outlet
//                column: straight-line path, 9 aligned 16-B loads + 9 aligned 16-B stores per
//                lane; the +-1 shifts along j are done in registers (lane shuffles; only the
//                two edge lanes of the wave issue an extra scalar load);
//   TILE_SOLID   every site solid: population swap (html:287-294), aligned vector copy;
//   TILE_INLET / TILE_OUTLET  far-field / zero-gradient columns without solids (html:301-322);
//   TILE_GENERAL everything else (body surface, ragged last tile): per-site code, all branches.
// The class is wave-uniform (read through readfirstlane), so the dispatch is a scalar branch.
//
// HBM traffic: 9 loads + 9 stores of sizeof(T) per site (72 B fp32 / 144 B fp64) + 1 B per
// tile of class; the byte mask is only read by GENERAL tiles.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels.hpp"

namespace wt {

template <typename T> struct VecOf;
template <> struct VecOf<float> { typedef float4 type; static constexpr int N = 4; };
template <> struct VecOf<double> { typedef double2 type; static constexpr int N = 2; };

template <typename T> __host__ __device__ constexpr int tile_j() { return 64 * VecOf<T>::N; }
static inline int tile_j_of(size_t esz) { return esz == 4 ? 256 : 128; }

// --------------------------------------------------------------------------------------------
// tile classification: one wave per tile
// --------------------------------------------------------------------------------------------
// tiles are `tj` rows tall and start every `stride` rows (stride = tj for the step kernel's own grid;
// 252-row stride / 256-row height for the window-aligned grid of the two-step mode, step_fused.hpp)
__global__ __launch_bounds__(256) void k_classify(const uint8_t *__restrict__ mask, uint8_t *__restrict__ tiles, Geom g,
                                                  int tiles_per_col, int tj, int stride)
{
    const int lane = threadIdx.x & 63;
    const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long ntiles = (long)g.nxl * tiles_per_col;
    if (tile >= ntiles) return;
    const int i = (int)(tile / tiles_per_col), jt = (int)(tile % tiles_per_col);
    const int j0 = jt * stride;
    const uint8_t *m = mask + g.pitch;
    int nb = 0, own_any = 0, own_all = 1;
    for (int jj = j0 - 1 + lane; jj <= j0 + tj; jj += 64) {
        if (jj < 0 || jj >= g.ny) continue;
        const int a = m[(long)(i - 1) * g.pitch + jj], b = m[(long)i * g.pitch + jj], c = m[(long)(i + 1) * g.pitch + jj];
        nb |= a | b | c;
        if (jj >= j0 && jj < j0 + tj) { own_any |= b; own_all &= (b != 0); }
    }
    const bool any_nb = __ballot(nb != 0) != 0ULL;
    const bool any_own = __ballot(own_any != 0) != 0ULL;
    const bool all_own = __ballot(own_all == 0) == 0ULL;
    const int gi = i + g.gi0;
    uint8_t cls;
    if (j0 + tj > g.ny) cls = TILE_GENERAL;                    // ragged or padding tile
    else if (gi == 0 && !any_own) cls = TILE_INLET;
    else if (gi == g.nx_g - 1 && !any_own) cls = TILE_OUTLET;
    else if (gi == 0 || gi == g.nx_g - 1) cls = TILE_GENERAL;
    else if (!any_nb) cls = TILE_FAST;
    else if (all_own) cls = TILE_SOLID;
    else cls = TILE_GENERAL;
    if (lane == 0) tiles[tile] = cls;
}

static inline int classify_tiles(const uint8_t *mask, uint8_t *tiles, const Geom &g, int tiles_per_col, hipStream_t st,
                                 int tj = 0, int stride = 0)
{
    if (tj == 0) tj = (int)(g.pitch / tiles_per_col);
    if (stride == 0) stride = tj;
    const long ntiles = (long)g.nxl * tiles_per_col;
    hipLaunchKernelGGL(k_classify, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, st, mask, tiles, g, tiles_per_col, tj, stride);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// --------------------------------------------------------------------------------------------
// vector helpers
// --------------------------------------------------------------------------------------------
template <typename T> struct Vec { T v[VecOf<T>::N]; };

// 16-byte load.  NT: nontemporal ("streamed once") hint — the populations of the source lattice
// are read exactly once per step, so they should not displace the freshly written lattice from
// the Infinity Cache (measured: tools/kbench3, DESIGN.md "Memory system").
// ALIGN4: the address is only element-aligned (the +-1 shifted rows).
template <typename T, bool NT = false, bool ALIGN4 = false>
__device__ __forceinline__ Vec<T> vload(const T *p)
{
    constexpr int N = VecOf<T>::N;
    typedef T VA __attribute__((ext_vector_type(N)));
    typedef T VU __attribute__((ext_vector_type(N), aligned(sizeof(T))));
    Vec<T> r;
    if constexpr (ALIGN4) {
        const VU x = NT ? __builtin_nontemporal_load(reinterpret_cast<const VU *>(p)) : *reinterpret_cast<const VU *>(p);
#pragma unroll
        for (int v = 0; v < N; v++) r.v[v] = x[v];
    } else {
        const VA x = NT ? __builtin_nontemporal_load(reinterpret_cast<const VA *>(p)) : *reinterpret_cast<const VA *>(p);
#pragma unroll
        for (int v = 0; v < N; v++) r.v[v] = x[v];
    }
    return r;
}

template <typename T>
__device__ __forceinline__ void vstore(T *p, const Vec<T> &r)
{
    typedef typename VecOf<T>::type V;
    V x;
    if constexpr (VecOf<T>::N == 4) { x.x = r.v[0]; x.y = r.v[1]; x.z = r.v[2]; x.w = r.v[3]; }
    else { x.x = r.v[0]; x.y = r.v[1]; }
    *reinterpret_cast<V *>(p) = x;
}

// value held by lane-1 (lane 0 keeps its own)
template <typename T> __device__ __forceinline__ T lane_up(T x) { return __shfl_up(x, 1); }
// value held by lane+1 (lane 63 keeps its own)
template <typename T> __device__ __forceinline__ T lane_down(T x) { return __shfl_down(x, 1); }

// sites j0..j0+N-1 need src[j-1]: {edge-or-left.w, r0, r1, ...}
template <typename T>
__device__ __forceinline__ Vec<T> shift_from_below(const Vec<T> &r, const T *p, int lane)
{
    constexpr int N = VecOf<T>::N;
    T left = lane_up(r.v[N - 1]);
    if (lane == 0) left = p[-1];
    Vec<T> o;
    o.v[0] = left;
#pragma unroll
    for (int v = 1; v < N; v++) o.v[v] = r.v[v - 1];
    return o;
}

// sites j0..j0+N-1 need src[j+1]: {r1, r2, ..., right.x-or-edge}
template <typename T>
__device__ __forceinline__ Vec<T> shift_from_above(const Vec<T> &r, const T *p, int lane)
{
    constexpr int N = VecOf<T>::N;
    T right = lane_down(r.v[0]);
    if (lane == 63) right = p[N];
    Vec<T> o;
#pragma unroll
    for (int v = 0; v < N - 1; v++) o.v[v] = r.v[v + 1];
    o.v[N - 1] = right;
    return o;
}

// --------------------------------------------------------------------------------------------
// the step kernel
// --------------------------------------------------------------------------------------------
// LOADMODE bit 0: nontemporal loads of the source lattice; bit 1: the +-1 shifted rows are read
// with element-aligned 16-B loads instead of aligned loads + lane shuffles.
// rev: walk the tiles backwards.  Successive steps alternate the direction so that a step starts
// by reading what the previous step wrote last — still resident in the 256 MB Infinity Cache.
// WIN: the tile belongs to a grid whose tiles start every `stride` rows (tile height stays 64*N) and
// only rows in [st_lo, st_hi) are stored — the window-aligned zone passes of step_fused.hpp.
template <typename T, bool EMIT, int LOADMODE, bool WIN = false>
__device__ __forceinline__ void step_tile(const T *__restrict__ fs, T *__restrict__ fd, T *__restrict__ macro,
                                          const uint8_t *__restrict__ mask, const uint8_t *__restrict__ tiles,
                                          int tiles_per_col, const Geom &g, int i_begin, T tau, T U0, long tile_local, int lane,
                                          int stride = 0, int st_lo = 0, int st_hi = 0)
{
    constexpr int N = VecOf<T>::N;
    constexpr int TJ = 64 * N;
    constexpr bool NT = (LOADMODE & 1) != 0;
    constexpr bool UNALIGNED = (LOADMODE & 2) != 0;
    const int i = i_begin + (int)(tile_local / tiles_per_col);
    const int jt = (int)(tile_local % tiles_per_col);
    const int row0 = WIN ? jt * stride : jt * TJ;
    const int cls = __builtin_amdgcn_readfirstlane((int)tiles[(long)i * tiles_per_col + jt]);
    const T *s = fs + g.pitch;
    T *d = fd + g.pitch;
    const long P = g.plane;
    const long mp = (long)g.nxl * g.pitch;

    if (cls == TILE_GENERAL) {
        const uint8_t *m = mask + g.pitch;
#pragma unroll 1
        for (int v = 0; v < N; v++) {
            const int j = row0 + v * 64 + lane;
            if (j < g.ny && (!WIN || (j >= st_lo && j < st_hi))) site_general<T>(s, d, macro, m, g, i, j, tau, U0, EMIT);
        }
        return;
    }

    const int j0 = row0 + lane * N;
    const long c = (long)i * g.pitch + j0;
    Vec<T> out[9];
    Vec<T> mrho, mux, muy;

    if (cls == TILE_FAST) {
        Vec<T> fin[9];
        // ey = 0: aligned
        fin[0] = vload<T, NT>(s + 0 * P + c);
        fin[1] = vload<T, NT>(s + 1 * P + c - g.pitch);
        fin[3] = vload<T, NT>(s + 3 * P + c + g.pitch);
        const T *p2 = s + 2 * P + c, *p5 = s + 5 * P + c - g.pitch, *p6 = s + 6 * P + c + g.pitch;   // ey=+1: source j-1
        const T *p4 = s + 4 * P + c, *p7 = s + 7 * P + c + g.pitch, *p8 = s + 8 * P + c - g.pitch;   // ey=-1: source j+1
        if constexpr (UNALIGNED) {
            fin[2] = vload<T, NT, true>(p2 - 1); fin[5] = vload<T, NT, true>(p5 - 1); fin[6] = vload<T, NT, true>(p6 - 1);
            fin[4] = vload<T, NT, true>(p4 + 1); fin[7] = vload<T, NT, true>(p7 + 1); fin[8] = vload<T, NT, true>(p8 + 1);
        } else {
            const Vec<T> r2 = vload<T, NT>(p2), r5 = vload<T, NT>(p5), r6 = vload<T, NT>(p6);
            const Vec<T> r4 = vload<T, NT>(p4), r7 = vload<T, NT>(p7), r8 = vload<T, NT>(p8);
            fin[2] = shift_from_below<T>(r2, p2, lane);
            fin[5] = shift_from_below<T>(r5, p5, lane);
            fin[6] = shift_from_below<T>(r6, p6, lane);
            fin[4] = shift_from_above<T>(r4, p4, lane);
            fin[7] = shift_from_above<T>(r7, p7, lane);
            fin[8] = shift_from_above<T>(r8, p8, lane);
        }
        T feq0[9];
        feq_all<T>(T(1.0), U0, T(0.0), feq0);      // far-field populations (html:314-322)
#pragma unroll
        for (int v = 0; v < N; v++) {
            T a[9], o[9], rho, ux, uy;
#pragma unroll
            for (int k = 0; k < 9; k++) a[k] = fin[k].v[v];
            collide<T>(a, tau, o, rho, ux, uy);
            const int j = j0 + v;
            const bool far = (j == 0) || (j == g.ny - 1);   // top / bottom rows
#pragma unroll
            for (int k = 0; k < 9; k++) out[k].v[v] = far ? feq0[k] : o[k];
            mrho.v[v] = far ? T(1.0) : rho;
            mux.v[v] = far ? U0 : ux;
            muy.v[v] = far ? T(0.0) : uy;
        }
    } else if (cls == TILE_SOLID) {
#pragma unroll
        for (int k = 0; k < 9; k++) out[k] = vload<T, NT>(s + opp_of(k) * P + c);
#pragma unroll
        for (int v = 0; v < N; v++) { mrho.v[v] = T(1.0); mux.v[v] = T(0.0); muy.v[v] = T(0.0); }
    } else if (cls == TILE_INLET) {
        T feq0[9];
        feq_all<T>(T(1.0), U0, T(0.0), feq0);
#pragma unroll
        for (int k = 0; k < 9; k++)
#pragma unroll
            for (int v = 0; v < N; v++) out[k].v[v] = feq0[k];
#pragma unroll
        for (int v = 0; v < N; v++) { mrho.v[v] = T(1.0); mux.v[v] = U0; muy.v[v] = T(0.0); }
    } else {   // TILE_OUTLET: copy the un-streamed populations of column i-1 (html:301-312)
#pragma unroll
        for (int k = 0; k < 9; k++) out[k] = vload<T, NT>(s + k * P + c - g.pitch);
#pragma unroll
        for (int v = 0; v < N; v++) {
            const T q0 = out[0].v[v], q1 = out[1].v[v], q2 = out[2].v[v], q3 = out[3].v[v], q4 = out[4].v[v],
                    q5 = out[5].v[v], q6 = out[6].v[v], q7 = out[7].v[v], q8 = out[8].v[v];
            const T rho = q0 + q1 + q2 + q3 + q4 + q5 + q6 + q7 + q8;
            mrho.v[v] = rho;
            mux.v[v] = (q1 + q5 + q8 - q3 - q6 - q7) / rho;
            muy.v[v] = (q2 + q5 + q6 - q4 - q7 - q8) / rho;
        }
    }
    if constexpr (WIN) {
        if (j0 < st_lo || j0 + N > st_hi) {        // window edge lanes: store only the rows inside [st_lo, st_hi)
#pragma unroll
            for (int v = 0; v < N; v++) {
                const int j = j0 + v;
                if (j < st_lo || j >= st_hi) continue;
#pragma unroll
                for (int k = 0; k < 9; k++) d[k * P + c + v] = out[k].v[v];
                if (EMIT) { macro[c + v] = mrho.v[v]; macro[mp + c + v] = mux.v[v]; macro[2 * mp + c + v] = muy.v[v]; }
            }
            return;
        }
    }
#pragma unroll
    for (int k = 0; k < 9; k++) vstore<T>(d + k * P + c, out[k]);
    if (EMIT) {
        vstore<T>(macro + c, mrho);
        vstore<T>(macro + mp + c, mux);
        vstore<T>(macro + 2 * mp + c, muy);
    }
}

// The kernel: one tile per wave, one-shot grid (a grid-stride / capped-grid variant measured 2 %
// slower at 4096^2 and no better on 544-column slabs).
template <typename T, bool EMIT, int LOADMODE>
__global__ __launch_bounds__(256) void k_step(const T *__restrict__ fs, T *__restrict__ fd, T *__restrict__ macro,
                                              const uint8_t *__restrict__ mask, const uint8_t *__restrict__ tiles,
                                              int tiles_per_col, Geom g, int i_begin, int i_end, T tau, T U0, int rev)
{
    const int lane = threadIdx.x & 63;
    const long ntiles = (long)(i_end - i_begin) * tiles_per_col;
    const long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= ntiles) return;
    step_tile<T, EMIT, LOADMODE>(fs, fd, macro, mask, tiles, tiles_per_col, g, i_begin, tau, U0, rev ? ntiles - 1 - t : t, lane);
}

// launch over local columns [i_begin, i_end)
#ifndef WT_LOADMODE
#define WT_LOADMODE 3
#endif
template <typename T, int LOADMODE = WT_LOADMODE>
static inline int step_columns(const T *fs, T *fd, T *macro, const uint8_t *mask, const uint8_t *tiles, int tiles_per_col,
                               const Geom &g, int i_begin, int i_end, T tau, T U0, bool emit, int rev, hipStream_t st)
{
    const long ntiles = (long)(i_end - i_begin) * tiles_per_col;
    const dim3 grid((unsigned)((ntiles + 3) / 4)), block(256);
    if (emit)
        hipLaunchKernelGGL((k_step<T, true, LOADMODE>), grid, block, 0, st, fs, fd, macro, mask, tiles, tiles_per_col, g, i_begin, i_end, tau, U0, rev);
    else
        hipLaunchKernelGGL((k_step<T, false, LOADMODE>), grid, block, 0, st, fs, fd, macro, mask, tiles, tiles_per_col, g, i_begin, i_end, tau, U0, rev);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace wt
