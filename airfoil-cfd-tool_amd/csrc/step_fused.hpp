// step_fused.hpp — TWO lattice steps per launch (opt-in: WT_FUSE2=1 or wt_set_option).
//
// Temporal fusion without LDS: a wave owns a WINDOW of 256 rows (j) and marches along a CHUNK of L
// columns (i).  Per iteration it (1) requests the 9 input vectors of column c+2 (software prefetch),
// (2) runs step 1 (STEP_FS main(), html:283-360) for column c+1 on the vectors requested one
// iteration earlier, (3) runs step 2 for column c from the post-collision populations of columns
// c-1, c, c+1, which never leave the register file: the +-1 shifts along j are lane shuffles, so the
// first and last row of a window cannot be produced and windows advance by 252 rows; (4) stores 9
// vectors.  HBM traffic per TWO site updates: 9 loads x (L+2)/L + 9 stores instead of 18 + 18.
//
// Only "plain" regions are fused: a unit (chunk x window) is fusable when no solid site lies within
// its input footprint and it stays clear of the inlet/outlet columns.  Everything else — the body
// surface, columns 0,1,NX-2,NX-1 — takes two ordinary single steps through a third lattice
// (A -> C on the tiles of T1, C -> B on the tiles of T2), see FusePlan below.  Every site is
// computed by exactly the same arithmetic as in k_step, so results stay bit-identical.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include "step_fast.hpp"

namespace wt {

static constexpr int FUSE_WIN_STRIDE = 252;     // rows a window advances by (256 loaded, 2+2 overlap)

struct FuseUnit { int ia, ib, w, pad; };        // output columns [ia, ib), window index

// value at j-1 / j+1 taken from the neighbouring lane (window-edge lanes get garbage: not stored)
template <typename T>
__device__ __forceinline__ Vec<T> from_below(const Vec<T> &r)
{
    constexpr int N = VecOf<T>::N;
    Vec<T> o;
    o.v[0] = lane_up(r.v[N - 1]);
#pragma unroll
    for (int v = 1; v < N; v++) o.v[v] = r.v[v - 1];
    return o;
}
template <typename T>
__device__ __forceinline__ Vec<T> from_above(const Vec<T> &r)
{
    constexpr int N = VecOf<T>::N;
    Vec<T> o;
#pragma unroll
    for (int v = 0; v < N - 1; v++) o.v[v] = r.v[v + 1];
    o.v[N - 1] = lane_down(r.v[0]);
    return o;
}

__device__ __forceinline__ void fuse_load_inputs(const float *__restrict__ s, const Geom &g, long P, int col, int j0, Vec<float> (&fin)[9])
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

// collide the 4 sites of a lane; top/bottom rows get the far-field populations (html:314-322)
template <bool WANT_MACRO>
__device__ __forceinline__ void fuse_collide(const Vec<float> (&fin)[9], const Geom &g, int j0, float tau, float U0,
                                             const float (&feq0)[9], Vec<float> (&G)[9], Vec<float> (&mac)[3])
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
        if (WANT_MACRO) {
            mac[0].v[v] = far ? 1.0f : rho;
            mac[1].v[v] = far ? U0 : ux;
            mac[2].v[v] = far ? 0.0f : uy;
        }
    }
}

template <bool EMIT>
__global__ __launch_bounds__(256) void k_step2(const float *__restrict__ fs, float *__restrict__ fd, float *__restrict__ macro,
                                               const FuseUnit *__restrict__ units, int nunits, Geom g, float tau, float U0, int rev)
{
    const int lane = threadIdx.x & 63;
    int u = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (u >= nunits) return;
    if (rev) u = nunits - 1 - u;
    const FuseUnit un = units[u];
    const int ia = __builtin_amdgcn_readfirstlane(un.ia), ib = __builtin_amdgcn_readfirstlane(un.ib);
    const int w = __builtin_amdgcn_readfirstlane(un.w);
    const int j0 = w * FUSE_WIN_STRIDE + lane * 4;
    const float *s = fs + g.pitch;
    float *d = fd + g.pitch;
    const long P = g.plane;
    const long mp = (long)g.nxl * g.pitch;
    float feq0[9];
    feq_all<float>(1.0f, U0, 0.0f, feq0);
    const bool first_win = (w == 0);

    Vec<float> G158m[3], G024c[3], G158c[3];      // what step 2 still needs from columns c-1 and c
    Vec<float> in[9], G[9], mac[3];
    fuse_load_inputs(s, g, P, ia - 1, j0, in);
    fuse_collide<false>(in, g, j0, tau, U0, feq0, G, mac);
    G158m[0] = G[1]; G158m[1] = G[5]; G158m[2] = G[8];
    fuse_load_inputs(s, g, P, ia, j0, in);
    fuse_collide<false>(in, g, j0, tau, U0, feq0, G, mac);
    G024c[0] = G[0]; G024c[1] = G[2]; G024c[2] = G[4];
    G158c[0] = G[1]; G158c[1] = G[5]; G158c[2] = G[8];
    fuse_load_inputs(s, g, P, ia + 1, j0, in);
#pragma unroll 1
    for (int c = ia; c < ib; c++) {
        Vec<float> nxt[9];
        fuse_load_inputs(s, g, P, (c + 2 <= ib) ? c + 2 : c + 1, j0, nxt);   // prefetch (last one: harmless re-load)
        fuse_collide<false>(in, g, j0, tau, U0, feq0, G, mac);               // step 1 of column c+1
        Vec<float> fin[9];
        fin[0] = G024c[0]; fin[1] = G158m[0]; fin[3] = G[3];
        fin[2] = from_below<float>(G024c[1]); fin[5] = from_below<float>(G158m[1]); fin[6] = from_below<float>(G[6]);
        fin[4] = from_above<float>(G024c[2]); fin[8] = from_above<float>(G158m[2]); fin[7] = from_above<float>(G[7]);
        Vec<float> out[9];
        fuse_collide<EMIT>(fin, g, j0, tau, U0, feq0, out, mac);             // step 2 of column c
        const long cc = (long)c * g.pitch + j0;
        if (j0 + 3 < g.ny) {
            if (lane == 0 && !first_win) {          // rows J0+2, J0+3 only
#pragma unroll
                for (int k = 0; k < 9; k++) *reinterpret_cast<float2 *>(d + k * P + cc + 2) = make_float2(out[k].v[2], out[k].v[3]);
                if (EMIT)
#pragma unroll
                    for (int a = 0; a < 3; a++) *reinterpret_cast<float2 *>(macro + a * mp + cc + 2) = make_float2(mac[a].v[2], mac[a].v[3]);
            } else if (lane == 63) {                // rows J0+252, J0+253 only
#pragma unroll
                for (int k = 0; k < 9; k++) *reinterpret_cast<float2 *>(d + k * P + cc) = make_float2(out[k].v[0], out[k].v[1]);
                if (EMIT)
#pragma unroll
                    for (int a = 0; a < 3; a++) *reinterpret_cast<float2 *>(macro + a * mp + cc) = make_float2(mac[a].v[0], mac[a].v[1]);
            } else {
#pragma unroll
                for (int k = 0; k < 9; k++) vstore<float>(d + k * P + cc, out[k]);
                if (EMIT)
#pragma unroll
                    for (int a = 0; a < 3; a++) vstore<float>(macro + a * mp + cc, mac[a]);
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

// Single-step kernel over a list of WINDOW-ALIGNED tiles (column x, window w): rows
// [252w, 252w+256) are computed; PASS 1 (A -> C) stores all of them, PASS 2 (C -> B) stores the
// window's own output rows [252w+2, 252w+254) (to the domain edge for the first / last window), whose
// stencil stays inside the rows pass 1 produced — no ring of extra tiles above or below the zone.
template <typename T, bool EMIT, int LOADMODE, int PASS>
__global__ __launch_bounds__(256) void k_step_list(const T *__restrict__ fs, T *__restrict__ fd, T *__restrict__ macro,
                                                   const uint8_t *__restrict__ mask, const uint8_t *__restrict__ wtiles,
                                                   int nwin, Geom g, const int *__restrict__ list, int nlist, T tau, T U0, int rev)
{
    const int lane = threadIdx.x & 63;
    int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= nlist) return;
    if (rev) t = nlist - 1 - t;
    const int tile = __builtin_amdgcn_readfirstlane(list[t]);
    const int w = tile % nwin;
    const int row0 = w * FUSE_WIN_STRIDE;
    int lo, hi;
    if (PASS == 1) { lo = row0; hi = row0 + 256; }
    else { lo = (w == 0) ? 0 : row0 + 2; hi = (w == nwin - 1) ? g.ny : row0 + 254; }
    step_tile<T, EMIT, LOADMODE, true>(fs, fd, macro, mask, wtiles, nwin, g, 0, tau, U0, tile, lane, FUSE_WIN_STRIDE, lo, hi);
}

// ------------------------------------------------------------------------------------------------
// host side: which units are fused, which tiles take the two single steps
// ------------------------------------------------------------------------------------------------
struct FusePlan {
    std::vector<FuseUnit> units;     // fusable units, chunk-major
    std::vector<int> t1, t2;         // window-tile ids (x*nwin + w) of pass 1 (A->C) and pass 2 (C->B)
    int chunk = 0, nwin = 0;
    bool usable = false;
};

// 2-D prefix sums of "column x has a solid site in rows [256t, 256t+256)" — one scan of the mask
struct FuseScan {
    int nx = 0, ny = 0, tpc = 0;
    std::vector<int> pre;
    bool any_solid(int x0, int x1, int t0, int t1) const {      // columns [x0,x1), 256-row tiles [t0,t1]
        x0 = x0 < 0 ? 0 : x0; x1 = x1 > nx ? nx : x1; t0 = t0 < 0 ? 0 : t0; t1 = t1 > tpc - 1 ? tpc - 1 : t1;
        if (x1 <= x0 || t1 < t0) return false;
        const long s = (long)pre[(size_t)x1 * (tpc + 1) + t1 + 1] - pre[(size_t)x0 * (tpc + 1) + t1 + 1] -
                       pre[(size_t)x1 * (tpc + 1) + t0] + pre[(size_t)x0 * (tpc + 1) + t0];
        return s != 0;
    }
    bool unit_fusable(int ia, int ib, int w) const {            // input footprint: columns [ia-2, ib+2), rows [J0-1, J0+257)
        const int J0 = w * FUSE_WIN_STRIDE;
        return !any_solid(ia - 2, ib + 2, (J0 - 1 < 0 ? 0 : J0 - 1) / 256, (J0 + 256) / 256);
    }
};

// mask: host [NY][NX] (non-zero = solid)
static inline FuseScan scan_mask(const uint8_t *mask, int nx, int ny)
{
    FuseScan sc;
    sc.nx = nx; sc.ny = ny; sc.tpc = (ny + 255) / 256;
    const int tpc = sc.tpc;
    sc.pre.assign((size_t)(nx + 1) * (tpc + 1), 0);
    std::vector<uint8_t> flag((size_t)nx * tpc, 0);
    for (int y = 0; y < ny; y++) {
        const uint8_t *row = mask + (size_t)y * nx;
        const int jt = y / 256;
        for (int x = 0; x < nx; x++) if (row[x]) flag[(size_t)x * tpc + jt] = 1;
    }
    for (int x = 0; x < nx; x++)
        for (int t = 0; t < tpc; t++)
            sc.pre[(size_t)(x + 1) * (tpc + 1) + t + 1] = flag[(size_t)x * tpc + t] + sc.pre[(size_t)x * (tpc + 1) + t + 1] +
                                                          sc.pre[(size_t)(x + 1) * (tpc + 1) + t] - sc.pre[(size_t)x * (tpc + 1) + t];
    return sc;
}

// Chunk length for a chip that keeps `capacity` fused waves resident (2 per SIMD).  Units are
// equal-sized and the dispatcher refills free slots, so the kernel takes about
//   units * (L+2) / capacity   (work, incl. the 2 redundant step-1 columns per chunk)
// + (L+2) / 2                  (ragged end: half a unit on average)
// column iterations; measured optimum on the 4096^2 bench body: L = 16 (tools/kfuse, bench --fuse-chunk).
static inline int auto_fuse_chunk(const FuseScan &sc, long capacity)
{
    const int ca = 2, cb = sc.nx - 2;
    const int nwin = (sc.ny - 2 + FUSE_WIN_STRIDE - 1) / FUSE_WIN_STRIDE;
    int best_L = 16;
    double best = 1e300;
    for (int L = 6; L <= 64; L++) {
        long units = 0;
        for (int ia = ca; ia < cb; ia += L)
            for (int w = 0; w < nwin; w++) units += sc.unit_fusable(ia, (ia + L < cb) ? ia + L : cb, w);
        if (units == 0) continue;
        const double cost = (double)units * (L + 2) / (double)capacity + 0.5 * (L + 2);
        if (cost < best) { best = cost; best_L = L; }
    }
    return best_L;
}

// Whole-lattice fp32 handles with NY % 4 == 0 only.
static inline FusePlan build_fuse_plan(const FuseScan &sc, int L)
{
    FusePlan p;
    p.chunk = L;
    const int nx = sc.nx, ny = sc.ny;
    if (ny % 4 != 0 || nx < 8 || ny < 8 || L < 1) return p;
    const int ca = 2, cb = nx - 2;
    const int nwin = (ny - 2 + FUSE_WIN_STRIDE - 1) / FUSE_WIN_STRIDE;
    const int nchunk = (cb - ca + L - 1) / L;
    std::vector<uint8_t> fus((size_t)nchunk * nwin, 0);
    for (int q = 0; q < nchunk; q++) {
        const int ia = ca + q * L, ib = (ia + L < cb) ? ia + L : cb;
        for (int w = 0; w < nwin; w++) {
            const bool ok = sc.unit_fusable(ia, ib, w);
            fus[(size_t)q * nwin + w] = ok;
            if (ok) p.units.push_back(FuseUnit{ia, ib, w, 0});
        }
    }
    // zone = the outputs of the non-fusable units + columns 0,1,NX-2,NX-1; pass 2 covers it with
    // window-aligned tiles (x, w), pass 1 additionally the columns left and right of it
    std::vector<uint8_t> in2((size_t)nx * nwin, 0), in1((size_t)nx * nwin, 0);
    for (int x = 0; x < nx; x++)
        for (int w = 0; w < nwin; w++) {
            bool covered = (x >= ca && x < cb);
            if (covered) covered = fus[(size_t)((x - ca) / L) * nwin + w] != 0;
            if (!covered) in2[(size_t)x * nwin + w] = 1;
        }
    for (int x = 0; x < nx; x++)
        for (int w = 0; w < nwin; w++) {
            if (!in2[(size_t)x * nwin + w]) continue;
            p.t2.push_back(x * nwin + w);
            for (int dx = -1; dx <= 1; dx++)
                if (x + dx >= 0 && x + dx < nx) in1[(size_t)(x + dx) * nwin + w] = 1;
        }
    for (int x = 0; x < nx; x++)
        for (int w = 0; w < nwin; w++)
            if (in1[(size_t)x * nwin + w]) p.t1.push_back(x * nwin + w);
    p.nwin = nwin;
    p.usable = !p.units.empty();
    return p;
}

}  // namespace wt
