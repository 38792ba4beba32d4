// step_march3.hpp — THREE or FOUR lattice steps per pass: 8-byte vectors per lane and direction —
//   <float, 2>  fp32, 2 sites per lane, 128-row windows (the fp32 default);
//   <double, 1> fp64, 1 site per lane, 64-row windows.
// Below, "128 rows" stands for WIN = 64 * S.
//
// Why: the SQ counters of the two-step kernel (profiles/r02_c_sq_counters.txt) show its vector-memory issue stalled on a
// full texture-addresser command FIFO for about as many cycles as the kernel runs, with the vector ALU 30 % busy — it is
// bound by the per-CU load/store path (~10 B/clk/CU of L1-miss traffic, the same limit k_step sits on), not by
// arithmetic.  The lever left is bytes per step: a pass that advances T steps moves 9 (L + 2(T-1))/L + 9 words
// per site instead of 18 T.  Every further level costs 21 register-resident vectors; with 8-byte vectors T = 3 takes
// 196-225 VGPRs and T = 4 216-256 (two waves per SIMD either way; 16-byte vectors do not fit beyond T = 2).
//
// Pipeline of one wave (window w, columns [ia, ib)), iteration x (DEPTH = 3; DEPTH = 4: one more level, march_unit4):
//     level 1 of column x      <- STEP_FS on the nine streamed vectors of column x        (march_align_in + march_step1)
//     level 2 of column x - 1  <- STEP_FS on level 1 of columns x-2, x-1, x               (march_stage)
//     level 3 of column x - 2  <- STEP_FS on level 2 of columns x-3, x-2, x-1 -> stored   (march_stage)
// The rows just outside the window — of the lattice itself (level 0: the populations are loaded without a row shift) and of every
// level below the last — come from the column's HALO LINE (below), built per pass by k_halo3 / k_halo4 from the seam buffer S3 the
// previous pass wrote (four rows on either side of every seam) or, when that is stale, from the lattice.
// Every site goes through the arithmetic of k_step DEPTH times: results are bit-identical to DEPTH single steps.
#pragma once
#include "step_march.hpp"

namespace wt {

static constexpr int M3_SREC = 80;          // S3 record: 2 halves x (9 slots x 4 rows + one pad slot) elements
static constexpr int M3_SHALF = 40;
// Halo LINES (round 4; one table in place of H1 / H2 / H3): HL[(b * (nxl+2) + c + 1) * 32 + 16 side + 4 q + slot], seam b = 1 .. nwin-1 between rows
// WIN b - 1 and WIN b, column c: everything the stages of the two windows next to the seam need from beyond it for column c, already gathered from
// the neighbouring columns —
//     side 0 (window b, from below):   slot 0, 1, 2 = populations 2 of column c, 5 of column c-1, 6 of column c+1 in row WIN b - 1 (they move up),
//     side 1 (window b-1, from above): slot 0, 1, 2 = populations 4 of column c, 7 of column c+1, 8 of column c-1 in row WIN b (they move down) —
// q = 0, 1, 2: after one, two, three steps (levels 1 .. depth-1); q = 3: LEVEL 0, the lattice's own rows, so that the populations themselves are
// loaded without a row shift (march_load_aligned).  One 128-byte line (fp32) per seam and column, written whole by one workgroup of the halo
// kernel; window w reads the first half of line (w, c) and the second half of line (w+1, c) with ONE load instruction (lanes 0..15 / 16..31) one
// iteration ahead of the column's first stage and hands the register from stage to stage.  Before, every iteration issued three loads (one per
// level) by lanes 0..5 that each touched three consecutive 32-byte records of two seams, and six of the nine population loads were shifted by one
// row and took a fifth line each: 46.7 line requests per column and window where 36 + 2 do (profiles/r04_u_fetch_calibration.txt).
static constexpr int M3_HL = 32;            // elements of one halo line
static constexpr int HL_COLS = 60;          // lines (columns) per workgroup of the halo kernels
static constexpr int H3_COLS = HL_COLS, H4_COLS = HL_COLS;

// ------------------------------------------------------------------------------------------------
// once per pass: the halo lines
// ------------------------------------------------------------------------------------------------
// One block = one seam x 60 columns (x0 .. x0+59), NL = depth - 1 levels:
//   level 1 of rows WIN b - NL .. WIN b + NL - 1 x (60 + 2 NL) columns from the seam buffer S3 (the record of the upstream column: rows
//           WIN b - 4 .. WIN b + 3 of the lattice this pass reads) where the site is plain interior fluid and S3 is valid, else through
//           site_step1 on the lattice -> LDS;  the same threads put the LEVEL-0 words of their column into the lines (they ARE three of the
//           site's nine pulled inputs);
//   level k of 2 (NL + 1 - k) rows x (60 + 2 (NL + 1 - k)) columns from level k-1 in LDS (halo_step: every branch of STEP_FS in the
//           reference's order);  the last level stays in registers, 62 columns x the two rows next to the seam;
//   the 60 lines are put together in LDS and stored as 60 x 128 contiguous bytes.
// (Every level is computed two columns wider than the next one needs: no word of a line comes from another block.)

// once per mask upload: flags3[(b - 1) * nxl + x], bit r4 set <=> site (x, 128 b - 2 + r4) is plain interior fluid (not solid, no
// solid neighbour, not on an inlet / outlet column or the first / last row) — one coalesced byte in place of eleven
// scattered mask / bounce-code bytes per halo thread
__global__ __launch_bounds__(256) void k_seam_flags3(const uint8_t *__restrict__ mask, const uint8_t *__restrict__ bcode, uint8_t *__restrict__ flags3,
                                                     Geom g, int nwin, int win)
{
    const long total = (long)(nwin - 1) * g.nxl;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int x = (int)(t % g.nxl);
    const int b = 1 + (int)(t / g.nxl);
    const uint8_t *m = mask + g.pitch;
    const int gi = x + g.gi0;
    unsigned f = 0;
    for (int r4 = 0; r4 < 4; r4++) {
        const int j = win * b - 2 + r4;
        if (j <= 0 || j >= g.ny - 1 || gi <= 0 || gi >= g.nx_g - 1) continue;
        const long c = (long)x * g.pitch + j;
        if (m[c] == 0 && bcode[c] == 0) f |= 1u << r4;
    }
    flags3[t] = (uint8_t)f;
}
// flags4[(b - 1) * nxl + x], bit r6 set <=> site (x, WIN b - 3 + r6) is plain interior fluid (see k_seam_flags3)
__global__ __launch_bounds__(256) void k_seam_flags4(const uint8_t *__restrict__ mask, const uint8_t *__restrict__ bcode, uint8_t *__restrict__ flags4,
                                                     Geom g, int nwin, int win)
{
    const long total = (long)(nwin - 1) * g.nxl;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int x = (int)(t % g.nxl);
    const int b = 1 + (int)(t / g.nxl);
    const uint8_t *m = mask + g.pitch;
    const int gi = x + g.gi0;
    unsigned f = 0;
    for (int r6 = 0; r6 < 6; r6++) {
        const int j = win * b - 3 + r6;
        if (j <= 0 || j >= g.ny - 1 || gi <= 0 || gi >= g.nx_g - 1) continue;
        const long c = (long)x * g.pitch + j;
        if (m[c] == 0 && bcode[c] == 0) f |= 1u << r6;
    }
    flags4[t] = (uint8_t)f;
}

// STEP_FS (html:283-360), every branch in the reference's order, for site (x, j) on the previous level's values:
// get(k, dx, dy) = population k of site (x + dx, j + dy) one level down
// `sc` = the site's bounce code (bits 0..7: bit k-1 set <=> the site upstream of direction k is solid, k_bounce_codes) | 256 if the site itself is
// solid — fetched by the caller at the start of the kernel, not behind a barrier
template <typename T, int FD, typename GET>
__device__ __forceinline__ void halo_step(unsigned sc, const Geom &g, int x, int j, bool plain, GET get, const FastDiv &fdv, T tau, T U0,
                                          T (&o)[9])
{
    if (j >= g.ny || x < 0 || x >= g.nxl) {
#pragma unroll
        for (int k = 0; k < 9; k++) o[k] = T(0);
        return;
    }
    const int gi = x + g.gi0;
    if (plain) {                                                       // plain interior fluid: html:324-359 without the mask reads
        T fin[9], rho, ux, uy;
#pragma unroll
        for (int k = 0; k < 9; k++) fin[k] = get(k, -ex_of(k), -ey_of(k));
        collide_t<T, FD>(fin, fdv, tau, o, rho, ux, uy);
    } else if (sc & 256u) {                                            // html:287-294 solid
#pragma unroll
        for (int k = 0; k < 9; k++) o[k] = get(opp_of(k), 0, 0);
    } else if (gi == g.nx_g - 1) {                                     // html:301-312 outlet
#pragma unroll
        for (int k = 0; k < 9; k++) o[k] = get(k, -1, 0);
    } else if (gi == 0 || j == g.ny - 1 || j == 0) {                   // html:314-322 far field
        feq_all<T>(T(1), U0, T(0), o);
    } else {                                                           // html:324-359 interior fluid
        T fin[9], rho, ux, uy;
        fin[0] = get(0, 0, 0);
#pragma unroll
        for (int k = 1; k < 9; k++) fin[k] = ((sc >> (k - 1)) & 1u) ? get(opp_of(k), 0, 0) : get(k, -ex_of(k), -ey_of(k));
        collide_t<T, FD>(fin, fdv, tau, o, rho, ux, uy);
    }
}

#ifdef WT_UNIT_CLOCKS
__device__ unsigned long long g_halo_clk[8];      // diagnostic build: clocks of the halo kernel's phases, summed over workgroups (+ count)
#define H4_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); hst_[i] = t_ - tprev_; tprev_ = t_; } while (0)      // (summed at the end of the kernel)
#else
#define H4_STAMP(i) do { } while (0)
#endif

// site_step1 (step_march.hpp) for a site next to a seam whose inputs are all in the seam buffer: `a` = the nine pulled populations (already
// fetched), `rec` = the record of column x, q0 = the site's row relative to row WIN b - 4.  A body / inlet / outlet site then costs nine more
// dwords of records that are in the cache anyway instead of nine scattered lines of the lattice (a quarter of the halo kernel's HBM reads on the
// bench lattice came from the few per cent of its sites that are not plain fluid).  Same branches, same order, same values (html:283-360).
template <typename T, int FD>
__device__ __forceinline__ void site_step1_seams(const T *__restrict__ rec, const T (&a)[9], int q0, unsigned sc, const Geom &g, int x, int j,
                                                 const FastDiv &fdv, T tau, T U0, T (&out)[9])
{
    const int gi = x + g.gi0;
    const int e0 = (q0 >> 2) * M3_SHALF + (q0 & 3);                // own-site element of direction k: e0 + 4 k
    if (sc & 256u) {                                               // html:287-294 solid
#pragma unroll
        for (int k = 0; k < 9; k++) out[k] = rec[e0 + 4 * opp_of(k)];
    } else if (gi == g.nx_g - 1) {                                 // html:301-312 outlet
#pragma unroll
        for (int k = 0; k < 9; k++) out[k] = rec[-M3_SREC + e0 + 4 * k];
    } else if (gi == 0 || j == g.ny - 1 || j == 0) {               // html:314-322 far field
        feq_all<T>(T(1), U0, T(0), out);
    } else {                                                       // html:324-359 interior fluid
        T fin[9], rho, ux, uy;
        fin[0] = a[0];
#pragma unroll
        for (int k = 1; k < 9; k++) fin[k] = ((sc >> (k - 1)) & 1u) ? rec[e0 + 4 * opp_of(k)] : a[k];
        collide_t<T, FD>(fin, fdv, tau, out, rho, ux, uy);
    }
}

template <typename T, int S, int FD, int DEPTH>
__device__ __forceinline__ void halo_lines_body(const T *__restrict__ fs, const T *__restrict__ seams3, const uint8_t *__restrict__ mask,
                                                const uint8_t *__restrict__ bcode, const uint8_t *__restrict__ flags, T *__restrict__ hl, const Geom &g, int nwin, int use_seams,
                                                const FastDiv &fdv, T tau, T U0, int xb0, int nbx)
{
    constexpr int WIN = 64 * S, NL = DEPTH - 1;
    constexpr int C1 = HL_COLS + 2 * NL, R1 = 2 * NL;                  // level 1: columns x0 - NL + c1, rows WIN b - NL + r
    constexpr int C2 = HL_COLS + 2 * (NL - 1), R2 = 2 * (NL - 1);      // level 2 (in LDS for DEPTH 4 only): columns x0 - (NL - 1) + c2
    constexpr int PADK = sizeof(T) == 4 ? 10 : 9;      // fp32: + 1 spreads the columns over the LDS banks; fp64 must stay below 64 KB
    __shared__ __attribute__((aligned(32))) T l1[C1][R1][PADK];
    __shared__ T l2[DEPTH == 4 ? C2 : 1][DEPTH == 4 ? R2 : 1][9];
    __shared__ __attribute__((aligned(32))) T lv0[HL_COLS][2][4];      // level-0 words until the lines are put together
    // the 60 lines are put together in l1's memory once nobody reads it any more (behind the third barrier): fp64 stays at three workgroups per CU
    static_assert(sizeof(T) * C1 * R1 * PADK >= sizeof(T) * HL_COLS * M3_HL, "the lines fit into l1");
    T (*ol)[M3_HL] = reinterpret_cast<T (*)[M3_HL]>(&l1[0][0][0]);
    // the launch covers the column blocks xb0 .. xb0 + nbx - 1 of every seam (a whole pass: all of them; the edge strips of a slab's fused
    // renewal: the blocks that hold its refreshed ghost columns, windtunnel.hip renew_fused)
    const int b = 1 + (int)(blockIdx.x / nbx);
    const int x0 = (xb0 + (int)(blockIdx.x % nbx)) * HL_COLS;
    const uint8_t *m = mask + g.pitch;
    const T *s = fs + g.pitch;                                          // column 0 of the lattice
    typedef T t4 __attribute__((ext_vector_type(4)));
#ifdef WT_UNIT_CLOCKS
    unsigned long long hst_[7] = {0, 0, 0, 0, 0, 0, 0}, tprev_ = __builtin_amdgcn_s_memtime();
#endif
    // the plain-fluid flags of the columns this thread works on in the LATER phases are requested now, beside level 1's inputs: fetched where they
    // are used they put a trip to memory behind every barrier of a kernel that is a chain of latencies
    // ... and so are the solid flag and the bounce code of the thread's sites (`sc`, halo_step)
    unsigned fl2 = 0, fl3 = 0, sc2 = 0, sc3 = 0;
    auto site_code = [&](int x, int j) -> unsigned {
        const long c = (long)x * g.pitch + j;
        return (unsigned)bcode[c] | (m[c] ? 256u : 0u);
    };
    if constexpr (DEPTH == 4) {
        const int x = x0 - (NL - 1) + (int)threadIdx.x / R2, j = WIN * b - (NL - 1) + (int)threadIdx.x % R2;
        if (x >= 0 && x < g.nxl) {
            fl2 = flags[(long)(b - 1) * g.nxl + x];
            if (j < g.ny && !((fl2 >> ((int)threadIdx.x % R2 + 1)) & 1)) sc2 = site_code(x, j);       // (plain sites — nearly all — need none)
        }
    }
    if (threadIdx.x < 2 * (HL_COLS + 2)) {
        const int x = x0 - 1 + (int)(threadIdx.x >> 1), j = WIN * b - 1 + (int)(threadIdx.x & 1);
        if (x >= 0 && x < g.nxl) {
            fl3 = flags[(long)(b - 1) * g.nxl + x];
            if (j < g.ny && !((fl3 >> (NL - 1 + (int)(threadIdx.x & 1))) & 1)) sc3 = site_code(x, j);
        }
    }
    // ---- level 1 (+ the lines' level-0 words)
    for (int w = threadIdx.x; w < C1 * R1; w += 256) {
        const int cl = w / R1, r = w % R1;
        const int x = x0 - NL + cl;
        const int j = WIN * b - NL + r;
        T o[9];
#pragma unroll
        for (int k = 0; k < 9; k++) o[k] = T(0);
        if (x >= 0 && x < g.nxl && j < g.ny) {
            // the nine seam-buffer inputs are requested together with the flag (not behind it): this kernel is a chain of memory
            // latencies, and the records of columns x-1 .. x+1 exist for every x (pad records at both ends)
            T a[9];
            const T *rec = seams3 + ((long)b * (g.nxl + 2) + x + 1) * M3_SREC;
#pragma unroll
            for (int k = 0; k < 9; k++) {
                const int q = 4 - NL + r - ey_of(k);             // row j - ey_k relative to row WIN b - 4
                a[k] = rec[-(long)ex_of(k) * M3_SREC + (q >> 2) * M3_SHALF + 4 * k + (q & 3)];
            }
            const bool plain = use_seams && ((flags[(long)(b - 1) * g.nxl + x] >> r) & 1) != 0;
            const int o0 = cl - NL;                              // the line of this column
            if ((r == NL || r == NL - 1) && o0 >= 0 && o0 < HL_COLS) {
                // level 0: row WIN b (r = NL) pulls populations 2,5,6 out of row WIN b - 1 — window b's words from below; row WIN b - 1 pulls 4,7,8
                // out of row WIN b — window b-1's words from above.  From the lattice when the seam buffer is stale (first pass after an upload).
                const bool below = r == NL;
                T v0 = below ? a[2] : a[4], v1 = below ? a[5] : a[7], v2 = below ? a[6] : a[8];
                if (!use_seams) {
                    const long c = (long)x * g.pitch + j;
                    if (below) { v0 = s[2 * g.plane + c - 1]; v1 = s[5 * g.plane + c - g.pitch - 1]; v2 = s[6 * g.plane + c + g.pitch - 1]; }
                    else { v0 = s[4 * g.plane + c + 1]; v1 = s[7 * g.plane + c + g.pitch + 1]; v2 = s[8 * g.plane + c - g.pitch + 1]; }
                }
                *reinterpret_cast<t4 *>(&lv0[o0][below ? 0 : 1][0]) = t4{v0, v1, v2, T(0)};
            }
            if (plain) {
                T rho, ux, uy;
                collide_t<T, FD>(a, fdv, tau, o, rho, ux, uy);
            } else if (use_seams) {
                site_step1_seams<T, FD>(rec, a, 4 - NL + r, site_code(x, j), g, x, j, fdv, tau, U0, o);
            } else {
                site_step1<T, FD>(s, m, g, x, j, fdv, tau, U0, o);
            }
        } else {
            const int o0 = cl - NL;
            if ((r == NL || r == NL - 1) && o0 >= 0 && o0 < HL_COLS) *reinterpret_cast<t4 *>(&lv0[o0][r == NL ? 0 : 1][0]) = t4{T(0), T(0), T(0), T(0)};
        }
#pragma unroll
        for (int k = 0; k < 9; k++) l1[cl][r][k] = o[k];
    }
    H4_STAMP(0);
    __syncthreads();
    H4_STAMP(1);
    if constexpr (DEPTH == 4) {
        // ---- level 2: columns x0 - 2 + c2, rows WIN b - 2 + r (exactly one item per thread)
        static_assert(DEPTH != 4 || C2 * R2 == 256, "one level-2 item per thread");
        for (int w = threadIdx.x; w < C2 * R2; w += 256) {
            const int c2 = w / R2, r = w % R2;
            const int x = x0 - (NL - 1) + c2;
            const int j = WIN * b - (NL - 1) + r;
            const bool plain = ((fl2 >> (r + 1)) & 1) != 0;              // (C2 R2 = 256: this thread's one item is the column fl2 was fetched for)
            auto get = [&](int k, int dx, int dy) { return l1[c2 + 1 + dx][r + 1 + dy][k]; };
            T o[9];
            halo_step<T, FD>(sc2, g, x, j, plain, get, fdv, tau, U0, o);
#pragma unroll
            for (int k = 0; k < 9; k++) l2[c2][r][k] = o[k];
        }
        H4_STAMP(2);
        __syncthreads();
        H4_STAMP(3);
    }
    // ---- the last level (threads 0 .. 123: columns x0 - 1 + c, the two rows next to the seam) beside the lines' words of the levels below it
    //      (threads 128 .. 247: line o, side): into registers first — the lines take l1's place
    T wl[3] = {T(0), T(0), T(0)};                 // last level: own word, the one moving in +x, the one moving in -x
    t4 wq[3];                                     // assembling threads: the words of level 1, (level 2,) level 0
    const bool last_thr = threadIdx.x < 2 * (HL_COLS + 2), asm_thr = threadIdx.x >= 128 && threadIdx.x < 128 + 2 * HL_COLS;
    if (last_thr) {
        const int c = threadIdx.x >> 1, side = threadIdx.x & 1;
        const int x = x0 - 1 + c;
        const int j = WIN * b - 1 + side;
        const bool plain = ((fl3 >> (NL - 1 + side)) & 1) != 0;
        T o[9];
        if constexpr (DEPTH == 4) {
            auto get = [&](int k, int dx, int dy) { return l2[c + 1 + dx][1 + side + dy][k]; };
            halo_step<T, FD>(sc3, g, x, j, plain, get, fdv, tau, U0, o);
        } else {
            auto get = [&](int k, int dx, int dy) { return l1[c + 1 + dx][1 + side + dy][k]; };
            halo_step<T, FD>(sc3, g, x, j, plain, get, fdv, tau, U0, o);
        }
        wl[0] = side ? o[4] : o[2]; wl[1] = side ? o[8] : o[5]; wl[2] = side ? o[7] : o[6];
    } else if (asm_thr) {
        const int t = threadIdx.x - 128, o = t >> 1, side = t & 1;
        auto words = [&](const auto &lk, int c, int r) {
            return side == 0 ? t4{lk[c][r][2], lk[c - 1][r][5], lk[c + 1][r][6], T(0)} : t4{lk[c][r][4], lk[c + 1][r][7], lk[c - 1][r][8], T(0)};
        };
        wq[0] = words(l1, o + NL, NL - 1 + side);
        if constexpr (DEPTH == 4) wq[1] = words(l2, o + NL - 1, NL - 2 + side);
        wq[2] = *reinterpret_cast<const t4 *>(&lv0[o][side][0]);
    }
    H4_STAMP(4);
    __syncthreads();
    if (last_thr) {
        // own word -> line c-1; the one moving in +x is pulled by column x+1 (line c), the one moving in -x by column x-1 (line c-2)
        const int c = threadIdx.x >> 1, side = threadIdx.x & 1;
        const int base = 16 * side + 4 * (NL - 1);
        if (c >= 1 && c <= HL_COLS) { ol[c - 1][base] = wl[0]; ol[c - 1][base + 3] = T(0); }
        if (c < HL_COLS) ol[c][base + (side ? 2 : 1)] = wl[1];
        if (c >= 2) ol[c - 2][base + (side ? 1 : 2)] = wl[2];
    } else if (asm_thr) {
        const int t = threadIdx.x - 128, o = t >> 1, side = t & 1;
        *reinterpret_cast<t4 *>(&ol[o][16 * side]) = wq[0];
        if constexpr (DEPTH == 4) *reinterpret_cast<t4 *>(&ol[o][16 * side + 4]) = wq[1];
        *reinterpret_cast<t4 *>(&ol[o][16 * side + 12]) = wq[2];
    }
    __syncthreads();
    // ---- the block's lines: contiguous in memory
    {
        const int nlines = g.nxl - x0 < HL_COLS ? g.nxl - x0 : HL_COLS;
        t4 *dst = reinterpret_cast<t4 *>(hl + ((long)b * (g.nxl + 2) + x0 + 1) * M3_HL);
        const t4 *src = reinterpret_cast<const t4 *>(&ol[0][0]);
        for (int t = threadIdx.x; t < nlines * (M3_HL / 4); t += 256) dst[t] = src[t];
    }
    H4_STAMP(5);
#ifdef WT_UNIT_CLOCKS
    if (threadIdx.x == 0) {
        for (int i = 0; i < 7; i++) atomicAdd(&g_halo_clk[i], hst_[i]);
        atomicAdd(&g_halo_clk[7], 1ULL);
    }
#endif
}

template <typename T, int S, int FD>
__global__ __launch_bounds__(256) void k_halo3(const T *__restrict__ fs, const T *__restrict__ seams3, const uint8_t *__restrict__ mask,
                                               const uint8_t *__restrict__ bcode, const uint8_t *__restrict__ flags3, T *__restrict__ hl, Geom g, int nwin, int use_seams,
                                               FastDiv fdv, T tau, T U0, int xb0, int nbx)
{
    halo_lines_body<T, S, FD, 3>(fs, seams3, mask, bcode, flags3, hl, g, nwin, use_seams, fdv, tau, U0, xb0, nbx);
}
template <typename T, int S, int FD>
__global__ __launch_bounds__(256) void k_halo4(const T *__restrict__ fs, const T *__restrict__ seams3, const uint8_t *__restrict__ mask,
                                               const uint8_t *__restrict__ bcode, const uint8_t *__restrict__ flags4, T *__restrict__ hl, Geom g, int nwin, int use_seams,
                                               FastDiv fdv, T tau, T U0, int xb0, int nbx)
{
    halo_lines_body<T, S, FD, 4>(fs, seams3, mask, bcode, flags4, hl, g, nwin, use_seams, fdv, tau, U0, xb0, nbx);
}

// ------------------------------------------------------------------------------------------------
// the marching kernel
// ------------------------------------------------------------------------------------------------

#ifdef WT_M3_STAMPS          // diagnostic build (tools/m3_stamps.py): where does an iteration of the lean three-step loop spend its clocks?
__device__ unsigned long long g_m3_stamps[8];
#define M3_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_[i] += t_ - tprev_; tprev_ = t_; } while (0)
#else
#define M3_STAMP(i) do { } while (0)
#endif

template <typename T, int S>
struct March3Addr {
    MarchAddr<T, S> a;                   // lattice / macro descriptors and offsets (its seam fields are unused here)
    __amdgpu_buffer_rsrc_t rs3;          // seam buffer S3
    unsigned voff_lo, voff_hi;           // lanes 0 .. 10 sizeof(T)/4 - 1: byte offsets of their 16-byte chunk in the two half records this window writes
    T *lds_w;                            // where this lane stages its S rows of direction 0 (direction k: + 4 k elements)
    const char *lds_r;                   // this lane's 16-byte chunk of the staged `below` half (`above`: + 40 elements)
};

// one more application of STEP_FS in registers: level k+1 of column c from level k of columns c-1 (populations 1,5,8: m158),
// c (all nine: Gc) and c+1 (3,6,7 of Gn); `hv`, `lb`: this column's halo line (lanes 0..23) and the level's first lane
template <bool BODY, bool WANT_MACRO, int FD, int LB, typename T, int S>
__device__ __forceinline__ void march_stage(const MarchParams<T> &p, int c, int j0, int lane, bool far_win, bool nf, bool allsolid,
                                            const T (&feq0)[9], const MV<T, S> (&m158)[3], const MV<T, S> (&Gc)[9], const MV<T, S> (&Gn)[9], T hv,
                                            MV<T, S> (&out)[9], MV<T, S> (&mac)[3], const uint32_t *pre = nullptr)
{
    typedef MV<T, S> V3;
    const Geom &g = p.g;
    constexpr bool OVL = (FD & MARCH_FD_OVL) != 0;
    // `hv` = this lane's element of column c's halo words, LB = 4 (level - 1): the lane of the level's first from-below word (from above: 48 + LB)
    V3 fin[9];
    fin[0] = Gc[0]; fin[1] = m158[0]; fin[3] = Gn[3];
    fin[2] = m_below_x<LB, OVL>(Gc[2], hv); fin[5] = m_below_x<LB + 1, OVL>(m158[1], hv); fin[6] = m_below_x<LB + 2, OVL>(Gn[6], hv);
    fin[4] = m_above_x<LB, OVL>(Gc[4], hv); fin[8] = m_above_x<LB + 2, OVL>(m158[2], hv); fin[7] = m_above_x<LB + 1, OVL>(Gn[7], hv);
    if (BODY) {
        const int gi = c + g.gi0;
        if (__builtin_expect(gi <= 0 || nf, 0)) {
            auto ownc = [&](int k) { return Gc[k]; };
            uint32_t solid4 = 0, code4 = 0;
            if (nf) {
                if (pre) { solid4 = pre[0]; code4 = pre[1]; }
                else {
                    solid4 = load_site_bytes<S>(p.mask + (long)(c + 1) * g.pitch + site_row<S, OVL>(j0, g.ny));
                    code4 = load_site_bytes<S>(p.bcode + (long)c * g.pitch + site_row<S, OVL>(j0, g.ny));
                }
            }
            const bool any_solid = __ballot(solid4 != 0) != 0ULL;
            if (gi <= 0) {
#pragma unroll
                for (int k = 0; k < 9; k++) out[k] = mv_splat<T, S>(feq0[k]);
                if (WANT_MACRO) { mac[0] = mv_splat<T, S>(T(1)); mac[1] = mv_splat<T, S>(p.U0); mac[2] = mv_splat<T, S>(T(0)); }
            } else if (allsolid) {
#pragma unroll
                for (int k = 0; k < 9; k++) out[k] = Gc[k];        // every site is overwritten by march_solid below
                if (WANT_MACRO) { mac[0] = mv_splat<T, S>(T(1)); mac[1] = mv_splat<T, S>(T(0)); mac[2] = mv_splat<T, S>(T(0)); }
            } else {
                march_bounce<T, S>(fin, code4, ownc);
                march_collide_general<T, S, FD, WANT_MACRO>(fin, solid4, j0, g.ny, p.fdv, p.tau, p.U0, feq0, out, mac);
            }
            if (any_solid && (gi <= 0 || allsolid)) march_solid<T, S, WANT_MACRO>(out, mac, solid4, ownc);
            return;
        }
    }
    march_collide<T, S, FD, WANT_MACRO>(fin, p.fdv, p.tau, out, mac);
    if (far_win) march_far_rows<T, S, WANT_MACRO>(j0, g.ny, p.U0, feq0, out, mac);
}

// nine lattice stores (+ three macro stores) of column `col`, then STAGE the four rows on either side of the window's
// seams in LDS (the first 4/S lanes hold rows 0..3, the last 4/S lanes the last four rows; every lane writes its 8 bytes
// — the others into a scratch area)
// (`voff_st`: the lane's store offset, or an out-of-range one to drop the column's stores without a branch)
template <bool EMIT, bool OVL = false, typename T, int S>
__device__ __forceinline__ void march3_store(const March3Addr<T, S> &m, unsigned voff_st, int col, const MV<T, S> (&out)[9], const MV<T, S> (&mac)[3])
{
    const MarchAddr<T, S> &a = m.a;
#pragma unroll
    for (int k = 0; k < 9; k++) (void)bstore<T, S>(a.rd, voff_st, lat_off(a, k, col, 0), out[k]);
    if (EMIT) {
        const unsigned mo = (unsigned)col * a.pitch4;
#pragma unroll
        for (int q = 0; q < 3; q++) (void)bstore<T, S>(a.rm, voff_st, (unsigned)q * a.mp4 + mo, mac[q]);
    }
    if constexpr (!OVL) {            // (overlapping windows keep no seam buffer)
#pragma unroll
        for (int k = 0; k < 9; k++) {
            u2v x;
            __builtin_memcpy(&x, &out[k], 8);
            *reinterpret_cast<u2v *>(m.lds_w + 4 * k) = x;
        }
    }
}
// Make the wait for the prefetched column land HERE (an empty asm that reads its 18 registers), before this iteration's
// stores are issued.  Left to itself hipcc waits for them at the top of the next iteration with vmcnt(0) — the loop-entry path
// has nothing younger in flight, and the merged counter state keeps that — which also waits for the eleven stores just
// issued: every iteration then ends by draining its own stores.
template <typename T, int S>
__device__ __forceinline__ void wait_for_column(const MV<T, S> (&c)[9], T h1)
{
#ifdef WT_M3_NOWAIT          // experiments: leave the waits to hipcc
    return;
#endif
    u2v r[9];                // the raw 8 bytes of every vector: the same registers, whatever T and S are
#pragma unroll
    for (int k = 0; k < 9; k++) __builtin_memcpy(&r[k], &c[k], 8);
    asm volatile("" ::"v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]), "v"(r[8]), "v"(h1));
}
template <typename T, int S>
__device__ __forceinline__ void wait_for_column(const MV<T, S> (&c)[9], T h1, T h2)
{
    wait_for_column(c, h1);
#ifndef WT_M3_NOWAIT
    asm volatile("" ::"v"(h2));
#endif
}
template <typename T, int S>
__device__ __forceinline__ void wait_for_column(const MV<T, S> (&c)[9], T h1, T h2, T h3)
{
    wait_for_column(c, h1);
#ifndef WT_M3_NOWAIT
    asm volatile("" ::"v"(h2), "v"(h3));
#endif
}
__device__ __forceinline__ void wait_for_bytes(const SiteBytes &b)
{
#ifndef WT_M3_NOWAIT
    asm volatile("" ::"v"(b.v[0]), "v"(b.v[1]));
#endif
}
// ... and keep it from drifting upwards: an asm that reads the column about to be stored is ordered before the one above
template <typename T, int S>
__device__ __forceinline__ void pin_after(const MV<T, S> (&c)[9])
{
    u2v r[9];
#pragma unroll
    for (int k = 0; k < 9; k++) __builtin_memcpy(&r[k], &c[k], 8);
    asm volatile("" ::"v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]), "v"(r[8]));
}
struct Seam3 { u4v below, above; };      // one 16-byte chunk of each staged half
template <bool OVL = false, typename T, int S>
__device__ __forceinline__ Seam3 seam3_fetch(const March3Addr<T, S> &m)
{
    Seam3 r;
    if constexpr (OVL) { r.below = u4v{0, 0, 0, 0}; r.above = r.below; return r; }
    r.below = *reinterpret_cast<const u4v *>(m.lds_r);                                      // the window's last four rows
    r.above = *reinterpret_cast<const u4v *>(m.lds_r + M3_SHALF * sizeof(T));              // rows 0..3
    return r;
}
template <bool OVL = false, typename T, int S>
__device__ __forceinline__ void seam3_flush(const March3Addr<T, S> &m, int col, const Seam3 &r)
{
    if constexpr (OVL) return;
    const unsigned so = (unsigned)(col + 1) * (unsigned)(M3_SREC * sizeof(T));
    const u4v d0 = r.below, d1 = r.above;
    __builtin_amdgcn_raw_buffer_store_b128(d0, m.rs3, m.voff_hi, so, 0);        // -> seam w+1, half 0
    __builtin_amdgcn_raw_buffer_store_b128(d1, m.rs3, m.voff_lo, so, 0);        // -> seam w,   half 1
    store_data_fence2(d0, d1);
}

template <bool BODY, bool EMIT, int FD, typename T, int S>
__device__ __forceinline__ void march_unit3(const MarchParams<T> &p, March3Addr<T, S> &m, __amdgpu_buffer_rsrc_t rh, unsigned hoff,
                                            int ia, int ib, int uflags, int j0, int lane, bool far_win, ClassMask nonfast_m,
                                            ClassMask solid_m, const T (&feq0)[9])
{
    constexpr bool OVLF = (FD & MARCH_FD_OVL) != 0;      // overlapping windows: no halo lines, no seam rows (step_chain.hpp k_march3)
    typedef MV<T, S> V3;
    constexpr unsigned HREC = M3_HL * sizeof(T);     // bytes of one halo line
    const Geom &g = p.g;
    const MarchAddr<T, S> &a = m.a;
#define NONFAST(x) (BODY && cm_bit(nonfast_m, (x) - ia + 2))
#define ALLSOLID(x) (BODY && cm_bit(solid_m, (x) - ia + 2))
    // a general column's own populations come out of this wave's LDS buffer (x & 1) (step_march.hpp own_prefetch): requested one iteration ahead in
    // the loop, on the spot (OWN_NOW) for the unit's first columns
#define OWN_BUF(x) (a.own_lds + ((x) & 1) * OWN_LDS_BYTES)
#define OWN_NOW(x) do { if (BODY && NONFAST(x)) own_prefetch<T, S, OVLF>(a, (x), OWN_BUF(x)); } while (0)
#define STEP1(x, in, G) march_step1<BODY, FD, T, S, BODY>(p, a, (x), j0, far_win, NONFAST(x), ALLSOLID(x), feq0, in, G, nullptr, OWN_BUF(x))
#define STEP1P(x, in, G, sb) march_step1<BODY, FD, T, S, BODY>(p, a, (x), j0, far_win, NONFAST(x), ALLSOLID(x), feq0, in, G, BODY ? (sb).v : nullptr, OWN_BUF(x))
    const bool outlet = BODY && (uflags & MU_OUTLET_AFTER) != 0;
    const int xend = outlet ? ib : ib + 1;       // last column whose level 1 is computed (the outlet column itself for the last unit)
    V3 s1m[3], s1c[9];           // level 1: populations 1,5,8 of column x-2; all nine of column x-1
    V3 s2m[3], s2c[9];           // level 2: populations 1,5,8 of column x-3; all nine of column x-2
    V3 in[9], G1[9], G2[9], mac[3];
    // ---- prologue: level 1 of columns ia-2 and ia-1 (columns left of the inlet do not exist: the inlet column's far-field
    //      value stands in — never used, the inlet column is a constant at every level)
#pragma unroll
    for (int k = 0; k < 9; k++) { s1c[k] = mv_splat<T, S>(feq0[k]); s2c[k] = s1c[k]; }
    s1m[0] = s1c[1]; s1m[1] = s1c[5]; s1m[2] = s1c[8];
    s2m[0] = s1c[1]; s2m[1] = s1c[5]; s2m[2] = s1c[8];
    // halo lines of the columns the first iteration's stages work on (afterwards: one line per iteration, fetched one iteration ahead like the
    // populations, and handed from stage to stage)
    T hv2 = halo_load_x<OVLF, T>(rh, hoff, (unsigned)(ia - 2 > 0 ? ia - 2 : 0) * HREC);          // column x-2
    T hv1 = halo_load_x<OVLF, T>(rh, hoff, (unsigned)(ia - 1 > 0 ? ia - 1 : 0) * HREC);          // column x-1
    T hv0 = halo_load_x<OVLF, T>(rh, hoff, (unsigned)ia * HREC);                                 // column x
    if (!BODY || ia - 2 + g.gi0 >= 0) {
        OWN_NOW(ia - 2);
        march_load_aligned(a, ia - 2, in);
        march_align_in<OVLF>(in, lane, hv2);
        STEP1(ia - 2, in, s1c);
    }
    s1m[0] = s1c[1]; s1m[1] = s1c[5]; s1m[2] = s1c[8];
    if (!BODY || ia - 1 + g.gi0 >= 0) {
        OWN_NOW(ia - 1);
        march_load_aligned(a, ia - 1, in);
        march_align_in<OVLF>(in, lane, hv1);
        STEP1(ia - 1, in, s1c);
    }
    OWN_NOW(ia);
    march_load_aligned(a, ia, in);
    // site bytes of columns x, x-1, x-2 (general loop only; see SiteBytes)
    SiteBytes sb0{{0, 0}}, sb1{{0, 0}}, sb2{{0, 0}};
    if (BODY) { sb0 = site_bytes_load<T, S, OVLF>(p, ia, j0); sb1 = site_bytes_load<T, S, OVLF>(p, ia - 1, j0); sb2 = site_bytes_load<T, S, OVLF>(p, ia - 2, j0); wait_for_bytes(sb0); wait_for_bytes(sb1); wait_for_bytes(sb2); }
    wait_for_column(in, hv0, hv1, hv2);      // no load pending at the loop header: see wait_for_column
    int seam_col = -1;           // column whose seam rows are staged in LDS (-1: none yet; the flush then lands on the pad record)
#ifdef WT_M3_STAMPS
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev_ = __builtin_amdgcn_s_memtime();
#endif
    // The loop body has no branch on the pipeline fill: during the first two iterations (x - 2 < ia) level 3 is computed on
    // don't-care values and its stores are dropped by an out-of-range offset — a scalar `if` around the stage and its
    // stores makes hipcc's waitcnt pass drain vmcnt(0) at the merge points of every iteration.
#pragma unroll 1
    for (int x = ia; x <= xend; x++) {
        V3 nxt[9];
        // a general column next: its own populations on their way into LDS (own_prefetch), AHEAD of this iteration's prefetch — loads return in
        // order, so the wait for the prefetched column at the end of the iteration covers them (behind stage 1 instead: the plain slabs ran 4 % slower)
        if (BODY && x + 1 <= xend && NONFAST(x + 1)) own_prefetch<T, S, OVLF>(a, x + 1, OWN_BUF(x + 1));
        march_load_aligned(a, (x + 1 <= xend) ? x + 1 : x, nxt);              // prefetch (last one: harmless re-load)
        const bool has2 = x - 2 >= ia;                                         // column x-2 is an output column
        const int c1 = x - 1, c2 = x - 2;
        const T hvn = halo_load_x<OVLF, T>(rh, hoff, (unsigned)(x + 1) * HREC);        // column x+1's halo line: the next iteration's first stage
        SiteBytes sbn{{0, 0}};
        if (BODY) sbn = site_bytes_load<T, S, OVLF>(p, x + 1, j0);
        const Seam3 sp = seam3_fetch<OVLF>(m);                                       // staged by the previous iteration's store
        M3_STAMP(0);                                                           // issue of the prefetch
        march_align_in<OVLF>(in, lane, hv0);
        STEP1P(x, in, G1, sb0);                                                // level 1 of column x
        M3_STAMP(1);
        // level 2 of column x-1 (a column left of the inlet takes the inlet branch: constants, no memory access)
        march_stage<BODY, false, FD, 0>(p, c1, j0, lane, far_win, NONFAST(c1), ALLSOLID(c1), feq0, s1m, s1c, G1, hv1, G2, mac, BODY ? sb1.v : nullptr);
#ifdef WT_M3_STAMPS
        pin_after(G2);
#endif
        M3_STAMP(2);
        V3 out[9];
        march_stage<BODY, EMIT, FD, 4>(p, c2, j0, lane, far_win, NONFAST(c2), ALLSOLID(c2), feq0, s2m, s2c, G2, hv2, out, mac, BODY ? sb2.v : nullptr);
        pin_after(out);
        M3_STAMP(3);
        wait_for_column(nxt, hvn);
        if (BODY) { wait_for_bytes(sbn); sb2 = sb1; sb1 = sb0; sb0 = sbn; }
#ifdef WT_M3_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        M3_STAMP(4);                                                           // the wait for the prefetched column
        hv2 = hv1; hv1 = hv0; hv0 = hvn;
        march3_store<EMIT, OVLF>(m, has2 ? a.voff_st : p.lat_bytes, has2 ? c2 : 0, out, mac);
        seam3_flush<OVLF>(m, seam_col, sp);
        M3_STAMP(5);                                                           // issue of the stores
        seam_col = has2 ? c2 : seam_col;
        if (BODY && outlet && x == xend) break;                                // the tail below needs the unshifted state
        s2m[0] = s2c[1]; s2m[1] = s2c[5]; s2m[2] = s2c[8];
        s1m[0] = s1c[1]; s1m[1] = s1c[5]; s1m[2] = s1c[8];
#pragma unroll
        for (int k = 0; k < 9; k++) { s2c[k] = G2[k]; s1c[k] = G1[k]; in[k] = nxt[k]; }
    }
#ifdef WT_M3_STAMPS
    if (!BODY && lane == 0) {
        for (int i = 0; i < 6; i++) atomicAdd(&g_m3_stamps[i], st_[i]);
        atomicAdd(&g_m3_stamps[6], (unsigned long long)(xend - ia + 1));
        atomicAdd(&g_m3_stamps[7], 1ULL);
    }
#endif
    if (BODY && outlet) {
        // Here x = ib = NX-1 (local): s1c = level 1 of NX-2, G1 = level 1 of NX-1, s2c = level 2 of NX-3, G2 = level 2 of NX-2;
        // level 3 of NX-3 is stored.  The outlet column copies the previous level of column NX-2 (html:301-312):
        //   level 2 of NX-1 = level 1 of NX-2;  level 3 of NX-1 = level 2 of NX-2;  solid sites: own previous level reversed.
        const int co = ib;                                   // outlet column
        uint32_t solid4 = 0;
        if (NONFAST(co)) solid4 = load_site_bytes<S>(p.mask + (long)(co + 1) * g.pitch + site_row<S, OVLF>(j0, g.ny));
        const bool any_solid = __ballot(solid4 != 0) != 0ULL;
        V3 L2o[9], out[9];
#pragma unroll
        for (int k = 0; k < 9; k++) L2o[k] = s1c[k];
        if (any_solid) { auto own1 = [&](int k) { return G1[k]; }; march_solid<T, S, false>(L2o, mac, solid4, own1); }
        // level 3 of column NX-2
        V3 t2m[3];
        t2m[0] = s2c[1]; t2m[1] = s2c[5]; t2m[2] = s2c[8];
        const T hv2t = halo_load_x<OVLF, T>(rh, hoff, (unsigned)(co - 1) * HREC);
        Seam3 sp = seam3_fetch<OVLF>(m);
        march_stage<BODY, EMIT, FD, 4>(p, co - 1, j0, lane, far_win, NONFAST(co - 1), ALLSOLID(co - 1), feq0, t2m, G2, L2o, hv2t, out, mac);
        march3_store<EMIT, OVLF>(m, a.voff_st, co - 1, out, mac);
        seam3_flush<OVLF>(m, seam_col, sp);
        seam_col = co - 1;
        // level 3 of the outlet column
#pragma unroll
        for (int k = 0; k < 9; k++) out[k] = G2[k];
        if (EMIT) march_outlet_macro(G2, mac);
        if (any_solid) { auto own2 = [&](int k) { return L2o[k]; }; march_solid<T, S, EMIT>(out, mac, solid4, own2); }
        sp = seam3_fetch<OVLF>(m);
        march3_store<EMIT, OVLF>(m, a.voff_st, co, out, mac);
        seam3_flush<OVLF>(m, seam_col, sp);
        seam_col = co;
    }
    seam3_flush<OVLF>(m, seam_col, seam3_fetch<OVLF>(m));
#undef NONFAST
#undef ALLSOLID
#undef STEP1
#undef STEP1P
#undef OWN_BUF
#undef OWN_NOW
}

// FOUR steps per pass: one more level (s3m / s3c) and one more stage than march_unit3; the pipeline starts one column earlier
// (level 2 of column ia-2 is needed) and delivers level 4 of column x-3.  Columns ia-3 .. ib+2 take part: class masks are indexed
// by x - ia + 3.
template <bool BODY, bool EMIT, int FD, typename T, int S>
__device__ __forceinline__ void march_unit4(const MarchParams<T> &p, March3Addr<T, S> &m, __amdgpu_buffer_rsrc_t rh,
                                            unsigned hoff, int ia, int ib, int uflags, int j0, int lane, bool far_win,
                                            ClassMask nonfast_m, ClassMask solid_m, const T (&feq0)[9])
{
    constexpr bool OVLF = (FD & MARCH_FD_OVL) != 0;      // overlapping windows: no halo lines, no seam rows (step_chain.hpp k_march3)
    typedef MV<T, S> V3;
    constexpr unsigned HREC = M3_HL * sizeof(T);
    const Geom &g = p.g;
    const MarchAddr<T, S> &a = m.a;
    // (the level-4 stage of the first iteration works on column ia-4, outside the masks: no class -> plain / inlet branch, no mask access)
#define NONFAST(x) (BODY && (x) >= ia - 3 && cm_bit(nonfast_m, (x) - ia + 3))
#define ALLSOLID(x) (BODY && (x) >= ia - 3 && cm_bit(solid_m, (x) - ia + 3))
#define OWN_BUF(x) (a.own_lds + ((x) & 1) * OWN_LDS_BYTES)
#define OWN_NOW(x) do { if (BODY && NONFAST(x)) own_prefetch<T, S, OVLF>(a, (x), OWN_BUF(x)); } while (0)      // (see march_unit3)
#define STEP1(x, in, G) march_step1<BODY, FD, T, S, BODY>(p, a, (x), j0, far_win, NONFAST(x), ALLSOLID(x), feq0, in, G, nullptr, OWN_BUF(x))
#define HCOL(c) ((unsigned)((c) > 0 ? (c) : 0) * HREC)
    const bool outlet = BODY && (uflags & MU_OUTLET_AFTER) != 0;
    const int xend = outlet ? ib : ib + 2;       // last column whose level 1 is computed
    V3 s1m[3], s1c[9];           // level 1: populations 1,5,8 of column x-2; all nine of column x-1
    V3 s2m[3], s2c[9];           // level 2: ... of column x-3; x-2
    V3 s3m[3], s3c[9];           // level 3: ... of column x-4; x-3
    V3 in[9], G1[9], G2[9], G3[9], mac[3];
#pragma unroll
    for (int k = 0; k < 9; k++) { s1c[k] = mv_splat<T, S>(feq0[k]); s2c[k] = s1c[k]; s3c[k] = s1c[k]; }
    s1m[0] = s1c[1]; s1m[1] = s1c[5]; s1m[2] = s1c[8];
    s2m[0] = s1c[1]; s2m[1] = s1c[5]; s2m[2] = s1c[8];
    s3m[0] = s1c[1]; s3m[1] = s1c[5]; s3m[2] = s1c[8];
    // ---- prologue: level 1 of columns ia-3 and ia-2 (columns left of the inlet do not exist: the far-field value stands in)
    const int xs = ia - 1;       // first loop column; when it lies left of the inlet its loads go to the inlet column (values unused)
#define LCOL(x) ((BODY && (x) + g.gi0 < 0) ? -g.gi0 : (x))
    // halo lines of columns x-3, x-2, x-1, x (afterwards: one line per iteration, fetched one iteration ahead, handed from stage to stage)
    T hv3 = halo_load_x<OVLF, T>(rh, hoff, HCOL(xs - 3)), hv2 = halo_load_x<OVLF, T>(rh, hoff, HCOL(xs - 2)), hv1 = halo_load_x<OVLF, T>(rh, hoff, HCOL(xs - 1)),
      hv0 = halo_load_x<OVLF, T>(rh, hoff, HCOL(xs));
    if (!BODY || ia - 3 + g.gi0 >= 0) {
        OWN_NOW(ia - 3);
        march_load_aligned(a, ia - 3, in);
        march_align_in<OVLF>(in, lane, hv2);          // ia - 3 = xs - 2
        STEP1(ia - 3, in, s1c);
    }
    s1m[0] = s1c[1]; s1m[1] = s1c[5]; s1m[2] = s1c[8];
    if (!BODY || ia - 2 + g.gi0 >= 0) {
        OWN_NOW(ia - 2);
        march_load_aligned(a, ia - 2, in);
        march_align_in<OVLF>(in, lane, hv1);          // ia - 2 = xs - 1
        STEP1(ia - 2, in, s1c);
    }
    OWN_NOW(xs);                                // (a column left of the inlet carries no class)
    march_load_aligned(a, LCOL(xs), in);
    // site bytes of columns x, x-1, x-2, x-3 (general loop only; see SiteBytes)
    SiteBytes sb0{{0, 0}}, sb1{{0, 0}}, sb2{{0, 0}}, sb3{{0, 0}};
    if (BODY) {
        sb0 = site_bytes_load<T, S, OVLF>(p, xs, j0); sb1 = site_bytes_load<T, S, OVLF>(p, xs - 1, j0); sb2 = site_bytes_load<T, S, OVLF>(p, xs - 2, j0); sb3 = site_bytes_load<T, S, OVLF>(p, xs - 3, j0);
        wait_for_bytes(sb0); wait_for_bytes(sb1); wait_for_bytes(sb2); wait_for_bytes(sb3);
    }
    wait_for_column(in, hv0, hv1, hv2);
    wait_for_column(in, hv3);
    int seam_col = -1;
#pragma unroll 1
    for (int x = xs; x <= xend; x++) {
        V3 nxt[9];
        if (BODY && x + 1 <= xend && NONFAST(x + 1)) own_prefetch<T, S, OVLF>(a, x + 1, OWN_BUF(x + 1));      // (see march_unit3)
        march_load_aligned(a, (x + 1 <= xend) ? x + 1 : x, nxt);
        const bool has3 = x - 3 >= ia;                                         // column x-3 is an output column
        const int c1 = x - 1, c2 = x - 2, c3 = x - 3;
        const T hvn = halo_load_x<OVLF, T>(rh, hoff, HCOL(x + 1));                      // column x+1's halo line: the next iteration's first stage
        SiteBytes sbn{{0, 0}};
        if (BODY) sbn = site_bytes_load<T, S, OVLF>(p, x + 1, j0);
        const Seam3 sp = seam3_fetch<OVLF>(m);
        march_align_in<OVLF>(in, lane, hv0);
        march_step1<BODY, FD, T, S, BODY>(p, a, x, j0, far_win, NONFAST(x), ALLSOLID(x), feq0, in, G1, BODY ? sb0.v : nullptr, OWN_BUF(x));      // level 1 of column x

        march_stage<BODY, false, FD, 0>(p, c1, j0, lane, far_win, NONFAST(c1), ALLSOLID(c1), feq0, s1m, s1c, G1, hv1, G2, mac, BODY ? sb1.v : nullptr);     // level 2 of x-1
        march_stage<BODY, false, FD, 4>(p, c2, j0, lane, far_win, NONFAST(c2), ALLSOLID(c2), feq0, s2m, s2c, G2, hv2, G3, mac, BODY ? sb2.v : nullptr);     // level 3 of x-2
        V3 out[9];
        march_stage<BODY, EMIT, FD, 8>(p, c3, j0, lane, far_win, NONFAST(c3), ALLSOLID(c3), feq0, s3m, s3c, G3, hv3, out, mac, BODY ? sb3.v : nullptr);     // level 4 of x-3
        pin_after(out);
        wait_for_column(nxt, hvn);
        if (BODY) { wait_for_bytes(sbn); sb3 = sb2; sb2 = sb1; sb1 = sb0; sb0 = sbn; }
        hv3 = hv2; hv2 = hv1; hv1 = hv0; hv0 = hvn;
        march3_store<EMIT, OVLF>(m, has3 ? a.voff_st : p.lat_bytes, has3 ? c3 : 0, out, mac);
        seam3_flush<OVLF>(m, seam_col, sp);
        seam_col = has3 ? c3 : seam_col;
        if (BODY && outlet && x == xend) break;                                // the tail below needs the unshifted state
        s3m[0] = s3c[1]; s3m[1] = s3c[5]; s3m[2] = s3c[8];
        s2m[0] = s2c[1]; s2m[1] = s2c[5]; s2m[2] = s2c[8];
        s1m[0] = s1c[1]; s1m[1] = s1c[5]; s1m[2] = s1c[8];
#pragma unroll
        for (int k = 0; k < 9; k++) { s3c[k] = G3[k]; s2c[k] = G2[k]; s1c[k] = G1[k]; in[k] = nxt[k]; }
    }
    if (BODY && outlet) {
        // x = ib = NX-1 = co (local).  Held: level 1 of co-1 (s1c) and co (G1); level 2 of co-2 (s2c) and co-1 (G2); level 3 of co-3
        // (s3c) and co-2 (G3); level 4 of co-3 is stored.  Outlet rule (html:301-312): level k+1 of co = level k of co-1; its solid
        // sites: own level k reversed.
        const int co = ib;
        uint32_t solid4 = 0;
        if (NONFAST(co)) solid4 = load_site_bytes<S>(p.mask + (long)(co + 1) * g.pitch + site_row<S, OVLF>(j0, g.ny));
        const bool any_solid = __ballot(solid4 != 0) != 0ULL;
        V3 O2[9], O3[9], L3a[9], out[9], tm[3];
        // level 2 of co
#pragma unroll
        for (int k = 0; k < 9; k++) O2[k] = s1c[k];
        if (any_solid) { auto own = [&](int k) { return G1[k]; }; march_solid<T, S, false>(O2, mac, solid4, own); }
        // level 3 of co-1
        tm[0] = s2c[1]; tm[1] = s2c[5]; tm[2] = s2c[8];
        march_stage<BODY, false, FD, 4>(p, co - 1, j0, lane, far_win, NONFAST(co - 1), ALLSOLID(co - 1), feq0, tm, G2, O2, halo_load_x<OVLF, T>(rh, hoff, HCOL(co - 1)), L3a, mac);
        // level 3 of co
#pragma unroll
        for (int k = 0; k < 9; k++) O3[k] = G2[k];
        if (any_solid) { auto own = [&](int k) { return O2[k]; }; march_solid<T, S, false>(O3, mac, solid4, own); }
        // level 4 of co-2
        tm[0] = s3c[1]; tm[1] = s3c[5]; tm[2] = s3c[8];
        Seam3 sp = seam3_fetch<OVLF>(m);
        // (a last unit of a single marched column does not own column co-2, and its level 3 of co-3 is not valid: drop the stores)
        const bool own2 = co - 2 >= ia;
        march_stage<BODY, EMIT, FD, 8>(p, co - 2, j0, lane, far_win, NONFAST(co - 2), ALLSOLID(co - 2), feq0, tm, G3, L3a, halo_load_x<OVLF, T>(rh, hoff, HCOL(co - 2)), out, mac);
        march3_store<EMIT, OVLF>(m, own2 ? a.voff_st : p.lat_bytes, own2 ? co - 2 : 0, out, mac);
        seam3_flush<OVLF>(m, seam_col, sp);
        seam_col = own2 ? co - 2 : seam_col;
        // level 4 of co-1
        tm[0] = G3[1]; tm[1] = G3[5]; tm[2] = G3[8];
        sp = seam3_fetch<OVLF>(m);
        march_stage<BODY, EMIT, FD, 8>(p, co - 1, j0, lane, far_win, NONFAST(co - 1), ALLSOLID(co - 1), feq0, tm, L3a, O3, halo_load_x<OVLF, T>(rh, hoff, HCOL(co - 1)), out, mac);
        march3_store<EMIT, OVLF>(m, a.voff_st, co - 1, out, mac);
        seam3_flush<OVLF>(m, seam_col, sp);
        seam_col = co - 1;
        // level 4 of co
#pragma unroll
        for (int k = 0; k < 9; k++) out[k] = L3a[k];
        if (EMIT) march_outlet_macro(L3a, mac);
        if (any_solid) { auto own = [&](int k) { return O3[k]; }; march_solid<T, S, EMIT>(out, mac, solid4, own); }
        sp = seam3_fetch<OVLF>(m);
        march3_store<EMIT, OVLF>(m, a.voff_st, co, out, mac);
        seam3_flush<OVLF>(m, seam_col, sp);
        seam_col = co;
    }
    seam3_flush<OVLF>(m, seam_col, seam3_fetch<OVLF>(m));
#undef NONFAST
#undef ALLSOLID
#undef STEP1
#undef OWN_BUF
#undef OWN_NOW
#undef HCOL
#undef LCOL
}

// The same machinery stopped after level 2: TWO steps per pass on the tables of the three-step plan (units, classes, halo lines, seam
// buffer S3), for the one or two steps a step count leaves over after its three-step passes.
template <bool BODY, bool EMIT, int FD, typename T, int S>
__device__ __forceinline__ void march_unit3_d2(const MarchParams<T> &p, March3Addr<T, S> &m, __amdgpu_buffer_rsrc_t rh, unsigned hoff, int ia, int ib,
                                               int uflags, int j0, int lane, bool far_win, ClassMask nonfast_m, ClassMask solid_m,
                                               const T (&feq0)[9])
{
    constexpr bool OVLF = (FD & MARCH_FD_OVL) != 0;      // overlapping windows: no halo lines, no seam rows (step_chain.hpp k_march3)
    typedef MV<T, S> V3;
    constexpr unsigned HREC = M3_HL * sizeof(T);
    const Geom &g = p.g;
    const MarchAddr<T, S> &a = m.a;
#define NONFAST(x) (BODY && cm_bit(nonfast_m, (x) - ia + 2))
#define ALLSOLID(x) (BODY && cm_bit(solid_m, (x) - ia + 2))
#define STEP1(x, in, G) march_step1<BODY, FD, T, S>(p, a, (x), j0, far_win, NONFAST(x), ALLSOLID(x), feq0, in, G)
    const bool outlet = BODY && (uflags & MU_OUTLET_AFTER) != 0;
    const int xend = ib;         // last column whose level 1 is computed
    V3 s1m[3], s1c[9];           // level 1: populations 1,5,8 of column x-2; all nine of column x-1
    V3 in[9], G1[9], mac[3];
#pragma unroll
    for (int k = 0; k < 9; k++) s1c[k] = mv_splat<T, S>(feq0[k]);
    s1m[0] = s1c[1]; s1m[1] = s1c[5]; s1m[2] = s1c[8];
    T hv1 = halo_load_x<OVLF, T>(rh, hoff, (unsigned)(ia - 1 > 0 ? ia - 1 : 0) * HREC), hv0 = halo_load_x<OVLF, T>(rh, hoff, (unsigned)ia * HREC);
    if (!BODY || ia - 1 + g.gi0 >= 0) {
        march_load_aligned(a, ia - 1, in);
        march_align_in<OVLF>(in, lane, hv1);
        STEP1(ia - 1, in, s1c);
    }
    march_load_aligned(a, ia, in);
    wait_for_column(in, hv0, hv1);
    int seam_col = -1;
#pragma unroll 1
    for (int x = ia; x <= xend; x++) {
        V3 nxt[9];
        march_load_aligned(a, (x + 1 <= xend) ? x + 1 : x, nxt);
        const bool has1 = x - 1 >= ia;                                         // column x-1 is an output column
        const int c1 = x - 1;
        const T hvn = halo_load_x<OVLF, T>(rh, hoff, (unsigned)(x + 1) * HREC);
        const Seam3 sp = seam3_fetch<OVLF>(m);
        march_align_in<OVLF>(in, lane, hv0);
        STEP1(x, in, G1);
        V3 out[9];
        march_stage<BODY, EMIT, FD, 0>(p, c1, j0, lane, far_win, NONFAST(c1), ALLSOLID(c1), feq0, s1m, s1c, G1, hv1, out, mac);
        pin_after(out);
        wait_for_column(nxt, hvn);
        hv1 = hv0; hv0 = hvn;
        march3_store<EMIT, OVLF>(m, has1 ? a.voff_st : p.lat_bytes, has1 ? c1 : 0, out, mac);
        seam3_flush<OVLF>(m, seam_col, sp);
        seam_col = has1 ? c1 : seam_col;
        if (BODY && outlet && x == xend) break;
        s1m[0] = s1c[1]; s1m[1] = s1c[5]; s1m[2] = s1c[8];
#pragma unroll
        for (int k = 0; k < 9; k++) { s1c[k] = G1[k]; in[k] = nxt[k]; }
    }
    if (BODY && outlet) {
        // x = ib = NX-1: s1c = level 1 of NX-2, G1 = level 1 of NX-1; level 2 of the outlet column = level 1 of NX-2 (html:301-312)
        const int co = ib;
        uint32_t solid4 = 0;
        if (NONFAST(co)) solid4 = load_site_bytes<S>(p.mask + (long)(co + 1) * g.pitch + site_row<S, OVLF>(j0, g.ny));
        V3 out[9];
#pragma unroll
        for (int k = 0; k < 9; k++) out[k] = s1c[k];
        if (EMIT) march_outlet_macro(s1c, mac);
        if (__ballot(solid4 != 0) != 0ULL) { auto own1 = [&](int k) { return G1[k]; }; march_solid<T, S, EMIT>(out, mac, solid4, own1); }
        const Seam3 sp = seam3_fetch<OVLF>(m);
        march3_store<EMIT, OVLF>(m, a.voff_st, co, out, mac);
        seam3_flush<OVLF>(m, seam_col, sp);
        seam_col = co;
    }
    seam3_flush<OVLF>(m, seam_col, seam3_fetch<OVLF>(m));
#undef NONFAST
#undef ALLSOLID
#undef STEP1
}

// Marched column range of a three-step pass: global edges as in march_range; a local slab edge loses THREE columns of
// validity per pass, and level 2 of column 1 would need column -2: the two columns next to a local edge are left alone.
static inline MarchRange march_range3(const Geom &g, int depth = 3)
{
    MarchRange r;
    r.i_begin = (g.gi0 == 0) ? 0 : depth - 1;
    r.outlet_after = (g.gi0 + g.nxl == g.nx_g) ? 1 : 0;
    r.i_end = r.outlet_after ? g.nxl - 1 : g.nxl - (depth - 1);
    return r;
}

}  // namespace wt
