// step_march.hpp — TWO lattice steps per launch over the WHOLE lattice, body included.
//
// Templated on the element type T and the sites per lane S (vector of S * sizeof(T) bytes per lane and direction):
//   <float, 4>  16-byte vectors, 256-row windows — wide fp32 lattices (the bench configuration);
//   <double, 2> 16-byte vectors, 128-row windows — fp64 lattices;
//   <float, 2>   8-byte vectors, 128-row windows — narrow fp32 lattices (column slabs), where 256-row windows give too
//               few units to fill the chip.
// Below, "256 rows" stands for WIN = 64 * S.
//
// A wave owns a WINDOW of 256 rows (j, the fast axis) and marches along a CHUNK of columns (i).
// Per iteration it (1) requests the 9 streamed input vectors of column c+2 (software prefetch),
// (2) runs step 1 (STEP_FS main(), html:283-360) for column c+1 on the vectors requested one
// iteration earlier, (3) runs step 2 for column c from the step-1 populations of columns c-1, c, c+1,
// which never leave the register file (the +-1 shifts along j are lane shuffles), (4) stores 9 vectors.
// HBM words per TWO site updates: 9 (L+2)/L + 9 instead of 18 + 18.
//
// Windows are 256 rows tall and 256 rows apart, so every access of a wave is one whole, 1-KiB-aligned
// kilobyte (measured with the arithmetic removed: a 252-row window pitch — overlapping windows that
// recompute their seam rows — costs 13-20 % of the pass time in line straddles and partial-line stores).
// What step 2 of a window's first and last row needs from the rows just outside the window — the step-1
// populations 2,5,6 of the row below and 4,7,8 of the row above — comes from a small table H that
// k_halo_rows fills before each pass (two rows per window seam, per-site code with every branch of
// STEP_FS); a wave fetches its six values per column with one dword load, one iteration ahead.
//
// The whole lattice is marched by ONE kernel (BODY = true; the BODY = false instantiation — plain fluid only —
// exists for tools/kmarch, where it measured no faster).  Round 1 sent everything near the body through a
// third lattice in two single-step passes; here each window-tile (column x, window w) has a class
//     (k_classify_windows, wave ballots):
//     FAST = no solid site in its 3 x 258 neighbourhood, SOLID = every own site solid, GENERAL = the rest;
//     a unit reads the classes of its columns once (one byte per lane, two ballots -> two 64-bit scalars)
//     and dispatches per column on a scalar bit test.  GENERAL tiles read one dword of solid flags and
//     one dword of BOUNCE CODES per lane (bit k-1 = the upstream neighbour of direction k is solid;
//     k_bounce_codes, once per mask upload) in place of the reference's nine mask texel reads
//     (html:324-334); the sites' own populations (step 1: nine aligned 16-B loads, step 2: the step-1
//     vectors of column c kept in registers) supply the bounced values.  The inlet (html:314-322) and
//     outlet (html:301-312) columns are marched too: the outlet column NX-1 is emitted while column
//     NX-2 is processed (its step-2 value is the step-1 state of NX-2).
// Addressing: buffer instructions — one resource descriptor per lattice in scalar registers, ONE
// per-lane byte offset for all 18 streams, the plane/column part of every address is a scalar add.
// Every site is computed by exactly the arithmetic of k_step: results are bit-identical.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <vector>
#include "step_fast.hpp"

#ifndef WT_LOAD_AUX
#define WT_LOAD_AUX 2       // cache policy of the streamed lattice loads: nt (read once per pass)
#endif
#ifndef WT_STORE_AUX
#define WT_STORE_AUX 2      // cache policy of the lattice stores: nt (written once, read by the NEXT pass); measured 2-4.5 % faster than 0
#endif

namespace wt {

static constexpr int MARCH_MAX_CHUNK = 60;       // two-step kernel: class bytes of columns ia-1 .. ib+1 must fit one wave (one ballot)
static constexpr int MARCH3_MAX_CHUNK = 124;     // three / four steps per pass: two ballots (ClassMask), columns ia-PAD .. ib+PAD-1 <= 128

// classes of up to 128 consecutive columns of a window, one bit each (lane l of the first ballot <-> the first column + l, of the second + 64 + l)
struct ClassMask { unsigned long long lo, hi; };
__device__ __forceinline__ bool cm_bit(const ClassMask &m, int idx) { return (((idx < 64) ? (m.lo >> idx) : (m.hi >> (idx - 64))) & 1ULL) != 0; }

enum : uint8_t { WC_FAST = 0, WC_GENERAL = 1, WC_SOLID = 2 };

// window height = window pitch = 64 lanes x S sites
static inline int march_nwin(int ny, int win) { return (ny + win - 1) / win; }

// collision for either element type: fp32 with the division selected by FD (d2q9.hpp), fp64 always IEEE
static constexpr int MARCH_FD_CONTRACTED = 8;    // bit 3 of FD: the opt-in contracted collision (d2q9.hpp collide_contracted; fp32 only)
static constexpr int MARCH_FD_TWOOP = 16;        // bit 4 of FD: the two-operation division by tau (d2q9.hpp; fp32, proved per tau like the three-operation one)
static constexpr int MARCH_FD_OVL = 32;          // bit 5 of FD (k_march3): OVERLAPPING windows — no halo lines, no seam buffer (step_chain.hpp k_march3); not an arithmetic choice,
                                                 // carried in FD because every function of the marching loop already takes it
template <typename T, int FD>
__device__ __forceinline__ void collide_t(const T (&fin)[9], const FastDiv &fdv, T tau, T (&fo)[9], T &rho, T &ux, T &uy)
{
    if constexpr (sizeof(T) == 4 && (FD & MARCH_FD_CONTRACTED) != 0) collide_contracted(fin, fdv.rtau, fo, rho, ux, uy);
    else if constexpr (sizeof(T) == 4) collide_fd<(FD & 3), (FD & MARCH_FD_TWOOP) != 0>(fin, fdv, fo, rho, ux, uy);
    else if constexpr ((FD & 3) != 0) collide_fd64(fin, fdv, fo, rho, ux, uy);       // fp64: the four-operation division, guarded (d2q9.hpp)
    else collide<T>(fin, tau, fo, rho, ux, uy);
}

// ------------------------------------------------------------------------------------------------
// once per mask upload
// ------------------------------------------------------------------------------------------------
// wcls[w * ld + (x + 1)], x = -1 .. nxl (the two extra columns are FAST), ld = nxl + 2
// (window w = rows w * stride + off .. + win - 1: stride = win, off = 0 for windows that tile the column; overlapping windows: k_march3)
__global__ __launch_bounds__(256) void k_classify_windows(const uint8_t *__restrict__ mask, uint8_t *__restrict__ wcls, Geom g, int nwin, int win, int stride, int off)
{
    const int lane = threadIdx.x & 63;
    const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int ld = g.nxl + 2;
    if (tile >= (long)ld * nwin) return;
    const int w = (int)(tile / ld), x = (int)(tile % ld) - 1;
    uint8_t cls = WC_FAST;
    if (x >= 0 && x < g.nxl) {
        const int j0 = w * stride + off;
        const uint8_t *m = mask + g.pitch;
        int nb = 0, own_all = 1;
        for (int jj = j0 - 1 + lane; jj <= j0 + win; jj += 64) {
            if (jj < 0 || jj >= g.ny) continue;
            const int a = m[(long)(x - 1) * g.pitch + jj], b = m[(long)x * g.pitch + jj], c = m[(long)(x + 1) * g.pitch + jj];
            nb |= a | b | c;
            if (jj >= j0 && jj < j0 + win) own_all &= (b != 0);      // (rows outside the lattice — a window that overlaps its end — do not count)
        }
        const bool any_nb = __ballot(nb != 0) != 0ULL;
        const bool all_own = __ballot(own_all == 0) == 0ULL;
        cls = !any_nb ? WC_FAST : (all_own ? WC_SOLID : WC_GENERAL);
    }
    if (lane == 0) wcls[(long)w * ld + x + 1] = cls;
}

// bcode[x * pitch + j], bit k-1 set <=> site (x - ex_k, j - ey_k) is solid (k = 1..8); 0 on rows 0 / NY-1;
// 0xFF on solid sites
__global__ __launch_bounds__(256) void k_bounce_codes(const uint8_t *__restrict__ mask, uint8_t *__restrict__ bcode, Geom g)
{
    const uint8_t *m = mask + g.pitch;
    const long total = (long)g.nxl * g.pitch;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int x = (int)(t / g.pitch), j = (int)(t % g.pitch);
        unsigned code = 0;
        if (j < g.ny && m[(long)x * g.pitch + j]) code = 0xffu;       // a solid site "bounces" every direction: fin[k] = own[opp(k)]
        else if (j >= 1 && j < g.ny - 1) {
#pragma unroll
            for (int k = 1; k < 9; k++)
                if (m[(long)(x - ex_of(k)) * g.pitch + (j - ey_of(k))]) code |= 1u << (k - 1);
        }
        bcode[t] = (uint8_t)code;
    }
}

// seam_plain[(b - 1) * nxl + x] = 1 <=> both sites next to seam b in column x (rows 256b-1 and 256b) are plain interior
// fluid: not solid, no solid neighbour, not on an inlet / outlet column or the top row.  The halo kernels then read one
// coalesced byte instead of four scattered mask / code bytes per thread.
__global__ __launch_bounds__(256) void k_seam_flags(const uint8_t *__restrict__ mask, const uint8_t *__restrict__ bcode, uint8_t *__restrict__ seam_plain,
                                                    Geom g, int nwin, int win)
{
    const long total = (long)(nwin - 1) * g.nxl;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int x = (int)(t % g.nxl);
    const int b = 1 + (int)(t / g.nxl);
    const int j = win * b - 1;
    const uint8_t *m = mask + g.pitch;
    const long c = (long)x * g.pitch + j;
    const int gi = x + g.gi0;
    bool plain = false;
    if (j + 1 < g.ny - 1 && gi > 0 && gi < g.nx_g - 1) plain = bcode[c] == 0 && bcode[c + 1] == 0 && m[c] == 0 && m[c + 1] == 0;
    seam_plain[t] = plain ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// once per pass: the halo table
// ------------------------------------------------------------------------------------------------
// One lattice site, step 1 only, every branch of STEP_FS main() (html:283-360) in the reference's order —
// the arithmetic of site_general (kernels.hpp) with the relaxation of collide_t.
template <typename T, int FD>
__device__ __forceinline__ void site_step1(const T *__restrict__ s, const uint8_t *__restrict__ m, const Geom &g, int i, int j,
                                           const FastDiv &fdv, T tau, T U0, T (&out)[9])
{
    const long c = (long)i * g.pitch + j;
    const int gi = i + g.gi0;
    if (m[c]) {                                                    // html:287-294 solid
#pragma unroll
        for (int k = 0; k < 9; k++) out[k] = s[opp_of(k) * g.plane + c];
    } else if (gi == g.nx_g - 1) {                                 // html:301-312 outlet
#pragma unroll
        for (int k = 0; k < 9; k++) out[k] = s[k * g.plane + c - g.pitch];
    } else if (gi == 0 || j == g.ny - 1 || j == 0) {               // html:314-322 far field
        feq_all<T>(T(1), U0, T(0), out);
    } else {                                                       // html:324-359 interior fluid
        T fin[9], rho, ux, uy;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const long src = c - (long)ex_of(k) * g.pitch - ey_of(k);
            fin[k] = m[src] ? s[opp_of(k) * g.plane + c] : s[k * g.plane + src];
        }
        collide_t<T, FD>(fin, fdv, tau, out, rho, ux, uy);
    }
}

// H[(b * (nxl+2) + x + 1) * 8 + ...], seam b = 1 .. nwin-1 lies between rows WIN*b-1 and WIN*b:
//   [0..2] = step-1 populations 2,5,6 of row WIN*b-1 (they move up into window b),
//   [4..6] = step-1 populations 4,7,8 of row WIN*b   (they move down into window b-1).
// One thread per (seam, column) computes both rows.  Plain fluid sites (no solid neighbour, interior column — the
// seam flags say so) take nine 4-element loads: rows WIN*b-2 .. WIN*b+1 of each direction's upstream column hold every
// input of the two sites; anything else falls back to site_step1.  The loads are a gather (neighbouring threads are
// one column = pitch elements apart); that, not the arithmetic, is this kernel's cost.
template <typename T, int FD>
__global__ __launch_bounds__(256) void k_halo_rows(const T *__restrict__ fs, const uint8_t *__restrict__ mask, const uint8_t *__restrict__ seam_plain,
                                                   T *__restrict__ halo, Geom g, int nwin, int win, FastDiv fdv, T tau, T U0)
{
    const long total = (long)(nwin - 1) * g.nxl;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    // neighbouring threads take the seams of ONE column, one window apart (neighbouring columns, a pitch apart, measured
    // the same: the kernel is bound by the rate of scattered 64-byte fetches, ~130 MB of them per 4096^2 fp32 lattice)
    const int b = 1 + (int)(t % (nwin - 1));
    const int x = (int)(t / (nwin - 1));
    const int j = win * b - 1;                       // rows j (below the seam) and j + 1 (above it)
    if (j >= g.ny) return;
    const T *s = fs + g.pitch;
    const uint8_t *m = mask + g.pitch;
    const long c = (long)x * g.pitch + j;
    T lo[9], hi[9];
    const bool two = (j + 1 < g.ny);
    const bool plain = seam_plain[(long)(b - 1) * g.nxl + x] != 0;
    if (plain) {
        typedef T q4u __attribute__((ext_vector_type(4), aligned(sizeof(T))));
        T a[9], d[9], rho, ux, uy;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const q4u q = *reinterpret_cast<const q4u *>(s + k * g.plane + c - (long)ex_of(k) * g.pitch - 1);     // rows j-1 .. j+2
            a[k] = q[1 - ey_of(k)];      // input of row j
            d[k] = q[2 - ey_of(k)];      // input of row j+1
        }
        collide_t<T, FD>(a, fdv, tau, lo, rho, ux, uy);
        collide_t<T, FD>(d, fdv, tau, hi, rho, ux, uy);
    } else {
        site_step1<T, FD>(s, m, g, x, j, fdv, tau, U0, lo);
        if (two) site_step1<T, FD>(s, m, g, x, j + 1, fdv, tau, U0, hi);
        else {
#pragma unroll
            for (int k = 0; k < 9; k++) hi[k] = T(0);
        }
    }
    typedef T t4 __attribute__((ext_vector_type(4)));
    t4 *rec = reinterpret_cast<t4 *>(halo + ((long)b * (g.nxl + 2) + x + 1) * 8);
    rec[0] = t4{lo[2], lo[5], lo[6], T(0)};
    rec[1] = t4{hi[4], hi[7], hi[8], T(0)};
}

// The same table from the seam buffer S that the PREVIOUS marching pass wrote beside the lattice it produced (valid
// only then: the library tracks it).  One thread per (seam, column, SIDE): side 0 = the row below the seam (its step-1
// populations 2,5,6 move up), side 1 = the row above it (4,7,8 move down) — twice the threads of k_halo_rows, because
// this kernel is latency-bound (a 4096^2 lattice has only 61 440 (seam, column) pairs).  Neighbouring threads read
// neighbouring 48-element records: coalesced, 13 MB instead of a 130 MB gather.  Sites near the body / on the inlet and
// outlet columns fall back to site_step1 on the lattice.
template <typename T, int FD>
__global__ __launch_bounds__(256) void k_halo_from_seams(const T *__restrict__ fs, const T *__restrict__ seams, const uint8_t *__restrict__ mask,
                                                         const uint8_t *__restrict__ seam_plain, T *__restrict__ halo, Geom g, int nwin, int win,
                                                         FastDiv fdv, T tau, T U0)
{
    const long total = (long)(nwin - 1) * g.nxl;
    const long t2 = (long)blockIdx.x * 256 + threadIdx.x;
    const int side = (int)(t2 & 1);
    const long t = t2 >> 1;
    if (t >= total) return;
    const int x = (int)(t % g.nxl);
    const int b = 1 + (int)(t / g.nxl);
    const int j = win * b - 1 + side;                // side 0: row win*b-1, side 1: row win*b
    T o[9];
    if (j >= g.ny) {
#pragma unroll
        for (int k = 0; k < 9; k++) o[k] = T(0);
    } else if (seam_plain[t] != 0) {
        T a[9], rho, ux, uy;
        const T *rec = seams + ((long)b * (g.nxl + 2) + x + 1) * 48;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            // record of the upstream column: slot k = {row win*b-2, row win*b-1 | +24: row win*b, row win*b+1}; the input of
            // row r for direction k is row r - ey_k
            const T *r = rec - (long)ex_of(k) * 48 + 2 * k;
            const int q = 1 + side - ey_of(k);       // 0..3 over the four rows around the seam
            a[k] = r[(q >> 1) * 24 + (q & 1)];
        }
        collide_t<T, FD>(a, fdv, tau, o, rho, ux, uy);
    } else {
        site_step1<T, FD>(fs + g.pitch, mask + g.pitch, g, x, j, fdv, tau, U0, o);
    }
    typedef T t4 __attribute__((ext_vector_type(4)));
    t4 *out = reinterpret_cast<t4 *>(halo + ((long)b * (g.nxl + 2) + x + 1) * 8);
    out[side] = side ? t4{o[4], o[7], o[8], T(0)} : t4{o[2], o[5], o[6], T(0)};
}

// ------------------------------------------------------------------------------------------------
// units
// ------------------------------------------------------------------------------------------------
struct MarchUnit { int ia, ib, w, flags; };      // marched columns [ia, ib), window; flags bit 0: emit the outlet column ib with column ib-1
enum { MU_OUTLET_AFTER = 1 };

template <typename T>
struct MarchParams {
    const T *fs;
    T *fd;
    T *macro;
    const uint8_t *mask;       // padded byte mask (column -1 first)
    const uint8_t *bcode;      // bounce codes, column 0 first
    const uint8_t *wcls;       // window-tile classes [nwin][nxl + 2]
    const T *halo;             // H[nwin + 1][nxl + 2][8], see k_halo_rows
    const T *hlines;           // k_march3 (step_march3.hpp): the halo lines HL[nwin + 1][nxl + 2][32], levels 0 .. depth-1 of every (seam, column)
    T *seams;                  // S[nwin + 1][nxl + 2][2][24]: the rows around every window seam of the DESTINATION lattice, see k_halo_from_seams
    const MarchUnit *units;
    int nunits;
    Geom g;
    unsigned lat_bytes;        // bytes of one lattice (9 planes) — below 4 GiB
    int nwin_total;            // windows per column
    int win_stride = 0;        // k_march3: rows from one window to the next (0: the window height, 64 S — windows that tile the column and take the rows
    int win_off = 0;           // beyond their seams out of the halo lines); OVERLAPPING windows (stride 64 S - 8, offset -4): see k_march3
    FastDiv fdv;               // fp32: tau and RN(1/tau)
    T tau;
    T U0;
    int rev;
    unsigned long long *clk = nullptr;   // tuning passes only (tune_fuse_plan): [unit] = {start, end} of every unit in s_memtime ticks
    unsigned int *stuck = nullptr;       // host-visible word: a chain unit that gave up waiting for its partner's hand-over sets it (step_chain.hpp)
};

// S consecutive rows of one column and direction, held by one lane
template <typename T, int S> struct MV { T v[S]; };
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
typedef unsigned int u2v __attribute__((ext_vector_type(2)));
template <int BYTES> struct RawOf;
template <> struct RawOf<16> { typedef u4v type; };
template <> struct RawOf<8> { typedef u2v type; };

template <typename T, int S> __device__ __forceinline__ MV<T, S> mv_splat(T x)
{
    MV<T, S> r;
#pragma unroll
    for (int v = 0; v < S; v++) r.v[v] = x;
    return r;
}

// the value held by lane - 1 / lane + 1 as ONE data-parallel-primitive move each (v_mov_b32_dpp wave_shr:1 / wave_shl:1; lane 0 /
// lane 63 keep their own value) — __shfl_up / __shfl_down compile to ds_bpermute_b32, an LDS round trip with a wait behind it
__device__ __forceinline__ int dpp_wave_up(int x) { return __builtin_amdgcn_update_dpp(x, x, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int dpp_wave_down(int x) { return __builtin_amdgcn_update_dpp(x, x, 0x130, 0xf, 0xf, false); }
__device__ __forceinline__ float wave_up(float x) { return __int_as_float(dpp_wave_up(__float_as_int(x))); }
__device__ __forceinline__ float wave_down(float x) { return __int_as_float(dpp_wave_down(__float_as_int(x))); }
__device__ __forceinline__ double wave_up(double x) { return __hiloint2double(dpp_wave_up(__double2hiint(x)), dpp_wave_up(__double2loint(x))); }
__device__ __forceinline__ double wave_down(double x) { return __hiloint2double(dpp_wave_down(__double2hiint(x)), dpp_wave_down(__double2loint(x))); }

// value at j-1 / j+1 taken from the neighbouring lane
// (lane 0 / lane 63 take the value of the row outside the window from the halo table: `edge`, wave-uniform)
template <typename T, int S> __device__ __forceinline__ MV<T, S> m_below(const MV<T, S> &r, int lane, T edge)
{
    MV<T, S> o;
    const T n = wave_up(r.v[S - 1]);
    o.v[0] = lane == 0 ? edge : n;
#pragma unroll
    for (int v = 1; v < S; v++) o.v[v] = r.v[v - 1];
    return o;
}
template <typename T, int S> __device__ __forceinline__ MV<T, S> m_above(const MV<T, S> &r, int lane, T edge)
{
    MV<T, S> o;
    const T n = wave_down(r.v[0]);
#pragma unroll
    for (int v = 0; v < S - 1; v++) o.v[v] = r.v[v + 1];
    o.v[S - 1] = lane == 63 ? edge : n;
    return o;
}

// The same with the row outside the window taken straight out of the column's halo register `hv` (step_march3.hpp: lanes 0..15 hold the words that
// come from below, lanes 48..63 those from above) by a ROW shift — lane N into lane 0 (resp. lane 48 + P into lane 63) — and the wave shift then leaving
// that lane alone: two data-parallel-primitive moves per direction instead of a v_readlane, a move and a select (the kernel is bound by the number of
// vector instructions it issues, and the six wave-uniform edge values of every stage no longer pass through scalar registers).
template <int N> __device__ __forceinline__ int dpp_shift_below(int x, int hv)
{
    static_assert(N >= 0 && N < 16, "from-below words live in lanes 0..15");
    // (mov_dpp: the lanes the row shift does not write are don't-care — only lane 0 of `e` is used —, so the move needs no copy of hv in its destination first)
    int e = hv;
    if constexpr (N > 0) e = __builtin_amdgcn_mov_dpp(hv, 0x100 + N, 0x1, 0xf, false);               // row_shl:N, row 0 only: lane 0 <- lane N
    return __builtin_amdgcn_update_dpp(e, x, 0x138, 0xf, 0xf, false);                                   // wave_shr:1: lane i <- lane i-1, lane 0 keeps e
}
template <int P> __device__ __forceinline__ int dpp_shift_above(int x, int hv)
{
    static_assert(P >= 0 && P < 16, "from-above words live in lanes 48..63");
    int e = hv;
    if constexpr (P < 15) e = __builtin_amdgcn_mov_dpp(hv, 0x110 + (15 - P), 0x8, 0xf, false);          // row_shr:15-P, row 3 only: lane 63 <- lane 48+P
    return __builtin_amdgcn_update_dpp(e, x, 0x130, 0xf, 0xf, false);                                   // wave_shl:1: lane i <- lane i+1, lane 63 keeps e
}
template <int N> __device__ __forceinline__ float shift_below_h(float x, float hv) { return __int_as_float(dpp_shift_below<N>(__float_as_int(x), __float_as_int(hv))); }
template <int N> __device__ __forceinline__ double shift_below_h(double x, double hv)
{
    return __hiloint2double(dpp_shift_below<N>(__double2hiint(x), __double2hiint(hv)), dpp_shift_below<N>(__double2loint(x), __double2loint(hv)));
}
template <int P> __device__ __forceinline__ float shift_above_h(float x, float hv) { return __int_as_float(dpp_shift_above<P>(__float_as_int(x), __float_as_int(hv))); }
template <int P> __device__ __forceinline__ double shift_above_h(double x, double hv)
{
    return __hiloint2double(dpp_shift_above<P>(__double2hiint(x), __double2hiint(hv)), dpp_shift_above<P>(__double2loint(x), __double2loint(hv)));
}
template <int N, typename T, int S> __device__ __forceinline__ MV<T, S> m_below_h(const MV<T, S> &r, T hv)
{
    MV<T, S> o;
    o.v[0] = shift_below_h<N>(r.v[S - 1], hv);
#pragma unroll
    for (int v = 1; v < S; v++) o.v[v] = r.v[v - 1];
    return o;
}
template <int P, typename T, int S> __device__ __forceinline__ MV<T, S> m_above_h(const MV<T, S> &r, T hv)
{
    MV<T, S> o;
#pragma unroll
    for (int v = 0; v < S - 1; v++) o.v[v] = r.v[v + 1];
    o.v[S - 1] = shift_above_h<P>(r.v[0], hv);
    return o;
}
// ... and for OVERLAPPING windows (OVL; k_march3): the first / last lane's row beyond the window is a margin row's input — don't-care —, so the wave
// shift alone (that lane keeps its own value: finite) does, one data-parallel-primitive move per direction
template <int N, bool OVL, typename T, int S> __device__ __forceinline__ MV<T, S> m_below_x(const MV<T, S> &r, T hv)
{
    if constexpr (!OVL) return m_below_h<N>(r, hv);
    else {
        MV<T, S> o;
        o.v[0] = wave_up(r.v[S - 1]);
#pragma unroll
        for (int v = 1; v < S; v++) o.v[v] = r.v[v - 1];
        return o;
    }
}
template <int P, bool OVL, typename T, int S> __device__ __forceinline__ MV<T, S> m_above_x(const MV<T, S> &r, T hv)
{
    if constexpr (!OVL) return m_above_h<P>(r, hv);
    else {
        MV<T, S> o;
#pragma unroll
        for (int v = 0; v < S - 1; v++) o.v[v] = r.v[v + 1];
        o.v[S - 1] = wave_down(r.v[0]);
        return o;
    }
}

// buffer addressing: rsrc = whole lattice; voff = the lane's byte offset (j0 * sizeof(T), loop-invariant);
// soff = scalar byte offset of (plane, column, row shift)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t march_rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
template <typename T, int S>
__device__ __forceinline__ MV<T, S> bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    MV<T, S> o;
    if constexpr (S * sizeof(T) == 16) {
        const u4v x = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, WT_LOAD_AUX);
        __builtin_memcpy(&o, &x, 16);
    } else {
        const u2v x = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, WT_LOAD_AUX);
        __builtin_memcpy(&o, &x, 8);
    }
    return o;
}
// returns the raw data registers of the store, for store_data_fence()
template <typename T, int S>
__device__ __forceinline__ typename RawOf<S * sizeof(T)>::type bstore(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, const MV<T, S> &v)
{
    typename RawOf<S * sizeof(T)>::type x;
    __builtin_memcpy(&x, &v, S * sizeof(T));
    if constexpr (S * sizeof(T) == 16) __builtin_amdgcn_raw_buffer_store_b128(x, r, voff, soff, WT_STORE_AUX);
    else __builtin_amdgcn_raw_buffer_store_b64(x, r, voff, soff, WT_STORE_AUX);
    return x;
}
// two elements (one seam slot)
template <typename T>
__device__ __forceinline__ typename RawOf<2 * sizeof(T)>::type bstore2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, T a, T b)
{
    typename RawOf<2 * sizeof(T)>::type x;
    const T ab[2] = {a, b};
    __builtin_memcpy(&x, ab, 2 * sizeof(T));
    if constexpr (sizeof(T) == 8) __builtin_amdgcn_raw_buffer_store_b128(x, r, voff, soff, 0);
    else __builtin_amdgcn_raw_buffer_store_b64(x, r, voff, soff, 0);
    return x;
}
// STORE-DATA HAZARD (measured on MI355X, ROCm 7.2): a buffer_store_dwordx4 whose soffset is an SGPR reads its data
// VGPRs a little after it issues.  A VALU instruction right behind it that overwrites one of them can win that race for
// the last lanes of each 16-lane group when the memory pipe is backed up — the store then writes the NEW register
// contents (seen as 4-lane groups of wrong macro values at the outlet column of a 16384 x 4096 lattice, nowhere else
// and not on every run).  hipcc pads this hazard only for stores WITHOUT a register soffset (LLVM GCNHazardRecognizer::
// createsVALUHazard), so the pad is ours: the asm keeps every data tuple of a store group live up to this point and
// its `s_nop 1` gives the two wait states after the group's last store.  tools/check_store_hazard.py verifies the
// generated ISA (run by tests/test_build_hazards.py).  (8-byte stores have no such hazard; the fence is harmless there.)
template <typename R>
__device__ __forceinline__ void store_data_fence(const R (&d)[12], int n)
{
    if (n == 12)
        asm volatile("s_nop 1" ::"v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(d[6]), "v"(d[7]), "v"(d[8]), "v"(d[9]), "v"(d[10]), "v"(d[11]));
    else
        asm volatile("s_nop 1" ::"v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(d[6]), "v"(d[7]), "v"(d[8]));
}
template <typename R>
__device__ __forceinline__ void store_data_fence2(const R &a, const R &b)
{
    if constexpr (sizeof(R) == 16) asm volatile("s_nop 1" ::"v"(a), "v"(b));
}

template <typename T, int S>
struct MarchAddr {
    __amdgpu_buffer_rsrc_t rs, rd, rm;   // source lattice, destination lattice, macro planes
    unsigned voff;                       // j0 * sizeof(T)
    unsigned voff_st;                    // the same for stores; lanes beyond the last row: out of range (dropped by the buffer check)
    __amdgpu_buffer_rsrc_t rseam;        // seam buffer S
    unsigned voff_lo, voff_hi;           // lanes 0..11: byte offsets of their slot in the two half records this window writes; other lanes: out of range
    T *lds_w, *lds_r;                    // this lane's LDS address for staging (write) and for the transposed read-back
    int lane;
    unsigned P4, pitch4, mp4;            // plane / column / macro-plane strides in bytes
    char *own_lds;                       // k_march3, general (BODY) units: this wave's two LDS buffers for a column's own populations (own_prefetch); else null
    unsigned voff_dma;                   // the lane's byte offset for a dword-per-lane LDS-DMA load of the window's first 64 dwords: row0 * sizeof(T) + 4 lane
};

// byte offset of (plane k, local column col, row shift dj) from the lattice base (one pad column in front)
template <typename T, int S>
__device__ __forceinline__ unsigned lat_off(const MarchAddr<T, S> &a, int k, int col, int dj)
{
    return (unsigned)k * a.P4 + (unsigned)(col + 1) * a.pitch4 + (unsigned)(dj * (int)sizeof(T));
}

// the nine streamed (pulled) input vectors of column `col`
template <typename T, int S>
__device__ __forceinline__ void march_load_stream(const MarchAddr<T, S> &a, int col, MV<T, S> (&fin)[9])
{
    fin[0] = bload<T, S>(a.rs, a.voff, lat_off(a, 0, col, 0));
    fin[1] = bload<T, S>(a.rs, a.voff, lat_off(a, 1, col - 1, 0));
    fin[3] = bload<T, S>(a.rs, a.voff, lat_off(a, 3, col + 1, 0));
    fin[2] = bload<T, S>(a.rs, a.voff, lat_off(a, 2, col, -1));
    fin[5] = bload<T, S>(a.rs, a.voff, lat_off(a, 5, col - 1, -1));
    fin[6] = bload<T, S>(a.rs, a.voff, lat_off(a, 6, col + 1, -1));
    fin[4] = bload<T, S>(a.rs, a.voff, lat_off(a, 4, col, 1));
    fin[7] = bload<T, S>(a.rs, a.voff, lat_off(a, 7, col + 1, 1));
    fin[8] = bload<T, S>(a.rs, a.voff, lat_off(a, 8, col - 1, 1));
}

// The same nine inputs WITHOUT the row shift (k_march3): every load is the window's own 64 S rows of the upstream column — 4 whole 128-byte
// lines (fp32) instead of 5, the fifth being a line of the neighbouring window that its wave fetches as well.  The shift by one row happens
// in registers (march_align_in), the row outside the window comes from the column's halo line (level-0 words, step_march3.hpp).
// Measured before: 46.7 line requests per column and window against 36 + 1 (profiles/r04_u_fetch_calibration.txt).
template <typename T, int S>
__device__ __forceinline__ void march_load_aligned(const MarchAddr<T, S> &a, int col, MV<T, S> (&fin)[9])
{
    fin[0] = bload<T, S>(a.rs, a.voff, lat_off(a, 0, col, 0));
    fin[1] = bload<T, S>(a.rs, a.voff, lat_off(a, 1, col - 1, 0));
    fin[3] = bload<T, S>(a.rs, a.voff, lat_off(a, 3, col + 1, 0));
    fin[2] = bload<T, S>(a.rs, a.voff, lat_off(a, 2, col, 0));
    fin[5] = bload<T, S>(a.rs, a.voff, lat_off(a, 5, col - 1, 0));
    fin[6] = bload<T, S>(a.rs, a.voff, lat_off(a, 6, col + 1, 0));
    fin[4] = bload<T, S>(a.rs, a.voff, lat_off(a, 4, col, 0));
    fin[7] = bload<T, S>(a.rs, a.voff, lat_off(a, 7, col + 1, 0));
    fin[8] = bload<T, S>(a.rs, a.voff, lat_off(a, 8, col - 1, 0));
}

// S mask / bounce-code bytes of a lane's sites, site v in bits 8v .. 8v+7
template <int S> __device__ __forceinline__ uint32_t load_site_bytes(const uint8_t *p)
{
    if constexpr (S == 4) return *reinterpret_cast<const uint32_t *>(p);
    else if constexpr (S == 2) return *reinterpret_cast<const uint16_t *>(p);
    else return *p;
}

// The solid flags and bounce codes of a lane's sites in one column do not depend on the level, and a unit's general loop needs them for
// the same column at consecutive iterations (STEP_FS at x, the next level at x - 1, ...).  Fetched inside the stage that uses them they
// sit behind a scalar branch, and hipcc's wait for them drains vmcnt to 0 — the prefetched column and the stores of the previous
// iteration included: a round trip to memory per level and iteration (27 such waits in the four-step general loop; per-unit clocks: a body
// column cost 3.2 plain ones).  Fetched ONE ITERATION AHEAD for every column of a general unit, beside the prefetched populations, and
// handed down from level to level in registers, they cost two byte loads per iteration and no wait of their own.
struct SiteBytes { uint32_t v[2]; };     // {solid4, code4}
// the row a lane's mask / bounce-code bytes are fetched from: its own first row, or — the margin lanes of an overlapping window below row 0 or beyond
// the last row (k_march3), whose results nobody stores — the nearest rows that exist
template <int S, bool OVL> __device__ __forceinline__ int site_row(int j0, int ny)
{
    if constexpr (OVL) return j0 < 0 ? 0 : (j0 > ny - S ? ny - S : j0);
    else return j0;
}
template <typename T, int S, bool OVL = false>
__device__ __forceinline__ SiteBytes site_bytes_load(const MarchParams<T> &p, int col, int j0)
{
    const int c = col < 0 ? 0 : (col > p.g.nxl - 1 ? p.g.nxl - 1 : col);       // (columns outside carry no class: their bytes are never used)
    const int jb = site_row<S, OVL>(j0, p.g.ny);
    SiteBytes r;
    r.v[0] = load_site_bytes<S>(p.mask + (long)(c + 1) * p.g.pitch + jb);
    r.v[1] = load_site_bytes<S>(p.bcode + (long)c * p.g.pitch + jb);
    return r;
}

// Two fp32 sites per lane as 2-vectors: the operations of collide_head / collide_tail (d2q9.hpp), element for element, written
// on float2 so that hipcc emits packed instructions (v_pk_add / v_pk_mul / v_pk_fma_f32) without the register-shuffling moves
// its SLP vectoriser pays for the same pairs (a third of the loop's vector instructions go away).  Selected by bit 2 of FD
// (MARCH_FD_PACKED): measured in alternating runs on one box it gains 3.4 % in the four-step kernel (158 -> 163 GLUPS), which
// is bound by its instruction stream, and loses 1-1.5 % in the three-step kernel on slab-sized lattices.
static constexpr int MARCH_FD_PACKED = 4;
typedef float f2v __attribute__((ext_vector_type(2)));
// a / b for two sites at once, operation for operation hipcc's own expansion of the IEEE binary32 division (LLVM AMDGPUTargetLowering::LowerFDIV32 with
// fp32 denormals on: v_div_scale x2, v_rcp, fma, fma, mul, fma, fma, fma, v_div_fmas, v_div_fixup) — the same instructions on the same values, so the same
// bits — with the six multiply-adds in between as PACKED instructions over the two sites: the marching kernel is bound by the number of vector
// instructions it issues (an un-packed collision costs it 20 %), and the two divisions per site were a sixth of them.
__device__ __forceinline__ f2v div2_ieee(f2v a, f2v b)
{
    bool s0, s1, dummy;
    const f2v ds = {__builtin_amdgcn_div_scalef(a[0], b[0], false, &dummy), __builtin_amdgcn_div_scalef(a[1], b[1], false, &dummy)};     // denominator scaled
    const f2v ns = {__builtin_amdgcn_div_scalef(a[0], b[0], true, &s0), __builtin_amdgcn_div_scalef(a[1], b[1], true, &s1)};             // numerator scaled
    const f2v rc = {__builtin_amdgcn_rcpf(ds[0]), __builtin_amdgcn_rcpf(ds[1])};
    const f2v one = {1.0f, 1.0f};
    const f2v f0 = __builtin_elementwise_fma(-ds, rc, one);
    const f2v f1 = __builtin_elementwise_fma(f0, rc, rc);
    const f2v mu = ns * f1;
    const f2v f2 = __builtin_elementwise_fma(-ds, mu, ns);
    const f2v f3 = __builtin_elementwise_fma(f2, f1, mu);
    const f2v f4 = __builtin_elementwise_fma(-ds, f3, ns);
    f2v q;
    q[0] = __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(f4[0], f1[0], f3[0], s0), b[0], a[0]);
    q[1] = __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(f4[1], f1[1], f3[1], s1), b[1], a[1]);
    return q;
}

template <int FD, bool TWO_OP>
__device__ __forceinline__ void collide2_packed(const f2v (&fin)[9], const FastDiv &fdv, f2v (&fo)[9], f2v &rho, f2v &ux, f2v &uy)
{
    f2v r, u, v;
    {   // moments (d2q9.hpp, html:335-338) with the two divisions packed
        // (0.0 + fin[0] of the reference's sum is fin[0] itself unless fin[0] = -0, and a density sum that starts at -0 instead of +0 ends on the same
        //  value unless all nine populations are -0 — a state whose clamped density, 0.5, and NaN velocity are the same either way: the addition is dropped)
        f2v rs = fin[0];
#pragma unroll
        for (int k = 1; k < 9; k++) rs += fin[k];
        r = rs;
        u = div2_ieee(fin[1] + fin[5] + fin[8] - fin[3] - fin[6] - fin[7], rs);
        v = div2_ieee(fin[2] + fin[5] + fin[6] - fin[4] - fin[7] - fin[8], rs);
    }
    const float rhoMin = 0.5f, rhoMax = 2.0f, uMax = 0.35f;        // html:344
#pragma unroll
    for (int i = 0; i < 2; i++) {
        float ri = r[i];
        ri = (ri < rhoMin) ? rhoMin : ri;
        ri = (rhoMax < ri) ? rhoMax : ri;
        r[i] = ri;
    }
    const f2v spd2 = u * u + v * v;
    bool safe = true;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const float m0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(fin[0][i]), __builtin_fabsf(fin[1][i])), __builtin_fabsf(fin[2][i]));
        const float m1 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(fin[3][i]), __builtin_fabsf(fin[4][i])), __builtin_fabsf(fin[5][i]));
        const float m2 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(fin[6][i]), __builtin_fabsf(fin[7][i])), __builtin_fabsf(fin[8][i]));
        const float m = __builtin_fmaxf(__builtin_fmaxf(m0, m1), m2);
        safe = safe && (m < 0x1p100f) && (r[i] == r[i]) && (spd2[i] == spd2[i]);
    }
    const bool fast = (FD == 2) || (FD == 1 && __ballot(!safe) == 0ULL);
#pragma unroll
    for (int i = 0; i < 2; i++) {
        if (spd2[i] > uMax * uMax) {
            const float k = uMax / wt_sqrt<float>(spd2[i]);
            u[i] *= k;
            v[i] *= k;
        }
    }
    f2v eq[9];
    feq_all<f2v>(r, u, v, eq);
    if (FD != 0 && fast) {
        const f2v rt = {fdv.rtau, fdv.rtau}, ta = {fdv.tau, fdv.tau}, rl = {fdv.rlo, fdv.rlo};
        // three directions abreast: a packed fp32 instruction whose result the very next instruction needs costs a wait state (s_nop) —
        // written chain by chain the nine relaxations were a fifth of the loop's issue slots in nops
#pragma unroll
        for (int g = 0; g < 9; g += 3) {
            f2v x[3], q0[3], e[3], t[3];
#pragma unroll
            for (int j = 0; j < 3; j++) x[j] = fin[g + j] - eq[g + j];
            if constexpr (TWO_OP) {      // div_by_tau_fast<true>: p = x rlo, q = fma(x, r, p)
#pragma unroll
                for (int j = 0; j < 3; j++) q0[j] = x[j] * rl;
#pragma unroll
                for (int j = 0; j < 3; j++) t[j] = __builtin_elementwise_fma(x[j], rt, q0[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 3; j++) q0[j] = x[j] * rt;
#pragma unroll
                for (int j = 0; j < 3; j++) e[j] = __builtin_elementwise_fma(-q0[j], ta, x[j]);
#pragma unroll
                for (int j = 0; j < 3; j++) t[j] = __builtin_elementwise_fma(e[j], rt, q0[j]);
            }
#pragma unroll
            for (int j = 0; j < 3; j++) fo[g + j] = fin[g + j] - t[j];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const f2v x = fin[k] - eq[k];
            f2v q;
            q[0] = x[0] / fdv.tau; q[1] = x[1] / fdv.tau;
            fo[k] = fin[k] - q;
        }
    }
    rho = r; ux = u; uy = v;
}

// collide S sites per lane; fp32: one wave-uniform decision between the fast and the IEEE division by tau
template <typename T, int S, int FD>
__device__ __forceinline__ void march_collide_sites(const MV<T, S> (&fin)[9], const FastDiv &fdv, T tau, MV<T, S> (&o)[9], MV<T, S> &rho4, MV<T, S> &ux4,
                                                    MV<T, S> &uy4)
{
    constexpr int FDV = FD & 3;          // how to divide by tau (d2q9.hpp)
    constexpr bool TWO = (FD & MARCH_FD_TWOOP) != 0;
    if constexpr (sizeof(T) == 4 && (FD & MARCH_FD_CONTRACTED) != 0) {
#pragma unroll
        for (int v = 0; v < S; v++) {
            float a[9], f[9], r, u, w;
#pragma unroll
            for (int k = 0; k < 9; k++) a[k] = fin[k].v[v];
            collide_contracted(a, fdv.rtau, f, r, u, w);
#pragma unroll
            for (int k = 0; k < 9; k++) o[k].v[v] = f[k];
            rho4.v[v] = r; ux4.v[v] = u; uy4.v[v] = w;
        }
    } else if constexpr (sizeof(T) == 4 && S == 2 && (FD & MARCH_FD_PACKED) != 0) {
        f2v a[9], f[9], r, u, w;
#pragma unroll
        for (int k = 0; k < 9; k++) a[k] = f2v{fin[k].v[0], fin[k].v[1]};
        collide2_packed<FDV, TWO>(a, fdv, f, r, u, w);
#pragma unroll
        for (int k = 0; k < 9; k++) { o[k].v[0] = f[k][0]; o[k].v[1] = f[k][1]; }
        rho4.v[0] = r[0]; rho4.v[1] = r[1]; ux4.v[0] = u[0]; ux4.v[1] = u[1]; uy4.v[0] = w[0]; uy4.v[1] = w[1];
    } else if constexpr (sizeof(T) == 4) {
        float r[S], u[S], w[S], s2[S];
        bool safe = true;
#pragma unroll
        for (int v = 0; v < S; v++) {
            float a[9];
            bool sv;
#pragma unroll
            for (int k = 0; k < 9; k++) a[k] = fin[k].v[v];
            collide_head(a, r[v], u[v], w[v], s2[v], sv);
            safe = safe && sv;
        }
        const bool fast = (FDV == 2) || (FDV == 1 && __ballot(!safe) == 0ULL);
#pragma unroll
        for (int v = 0; v < S; v++) {
            float a[9], f[9];
#pragma unroll
            for (int k = 0; k < 9; k++) a[k] = fin[k].v[v];
            if (FDV != 0 && fast) collide_tail<true, TWO>(a, fdv, r[v], u[v], w[v], s2[v], f);
            else collide_tail<false>(a, fdv, r[v], u[v], w[v], s2[v], f);
#pragma unroll
            for (int k = 0; k < 9; k++) o[k].v[v] = f[k];
            rho4.v[v] = r[v]; ux4.v[v] = u[v]; uy4.v[v] = w[v];
        }
    } else {
#pragma unroll
        for (int v = 0; v < S; v++) {
            T a[9], f[9], r, u, w;
#pragma unroll
            for (int k = 0; k < 9; k++) a[k] = fin[k].v[v];
            collide_t<T, FD>(a, fdv, tau, f, r, u, w);
#pragma unroll
            for (int k = 0; k < 9; k++) o[k].v[v] = f[k];
            rho4.v[v] = r; ux4.v[v] = u; uy4.v[v] = w;
        }
    }
}

template <typename T, int S, int FD, bool WANT_MACRO>
__device__ __forceinline__ void march_collide(const MV<T, S> (&fin)[9], const FastDiv &fdv, T tau, MV<T, S> (&G)[9], MV<T, S> (&mac)[3])
{
    MV<T, S> rho4, ux4, uy4;
    march_collide_sites<T, S, FD>(fin, fdv, tau, G, rho4, ux4, uy4);
    if (WANT_MACRO) { mac[0] = rho4; mac[1] = ux4; mac[2] = uy4; }
}

// rows 0 and NY-1 carry the far-field populations (html:314-322); only called for windows that hold one of them
template <typename T, int S, bool WANT_MACRO>
__device__ __forceinline__ void march_far_rows(int j0, int ny, T U0, const T (&feq0)[9], MV<T, S> (&G)[9], MV<T, S> (&mac)[3])
{
#pragma unroll
    for (int v = 0; v < S; v++) {
        const int j = j0 + v;
        const bool far = (j == 0) || (j == ny - 1);
#pragma unroll
        for (int k = 0; k < 9; k++) G[k].v[v] = far ? feq0[k] : G[k].v[v];
        if (WANT_MACRO) { mac[0].v[v] = far ? T(1) : mac[0].v[v]; mac[1].v[v] = far ? U0 : mac[1].v[v]; mac[2].v[v] = far ? T(0) : mac[2].v[v]; }
    }
}

// GENERAL tile, before the collision: half-way bounce-back (html:324-334) as an in-place select —
// direction k of a site takes the site's OWN population opp(k) where the bounce code has bit k-1 set.
// `own(k)` yields the vector of the sites' own populations of direction k (step 1: an aligned load,
// step 2: the step-1 vector of column c held in registers); one vector is live at a time.
template <typename T, int S, typename OWN>
__device__ __forceinline__ void march_bounce(MV<T, S> (&fin)[9], uint32_t code4, OWN own)
{
#pragma unroll
    for (int k = 1; k < 9; k++) {
        const MV<T, S> o = own(opp_of(k));
#pragma unroll
        for (int v = 0; v < S; v++) fin[k].v[v] = ((code4 >> (8 * v + k - 1)) & 1u) ? o.v[v] : fin[k].v[v];
    }
}

// GENERAL tile, the collision itself.  `fin` went through march_bounce with code 0xFF on solid sites, i.e. a solid
// site's fin[k] already IS its own population opp(k) — the value the reference stores for it (html:287-294) — so the
// solid select needs no further loads.  Reference order: solid, far field (html:314-322), interior (html:335-359).
template <typename T, int S, int FD, bool WANT_MACRO>
__device__ __forceinline__ void march_collide_general(const MV<T, S> (&fin)[9], uint32_t solid4, int j0, int ny, const FastDiv &fdv, T tau, T U0,
                                                      const T (&feq0)[9], MV<T, S> (&G)[9], MV<T, S> (&mac)[3])
{
    MV<T, S> o[9], rho4, ux4, uy4;
    march_collide_sites<T, S, FD>(fin, fdv, tau, o, rho4, ux4, uy4);
#pragma unroll
    for (int v = 0; v < S; v++) {
        const bool solid = ((solid4 >> (8 * v)) & 0xffu) != 0;
        const int j = j0 + v;
        const bool far = (j == 0) || (j == ny - 1);
#pragma unroll
        for (int k = 0; k < 9; k++) G[k].v[v] = solid ? fin[k].v[v] : (far ? feq0[k] : o[k].v[v]);
        if (WANT_MACRO) {
            mac[0].v[v] = (solid || far) ? T(1) : rho4.v[v];
            mac[1].v[v] = solid ? T(0) : (far ? U0 : ux4.v[v]);
            mac[2].v[v] = (solid || far) ? T(0) : uy4.v[v];
        }
    }
}

// after the collision: solid sites carry their own populations reversed (html:287-294), macro (1,0,0)
template <typename T, int S, bool WANT_MACRO, typename OWN>
__device__ __forceinline__ void march_solid(MV<T, S> (&G)[9], MV<T, S> (&mac)[3], uint32_t solid4, OWN own)
{
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const MV<T, S> o = own(opp_of(k));
#pragma unroll
        for (int v = 0; v < S; v++) G[k].v[v] = ((solid4 >> (8 * v)) & 0xffu) ? o.v[v] : G[k].v[v];
    }
    if (WANT_MACRO) {
#pragma unroll
        for (int v = 0; v < S; v++) {
            const bool solid = ((solid4 >> (8 * v)) & 0xffu) != 0;
            mac[0].v[v] = solid ? T(1) : mac[0].v[v]; mac[1].v[v] = solid ? T(0) : mac[1].v[v]; mac[2].v[v] = solid ? T(0) : mac[2].v[v];
        }
    }
}

// outlet column (html:301-312): macro = moments of the copied populations, not clamped
template <typename T, int S>
__device__ __forceinline__ void march_outlet_macro(const MV<T, S> (&q9)[9], MV<T, S> (&mac)[3])
{
#pragma unroll
    for (int v = 0; v < S; v++) {
        T q[9], rho, ux, uy;
#pragma unroll
        for (int k = 0; k < 9; k++) q[k] = q9[k].v[v];
        moments(q, rho, ux, uy);
        mac[0].v[v] = rho; mac[1].v[v] = ux; mac[2].v[v] = uy;
    }
}

// store the window's rows of column `col`.  No branch: lanes beyond the last row carry an out-of-range offset and
// the buffer range check drops their stores.  (A divergent `if (row < NY)` around the stores makes hipcc's waitcnt
// pass merge the "stored" and "not stored" paths and drain vmcnt to 0 at the loop tail — every iteration then waits
// for its own nine stores to complete before the next one starts.)
template <bool EMIT, typename T, int S>
__device__ __forceinline__ void march_store(const MarchAddr<T, S> &a, int col, const MV<T, S> (&out)[9], const MV<T, S> (&mac)[3])
{
    typename RawOf<S * sizeof(T)>::type d[12];
#pragma unroll
    for (int k = 0; k < 9; k++) d[k] = bstore<T, S>(a.rd, a.voff_st, lat_off(a, k, col, 0), out[k]);
    if (EMIT) {
        const unsigned mo = (unsigned)col * a.pitch4;
#pragma unroll
        for (int q = 0; q < 3; q++) d[9 + q] = bstore<T, S>(a.rm, a.voff_st, (unsigned)q * a.mp4 + mo, mac[q]);
    }
    store_data_fence(d, EMIT ? 12 : 9);
    // The two rows on either side of the window seams go, once more, into the seam buffer S (lane 0 holds rows 0,1 — above
    // seam w; lane 63 the window's last two rows — below seam w+1).  They pass through LDS so that 12 lanes write each half
    // record as contiguous, sector-aligned bytes (whole memory sectors: two-lane stores straight from lanes 0 / 63 cost more
    // than the table saves — partial sectors are read-modify-written).  Software-pipelined: this call only STAGES the
    // values (one LDS write per direction, every lane, no branch: lanes 1..62 hit a scratch slot); seam_fetch() at the
    // top of the next iteration reads them back transposed and seam_flush() stores them beside that iteration's stores,
    // so no LDS latency is exposed.  The next pass builds its halo table from S with coalesced loads
    // (k_halo_from_seams) instead of a gather (k_halo_rows).
    typedef T t2 __attribute__((ext_vector_type(2)));
    const bool top = a.lane == 63;
#pragma unroll
    for (int k = 0; k < 9; k++)
        *reinterpret_cast<t2 *>(a.lds_w + 2 * k) = t2{top ? out[k].v[S - 2] : out[k].v[0], top ? out[k].v[S - 1] : out[k].v[1]};
}

// seam values staged by the previous march_store, transposed: lane k < 9 gets direction k's pair of rows
template <typename T> struct SeamPair { T below[2], above[2]; };
template <typename T, int S>
__device__ __forceinline__ SeamPair<T> seam_fetch(const MarchAddr<T, S> &a)
{
    typedef T t2 __attribute__((ext_vector_type(2)));
    SeamPair<T> r;
    const t2 b = *reinterpret_cast<const t2 *>(a.lds_r);          // the window's last two rows (staged by lane 63)
    const t2 t = *reinterpret_cast<const t2 *>(a.lds_r + 24);     // rows 0,1                   (staged by lane 0)
    r.below[0] = b[0]; r.below[1] = b[1]; r.above[0] = t[0]; r.above[1] = t[1];
    return r;
}
template <typename T, int S>
__device__ __forceinline__ void seam_flush(const MarchAddr<T, S> &a, int col, const SeamPair<T> &r)
{
    const unsigned so = (unsigned)(col + 1) * (unsigned)(48 * sizeof(T));
    const auto d0 = bstore2<T>(a.rseam, a.voff_hi, so, r.below[0], r.below[1]);        // -> seam w+1, half 0
    const auto d1 = bstore2<T>(a.rseam, a.voff_lo, so, r.above[0], r.above[1]);        // -> seam w,   half 1
    store_data_fence2(d0, d1);
}

// wave-uniform value of lane l of a halo-table word
template <typename T> __device__ __forceinline__ T readlane_t(T x, int l)
{
    if constexpr (sizeof(T) == 4) return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x), l));
    else {
        const unsigned lo = __builtin_amdgcn_readlane((unsigned)__double2loint(x), l), hi = __builtin_amdgcn_readlane((unsigned)__double2hiint(x), l);
        return __hiloint2double((int)hi, (int)lo);
    }
}
template <typename T> __device__ __forceinline__ T halo_load(__amdgpu_buffer_rsrc_t rh, unsigned hoff, unsigned soff)
{
    if constexpr (sizeof(T) == 4) return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rh, hoff, soff, 0));
    else {
        const u2v x = __builtin_amdgcn_raw_buffer_load_b64(rh, hoff, soff, 0);
        return __hiloint2double((int)x.y, (int)x.x);
    }
}

// march_load_aligned's inputs of one column -> the streamed (pulled) inputs: populations 2,5,6 come from one row below, 4,7,8 from one row above;
// `hv` = the column's halo words, level 0 in lanes 12..14 (from below) and 60..62 (from above)
template <bool OVL = false, typename T, int S>
__device__ __forceinline__ void march_align_in(MV<T, S> (&in)[9], int lane, T hv)
{
    in[2] = m_below_x<12, OVL>(in[2], hv); in[5] = m_below_x<13, OVL>(in[5], hv); in[6] = m_below_x<14, OVL>(in[6], hv);
    in[4] = m_above_x<12, OVL>(in[4], hv); in[7] = m_above_x<13, OVL>(in[7], hv); in[8] = m_above_x<14, OVL>(in[8], hv);
}
// a column's halo line — none for overlapping windows
template <bool OVL, typename T> __device__ __forceinline__ T halo_load_x(__amdgpu_buffer_rsrc_t rh, unsigned hoff, unsigned soff)
{
    if constexpr (OVL) return T(0);
    else return halo_load<T>(rh, hoff, soff);
}

// ---- a general column's OWN populations, prefetched into LDS (round 5) ----
// Step 1 of a column that touches the body needs the sites' own populations of all directions — the bounced values of half-way bounce-back
// (html:324-334: fin[k] = f[opp(k)] of the site itself where the upstream site is solid) and the reversed populations of solid sites
// (html:287-294).  Loaded where they are used they put a full trip to memory behind a scalar branch in the middle of an iteration: on the slab
// over the thick part of the body these nine loads ALONE were 20 % of the pass (profiles/r05_g_own_loads.txt: 21.1 -> 16.7 us per step with the
// loads stubbed out), and they are why a body column cost three plain ones.  There is no register to prefetch them into (242-253 VGPRs), so they
// go to LDS instead: at the top of the iteration BEFORE the one that needs them, ahead of that iteration's own prefetch, as dword-per-lane LDS-DMA
// loads (`buffer_load_dword ... lds`: lane i's dword lands at M0 + 4 i, so two instructions 256 bytes apart lay the window's 64 S sizeof(T) = 512
// bytes of one population down contiguously and every lane reads its S rows back with one ds_read_b64 — layout checked by tools/kldsdma.hip).
// Loads return in order, so the wait for the prefetched column at the end of that iteration covers them (and hipcc's wait-count pass, which knows
// that these loads write LDS, puts an s_waitcnt vmcnt(n) in front of the reads where it cannot see that); two buffers per wave take turns.  The buffers live in the LDS of the chain blocks' hand-over slots, which a
// workgroup of solo units does not use.
static constexpr int OWN_LDS_BYTES = 9 * 512;      // one buffer: nine populations x 512 bytes
typedef __attribute__((address_space(3))) void *lds_void_p;
template <typename T, int S, bool OVL = false>
__device__ __forceinline__ void own_prefetch(const MarchAddr<T, S> &a, int col, char *buf)
{
    static_assert(64 * S * sizeof(T) == 512, "one population of a window's column is 512 bytes");
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const unsigned soff = lat_off(a, k, col, 0) - (OVL ? 256u : 0u);      // (OVL: a.voff_dma is kept 256 bytes high, k_march3)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a.rs, (lds_void_p)(buf + 512 * k), 4, a.voff_dma, soff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a.rs, (lds_void_p)(buf + 512 * k), 4, a.voff_dma, soff, 256, 0);
    }
}
template <typename T, int S>
__device__ __forceinline__ MV<T, S> own_read(const char *buf, int k, int lane)
{
    MV<T, S> o;
    const u2v x = *reinterpret_cast<const u2v *>(buf + 512 * k + 8 * lane);
    __builtin_memcpy(&o, &x, 8);
    return o;
}

// Step 1 of column x -> G (all nine directions).  `in` holds the streamed inputs of column x (modified in place).
// OWN_LDS: the column's own populations (general columns only) are in the LDS buffer `own_buf`, requested by own_prefetch — one iteration ago in the
// units' loops, just now for a unit's first columns; false (the two-step kernel): they are loaded here.
// `pre` (optional): the column's solid flags and bounce codes {solid4, code4}, fetched ahead by the caller (SiteBytes below)
template <bool BODY, int FD, typename T, int S, bool OWN_LDS = false>
__device__ __forceinline__ void march_step1(const MarchParams<T> &p, const MarchAddr<T, S> &a, int x, int j0, bool far_win, bool nonfast, bool allsolid,
                                            const T (&feq0)[9], MV<T, S> (&in)[9], MV<T, S> (&G)[9], const uint32_t *pre = nullptr, const char *own_buf = nullptr)
{
    MV<T, S> mac[3];
    if (BODY) {
        const Geom &g = p.g;
        const int gi = x + g.gi0;
#ifdef WT_EXP_NO_OWN       // timing experiment (WRONG results): what do the stage-1 loads of a general column's own populations cost?
        auto own = [&](int k) { return in[k]; };
#else
        auto own = [&](int k) {
            if constexpr (OWN_LDS) return own_read<T, S>(own_buf, k, a.lane);
            else return bload<T, S>(a.rs, a.voff, lat_off(a, k, x, 0));
        };
#endif
        if (__builtin_expect(gi <= 0 || gi >= g.nx_g - 1 || nonfast, 0)) {
            // rare paths (scalar branches): inlet / outlet columns, body surface, body interior
            uint32_t solid4 = 0, code4 = 0;
            if (nonfast) {
                if (pre) { solid4 = pre[0]; code4 = pre[1]; }
                else {
                    solid4 = load_site_bytes<S>(p.mask + (long)(x + 1) * g.pitch + site_row<S, (FD & MARCH_FD_OVL) != 0>(j0, g.ny));
                    code4 = load_site_bytes<S>(p.bcode + (long)x * g.pitch + site_row<S, (FD & MARCH_FD_OVL) != 0>(j0, g.ny));
                }
            }
            const bool any_solid = __ballot(solid4 != 0) != 0ULL;
            if (gi <= 0) {
#pragma unroll
                for (int k = 0; k < 9; k++) G[k] = mv_splat<T, S>(feq0[k]);
            } else if (gi >= g.nx_g - 1) {
#pragma unroll
                for (int k = 0; k < 9; k++) G[k] = bload<T, S>(a.rs, a.voff, lat_off(a, k, x - 1, 0));
            } else if (allsolid) {
#pragma unroll
                for (int k = 0; k < 9; k++) G[k] = own(opp_of(k));
                return;
            } else {
                march_bounce<T, S>(in, code4, own);
                march_collide_general<T, S, FD, false>(in, solid4, j0, g.ny, p.fdv, p.tau, p.U0, feq0, G, mac);
                return;
            }
            if (any_solid) march_solid<T, S, false>(G, mac, solid4, own);      // inlet / outlet columns with solid sites
            return;
        }
    }
    march_collide<T, S, FD, false>(in, p.fdv, p.tau, G, mac);
    if (far_win) march_far_rows<T, S, false>(j0, p.g.ny, p.U0, feq0, G, mac);
}

// One unit: marched columns [ia, ib) of window w.  BODY = false: the unit's footprint is plain interior fluid — no
// class tests, no mask, no inlet / outlet logic, fewer live registers (no spills); BODY = true: everything.
template <bool BODY, bool EMIT, int FD, typename T, int S>
__device__ __forceinline__ void march_unit(const MarchParams<T> &p, MarchAddr<T, S> &a, __amdgpu_buffer_rsrc_t rh, unsigned hoff, int ia, int ib, int uflags,
                                           int j0, int lane, bool far_win, unsigned long long nonfast_m, unsigned long long solid_m,
                                           const T (&feq0)[9])
{
    typedef MV<T, S> V;
    const Geom &g = p.g;
    constexpr unsigned HREC = 8 * sizeof(T);     // bytes of one halo-table record
#define NONFAST(x) (BODY && ((nonfast_m >> ((x) - ia + 1)) & 1ULL) != 0)
#define ALLSOLID(x) (BODY && ((solid_m >> ((x) - ia + 1)) & 1ULL) != 0)
#define STEP1(x, in, G) march_step1<BODY, FD, T, S>(p, a, (x), j0, far_win, NONFAST(x), ALLSOLID(x), feq0, in, G)

    V G158m[3];                  // step-1 populations 1,5,8 of column c-1
    V Gc[9];                     // step-1 populations of column c (BODY = false: only 0,2,4 and 1,5,8 stay live)
    V in[9], G[9], mac[3];
    if (!BODY || ia + g.gi0 > 0) {        // column ia-1 exists (ia = 0 on the inlet side: its step 2 is the far field)
        march_load_stream(a, ia - 1, in);
        STEP1(ia - 1, in, G);
        G158m[0] = G[1]; G158m[1] = G[5]; G158m[2] = G[8];
    } else {
        G158m[0] = mv_splat<T, S>(feq0[1]); G158m[1] = mv_splat<T, S>(feq0[5]); G158m[2] = mv_splat<T, S>(feq0[8]);
    }
    march_load_stream(a, ia, in);
    STEP1(ia, in, Gc);
    march_load_stream(a, ia + 1, in);
    T hv = halo_load<T>(rh, hoff, (unsigned)ia * HREC);
    int seam_col = -1;           // column whose seam rows are staged in LDS (-1: none yet; the flush then lands on the pad record)
#pragma unroll 1
    for (int c = ia; c < ib; c++) {
        V nxt[9];
        march_load_stream(a, (c + 2 <= ib) ? c + 2 : c + 1, nxt);             // prefetch (last one: harmless re-load)
        const T hv_next = halo_load<T>(rh, hoff, (unsigned)(c + 1) * HREC);
        const SeamPair<T> sp = seam_fetch(a);                                 // staged by the previous iteration's march_store
        STEP1(c + 1, in, G);                                                  // step 1 of column c+1
        const T hb2 = readlane_t(hv, 0), hb5 = readlane_t(hv, 1), hb6 = readlane_t(hv, 2), ha4 = readlane_t(hv, 3), ha7 = readlane_t(hv, 4),
                ha8 = readlane_t(hv, 5);
        // ---- step 2 of column c
        V fin[9], out[9];
        fin[0] = Gc[0]; fin[1] = G158m[0]; fin[3] = G[3];
        fin[2] = m_below(Gc[2], lane, hb2); fin[5] = m_below(G158m[1], lane, hb5); fin[6] = m_below(G[6], lane, hb6);
        fin[4] = m_above(Gc[4], lane, ha4); fin[8] = m_above(G158m[2], lane, ha8); fin[7] = m_above(G[7], lane, ha7);
        bool plain = true;
        if (BODY) {
            const int gi = c + g.gi0;
            const bool nf = NONFAST(c);
            if (__builtin_expect(gi <= 0 || nf, 0)) {
                plain = false;
                auto ownc = [&](int k) { return Gc[k]; };
                uint32_t solid4 = 0, code4 = 0;
                if (nf) {
                    solid4 = load_site_bytes<S>(p.mask + (long)(c + 1) * g.pitch + j0);
                    code4 = load_site_bytes<S>(p.bcode + (long)c * g.pitch + j0);
                }
                const bool any_solid = __ballot(solid4 != 0) != 0ULL;
                if (gi <= 0) {
#pragma unroll
                    for (int k = 0; k < 9; k++) out[k] = mv_splat<T, S>(feq0[k]);
                    if (EMIT) { mac[0] = mv_splat<T, S>(T(1)); mac[1] = mv_splat<T, S>(p.U0); mac[2] = mv_splat<T, S>(T(0)); }
                } else if (ALLSOLID(c)) {
#pragma unroll
                    for (int k = 0; k < 9; k++) out[k] = Gc[k];        // every site is overwritten by march_solid below
                    if (EMIT) { mac[0] = mv_splat<T, S>(T(1)); mac[1] = mv_splat<T, S>(T(0)); mac[2] = mv_splat<T, S>(T(0)); }
                } else {
                    march_bounce<T, S>(fin, code4, ownc);
                    march_collide_general<T, S, FD, EMIT>(fin, solid4, j0, g.ny, p.fdv, p.tau, p.U0, feq0, out, mac);
                }
                if (any_solid && (gi <= 0 || ALLSOLID(c))) march_solid<T, S, EMIT>(out, mac, solid4, ownc);
            }
        }
        if (plain) {
            march_collide<T, S, FD, EMIT>(fin, p.fdv, p.tau, out, mac);
            if (far_win) march_far_rows<T, S, EMIT>(j0, g.ny, p.U0, feq0, out, mac);
        }
        march_store<EMIT>(a, c, out, mac);
        seam_flush(a, seam_col, sp);
        seam_col = c;
        if (BODY && __builtin_expect((uflags & MU_OUTLET_AFTER) && c + 1 == ib, 0)) {
            // outlet column NX-1 (html:301-312): its step-2 value is the step-1 state of column NX-2 (= Gc), its own
            // step-1 state (solid sites only) is G
            uint32_t solid4 = 0;
            if (NONFAST(c + 1)) solid4 = load_site_bytes<S>(p.mask + (long)(c + 2) * g.pitch + j0);
            auto ownp = [&](int k) { return G[k]; };
#pragma unroll
            for (int k = 0; k < 9; k++) out[k] = Gc[k];
            if (EMIT) march_outlet_macro(Gc, mac);
            if (__ballot(solid4 != 0) != 0ULL) march_solid<T, S, EMIT>(out, mac, solid4, ownp);
            seam_flush(a, seam_col, seam_fetch(a));          // column c's seam rows, before the staging area is reused
            march_store<EMIT>(a, c + 1, out, mac);
            seam_col = c + 1;
        }
        G158m[0] = Gc[1]; G158m[1] = Gc[5]; G158m[2] = Gc[8];
#pragma unroll
        for (int k = 0; k < 9; k++) { Gc[k] = G[k]; in[k] = nxt[k]; }
        hv = hv_next;
    }
    seam_flush(a, seam_col, seam_fetch(a));
#undef NONFAST
#undef ALLSOLID
#undef STEP1
}

template <typename T, int S, bool EMIT, int FD>
__global__ __launch_bounds__(256, 2) void k_march(MarchParams<T> p)
{
    constexpr int WIN = 64 * S;
    constexpr unsigned EB = sizeof(T);
    const Geom &g = p.g;
    const int lane = threadIdx.x & 63;
    int u = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (u >= p.nunits) return;
    if (p.rev & 1) u = p.nunits - 1 - u;
    const MarchUnit un = p.units[u];
    const int ia = __builtin_amdgcn_readfirstlane(un.ia), ib = __builtin_amdgcn_readfirstlane(un.ib);
    const int w = __builtin_amdgcn_readfirstlane(un.w), uflags = __builtin_amdgcn_readfirstlane(un.flags);
    if (ib <= ia) return;                                          // padding unit (keeps the block -> XCD pattern of the list)
    const int row0 = w * WIN;
    const int j0 = row0 + lane * S;
    const bool far_win = (w == 0) || (row0 + WIN >= g.ny);         // the window holds row 0 or row NY-1
    MarchAddr<T, S> a;
    a.rs = march_rsrc(p.fs, p.lat_bytes);
    a.rd = march_rsrc(p.fd, p.lat_bytes);
    a.rm = march_rsrc(p.macro, (unsigned)(3u * (unsigned)g.nxl * (unsigned)g.pitch * EB));
    a.voff = (unsigned)((j0 < g.ny) ? j0 : row0) * EB;           // lanes beyond the last row re-read the window's first rows (cached) and store nothing
    a.voff_st = (j0 < g.ny) ? (unsigned)j0 * EB : p.lat_bytes;    // >= num_records of both the lattice and the macro buffer
    a.P4 = (unsigned)g.plane * EB; a.pitch4 = (unsigned)g.pitch * EB; a.mp4 = (unsigned)g.nxl * (unsigned)g.pitch * EB;
    a.own_lds = nullptr; a.voff_dma = 0;                          // (the LDS prefetch of a general column's own populations belongs to k_march3)
    // halo table: lanes 0..5 fetch, for step 2 of column c, {G2(c), G5(c-1), G6(c+1)} of the row below the window
    // (seam w) and {G4(c), G7(c+1), G8(c-1)} of the row above it (seam w+1); record = 8 elements per (seam, column)
    const __amdgpu_buffer_rsrc_t rh = march_rsrc(p.halo, (unsigned)((unsigned)(p.nwin_total + 1) * (unsigned)(g.nxl + 2) * 8u * EB));
    unsigned hoff;
    {
        const int hl = lane < 6 ? lane : 0;
        const int dx = (hl == 1 || hl == 5) ? -1 : ((hl == 2 || hl == 4) ? 1 : 0);
        const int slot = hl < 3 ? hl : hl + 1;
        const int seam = hl < 3 ? w : w + 1;
        hoff = (unsigned)((seam * (g.nxl + 2) + 1 + dx) * 8 + slot) * EB;
    }
    {
        // S record of (seam b, column x) = 48 elements: half 0 = rows WIN*b-2, WIN*b-1 (written by window b-1), half 1 = rows
        // WIN*b, WIN*b+1 (written by window b); a half = 12 slots of 2 elements, slot k < 9 = direction k
        // per wave: below[24] (lane 63 stages pairs 0..8, slots 9..11 stay zero), above[24] (lane 0), then a scratch area
        // the other 62 lanes stage into (an LDS write per lane is cheaper than a branch around two-lane writes)
        __shared__ T seam_lds[4][48 + 160];
        T *wl = &seam_lds[threadIdx.x >> 6][0];
        a.lane = lane;
        a.lds_w = lane == 63 ? wl : (lane == 0 ? wl + 24 : wl + 48 + 2 * lane);
        a.lds_r = wl + 2 * (lane < 12 ? lane : 0);
        if (lane < 48) wl[lane] = T(0);
        const unsigned sbytes = (unsigned)(p.nwin_total + 1) * (unsigned)(g.nxl + 2) * 48u * EB;
        a.rseam = march_rsrc(p.seams, sbytes);
        const unsigned rec = (unsigned)(g.nxl + 2) * 48u * EB;
        a.voff_lo = lane < 12 ? (unsigned)w * rec + 24u * EB + (unsigned)lane * 2u * EB : sbytes;
        a.voff_hi = lane < 12 ? (unsigned)(w + 1) * rec + (unsigned)lane * 2u * EB : sbytes;
    }
    T feq0[9];
    feq_all<T>(T(1), p.U0, T(0), feq0);                           // far-field populations (html:314-322)

    // classes of columns ia-1 .. ib (lane l <-> column ia-1+l): two 64-bit scalars
    unsigned long long nonfast_m, solid_m;
    {
        const int n = ib - ia + 2;
        uint8_t cls = WC_FAST;
        if (lane < n) cls = p.wcls[(long)w * (g.nxl + 2) + ia + lane];
        nonfast_m = __ballot(cls != WC_FAST);
        solid_m = __ballot(cls == WC_SOLID);
    }
    // a unit whose whole footprint (columns ia-1 .. ib) is plain interior fluid takes the lean loop
    const bool lean = nonfast_m == 0ULL && ia + g.gi0 >= 2 && ib + g.gi0 <= g.nx_g - 2 && !(uflags & MU_OUTLET_AFTER) && !(p.rev & 2);
    if (lean) march_unit<false, EMIT, FD, T, S>(p, a, rh, hoff, ia, ib, uflags, j0, lane, far_win, 0ULL, 0ULL, feq0);
    else march_unit<true, EMIT, FD, T, S>(p, a, rh, hoff, ia, ib, uflags, j0, lane, far_win, nonfast_m, solid_m, feq0);
}

// ------------------------------------------------------------------------------------------------
// host side: the unit lists of one mask
// ------------------------------------------------------------------------------------------------
// Marched column range of a handle: global edges are part of the march (inlet column 0; outlet column
// emitted with NX-2), local slab edges are not (their ghost columns lose two columns of validity per pass).
struct MarchRange { int i_begin, i_end, outlet_after; };
static inline MarchRange march_range(const Geom &g)
{
    MarchRange r;
    r.i_begin = (g.gi0 == 0) ? 0 : 1;
    r.i_end = g.nxl - 1;
    r.outlet_after = (g.gi0 + g.nxl == g.nx_g) ? 1 : 0;
    return r;
}

struct MarchPlan {
    std::vector<MarchUnit> units;
    int nwin = 0;
    int chunk = 0;       // columns of a typical unit
};

// wcls: host copy of the window-tile classes [nwin][nxl+2].  All units of a launch run side by side in whole
// "rounds" of `slots` resident waves and take about equally long, so the launch time is rounds x unit time:
// a unit count just above a multiple of `slots` wastes most of a round (measured on 4096^2: 4096 units 126 us
// per step, 4298 units 162 us).  The marched columns of every window are therefore cut into units of equal
// COST — a FAST column costs 1, any other column 1 + alpha (its step 1 waits for nine more loads) — such that
// the total is at most `target_units` (a multiple of `slots` chosen by the caller), or, when max_cost > 0, into
// units of at most max_cost (tests, experiments).
// min_last: least number of marched columns of a window's LAST unit (a four-step pass needs 2: the unit before the outlet unit
// must end two columns short of the outlet column); max_len: most columns of a unit; parts_multiple: the units of a window come in
// whole groups of that many where the window has at least one group (step_chain.hpp: blocks of four).
static inline MarchPlan build_march_plan(const uint8_t *wcls, const Geom &g, int win, long target_units, int max_cost = 0, double alpha = 1.0,
                                         const MarchRange *range = nullptr, int min_last = 1, int max_len = MARCH_MAX_CHUNK, int parts_multiple = 1)
{
    MarchPlan pl;
    const int nwin = march_nwin(g.ny, win), ld = g.nxl + 2;
    pl.nwin = nwin;
    const MarchRange r = range ? *range : march_range(g);
    const int ncol = r.i_end - r.i_begin;
    if (ncol <= 0) return pl;
    std::vector<double> wcost((size_t)nwin, 0.0);
    double total = 0.0;
    for (int w = 0; w < nwin; w++) {
        const uint8_t *c = wcls + (size_t)w * ld + 1;
        for (int x = r.i_begin; x < r.i_end; x++) wcost[w] += (c[x] == WC_FAST) ? 1.0 : 1.0 + alpha;
        total += wcost[w];
    }
    double target;       // cost of one unit
    if (max_cost > 0) target = (double)max_cost;
    else {
        if (target_units < nwin) target_units = nwin;
        target = total / (double)target_units;
    }
    if (target > (double)max_len) target = (double)max_len;      // a unit's class bytes must fit one wave
    if (target < 1.0) target = 1.0;
    pl.chunk = (int)(target + 0.5);
    for (int w = 0; w < nwin; w++) {
        const uint8_t *c = wcls + (size_t)w * ld + 1;
        // whole parts only, never more than wcost / target of them: the total stays <= target_units
        int parts = (int)(wcost[w] / target);
        if (max_cost > 0) parts = (int)((wcost[w] + target - 1e-9) / target);
        if (parts < 1) parts = 1;
        if (max_cost <= 0 && parts_multiple > 1 && parts >= parts_multiple && (ncol + parts - 1) / parts < max_len) parts -= parts % parts_multiple;
        const double per = wcost[w] / parts;
        int ia = r.i_begin, done = 0;
        double acc = 0.0;
        for (int x = r.i_begin; x < r.i_end; x++) {
            acc += (c[x] == WC_FAST) ? 1.0 : 1.0 + alpha;
            const bool last = (x + 1 == r.i_end);
            const bool tail_short = !last && r.i_end - (x + 1) < min_last;      // a cut here would leave too short a last unit
            if (last || (!tail_short && ((done + 1 < parts && acc >= per * (done + 1) - 1e-9) || x + 1 - ia >= max_len))) {
                pl.units.push_back(MarchUnit{ia, x + 1, w, (r.outlet_after && last) ? MU_OUTLET_AFTER : 0});
                ia = x + 1;
                done++;
            }
        }
    }
    // chunk-major order: the windows of one chunk are neighbours in the list (adjacent waves read adjacent kilobytes)
    std::stable_sort(pl.units.begin(), pl.units.end(), [](const MarchUnit &x, const MarchUnit &y) { return x.ia != y.ia ? x.ia < y.ia : x.w < y.w; });
    return pl;
}

// The same cut by TIME instead of by owned columns.  All units of a launch are resident together (whole rounds), so the launch takes as
// long as its slowest unit, and a unit's time is the cost of every column it ITERATES over — its own and the `over` columns it recomputes
// for its neighbours' sake (pipeline fill and drain: 2.7 iterations of a three-step pass, 4.5 of a four-step one), which in the body zone
// are body columns too (per-unit clocks, tools/unit_clocks.py: with the cut by owned columns a two-column body unit of the bench mask took
// 1.55 x as long as a ten-column plain one and set the pace of the whole launch).  time(ia, ib) = C(ib + over / 2) - C(ia - over / 2) with
// C the running cost of the window's columns (a FAST column 1, a column of an all-solid tile 1 + alpha_solid, any other 1 + alpha, nothing beyond the tunnel's ends) and `tail` more
// for the unit that also emits the outlet column; the smallest t for which every window cut greedily into units of time <= t gives at
// most target_units in total is found by bisection.
static inline MarchPlan build_march_plan_timed(const uint8_t *wcls, const Geom &g, int win, long target_units, double alpha, const MarchRange &r, int min_last,
                                               int max_len, double over, double tail, double alpha_solid, const float *colw = nullptr)
{
    MarchPlan pl;
    const int nwin = march_nwin(g.ny, win), ld = g.nxl + 2;
    pl.nwin = nwin;
    const int ncol = r.i_end - r.i_begin;
    if (ncol <= 0) return pl;
    if (target_units < nwin) target_units = nwin;
    const int E = (int)(over / 2) + 2;                         // columns beyond the marched range that enter a unit's time
    const int n = ncol + 2 * E;
    std::vector<double> C((size_t)nwin * (n + 1), 0.0);        // C[w][k] = cost of columns i_begin - E .. i_begin - E + k - 1
    for (int w = 0; w < nwin; w++) {
        const uint8_t *c = wcls + (size_t)w * ld + 1;
        double *Cw = &C[(size_t)w * (n + 1)];
        for (int k = 0; k < n; k++) {
            const int x = r.i_begin - E + k, gi = x + g.gi0;
            double cost = 0.0;
            if (gi >= 0 && gi < g.nx_g) cost = (x >= -1 && x <= g.nxl && c[x] != WC_FAST) ? 1.0 + (c[x] == WC_SOLID ? alpha_solid : alpha) : 1.0;
            if (colw && x >= -1 && x <= g.nxl) cost *= colw[(size_t)w * ld + x + 1];      // measured correction (tune_fuse_plan)
            Cw[k + 1] = Cw[k] + cost;
        }
    }
    auto Cf = [&](const double *Cw, double x) {                // running cost up to (fractional) column x
        double k = x - (r.i_begin - E);
        if (k < 0) k = 0;
        if (k > n) k = n;
        const int k0 = (int)k;
        return k0 >= n ? Cw[n] : Cw[k0] + (k - k0) * (Cw[k0 + 1] - Cw[k0]);
    };
    auto unit_time = [&](const double *Cw, int ia, int ib) {
        return Cf(Cw, ib + over / 2) - Cf(Cw, ia - over / 2) + ((r.outlet_after && ib == r.i_end) ? tail : 0.0);
    };
    // cut one window for a time limit t; returns the number of units (and the cuts when `out` is given)
    auto cut = [&](int w, double t, std::vector<MarchUnit> *out) {
        const double *Cw = &C[(size_t)w * (n + 1)];
        int ia = r.i_begin, count = 0;
        while (ia < r.i_end) {
            int ib = ia + 1;
            const int cap = std::min(r.i_end, ia + max_len);
            while (ib < cap && unit_time(Cw, ia, ib + 1) <= t) ib++;
            if (r.i_end - ib > 0 && r.i_end - ib < min_last) {     // too short a last unit: take it in, or leave it min_last columns
                if (r.i_end - min_last <= ia || (r.i_end - ia <= max_len + min_last - 1 && unit_time(Cw, ia, r.i_end) <= t)) ib = r.i_end;
                else ib = r.i_end - min_last;
            }
            if (out) out->push_back(MarchUnit{ia, ib, w, (r.outlet_after && ib == r.i_end) ? MU_OUTLET_AFTER : 0});
            ia = ib;
            count++;
        }
        return count;
    };
    auto total = [&](double t) { long s = 0; for (int w = 0; w < nwin; w++) s += cut(w, t, nullptr); return s; };
    double lo = 0.0, hi = (double)max_len * (1.0 + alpha) + over * (1.0 + alpha) + tail + 1.0;
    if (total(hi) > target_units) lo = hi;                     // the length cap forces more units than asked for: longest units
    else {
        for (int it = 0; it < 24; it++) {       // (t to 6e-8 of its range: unit counts change at discrete t)
            const double mid = 0.5 * (lo + hi);
            if (total(mid) <= target_units) hi = mid; else lo = mid;
        }
    }
    const double t = hi;
    pl.chunk = 0;
    for (int w = 0; w < nwin; w++) cut(w, t, &pl.units);
    long cols = 0;
    for (const MarchUnit &u : pl.units) cols += u.ib - u.ia;
    pl.chunk = pl.units.empty() ? 0 : (int)((double)cols / (double)pl.units.size() + 0.5);
    // chunk-major order: the windows of one chunk are neighbours in the list (adjacent waves read adjacent kilobytes)
    std::stable_sort(pl.units.begin(), pl.units.end(), [](const MarchUnit &x, const MarchUnit &y) { return x.ia != y.ia ? x.ia < y.ia : x.w < y.w; });
    return pl;
}

}  // namespace wt
