// step_chain.hpp — marching units that SHARE their edge columns instead of recomputing them.
//
// A solo unit of a T-step pass (step_march3.hpp) recomputes 2 (T-1) columns of its neighbours: level 1 of T-1 columns on either side,
// level 2 of T-2, ...  On a column slab of an 8-way split a unit has 8-9 columns of its own and a third of its work is that overlap
// (3 L + 8 stage executions for 3 L useful ones).  Here the four waves of a workgroup take four consecutive units of one window and
// march them in ALTERNATING directions,
//
//        <---- A0 ----|---- B0 ---->   <---- A1 ----|---- B1 ---->
//               start seam        end seam        start seam
//
// so that both units of a pair begin at their common seam, and the inner units B0 and A1 finish at theirs.  At a start seam each
// unit publishes, level by level, the three populations of its first column that move across the seam (LDS: data, then a flag the
// partner polls) and takes its partner's in place of the "column behind" it would otherwise have recomputed; at an end seam the same
// happens with the last column and the "column ahead".  No site is computed twice inside the block; only its two outer ends
// (A0's left, B1's right) still overlap with the neighbouring blocks.  Stage executions per unit: 3 L + 1.5 on average instead of
// 3 L + 8 (T = 3), 4 L + 3 instead of 4 L + 18 (T = 4).
//
// The fill and drain iterations are PEELED (an iteration template with a compile-time stage mask and hand-over hooks) rather than
// run on don't-care values; there is ONE workgroup barrier, behind the clearing of the flags at the start of the kernel.
//
// Chain blocks are plain interior fluid throughout (the planner checks the window-tile classes of the block's whole footprint and
// keeps it away from the inlet / outlet columns; every unit holds at least depth + 1 columns): no mask, no bounce codes, no class
// tests.  Everything else — body, inlet, outlet, short units, two-step passes — stays with the solo units of step_march3.hpp, in the
// same launch.  Units are cut by time and placed by build_chain_plan_timed (bottom of this file).
// Every site still goes through the arithmetic of k_step once per level: bit-identical results.
#pragma once
#include "step_march3.hpp"

namespace wt {

enum { MU_CHAIN = 2, MU_DIR_NEG = 4, MU_END_SHARED = 8 };      // MarchUnit::flags, beside MU_OUTLET_AFTER

template <typename T, int S, int DEPTH>
struct ChainLds {
    // [phase: 0 start seam, 1 end seam][level - 1][position of the unit in its block][population of the triple][lane]
    MV<T, S> x[2][DEPTH - 1][4][3][64];
    int flag[2][DEPTH - 1][4];           // 1 once the triple above has been written (each slot is written once per launch)
};

typedef __attribute__((address_space(3))) int lds_int_t;      // the hand-over flags live in LDS: say so (see chain_publish)
// Hand-over of a triple between two waves of a workgroup WITHOUT a barrier: the producer writes the data, then the flag (LDS operations of one
// wave complete in order; the wait in between makes that explicit); the consumer polls the flag just before the stage that needs the data —
// one stage or more after its partner published, so the poll almost always succeeds at once.  (A workgroup barrier per level cost 8 % of a
// three-step unit and 18 % of a four-step one on a 544-column slab: four waves on four SIMDs, each sharing its SIMD with another workgroup's
// wave, are never in step.)  Both waves are resident (same workgroup), so the poll cannot dead-lock.
template <typename V>
__device__ __forceinline__ void chain_publish(V (&slot)[3][64], int &flag, int lane, const V &a0, const V &a1, const V &a2)
{
    slot[0][lane] = a0;
    slot[1][lane] = a1;
    slot[2][lane] = a2;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // (through an LDS-qualified pointer: as a generic volatile access the flag came out as flat_store_dword sc0 sc1 + s_waitcnt vmcnt(0) — every
    //  hand-over drained the wave's prefetched column and its stores; the poll below likewise as flat_load_dword + vmcnt(0).  Round 5.)
    *(volatile lds_int_t *)&flag = 1;
}
// The poll is BOUNDED: on a well-formed plan the partner publishes within a stage or two — the library checks every plan
// it uploads (sanitize_chain_plan) — but a wave must never be able to spin for ever on a GPU other people share.  After CHAIN_POLL_LIMIT polls
// (seconds; a healthy hand-over takes microseconds) the unit raises the host-visible word `stuck`, goes on with whatever the slot holds and
// terminates normally; the host turns the word into WT_ERR_STATE at its next synchronisation (windtunnel.hip check_stuck).
static constexpr int CHAIN_POLL_LIMIT = 1 << 24;
template <typename V>
__device__ __forceinline__ void chain_receive(const V (&slot)[3][64], const int &flag, int lane, V (&r)[3], unsigned int *stuck)
{
#ifndef WT_CHAIN_NOBARRIER   // (timing experiment: no synchronisation at all — wrong results)
    if (__builtin_expect(__builtin_amdgcn_readfirstlane(*(const volatile lds_int_t *)&flag) == 0, 0)) {
        int polls = 0;
        while (__builtin_amdgcn_readfirstlane(*(const volatile lds_int_t *)&flag) == 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++polls > CHAIN_POLL_LIMIT) {
                if (lane == 0 && stuck) atomicOr(stuck, 1u);
                break;
            }
        }
    }
#endif
    asm volatile("" ::: "memory");
    r[0] = slot[0][lane];
    r[1] = slot[1][lane];
    r[2] = slot[2][lane];
}

// populations that move in the marching direction (they come from the column BEHIND) and against it (from the column AHEAD),
// in the order (straight, moving up, moving down)
template <int DIR> struct ChainPops;
template <> struct ChainPops<1> { static constexpr int B0 = 1, B1 = 5, B2 = 8, A0 = 3, A1 = 6, A2 = 7; };
template <> struct ChainPops<-1> { static constexpr int B0 = 3, B1 = 6, B2 = 7, A0 = 1, A1 = 5, A2 = 8; };

// one more application of STEP_FS in registers (plain interior fluid): level k+1 of column c from level k of the column behind
// (`mb`: its three populations moving forward), of column c itself (`Gc`) and of the column ahead (`ma`: three populations moving
// backward); `hv`, `lb` = column c's halo line and the level's first lane (as in march_stage)
template <int DIR, bool WANT_MACRO, int FD, int LB, typename T, int S>
__device__ __forceinline__ void chain_stage(const MarchParams<T> &p, int j0, int lane, bool far_win, const T (&feq0)[9], const MV<T, S> (&mb)[3],
                                            const MV<T, S> (&Gc)[9], const MV<T, S> (&ma)[3], T hv, MV<T, S> (&out)[9], MV<T, S> (&mac)[3])
{
    typedef MV<T, S> V3;
    typedef ChainPops<DIR> P;
    constexpr bool OVL = (FD & MARCH_FD_OVL) != 0;
    // the word of population k's row outside the window: slot 0 = 2 / 4 (own column), 1 = 5 / 7, 2 = 6 / 8 (step_march3.hpp, halo lines)
    constexpr int SB1 = P::B1 == 5 ? 1 : 2, SA1 = P::A1 == 5 ? 1 : 2;          // from below: populations 5, 6
    constexpr int SB2 = P::B2 == 7 ? 1 : 2, SA2 = P::A2 == 7 ? 1 : 2;          // from above: populations 7, 8
    V3 fin[9];
    fin[0] = Gc[0];
    fin[2] = m_below_x<LB, OVL>(Gc[2], hv);
    fin[4] = m_above_x<LB, OVL>(Gc[4], hv);
    fin[P::B0] = mb[0];
    fin[P::B1] = m_below_x<LB + SB1, OVL>(mb[1], hv);
    fin[P::B2] = m_above_x<LB + SB2, OVL>(mb[2], hv);
    fin[P::A0] = ma[0];
    fin[P::A1] = m_below_x<LB + SA1, OVL>(ma[1], hv);
    fin[P::A2] = m_above_x<LB + SA2, OVL>(ma[2], hv);
    march_collide<T, S, FD, WANT_MACRO>(fin, p.fdv, p.tau, out, mac);
    if (far_win) march_far_rows<T, S, WANT_MACRO>(j0, p.g.ny, p.U0, feq0, out, mac);
}

template <int DIR, int DEPTH, bool EMIT, int FD, typename T, int S>
struct ChainUnit {
    typedef MV<T, S> V3;
    typedef ChainPops<DIR> P;
    static constexpr unsigned HREC = M3_HL * sizeof(T);
    static constexpr int FULL = (1 << DEPTH) - 1;
    static constexpr bool OVLF = (FD & MARCH_FD_OVL) != 0;

    const MarchParams<T> &p;
    March3Addr<T, S> &m;
    const __amdgpu_buffer_rsrc_t &rh;
    const T (&feq0)[9];
    ChainLds<T, S, DEPTH> &lds;
    unsigned hoff;
    int j0, lane, pos;
    bool far_win;
    int x_lo, x_hi;              // columns whose streamed inputs may be requested (prefetches beyond the unit's last column re-load it)
    int seam_col;

    V3 in[9];                    // streamed inputs of the next level-1 column
    V3 sm[DEPTH - 1][3];         // level k (index k-1): forward-moving populations of the column two behind the front of that level
    V3 sc[DEPTH - 1][9];         //                      all nine of the column one behind it
    T hv[DEPTH];                 // halo lines of the columns this iteration's stages 1 .. DEPTH work on: hv[j] = column x - j DIR

    __device__ __forceinline__ ChainUnit(const MarchParams<T> &p_, March3Addr<T, S> &m_, const __amdgpu_buffer_rsrc_t &rh_, const T (&feq0_)[9],
                                         ChainLds<T, S, DEPTH> &lds_)
        : p(p_), m(m_), rh(rh_), feq0(feq0_), lds(lds_) {}

    __device__ __forceinline__ unsigned hcol(int c) const { return (unsigned)(c > 0 ? c : 0) * HREC; }
    __device__ __forceinline__ int clampx(int x) const { return x < x_lo ? x_lo : (x > x_hi ? x_hi : x); }

    // One iteration with the level-1 front at column x: stage 1 (bit 0 of MASK) = STEP_FS on the streamed inputs of column x; stage k
    // (bit k-1) = level k of column x - (k-1) DIR; the last stage stores.  Hand-over hooks (0 = none):
    //   RS: take the start-seam partner's level-RS triple as the column behind, just before stage RS + 1;
    //   PS: publish this unit's level-PS triple for the start-seam partner, right after stage PS;
    //   RE: take the end-seam partner's level-RE triple as the column ahead of stage RE + 1, the first stage of this (drain) iteration;
    //   PE: publish this unit's level-PE triple for the end-seam partner, right after stage PE.
    // stage K of an iteration (see iter): level K of column x - (K-1) DIR
    template <int K, int MASK, int RS, int PS, int RE, int PE>
    __device__ __forceinline__ void stage_k(V3 (&G)[DEPTH + 1][9], V3 (&mac)[3])
    {
        if constexpr (K <= DEPTH && ((MASK >> (K - 1)) & 1) != 0) {
            V3 ma[3];
            if constexpr (((MASK >> (K - 2)) & 1) != 0) { ma[0] = G[K - 1][P::A0]; ma[1] = G[K - 1][P::A1]; ma[2] = G[K - 1][P::A2]; }
            else chain_receive(lds.x[1][K - 2][pos ^ 3], lds.flag[1][K - 2][pos ^ 3], lane, ma, p.stuck);           // RE == K - 1 (positions 1 <-> 2)
            if constexpr (RS == K - 1) chain_receive(lds.x[0][K - 2][pos ^ 1], lds.flag[0][K - 2][pos ^ 1], lane, sm[K - 2], p.stuck);
            chain_stage<DIR, (K == DEPTH) && EMIT, FD, 4 * (K - 2)>(p, j0, lane, far_win, feq0, sm[K - 2], sc[K - 2], ma, hv[K - 1], G[K], mac);
            if constexpr (K < DEPTH && PS == K) chain_publish(lds.x[0][K - 1][pos], lds.flag[0][K - 1][pos], lane, G[K][P::A0], G[K][P::A1], G[K][P::A2]);
            if constexpr (K < DEPTH && PE == K) chain_publish(lds.x[1][K - 1][pos], lds.flag[1][K - 1][pos], lane, G[K][P::B0], G[K][P::B1], G[K][P::B2]);
        }
    }

    template <int MASK, int RS, int PS, int RE, int PE>
    __device__ __forceinline__ void iter(int x)
    {
        const MarchAddr<T, S> &a = m.a;
        constexpr bool S1 = (MASK & 1) != 0, LAST = (MASK >> (DEPTH - 1)) != 0;
        V3 nxt[9];
        if (S1) march_load_aligned(a, clampx(x + DIR), nxt);
        const T hvn = halo_load_x<OVLF, T>(rh, hoff, hcol(x + DIR));    // column x + DIR's halo line: the next iteration's stage 1, then handed on
        Seam3 sp;
        if (LAST) sp = seam3_fetch<OVLF>(m);
        V3 G[DEPTH + 1][9], mac[3];          // G[k] = level k computed in this iteration (G[DEPTH] = what is stored)
        if (S1) {
            march_align_in<OVLF>(in, lane, hv[0]);
            march_step1<false, FD, T, S>(p, a, x, j0, far_win, false, false, feq0, in, G[1]);
            if (PS == 1) chain_publish(lds.x[0][0][pos], lds.flag[0][0][pos], lane, G[1][P::A0], G[1][P::A1], G[1][P::A2]);
            if (PE == 1) chain_publish(lds.x[1][0][pos], lds.flag[1][0][pos], lane, G[1][P::B0], G[1][P::B1], G[1][P::B2]);
        }
        stage_k<2, MASK, RS, PS, RE, PE>(G, mac);
        stage_k<3, MASK, RS, PS, RE, PE>(G, mac);
        stage_k<4, MASK, RS, PS, RE, PE>(G, mac);
        static_assert(RE == 0 || (((MASK >> RE) & 1) && !((MASK >> (RE - 1)) & 1)), "RE names the level below the first stage of a drain iteration");
        if (LAST) {
            pin_after(G[DEPTH]);
            if (S1) wait_for_column(nxt, hvn);
            const int c = x - (DEPTH - 1) * DIR;
            march3_store<EMIT, OVLF>(m, a.voff_st, c, G[DEPTH], mac);
            seam3_flush<OVLF>(m, seam_col, sp);
            seam_col = c;
        }
#pragma unroll
        for (int k = DEPTH - 1; k > 0; k--) hv[k] = hv[k - 1];
        hv[0] = hvn;
#pragma unroll
        for (int k = 1; k < DEPTH; k++) {
            if (!((MASK >> (k - 1)) & 1)) continue;      // level k was not advanced: its carried columns stay
            sm[k - 1][0] = sc[k - 1][P::B0]; sm[k - 1][1] = sc[k - 1][P::B1]; sm[k - 1][2] = sc[k - 1][P::B2];
#pragma unroll
            for (int q = 0; q < 9; q++) sc[k - 1][q] = G[k][q];
        }
        if (S1) {
#pragma unroll
            for (int q = 0; q < 9; q++) in[q] = nxt[q];
        }
    }

    // marched columns [ia, ib), at least DEPTH + 1 of them; the unit starts at its seam with unit pos ^ 1 and ends, when end_shared, at its
    // seam with the other inner unit (positions 1 and 2), else on its own (DEPTH - 1 recomputed columns of the neighbouring block)
    __device__ __forceinline__ void run(int ia, int ib, bool end_shared)
    {
        const MarchAddr<T, S> &a = m.a;
        const int L = ib - ia;
        const int xf = DIR > 0 ? ia : ib - 1;                                 // first column
        const int p_end = end_shared ? L - 1 : L - 1 + (DEPTH - 1);           // last position whose level 1 is computed here
        x_lo = DIR > 0 ? ia : ib - 1 - p_end;
        x_hi = DIR > 0 ? ia + p_end : ib - 1;
        seam_col = -1;
#pragma unroll
        for (int k = 0; k < DEPTH - 1; k++) {
#pragma unroll
            for (int q = 0; q < 9; q++) sc[k][q] = mv_splat<T, S>(feq0[q]);
            sm[k][0] = sc[k][1]; sm[k][1] = sc[k][5]; sm[k][2] = sc[k][8];
        }
        march_load_aligned(a, xf, in);
#pragma unroll
        for (int k = 0; k < DEPTH; k++) hv[k] = halo_load_x<OVLF, T>(rh, hoff, hcol(xf - k * DIR));
        // ---- start seam: iteration q brings level q + 1 of the first column into being and publishes its backward-moving populations; the
        //      partner's forward-moving ones are taken one iteration later, just before the stage that needs them
        iter<1, 0, 1, 0, 0>(xf);
        iter<3, 1, 2, 0, 0>(xf + DIR);
        if constexpr (DEPTH == 3) {
            iter<7, 2, 0, 0, 0>(xf + 2 * DIR);
        } else {
            iter<7, 2, 3, 0, 0>(xf + 2 * DIR);
            iter<15, 3, 0, 0, 0>(xf + 3 * DIR);
        }
        // ---- the body of the unit
        const int q_last = end_shared ? L - 2 : p_end;
#pragma unroll 1
        for (int q = DEPTH; q <= q_last; q++) iter<FULL, 0, 0, 0, 0>(xf + q * DIR);
        // ---- end seam: the last column's forward-moving populations go out as soon as each level exists; the partner's backward-moving
        //      ones are taken at the head of the next (drain) iteration
        if (end_shared) {
            const int xl = xf + (L - 1) * DIR;
            iter<FULL, 0, 0, 0, 1>(xl);
            iter<(FULL & ~1), 0, 0, 1, 2>(xl + DIR);
            if constexpr (DEPTH == 3) {
                iter<(FULL & ~3), 0, 0, 2, 0>(xl + 2 * DIR);
            } else {
                iter<(FULL & ~3), 0, 0, 2, 3>(xl + 2 * DIR);
                iter<(FULL & ~7), 0, 0, 3, 0>(xl + 3 * DIR);
            }
        }
        seam3_flush<OVLF>(m, seam_col, seam3_fetch<OVLF>(m));
    }
};

// ------------------------------------------------------------------------------------------------
// the marching kernel: solo units (step_march3.hpp) and chain blocks in one launch
// ------------------------------------------------------------------------------------------------
// How long does every unit of a pass take?  Tuning passes (tune_fuse_plan in windtunnel.hip; tools/unit_clocks.py) hand a buffer over:
// [unit] = {start, end} in s_memtime ticks.  Ordinary passes carry a null pointer and pay one scalar read of the clock.
#ifdef WT_CLOCK_REALTIME      // diagnostic build (tools/unit_timeline.py): the constant 100 MHz counter all XCDs share, for a timeline of one launch
#define WT_UNIT_TICKS() __builtin_amdgcn_s_memrealtime()
#else
#define WT_UNIT_TICKS() __builtin_amdgcn_s_memtime()
#endif
struct UnitClock {
    unsigned long long *clk;
    int u, lane;
    unsigned long long t0;
    __device__ __forceinline__ UnitClock(unsigned long long *clk_, int u_, int lane_) : clk(clk_), u(u_), lane(lane_), t0(clk_ ? WT_UNIT_TICKS() : 0ULL) {}
    __device__ __forceinline__ ~UnitClock()
    {
        if (clk) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            const unsigned long long t1 = WT_UNIT_TICKS();
            if (lane == 0) { clk[2 * u] = t0; clk[2 * u + 1] = t1; }
        }
    }
};

template <typename T, int S, int DEPTH, bool EMIT, int FD>
__global__ __launch_bounds__(256, 2) void k_march3(MarchParams<T> p)
{
    constexpr int M3_WIN = 64 * S;
    constexpr unsigned EB = sizeof(T);
    const Geom &g = p.g;
    const int lane = threadIdx.x & 63;
    int u = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (u >= p.nunits) return;
    if (p.rev & 1) u = p.nunits - 1 - u;
    const MarchUnit un = p.units[u];
    const int ia = __builtin_amdgcn_readfirstlane(un.ia), ib = __builtin_amdgcn_readfirstlane(un.ib);
    const int w = __builtin_amdgcn_readfirstlane(un.w), uflags = __builtin_amdgcn_readfirstlane(un.flags);
    if (ib <= ia) return;
    UnitClock unit_clock(p.clk, u, lane);
    // Window w = rows row0 .. row0 + WIN - 1.  Windows that TILE the column (row0 = w WIN) take the rows beyond their seams, level by level, out of the
    // halo lines the halo kernel builds before every pass.  OVERLAPPING windows (the instantiations with MARCH_FD_OVL in FD: row0 = w (WIN - 8) - 4)
    // carry a margin of four rows on either side instead: a margin row loses one row of validity per level — whatever arrives from beyond the window
    // is don't-care —, so after up to four levels the WIN - 8 rows in between are exact, and only those are stored.  No halo lines, no halo kernel, no
    // seam buffer to keep valid, six data-parallel moves per stage instead of twelve; 128 / 120 of the arithmetic (fp32).  The host launches these
    // instantiations for a plan whose tables were built for such windows (windtunnel.hip build_fuse_plan: option window_overlap, p.win_stride).
    constexpr bool OVL = (FD & MARCH_FD_OVL) != 0;
    constexpr int STRIDE = OVL ? M3_WIN - 8 : M3_WIN, ROW_OFF = OVL ? -4 : 0;
    const int row0 = w * STRIDE + ROW_OFF;
    const int j0 = row0 + lane * S;
    const bool far_win = OVL ? (row0 <= 0 || row0 + M3_WIN >= g.ny) : ((w == 0) || (row0 + M3_WIN >= g.ny));
    // rows a lane owns (stores): tiling windows all of theirs that exist; overlapping windows those outside the two four-row margins
    const bool owns = OVL ? (j0 >= 0 && j0 < g.ny && lane >= 4 / S && lane < 64 - 4 / S) : (j0 < g.ny);
    March3Addr<T, S> m;
    MarchAddr<T, S> &a = m.a;
    a.rs = march_rsrc(p.fs, p.lat_bytes);
    a.rd = march_rsrc(p.fd, p.lat_bytes);
    a.rm = march_rsrc(p.macro, (unsigned)(3u * (unsigned)g.nxl * (unsigned)g.pitch * EB));
    if constexpr (OVL) a.voff = (unsigned)((j0 >= 0 && j0 < g.ny) ? j0 : (row0 > 0 ? row0 : 0)) * EB;      // (rows that do not exist: any that do — don't-care values)
    else a.voff = (unsigned)((j0 < g.ny) ? j0 : row0) * EB;
    a.voff_st = owns ? (unsigned)j0 * EB : p.lat_bytes;
    a.P4 = (unsigned)g.plane * EB; a.pitch4 = (unsigned)g.pitch * EB; a.mp4 = (unsigned)g.nxl * (unsigned)g.pitch * EB;
    a.lane = lane;
    // One LDS pool per workgroup: the hand-over slots of a chain block, or — a workgroup is a chain block or four solo units, never both — the
    // solo units' buffers for the own populations of general columns (own_prefetch, step_march.hpp): two per wave.
    // (a union, so that the chain path keeps addressing its slots and flags as members of a __shared__ object: through a reinterpreted char
    //  array the flag polls came out as FLAT loads with a vmcnt(0) wait behind them)
    union LdsPool { ChainLds<T, S, (DEPTH >= 3 ? DEPTH : 3)> chain; char own[8 * OWN_LDS_BYTES]; };
    __shared__ __attribute__((aligned(16))) LdsPool lds_pool;
    a.own_lds = &lds_pool.own[0] + (threadIdx.x >> 6) * 2 * OWN_LDS_BYTES;
    // (overlapping windows: window 0 starts four rows below row 0, so the offset is kept 256 bytes high and own_prefetch takes them off the scalar
    //  offset again — the margin's dwords then come out of the end of the column before, which exists: don't-care values)
    a.voff_dma = (unsigned)(row0 * (int)EB + lane * 4 + (OVL ? 256 : 0));
    // halo lines (step_march3.hpp): lanes 0..15 hold the first half of the line of (seam w, column c) — window w's words from below, levels 1, 2, 3, 0 in
    // four lanes each —, lanes 48..63 the second half of the line of (seam w+1, column c): its words from above (row 0 / row 3 of the wave: the stages
    // fetch them with row shifts, m_below_h / m_above_h)
    const unsigned hbytes = (unsigned)(p.nwin_total + 1) * (unsigned)(g.nxl + 2) * (unsigned)M3_HL * EB;
    const __amdgpu_buffer_rsrc_t rh = march_rsrc(p.hlines, hbytes);
    const unsigned hoff = (unsigned)(lane >= 48 ? ((w + 1) * (g.nxl + 2) + 1) * M3_HL + lane - 32 : (w * (g.nxl + 2) + 1) * M3_HL + (lane < 16 ? lane : 0)) * EB;
    if constexpr (OVL) { m.lds_w = nullptr; m.lds_r = nullptr; m.voff_lo = m.voff_hi = 0; m.rs3 = a.rs; }
    else {
        // per wave, in elements: below[40] (slot k = the window's last four rows of direction k, staged by the last 4/S lanes; slot
        // 9 stays zero), above[40] (rows 0..3, the first 4/S lanes), then a scratch area for the other lanes' writes
        constexpr int EDGE = 4 / S;                       // lanes that hold four rows
        constexpr int NCH = 10 * (int)sizeof(T) / 4;      // 16-byte chunks of one half
        __shared__ __attribute__((aligned(16))) T seam_lds[4][2 * M3_SHALF + 176];
        T *wl = &seam_lds[threadIdx.x >> 6][0];
        m.lds_w = lane >= 64 - EDGE ? wl + (lane - (64 - EDGE)) * S : (lane < EDGE ? wl + M3_SHALF + lane * S : wl + 2 * M3_SHALF + S * lane);
        m.lds_r = reinterpret_cast<const char *>(wl) + 16 * (lane < NCH ? lane : 0);
        if (lane < M3_SHALF) { wl[lane] = T(0); wl[lane + M3_SHALF] = T(0); }
        const unsigned sbytes = (unsigned)(p.nwin_total + 1) * (unsigned)(g.nxl + 2) * (unsigned)(M3_SREC * EB);
        m.rs3 = march_rsrc(p.seams, sbytes);
        const unsigned rec = (unsigned)(g.nxl + 2) * (unsigned)(M3_SREC * EB);
        m.voff_lo = lane < NCH ? (unsigned)w * rec + (unsigned)(M3_SHALF * EB) + (unsigned)lane * 16u : sbytes;
        m.voff_hi = lane < NCH ? (unsigned)(w + 1) * rec + (unsigned)lane * 16u : sbytes;
    }
    T feq0[9];
    feq_all<T>(T(1), p.U0, T(0), feq0);

    // classes of columns ia-PAD .. ib+PAD-1 (lane l <-> columns ia-PAD+l and ia-PAD+64+l): two ballots each (ClassMask), up to 128 columns
    constexpr int PAD = DEPTH == 4 ? 3 : 2;
    ClassMask nonfast_m, solid_m;
    {
        const int n = ib - ia + 2 * PAD;
        const int col = ia - PAD + lane;
        uint8_t cls = WC_FAST, cls2 = WC_FAST;
        if (lane < n && col >= -1 && col <= g.nxl) cls = p.wcls[(long)w * (g.nxl + 2) + col + 1];
        if (lane + 64 < n && col + 64 >= -1 && col + 64 <= g.nxl) cls2 = p.wcls[(long)w * (g.nxl + 2) + col + 64 + 1];
        nonfast_m.lo = __ballot(cls != WC_FAST); nonfast_m.hi = __ballot(cls2 != WC_FAST);
        solid_m.lo = __ballot(cls == WC_SOLID); solid_m.hi = __ballot(cls2 == WC_SOLID);
    }
    const ClassMask no_m{0ULL, 0ULL};
    const bool lean = (nonfast_m.lo | nonfast_m.hi) == 0ULL && ia + g.gi0 >= PAD + 1 && ib + g.gi0 <= g.nx_g - PAD - 1 && !(uflags & MU_OUTLET_AFTER) && !(p.rev & 2);
    if constexpr (DEPTH >= 3) {
        // chain blocks (step_chain.hpp): the four waves of this workgroup share the edge columns of their units through LDS.  The flag is
        // per block (set by chain_blocks on the host for all four units or none), so the barriers inside are workgroup-uniform.
        if (uflags & MU_CHAIN) {
            constexpr int FDP = (DEPTH == 4 && sizeof(T) == 4) ? (FD | MARCH_FD_PACKED) : FD;
            ChainLds<T, S, DEPTH> &chain_lds = lds_pool.chain;
            const bool end_shared = (uflags & MU_END_SHARED) != 0;
            if (lane < 2 * (DEPTH - 1)) chain_lds.flag[lane / (DEPTH - 1)][lane % (DEPTH - 1)][u & 3] = 0;     // my own hand-over flags ...
            __syncthreads();                                                                              // ... before anybody polls them
            if (uflags & MU_DIR_NEG) {
                ChainUnit<-1, DEPTH, EMIT, FDP, T, S> cu(p, m, rh, feq0, chain_lds);
                cu.hoff = hoff; cu.j0 = j0; cu.lane = lane; cu.pos = u & 3; cu.far_win = far_win;
                cu.run(ia, ib, end_shared);
            } else {
                ChainUnit<1, DEPTH, EMIT, FDP, T, S> cu(p, m, rh, feq0, chain_lds);
                cu.hoff = hoff; cu.j0 = j0; cu.lane = lane; cu.pos = u & 3; cu.far_win = far_win;
                cu.run(ia, ib, end_shared);
            }
            return;
        }
    }
    if (DEPTH == 4) {
        constexpr int FDP = sizeof(T) == 4 ? (FD | MARCH_FD_PACKED) : FD;     // fp32: the packed two-site collision (step_march.hpp)
        if (lean) march_unit4<false, EMIT, FDP, T, S>(p, m, rh, hoff, ia, ib, uflags, j0, lane, far_win, no_m, no_m, feq0);
        else march_unit4<true, EMIT, FDP, T, S>(p, m, rh, hoff, ia, ib, uflags, j0, lane, far_win, nonfast_m, solid_m, feq0);
    } else if (DEPTH == 3) {
        if (lean) march_unit3<false, EMIT, FD, T, S>(p, m, rh, hoff, ia, ib, uflags, j0, lane, far_win, no_m, no_m, feq0);
        else march_unit3<true, EMIT, FD, T, S>(p, m, rh, hoff, ia, ib, uflags, j0, lane, far_win, nonfast_m, solid_m, feq0);
    } else {
        if (lean) march_unit3_d2<false, EMIT, FD, T, S>(p, m, rh, hoff, ia, ib, uflags, j0, lane, far_win, no_m, no_m, feq0);
        else march_unit3_d2<true, EMIT, FD, T, S>(p, m, rh, hoff, ia, ib, uflags, j0, lane, far_win, nonfast_m, solid_m, feq0);
    }
}

// ------------------------------------------------------------------------------------------------
// host side: blocks of four units
// ------------------------------------------------------------------------------------------------
// Re-orders a plan into blocks of four consecutive units of one window (one workgroup each), pads every window to whole blocks with
// empty units, and marks the blocks that can run as a chain: four units of at least `depth` columns each whose whole footprint
// (`pad` columns beyond either end included) is FAST and clear of the inlet / outlet columns.  Blocks are listed chunk-major (the
// windows of one column range are neighbours), like the units of the solo plan.
static inline void chain_blocks(MarchPlan &pl, const uint8_t *wcls, const Geom &g, int depth, bool enable)
{
    const int ld = g.nxl + 2, pad = depth == 4 ? 3 : 2;
    std::vector<std::vector<MarchUnit>> per((size_t)pl.nwin);
    for (const MarchUnit &u : pl.units) per[(size_t)u.w].push_back(u);
    struct Block { MarchUnit u[4]; };
    std::vector<Block> blocks;
    for (int w = 0; w < pl.nwin; w++) {
        std::vector<MarchUnit> &v = per[(size_t)w];
        std::sort(v.begin(), v.end(), [](const MarchUnit &a, const MarchUnit &b) { return a.ia < b.ia; });
        for (size_t i = 0; i < v.size(); i += 4) {
            Block b;
            const size_t n = std::min<size_t>(4, v.size() - i);
            for (size_t k = 0; k < 4; k++) b.u[k] = k < n ? v[i + k] : MarchUnit{0, 0, w, 0};
            bool chain = enable && n == 4;
            if (chain) {
                const int lo = b.u[0].ia - pad, hi = b.u[3].ib + pad;            // footprint [lo, hi)
                chain = lo + g.gi0 >= 1 && hi + g.gi0 <= g.nx_g - 1 && lo >= 0 && hi <= g.nxl;
                for (int k = 0; k < 4 && chain; k++)
                    chain = b.u[k].ib - b.u[k].ia >= depth + 1 && !(b.u[k].flags & MU_OUTLET_AFTER) && (k == 0 || b.u[k].ia == b.u[k - 1].ib);
                const uint8_t *c = wcls + (size_t)w * ld + 1;
                for (int x = lo; x < hi && chain; x++) chain = c[x] == WC_FAST;
            }
            if (chain) {
                b.u[0].flags |= MU_CHAIN | MU_DIR_NEG;
                b.u[1].flags |= MU_CHAIN | MU_END_SHARED;
                b.u[2].flags |= MU_CHAIN | MU_DIR_NEG | MU_END_SHARED;
                b.u[3].flags |= MU_CHAIN;
            }
            blocks.push_back(b);
        }
    }
    std::stable_sort(blocks.begin(), blocks.end(), [](const Block &x, const Block &y) { return x.u[0].ia != y.u[0].ia ? x.u[0].ia < y.u[0].ia : x.u[0].w < y.u[0].w; });
    pl.units.clear();
    for (const Block &b : blocks)
        for (int k = 0; k < 4; k++) pl.units.push_back(b.u[k]);
}

// The kernel's hand-over is safe only on a well-formed plan: chain_receive polls an LDS flag that only the partner wave of the same workgroup
// sets, and a workgroup barrier sits behind `uflags & MU_CHAIN` — a block with three chain units and one solo unit, a chain unit shorter than
// the pipeline, or two "partners" that are not neighbours would spin for ever or read columns nobody published.  The planners above produce
// well-formed plans by construction (tests/_march_plan_check.hip); this is the same statement checked on EVERY plan the library uploads
// (windtunnel.hip upload_units), whoever built it.  A group of four that carries a chain flag must be: four chain units of one window,
// contiguous, at least depth + 1 columns each, flagged (DIR_NEG | -), (END_SHARED), (DIR_NEG | END_SHARED), (-), none the outlet unit, the
// footprint (`pad` columns beyond either end) inside the local lattice, clear of the tunnel's ends and FAST throughout.
static inline bool chain_group_ok(const MarchUnit *u, const uint8_t *wcls, const Geom &g, int depth)
{
    const int ld = g.nxl + 2, pad = depth == 4 ? 3 : 2;
    static const int want[4] = {MU_CHAIN | MU_DIR_NEG, MU_CHAIN | MU_END_SHARED, MU_CHAIN | MU_DIR_NEG | MU_END_SHARED, MU_CHAIN};
    for (int k = 0; k < 4; k++) {
        if (u[k].flags != want[k] || u[k].w != u[0].w || u[k].ib - u[k].ia < depth + 1) return false;
        if (k && u[k].ia != u[k - 1].ib) return false;
    }
    const int lo = u[0].ia - pad, hi = u[3].ib + pad;
    if (lo < 0 || hi > g.nxl || lo + g.gi0 < 1 || hi + g.gi0 > g.nx_g - 1) return false;
    const uint8_t *c = wcls + (size_t)u[0].w * ld + 1;
    for (int x = lo; x < hi; x++)
        if (c[x] != WC_FAST) return false;
    return true;
}

// Checks every aligned group of four units; a group that carries chain flags without being a well-formed chain block is DOWNGRADED to solo units
// (flags cleared; a unit longer than a solo unit may be — its class masks cover max_solo columns — is split), never launched as it is.  Returns the
// number of groups downgraded; the list stays a whole number of groups (well-formed chain blocks first where anything was rebuilt).
static inline int sanitize_chain_plan(MarchPlan &pl, const uint8_t *wcls, const Geom &g, int depth, int max_solo)
{
    const size_t n = pl.units.size();
    bool any_chain = false;
    for (const MarchUnit &u : pl.units) any_chain = any_chain || (u.flags & (MU_CHAIN | MU_DIR_NEG | MU_END_SHARED)) != 0;
    if (!any_chain) return 0;
    int bad = 0;
    std::vector<char> ok((n + 3) / 4, 1);
    for (size_t b = 0; b < n; b += 4) {
        int nc = 0;
        const size_t m = std::min<size_t>(4, n - b);
        for (size_t k = 0; k < m; k++) nc += (pl.units[b + k].flags & (MU_CHAIN | MU_DIR_NEG | MU_END_SHARED)) != 0;
        if (nc == 0) continue;
        if (m < 4 || depth < 3 || !chain_group_ok(&pl.units[b], wcls, g, depth)) { ok[b / 4] = 0; bad++; }
    }
    if (bad == 0 && n % 4 == 0) return 0;
    std::vector<MarchUnit> chain, solo;
    for (size_t b = 0; b < n; b += 4) {
        const size_t m = std::min<size_t>(4, n - b);
        const bool is_chain = ok[b / 4] && m == 4 && (pl.units[b].flags & MU_CHAIN);
        for (size_t k = 0; k < m; k++) {
            MarchUnit u = pl.units[b + k];
            if (is_chain) { chain.push_back(u); continue; }
            if (u.ib <= u.ia) continue;                               // padding
            u.flags &= MU_OUTLET_AFTER;
            const int outlet = u.flags;
            for (int a = u.ia; a < u.ib;) {
                int e = std::min(u.ib, a + max_solo);
                if (e < u.ib && u.ib - e < 2 && u.ib - 2 > a) e = u.ib - 2;      // (a four-step pass wants two columns in a window's last unit)
                solo.push_back(MarchUnit{a, e, u.w, e == u.ib ? outlet : 0});
                a = e;
            }
        }
    }
    while (solo.size() % 4) solo.push_back(MarchUnit{0, 0, solo.empty() ? 0 : solo.back().w, 0});
    pl.units = chain;
    pl.units.insert(pl.units.end(), solo.begin(), solo.end());
    return bad;
}

// The cut by time (build_march_plan_timed) with chain blocks: sweeping a window from the inlet side, a chain block of four units — outer,
// inner, inner, outer, with as many columns as the time limit t leaves after the chain overheads — is placed wherever its whole footprint is
// plain fluid and clear of the tunnel's ends; elsewhere solo units are cut by time as before.  A chain block counts four units, and the
// smallest t whose plan has at most target_units units is found by bisection.  Units come out in chunk-major order with every aligned group
// of four either one chain block or four solo units (the last solo group of the list is padded with empty units).
// solo: columns iterated beyond the unit's own, outlet extra, slow-down of a unit that runs the general loop (inlet / outlet / body in its footprint:
// class tests and scalar branches per column), cost of an iteration on a column beyond the tunnel's end; chain: per-unit overheads in columns
// (solid: what a column of an all-solid tile costs beyond a plain one — such tiles skip the collision, yet their units run the general loop)
struct ChainCost { double over, tail, ov_inner, ov_outer, beta, outside; int max_chain; double solid; };
static inline MarchPlan build_chain_plan_timed(const uint8_t *wcls, const Geom &g, int win, long target_units, double alpha, const MarchRange &r, int min_last,
                                               int max_len, int depth, const ChainCost &cc, const float *colw = nullptr)
{
    MarchPlan pl;
    const int nwin = march_nwin(g.ny, win), ld = g.nxl + 2, pad = depth == 4 ? 3 : 2;
    pl.nwin = nwin;
    const int ncol = r.i_end - r.i_begin;
    if (ncol <= 0) return pl;
    if (target_units < nwin) target_units = nwin;
    const int E = (int)(cc.over / 2) + 2;
    const int n = ncol + 2 * E;
    std::vector<double> C((size_t)nwin * (n + 1), 0.0);        // running cost, as in build_march_plan_timed
    std::vector<int> NF((size_t)nwin * (g.nxl + 3), 0);        // NF[w][x + 2] = non-FAST columns among -1 .. x
    for (int w = 0; w < nwin; w++) {
        const uint8_t *c = wcls + (size_t)w * ld + 1;
        double *Cw = &C[(size_t)w * (n + 1)];
        for (int k = 0; k < n; k++) {
            const int x = r.i_begin - E + k, gi = x + g.gi0;
            double cost = cc.outside;
            if (gi >= 0 && gi < g.nx_g) cost = (x >= -1 && x <= g.nxl && c[x] != WC_FAST) ? 1.0 + (c[x] == WC_SOLID ? cc.solid : alpha) : 1.0;
            if (colw && x >= -1 && x <= g.nxl) cost *= colw[(size_t)w * ld + x + 1];      // measured correction (tune_fuse_plan)
            Cw[k + 1] = Cw[k] + cost;
        }
        int *Nw = &NF[(size_t)w * (g.nxl + 3)];
        for (int x = -1; x <= g.nxl; x++) Nw[x + 2] = Nw[x + 1] + (c[x] != WC_FAST);
    }
    auto Cf = [&](const double *Cw, double x) {
        double k = x - (r.i_begin - E);
        if (k < 0) k = 0;
        if (k > n) k = n;
        const int k0 = (int)k;
        return k0 >= n ? Cw[n] : Cw[k0] + (k - k0) * (Cw[k0 + 1] - Cw[k0]);
    };
    auto plain = [&](int w, int lo, int hi) {                  // columns [lo, hi) all FAST, inside the lattice and clear of the tunnel's ends
        if (lo < 0 || hi > g.nxl || lo + g.gi0 < 1 || hi + g.gi0 > g.nx_g - 1) return false;
        const int *Nw = &NF[(size_t)w * (g.nxl + 3)];
        return Nw[hi + 1] - Nw[lo + 1] == 0;
    };
    auto unit_time = [&](int w, const double *Cw, int ia, int ib) {
        const double t0 = Cf(Cw, ib + cc.over / 2) - Cf(Cw, ia - cc.over / 2) + ((r.outlet_after && ib == r.i_end) ? cc.tail : 0.0);
        return plain(w, ia - pad, ib + pad) ? t0 : cc.beta * t0;
    };
    struct Item { MarchUnit u[4]; int n; };                    // a chain block (n = 4) or one solo unit (n = 1)
    auto cut = [&](int w, double t, std::vector<Item> *out) {
        const double *Cw = &C[(size_t)w * (n + 1)];
        // (chain units read no class masks: their length is not bound by the 64-bit masks of the solo units)
        const int Li = std::min((int)(t - cc.ov_inner), cc.max_chain), Lo = std::min((int)(t - cc.ov_outer), cc.max_chain);
        const int B = 2 * Li + 2 * Lo;
        int ia = r.i_begin;
        long count = 0;
        while (ia < r.i_end) {
            const int rest = r.i_end - (ia + B);
            if (Li >= depth + 1 && Lo >= depth + 1 && rest >= 0 && (rest == 0 ? !r.outlet_after : rest >= min_last) && plain(w, ia - pad, ia + B + pad)) {
                if (out) {
                    Item it; it.n = 4;
                    const int len[4] = {Lo, Li, Li, Lo};
                    const int fl[4] = {MU_CHAIN | MU_DIR_NEG, MU_CHAIN | MU_END_SHARED, MU_CHAIN | MU_DIR_NEG | MU_END_SHARED, MU_CHAIN};
                    int x = ia;
                    for (int k = 0; k < 4; k++) { it.u[k] = MarchUnit{x, x + len[k], w, fl[k]}; x += len[k]; }
                    out->push_back(it);
                }
                ia += B;
                count += 4;
                continue;
            }
            int ib = ia + 1;
            const int cap = std::min(r.i_end, ia + max_len);
            while (ib < cap && unit_time(w, Cw, ia, ib + 1) <= t) ib++;
            if (r.i_end - ib > 0 && r.i_end - ib < min_last) {
                if (r.i_end - min_last <= ia || (r.i_end - ia <= max_len + min_last - 1 && unit_time(w, Cw, ia, r.i_end) <= t)) ib = r.i_end;
                else ib = r.i_end - min_last;
            }
            if (out) { Item it; it.n = 1; it.u[0] = MarchUnit{ia, ib, w, (r.outlet_after && ib == r.i_end) ? MU_OUTLET_AFTER : 0}; out->push_back(it); }
            ia = ib;
            count++;
        }
        return count;
    };
    auto total = [&](double t) { long s = 0; for (int w = 0; w < nwin; w++) s += cut(w, t, nullptr); return s; };
    double lo = 0.0, hi = (double)std::max(max_len, cc.max_chain) * (1.0 + alpha) + cc.over * (1.0 + alpha) + cc.tail + cc.ov_inner + 1.0;
    if (total(hi) > target_units) lo = hi;
    else {
        // the count is not monotonic in t where a block starts or stops fitting, so scan downwards from a bisection estimate
        for (int it = 0; it < 24; it++) {       // (t to 6e-8 of its range: unit counts change at discrete t)
            const double mid = 0.5 * (lo + hi);
            if (total(mid) <= target_units) hi = mid; else lo = mid;
        }
    }
    std::vector<Item> items;
    for (int w = 0; w < nwin; w++) cut(w, hi, &items);
    std::stable_sort(items.begin(), items.end(), [](const Item &x, const Item &y) { return x.u[0].ia != y.u[0].ia ? x.u[0].ia < y.u[0].ia : x.u[0].w < y.u[0].w; });
    std::vector<MarchUnit> solo;
    long cols = 0, live = 0;
    for (const Item &it : items) {
        for (int k = 0; k < it.n; k++) { cols += it.u[k].ib - it.u[k].ia; live++; }
        if (it.n == 4) { for (int k = 0; k < 4; k++) pl.units.push_back(it.u[k]); continue; }
        solo.push_back(it.u[0]);
        if (solo.size() == 4) { for (const MarchUnit &u : solo) pl.units.push_back(u); solo.clear(); }
    }
    if (!solo.empty()) {
        while (solo.size() < 4) solo.push_back(MarchUnit{0, 0, solo[0].w, 0});
        for (const MarchUnit &u : solo) pl.units.push_back(u);
    }
    pl.chunk = live ? (int)((double)cols / (double)live + 0.5) : 0;
    return pl;
}

// Workgroup k of a launch runs on XCD k mod 8 (each with its own L2).  The planners list the workgroups chunk-major — the windows of one column range
// next to one another —, so dealt out round-robin two windows that are vertical neighbours never share an L2.  For OVERLAPPING windows that matters:
// neighbours read one 128-byte line in common per population and column and write the two halves of another.  Here the list is cut into eight
// contiguous runs and run x is dealt to the positions x, x + 8, x + 16 ...: every XCD gets a stretch of column ranges with all their windows.
// (padded with empty workgroups to a multiple of eight; the reversed launch order of every other pass keeps the runs together)
static inline void xcd_order(std::vector<MarchUnit> &units)
{
    const size_t nb = (units.size() + 3) / 4;
    if (nb < 16) return;
    const size_t L = (nb + 7) / 8;
    std::vector<MarchUnit> out(L * 8 * 4, MarchUnit{0, 0, units.empty() ? 0 : units[0].w, 0});
    for (size_t b = 0; b < nb; b++) {
        const size_t x = b / L, q = b % L, i = 8 * q + x;
        for (size_t k = 0; k < 4 && 4 * b + k < units.size(); k++) out[4 * i + k] = units[4 * b + k];
    }
    units.swap(out);
}

}  // namespace wt
