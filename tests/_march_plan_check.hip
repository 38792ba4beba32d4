// Host-only check of build_march_plan (csrc/step_march.hpp): compiled by tests/test_march_plan.py with hipcc and run on the CPU —
// it makes no HIP runtime call.  Invariants: every marched column of every window belongs to exactly one unit, units respect
// the length cap and the minimum length of a window's last unit, the outlet flag sits on the last unit only, and with a target
// the unit count does not exceed it (whole resident rounds).
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include "../airfoil-cfd-tool_amd/csrc/step_chain.hpp"
using namespace wt;

static int check(const char *name, int nxl, int ny, int gi0, int nx_g, int win, int depth, long target, int max_cost, unsigned seed, int timed = 1)
{
    Geom g{};
    g.nxl = nxl; g.ny = ny; g.gi0 = gi0; g.nx_g = nx_g; g.pitch = (ny + 255) / 256 * 256; g.plane = (long)(nxl + 2) * g.pitch + 4352;
    const int nwin = march_nwin(ny, win), ld = nxl + 2;
    std::vector<uint8_t> wcls((size_t)nwin * ld, WC_FAST);
    std::mt19937 rng(seed);
    for (int w = 0; w < nwin; w++)
        for (int x = 0; x < nxl; x++) {     // a body-like cluster of non-FAST tiles in a third of the windows, a few stray ones elsewhere
            const unsigned r = rng() % 1000;
            const bool body = (w % 3 == 1) && x > nxl / 3 && x < nxl / 2;
            if (body ? r < 600 : r < 2) wcls[(size_t)w * ld + x + 1] = r % 3 == 0 ? WC_SOLID : WC_GENERAL;
        }
    const MarchRange r = depth >= 3 ? march_range3(g, depth) : march_range(g);
    const int min_last = depth == 4 ? 2 : 1, max_len = depth == 4 ? MARCH3_MAX_CHUNK - 3 : (depth == 3 ? MARCH3_MAX_CHUNK : MARCH_MAX_CHUNK);
    // max_cost > 0 (the fuse_chunk option): the cut by owned columns; otherwise the library's default, the cut by time
    const double over = depth == 4 ? 4.5 : (depth == 3 ? 2.7 : 1.5), tail = depth == 4 ? 1.25 : (depth == 3 ? 1.0 : 0.5);
    const bool chain_timed = depth >= 3 && max_cost <= 0 && (timed == 1 || timed == 3);
    // timed 3 / 4: with measured corrections of the column costs (tune_fuse_plan), here random factors within the library's clamp
    std::vector<float> colw;
    if (timed >= 3) { colw.resize(wcls.size()); for (float &c : colw) c = 0.2f + 5.8f * (float)(rng() % 1000) / 999.0f; }
    const float *cw = colw.empty() ? nullptr : colw.data();
    const ChainCost cc{over, tail, depth == 4 ? 2.0 : 1.4, depth == 4 ? 2.0 : 1.4, 1.25, 0.6, 160, 0.3};
    MarchPlan pl = (max_cost > 0 || timed == 0) ? build_march_plan(wcls.data(), g, win, target, max_cost, 2.0, &r, min_last, max_len, depth >= 3 ? 4 : 1)
                   : chain_timed                ? build_chain_plan_timed(wcls.data(), g, win, target, 2.2, r, min_last, max_len, depth, cc, cw)
                                                : build_march_plan_timed(wcls.data(), g, win, target, 2.2, r, min_last, max_len, over, tail, 0.3, cw);
    int bad = 0;
    size_t n_chain = 0;
    if (depth >= 3) {
        // step_chain.hpp: whole blocks of four units of one window; chain flags on all four or none; a chain block is contiguous, plain
        // fluid over its whole footprint, clear of the tunnel ends, and its units hold at least `depth` columns each
        if (!chain_timed) chain_blocks(pl, wcls.data(), g, depth, true);
        const int pad = depth == 4 ? 3 : 2;
        if (pl.units.size() % 4) bad++;
        for (size_t b = 0; b + 3 < pl.units.size(); b += 4) {
            const MarchUnit *u = &pl.units[b];
            int nc = 0;
            for (int k = 0; k < 4; k++) nc += (u[k].flags & MU_CHAIN) != 0;
            for (int k = 0; k < 4; k++) if ((nc || !chain_timed) && u[k].w != u[0].w) bad++;      // (solo groups of the cut by time may mix windows)
            if (nc != 0 && nc != 4) { bad++; continue; }
            if (!nc) { for (int k = 0; k < 4; k++) if (u[k].flags & (MU_DIR_NEG | MU_END_SHARED)) bad++; continue; }
            n_chain += 4;
            if ((u[0].flags & ~MU_CHAIN) != MU_DIR_NEG || (u[1].flags & ~MU_CHAIN) != MU_END_SHARED ||
                (u[2].flags & ~MU_CHAIN) != (MU_DIR_NEG | MU_END_SHARED) || (u[3].flags & ~MU_CHAIN) != 0) bad++;
            for (int k = 0; k < 4; k++) { if (u[k].ib - u[k].ia < depth + 1) bad++; if (k && u[k].ia != u[k - 1].ib) bad++; }
            const int lo = u[0].ia - pad, hi = u[3].ib + pad;
            if (lo < 0 || hi > nxl || lo + gi0 < 1 || hi + gi0 > nx_g - 1) { bad++; continue; }
            for (int x = lo; x < hi; x++) if (wcls[(size_t)u[0].w * ld + x + 1] != WC_FAST) bad++;
        }
    }
    std::vector<int> cover((size_t)nwin * nxl, 0);
    std::vector<int> last_len(nwin, -1), n_outlet(nwin, 0);
    for (const MarchUnit &u : pl.units) {
        if (depth >= 3 && u.ib == u.ia && u.flags == 0) continue;      // padding unit of a window's last block
        if (u.w < 0 || u.w >= nwin || u.ia < r.i_begin || u.ib > r.i_end || u.ib <= u.ia) { bad++; continue; }
        if (u.ib - u.ia > ((u.flags & MU_CHAIN) ? 160 : max_len + min_last - 1)) bad++;      // a suppressed cut before a short last unit may add min_last - 1 columns
        for (int x = u.ia; x < u.ib; x++) cover[(size_t)u.w * nxl + x]++;
        if (u.ib == r.i_end) last_len[u.w] = u.ib - u.ia;
        if (u.flags & MU_OUTLET_AFTER) { n_outlet[u.w]++; if (u.ib != r.i_end || !r.outlet_after) bad++; }
    }
    for (int w = 0; w < nwin; w++) {
        for (int x = 0; x < nxl; x++) if (cover[(size_t)w * nxl + x] != ((x >= r.i_begin && x < r.i_end) ? 1 : 0)) bad++;
        if (r.i_end - r.i_begin >= min_last && last_len[w] < min_last) bad++;
        if (n_outlet[w] != (r.outlet_after ? 1 : 0)) bad++;
    }
    if (timed && max_cost <= 0) {       // the cut by time: at most `target` units before block padding (unless the length cap forces more)
        size_t live = 0; for (const MarchUnit &u : pl.units) live += u.ib > u.ia;
        if (target >= nwin && pl.chunk < max_len - 1 && (long)live > target) bad++;
    }
    if (max_cost <= 0 && target >= nwin && pl.chunk < max_len && (long)pl.units.size() > target + 3 * nwin) bad++;      // whole rounds (+ block padding), unless the length cap forces more units
    printf("%-28s nxl %5d ny %5d win %3d depth %d target %5ld max_cost %2d: %6zu units (%zu in chain blocks), chunk %2d  %s\n", name, nxl, ny, win, depth, target, max_cost,
           pl.units.size(), n_chain, pl.chunk, bad ? "FAIL" : "ok");
    return bad;
}

// The library checks every plan it uploads (sanitize_chain_plan, step_chain.hpp): a group of four units that carries chain flags without being a
// well-formed chain block must come out as solo units — never reach the kernel, whose LDS hand-over would spin on it.  Corrupt a good plan in
// the ways a planner bug could, and look at what the guard hands on.
static int check_sanitize(int depth, int corruption, unsigned seed)
{
    Geom g{};
    const int nxl = 544, ny = 4096, gi0 = 1760, nx_g = 4096, win = 128;
    g.nxl = nxl; g.ny = ny; g.gi0 = gi0; g.nx_g = nx_g; g.pitch = (ny + 255) / 256 * 256; g.plane = (long)(nxl + 2) * g.pitch + 4352;
    const int nwin = march_nwin(ny, win), ld = nxl + 2;
    std::vector<uint8_t> wcls((size_t)nwin * ld, WC_FAST);
    std::mt19937 rng(seed);
    for (int w = 0; w < nwin; w++)
        for (int x = 0; x < nxl; x++)
            if ((w % 3 == 1) && x > nxl / 3 && x < nxl / 2 && rng() % 1000 < 600) wcls[(size_t)w * ld + x + 1] = WC_GENERAL;
    const MarchRange r = march_range3(g, depth);
    const int min_last = depth == 4 ? 2 : 1, max_len = depth == 4 ? MARCH3_MAX_CHUNK - 3 : MARCH3_MAX_CHUNK;
    const double over = depth == 4 ? 4.5 : 2.7, tail = depth == 4 ? 1.25 : 1.0;
    const ChainCost cc{over, tail, depth == 4 ? 2.0 : 1.4, depth == 4 ? 2.0 : 1.4, 1.25, 0.6, 160, 0.3};
    // (a small target: long chain units, so that a downgraded one has to be split for the solo kernel's class masks)
    MarchPlan pl = build_chain_plan_timed(wcls.data(), g, win, corruption == 5 ? 192 : 2048, 2.2, r, min_last, max_len, depth, cc, nullptr);
    MarchPlan good = pl;
    int bad = 0;
    if (sanitize_chain_plan(good, wcls.data(), g, depth, max_len - 2) != 0 || good.units.size() != pl.units.size()) { printf("sanitize touched a good plan\n"); bad++; }
    std::vector<size_t> chain_groups;
    for (size_t b = 0; b + 3 < pl.units.size(); b += 4) if (pl.units[b].flags & MU_CHAIN) chain_groups.push_back(b);
    if (chain_groups.empty()) { printf("no chain block to corrupt\n"); return 1; }
    const size_t b = chain_groups[rng() % chain_groups.size()];
    MarchUnit *u = &pl.units[b];
    switch (corruption) {
    case 0: u[2].flags &= ~MU_CHAIN; break;                                   // three chain units and a solo one: the barrier would not be uniform
    case 1: u[1].ib -= (u[1].ib - u[1].ia) - depth; u[2].ia = u[1].ib; break;  // a chain unit shorter than the pipeline (depth columns)
    case 2: std::swap(u[1], u[2]); break;                                     // partners that are not neighbours
    case 3: u[0].flags = MU_CHAIN; break;                                     // wrong direction flag
    case 4: wcls[(size_t)u[0].w * ld + u[1].ia + 1] = WC_GENERAL; break;       // a body column inside the footprint
    case 5: u[3].flags |= MU_CHAIN | MU_END_SHARED; break;                     // (long units) an end seam nobody shares
    }
    if (corruption == 1 && u[2].ib - u[2].ia > 160) return 0;
    const int nd = sanitize_chain_plan(pl, wcls.data(), g, depth, max_len - 2);
    if (nd < 1) { printf("corruption %d (depth %d) not detected\n", corruption, depth); bad++; }
    if (pl.units.size() % 4) bad++;
    std::vector<int> cover((size_t)nwin * nxl, 0);
    for (size_t q = 0; q + 3 < pl.units.size(); q += 4) {
        int nc = 0;
        for (int k = 0; k < 4; k++) nc += (pl.units[q + k].flags & (MU_CHAIN | MU_DIR_NEG | MU_END_SHARED)) != 0;
        if (nc && !chain_group_ok(&pl.units[q], wcls.data(), g, depth)) { printf("an ill-formed chain group survived\n"); bad++; }
    }
    for (const MarchUnit &x : pl.units) {
        if (x.ib <= x.ia) continue;
        if (!(x.flags & MU_CHAIN) && x.ib - x.ia > max_len) bad++;
        for (int c = x.ia; c < x.ib; c++) cover[(size_t)x.w * nxl + c]++;
    }
    for (int w = 0; w < nwin; w++)
        for (int x = 0; x < nxl; x++) if (cover[(size_t)w * nxl + x] != ((x >= r.i_begin && x < r.i_end) ? 1 : 0)) { bad++; break; }
    printf("sanitize: depth %d corruption %d -> %d group(s) downgraded, %zu units  %s\n", depth, corruption, nd, pl.units.size(), bad ? "FAIL" : "ok");
    return bad;
}

// xcd_order (step_chain.hpp): the workgroups (aligned groups of four units) of a plan dealt to the XCDs in contiguous runs — a permutation of the
// groups, padded with empty groups to a multiple of eight, every group intact, group b of the chunk-major list on XCD b / ceil(groups / 8)
static int check_xcd_order(int ngroups, unsigned seed)
{
    std::mt19937 rng(seed);
    std::vector<MarchUnit> u;
    for (int b = 0; b < ngroups; b++)
        for (int k = 0; k < 4; k++) u.push_back(MarchUnit{(int)(1 + b * 40 + k * 10), (int)(1 + b * 40 + k * 10 + 10), (int)(rng() % 35), (int)(rng() % 16)});
    const std::vector<MarchUnit> in = u;
    xcd_order(u);
    int bad = 0;
    if (ngroups < 16) { bad += u.size() != in.size(); for (size_t i = 0; i < in.size() && !bad; i++) bad += u[i].ia != in[i].ia; }
    else {
        const size_t L = ((size_t)ngroups + 7) / 8;
        bad += u.size() != L * 32;
        std::vector<int> seen((size_t)ngroups, 0);
        for (size_t i = 0; i + 3 < u.size(); i += 4) {
            if (u[i].ib <= u[i].ia) { for (int k = 0; k < 4; k++) bad += u[i + k].ib > u[i + k].ia; continue; }       // an empty group is empty throughout
            const int b = (u[i].ia - 1) / 40;
            for (int k = 0; k < 4; k++) bad += u[i + k].ia != in[(size_t)4 * b + k].ia || u[i + k].w != in[(size_t)4 * b + k].w || u[i + k].flags != in[(size_t)4 * b + k].flags;
            seen[(size_t)b]++;
            bad += (i / 4) % 8 != (size_t)b / L;                          // workgroup position mod 8 = the XCD = the run the group belongs to
        }
        for (int b = 0; b < ngroups; b++) bad += seen[(size_t)b] != 1;
    }
    printf("xcd_order, %4d groups: %s\n", ngroups, bad ? "FAIL" : "ok");
    return bad;
}

int main()
{
    int bad = 0;
    for (int n : {3, 15, 16, 17, 255, 256, 512, 513, 519}) bad += check_xcd_order(n, 7u + (unsigned)n);
    for (int depth : {3, 4})
        for (int c = 0; c < 6; c++) bad += check_sanitize(depth, c, 100u + (unsigned)(10 * depth + c));
    for (int depth : {2, 3, 4}) {
        const int win = depth == 2 ? 256 : 128;
        bad += check("whole lattice", 4096, 4096, 0, 4096, win, depth, 4096, 0, 1);
        bad += check("whole lattice, tiny target", 512, 256, 0, 512, win, depth, 2048, 0, 2);      // one-column units
        bad += check("whole lattice, max_cost", 512, 256, 0, 512, win, depth, 2048, 7, 3);
        bad += check("whole lattice, max_cost 1", 300, 520, 0, 300, win, depth, 2048, 1, 4);
        bad += check("left slab", 544, 4096, 0, 4096, win, depth, 2048, 0, 5);
        bad += check("middle slab", 544, 4096, 1760, 4096, win, depth, 2048, 0, 6);
        bad += check("right slab", 544, 4096, 3552, 4096, win, depth, 2048, 0, 7);
        bad += check("narrow slab", 24, 1000, 64, 4096, win, depth, 2048, 0, 8);
        bad += check("fp64-like windows", 4096, 2048, 0, 4096, 64, depth, 6144, 0, 9);
        bad += check("long units", 16384, 256, 0, 16384, win, depth, 16, 0, 10);                    // cap-forced cuts
        bad += check("whole lattice, by columns", 4096, 4096, 0, 4096, win, depth, 4096, 0, 11, 0);
        bad += check("short window", 9, 300, 0, 9, win, depth, 2048, 0, 12);
        bad += check("whole lattice, by time, solo", 4096, 4096, 0, 4096, win, depth, 4096, 0, 13, 2);
        bad += check("middle slab, by time, solo", 544, 4096, 1760, 4096, win, depth, 2048, 0, 14, 2);
        bad += check("whole lattice, measured costs", 4096, 4096, 0, 4096, win, depth, 2048, 0, 15, 3);
        bad += check("middle slab, measured costs", 544, 4096, 1760, 4096, win, depth, 2048, 0, 16, 3);
        bad += check("slab, measured costs, solo", 544, 4096, 1760, 4096, win, depth, 2048, 0, 17, 4);
    }
    printf("%s\n", bad ? "PLAN CHECK FAILED" : "plan check passed");
    return bad ? 1 : 0;
}
