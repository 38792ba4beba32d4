// windtunnel.hip — C-ABI implementation of libwindtunnel.so (see include/windtunnel.h).
//
// Replaces the JS LBM runtime of the reference (pages/airfoil_flow_lbm_aerolab.html:424-700):
// texture/FBO ping-pong -> two SoA lattices in HBM; simStep -> step kernels on a HIP stream;
// readMacro/updateFieldsFromMacro/computeForces/renderField -> on-demand kernels.
#include <hip/hip_runtime.h>
#include "rccl_bind.hpp"      // <rccl/rccl.h> for the types; the entry points are bound at run time to the one RCCL of the process

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <algorithm>
#include <vector>

#include "../../include/windtunnel.h"
#include "kernels.hpp"
#include "step_fast.hpp"
#include "step_march.hpp"
#include "step_march3.hpp"
#include "step_chain.hpp"
#include "canvas.hpp"
#ifndef WT_LOAD_AUX
#define WT_LOAD_AUX 2
#endif

using namespace wt;

// Environment knobs.  Five are part of the interface (include/windtunnel.h: WT_FUSE2, WT_FUSE_CHUNK, WT_FAST_DIV, WT_CHAIN, WT_TUNE): they preset a
// handle's options at wt_create.  The four that decide the pass schedule (fuse_steps, fuse_chunk, fast_div, chain) enter the cross-rank fingerprint
// through their options (schedule_fingerprint lists exactly what is compared); WT_TUNE / "tune" does NOT and must not — the measured cut is a rank's
// own affair (its trial passes exchange nothing) — and neither do "trim_ghosts", "fast_div_two_op" and "exchange_timing" (ADVICE r4).
// The others belong to the experiments under tools/ and exist only in a library built with -DWT_EXPERIMENT_KNOBS (`make lib EXPERIMENT=1`): a
// production library cannot be steered into a plan its neighbours do not share by a stray environment variable.
static inline const char *exp_env(const char *name)
{
#ifdef WT_EXPERIMENT_KNOBS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(e_ == hipErrorOutOfMemory ? WT_ERR_OOM : WT_ERR_HIP, "%s failed: %s (%s:%d)",   \
                        #expr, hipGetErrorString(e_), __FILE__, __LINE__);                              \
    } while (0)

#define NCCL_TRY(expr)                                                                                  \
    do {                                                                                                \
        ncclResult_t r_ = (expr);                                                                       \
        if (r_ != ncclSuccess)                                                                          \
            return fail(WT_ERR_RCCL, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__,  \
                        __LINE__);                                                                      \
    } while (0)

#define WT_TRY(expr)                \
    do {                            \
        int rc_ = (expr);           \
        if (rc_ != WT_OK) return rc_; \
    } while (0)

// ------------------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------------------
enum Transport { TR_NONE = 0, TR_LOCAL = 1, TR_RCCL = 2 };

struct wt_handle {
    int nx_g = 0, ny = 0, dtype = WT_F32, device = 0;
    int rank = 0, nranks = 1, halo = 0;
    int x0 = 0, width = 0;       // owned global columns [x0, x0+width)
    int min_width = 0;           // owned columns of the narrowest slab of the tunnel
    int max_width = 0;           // ... and of the widest one (every rank knows both from the split)
    std::vector<int> edges;      // the split: slab r owns [edges[r], edges[r+1])
    int plan_columns = 0;        // option "plan_columns": choose the steps per pass as for a lattice of this many local columns (0: this one's)
    int gl = 0, gr = 0;          // ghost columns on the left / right
    Geom g{};                    // local geometry
    size_t esz = 4;              // element size
    void *f[2] = {nullptr, nullptr};
    int cur = 0;
    void *macro = nullptr;       // 3 * nxl * pitch
    uint8_t *mask = nullptr;     // (nxl+2) * pitch
    uint8_t *tiles = nullptr;    // nxl * tiles_per_col
    int tiles_per_col = 0;
    void *stage = nullptr;       // device staging for layout conversion
    size_t stage_bytes = 0;
    void *partials = nullptr;    // reduction partials (device)
    void *partials_host = nullptr;
    hipStream_t s_compute = nullptr, s_comm = nullptr;
    hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_state = nullptr, ev_halo = nullptr;
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;      // around the marching kernel of a tuning pass
    bool mask_set = false, inited = false;
    bool macro_stale = false;    // wt_write_f replaced the populations: (rho,ux,uy) describe an older state until a step emits them
    int ghost_valid = 0;         // ghost columns still exact (both sides)
    long long steps_done = 0;
    long long device_bytes = 0;
    int transport = TR_NONE;
    ncclComm_t comm = nullptr;
    wt_handle *peer_l = nullptr, *peer_r = nullptr;   // TR_LOCAL
    // two-steps-per-launch mode (step_march.hpp)
    bool fuse = false;
    int fuse_sites = 0;                  // option: sites per lane of the marching kernel (0 = automatic; 2 or 4)
    int fuse_depth = 0;                  // option: steps per pass (0 = automatic; 2 or 3)
    int march_depth = 0;                 // steps per pass the plan's tables are built for (3 / 4: step_march3.hpp; 2: step_march.hpp)
    int pass_cap = 0;                    // longest pass actually taken on those tables (0 = march_depth): fp64 with fuse_depth = 2
    int tau_cap = 0;                     // fp32, set per stepping call: 3 when the fast division by tau is not proved for this tau (IEEE division:
                                         // the four-step kernel would spill registers and is not built for it), else 0
    void *hlines = nullptr;              // depth 3 / 4: the halo lines of k_march3 (step_march3.hpp), one 32-element line per (window, column)
    long long passes = 0;
    long long single_steps = 0;          // k_step launches of whole steps (option "single_steps"): what a fused plan falls back to
    long long march_table_bytes = 0;     // wcls + halo_tab + seams + seam_plain (part of device_bytes)
    int march_s = 0;                     // sites per lane in use (4: fp32 256-row windows; 2: fp64, or fp32 on narrow lattices)
    bool fuse_force = false;             // fuse_steps = 2: also when the lattice is too small for it to pay
    int fuse_chunk = 0;                  // cost limit of a unit (columns); 0 = whole resident rounds of units (build_march_plan)
    bool fuse_ready = false;
    int fuse_chunk_used = 0;
    uint8_t *bcode = nullptr;            // bounce codes, (nxl+2) * pitch
    uint8_t *wcls = nullptr;             // window-tile classes, nwin * (nxl+2)
    void *halo_tab = nullptr;            // halo table of the two-step marching kernel, (nwin+1) * (nxl+2) * 8 elements (depth 3 / 4: hlines)
    void *seams = nullptr;               // seam rows written by a marching pass beside its output lattice, (nwin+1) * (nxl+2) * 48 elements
    uint8_t *seam_plain = nullptr;       // per (seam, column): both sites next to the seam are plain interior fluid, (nwin-1) * nxl
    bool seams_valid = false;            // `seams` describes lattice f[cur] (set by a marching pass, cleared by everything else that writes f)
    MarchUnit *d_units = nullptr;
    size_t units_cap = 0;
    int n_units = 0, n_win = 0, nonfast_tiles = 0;
    std::vector<uint8_t> host_wcls;
    // fast division by tau (d2q9.hpp): proved per tau on the device before it is used
    unsigned int *d_nbad = nullptr;
    float fd_tau = 0.0f;
    bool fd_checked = false, fd_ok = false;       // the three-operation division by tau is proved for fd_tau (d2q9.hpp)
    bool fd2_ok = false;                          // ... and so is the two-operation one (the four-step fp32 kernel's; rank-local: same bits either way)
    bool fast_div = true;                // option "fast_div"
    double selftest_tau = 0.58;          // option "selftest_tau": the tau the read-only options "selftest_fastdiv32_*" / "selftest_fastdiv64" check
    bool two_op = true;                  // option "fast_div_two_op": the four-step fp32 kernel divides by tau in two operations where proved (d2q9.hpp)
    bool fast_math = false;              // option "fast_math": contracted collision in the marching kernels (opt-in; tolerance, not bit-equality)
    bool chain = true;                   // option "chain": plain-fluid workgroups share their units' edge columns (step_chain.hpp)
    int win_overlap = -1;                // option "window_overlap": -1 automatic (fp32 slabs), 0 windows that tile the column + halo lines, 1 overlapping windows
    bool ovl = false;                    // the present plan's windows overlap (k_march3, step_chain.hpp): no halo lines, no halo kernel, no seam buffer in use
    int n_chain_units = 0;
    // the plan's units on the host, and their refinement by measured unit times (tune_fuse_plan)
    std::vector<MarchUnit> host_units;
    long plan_target = 0;
    bool tune = true;                    // option "tune"
    bool plan_tuned = false;
    // masks that change every few passes (a slider being dragged) are not worth the trial passes: a mask that follows one which lived fewer than
    // TUNE_LIVE_PASSES passes is timed only once it has lived that long itself
    long long passes_total = 0, passes_at_mask = 0;
    long long masks_set = 0;
    bool tune_deferred = false;
    int tune_rounds = 0;                 // refinements tried for the present plan
    double tune_gain = 0.0;              // makespan of the modelled plan / makespan of the plan kept (unit clocks)
    unsigned long long *d_clk = nullptr;  // unit clocks of tuning passes: two records of n_units {start, end}
    size_t clk_cap = 0;
    bool clk_on = false;
    size_t clk_off = 0;
    long wave_slots = 0;                 // resident marching waves of the device: CUs x 4 SIMDs x 2 (x WT_MARCH_WAVES / 2 in an experiment build)
    // the page's canvas on the device (canvas.hpp): particle layer, text alphas, polygon / bar / segment staging, the RGBA8 result
    int cv_scale = 0;
    double4 *cv_layer = nullptr;
    float *cv_text = nullptr;
    bool cv_text_set = false;
    bool cv_layer_live = false;          // strokes have been drawn on the particle layer since it was last cleared (a canvas of ANOTHER scale would wipe them)
    uchar4 *cv_out = nullptr;
    double *cv_small = nullptr;          // polygon (up to CV_MAX_POLY points) + bar rows
    double *cv_seg = nullptr;
    size_t cv_seg_cap = 0;
    // Trimmed ghost marching (slab handles, three / four steps per pass): a pass that starts with gv exact ghost columns and advances k steps
    // leaves gv - k of them exact — marching the others is work whose result nobody may read.  One unit list per "exact ghost columns after
    // the pass" (the kept plan's range shrunk at the slab's local edges), cut lazily with the kept plan's measured column costs.
    struct TrimPlan { int v_after = -1; MarchUnit *d_units = nullptr; size_t cap = 0; int n_units = 0; bool valid = false; };
    std::vector<TrimPlan> trim_plans;
    bool trim_prebuilt = false;          // the lists of the two standard cycles exist (prebuild_trim_plans)
    // refresh = 2 (renew_plan_for): per pass length, the unit lists of the interior columns [0] and of the two edge strips [1]
    struct RenewPlan { int depth = 0; MarchUnit *d_units[2] = {nullptr, nullptr}; size_t cap[2] = {0, 0}; int n_units[2] = {0, 0}; int strip_lo[2] = {0, 0}, strip_hi[2] = {0, 0}; bool valid = false; };
    std::vector<RenewPlan> renew_plans;
    long long fused_renewals = 0;        // ghost renewals taken inside a fused pass (option "fused_renewals")
    std::vector<float> colw_kept;        // column-cost corrections of the plan tune_fuse_plan kept (empty: the modelled costs)
    bool trim = true;                    // option "trim_ghosts"
    int refresh_mode = 0;                // option "refresh": 0 = a single step with the exchange beside its interior columns, 1 = the exchange at a
                                         // pass boundary (not overlapped), every step of the cycle fused
    long long boundary_exchanges = 0;    // exchanges taken at a pass boundary (option "boundary_exchanges")
    long long trimmed_passes = 0;        // passes that marched a trimmed range (option "trimmed_passes")
    unsigned int *stuck_host = nullptr;  // host-mapped word a chain unit raises when it gives up waiting for a hand-over (step_chain.hpp chain_receive)
    unsigned int *stuck_dev = nullptr;
    int chain_downgrades = 0;            // groups of four units whose chain flags failed sanitize_chain_plan (option "chain_downgrades"; 0 by construction)
    // cross-rank agreement (slab handles): everything that decides the sequence of passes / single steps / refreshes must be the same on every
    // slab of a tunnel — checked, not assumed (agree_rccl / agree_local)
    bool agree_check = true;             // option "agree_check"
    bool agree_dirty = true;             // a schedule input changed since the last check
    long long *d_agree = nullptr;        // device scratch of the all-reduce
    long long agree_checks = 0;          // checks made (option "agree_checks")
    float fd_agreed_tau = 0.0f;          // tau whose fast-division verdict the ranks have agreed on
    bool fd_agreed = false;
    int comm_ranks = 0;                  // ncclCommCount of the handle's communicator (option "comm_ranks")
    // exchange timing (option "exchange_timing"; bench.py --gpus N): events around the ghost exchange on the comm stream and around the interior
    // kernel / the wait for the exchange on the compute stream of every refresh step
    bool xt_on = false;
    std::vector<hipEvent_t> xt_ev;       // XT_RING x {x0, x1, i0, i1, w}
    int xt_n = 0;                        // refreshes recorded and not yet resolved
    double xt_exchange_ms = 0.0, xt_interior_ms = 0.0, xt_exposed_ms = 0.0;
    long long xt_count = 0;
};
static const int XT_RING = 32;

static const int kReduceBlocks = 1024;
static const long long TUNE_LIVE_PASSES = 16;

template <typename T> static T *fptr(wt_handle *h, int which) { return reinterpret_cast<T *>(h->f[which]); }

static int ensure_stage(wt_handle *h, size_t bytes)
{
    if (h->stage_bytes >= bytes) return WT_OK;
    if (h->stage) { HIP_TRY(hipFree(h->stage)); h->device_bytes -= (long long)h->stage_bytes; h->stage = nullptr; h->stage_bytes = 0; }
    HIP_TRY(hipMalloc(&h->stage, bytes));
    h->stage_bytes = bytes;
    h->device_bytes += (long long)bytes;
    return WT_OK;
}

static int check_handle(const wt_handle *h)
{
    if (!h) return fail(WT_ERR_ARG, "null handle");
    return WT_OK;
}

// ------------------------------------------------------------------------------------------
// life cycle
// ------------------------------------------------------------------------------------------
static int create_impl(int nx_g, int ny, int dtype, int device, int rank, int nranks, int halo, const int *edges, wt_handle **out)
{
    if (!out) return fail(WT_ERR_ARG, "out is null");
    *out = nullptr;
    if (nx_g < 3 || ny < 3) return fail(WT_ERR_ARG, "lattice must be at least 3x3 (got %dx%d)", nx_g, ny);
    if ((long long)nx_g * ny > (1LL << 33)) return fail(WT_ERR_ARG, "lattice too large");
    if (dtype != WT_F32 && dtype != WT_F64) return fail(WT_ERR_ARG, "dtype must be WT_F32 or WT_F64");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(WT_ERR_ARG, "bad rank %d of %d", rank, nranks);
    if (nranks > 1 && halo < 1) return fail(WT_ERR_ARG, "slab handles need halo >= 1");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(WT_ERR_HIP, "no HIP device available (%s); libwindtunnel has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(WT_ERR_ARG, "device %d out of range (0..%d)", device, ndev - 1);
    HIP_TRY(hipSetDevice(device));

    wt_handle *h = new (std::nothrow) wt_handle();
    if (h && getenv("WT_TUNE")) h->tune = atoi(getenv("WT_TUNE")) != 0;      // experiments; the option "tune" is the interface
    if (!h) return fail(WT_ERR_OOM, "host allocation failed");
    h->nx_g = nx_g; h->ny = ny; h->dtype = dtype; h->device = device;
    h->rank = rank; h->nranks = nranks;
    // the split: the caller's (wt_create_slab_at) or equal widths; the narrowest slab of all decides the plan depth of every rank
    h->min_width = nx_g;
    h->max_width = 0;
    h->edges.assign((size_t)nranks + 1, 0);
    for (int r = 0; r < nranks; r++) {
        const int a = edges ? edges[r] : (int)((long long)r * nx_g / nranks), b = edges ? edges[r + 1] : (int)((long long)(r + 1) * nx_g / nranks);
        if (edges && (a < 0 || b > nx_g || b <= a || (r == 0 && a != 0) || (r == nranks - 1 && b != nx_g))) {
            delete h;
            return fail(WT_ERR_ARG, "edges must rise from 0 to nx_global (slab %d: [%d, %d))", r, a, b);
        }
        if (b - a < h->min_width) h->min_width = b - a;
        if (b - a > h->max_width) h->max_width = b - a;
        h->edges[(size_t)r] = a; h->edges[(size_t)r + 1] = b;
        if (r == rank) { h->x0 = a; h->width = b - a; }
    }
    if (nranks > 1 && halo > h->min_width) { delete h; return fail(WT_ERR_ARG, "halo %d wider than the narrowest slab (%d columns)", halo, h->min_width); }
    if (nranks > 1 && h->min_width < 2) { delete h; return fail(WT_ERR_ARG, "slab narrower than 2 columns"); }
    h->halo = nranks > 1 ? halo : 0;
    h->gl = (rank > 0) ? h->halo : 0;
    h->gr = (rank < nranks - 1) ? h->halo : 0;
    h->esz = dtype == WT_F32 ? 4 : 8;
    {
        int cus = 0;
        hipError_t ea = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
        if (ea != hipSuccess || cus <= 0) { delete h; return fail(WT_ERR_HIP, "hipDeviceGetAttribute(multiprocessor count) failed: %s", hipGetErrorString(ea)); }
        long waves = 2;                                                  // resident marching waves per SIMD (229-252 VGPRs)
        if (const char *e = exp_env("WT_MARCH_WAVES")) waves = atoi(e) > 0 ? atoi(e) : 2;
        h->wave_slots = (long)cus * 4 * waves;
    }
    Geom &g = h->g;
    g.nxl = h->gl + h->width + h->gr;
    g.ny = ny;
    g.gi0 = h->x0 - h->gl;
    g.nx_g = nx_g;
    g.pitch = ((long)ny + 255) / 256 * 256;
    {
        // Plane stride: the nine population planes are walked in lock-step, so a stride that is a
        // multiple of a large power of two would park all 18 streams on the same HBM channels.
        // Round the plane to 4 KiB and add an odd multiple of 1 KiB (measured: tools/kbench3).
        const long plane_bytes = ((long)(g.nxl + 2) * g.pitch * (long)h->esz + 4095) / 4096 * 4096 + 17408;
        g.plane = plane_bytes / (long)h->esz;
    }
    h->tiles_per_col = (int)(g.pitch / tile_j_of(h->esz));

    auto cleanup = [&](int rc) { wt_destroy(h); return rc; };
    const size_t lat_bytes = (size_t)9 * g.plane * h->esz;
    const size_t macro_bytes = (size_t)3 * g.nxl * g.pitch * h->esz;
    const size_t mask_bytes = (size_t)(g.nxl + 2) * g.pitch;
    const size_t tile_bytes = (size_t)g.nxl * h->tiles_per_col;
#define CREATE_TRY(expr)                                                                               \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return cleanup(fail(e_ == hipErrorOutOfMemory ? WT_ERR_OOM : WT_ERR_HIP, "%s failed: %s",  \
                                #expr, hipGetErrorString(e_)));                                        \
    } while (0)
    CREATE_TRY(hipMalloc(&h->f[0], lat_bytes));
    CREATE_TRY(hipMalloc(&h->f[1], lat_bytes));
    CREATE_TRY(hipMalloc(&h->macro, macro_bytes));
    CREATE_TRY(hipMalloc((void **)&h->mask, mask_bytes));
    CREATE_TRY(hipMalloc((void **)&h->tiles, tile_bytes));
    CREATE_TRY(hipMalloc(&h->partials, kReduceBlocks * sizeof(ForcePartial)));
    CREATE_TRY(hipHostMalloc(&h->partials_host, kReduceBlocks * sizeof(ForcePartial)));
    CREATE_TRY(hipHostMalloc((void **)&h->stuck_host, sizeof(unsigned int), hipHostMallocMapped));
    *h->stuck_host = 0;
    CREATE_TRY(hipHostGetDevicePointer((void **)&h->stuck_dev, h->stuck_host, 0));
    h->device_bytes = (long long)(2 * lat_bytes + macro_bytes + mask_bytes + tile_bytes + kReduceBlocks * sizeof(ForcePartial));
    int lo = 0, hi = 0;
    CREATE_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CREATE_TRY(hipStreamCreateWithPriority(&h->s_compute, hipStreamNonBlocking, lo));
    CREATE_TRY(hipStreamCreateWithPriority(&h->s_comm, hipStreamNonBlocking, hi));
    CREATE_TRY(hipEventCreate(&h->ev_a));
    CREATE_TRY(hipEventCreate(&h->ev_b));
    CREATE_TRY(hipEventCreate(&h->ev_t0));
    CREATE_TRY(hipEventCreate(&h->ev_t1));
    CREATE_TRY(hipEventCreateWithFlags(&h->ev_state, hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&h->ev_halo, hipEventDisableTiming));
    CREATE_TRY(hipMemsetAsync(h->f[0], 0, lat_bytes, h->s_compute));
    CREATE_TRY(hipMemsetAsync(h->f[1], 0, lat_bytes, h->s_compute));
    CREATE_TRY(hipMemsetAsync(h->macro, 0, macro_bytes, h->s_compute));
    CREATE_TRY(hipMemsetAsync(h->mask, 0, mask_bytes, h->s_compute));
    CREATE_TRY(hipMemsetAsync(h->tiles, 0, tile_bytes, h->s_compute));
    CREATE_TRY(hipStreamSynchronize(h->s_compute));
#undef CREATE_TRY
    {
        // Two steps per pass (step_march.hpp): on by default, engaging only where the plan has enough units to pay
        // (fuse_steps = 1: whole lattices from about 1000 x 4096 up; one-GPU measurements of slab-sized lattices:
        // 2080 columns 98 -> 79 us per step, 1056 columns 53 -> 47, 544 columns no gain and left on the single-step
        // kernel).  Slabs march between ghost refreshes, the refresh step itself stays a single step with the exchange
        // beside its interior columns.  WT_FUSE2 = 0 | 1 | 2 overrides (off / where it pays / always).
        h->fuse = true;
        const char *e = getenv("WT_FUSE2");
        if (e) { h->fuse = atoi(e) != 0; h->fuse_force = atoi(e) >= 2; }
        const char *c = getenv("WT_FUSE_CHUNK");
        if (c && atoi(c) >= 0) h->fuse_chunk = atoi(c);
        const char *fdv = getenv("WT_FAST_DIV");
        if (fdv) h->fast_div = atoi(fdv) != 0;
        const char *ch = getenv("WT_CHAIN");
        if (ch) h->chain = atoi(ch) != 0;
    }
    *out = h;
    return WT_OK;
}

extern "C" int wt_create(int nx, int ny, int dtype, int device, wt_handle **out)
{
    return create_impl(nx, ny, dtype, device, 0, 1, 0, nullptr, out);
}

extern "C" int wt_create_slab(int nx_global, int ny, int dtype, int device, int rank, int nranks, int halo,
                              wt_handle **out)
{
    return create_impl(nx_global, ny, dtype, device, rank, nranks, halo, nullptr, out);
}

extern "C" int wt_create_slab_at(int nx_global, int ny, int dtype, int device, int rank, int nranks, int halo, const int *edges,
                                 wt_handle **out)
{
    if (!edges) return fail(WT_ERR_ARG, "edges is null");
    if (nranks < 1) return fail(WT_ERR_ARG, "bad rank %d of %d", rank, nranks);
    return create_impl(nx_global, ny, dtype, device, rank, nranks, halo, edges, out);
}

extern "C" int wt_destroy(wt_handle *h)
{
    if (!h) return WT_OK;
    (void)hipSetDevice(h->device);
    if (h->s_compute) (void)hipStreamSynchronize(h->s_compute);
    if (h->s_comm) (void)hipStreamSynchronize(h->s_comm);
    if (h->comm) { (void)ncclCommDestroy(h->comm); h->comm = nullptr; }
    if (h->peer_l) h->peer_l->peer_r = nullptr;
    if (h->peer_r) h->peer_r->peer_l = nullptr;
    for (int i = 0; i < 2; i++) if (h->f[i]) (void)hipFree(h->f[i]);
    if (h->macro) (void)hipFree(h->macro);
    if (h->mask) (void)hipFree(h->mask);
    if (h->tiles) (void)hipFree(h->tiles);
    if (h->stage) (void)hipFree(h->stage);
    if (h->bcode) (void)hipFree(h->bcode);
    if (h->wcls) (void)hipFree(h->wcls);
    if (h->halo_tab) (void)hipFree(h->halo_tab);
    if (h->seams) (void)hipFree(h->seams);
    if (h->seam_plain) (void)hipFree(h->seam_plain);
    if (h->d_units) (void)hipFree(h->d_units);
    for (auto &tp : h->trim_plans) if (tp.d_units) (void)hipFree(tp.d_units);
    for (auto &rp : h->renew_plans) for (int i = 0; i < 2; i++) if (rp.d_units[i]) (void)hipFree(rp.d_units[i]);
    if (h->d_clk) (void)hipFree(h->d_clk);
    if (h->d_nbad) (void)hipFree(h->d_nbad);
    if (h->d_agree) (void)hipFree(h->d_agree);
    if (h->stuck_host) (void)hipHostFree(h->stuck_host);
    if (h->cv_layer) (void)hipFree(h->cv_layer);
    if (h->cv_text) (void)hipFree(h->cv_text);
    if (h->cv_out) (void)hipFree(h->cv_out);
    if (h->cv_small) (void)hipFree(h->cv_small);
    if (h->cv_seg) (void)hipFree(h->cv_seg);
    for (hipEvent_t e : h->xt_ev) if (e) (void)hipEventDestroy(e);
    if (h->partials) (void)hipFree(h->partials);
    if (h->partials_host) (void)hipHostFree(h->partials_host);
    if (h->ev_a) (void)hipEventDestroy(h->ev_a);
    if (h->ev_b) (void)hipEventDestroy(h->ev_b);
    if (h->ev_t0) (void)hipEventDestroy(h->ev_t0);
    if (h->ev_t1) (void)hipEventDestroy(h->ev_t1);
    if (h->ev_state) (void)hipEventDestroy(h->ev_state);
    if (h->ev_halo) (void)hipEventDestroy(h->ev_halo);
    if (h->s_compute) (void)hipStreamDestroy(h->s_compute);
    if (h->s_comm) (void)hipStreamDestroy(h->s_comm);
    delete h;
    return WT_OK;
}

extern "C" int wt_get_info(const wt_handle *h, wt_info *info)
{
    WT_TRY(check_handle(h));
    if (!info) return fail(WT_ERR_ARG, "info is null");
    info->nx_global = h->nx_g; info->ny = h->ny; info->dtype = h->dtype; info->device = h->device;
    info->rank = h->rank; info->nranks = h->nranks; info->x0 = h->x0; info->width = h->width;
    info->halo = h->halo; info->reserved = 0; info->steps_done = h->steps_done; info->device_bytes = h->device_bytes;
    return WT_OK;
}

extern "C" const char *wt_last_error(void) { return g_err; }
// "libwindtunnel 0.2 (...); RCCL <version> at <path> (<how it was bound>)" — or "RCCL: not bound yet" before the first wt_comm_* call of the process
extern "C" const char *wt_version(void)
{
    static thread_local std::string v;
    v = "libwindtunnel 0.2 (gfx950, D2Q9 pull, column-major SoA); " + rccl_describe(false);
    return v.c_str();
}
// every wt_comm_* entry point: bind RCCL (once per process) or say why not
static int rccl_require()
{
    const RcclApi &a = rccl_api();
    if (!a.bound) return fail(WT_ERR_RCCL, "%s", a.error.c_str());
    return WT_OK;
}

// A chain unit that gave up waiting for its partner's hand-over (step_chain.hpp: the poll is bounded) has raised the host-mapped word: the
// state it produced is not to be trusted.  Looked at wherever the host has just synchronised with the device AND returns data: wt_sync, every
// read-back (populations, macro planes, fields, colours, the canvas), the reductions (ranges, forces, clamp events), the tracers and the timed
// stepping calls; wt_init_equilibrium and wt_write_f — which replace the state — clear it.
static int check_stuck(wt_handle *h)
{
    if (h->stuck_host && *h->stuck_host != 0)
        return fail(WT_ERR_STATE, "a chain unit of a marching pass gave up waiting for its partner's hand-over (an ill-formed unit plan reached the device): "
                                  "the lattice state is not valid; re-initialise the handle");
    return WT_OK;
}

// (hipStreamSynchronize on a stream that is already idle still costs 8 us on this stack — a timed region of 20 steps is 1.6 ms and ends with one of these per
//  stream —; asking first costs 1)
static inline hipError_t stream_sync(hipStream_t s)
{
    const hipError_t q = hipStreamQuery(s);
    if (q == hipSuccess) return hipSuccess;
    if (q != hipErrorNotReady) { (void)hipGetLastError(); }
    return hipStreamSynchronize(s);
}
extern "C" int wt_sync(wt_handle *h)
{
    WT_TRY(check_handle(h));
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(stream_sync(h->s_compute));
    HIP_TRY(stream_sync(h->s_comm));
    return check_stuck(h);
}

// ------------------------------------------------------------------------------------------
// two-steps-per-launch plan (step_march.hpp)
// ------------------------------------------------------------------------------------------
// whole lattices and column slabs alike (a slab plans over its LOCAL columns, ghosts included); the
// marching kernels address a lattice through one 32-bit buffer descriptor
// A slab decides from the SPLIT, not from its own width: the widest slab's lattice (its width + both halos — what an interior
// slab of that width would hold) must stay below the 4 GiB a buffer descriptor addresses and the narrowest one must have 8 local columns,
// so that either every rank of a tunnel marches or none does, whatever the cut of the slabs.
static unsigned long long plane_elems_of(int nxl, long pitch, size_t esz)
{
    return (unsigned long long)((((long)(nxl + 2) * pitch * (long)esz + 4095) / 4096 * 4096 + 17408) / (long)esz);
}
static bool fuse_eligible_s(const wt_handle *h, int sites)
{
    const unsigned long long eb = h->dtype == WT_F32 ? 4 : 8;
    if (sites * eb != 8) return false;                                      // 8-byte vectors: fp32 with 2 sites per lane, fp64 with 1
    const int nxl_max = h->nranks > 1 ? h->max_width + 2 * h->halo : h->g.nxl;
    const int nxl_min = h->nranks > 1 ? h->min_width + h->halo : h->g.nxl;
    const unsigned long long plane = std::max<unsigned long long>((unsigned long long)h->g.plane, plane_elems_of(nxl_max, h->g.pitch, (size_t)eb));
    return h->g.ny % sites == 0 && std::min(h->g.nxl, nxl_min) >= 8 && 9ULL * plane * eb < (1ULL << 32) - (1ULL << 20);
}
static bool fuse_eligible(const wt_handle *h) { return fuse_eligible_s(h, h->dtype == WT_F32 ? 2 : 1); }
static inline int plan_depth(const wt_handle *h) { return h->pass_cap > 0 && h->pass_cap < h->march_depth ? h->pass_cap : h->march_depth; }
static inline int eff_depth(const wt_handle *h)
{
    const int d = plan_depth(h);
    return h->tau_cap > 0 && h->tau_cap < d ? h->tau_cap : d;
}

static void free_march_tables(wt_handle *h)
{
    if (h->wcls) { (void)hipFree(h->wcls); h->wcls = nullptr; }
    if (h->halo_tab) { (void)hipFree(h->halo_tab); h->halo_tab = nullptr; }
    if (h->seams) { (void)hipFree(h->seams); h->seams = nullptr; }
    if (h->seam_plain) { (void)hipFree(h->seam_plain); h->seam_plain = nullptr; }
    if (h->hlines) { (void)hipFree(h->hlines); h->hlines = nullptr; }
    h->seams_valid = false;
    h->n_win = 0;
    h->device_bytes -= h->march_table_bytes;
    h->march_table_bytes = 0;
}

// The unit list of the handle's tables (host_wcls, march_s, march_depth, plan_target); colw: per (window, column) corrections of the column
// costs measured by tune_fuse_plan, or null.
static bool plan_by_time(const wt_handle *h)
{
    static const int timed = exp_env("WT_PLAN_TIMED") ? atoi(exp_env("WT_PLAN_TIMED")) : 1;
    return h->fuse_chunk <= 0 && timed;
}
static void cut_units(wt_handle *h, const float *colw, MarchPlan *out, const MarchRange *range = nullptr)
{
    const Geom &g = h->g;
    const int depth = h->march_depth, win = 64 * h->march_s - (h->ovl ? 8 : 0);      // (the planners want the rows from window to window: the window count)
    const long target = h->plan_target;
    const MarchRange r = range ? *range : (depth >= 3 ? march_range3(g, depth) : march_range(g));
    // four steps per pass: class masks cover ib - ia + 6 columns, and the last unit of a window marches at least two
    // cost of a column that is not plain fluid, in plain columns: 1 + alpha; `over`: columns a unit iterates over beyond its own (pipeline
    // fill and drain of a `depth`-step pass); `tail`: the outlet column's extra stages — see build_march_plan_timed
    static const double alpha = exp_env("WT_ALPHA") ? atof(exp_env("WT_ALPHA")) : 1.6;
    static const double alpha_solid = exp_env("WT_ALPHA_SOLID") ? atof(exp_env("WT_ALPHA_SOLID")) : alpha;
    static const int timed = exp_env("WT_PLAN_TIMED") ? atoi(exp_env("WT_PLAN_TIMED")) : 1;
    const int min_last = depth == 4 ? 2 : 1, max_len = depth == 4 ? MARCH3_MAX_CHUNK - 3 : (depth == 3 ? MARCH3_MAX_CHUNK : MARCH_MAX_CHUNK);
    const double over = depth == 4 ? 4.5 : (depth == 3 ? 2.7 : 1.5), tail = depth == 4 ? 1.25 : (depth == 3 ? 1.0 : 0.5);
    const bool by_time = plan_by_time(h);
    const bool chain = depth >= 3 && h->chain;
    // chain overheads in columns, from per-unit clocks on a 544-column lattice (tools/unit_clocks.py): a four-step chain unit of 8.5 columns takes as
    // long as 10.5 solo iterations, a three-step one as 9.9
    static const double beta = exp_env("WT_BETA") ? atof(exp_env("WT_BETA")) : 1.25;
    static const int max_chain = exp_env("WT_MAX_CHAIN") ? atoi(exp_env("WT_MAX_CHAIN")) : 160;
    const ChainCost cc{over, tail, depth == 4 ? 2.0 : 1.4, depth == 4 ? 2.0 : 1.4, beta, 0.6, max_chain, alpha_solid};
    // experiments (tools/r5_asym.py): the workgroups of the first half of the launch (dispatched first: one per CU, they win the issue arbitration of their SIMDs
    // against the second half's) run ahead of the others — with a FIXED launch order (WT_MARCH_REV=0) the columns of the list's first half can be made cheaper
    // and the others dearer by a factor 1 -+ gamma, so that the older units get the longer stretches
    std::vector<float> asym;
    if (const char *ea = exp_env("WT_ASYM")) {
        const double gamma = atof(ea);
        if (gamma > 0.0 && gamma < 0.9) {
            const int ld = g.nxl + 2, nwin = h->n_win;
            asym.assign((size_t)nwin * ld, 1.0f);
            const double xs = r.i_begin + (r.i_end - r.i_begin) * (1.0 + gamma) / 2.0;
            for (int w = 0; w < nwin; w++)
                for (int x = -1; x <= g.nxl; x++)
                    asym[(size_t)w * ld + x + 1] = (float)((x < xs ? 1.0 / (1.0 + gamma) : 1.0 / (1.0 - gamma)) * (colw ? colw[(size_t)w * ld + x + 1] : 1.0f));
            colw = asym.data();
        }
    }
    MarchPlan pl = !by_time ? build_march_plan(h->host_wcls.data(), g, win, target, h->fuse_chunk, timed ? alpha : 4.0, &r, min_last, max_len, chain ? 4 : 1)
                   : chain  ? build_chain_plan_timed(h->host_wcls.data(), g, win, target, alpha, r, min_last, max_len, depth, cc, colw)
                            : build_march_plan_timed(h->host_wcls.data(), g, win, target, alpha, r, min_last, max_len, over, tail, alpha_solid, colw);
    if (chain && !by_time) chain_blocks(pl, h->host_wcls.data(), g, depth, true);      // the fuse_chunk option: blocks of four consecutive units of the cut by columns
    *out = std::move(pl);
}

// xcd_order (step_chain.hpp) is on for plans with overlapping windows — slabs and small whole lattices — (profiles/r05_s_xcd_order.txt: stand-alone slabs of the 8-way split of 4096^2 12.4-12.6 against
// 13.0-13.8 us per step plain, 13.9-14.7 against 14.7-14.9 over the body; windows that tile the column gain 2-4 % on a slab and lose on the whole lattice).
static bool xcd_order_on(const wt_handle *h)
{
    static const int e = exp_env("WT_XCD_ORDER") ? atoi(exp_env("WT_XCD_ORDER")) : -1;
    return e >= 0 ? e != 0 : h->ovl;
}

static int upload_units(wt_handle *h, const MarchPlan &plan_in)
{
    // the chain-block invariant the kernel's LDS hand-over rests on, checked on every plan that reaches the device (sanitize_chain_plan,
    // step_chain.hpp): an ill-formed group is downgraded to solo units and counted, never launched
    MarchPlan pl = plan_in;
    if (!h->host_wcls.empty()) {
        const int max_solo = (h->march_depth == 4 ? MARCH3_MAX_CHUNK - 3 : (h->march_depth == 3 ? MARCH3_MAX_CHUNK : MARCH_MAX_CHUNK)) - 2;
        h->chain_downgrades += sanitize_chain_plan(pl, h->host_wcls.data(), h->g, h->march_depth, max_solo);
    }
    if (const char *e = exp_env("WT_DEBUG_CORRUPT_PLAN")) {
        // experiment builds only (tools/r4_stuck_check.py): strip the chain flags of ONE unit of the first chain block BEHIND the guard above, so that
        // its start-seam partner waits for a hand-over that never comes — the bounded poll of chain_receive must end the unit and raise `stuck`
        if (atoi(e) != 0)
            for (size_t b = 0; b + 3 < pl.units.size(); b += 4)
                if (pl.units[b].flags & MU_CHAIN) { pl.units[b + 1].flags = 0; break; }
    }
    if (xcd_order_on(h)) xcd_order(pl.units);
    for (auto &tp : h->trim_plans) tp.valid = false;       // cut from the kept plan's costs: stale now
    h->trim_prebuilt = false;
    for (auto &rp : h->renew_plans) rp.valid = false;
    h->n_chain_units = 0;
    for (const MarchUnit &u : pl.units) h->n_chain_units += (u.flags & MU_CHAIN) != 0;
    const size_t total = pl.units.size();
    h->n_units = (int)total;
    h->fuse_chunk_used = pl.chunk;
    h->host_units = pl.units;
    if (total == 0) return WT_OK;
    if (total > h->units_cap) {
        if (h->d_units) { HIP_TRY(hipFree(h->d_units)); h->d_units = nullptr; h->units_cap = 0; }
        const size_t cap = total + total / 4 + 64;
        HIP_TRY(hipMalloc((void **)&h->d_units, cap * sizeof(MarchUnit)));
        h->units_cap = cap;
    }
    HIP_TRY(hipMemcpyAsync(h->d_units, pl.units.data(), total * sizeof(MarchUnit), hipMemcpyHostToDevice, h->s_compute));
    HIP_TRY(hipStreamSynchronize(h->s_compute));
    return WT_OK;
}

// Overlapping windows (k_march3, step_chain.hpp: four margin rows on either side in place of the halo lines): 128 / 120 of the arithmetic and loads /
// stores that straddle lines for no halo kernel — 10 of the 58 us of a pass on a slab of an 8-way split of 4096^2, 21 of 307 on the whole lattice.
// Automatic: fp32 lattices — whole ones and the slabs of a split alike, by their LOCAL size — of up to 7.5 M sites.  Measured
// (profiles/r05_zy_overlap_sizes.txt, us per step tiling / overlapping): 1024x512 11.8 / 9.3, 2048x1024 19.5 / 14.5, 2048^2 24.8 / 23.2, 3584x2048
// 37.2 / 35.4, but 4096x2048 42.5 / 46.9, 3072^2 45.8 / 51.2, 4096^2 77 / 90-97: the large ones are half bound by their line traffic; slabs of 4096
// rows: 600 columns (8-way split of 4096^2) 19.9 / 17.5, 1150 columns (4-way) 28.4 / 25.4, but 2110 columns (2-way) 44.1 / 46.4-48.3, and the
// 2170-column slabs of 16384x4096 47.6-47.9 / 47.4-50.5.  fp64 windows would keep 56 rows of 64; the contracted kernels (fast_math) know no such windows.
static bool want_overlap(const wt_handle *h)
{
    if (h->dtype != WT_F32 || h->fast_math) return false;
    if (h->win_overlap >= 0) return h->win_overlap > 0;
    return (long)h->g.nxl * h->g.ny <= 7500000L;
}

// Classes, bounce codes and the unit lists of the current mask for windows of 64 * sites rows.  Everything but the
// cuts of the column ranges runs on the device; the host reads nwin x (nxl+2) class bytes back.
static int build_fuse_plan(wt_handle *h, int sites, long target, int depth)
{
    const Geom &g = h->g;
    const int win = 64 * sites;
    const size_t eb = h->dtype == WT_F32 ? 4 : 8;
    h->ovl = depth >= 3 && want_overlap(h);
    const int wstride = h->ovl ? win - 8 : win, woff = h->ovl ? -4 : 0;
    const int nwin = march_nwin(g.ny, wstride);
    const size_t wbytes = (size_t)nwin * (g.nxl + 2), cbytes = (size_t)(g.nxl + 2) * g.pitch;
    if (h->n_win != nwin || h->march_s != sites || h->march_depth != depth) free_march_tables(h);
    long long added = 0;
    if (!h->wcls) { HIP_TRY(hipMalloc((void **)&h->wcls, wbytes)); added += (long long)wbytes; }
    if (depth < 3 && !h->halo_tab) {      // the two-step kernel's table (k_halo_rows / k_halo_from_seams)
        const size_t hbytes = (size_t)(nwin + 1) * (g.nxl + 2) * 8 * eb;
        HIP_TRY(hipMalloc(&h->halo_tab, hbytes));
        HIP_TRY(hipMemsetAsync(h->halo_tab, 0, hbytes, h->s_compute));
        added += (long long)hbytes;
    }
    if (!h->seams) {
        const size_t sbytes = (size_t)(nwin + 1) * (g.nxl + 2) * (depth >= 3 ? M3_SREC : 48) * eb;
        HIP_TRY(hipMalloc(&h->seams, sbytes));
        HIP_TRY(hipMemsetAsync(h->seams, 0, sbytes, h->s_compute));
        h->seams_valid = false;
        added += (long long)sbytes;
    }
    if (!h->seam_plain && nwin > 1) { HIP_TRY(hipMalloc((void **)&h->seam_plain, (size_t)(nwin - 1) * g.nxl)); added += (long long)(nwin - 1) * g.nxl; }
    if (depth >= 3 && !h->hlines) {       // k_march3's halo lines: zeroed once — the slots nobody writes (the lattice's first / last column, the
        const size_t hbytes = (size_t)(nwin + 1) * (g.nxl + 2) * M3_HL * eb;       // bottom / top window) must stay finite don't-cares
        HIP_TRY(hipMalloc(&h->hlines, hbytes));
        HIP_TRY(hipMemsetAsync(h->hlines, 0, hbytes, h->s_compute));
        added += (long long)hbytes;
    }
    h->march_table_bytes += added;
    h->device_bytes += added;
    if (!h->bcode) {
        HIP_TRY(hipMalloc((void **)&h->bcode, cbytes));
        HIP_TRY(hipMemsetAsync(h->bcode, 0, cbytes, h->s_compute));
        h->device_bytes += (long long)cbytes;
    }
    if (!h->d_nbad) HIP_TRY(hipMalloc((void **)&h->d_nbad, 2 * sizeof(unsigned int)));
    h->n_win = nwin;
    h->march_s = sites;
    h->march_depth = depth;
    const long nt = (long)(g.nxl + 2) * nwin;
    hipLaunchKernelGGL(k_classify_windows, dim3((unsigned)((nt + 3) / 4)), dim3(256), 0, h->s_compute, (const uint8_t *)h->mask, h->wcls, g, nwin, win, wstride, woff);
    hipLaunchKernelGGL(k_bounce_codes, dim3(2048), dim3(256), 0, h->s_compute, (const uint8_t *)h->mask, h->bcode, g);
    if (nwin > 1 && !h->ovl) {
        const long nth = (long)(nwin - 1) * g.nxl;
        if (depth == 4)
            hipLaunchKernelGGL(k_seam_flags4, dim3((unsigned)((nth + 255) / 256)), dim3(256), 0, h->s_compute, (const uint8_t *)h->mask, (const uint8_t *)h->bcode,
                               h->seam_plain, g, nwin, win);
        else if (depth == 3)
            hipLaunchKernelGGL(k_seam_flags3, dim3((unsigned)((nth + 255) / 256)), dim3(256), 0, h->s_compute, (const uint8_t *)h->mask, (const uint8_t *)h->bcode,
                               h->seam_plain, g, nwin, win);
        else
            hipLaunchKernelGGL(k_seam_flags, dim3((unsigned)((nth + 255) / 256)), dim3(256), 0, h->s_compute, (const uint8_t *)h->mask, (const uint8_t *)h->bcode,
                               h->seam_plain, g, nwin, win);
    }
    HIP_TRY(hipGetLastError());
    h->host_wcls.resize(wbytes);
    HIP_TRY(hipMemcpyAsync(h->host_wcls.data(), h->wcls, wbytes, hipMemcpyDeviceToHost, h->s_compute));
    HIP_TRY(hipStreamSynchronize(h->s_compute));

    h->plan_target = target;
    h->plan_tuned = false;
    h->colw_kept.clear();
    MarchPlan pl;
    cut_units(h, nullptr, &pl);
    WT_TRY(upload_units(h, pl));
    const MarchRange r = depth >= 3 ? march_range3(g, depth) : march_range(g);
    for (int w = 0; w < nwin; w++)
        for (int x = r.i_begin; x < r.i_end; x++) h->nonfast_tiles += h->host_wcls[(size_t)w * (g.nxl + 2) + x + 1] != WC_FAST;
    return WT_OK;
}

// Whole resident rounds of units (see build_march_plan): two rounds where that leaves at least 12 columns per unit
// (4096^2: 4096 units of 16 columns; the two re-read halo columns of a unit are then 12 % of its loads), else one.
// Returns 0 when the plan would not pay: with fewer than one round of min_cols-column units the two redundant step-1
// columns of every unit outweigh the saved traffic (measured with 2 sites per lane on 4096 rows: 288 columns = 4.5 per
// unit 18.7 us/step against 21.2 for single steps; 544 columns 25.1 against 33.5).
static long march_target_units(const wt_handle *h, int sites, long slots, bool force, int min_cols)
{
    const MarchRange r = march_range(h->g);
    const long tiles = (long)(r.i_end - r.i_begin) * march_nwin(h->g.ny, 64 * sites);
    if (!force && h->fuse_chunk <= 0 && tiles / slots < min_cols) return 0;      // columns per unit in ONE round
    long target = 2 * slots;
    if (tiles / target < 12) target = slots;
    return target;
}

// Steps per pass.  Every marching kernel holds 8-byte vectors per lane and direction (fp32: 2 sites per lane, 128-row windows; fp64: 1
// site, 64-row windows).  THREE steps per pass (step_march3.hpp) where eligible, FOUR where the units are long; fuse_depth = 2 selects two
// steps per pass: fp32 the two-step kernel of step_march.hpp on its own (smaller) tables — also the automatic choice for lattices too
// narrow for three steps to pay —, fp64 two-step passes of the three-step kernel on its tables.  (Round 2 also shipped 16-byte-vector
// instantiations of the two-step kernel — fp32 with 4 sites per lane, fp64 with 2; they spilled registers, were never the automatic
// choice on any lattice and are gone.)  The choice depends on the geometry only, so a mask change rebuilds the tables in place.
static int rebuild_fuse_plan(wt_handle *h)
{
    h->fuse_ready = false;
    h->n_units = h->nonfast_tiles = 0;
    h->pass_cap = 0;
    if (!h->fuse || !fuse_eligible(h) || !h->mask_set) return WT_OK;
    const long slots = h->wave_slots;
    const int s3 = h->dtype == WT_F32 ? 2 : 1;
    const bool two_on_three = h->dtype != WT_F32 && h->fuse_depth == 2;  // fp64: two-step passes on the three-step tables
    // Every slab of a tunnel must take the SAME sequence of passes and refresh steps (the exchange is collective: over RCCL each rank
    // decides on its own), so the automatic choices below look at the NARROWEST slab of the split — an edge slab, W + halo columns —
    // whatever this slab's own width is: a quantity every rank computes alike from the split and the halo depth.
    const int plan_nxl = h->plan_columns > 0 ? h->plan_columns : (h->nranks > 1 ? h->min_width + h->halo : h->g.nxl);
    const bool depth3_ok = std::min(h->g.nxl, plan_nxl) >= 16 && (h->fuse_depth != 2 || two_on_three);
    if (depth3_ok) {
        // Steps per pass by columns per resident unit (tools/r3_depth_at_widths.py on the kernels of round 3 — chain blocks, units cut by measured
        // time; 4096 rows, us per step as single steps / two / three / four per pass, a plain slab | a slab over the thick part of the body):
        //   fp32  64 columns (0 per unit)   9.8 /  9.8 /  8.8 /  9.8 | 10.3 / 10.3 / 10.4 / 12.4
        //        100 (1)                   13.2 / 13.2 /  9.3 / 10.1 | 13.2 / 13.2 / 12.2 / 14.8
        //        200 (3)                   14.3 / 14.2 / 10.2 / 10.4 | 15.9 / 16.0 / 13.3 / 15.8
        //        300 (4)                   18.9 / 14.5 / 12.2 / 11.6 | 20.8 / 17.3 / 14.8 / 16.5
        //        340 (5)                   21.5 / 15.5 / 13.2 / 11.7 | 23.5 / 18.1 / 16.3 / 17.0
        //        420 (6)                   24.6 / 17.6 / 14.7 / 13.2 | 26.6 / 20.9 / 18.8 / 18.7
        //        500 (7)                   29.1 / 19.9 / 16.2 / 14.5 | 31.1 / 23.8 / 20.3 / 21.4      (544 and up: run_width_sweep.sh, four)
        //   fp64  64 (1)                   12.6 / 10.9 /  9.6 / 10.9 | 12.6 / 15.2 / 14.3 / 15.5
        //        150 (4)                   18.3 / 15.8 / 12.5 / 13.7 | 18.7 / 19.3 / 16.6 / 17.7
        //        300 (9)                   30.8 / 26.6 / 19.1 / 19.4 | 31.1 / 31.3 / 22.7 / 23.8
        //        400 (12)                  39.9 / 34.9 / 24.1 / 23.0 | 41.1 / 37.8 / 27.7 / 26.7
        // Whole tunnels with their body inside (tools/r3_small_lattices.py: 2048x1024, 7 per unit: 32.3 / - / 21.7 / 23.5; 1024x1024, 3 per unit:
        // 18.5 / - / 15.3 / 17.1; 1024x512, 1 per unit: 13.6 / - / 13.4 / 16.1; 512x256, 0 per unit: 9.9 / - / 11.1 / 13.5) follow the body columns.
        // fp32: three steps per pass from one column per unit up, four from eight — from five for the slabs of a split, which are mostly plain
        // and are cut by cost (the widest, plain ones set the pace); fp64: four from 12, three from four; single steps below.
        // (Round 2's rule — two steps per pass below eight columns per unit — predates the chain blocks: the two-step kernel is never the
        // best choice any more and runs only when fuse_depth = 2 asks for it.)
        const long tiles3 = (long)(plan_nxl - 4) * march_nwin(h->g.ny, 64 * s3);
        const bool f32 = h->dtype == WT_F32;
        const long cpu = tiles3 / slots;
        static const long min4_env = exp_env("WT_DEPTH4_MIN") ? atol(exp_env("WT_DEPTH4_MIN")) : 0;      // experiments
        // (whole lattices on overlapping windows — want_overlap: the small ones — pay no halo kernel per pass: four steps from four columns per unit,
        //  1536x768 11.8 against 12.6 us per step, 2048x1024 14.5 against 16.8; 1024x512, two per unit, stays at three: 9.2 against 10.0)
        const long min4 = min4_env > 0 ? min4_env : ((h->nranks > 1 || h->plan_columns > 0) ? 5 : (want_overlap(h) ? 4 : 8));
        static const long min3 = exp_env("WT_DEPTH3_MIN") ? atol(exp_env("WT_DEPTH3_MIN")) : 1;
        const bool force = h->fuse_force || h->fuse_depth >= 2;
        if (!force && h->fuse_chunk <= 0 && cpu < (f32 ? min3 : 4)) return WT_OK;
        int depth = h->fuse_depth == 4 || (h->fuse_depth == 0 && cpu >= (f32 ? min4 : 12)) ? 4 : 3;
        // The units of a depth-D plan leave the D-1 columns next to a local slab edge unwritten: those must all be GHOST columns, so a slab
        // with fewer than D-1 of them plans shallower (halo 2: three steps per pass at most; halo 1: the two-step kernel, whose units leave one).
        if (h->nranks > 1 && h->halo < depth - 1) depth = h->halo + 1;
        if (depth < 3) { if (f32) goto two_step; return WT_OK; }
        const MarchRange r = march_range3(h->g, depth);
        const long tiles = (long)(r.i_end - r.i_begin) * march_nwin(h->g.ny, 64 * s3);
        {
            // ONE resident round of units with chain blocks (their units can be as long as the lattice asks for; tools/r3_rounds.sh: 2080 columns 52.9
            // against 55.8 us per step with two rounds of half the length, 3000^2 56.7 / 60.5, 4096^2 90.4 / 92.9); without them two rounds
            // where that leaves at least 12 columns per unit, as in round 2
            long target = h->chain && h->fuse_chunk <= 0 ? slots : 2 * slots;
            if (tiles / target < 12) target = slots;
            if (const char *e = exp_env("WT_MARCH_ROUNDS")) { if (atoi(e) > 0) target = atoi(e) * slots; }      // experiments
            if (!h->chain || h->fuse_chunk > 0)
                while (tiles / target > MARCH3_MAX_CHUNK - 6) target += slots;     // a solo unit holds at most MARCH3_MAX_CHUNK columns: more rounds
            WT_TRY(build_fuse_plan(h, s3, target, depth));
            h->fuse_ready = h->n_units > 0;
            h->pass_cap = two_on_three ? 2 : 0;
            return WT_OK;
        }
    }
two_step:
    if (h->fuse_depth >= 3 || h->dtype != WT_F32) return WT_OK;
    // fp32, two steps per pass on 128-row windows (measured with 4096 rows: 288 columns = 4.5 per unit 16.6 us/step against 20.9 for
    // single steps and 18.5 for three steps per pass)
    if (!h->fuse_force && h->fuse_chunk <= 0 && (long)(plan_nxl - 2) * march_nwin(h->g.ny, 128) / slots < 4) return WT_OK;     // (the narrowest slab decides)
    const long target = march_target_units(h, 2, slots, true, 4);
    if (target == 0) return WT_OK;
    WT_TRY(build_fuse_plan(h, 2, target, 2));
    h->fuse_ready = h->n_units > 0;
    return WT_OK;
}

#ifdef WT_M3_STAMPS          // diagnostic build only (tools/m3_stamps.py); not part of the ABI
extern "C" WT_API int wt_debug_m3_stamps(unsigned long long *out, int reset)
{
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(wt::g_m3_stamps), 8 * sizeof(unsigned long long)));
    if (reset) { unsigned long long z[8] = {0}; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(wt::g_m3_stamps), z, sizeof(z))); }
    return WT_OK;
}
#endif

#ifdef WT_UNIT_CLOCKS         // diagnostic build only (tools/unit_clocks.py); not part of the ABI
extern "C" WT_API int wt_debug_halo_clocks(unsigned long long *out, int reset)
{
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(wt::g_halo_clk), 8 * sizeof(unsigned long long)));
    if (reset) { unsigned long long z[8] = {0}; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(wt::g_halo_clk), z, sizeof(z))); }
    return WT_OK;
}
extern "C" WT_API int wt_debug_unit_clocks(wt_handle *h, unsigned long long *clk, int *units4, int cap)
{
    WT_TRY(check_handle(h));
    HIP_TRY(hipDeviceSynchronize());
    const int n = h->n_units < cap ? h->n_units : cap;
    if (!h->d_clk) return fail(WT_ERR_STATE, "no pass has recorded unit clocks yet");
    HIP_TRY(hipMemcpy(clk, h->d_clk, (size_t)2 * n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(units4, h->d_units, (size_t)n * sizeof(MarchUnit), hipMemcpyDeviceToHost));
    return n;
}
#endif

// Exchange timing: slot k of the ring holds the events of one refresh step — x0 / x1 around the exchange on the comm stream, i0 / i1 around the
// interior kernel and w behind the wait for the exchange on the compute stream.  exposed = i1 -> w: what the compute stream waited for the
// exchange after its interior kernel (a few microseconds of event overhead even when the exchange hid completely).
static inline hipEvent_t xt_event(wt_handle *h, int which) { return h->xt_ev[(size_t)(h->xt_n % XT_RING) * 5 + which]; }
static int xt_resolve(wt_handle *h)
{
    if (h->xt_ev.empty() || h->xt_n == 0) return WT_OK;
    HIP_TRY(hipSetDevice(h->device));
    for (int k = 0; k < h->xt_n; k++) {
        hipEvent_t *e = &h->xt_ev[(size_t)k * 5];
        float x = 0.0f, in = 0.0f, w = 0.0f;
        HIP_TRY(hipEventSynchronize(e[1]));
        HIP_TRY(hipEventSynchronize(e[4]));
        HIP_TRY(hipEventElapsedTime(&x, e[0], e[1]));
        HIP_TRY(hipEventElapsedTime(&in, e[2], e[3]));
        HIP_TRY(hipEventElapsedTime(&w, e[3], e[4]));
        h->xt_exchange_ms += x; h->xt_interior_ms += in; h->xt_exposed_ms += w;
        h->xt_count += 1;
    }
    h->xt_n = 0;
    return WT_OK;
}

extern "C" int wt_set_option(wt_handle *h, const char *name, double value)
{
    WT_TRY(check_handle(h));
    if (!name) return fail(WT_ERR_ARG, "option name is null");
    HIP_TRY(hipSetDevice(h->device));
    if (strcmp(name, "agree_check") == 0) {
        // slab handles: check that every slab of the tunnel takes the same schedule (agree_rccl / agree_local); 0 only for tests that mix on purpose
        h->agree_check = value != 0.0;
        h->agree_dirty = true;
        return WT_OK;
    }
    if (strcmp(name, "exchange_timing") == 0) {
        // slab handles: HIP events around every ghost exchange and the interior kernel beside it (reported by "exchange_ms", "interior_ms",
        // "exchange_exposed_ms", "exchanges"); off by default — the events cost a few microseconds per refresh
        h->xt_on = value != 0.0;
        if (h->xt_on && h->xt_ev.empty()) {
            h->xt_ev.assign((size_t)XT_RING * 5, nullptr);
            for (hipEvent_t &e : h->xt_ev) HIP_TRY(hipEventCreate(&e));
        }
        h->xt_n = 0; h->xt_count = 0;
        h->xt_exchange_ms = h->xt_interior_ms = h->xt_exposed_ms = 0.0;
        return WT_OK;
    }
    h->agree_dirty = true;               // any other option may change the schedule: the next stepping call of a slab handle checks again
    if (strcmp(name, "fuse_steps") == 0) {
        if (value != 0.0 && !fuse_eligible(h))
            return fail(WT_ERR_STATE, "fuse_steps needs at least 8 local columns, a lattice below 4 GiB and (fp32) an even NY");
        h->fuse = value != 0.0;
        h->fuse_force = value >= 2.0;
        return rebuild_fuse_plan(h);
    }
    if (strcmp(name, "fuse_sites") == 0) {
        // kept for callers of round 2: the sites per lane are fixed by the element type now (see rebuild_fuse_plan)
        const int s3 = h->dtype == WT_F32 ? 2 : 1;
        if (!(value == 0.0 || value == (double)s3))
            return fail(WT_ERR_ARG, "fuse_sites must be 0 (automatic) or %d for this handle (the 16-byte-vector kernels of round 2 are gone)", s3);
        h->fuse_sites = (int)value;
        return WT_OK;
    }
    if (strcmp(name, "fuse_depth") == 0) {
        if (!(value == 0.0 || value == 2.0 || value == 3.0 || value == 4.0)) return fail(WT_ERR_ARG, "fuse_depth must be 0 (automatic), 2, 3 or 4");
        if (value >= 3.0 && !(fuse_eligible_s(h, h->dtype == WT_F32 ? 2 : 1) && h->g.nxl >= 16))
            return fail(WT_ERR_STATE, "fuse_depth 3 / 4 needs at least 16 local columns, a lattice below 4 GiB and (fp32) an even NY");
        h->fuse_depth = (int)value;
        return rebuild_fuse_plan(h);
    }
    if (strcmp(name, "fuse_chunk") == 0) {
        if (!(value >= 0.0 && value <= 4096.0)) return fail(WT_ERR_ARG, "fuse_chunk out of range (0 = automatic)");
        h->fuse_chunk = (int)value;
        return rebuild_fuse_plan(h);
    }
    if (strcmp(name, "fast_div") == 0) {
        h->fast_div = value != 0.0;
        return WT_OK;
    }
    if (strcmp(name, "selftest_tau") == 0) {
        if (!(value > 0.0) || !std::isfinite(value)) return fail(WT_ERR_ARG, "selftest_tau must be positive and finite");
        h->selftest_tau = value;
        return WT_OK;
    }
    if (strcmp(name, "fast_div_two_op") == 0) {       // rank-local (the same bits either way): not part of the cross-rank fingerprint
        h->two_op = value != 0.0;
        return WT_OK;
    }
    if (strcmp(name, "chain") == 0) {
        h->chain = value != 0.0;
        return rebuild_fuse_plan(h);
    }
    if (strcmp(name, "window_overlap") == 0) {
        // -1 automatic, 0 windows that tile the column (halo lines), 1 overlapping windows (k_march3).  The same bits either way, and the sequence of
        // passes does not depend on it (the steps per pass are chosen from the tiling windows' count): a rank's own affair, like `tune`.
        if (!(value == -1.0 || value == 0.0 || value == 1.0)) return fail(WT_ERR_ARG, "window_overlap must be -1 (automatic), 0 or 1");
        h->win_overlap = (int)value;
        return rebuild_fuse_plan(h);
    }
    if (strcmp(name, "plan_columns") == 0) {
        // a stand-alone handle that stands in for one slab of a split plans like that split's narrowest slab (distributed.measure_slab_cost)
        if (value < 0) return fail(WT_ERR_ARG, "plan_columns must be >= 0");
        h->plan_columns = (int)value;
        return rebuild_fuse_plan(h);
    }
    if (strcmp(name, "tune") == 0) {
        // measured refinement of the marching units before the first pass on a plan (tune_fuse_plan); 0 keeps the modelled cut
        h->tune = value != 0.0;
        return h->plan_tuned ? rebuild_fuse_plan(h) : WT_OK;
    }
    if (strcmp(name, "trim_ghosts") == 0) {
        h->trim = value != 0.0;
        return WT_OK;
    }
    if (strcmp(name, "refresh") == 0) {
        if (!(value == 0.0 || value == 1.0 || value == 2.0))
            return fail(WT_ERR_ARG, "refresh must be 0 (overlapped single step), 1 (exchange at a pass boundary) or 2 (exchange beside the interior of a fused pass)");
        h->refresh_mode = (int)value;
        h->trim_prebuilt = false;        // (mode 2's unit lists are cut with the others before the next stepping call)
        return WT_OK;
    }
    if (strcmp(name, "fast_math") == 0) {
        if (value != 0.0 && h->dtype != WT_F32) return fail(WT_ERR_STATE, "fast_math is an fp32 option");
        const bool was = h->fast_math;
        h->fast_math = value != 0.0;
        // (the contracted kernels know no overlapping windows: a plan that has them is cut again, and the other way round)
        return (was != h->fast_math && (h->ovl || h->win_overlap != 0)) ? rebuild_fuse_plan(h) : WT_OK;
    }
    return fail(WT_ERR_ARG, "unknown option '%s'", name);
}

extern "C" int wt_get_option(const wt_handle *h, const char *name, double *value)
{
    WT_TRY(check_handle(h));
    if (!name || !value) return fail(WT_ERR_ARG, "null argument");
    if (strcmp(name, "fuse_steps") == 0) { *value = h->fuse ? (h->fuse_force ? 2.0 : 1.0) : 0.0; return WT_OK; }
    if (strcmp(name, "fuse_active") == 0) { *value = h->fuse_ready ? 1.0 : 0.0; return WT_OK; }
    if (strcmp(name, "fuse_chunk") == 0) { *value = h->fuse_ready ? h->fuse_chunk_used : h->fuse_chunk; return WT_OK; }
    if (strcmp(name, "fuse_units") == 0) { *value = h->n_units; return WT_OK; }
    if (strcmp(name, "fuse_sites") == 0) { *value = h->fuse_ready ? h->march_s : h->fuse_sites; return WT_OK; }
    if (strcmp(name, "fuse_depth") == 0) { *value = h->fuse_ready ? plan_depth(h) : h->fuse_depth; return WT_OK; }
    if (strcmp(name, "fuse_tiles_general") == 0) { *value = h->nonfast_tiles; return WT_OK; }   // window-tiles that take the body paths
    if (strcmp(name, "fast_div") == 0) { *value = h->fast_div ? 1.0 : 0.0; return WT_OK; }
    if (strcmp(name, "chain") == 0) { *value = h->chain ? 1.0 : 0.0; return WT_OK; }
    if (strcmp(name, "window_overlap") == 0) { *value = h->fuse_ready ? (h->ovl ? 1.0 : 0.0) : (double)h->win_overlap; return WT_OK; }     // the present plan's layout
    if (strcmp(name, "fast_math") == 0) { *value = h->fast_math ? 1.0 : 0.0; return WT_OK; }
    if (strcmp(name, "chain_units") == 0) { *value = h->fuse_ready ? h->n_chain_units : 0; return WT_OK; }     // units that run in chain blocks
    if (strcmp(name, "plan_columns") == 0) { *value = h->plan_columns; return WT_OK; }
    if (strcmp(name, "tune") == 0) { *value = h->tune ? 1.0 : 0.0; return WT_OK; }
    if (strcmp(name, "tune_rounds") == 0) { *value = h->plan_tuned ? h->tune_rounds : 0; return WT_OK; }        // plans measured for the present mask
    if (strcmp(name, "tune_gain") == 0) { *value = h->plan_tuned ? h->tune_gain : 0.0; return WT_OK; }          // makespan modelled plan / kept plan
    if (strcmp(name, "single_steps") == 0) { *value = (double)h->single_steps; return WT_OK; }   // whole steps taken by k_step since the last init / write_f
    if (strcmp(name, "passes") == 0) { *value = (double)h->passes; return WT_OK; }
    if (strcmp(name, "trim_ghosts") == 0) { *value = h->trim ? 1.0 : 0.0; return WT_OK; }
    if (strcmp(name, "refresh") == 0) { *value = h->refresh_mode; return WT_OK; }
    if (strcmp(name, "fused_renewals") == 0) { *value = (double)h->fused_renewals; return WT_OK; }
    if (strcmp(name, "boundary_exchanges") == 0) { *value = (double)h->boundary_exchanges; return WT_OK; }
    if (strcmp(name, "trimmed_passes") == 0) { *value = (double)h->trimmed_passes; return WT_OK; }
    if (strcmp(name, "pass_depth") == 0) { *value = h->fuse_ready ? eff_depth(h) : 0; return WT_OK; }       // steps a full pass takes for the tau of the last stepping call
    if (strcmp(name, "chain_downgrades") == 0) { *value = h->chain_downgrades; return WT_OK; }
    if (strcmp(name, "agree_check") == 0) { *value = h->agree_check ? 1.0 : 0.0; return WT_OK; }
    if (strcmp(name, "agree_checks") == 0) { *value = (double)h->agree_checks; return WT_OK; }
    if (strcmp(name, "comm_ranks") == 0) { *value = h->comm_ranks; return WT_OK; }
    if (strcmp(name, "wave_slots") == 0) { *value = (double)h->wave_slots; return WT_OK; }
    if (strcmp(name, "exchange_timing") == 0) { *value = h->xt_on ? 1.0 : 0.0; return WT_OK; }
    if (strcmp(name, "exchanges") == 0 || strcmp(name, "exchange_ms") == 0 || strcmp(name, "interior_ms") == 0 || strcmp(name, "exchange_exposed_ms") == 0) {
        if (xt_resolve(const_cast<wt_handle *>(h)) != WT_OK) return WT_ERR_HIP;
        *value = strcmp(name, "exchanges") == 0 ? (double)h->xt_count : strcmp(name, "exchange_ms") == 0 ? h->xt_exchange_ms
                 : strcmp(name, "interior_ms") == 0 ? h->xt_interior_ms : h->xt_exposed_ms;
        return WT_OK;
    }
    if (strcmp(name, "fast_div_active") == 0) {
        // fp32: the proof accepted the tau of the last stepping call; fp64: the four-operation sequence is on (a theorem for every tau, d2q9.hpp)
        if (h->dtype != WT_F32) { *value = h->fast_div ? 1.0 : 0.0; return WT_OK; }
        *value = (h->fd_checked && h->fd_ok && h->fast_div) ? 1.0 : 0.0;
        return WT_OK;
    }
    if (strcmp(name, "fast_div_two_op") == 0) { *value = h->two_op ? 1.0 : 0.0; return WT_OK; }
    if (strcmp(name, "canvas_scale") == 0) { *value = h->cv_scale; return WT_OK; }                    // 0: no canvas allocated yet
    if (strcmp(name, "canvas_text_set") == 0) { *value = h->cv_text_set ? 1.0 : 0.0; return WT_OK; }  // the label map of the present canvas has been uploaded
    if (strcmp(name, "canvas_layer_live") == 0) { *value = h->cv_layer_live ? 1.0 : 0.0; return WT_OK; }
    if (strncmp(name, "selftest_fastdiv", 16) == 0) {
        // mismatches of the fast divisions by "selftest_tau" against the IEEE quotient, counted on the device (tests/test_gpu_fastdiv.py):
        //   selftest_fastdiv32_3 / _2: the three- / two-operation binary32 forms over all 2^23 significands and both signs (the library's own proof);
        //   selftest_fastdiv64: the four-operation binary64 form on 2^28 pseudo-random and boundary-hugging numerators
        wt_handle *hm = const_cast<wt_handle *>(h);
        const FastDiv fdv = make_fastdiv(strcmp(name, "selftest_fastdiv64") == 0 ? h->selftest_tau : (double)(float)h->selftest_tau);
        if (hipSetDevice(h->device) != hipSuccess) return fail(WT_ERR_HIP, "hipSetDevice");
        if (!hm->d_nbad && hipMalloc((void **)&hm->d_nbad, 2 * sizeof(unsigned int)) != hipSuccess) return fail(WT_ERR_HIP, "selftest: hipMalloc");
        if (hipMemsetAsync(hm->d_nbad, 0, 2 * sizeof(unsigned int), h->s_compute) != hipSuccess) return fail(WT_ERR_HIP, "selftest: memset");
        const bool f64 = strcmp(name, "selftest_fastdiv64") == 0;
        if (f64) hipLaunchKernelGGL(k_check_fastdiv64, dim3(1024), dim3(256), 0, h->s_compute, fdv, 0x5eedULL, 1024, hm->d_nbad);
        else if (strcmp(name, "selftest_fastdiv32_3") == 0 || strcmp(name, "selftest_fastdiv32_2") == 0)
            hipLaunchKernelGGL(k_verify_fastdiv, dim3(1024), dim3(256), 0, h->s_compute, fdv, hm->d_nbad);
        else return fail(WT_ERR_ARG, "unknown option '%s'", name);
        unsigned int nbad[2] = {1, 1};
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(nbad, hm->d_nbad, sizeof(nbad), hipMemcpyDeviceToHost, h->s_compute) != hipSuccess ||
            hipStreamSynchronize(h->s_compute) != hipSuccess) return fail(WT_ERR_HIP, "selftest: launch");
        *value = (double)nbad[(!f64 && name[strlen(name) - 1] == '2') ? 1 : 0];
        return WT_OK;
    }
    if (strcmp(name, "fast_div_two_op_active") == 0) { *value = (h->dtype == WT_F32 && h->fd_checked && h->fd_ok && h->fd2_ok && h->fast_div && h->two_op) ? 1.0 : 0.0; return WT_OK; }
    return fail(WT_ERR_ARG, "unknown option '%s'", name);
}

// ------------------------------------------------------------------------------------------
// mask
// ------------------------------------------------------------------------------------------
__global__ void k_mask_rows_to_cols(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int ncols, int ny,
                                    long pitch, long ld)
{
    // src[j*ld + x], x in [0,ncols) -> dst[x*pitch + j] as 0/1
    __shared__ uint8_t tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int j = by + r, x = bx + threadIdx.x;
        if (x < ncols && j < ny) tile[r][threadIdx.x] = src[(long)j * ld + x] ? 1 : 0;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int x = bx + r, j = by + threadIdx.x;
        if (x < ncols && j < ny) dst[(long)x * pitch + j] = tile[threadIdx.x][r];
    }
}

extern "C" int wt_set_mask(wt_handle *h, const uint8_t *mask)
{
    WT_TRY(check_handle(h));
    if (!mask) return fail(WT_ERR_ARG, "mask is null");
    HIP_TRY(hipSetDevice(h->device));
    const Geom &g = h->g;
    // local columns -1 .. nxl  <->  global columns gi0-1 .. gi0+nxl, clipped to the tunnel
    const int glo = g.gi0 - 1, ghi = g.gi0 + g.nxl;                  // inclusive
    const int clo = glo < 0 ? 0 : glo, chi = ghi > g.nx_g - 1 ? g.nx_g - 1 : ghi;
    const int ncols = chi - clo + 1;
    WT_TRY(ensure_stage(h, (size_t)ncols * g.ny));
    HIP_TRY(hipStreamSynchronize(h->s_compute));
    HIP_TRY(hipMemcpy2D(h->stage, (size_t)ncols, mask + clo, (size_t)g.nx_g, (size_t)ncols, (size_t)g.ny,
                        hipMemcpyHostToDevice));
    HIP_TRY(hipMemsetAsync(h->mask, 0, (size_t)(g.nxl + 2) * g.pitch, h->s_compute));
    dim3 blk(32, 8), grd((ncols + 31) / 32, (g.ny + 31) / 32);
    // destination: local column (clo - g.gi0) lives at row (clo - g.gi0 + 1) of the padded mask
    uint8_t *dst = h->mask + (long)(clo - g.gi0 + 1) * g.pitch;
    hipLaunchKernelGGL(k_mask_rows_to_cols, grd, blk, 0, h->s_compute, (const uint8_t *)h->stage, dst, ncols, g.ny,
                       g.pitch, (long)ncols);
    HIP_TRY(hipGetLastError());
    WT_TRY(classify_tiles(h->mask, h->tiles, g, h->tiles_per_col, h->s_compute));
    HIP_TRY(hipStreamSynchronize(h->s_compute));
    h->mask_set = true;
    h->agree_dirty = true;
    h->tune_deferred = h->masks_set > 0 && h->passes_total - h->passes_at_mask < TUNE_LIVE_PASSES;
    h->passes_at_mask = h->passes_total;
    h->masks_set += 1;
    WT_TRY(rebuild_fuse_plan(h));
    return WT_OK;
}

// ------------------------------------------------------------------------------------------
// init
// ------------------------------------------------------------------------------------------
template <typename T>
static int init_impl(wt_handle *h, double u0)
{
    // html:474-490: JS doubles, rounded to the storage type
    const double w0 = 4.0 / 9.0, ws = 1.0 / 9.0, wd = 1.0 / 36.0;
    Init9<T> iv;
    for (int k = 0; k < 9; k++) {
        const double w = (k == 0) ? w0 : (k <= 4 ? ws : wd);
        const double eu = ex_of(k) * u0, uu = u0 * u0;
        iv.v[k] = (T)(w * (1 + 3 * eu + 4.5 * eu * eu - 1.5 * uu));
    }
    iv.u0 = (T)u0;
    hipLaunchKernelGGL(k_fill_equilibrium<T>, dim3(2048), dim3(256), 0, h->s_compute, fptr<T>(h, 0), fptr<T>(h, 1),
                       reinterpret_cast<T *>(h->macro), h->g, iv);
    HIP_TRY(hipGetLastError());
    return WT_OK;
}

extern "C" int wt_init_equilibrium(wt_handle *h, double u0)
{
    WT_TRY(check_handle(h));
    HIP_TRY(hipSetDevice(h->device));
    WT_TRY(h->dtype == WT_F32 ? init_impl<float>(h, u0) : init_impl<double>(h, u0));
    h->cur = 0;
    h->inited = true;
    if (h->stuck_host) { HIP_TRY(hipStreamSynchronize(h->s_compute)); *h->stuck_host = 0; }
    h->macro_stale = false;
    h->seams_valid = false;
    h->steps_done = 0;
    h->single_steps = 0;
    h->passes = 0;
    h->agree_dirty = true;
    h->ghost_valid = h->halo;     // a uniform state is exact everywhere, ghosts included
    return WT_OK;
}

// ------------------------------------------------------------------------------------------
// stepping
// ------------------------------------------------------------------------------------------
template <typename T>
static int launch_step(wt_handle *h, int i_begin, int i_end, double tau, double u0, bool emit, hipStream_t st)
{
    if (i_end <= i_begin) return WT_OK;
    return step_columns<T>(fptr<T>(h, h->cur), fptr<T>(h, 1 - h->cur), reinterpret_cast<T *>(h->macro), h->mask,
                           h->tiles, h->tiles_per_col, h->g, i_begin, i_end, (T)tau, (T)u0, emit, (int)(h->steps_done & 1), st);
}

static int launch_step_any(wt_handle *h, int i_begin, int i_end, double tau, double u0, bool emit, hipStream_t st)
{
    int rc = h->dtype == WT_F32 ? launch_step<float>(h, i_begin, i_end, tau, u0, emit, st)
                                : launch_step<double>(h, i_begin, i_end, tau, u0, emit, st);
    if (rc != WT_OK) return fail(rc, "step kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
    return WT_OK;
}

// byte ranges of `ncol` whole columns starting at local column `i` in population k of lattice `which`
static inline char *col_ptr(wt_handle *h, int which, int k, int i)
{
    return reinterpret_cast<char *>(h->f[which]) + ((size_t)k * h->g.plane + (size_t)(i + 1) * h->g.pitch) * h->esz;
}

// Refresh this slab's ghost columns from its neighbours' owned edge columns (lattice `cur`).
// TR_RCCL: grouped send/recv on the comm stream.  Ghost columns are contiguous per population
// (column-major layout), so the lattice itself is the send and the receive buffer.
static int exchange_rccl(wt_handle *h)
{
    const size_t count = (size_t)h->halo * h->g.pitch;   // elements per population per side
    const ncclDataType_t dt = h->dtype == WT_F32 ? ncclFloat32 : ncclFloat64;
    ncclResult_t rc = ncclGroupStart();
    for (int k = 0; k < 9 && rc == ncclSuccess; k++) {
        if (h->gl) {   // left neighbour: send my first `halo` owned columns, receive my left ghosts
            rc = ncclSend(col_ptr(h, h->cur, k, h->gl), count, dt, h->rank - 1, h->comm, h->s_comm);
            if (rc == ncclSuccess) rc = ncclRecv(col_ptr(h, h->cur, k, 0), count, dt, h->rank - 1, h->comm, h->s_comm);
        }
        if (h->gr && rc == ncclSuccess) {   // right neighbour: send my last `halo` owned columns, receive my right ghosts
            rc = ncclSend(col_ptr(h, h->cur, k, h->gl + h->width - h->halo), count, dt, h->rank + 1, h->comm, h->s_comm);
            if (rc == ncclSuccess) rc = ncclRecv(col_ptr(h, h->cur, k, h->gl + h->width), count, dt, h->rank + 1, h->comm, h->s_comm);
        }
    }
    const ncclResult_t rc_end = ncclGroupEnd();            // always close the group, even after a failure
    if (rc == ncclSuccess) rc = rc_end;
    if (rc != ncclSuccess) return fail(WT_ERR_RCCL, "ghost-column exchange failed: %s", ncclGetErrorString(rc));
    return WT_OK;
}

// TR_LOCAL: pull the ghost columns from the peers' lattices on MY comm stream — one copy kernel when the peer lives on the same
// device (it shows up in a kernel trace beside the interior step kernel; 18 separate hipMemcpyPeerAsync calls on one device cost far
// more than the data), peer copies when it does not.  The caller guarantees the peers' `cur` lattices are final (wt_step_group
// event-joins them).
struct GhostCopy {
    const char *src[2];          // side 0: left ghosts <- left peer's last owned columns; side 1: right ghosts <- right peer's first owned columns
    char *dst[2];
    size_t src_plane[2], dst_plane;      // bytes between two populations
    size_t bytes;                // per population and side
};
__global__ __launch_bounds__(256) void k_ghost_copy(GhostCopy c)
{
    const int k = blockIdx.y, side = blockIdx.z;
    if (!c.src[side]) return;
    const uint4 *s = reinterpret_cast<const uint4 *>(c.src[side] + (size_t)k * c.src_plane[side]);
    uint4 *d = reinterpret_cast<uint4 *>(c.dst[side] + (size_t)k * c.dst_plane);
    const size_t n = c.bytes / 16;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s[i];
}

static int exchange_local(wt_handle *h)
{
    const size_t bytes = (size_t)h->halo * h->g.pitch * h->esz;
    if ((h->gl && !h->peer_l) || (h->gr && !h->peer_r))
        return fail(WT_ERR_STATE, "a neighbouring slab of this locally linked group has been destroyed");
    const bool same_l = !h->gl || h->peer_l->device == h->device, same_r = !h->gr || h->peer_r->device == h->device;
    if (same_l && same_r) {
        GhostCopy c{};
        c.bytes = bytes;
        c.dst_plane = (size_t)h->g.plane * h->esz;
        if (h->gl) {
            wt_handle *p = h->peer_l;
            c.src[0] = col_ptr(p, p->cur, 0, p->gl + p->width - p->halo); c.dst[0] = col_ptr(h, h->cur, 0, 0); c.src_plane[0] = (size_t)p->g.plane * p->esz;
        }
        if (h->gr) {
            wt_handle *p = h->peer_r;
            c.src[1] = col_ptr(p, p->cur, 0, p->gl); c.dst[1] = col_ptr(h, h->cur, 0, h->gl + h->width); c.src_plane[1] = (size_t)p->g.plane * p->esz;
        }
        const unsigned nb = (unsigned)std::min<size_t>(64, (bytes / 16 + 255) / 256);
        hipLaunchKernelGGL(k_ghost_copy, dim3(nb ? nb : 1, 9, 2), dim3(256), 0, h->s_comm, c);
        HIP_TRY(hipGetLastError());
        return WT_OK;
    }
    for (int k = 0; k < 9; k++) {
        if (h->gl) {
            wt_handle *p = h->peer_l;
            HIP_TRY(hipMemcpyPeerAsync(col_ptr(h, h->cur, k, 0), h->device,
                                       col_ptr(p, p->cur, k, p->gl + p->width - p->halo), p->device, bytes, h->s_comm));
        }
        if (h->gr) {
            wt_handle *p = h->peer_r;
            HIP_TRY(hipMemcpyPeerAsync(col_ptr(h, h->cur, k, h->gl + h->width), h->device,
                                       col_ptr(p, p->cur, k, p->gl), p->device, bytes, h->s_comm));
        }
    }
    return WT_OK;
}

static int check_steppable(wt_handle *h, int nsteps, double tau, double u0)
{
    WT_TRY(check_handle(h));
    if (nsteps < 0) return fail(WT_ERR_ARG, "nsteps < 0");
    if (!(tau > 0.0) || !std::isfinite(tau)) return fail(WT_ERR_ARG, "tau must be positive and finite");
    if (!std::isfinite(u0)) return fail(WT_ERR_ARG, "u0 must be finite");
    if (!h->inited) return fail(WT_ERR_STATE, "wt_init_equilibrium (or wt_write_f) has not been called");
    if (!h->mask_set) return fail(WT_ERR_STATE, "wt_set_mask has not been called");
    return WT_OK;
}

// Enqueue the refresh of this slab's ghost columns (lattice `cur`) on its comm stream.
static int halo_begin(wt_handle *h)
{
    h->seams_valid = false;              // the refreshed ghost columns are not in the seam buffer
    HIP_TRY(hipEventRecord(h->ev_state, h->s_compute));
    HIP_TRY(hipStreamWaitEvent(h->s_comm, h->ev_state, 0));
    if (h->xt_on) {
        if (h->xt_n >= XT_RING) WT_TRY(xt_resolve(h));          // ring full: fold the finished refreshes into the sums (blocks on the oldest)
        HIP_TRY(hipEventRecord(xt_event(h, 0), h->s_comm));
    }
    if (h->transport == TR_RCCL) WT_TRY(exchange_rccl(h)); else WT_TRY(exchange_local(h));
    if (h->xt_on) HIP_TRY(hipEventRecord(xt_event(h, 1), h->s_comm));
    HIP_TRY(hipEventRecord(h->ev_halo, h->s_comm));
    return WT_OK;
}

static inline bool needs_halo(const wt_handle *h) { return h->nranks > 1 && h->ghost_valid <= 0; }

// refresh = 1: is an exchange due before the next step of `left` steps to go?  (A fused pass needs two exact ghost columns, a last single step one.)
static inline bool boundary_exchange_due(const wt_handle *h, int left)
{
    return h->nranks > 1 && h->refresh_mode == 1 && h->fuse_ready && left > 0 && h->ghost_valid < std::min(left, 2);
}
// ... and the exchange itself: all `halo` ghost columns of lattice `cur` from the neighbours, the compute stream waits for them.  Not overlapped
// with anything; in exchange no step of the cycle is a single k_step (which runs at the un-fused rate).
static int exchange_at_boundary(wt_handle *h)
{
    WT_TRY(halo_begin(h));
    if (h->xt_on) { HIP_TRY(hipEventRecord(xt_event(h, 2), h->s_compute)); HIP_TRY(hipEventRecord(xt_event(h, 3), h->s_compute)); }      // (no interior kernel beside it)
    HIP_TRY(hipStreamWaitEvent(h->s_compute, h->ev_halo, 0));
    if (h->xt_on) { HIP_TRY(hipEventRecord(xt_event(h, 4), h->s_compute)); h->xt_n += 1; }
    h->ghost_valid = h->halo;
    h->boundary_exchanges += 1;
    return WT_OK;
}

// One step of one handle.  `refreshed`: halo_begin was enqueued for this step; the refresh is
// overlapped with the interior columns:  [ghost refresh on s_comm] || [interior on s_compute]
// -> edge strips once the ghosts have landed.
static int step_compute(wt_handle *h, double tau, double u0, bool emit, bool refreshed)
{
    const Geom &g = h->g;
    if (h->nranks == 1) {
        WT_TRY(launch_step_any(h, 0, g.nxl, tau, u0, emit, h->s_compute));
    } else if (!refreshed) {
        if (h->ghost_valid <= 0) return fail(WT_ERR_STATE, "internal: stale ghost columns");
        WT_TRY(launch_step_any(h, 0, g.nxl, tau, u0, emit, h->s_compute));
        h->ghost_valid -= 1;
    } else {
        const int ib = h->gl ? h->gl + 1 : 0;                     // first column whose stencil avoids the left ghosts
        const int ie = h->gr ? h->gl + h->width - 1 : g.nxl;      // one past the last such column
        if (h->xt_on) HIP_TRY(hipEventRecord(xt_event(h, 2), h->s_compute));
        WT_TRY(launch_step_any(h, ib, ie, tau, u0, emit, h->s_compute));
        if (h->xt_on) HIP_TRY(hipEventRecord(xt_event(h, 3), h->s_compute));
        HIP_TRY(hipStreamWaitEvent(h->s_compute, h->ev_halo, 0));
        if (h->xt_on) { HIP_TRY(hipEventRecord(xt_event(h, 4), h->s_compute)); h->xt_n += 1; }
        if (h->gl) WT_TRY(launch_step_any(h, 0, ib, tau, u0, emit, h->s_compute));
        if (h->gr) WT_TRY(launch_step_any(h, ie, g.nxl, tau, u0, emit, h->s_compute));
        h->ghost_valid = h->halo - 1;
    }
    h->cur = 1 - h->cur;
    h->steps_done += 1;
    h->single_steps += 1;
    h->seams_valid = false;
    return WT_OK;
}

static int step_once(wt_handle *h, double tau, double u0, bool emit)
{
    if (h->nranks > 1 && h->transport == TR_NONE)
        return fail(WT_ERR_STATE, "slab handle has no transport (wt_comm_init_rank / wt_link_local)");
    const bool refresh = needs_halo(h);
    if (refresh) WT_TRY(halo_begin(h));
    return step_compute(h, tau, u0, emit, refresh);
}

// Are the three- and the two-operation division by tau exact for this tau?  Exhaustive check over all 2^23 significands on the
// device (d2q9.hpp), once per tau; the answers select the kernel instantiation (`use`: the three-operation form, which every rank
// agrees on — agree_fastdiv —; `use2`: the two-operation form of the four-step fp32 kernel, a rank's own affair: the same bits).
static int fastdiv_for(wt_handle *h, float tau, bool *use, bool *use2 = nullptr)
{
    *use = false;
    if (use2) *use2 = false;
    if (!h->fast_div) return WT_OK;
    if (!h->fd_checked || h->fd_tau != tau) {
        const FastDiv fdv = make_fastdiv((double)tau);
        HIP_TRY(hipMemsetAsync(h->d_nbad, 0, 2 * sizeof(unsigned int), h->s_compute));
        hipLaunchKernelGGL(k_verify_fastdiv, dim3(1024), dim3(256), 0, h->s_compute, fdv, h->d_nbad);
        HIP_TRY(hipGetLastError());
        unsigned int nbad[2] = {1, 1};
        HIP_TRY(hipMemcpyAsync(nbad, h->d_nbad, sizeof(nbad), hipMemcpyDeviceToHost, h->s_compute));
        HIP_TRY(hipStreamSynchronize(h->s_compute));
        const bool range_ok = std::isfinite(fdv.rtau) && tau >= 0x1p-20f && tau <= 0x1p20f;
        h->fd_tau = tau;
        h->fd_ok = (nbad[0] == 0) && range_ok;
        h->fd2_ok = h->fd_ok && (nbad[1] == 0);
        h->fd_checked = true;
    }
    *use = h->fd_ok;
    if (use2) *use2 = h->fd_ok && h->fd2_ok && h->two_op;
    return WT_OK;
}
static inline FastDiv fastdiv_params(const wt_handle *h, double tau)
{
    FastDiv f = make_fastdiv(tau);
    // fp64: the four-operation sequence is a theorem for every tau (d2q9.hpp); kept away from tau so close to the range limits that 1/tau is not normal
    f.on64 = (h->fast_div && std::isfinite(f.rhi64) && tau >= 0x1p-20 && tau <= 0x1p20) ? 1 : 0;
    return f;
}

template <typename T, int S, bool EMIT, int FD>
static void launch_march(const MarchParams<T> &p, hipStream_t st)
{
    if (p.nunits <= 0) return;
    hipLaunchKernelGGL((k_march<T, S, EMIT, FD>), dim3((unsigned)((p.nunits + 3) / 4)), dim3(256), 0, st, p);
}

// Two steps in one pass over the lattice (step_march.hpp).  A = f[cur] (time t), B = f[1-cur] (receives time t+2).
template <typename T, int S, int FD>
static int step_pair_fused_t(wt_handle *h, double tau, double u0, bool emit)
{
    const Geom &g = h->g;
    MarchParams<T> p;
    p.fs = fptr<T>(h, h->cur);
    p.fd = fptr<T>(h, 1 - h->cur);
    p.macro = reinterpret_cast<T *>(h->macro);
    p.mask = h->mask; p.bcode = h->bcode; p.wcls = h->wcls;
    p.halo = reinterpret_cast<const T *>(h->halo_tab); p.seams = reinterpret_cast<T *>(h->seams);
    p.g = g;
    p.lat_bytes = (unsigned)((size_t)9 * g.plane * sizeof(T));
    p.nwin_total = h->n_win;
    p.fdv = fastdiv_params(h, tau);
    p.tau = (T)tau;
    p.U0 = (T)u0;
    p.rev = (int)((h->steps_done >> 1) & 1);
    {
        static const int rev_mode = exp_env("WT_MARCH_REV") ? atoi(exp_env("WT_MARCH_REV")) : 2;     // experiments: 0 / 1 = fixed order
        if (rev_mode == 0 || rev_mode == 1) p.rev = rev_mode;
    }
    hipStream_t st = h->s_compute;
    if (h->n_win > 1) {        // the step-1 populations that cross the window seams
        const long nth = (long)(h->n_win - 1) * g.nxl;
        const dim3 grid((unsigned)((nth + 255) / 256)), block(256);
        const uint8_t *mk = h->mask, *sp = h->seam_plain;
        T *ht = reinterpret_cast<T *>(h->halo_tab);
        if (h->seams_valid)     // the previous pass left the seam rows of this lattice in `seams`: coalesced loads
            hipLaunchKernelGGL((k_halo_from_seams<T, FD>), dim3((unsigned)((2 * nth + 255) / 256)), block, 0, st, p.fs, (const T *)p.seams, mk, sp, ht, g, h->n_win, 64 * S, p.fdv, p.tau, p.U0);
        else                    // gather from the lattice (first pass after a single step, an upload, a ghost refresh)
            hipLaunchKernelGGL((k_halo_rows<T, FD>), grid, block, 0, st, p.fs, mk, sp, ht, g, h->n_win, 64 * S, p.fdv, p.tau, p.U0);
    }
    p.units = h->d_units; p.nunits = h->n_units;
    if (emit) launch_march<T, S, true, FD>(p, st);
    else launch_march<T, S, false, FD>(p, st);
    HIP_TRY(hipGetLastError());
    h->cur = 1 - h->cur;
    h->steps_done += 2;
    h->passes += 1;
    h->passes_total += 1;
    h->seams_valid = true;                       // the pass wrote the seam rows of the lattice it produced
    if (h->nranks > 1) h->ghost_valid -= 2;      // two columns of ghost validity consumed
    return WT_OK;
}

static int step_pair_fused(wt_handle *h, double tau, double u0, bool emit)
{
    if (h->dtype != WT_F32 || h->march_s != 2) return fail(WT_ERR_STATE, "internal: the two-step kernel runs fp32 handles with 2 sites per lane only");
    if (h->fast_math) return step_pair_fused_t<float, 2, MARCH_FD_CONTRACTED>(h, tau, u0, emit);
    bool fd = false;
    WT_TRY(fastdiv_for(h, (float)tau, &fd));
    return fd ? step_pair_fused_t<float, 2, 1>(h, tau, u0, emit) : step_pair_fused_t<float, 2, 0>(h, tau, u0, emit);
}

static inline bool tune_due(const wt_handle *h, int nsteps)
{
    return !h->plan_tuned && nsteps >= 2 && h->fuse_ready && (!h->tune_deferred || h->passes_total - h->passes_at_mask >= TUNE_LIVE_PASSES);
}

static int ensure_clocks(wt_handle *h)
{
    const size_t need = (size_t)4 * (size_t)h->n_units + 8;       // two records
    if (need > h->clk_cap) {
        if (h->d_clk) { HIP_TRY(hipFree(h->d_clk)); h->d_clk = nullptr; h->clk_cap = 0; }
        HIP_TRY(hipMalloc((void **)&h->d_clk, need * sizeof(unsigned long long)));
        h->clk_cap = need;
    }
    return WT_OK;
}

// The unit list of a pass that leaves `v_after` exact ghost columns (fewer than the kept plan's range covers): the kept plan's range shrunk to
// the owned columns + v_after ghost columns at every LOCAL slab edge (the tunnel's own ends are marched as always).  Same planners, same
// invariants, same tables; cut with the column costs the kept plan was cut with.  Every plan computes the same bits in the columns it marches.
static int trim_plan_for(wt_handle *h, int v_after, const MarchUnit **units, int *nunits)
{
    wt_handle::TrimPlan *tp = nullptr;
    for (auto &t : h->trim_plans) if (t.v_after == v_after) tp = &t;
    if (!tp) { h->trim_plans.emplace_back(); tp = &h->trim_plans.back(); tp->v_after = v_after; }
    if (!tp->valid) {
        const Geom &g = h->g;
        MarchRange r = march_range3(g, h->march_depth);
        if (h->gl) r.i_begin = std::max(r.i_begin, h->gl - v_after);
        if (h->gr) r.i_end = std::min(r.i_end, h->gl + h->width + v_after);
        MarchPlan pl;
        cut_units(h, h->colw_kept.empty() ? nullptr : h->colw_kept.data(), &pl, &r);
        if (!h->host_wcls.empty()) {
            const int max_solo = (h->march_depth == 4 ? MARCH3_MAX_CHUNK - 3 : MARCH3_MAX_CHUNK) - 2;
            h->chain_downgrades += sanitize_chain_plan(pl, h->host_wcls.data(), g, h->march_depth, max_solo);
        }
        if (xcd_order_on(h)) xcd_order(pl.units);
        const size_t total = pl.units.size();
        if (total > tp->cap) {
            if (tp->d_units) { HIP_TRY(hipFree(tp->d_units)); tp->d_units = nullptr; tp->cap = 0; }
            const size_t cap = total + total / 4 + 64;
            HIP_TRY(hipMalloc((void **)&tp->d_units, cap * sizeof(MarchUnit)));
            tp->cap = cap;
        }
        if (total > 0) {
            HIP_TRY(hipMemcpyAsync(tp->d_units, pl.units.data(), total * sizeof(MarchUnit), hipMemcpyHostToDevice, h->s_compute));
            HIP_TRY(hipStreamSynchronize(h->s_compute));
        }
        tp->n_units = (int)total;
        tp->valid = true;
    }
    *units = tp->d_units;
    *nunits = tp->n_units;
    return WT_OK;
}

// ---- one pass of k_march3 (step_march3.hpp): the parameter block, the halo-line kernel over a range of column blocks, the marching kernel ----
template <typename T>
static void march3_params(wt_handle *h, double tau, double u0, MarchParams<T> &p)
{
    const Geom &g = h->g;
    p.fs = fptr<T>(h, h->cur);
    p.fd = fptr<T>(h, 1 - h->cur);
    p.macro = reinterpret_cast<T *>(h->macro);
    p.mask = h->mask; p.bcode = h->bcode; p.wcls = h->wcls;
    p.halo = nullptr; p.hlines = reinterpret_cast<const T *>(h->hlines);
    p.seams = reinterpret_cast<T *>(h->seams);
    p.g = g;
    p.lat_bytes = (unsigned)((size_t)9 * g.plane * sizeof(T));
    p.nwin_total = h->n_win;
    p.win_stride = h->ovl ? 64 * h->march_s - 8 : 0;
    p.win_off = h->ovl ? -4 : 0;
    p.fdv = fastdiv_params(h, tau);
    p.tau = (T)tau;
    p.U0 = (T)u0;
    p.rev = (int)(h->passes & 1);
    {
        static const int rev_mode = exp_env("WT_MARCH_REV") ? atoi(exp_env("WT_MARCH_REV")) : 2;     // experiments: 0 / 1 = fixed order
        if (rev_mode == 0 || rev_mode == 1) p.rev = rev_mode;
    }
    p.stuck = h->stuck_dev;
    p.units = h->d_units; p.nunits = h->n_units;
}

// level-0 .. level-(D-1) values of the rows around the window seams for the column blocks xb0 .. xb0 + nbx - 1 (HL_COLS columns each; nbx < 0: all)
template <typename T, int S, int FD>
static void launch_halo_lines(wt_handle *h, const MarchParams<T> &p, int use_seams, int xb0, int nbx, hipStream_t st)
{
    const Geom &g = h->g;
    if (h->n_win <= 1 || h->ovl) return;      // (overlapping windows read no halo lines)
    const int all = (g.nxl + HL_COLS - 1) / HL_COLS;
    if (nbx < 0) { xb0 = 0; nbx = all; }
    if (xb0 < 0) { nbx += xb0; xb0 = 0; }
    if (xb0 + nbx > all) nbx = all - xb0;
    if (nbx <= 0) return;
    const unsigned nblk = (unsigned)(h->n_win - 1) * (unsigned)nbx;
    if (h->march_depth == 4)        // the plan's tables are those of the four-step pass, whatever this pass advances
        hipLaunchKernelGGL((k_halo4<T, S, FD>), dim3(nblk), dim3(256), 0, st, p.fs, (const T *)p.seams, (const uint8_t *)h->mask, (const uint8_t *)h->bcode,
                           (const uint8_t *)h->seam_plain, reinterpret_cast<T *>(h->hlines), g, h->n_win, use_seams, p.fdv, p.tau, p.U0, xb0, nbx);
    else
        hipLaunchKernelGGL((k_halo3<T, S, FD>), dim3(nblk), dim3(256), 0, st, p.fs, (const T *)p.seams, (const uint8_t *)h->mask, (const uint8_t *)h->bcode,
                           (const uint8_t *)h->seam_plain, reinterpret_cast<T *>(h->hlines), g, h->n_win, use_seams, p.fdv, p.tau, p.U0, xb0, nbx);
}

// the marching kernel over p.units (depth = steps this pass advances, on the tables of h->march_depth); `ovl`: the plan's windows overlap
// (p.win_stride: the instantiations without halo lines and seam rows, fp32 only — build_fuse_plan plans no other)
template <typename T, int S, int FD, int DEPTH>
static void launch_march3_k(const MarchParams<T> &p, bool emit, hipStream_t st)
{
    const dim3 grid((unsigned)((p.nunits + 3) / 4));
    if (emit) hipLaunchKernelGGL((k_march3<T, S, DEPTH, true, FD>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((k_march3<T, S, DEPTH, false, FD>), grid, dim3(256), 0, st, p);
}
template <typename T, int S, int FD>
static int launch_march3(const MarchParams<T> &p, int depth, bool emit, bool two_op, hipStream_t st)
{
    if (p.nunits <= 0) return WT_OK;
    const bool ovl = p.win_stride > 0;
    constexpr bool CAN_OVL = sizeof(T) == 4 && (FD & MARCH_FD_CONTRACTED) == 0;
    if (ovl && !CAN_OVL) return fail(WT_ERR_STATE, "internal: overlapping windows planned for a kernel that has none");
    if (depth == 4) {
        // (fp32 with the IEEE division by tau: not built — 40-80 bytes of scratch per lane; set_tau_cap keeps such a call at three steps per pass)
        if constexpr (sizeof(T) == 4 && FD == 0) return fail(WT_ERR_STATE, "internal: four-step pass with the IEEE division");
        else if constexpr (sizeof(T) == 4 && FD == 1) {
            // the division by tau in two operations where the device has proved it for this tau (fastdiv_for), in three otherwise
            if (ovl) { if (two_op) launch_march3_k<T, S, FD | MARCH_FD_TWOOP | MARCH_FD_OVL, 4>(p, emit, st); else launch_march3_k<T, S, FD | MARCH_FD_OVL, 4>(p, emit, st); }
            else if (two_op) launch_march3_k<T, S, FD | MARCH_FD_TWOOP, 4>(p, emit, st);
            else launch_march3_k<T, S, FD, 4>(p, emit, st);
        }
        else launch_march3_k<T, S, FD, 4>(p, emit, st);
    } else if (depth == 3) {
        if constexpr (CAN_OVL) { if (ovl) { launch_march3_k<T, S, FD | MARCH_FD_OVL, 3>(p, emit, st); return WT_OK; } }
        launch_march3_k<T, S, FD, 3>(p, emit, st);
    } else {
        if constexpr (CAN_OVL) { if (ovl) { launch_march3_k<T, S, FD | MARCH_FD_OVL, 2>(p, emit, st); return WT_OK; } }
        launch_march3_k<T, S, FD, 2>(p, emit, st);
    }
    return WT_OK;
}

// Three steps in one pass (step_march3.hpp), or two on the same tables (depth = 2: what a step count leaves over).
// A = f[cur] (time t), B = f[1-cur] (receives time t + depth).
template <typename T, int S, int FD>
static int step_triple_fused_t(wt_handle *h, double tau, double u0, bool emit, int depth, bool two_op = false)
{
    MarchParams<T> p;
    march3_params<T>(h, tau, u0, p);
    if (h->clk_on) p.clk = h->d_clk + h->clk_off;
#ifdef WT_UNIT_CLOCKS         // diagnostic build (tools/unit_clocks.py): every pass records its units
    else { WT_TRY(ensure_clocks(h)); p.clk = h->d_clk; }
#endif
    hipStream_t st = h->s_compute;
    launch_halo_lines<T, S, FD>(h, p, h->seams_valid ? 1 : 0, 0, -1, st);
    if (h->nranks > 1 && h->trim && !h->clk_on) {
        // ghost columns that will still be exact after this pass: the others are not marched (trim_plan_for)
        const int v_full = h->halo - (h->march_depth - 1), v_after = std::max(0, std::min(h->ghost_valid - depth, v_full));
        if (v_after < v_full) {
            const MarchUnit *tu = nullptr;
            int tn = 0;
            WT_TRY(trim_plan_for(h, v_after, &tu, &tn));
            if (tn > 0) { p.units = tu; p.nunits = tn; h->trimmed_passes += 1; }
        }
    }
    if (h->clk_on) HIP_TRY(hipEventRecord(h->ev_t0, st));       // tuning passes: the marching kernel alone is timed
    WT_TRY((launch_march3<T, S, FD>(p, depth, emit, two_op, st)));
    HIP_TRY(hipGetLastError());
    if (h->clk_on) HIP_TRY(hipEventRecord(h->ev_t1, st));
    h->cur = 1 - h->cur;
    h->steps_done += depth;
    h->passes += 1;
    h->passes_total += 1;
    h->seams_valid = true;
    // One column of ghost validity is consumed per step — and the units of a depth-D plan leave the D-1 columns next to a local edge
    // unwritten whatever the pass advances (march_range3), so a SHORTER pass on those tables still costs D-1 columns of the fresh ghosts
    // (found by the mixed-depth group test: two-step passes on four-step tables right after an initialisation).
    if (h->nranks > 1) h->ghost_valid = std::max(0, std::min(h->ghost_valid - depth, h->halo - (h->march_depth - 1)));
    return WT_OK;
}

// ---- refresh = 2: the ghost columns are renewed INSIDE a fused pass ----
// The exchange (comm stream) runs beside the marching of the INTERIOR columns — those whose `depth`-step cone stays inside the owned columns,
// [gl + depth, gl + width - depth) —, the two EDGE STRIPS — [gl - (halo - depth), gl + depth) and its mirror image: the owned columns next to
// the edges and the ghost columns that are still exact after the pass — are marched once the ghosts have landed.  No step of the cycle is a
// single k_step, no exchange stands alone at a pass boundary, and the seam buffer is never stale for the owned columns: only the strips'
// halo lines are built by the gather path (their ghost columns are new to this rank).  Two unit lists per pass length, cut with the kept
// plan's column costs like the trimmed lists.
static int renew_plan_for(wt_handle *h, int depth, wt_handle::RenewPlan **out)
{
    wt_handle::RenewPlan *rp = nullptr;
    for (auto &t : h->renew_plans) if (t.depth == depth) rp = &t;
    if (!rp) { h->renew_plans.emplace_back(); rp = &h->renew_plans.back(); rp->depth = depth; }
    if (!rp->valid) {
        const Geom &g = h->g;
        const MarchRange full = march_range3(g, h->march_depth);
        const int v_after = std::max(0, std::min(h->halo - depth, h->halo - (h->march_depth - 1)));
        MarchRange ri = full;                                  // interior
        if (h->gl) ri.i_begin = h->gl + depth;
        if (h->gr) ri.i_end = h->gl + h->width - depth;
        if (h->gr) ri.outlet_after = 0;
        if (ri.i_end - ri.i_begin < 2 * h->march_depth + 2) return fail(WT_ERR_STATE, "slab too narrow for a fused renewal (refresh = 2): %d interior columns", ri.i_end - ri.i_begin);
        const int max_solo = (h->march_depth == 4 ? MARCH3_MAX_CHUNK - 3 : MARCH3_MAX_CHUNK) - 2;
        const float *colw = h->colw_kept.empty() ? nullptr : h->colw_kept.data();
        std::vector<MarchUnit> lists[2];
        {
            MarchPlan pl;
            cut_units(h, colw, &pl, &ri);
            if (!h->host_wcls.empty()) h->chain_downgrades += sanitize_chain_plan(pl, h->host_wcls.data(), g, h->march_depth, max_solo);
            if (xcd_order_on(h)) xcd_order(pl.units);
            lists[0] = pl.units;
        }
        // the strips: a handful of columns per window.  They run AFTER the exchange, with nothing beside them: what counts is how long their slowest
        // unit takes, not how many columns are recomputed — so they are cut by time into as many units as the device holds (a plain strip column
        // becomes a unit of its own: 1 + the pipeline's fill instead of halo + fill iterations; measured on the 8-way split of 4096^2,
        // profiles/r05_d_slab_costs_cfg2.txt)
        const long target_save = h->plan_target;
        for (int side = 0; side < 2; side++) {
            if (!(side ? h->gr : h->gl)) continue;
            MarchRange rs = full;
            rs.outlet_after = 0;
            if (side == 0) { rs.i_begin = std::max(full.i_begin, h->gl - v_after); rs.i_end = h->gl + depth; }
            else { rs.i_begin = h->gl + h->width - depth; rs.i_end = std::min(full.i_end, h->gl + h->width + v_after); }
            if (rs.i_end <= rs.i_begin) continue;
            h->plan_target = std::max<long>(h->n_win, target_save / ((h->gl ? 1 : 0) + (h->gr ? 1 : 0)));
            MarchPlan pl;
            cut_units(h, colw, &pl, &rs);
            h->plan_target = target_save;
            if (!h->host_wcls.empty()) h->chain_downgrades += sanitize_chain_plan(pl, h->host_wcls.data(), g, h->march_depth, max_solo);
            lists[1].insert(lists[1].end(), pl.units.begin(), pl.units.end());
        }
        h->plan_target = target_save;
        while (lists[1].size() % 4) lists[1].push_back(MarchUnit{0, 0, 0, 0});
        for (int i = 0; i < 2; i++) {
            const size_t total = lists[i].size();
            if (total > rp->cap[i]) {
                if (rp->d_units[i]) { HIP_TRY(hipFree(rp->d_units[i])); rp->d_units[i] = nullptr; rp->cap[i] = 0; }
                const size_t cap = total + total / 4 + 64;
                HIP_TRY(hipMalloc((void **)&rp->d_units[i], cap * sizeof(MarchUnit)));
                rp->cap[i] = cap;
            }
            if (total > 0) HIP_TRY(hipMemcpyAsync(rp->d_units[i], lists[i].data(), total * sizeof(MarchUnit), hipMemcpyHostToDevice, h->s_compute));
            rp->n_units[i] = (int)total;
        }
        HIP_TRY(hipStreamSynchronize(h->s_compute));
        rp->strip_lo[0] = std::max(full.i_begin, h->gl - v_after); rp->strip_hi[0] = h->gl + depth;
        rp->strip_lo[1] = h->gl + h->width - depth; rp->strip_hi[1] = std::min(full.i_end, h->gl + h->width + v_after);
        rp->valid = true;
    }
    *out = rp;
    return WT_OK;
}

// The interior half: to be enqueued right after halo_begin(h) (the exchange is then on its way on the comm stream).
template <typename T, int S, int FD>
static int renew_interior_t(wt_handle *h, double tau, double u0, bool emit, int depth, bool two_op, bool seams_were_valid)
{
    wt_handle::RenewPlan *rp = nullptr;
    WT_TRY(renew_plan_for(h, depth, &rp));
    MarchParams<T> p;
    march3_params<T>(h, tau, u0, p);
    hipStream_t st = h->s_compute;
    if (h->xt_on) HIP_TRY(hipEventRecord(xt_event(h, 2), st));
    launch_halo_lines<T, S, FD>(h, p, seams_were_valid ? 1 : 0, 0, -1, st);      // (the lines of the ghost columns come out of stale records: rebuilt below, unused here)
    p.units = rp->d_units[0]; p.nunits = rp->n_units[0];
    WT_TRY((launch_march3<T, S, FD>(p, depth, emit, two_op, st)));
    HIP_TRY(hipGetLastError());
    if (h->xt_on) HIP_TRY(hipEventRecord(xt_event(h, 3), st));
    return WT_OK;
}
// The edge strips: after the compute stream has waited for the exchange.
template <typename T, int S, int FD>
static int renew_strips_t(wt_handle *h, double tau, double u0, bool emit, int depth, bool two_op)
{
    wt_handle::RenewPlan *rp = nullptr;
    WT_TRY(renew_plan_for(h, depth, &rp));
    MarchParams<T> p;
    march3_params<T>(h, tau, u0, p);
    hipStream_t st = h->s_compute;
    HIP_TRY(hipStreamWaitEvent(st, h->ev_halo, 0));
    if (h->xt_on) { HIP_TRY(hipEventRecord(xt_event(h, 4), st)); h->xt_n += 1; }
    // halo lines of the strips' columns from the lattice (gather path): every line a strip unit reads lies within march_depth columns of its range
    const int D = h->march_depth;
    for (int side = 0; side < 2; side++) {
        if (!(side ? h->gr : h->gl) || rp->strip_hi[side] <= rp->strip_lo[side]) continue;
        const int lo = std::max(0, rp->strip_lo[side] - D), hi = std::min(h->g.nxl, rp->strip_hi[side] + D);
        launch_halo_lines<T, S, FD>(h, p, 0, lo / HL_COLS, (hi - 1) / HL_COLS - lo / HL_COLS + 1, st);
    }
    p.units = rp->d_units[1]; p.nunits = rp->n_units[1];
    p.rev = 0;
    WT_TRY((launch_march3<T, S, FD>(p, depth, emit, two_op, st)));
    HIP_TRY(hipGetLastError());
    h->cur = 1 - h->cur;
    h->steps_done += depth;
    h->passes += 1;
    h->passes_total += 1;
    h->seams_valid = true;                       // interior and strips together wrote the seam rows of every column that is still exact
    h->ghost_valid = std::max(0, std::min(h->halo - depth, h->halo - (h->march_depth - 1)));
    h->fused_renewals += 1;
    return WT_OK;
}

// Steps the next fused pass advances given `left` steps to go (0: none — take a single step).  A pass needs as many exact ghost
// columns as it advances steps.  The tables of a depth-D plan also run every shorter pass down to two steps, and a remainder of
// ONE step is never left behind where two fused passes fit (5 = 3 + 2 on a four-step plan, 4 = 2 + 2 on a three-step plan): a
// single k_step clears the seam buffer, and the next pass would then build its halo lines by the gather path.
static inline int fuse_pick(int depth, int avail)
{
    if (avail < 2) return 0;
    if (depth <= 2) return 2;                                   // the two-step kernel (step_march.hpp) has no shorter pass
    if (avail >= depth) return (avail - depth == 1) ? depth - 1 : depth;
    return avail;
}
static inline int fuse_avail(const wt_handle *h, int left)
{
    return h->nranks == 1 ? left : (left < h->ghost_valid ? left : h->ghost_valid);
}
static inline int fuse_stride(const wt_handle *h, int left)
{
    if (!h->fuse_ready) return 0;
    return fuse_pick(eff_depth(h), fuse_avail(h, left));
}

static int step_fused(wt_handle *h, double tau, double u0, bool emit, int k)
{
    if (h->march_depth >= 3) {
        if (h->dtype != WT_F32) return step_triple_fused_t<double, 1, 1>(h, tau, u0, emit, k);      // (FD 1: the guarded four-operation division, switched by fdv.on64)
        if (h->fast_math) return step_triple_fused_t<float, 2, MARCH_FD_CONTRACTED>(h, tau, u0, emit, k);
        bool fd = false, fd2 = false;
        WT_TRY(fastdiv_for(h, (float)tau, &fd, &fd2));
        return fd ? step_triple_fused_t<float, 2, 1>(h, tau, u0, emit, k, fd2) : step_triple_fused_t<float, 2, 0>(h, tau, u0, emit, k);
    }
    return step_pair_fused(h, tau, u0, emit);
}

// refresh = 2: is the next thing a slab does a fused renewal — a pass of k > 0 steps with the exchange beside its interior columns?
// Due when a fused pass is wanted (at least two steps to go) and fewer than two exact ghost columns are left for it
// ... or when the pass the remaining ghost columns allow would leave exactly ONE step behind (3 steps to go on 2 exact ghost columns): a renewal may
// come early — it renews everything —, and with fresh ghosts the steps to go split into fused passes (3, or 3 + 2 for 5).
static inline int renew_stride(const wt_handle *h, int left)
{
    if (!(h->nranks > 1 && h->refresh_mode == 2 && h->fuse_ready && h->march_depth >= 3) || left < 2) return 0;
    const int k = fuse_pick(eff_depth(h), std::min(left, h->ghost_valid));
    if (h->ghost_valid >= 2 && k > 0 && left - k != 1) return 0;
    return fuse_pick(eff_depth(h), std::min(left, h->halo));
}
// the two halves of such a pass (see renew_interior_t): between them every slab's exchange is in flight
static int renew_interior(wt_handle *h, double tau, double u0, bool emit, int k, bool seams_were_valid)
{
    if (h->dtype != WT_F32) return renew_interior_t<double, 1, 1>(h, tau, u0, emit, k, false, seams_were_valid);
    if (h->fast_math) return renew_interior_t<float, 2, MARCH_FD_CONTRACTED>(h, tau, u0, emit, k, false, seams_were_valid);
    bool fd = false, fd2 = false;
    WT_TRY(fastdiv_for(h, (float)tau, &fd, &fd2));
    return fd ? renew_interior_t<float, 2, 1>(h, tau, u0, emit, k, fd2, seams_were_valid) : renew_interior_t<float, 2, 0>(h, tau, u0, emit, k, false, seams_were_valid);
}
static int renew_strips(wt_handle *h, double tau, double u0, bool emit, int k)
{
    if (h->dtype != WT_F32) return renew_strips_t<double, 1, 1>(h, tau, u0, emit, k, false);
    if (h->fast_math) return renew_strips_t<float, 2, MARCH_FD_CONTRACTED>(h, tau, u0, emit, k, false);
    bool fd = false, fd2 = false;
    WT_TRY(fastdiv_for(h, (float)tau, &fd, &fd2));
    return fd ? renew_strips_t<float, 2, 1>(h, tau, u0, emit, k, fd2) : renew_strips_t<float, 2, 0>(h, tau, u0, emit, k, false);
}

// Measure, then cut again.  The cut by time rests on a model of what a column costs (cut_units), and a launch takes as long as its
// slowest unit; how good the model is depends on the mask and on the slab (the piece of the body a slab of an 8-way split holds ran 15 %
// behind its plain units with constants fitted on the whole lattice: tools/r3_slab_costs.py).  So before the first pass on a new plan the
// library times the units themselves: two passes (one per launch order) from the present state into the lattice nobody reads, with the
// kernel stamping every unit's start and end (MarchParams::clk); a solo unit that ran r times as long as the median chain unit (or the
// median unit, where no chain fits) has the cost of its own columns multiplied by r^0.6, the columns are cut again, and after a few rounds
// the plan with the shortest measured makespan is kept (the modelled one if nothing beat it).  The populations are not touched: the
// passes write f[1 - cur] and the scratch tables, the handle's counters are put back, and only the seam buffer is marked stale.
// The cut changes no result — every plan computes the same bits (tests/test_gpu_fused.py runs tuned and untuned plans against the oracle).
static int step_fused(wt_handle *h, double tau, double u0, bool emit, int k);
static int tune_fuse_plan(wt_handle *h, double tau, double u0)
{
    h->plan_tuned = true;
    h->tune_rounds = 0;
    h->tune_gain = 1.0;
    static const int rounds = exp_env("WT_TUNE_ROUNDS") ? atoi(exp_env("WT_TUNE_ROUNDS")) : 6;
    static const int trace = exp_env("WT_TUNE_TRACE") ? atoi(exp_env("WT_TUNE_TRACE")) : 0;
    static const double damp = exp_env("WT_TUNE_DAMP") ? atof(exp_env("WT_TUNE_DAMP")) : 0.6;
    if (!h->tune || rounds <= 0 || !h->fuse_ready || h->march_depth < 3 || !plan_by_time(h) || h->n_units < 8) return WT_OK;
    const int k = fuse_pick(eff_depth(h), 1 << 20);
    if (k < 2) return WT_OK;
    const Geom &g = h->g;
    const int ld = g.nxl + 2;
    std::vector<float> colw((size_t)h->n_win * ld, 1.0f);
    std::vector<float> colw_cur, colw_best;          // the corrections the present / the best plan was cut with (empty: the modelled costs)
    std::vector<MarchUnit> best = h->host_units;
    double best_span = 0.0, first_span = 0.0;
    std::vector<unsigned long long> clk;
    std::vector<double> dur, chain_dur;
    const int s_cur = h->cur, s_gv = h->ghost_valid;
    const long long s_steps = h->steps_done, s_passes = h->passes, s_total = h->passes_total;
    bool have_best = false, uploaded_best = true;
    for (int it = 0; it < rounds; it++) {
        const int n = h->n_units;
        WT_TRY(ensure_clocks(h));
        HIP_TRY(hipMemsetAsync(h->d_clk, 0, (size_t)4 * n * sizeof(unsigned long long), h->s_compute));
        // the plan's score: the time of its marching kernel, one launch per launch order (the units' clocks are per XCD and not
        // comparable across units — only their durations are used); the very first launch is a warm-up
        double span = 0.0;
        for (int rep = (it == 0 ? -1 : 0); rep < 2; rep++) {
            h->clk_on = true; h->clk_off = (size_t)2 * n * (rep < 0 ? 0 : rep);
            h->passes = rep & 1;                               // the launch order alternates with the pass count
            const int rc = step_fused(h, tau, u0, false, k);
            h->clk_on = false;
            h->cur = s_cur; h->ghost_valid = s_gv; h->steps_done = s_steps; h->passes = s_passes; h->passes_total = s_total;
            h->seams_valid = false;                            // the seam rows now describe the lattice that was thrown away
            if (rc != WT_OK) return rc;
            HIP_TRY(hipEventSynchronize(h->ev_t1));
            float ms = 0.0f;
            HIP_TRY(hipEventElapsedTime(&ms, h->ev_t0, h->ev_t1));
            if (rep >= 0) span += 0.5 * (double)ms;
        }
        clk.resize((size_t)4 * n);
        HIP_TRY(hipMemcpyAsync(clk.data(), h->d_clk, clk.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->s_compute));
        HIP_TRY(hipStreamSynchronize(h->s_compute));
        dur.assign(n, 0.0);
        chain_dur.clear();
        double dmax = 0.0;
        for (int u = 0; u < n; u++) {
            for (int rep = 0; rep < 2; rep++) {
                const unsigned long long t0 = clk[(size_t)2 * n * rep + 2 * u], t1 = clk[(size_t)2 * n * rep + 2 * u + 1];
                if (t1 != 0) dur[u] += 0.5 * (double)(t1 - t0);          // (zero: an empty unit, padding of the last solo group)
            }
            dmax = std::max(dmax, dur[u]);
        }
        if (trace) {
            std::vector<double> cd, sd;
            int umax = 0;
            for (int u = 0; u < n; u++) {
                if (dur[u] <= 0.0) continue;
                ((h->host_units[u].flags & MU_CHAIN) ? cd : sd).push_back(dur[u]);
                if (dur[u] > dur[umax]) umax = u;
            }
            std::sort(cd.begin(), cd.end()); std::sort(sd.begin(), sd.end());
            for (int pos = 0; pos < 4; pos++) {                   // chain units by their place in the block (outer, inner, inner, outer)
                std::vector<double> pd;
                for (int u = pos; u < n; u += 4) if (dur[u] > 0.0 && (h->host_units[u].flags & MU_CHAIN)) pd.push_back(dur[u]);
                std::sort(pd.begin(), pd.end());
                if (!pd.empty()) fprintf(stderr, "[wt tune]   chain place %d: %zu units of %d columns, median %.0f p90 %.0f max %.0f\n", pos, pd.size(),
                                         h->host_units[pos].ib - h->host_units[pos].ia, pd[pd.size() / 2], pd[pd.size() * 9 / 10], pd.back());
            }
            const MarchUnit &um = h->host_units[umax];
            fprintf(stderr, "[wt tune] round %d: %d units, kernel %.4f ms; chain %zu: median %.0f max %.0f; solo %zu: median %.0f p90 %.0f max %.0f; slowest: window %d columns [%d, %d) flags %d\n",
                    it, n, span, cd.size(), cd.empty() ? 0.0 : cd[cd.size() / 2], cd.empty() ? 0.0 : cd.back(), sd.size(), sd.empty() ? 0.0 : sd[sd.size() / 2],
                    sd.empty() ? 0.0 : sd[sd.size() * 9 / 10], sd.empty() ? 0.0 : sd.back(), um.w, um.ia, um.ib, um.flags);
        }
        if (span <= 0.0) break;
        if (it == 0) first_span = span;
        if (!have_best || span < best_span) { best_span = span; best = h->host_units; colw_best = colw_cur; have_best = true; uploaded_best = true; }
        else uploaded_best = false;
        h->tune_rounds = it + 1;
        if (it + 1 == rounds) break;
        std::vector<double> all;
        for (int u = 0; u < n; u++) {
            if (dur[u] <= 0.0) continue;
            all.push_back(dur[u]);
            if (h->host_units[u].flags & MU_CHAIN) chain_dur.push_back(dur[u]);
        }
        std::vector<double> &refv = chain_dur.size() >= 16 ? chain_dur : all;
        if (refv.empty()) break;
        std::nth_element(refv.begin(), refv.begin() + refv.size() / 2, refv.end());
        const double ref = refv[refv.size() / 2];
        for (int u = 0; u < n; u++) {
            const MarchUnit &un = h->host_units[u];
            if (dur[u] <= 0.0 || (un.flags & MU_CHAIN)) continue;
            const float r = (float)pow(dur[u] / ref, damp);
            for (int x = un.ia; x < un.ib; x++) {
                float &c = colw[(size_t)un.w * ld + x + 1];
                c = std::min(6.0f, std::max(0.2f, c * r));
            }
        }
        MarchPlan pl;
        cut_units(h, colw.data(), &pl);
        if (pl.units.empty()) break;
        WT_TRY(upload_units(h, pl));
        colw_cur = colw;
        uploaded_best = false;
    }
    h->colw_kept = colw_best;                        // the trimmed plans of a slab (trim_plan_for) are cut with the kept plan's costs
    if (have_best && !uploaded_best) {
        MarchPlan pl;
        pl.units = best;
        pl.chunk = h->fuse_chunk_used;
        WT_TRY(upload_units(h, pl));
    }
    if (best_span > 0.0) h->tune_gain = first_span / best_span;
    return WT_OK;
}

// fp32 plans of depth 4 take three-step passes while the division by tau has to be the IEEE one (fast_div off, or a tau the proof rejects)
static int set_tau_cap(wt_handle *h, double tau)
{
    h->tau_cap = 0;
    if (h->fuse_ready && h->dtype == WT_F32 && h->march_depth == 4 && !h->fast_math) {
        bool fd = false;
        WT_TRY(fastdiv_for(h, (float)tau, &fd));
        if (!fd) h->tau_cap = 3;
    }
    return WT_OK;
}


// ------------------------------------------------------------------------------------------
// cross-rank agreement
// ------------------------------------------------------------------------------------------
// Everything that decides the SEQUENCE of fused passes, single steps and ghost refreshes a slab takes (run_steps, wt_plan_steps).  Over RCCL every
// rank decides alone and the exchange is collective: two ranks that differ in any of these do not produce wrong numbers, they hang.  Inputs
// first — so that a mismatch can be NAMED — then the outcome of the planner.  (What a rank may decide for itself is absent on purpose: the cut
// of its units, measured or modelled, its mask columns, its device ordinal.)
struct SchedField { const char *name; long long v; };
static const int SCHED_FIELDS_MAX = 32;
static long wave_slots_of(const wt_handle *h) { return h->wave_slots; }
static int schedule_fingerprint(const wt_handle *h, SchedField *f)
{
    int n = 0;
    unsigned long long eh = 1469598103934665603ULL;                  // FNV-1a over the split
    for (int e : h->edges) { eh ^= (unsigned long long)(unsigned)e; eh *= 1099511628211ULL; }
    auto put = [&](const char *name, long long v) { if (n < SCHED_FIELDS_MAX) { f[n].name = name; f[n].v = v; n++; } };
    put("nx_global", h->nx_g); put("ny", h->ny); put("nranks", h->nranks); put("halo", h->halo); put("dtype", h->dtype);
    put("edges (hash of the split)", (long long)(eh >> 1));
    put("option fuse_steps (WT_FUSE2)", h->fuse ? (h->fuse_force ? 2 : 1) : 0);
    put("option fuse_depth", h->fuse_depth);
    put("option fuse_chunk (WT_FUSE_CHUNK)", h->fuse_chunk);
    put("option chain (WT_CHAIN)", h->chain ? 1 : 0);
    put("option fast_div (WT_FAST_DIV)", h->fast_div ? 1 : 0);
    put("option fast_math", h->fast_math ? 1 : 0);
    put("option plan_columns", h->plan_columns);
    put("option refresh", h->refresh_mode);
    put("resident wave slots of the device (CUs x 4 SIMDs x waves)", wave_slots_of(h));
    put("marching kernels eligible (widest slab below 4 GiB, even NY)", fuse_eligible(h) ? 1 : 0);
    put("mask set", h->mask_set ? 1 : 0);
    put("planner outcome: fused plan ready", h->fuse_ready ? 1 : 0);
    put("planner outcome: steps per pass of the tables", h->fuse_ready ? h->march_depth : 0);
    put("planner outcome: pass cap", h->fuse_ready ? h->pass_cap : 0);
    put("state: steps done", h->steps_done);
    put("state: exact ghost columns", h->nranks > 1 ? h->ghost_valid : 0);
    return n;
}

static int agree_mismatch(const wt_handle *h, const SchedField &f, long long lo, long long hi, const char *when)
{
    return fail(WT_ERR_STATE, "slab ranks disagree on '%s' (%s): rank %d has %lld, the ranks span %lld .. %lld — every slab of a tunnel must take "
                              "the same sequence of passes and ghost refreshes (same options, same documented WT_* environment, same split and halo "
                              "on every rank)", f.name, when, h->rank, f.v, lo, hi);
}

// TR_RCCL: one all-reduce(max) over {v, -v} yields max and min of every field on every rank; all ranks see the same table, so all of them fail
// together (nobody is left waiting in an exchange).  COLLECTIVE: made inside wt_comm_init_rank and at the first stepping call after a schedule
// input changed — wt_set_option / wt_set_mask / wt_init_equilibrium / wt_write_f on a slab handle are collective in that sense (every rank
// makes the same call), as they always had to be.
// `agree_dirty` is cleared only once the comparison has SUCCEEDED: after a refusal, or a HIP / RCCL error inside the check, the next stepping call
// checks again and is refused again (cleared up front, a refused handle could be stepped unchecked — over RCCL, the hang this exists for).
static int agree_rccl(wt_handle *h, const char *when)
{
    if (!h->agree_check || h->transport != TR_RCCL) { h->agree_dirty = false; return WT_OK; }
    SchedField f[SCHED_FIELDS_MAX];
    const int n = schedule_fingerprint(h, f);
    long long host[2 * SCHED_FIELDS_MAX];
    for (int i = 0; i < n; i++) { host[i] = f[i].v; host[n + i] = -f[i].v; }
    if (!h->d_agree) HIP_TRY(hipMalloc((void **)&h->d_agree, 2 * SCHED_FIELDS_MAX * sizeof(long long)));
    HIP_TRY(hipMemcpyAsync(h->d_agree, host, 2 * n * sizeof(long long), hipMemcpyHostToDevice, h->s_comm));
    NCCL_TRY(ncclAllReduce(h->d_agree, h->d_agree, (size_t)2 * n, ncclInt64, ncclMax, h->comm, h->s_comm));
    HIP_TRY(hipMemcpyAsync(host, h->d_agree, 2 * n * sizeof(long long), hipMemcpyDeviceToHost, h->s_comm));
    HIP_TRY(hipStreamSynchronize(h->s_comm));
    h->agree_checks += 1;
    for (int i = 0; i < n; i++)
        if (host[i] != -host[n + i]) return agree_mismatch(h, f[i], -host[n + i], host[i], when);
    h->agree_dirty = false;
    return WT_OK;
}

// TR_LOCAL: the same table compared on the host over the handles of the group.  (The flags stay set on a refusal: the next call checks again.)
static int agree_local(wt_handle **hs, int n_h, const char *when)
{
    bool due = false, on = true;
    for (int r = 0; r < n_h; r++) { due = due || hs[r]->agree_dirty; on = on && hs[r]->agree_check; }
    if (!due || !on || n_h < 2) {
        for (int r = 0; r < n_h; r++) hs[r]->agree_dirty = false;
        return WT_OK;
    }
    SchedField f0[SCHED_FIELDS_MAX], fr[SCHED_FIELDS_MAX];
    const int n = schedule_fingerprint(hs[0], f0);
    for (int r = 1; r < n_h; r++) {
        schedule_fingerprint(hs[r], fr);
        for (int i = 0; i < n; i++)
            if (fr[i].v != f0[i].v) return agree_mismatch(hs[r], fr[i], std::min(fr[i].v, f0[i].v), std::max(fr[i].v, f0[i].v), when);
    }
    for (int r = 0; r < n_h; r++) { hs[r]->agree_checks += 1; hs[r]->agree_dirty = false; }
    return WT_OK;
}

// The fast-division verdict decides the pass length of an fp32 four-step plan (set_tau_cap): all ranks take the kernel the LEAST lucky rank can
// take — an all-reduce(min), the first time a tau meets a slab handle.  (The proof is deterministic arithmetic; a verdict that differs between
// devices would be a broken device — and would otherwise show up as a hang.)
static int agree_tau_cap_rccl(wt_handle *h, float tau)
{
    if (!h->agree_check || h->transport != TR_RCCL) return WT_OK;
    if (!(h->fuse_ready && h->dtype == WT_F32 && h->march_depth == 4 && !h->fast_math)) return WT_OK;
    if (h->fd_agreed && h->fd_agreed_tau == tau) { if (!h->fd_ok) h->tau_cap = 3; return WT_OK; }
    long long mine = h->tau_cap == 0 ? 1 : 0, all = 0;
    if (!h->d_agree) HIP_TRY(hipMalloc((void **)&h->d_agree, 2 * SCHED_FIELDS_MAX * sizeof(long long)));
    HIP_TRY(hipMemcpyAsync(h->d_agree, &mine, sizeof(mine), hipMemcpyHostToDevice, h->s_comm));
    NCCL_TRY(ncclAllReduce(h->d_agree, h->d_agree, 1, ncclInt64, ncclMin, h->comm, h->s_comm));
    HIP_TRY(hipMemcpyAsync(&all, h->d_agree, sizeof(all), hipMemcpyDeviceToHost, h->s_comm));
    HIP_TRY(hipStreamSynchronize(h->s_comm));
    h->fd_agreed = true; h->fd_agreed_tau = tau;
    if (all == 0) { h->fd_ok = false; h->tau_cap = 3; }          // somebody divides in IEEE arithmetic: so does everybody
    return WT_OK;
}

// The trimmed unit lists used to be cut the first time a pass needed them — 0.3 ms of planning on the host, a device allocation and a stream
// synchronisation in the middle of the stepping loop, fifteen times per cycle at halo 61 and again for every residue a call's step count leaves behind
// (a locally linked 8-slab group: 194 us per step over the first 610 steps, 119 once every list existed).  They are cut HERE now, all halo - 3 of them,
// before the first launch of the first stepping call on a plan (0.3 ms each; trim_plan_for stays lazy for whoever calls it first).
static int prebuild_trim_plans(wt_handle *h)
{
    if (h->trim_prebuilt || !(h->nranks > 1 && h->trim && h->fuse_ready && h->march_depth >= 3)) return WT_OK;
    const int v_full = h->halo - (h->march_depth - 1);
    for (int v_after = 0; v_after < v_full; v_after++) {
        const MarchUnit *tu = nullptr;
        int tn = 0;
        WT_TRY(trim_plan_for(h, v_after, &tu, &tn));
    }
    if (h->refresh_mode == 2) {
        wt_handle::RenewPlan *rp = nullptr;
        const int k = fuse_pick(eff_depth(h), h->halo);
        if (k > 0) WT_TRY(renew_plan_for(h, k, &rp));
    }
    h->trim_prebuilt = true;
    return WT_OK;
}

// What a stepping call does before its first launch: the pass cap of this tau, the cross-rank checks that are due, the measured cut of a new plan.
// (wt_step_timed runs it BEFORE its first event: the 13 trial passes of a new plan are not step time — ADVICE r3.)
static int prepare_steps(wt_handle *h, int nsteps, double tau, double u0)
{
    WT_TRY(set_tau_cap(h, tau));
    if (h->transport == TR_RCCL) {
        if (h->agree_dirty) WT_TRY(agree_rccl(h, "first stepping call after a change"));
        WT_TRY(agree_tau_cap_rccl(h, (float)tau));
    }
    if (tune_due(h, nsteps)) WT_TRY(tune_fuse_plan(h, tau, u0));
    WT_TRY(prebuild_trim_plans(h));
    return WT_OK;
}

static int run_steps(wt_handle *h, int nsteps, double tau, double u0)
{
    WT_TRY(prepare_steps(h, nsteps, tau, u0));
    int s = 0;
    while (s < nsteps) {
        if (boundary_exchange_due(h, nsteps - s)) {
            if (h->transport == TR_NONE) return fail(WT_ERR_STATE, "slab handle has no transport (wt_comm_init_rank / wt_link_local)");
            WT_TRY(exchange_at_boundary(h));
        }
        if (const int kr = renew_stride(h, nsteps - s)) {         // refresh = 2: exchange || interior columns, then the edge strips
            if (h->transport == TR_NONE) return fail(WT_ERR_STATE, "slab handle has no transport (wt_comm_init_rank / wt_link_local)");
            const bool sv = h->seams_valid;
            WT_TRY(halo_begin(h));
            WT_TRY(renew_interior(h, tau, u0, s + kr == nsteps, kr, sv));
            WT_TRY(renew_strips(h, tau, u0, s + kr == nsteps, kr));
            s += kr;
            continue;
        }
        const int k = fuse_stride(h, nsteps - s);
        if (k > 0) {
            WT_TRY(step_fused(h, tau, u0, s + k == nsteps, k));
            s += k;
        } else {
            WT_TRY(step_once(h, tau, u0, s + 1 == nsteps));
            s += 1;
        }
    }
    if (nsteps > 0) h->macro_stale = false;     // the last step emitted (rho,ux,uy)
    return WT_OK;
}

// How would `nsteps` steps be taken from the handle's present state?  The decisions of run_steps without a launch: +k = one fused pass of k
// steps, 1 = a single step, -1 = a single step that refreshes the ghost columns first.  Every slab of a tunnel must produce the SAME sequence
// (over RCCL each rank decides alone and the exchange is collective): tests/test_gpu_slabs.py compares the ranks' answers.
extern "C" int wt_plan_steps(wt_handle *h, int nsteps, double tau, int *seq, int cap)
{
    WT_TRY(check_handle(h));
    if (nsteps < 0 || !seq || cap < 0) return fail(WT_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    WT_TRY(set_tau_cap(h, tau));
    const int gv0 = h->ghost_valid;
    int n = 0, s = 0;
    while (s < nsteps) {
        if (boundary_exchange_due(h, nsteps - s)) {          // refresh = 1: -2 = an exchange at a pass boundary (no step)
            if (n < cap) seq[n] = -2;
            n++;
            h->ghost_valid = h->halo;
        }
        if (const int kr = renew_stride(h, nsteps - s)) {     // refresh = 2: 100 + k = a fused pass of k steps that renews the ghost columns
            if (n < cap) seq[n] = 100 + kr;
            n++;
            h->ghost_valid = std::max(0, std::min(h->halo - kr, h->halo - (h->march_depth - 1)));
            s += kr;
            continue;
        }
        const int k = fuse_stride(h, nsteps - s);
        int code;
        if (k > 0) {
            code = k;
            if (h->nranks > 1) h->ghost_valid = std::max(0, std::min(h->ghost_valid - k, h->march_depth >= 3 ? h->halo - (h->march_depth - 1) : h->ghost_valid - k));
            s += k;
        } else {
            const bool refresh = needs_halo(h);
            code = refresh ? -1 : 1;
            if (h->nranks > 1) h->ghost_valid = refresh ? h->halo - 1 : h->ghost_valid - 1;
            s += 1;
        }
        if (n < cap) seq[n] = code;
        n++;
    }
    h->ghost_valid = gv0;
    return n;
}

extern "C" int wt_step(wt_handle *h, int nsteps, double tau, double u0)
{
    WT_TRY(check_steppable(h, nsteps, tau, u0));
    if (h->transport == TR_LOCAL) return fail(WT_ERR_STATE, "locally linked slabs are stepped with wt_step_group");
    HIP_TRY(hipSetDevice(h->device));
    return run_steps(h, nsteps, tau, u0);
}

extern "C" int wt_step_timed(wt_handle *h, int nsteps, double tau, double u0, float *elapsed_ms)
{
    WT_TRY(check_steppable(h, nsteps, tau, u0));
    if (!elapsed_ms) return fail(WT_ERR_ARG, "elapsed_ms is null");
    if (h->transport == TR_LOCAL) return fail(WT_ERR_STATE, "locally linked slabs are stepped with wt_step_group");
    HIP_TRY(hipSetDevice(h->device));
    WT_TRY(prepare_steps(h, nsteps, tau, u0));          // (the trial passes of a new plan are not step time)
    HIP_TRY(hipEventRecord(h->ev_a, h->s_compute));
    WT_TRY(run_steps(h, nsteps, tau, u0));
    HIP_TRY(hipEventRecord(h->ev_b, h->s_compute));
    HIP_TRY(hipEventSynchronize(h->ev_b));
    HIP_TRY(hipEventElapsedTime(elapsed_ms, h->ev_a, h->ev_b));
    return check_stuck(h);                              // (the host has just synchronised with the device: look at the word the bounded poll raises)
}

// ------------------------------------------------------------------------------------------
// transports
// ------------------------------------------------------------------------------------------
extern "C" int wt_comm_unique_id(void *id_out)
{
    if (!id_out) return fail(WT_ERR_ARG, "id_out is null");
    WT_TRY(rccl_require());
    static_assert(sizeof(ncclUniqueId) <= WT_COMM_ID_BYTES, "ncclUniqueId larger than WT_COMM_ID_BYTES");
    ncclUniqueId id;
    NCCL_TRY(ncclGetUniqueId(&id));
    memset(id_out, 0, WT_COMM_ID_BYTES);
    memcpy(id_out, &id, sizeof(id));
    return WT_OK;
}

extern "C" int wt_comm_init_rank(wt_handle *h, const void *id_in)
{
    WT_TRY(check_handle(h));
    if (!id_in) return fail(WT_ERR_ARG, "id is null");
    if (h->nranks < 2) return fail(WT_ERR_STATE, "not a slab handle");
    if (h->transport != TR_NONE) return fail(WT_ERR_STATE, "handle already has a transport");
    WT_TRY(rccl_require());
    HIP_TRY(hipSetDevice(h->device));
    ncclUniqueId id;
    memcpy(&id, id_in, sizeof(id));
    NCCL_TRY(ncclCommInitRank(&h->comm, h->nranks, id, h->rank));
    h->transport = TR_RCCL;
    int rc = WT_OK;
    {
        const ncclResult_t r = ncclCommCount(h->comm, &h->comm_ranks);
        if (r != ncclSuccess) rc = fail(WT_ERR_RCCL, "ncclCommCount failed: %s", ncclGetErrorString(r));
        else if (h->comm_ranks != h->nranks) rc = fail(WT_ERR_RCCL, "the communicator holds %d ranks, the tunnel has %d slabs", h->comm_ranks, h->nranks);
    }
    // every rank proves, before the first exchange, that it will take the schedule its neighbours take
    if (rc == WT_OK) rc = agree_rccl(h, "wt_comm_init_rank");
    if (rc != WT_OK) {
        // a refused handle must not stay steppable: the communicator is torn down and the handle is back to "no transport" — every rank
        // reaches this branch together (the agreement table is the same on all of them), so nobody is left inside a collective
        (void)ncclCommDestroy(h->comm);
        h->comm = nullptr;
        h->transport = TR_NONE;
        h->comm_ranks = 0;
        h->agree_dirty = true;
    }
    return rc;
}

// RCCL plumbing check on ONE GPU: a one-rank communicator, then the same grouped
// ncclSend/ncclRecv pattern exchange_rccl uses (18 messages, to self), on a non-blocking stream.
// Exercises the library's RCCL linkage and call sequence where no second GPU is available.
extern "C" int wt_comm_selftest(int device, int ny)
{
    if (ny < 1 || ny > (1 << 20)) return fail(WT_ERR_ARG, "ny out of range");
    WT_TRY(rccl_require());
    HIP_TRY(hipSetDevice(device));
    ncclUniqueId id;
    NCCL_TRY(ncclGetUniqueId(&id));
    ncclComm_t comm = nullptr;
    NCCL_TRY(ncclCommInitRank(&comm, 1, id, 0));   // (nothing to release before this point)
    const size_t count = (size_t)ny * 16;          // 16 ghost columns
    float *src = nullptr, *dst = nullptr;
    hipStream_t st = nullptr;
    int rc = WT_OK;
    std::vector<float> host(9 * count), back(9 * count);
    for (size_t i = 0; i < host.size(); i++) host[i] = (float)(i % 9973) * 0.25f;
    do {
        if (hipMalloc((void **)&src, 9 * count * 4) != hipSuccess || hipMalloc((void **)&dst, 9 * count * 4) != hipSuccess) { rc = fail(WT_ERR_OOM, "selftest alloc"); break; }
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { rc = fail(WT_ERR_HIP, "selftest stream"); break; }
        if (hipMemcpy(src, host.data(), 9 * count * 4, hipMemcpyHostToDevice) != hipSuccess) { rc = fail(WT_ERR_HIP, "selftest upload"); break; }
        if (hipMemset(dst, 0, 9 * count * 4) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { rc = fail(WT_ERR_HIP, "selftest memset"); break; }
        ncclResult_t r = ncclGroupStart();
        for (int k = 0; k < 9 && r == ncclSuccess; k++) {
            r = ncclSend(src + k * count, count, ncclFloat32, 0, comm, st);
            if (r == ncclSuccess) r = ncclRecv(dst + k * count, count, ncclFloat32, 0, comm, st);
        }
        const ncclResult_t re = ncclGroupEnd();
        if (r == ncclSuccess) r = re;
        if (r != ncclSuccess) { rc = fail(WT_ERR_RCCL, "selftest exchange failed: %s", ncclGetErrorString(r)); break; }
        if (hipStreamSynchronize(st) != hipSuccess) { rc = fail(WT_ERR_HIP, "selftest sync"); break; }
        if (hipMemcpy(back.data(), dst, 9 * count * 4, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(WT_ERR_HIP, "selftest download"); break; }
        if (memcmp(host.data(), back.data(), 9 * count * 4) != 0) { rc = fail(WT_ERR_RCCL, "selftest: received data differ from sent data"); break; }
    } while (0);
    if (st) (void)hipStreamDestroy(st);
    if (src) (void)hipFree(src);
    if (dst) (void)hipFree(dst);
    (void)ncclCommDestroy(comm);
    return rc;
}

extern "C" int wt_link_local(wt_handle **hs, int n)
{
    if (hs && n == 1 && hs[0] && hs[0]->nranks > 1) {
        // ONE slab handle linked to ITSELF: its ghost columns are refreshed from its own owned edge columns (a tunnel periodic in x over this slab).
        // The timing stand-in for one slab of a split (distributed.measure_slab_real, tools/r5_slab_costs.py): the whole slab state machine — trimmed
        // ghost marching, the refresh mode, the exchange beside the interior — runs on one handle alone on the GPU, with a copy kernel as the exchange.
        wt_handle *h = hs[0];
        if (h->transport != TR_NONE) return fail(WT_ERR_STATE, "handle already has a transport");
        if (h->width < 2 * h->halo) return fail(WT_ERR_ARG, "a self-linked slab needs at least 2 x halo owned columns");
        h->peer_l = h->gl ? h : nullptr;
        h->peer_r = h->gr ? h : nullptr;
        h->transport = TR_LOCAL;
        return WT_OK;
    }
    if (!hs || n < 2) return fail(WT_ERR_ARG, "need at least two slab handles (or one slab handle, linked to itself)");
    for (int r = 0; r < n; r++) {
        wt_handle *h = hs[r];
        WT_TRY(check_handle(h));
        if (h->nranks != n || h->rank != r) return fail(WT_ERR_ARG, "handle %d is rank %d of %d", r, h->rank, h->nranks);
        if (h->transport != TR_NONE) return fail(WT_ERR_STATE, "handle %d already has a transport", r);
        if (h->nx_g != hs[0]->nx_g || h->ny != hs[0]->ny || h->dtype != hs[0]->dtype || h->halo != hs[0]->halo)
            return fail(WT_ERR_ARG, "slab %d does not match slab 0", r);
    }
    for (int r = 0; r < n; r++) {
        wt_handle *h = hs[r];
        h->peer_l = r > 0 ? hs[r - 1] : nullptr;
        h->peer_r = r < n - 1 ? hs[r + 1] : nullptr;
        h->transport = TR_LOCAL;
        for (wt_handle *p : {h->peer_l, h->peer_r}) {
            if (p && p->device != h->device) {
                int can = 0;
                HIP_TRY(hipDeviceCanAccessPeer(&can, h->device, p->device));
                if (can) {
                    HIP_TRY(hipSetDevice(h->device));
                    hipError_t e = hipDeviceEnablePeerAccess(p->device, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                        return fail(WT_ERR_HIP, "hipDeviceEnablePeerAccess failed: %s", hipGetErrorString(e));
                    (void)hipGetLastError();
                }
            }
        }
    }
    return WT_OK;
}

static int group_prepare(wt_handle **hs, int n, int nsteps, double tau, double u0)
{
    for (int r = 0; r < n; r++) {
        HIP_TRY(hipSetDevice(hs[r]->device));
        WT_TRY(set_tau_cap(hs[r], tau));
    }
    WT_TRY(agree_local(hs, n, "wt_step_group"));
    for (int r = 0; r < n; r++) {
        HIP_TRY(hipSetDevice(hs[r]->device));
        if (tune_due(hs[r], nsteps)) WT_TRY(tune_fuse_plan(hs[r], tau, u0));
        WT_TRY(prebuild_trim_plans(hs[r]));
    }
    return WT_OK;
}

extern "C" int wt_step_group(wt_handle **hs, int n, int nsteps, double tau, double u0)
{
    if (!hs || n < 1) return fail(WT_ERR_ARG, "no handles");
    for (int r = 0; r < n; r++) {
        WT_TRY(check_steppable(hs[r], nsteps, tau, u0));
        if (n > 1 && hs[r]->transport != TR_LOCAL) return fail(WT_ERR_STATE, "handle %d is not locally linked", r);
        if (hs[r]->steps_done != hs[0]->steps_done) return fail(WT_ERR_STATE, "slabs are not at the same step");
    }
    WT_TRY(group_prepare(hs, n, nsteps, tau, u0));
    const bool multi = n > 1 || (hs[0]->nranks > 1 && hs[0]->transport == TR_LOCAL);      // (one handle: a whole tunnel, or a slab linked to itself)
    int s = 0;
    while (s < nsteps) {
        // refresh = 1 (agreed by the group): an exchange at a pass boundary as soon as any slab is short of exact ghost columns
        bool xdue = false;
        for (int r = 0; r < n && multi; r++) xdue = xdue || boundary_exchange_due(hs[r], nsteps - s);
        if (xdue) {
            for (int r = 0; r < n; r++) {                      // every slab's comm stream must see its neighbours' finished lattices
                HIP_TRY(hipSetDevice(hs[r]->device));
                HIP_TRY(hipEventRecord(hs[r]->ev_state, hs[r]->s_compute));
            }
            for (int r = 0; r < n; r++) {
                HIP_TRY(hipSetDevice(hs[r]->device));
                if (hs[r]->peer_l) HIP_TRY(hipStreamWaitEvent(hs[r]->s_comm, hs[r]->peer_l->ev_state, 0));
                if (hs[r]->peer_r) HIP_TRY(hipStreamWaitEvent(hs[r]->s_comm, hs[r]->peer_r->ev_state, 0));
            }
            for (int r = 0; r < n; r++) {
                HIP_TRY(hipSetDevice(hs[r]->device));
                WT_TRY(exchange_at_boundary(hs[r]));
            }
            for (int r = 0; r < n; r++) {                      // a peer's pass after next overwrites the lattice my copies read: it waits for them
                HIP_TRY(hipSetDevice(hs[r]->device));
                if (hs[r]->peer_l) HIP_TRY(hipStreamWaitEvent(hs[r]->s_compute, hs[r]->peer_l->ev_halo, 0));
                if (hs[r]->peer_r) HIP_TRY(hipStreamWaitEvent(hs[r]->s_compute, hs[r]->peer_r->ev_halo, 0));
            }
        }
        // refresh = 2 (agreed by the group): the renewal inside a fused pass — every exchange is enqueued, then every slab marches its interior
        // columns beside it, then the edge strips once its own ghosts have landed
        {
            int kr = 1 << 30;
            for (int r = 0; r < n && multi; r++) kr = std::min(kr, renew_stride(hs[r], nsteps - s));
            if (multi && kr > 0 && kr < (1 << 30)) {
                std::vector<char> sv((size_t)n);
                for (int r = 0; r < n; r++) {
                    HIP_TRY(hipSetDevice(hs[r]->device));
                    HIP_TRY(hipEventRecord(hs[r]->ev_state, hs[r]->s_compute));
                    sv[(size_t)r] = hs[r]->seams_valid ? 1 : 0;
                }
                for (int r = 0; r < n; r++) {
                    HIP_TRY(hipSetDevice(hs[r]->device));
                    if (hs[r]->peer_l) HIP_TRY(hipStreamWaitEvent(hs[r]->s_comm, hs[r]->peer_l->ev_state, 0));
                    if (hs[r]->peer_r) HIP_TRY(hipStreamWaitEvent(hs[r]->s_comm, hs[r]->peer_r->ev_state, 0));
                }
                for (int r = 0; r < n; r++) { HIP_TRY(hipSetDevice(hs[r]->device)); WT_TRY(halo_begin(hs[r])); }
                const bool emit_r = s + kr == nsteps;
                for (int r = 0; r < n; r++) { HIP_TRY(hipSetDevice(hs[r]->device)); WT_TRY(renew_interior(hs[r], tau, u0, emit_r, kr, sv[(size_t)r] != 0)); }
                for (int r = 0; r < n; r++) { HIP_TRY(hipSetDevice(hs[r]->device)); WT_TRY(renew_strips(hs[r], tau, u0, emit_r, kr)); }
                for (int r = 0; r < n; r++) {                  // a peer's NEXT pass overwrites the lattice my copies read: it waits for them
                    HIP_TRY(hipSetDevice(hs[r]->device));
                    if (hs[r]->peer_l) HIP_TRY(hipStreamWaitEvent(hs[r]->s_compute, hs[r]->peer_l->ev_halo, 0));
                    if (hs[r]->peer_r) HIP_TRY(hipStreamWaitEvent(hs[r]->s_compute, hs[r]->peer_r->ev_halo, 0));
                }
                s += kr;
                continue;
            }
        }
        // one pass length for the whole group: the shortest plan depth and the fewest exact ghost columns of any slab (edge
        // slabs are narrower than interior ones and may have chosen another depth; a depth-4 table also runs 2- and 3-step passes)
        int depth = 1 << 30, avail = nsteps - s;
        bool fused = true;
        for (int r = 0; r < n && fused; r++) {
            fused = hs[r]->fuse_ready;
            depth = std::min(depth, eff_depth(hs[r]));
            avail = std::min(avail, fuse_avail(hs[r], nsteps - s));
        }
        const int k = fused ? fuse_pick(depth, avail) : 0;
        fused = k > 0;
        if (fused) {                                  // k steps per pass on every slab; no exchange involved
            for (int r = 0; r < n; r++) {
                HIP_TRY(hipSetDevice(hs[r]->device));
                WT_TRY(step_fused(hs[r], tau, u0, s + k == nsteps, k));
            }
            s += k;
            continue;
        }
        const bool emit = (s == nsteps - 1);
        bool refresh = false;                             // as soon as ANY slab has no exact ghost column left, all of them refresh
        for (int r = 0; r < n && multi; r++) refresh = refresh || hs[r]->ghost_valid <= 0;
        if (refresh) {
            // every slab's comm stream must see its neighbours' finished lattices ...
            for (int r = 0; r < n; r++) {
                HIP_TRY(hipSetDevice(hs[r]->device));
                HIP_TRY(hipEventRecord(hs[r]->ev_state, hs[r]->s_compute));
            }
            for (int r = 0; r < n; r++) {
                HIP_TRY(hipSetDevice(hs[r]->device));
                if (hs[r]->peer_l) HIP_TRY(hipStreamWaitEvent(hs[r]->s_comm, hs[r]->peer_l->ev_state, 0));
                if (hs[r]->peer_r) HIP_TRY(hipStreamWaitEvent(hs[r]->s_comm, hs[r]->peer_r->ev_state, 0));
            }
            // ... and ALL refreshes are enqueued before any slab flips its lattice index
            for (int r = 0; r < n; r++) {
                HIP_TRY(hipSetDevice(hs[r]->device));
                WT_TRY(halo_begin(hs[r]));
            }
        }
        for (int r = 0; r < n; r++) {
            HIP_TRY(hipSetDevice(hs[r]->device));
            WT_TRY(step_compute(hs[r], tau, u0, emit, refresh));
        }
        if (refresh) {
            // the peers' NEXT step overwrites the lattice my copies just read: their compute
            // streams wait for my halo copies
            for (int r = 0; r < n; r++) {
                HIP_TRY(hipSetDevice(hs[r]->device));
                if (hs[r]->peer_l) HIP_TRY(hipStreamWaitEvent(hs[r]->s_compute, hs[r]->peer_l->ev_halo, 0));
                if (hs[r]->peer_r) HIP_TRY(hipStreamWaitEvent(hs[r]->s_compute, hs[r]->peer_r->ev_halo, 0));
            }
        }
        s += 1;
    }
    if (nsteps > 0) for (int r = 0; r < n; r++) hs[r]->macro_stale = false;
    return WT_OK;
}

extern "C" int wt_step_group_timed(wt_handle **hs, int n, int nsteps, double tau, double u0, float *elapsed_ms)
{
    if (!hs || n < 1) return fail(WT_ERR_ARG, "no handles");
    if (!elapsed_ms) return fail(WT_ERR_ARG, "elapsed_ms is null");
    for (int r = 0; r < n; r++) WT_TRY(check_steppable(hs[r], nsteps, tau, u0));
    WT_TRY(group_prepare(hs, n, nsteps, tau, u0));          // (the trial passes of a new plan are not step time)
    for (int r = 0; r < n; r++) {
        HIP_TRY(hipSetDevice(hs[r]->device));
        HIP_TRY(hipEventRecord(hs[r]->ev_a, hs[r]->s_compute));
    }
    WT_TRY(wt_step_group(hs, n, nsteps, tau, u0));
    for (int r = 0; r < n; r++) {
        HIP_TRY(hipSetDevice(hs[r]->device));
        HIP_TRY(hipEventRecord(hs[r]->ev_b, hs[r]->s_compute));
    }
    for (int r = 0; r < n; r++) {
        HIP_TRY(hipSetDevice(hs[r]->device));
        HIP_TRY(hipEventSynchronize(hs[r]->ev_b));
        HIP_TRY(hipEventElapsedTime(&elapsed_ms[r], hs[r]->ev_a, hs[r]->ev_b));
    }
    for (int r = 0; r < n; r++) WT_TRY(check_stuck(hs[r]));
    return WT_OK;
}

// ------------------------------------------------------------------------------------------
// populations in / out
// ------------------------------------------------------------------------------------------
template <typename T>
static int read_plane(wt_handle *h, const T *src_cols, T *host_dst)
{
    const Geom &g = h->g;
    const size_t bytes = (size_t)h->width * g.ny * sizeof(T);
    WT_TRY(ensure_stage(h, bytes));
    dim3 blk(32, 8), grd((g.ny + 31) / 32, (h->width + 31) / 32);
    hipLaunchKernelGGL(k_cols_to_rows<T>, grd, blk, 0, h->s_compute, src_cols, reinterpret_cast<T *>(h->stage), h->gl,
                       h->width, g.ny, g.pitch);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(host_dst, h->stage, bytes, hipMemcpyDeviceToHost, h->s_compute));
    HIP_TRY(hipStreamSynchronize(h->s_compute));
    return check_stuck(h);
}

template <typename T>
static int read_f_impl(wt_handle *h, void *out)
{
    const size_t n = (size_t)h->width * h->g.ny;
    for (int k = 0; k < 9; k++)
        WT_TRY(read_plane<T>(h, fptr<T>(h, h->cur) + k * h->g.plane + h->g.pitch, reinterpret_cast<T *>(out) + k * n));
    return WT_OK;
}

extern "C" int wt_read_f(wt_handle *h, void *f_out)
{
    WT_TRY(check_handle(h));
    if (!f_out) return fail(WT_ERR_ARG, "f_out is null");
    if (!h->inited) return fail(WT_ERR_STATE, "no state to read");
    HIP_TRY(hipSetDevice(h->device));
    return h->dtype == WT_F32 ? read_f_impl<float>(h, f_out) : read_f_impl<double>(h, f_out);
}

template <typename T>
static int write_f_impl(wt_handle *h, const void *in)
{
    const Geom &g = h->g;
    const size_t n = (size_t)h->width * g.ny;
    WT_TRY(ensure_stage(h, n * sizeof(T)));
    dim3 blk(32, 8), grd((h->width + 31) / 32, (g.ny + 31) / 32);
    for (int k = 0; k < 9; k++) {
        HIP_TRY(hipMemcpyAsync(h->stage, reinterpret_cast<const T *>(in) + k * n, n * sizeof(T), hipMemcpyHostToDevice,
                               h->s_compute));
        hipLaunchKernelGGL(k_rows_to_cols<T>, grd, blk, 0, h->s_compute, reinterpret_cast<const T *>(h->stage),
                           fptr<T>(h, h->cur) + k * g.plane + g.pitch, h->gl, h->width, g.ny, g.pitch, (long)h->width);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(h->s_compute));
    }
    return WT_OK;
}

extern "C" int wt_write_f(wt_handle *h, const void *f_in)
{
    WT_TRY(check_handle(h));
    if (!f_in) return fail(WT_ERR_ARG, "f_in is null");
    HIP_TRY(hipSetDevice(h->device));
    if (!h->inited) {   // make pads / ghosts finite
        WT_TRY(h->dtype == WT_F32 ? init_impl<float>(h, 0.0) : init_impl<double>(h, 0.0));
        h->cur = 0;
    }
    WT_TRY(h->dtype == WT_F32 ? write_f_impl<float>(h, f_in) : write_f_impl<double>(h, f_in));
    if (h->stuck_host) { HIP_TRY(hipStreamSynchronize(h->s_compute)); *h->stuck_host = 0; }       // the state is replaced, as by wt_init_equilibrium
    h->inited = true;
    h->seams_valid = false;
    h->macro_stale = true;   // the macro planes still hold the previous state's (rho,ux,uy)
    h->ghost_valid = 0;      // ghosts must be refreshed from the neighbours before the next step
    h->steps_done = 0;
    h->single_steps = 0;
    h->passes = 0;
    h->agree_dirty = true;
    return WT_OK;
}

// ------------------------------------------------------------------------------------------
// macro read-back, reductions, field
// ------------------------------------------------------------------------------------------
template <typename T>
static int read_macro_impl(wt_handle *h, void *rho, void *ux, void *uy)
{
    const long mp = (long)h->g.nxl * h->g.pitch;
    const T *m = reinterpret_cast<const T *>(h->macro);
    void *dst[3] = {rho, ux, uy};
    for (int a = 0; a < 3; a++)
        if (dst[a]) WT_TRY(read_plane<T>(h, m + a * mp, reinterpret_cast<T *>(dst[a])));
    return WT_OK;
}

extern "C" int wt_read_macro(wt_handle *h, void *rho, void *ux, void *uy)
{
    WT_TRY(check_handle(h));
    if (!h->inited) return fail(WT_ERR_STATE, "no state to read");
    if (h->macro_stale) return fail(WT_ERR_STATE, "wt_write_f replaced the populations: (rho,ux,uy) are emitted by the next wt_step");
    HIP_TRY(hipSetDevice(h->device));
    return h->dtype == WT_F32 ? read_macro_impl<float>(h, rho, ux, uy) : read_macro_impl<double>(h, rho, ux, uy);
}

extern "C" int wt_reduce_ranges(wt_handle *h, double u0, double *max_s, double *cp_min, double *cp_max)
{
    WT_TRY(check_handle(h));
    if (!max_s || !cp_min || !cp_max) return fail(WT_ERR_ARG, "null output");
    if (!h->inited || !h->mask_set) return fail(WT_ERR_STATE, "state or mask missing");
    if (h->macro_stale) return fail(WT_ERR_STATE, "wt_write_f replaced the populations: (rho,ux,uy) are emitted by the next wt_step");
    if (!(u0 != 0.0)) return fail(WT_ERR_ARG, "u0 must be non-zero");
    HIP_TRY(hipSetDevice(h->device));
    const long total = (long)h->width * h->g.ny;
    int nb = (int)((total + 255) / 256);
    if (nb > kReduceBlocks) nb = kReduceBlocks;
    RangePartial *dp = reinterpret_cast<RangePartial *>(h->partials);
    if (h->dtype == WT_F32)
        hipLaunchKernelGGL(k_ranges<float>, dim3(nb), dim3(256), 0, h->s_compute, (const float *)h->macro, h->mask, h->g,
                           h->gl, h->width, u0, dp);
    else
        hipLaunchKernelGGL(k_ranges<double>, dim3(nb), dim3(256), 0, h->s_compute, (const double *)h->macro, h->mask,
                           h->g, h->gl, h->width, u0, dp);
    HIP_TRY(hipGetLastError());
    RangePartial *hp = reinterpret_cast<RangePartial *>(h->partials_host);
    HIP_TRY(hipMemcpyAsync(hp, dp, nb * sizeof(RangePartial), hipMemcpyDeviceToHost, h->s_compute));
    HIP_TRY(hipStreamSynchronize(h->s_compute));
    double mx = 0.0, cmin = std::numeric_limits<double>::infinity(), cmax = -cmin;
    for (int b = 0; b < nb; b++) {
        if (hp[b].max_s > mx) mx = hp[b].max_s;
        if (hp[b].cp_min < cmin) cmin = hp[b].cp_min;
        if (hp[b].cp_max > cmax) cmax = hp[b].cp_max;
    }
    *max_s = mx; *cp_min = cmin; *cp_max = cmax;
    return check_stuck(h);
}

extern "C" int wt_forces(wt_handle *h, double *fx, double *fy, int64_t *surf, int64_t *rev)
{
    WT_TRY(check_handle(h));
    if (!fx || !fy || !surf || !rev) return fail(WT_ERR_ARG, "null output");
    if (!h->inited || !h->mask_set) return fail(WT_ERR_STATE, "state or mask missing");
    if (h->macro_stale) return fail(WT_ERR_STATE, "wt_write_f replaced the populations: (rho,ux,uy) are emitted by the next wt_step");
    HIP_TRY(hipSetDevice(h->device));
    const long total = (long)h->width * h->g.ny;
    int nb = (int)((total + 255) / 256);
    if (nb > kReduceBlocks) nb = kReduceBlocks;
    ForcePartial *dp = reinterpret_cast<ForcePartial *>(h->partials);
    if (h->dtype == WT_F32)
        hipLaunchKernelGGL(k_forces<float>, dim3(nb), dim3(256), 0, h->s_compute, (const float *)h->macro, h->mask, h->g,
                           h->gl, h->width, dp);
    else
        hipLaunchKernelGGL(k_forces<double>, dim3(nb), dim3(256), 0, h->s_compute, (const double *)h->macro, h->mask,
                           h->g, h->gl, h->width, dp);
    HIP_TRY(hipGetLastError());
    ForcePartial *hp = reinterpret_cast<ForcePartial *>(h->partials_host);
    HIP_TRY(hipMemcpyAsync(hp, dp, nb * sizeof(ForcePartial), hipMemcpyDeviceToHost, h->s_compute));
    HIP_TRY(hipStreamSynchronize(h->s_compute));
    double sx = 0.0, sy = 0.0;
    long long ns = 0, nr = 0;
    for (int b = 0; b < nb; b++) { sx += hp[b].fx; sy += hp[b].fy; ns += hp[b].surf; nr += hp[b].rev; }
    *fx = sx; *fy = sy; *surf = ns; *rev = nr;
    return check_stuck(h);
}

extern "C" int wt_clamp_events(wt_handle *h, int64_t *rho_events, int64_t *u_events)
{
    WT_TRY(check_handle(h));
    if (!rho_events || !u_events) return fail(WT_ERR_ARG, "null output");
    if (!h->inited || !h->mask_set) return fail(WT_ERR_STATE, "state or mask missing");
    if (h->macro_stale) return fail(WT_ERR_STATE, "wt_write_f replaced the populations: (rho,ux,uy) are emitted by the next wt_step");
    HIP_TRY(hipSetDevice(h->device));
    const long total = (long)h->width * h->g.ny;
    int nb = (int)((total + 255) / 256);
    if (nb > kReduceBlocks) nb = kReduceBlocks;
    ClampPartial *dp = reinterpret_cast<ClampPartial *>(h->partials);
    if (h->dtype == WT_F32)
        hipLaunchKernelGGL(k_clamp_events<float>, dim3(nb), dim3(256), 0, h->s_compute, (const float *)h->macro, h->mask, h->g, h->gl, h->width, dp);
    else
        hipLaunchKernelGGL(k_clamp_events<double>, dim3(nb), dim3(256), 0, h->s_compute, (const double *)h->macro, h->mask, h->g, h->gl, h->width, dp);
    HIP_TRY(hipGetLastError());
    ClampPartial *hp = reinterpret_cast<ClampPartial *>(h->partials_host);
    HIP_TRY(hipMemcpyAsync(hp, dp, nb * sizeof(ClampPartial), hipMemcpyDeviceToHost, h->s_compute));
    HIP_TRY(hipStreamSynchronize(h->s_compute));
    long long nr = 0, nu = 0;
    for (int b = 0; b < nb; b++) { nr += hp[b].rho_events; nu += hp[b].u_events; }
    *rho_events = nr; *u_events = nu;
    return check_stuck(h);
}

// Vorticity (html:411-417) needs uy of the columns left and right of the slab: fetch the
// neighbours' edge columns of the macro uy plane into this slab's innermost ghost columns.
// TR_RCCL: collective — every rank must be inside wt_field(WT_FIELD_VORT) together.
static int refresh_macro_ghosts(wt_handle *h)
{
    if (h->transport == TR_NONE) return fail(WT_ERR_STATE, "slab handle has no transport");
    const Geom &g = h->g;
    const size_t mp = (size_t)g.nxl * g.pitch;
    auto uy_col = [&](wt_handle *q, int i) { return reinterpret_cast<char *>(q->macro) + (2 * (size_t)q->g.nxl * q->g.pitch + (size_t)i * q->g.pitch) * q->esz; };
    (void)mp;
    HIP_TRY(hipStreamSynchronize(h->s_compute));
    if (h->transport == TR_RCCL) {
        const ncclDataType_t dt = h->dtype == WT_F32 ? ncclFloat32 : ncclFloat64;
        const size_t count = (size_t)g.pitch;
        ncclResult_t rc = ncclGroupStart();
        if (h->gl && rc == ncclSuccess) {
            rc = ncclSend(uy_col(h, h->gl), count, dt, h->rank - 1, h->comm, h->s_comm);
            if (rc == ncclSuccess) rc = ncclRecv(uy_col(h, h->gl - 1), count, dt, h->rank - 1, h->comm, h->s_comm);
        }
        if (h->gr && rc == ncclSuccess) {
            rc = ncclSend(uy_col(h, h->gl + h->width - 1), count, dt, h->rank + 1, h->comm, h->s_comm);
            if (rc == ncclSuccess) rc = ncclRecv(uy_col(h, h->gl + h->width), count, dt, h->rank + 1, h->comm, h->s_comm);
        }
        const ncclResult_t rc_end = ncclGroupEnd();            // always close the group, even after a failure
        if (rc == ncclSuccess) rc = rc_end;
        if (rc != ncclSuccess) return fail(WT_ERR_RCCL, "macro ghost-column exchange failed: %s", ncclGetErrorString(rc));
    } else {
        const size_t bytes = (size_t)g.pitch * h->esz;
        if (h->gl) {
            wt_handle *p = h->peer_l;
            HIP_TRY(hipSetDevice(p->device)); HIP_TRY(hipStreamSynchronize(p->s_compute)); HIP_TRY(hipSetDevice(h->device));
            HIP_TRY(hipMemcpyPeerAsync(uy_col(h, h->gl - 1), h->device, uy_col(p, p->gl + p->width - 1), p->device, bytes, h->s_comm));
        }
        if (h->gr) {
            wt_handle *p = h->peer_r;
            HIP_TRY(hipSetDevice(p->device)); HIP_TRY(hipStreamSynchronize(p->s_compute)); HIP_TRY(hipSetDevice(h->device));
            HIP_TRY(hipMemcpyPeerAsync(uy_col(h, h->gl + h->width), h->device, uy_col(p, p->gl), p->device, bytes, h->s_comm));
        }
    }
    HIP_TRY(hipStreamSynchronize(h->s_comm));
    return WT_OK;
}

template <typename T>
static int field_impl(wt_handle *h, int mode, double u0, double max_s, double cp_min, double cp_max, double vs, void *out)
{
    const Geom &g = h->g;
    const size_t bytes = (size_t)h->width * g.ny * sizeof(T);
    WT_TRY(ensure_stage(h, bytes));
    FieldParams<T> fp;
    fp.U0 = (T)u0; fp.maxS = (T)max_s; fp.cpMin = (T)cp_min; fp.cpMax = (T)cp_max; fp.vortScale = (T)vs; fp.mode = mode;
    dim3 blk(32, 8), grd((g.ny + 31) / 32, (h->width + 31) / 32);
    hipLaunchKernelGGL(k_field<T>, grd, blk, 0, h->s_compute, reinterpret_cast<const T *>(h->macro), h->mask, g, h->gl,
                       h->width, fp, reinterpret_cast<T *>(h->stage));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, h->stage, bytes, hipMemcpyDeviceToHost, h->s_compute));
    HIP_TRY(hipStreamSynchronize(h->s_compute));
    return check_stuck(h);
}

extern "C" int wt_field(wt_handle *h, int mode, double u0, double max_s, double cp_min, double cp_max,
                        double vort_scale, void *t_out)
{
    WT_TRY(check_handle(h));
    if (!t_out) return fail(WT_ERR_ARG, "t_out is null");
    if (mode < 0 || mode > 2) return fail(WT_ERR_ARG, "mode must be 0 (speed), 1 (cp) or 2 (vort)");
    if (!h->inited || !h->mask_set) return fail(WT_ERR_STATE, "state or mask missing");
    if (h->macro_stale) return fail(WT_ERR_STATE, "wt_write_f replaced the populations: (rho,ux,uy) are emitted by the next wt_step");
    HIP_TRY(hipSetDevice(h->device));
    if (mode == WT_FIELD_VORT && h->nranks > 1) WT_TRY(refresh_macro_ghosts(h));
    return h->dtype == WT_F32 ? field_impl<float>(h, mode, u0, max_s, cp_min, cp_max, vort_scale, t_out)
                              : field_impl<double>(h, mode, u0, max_s, cp_min, cp_max, vort_scale, t_out);
}

template <typename T>
static int render_impl(wt_handle *h, int mode, double u0, double max_s, double cp_min, double cp_max, double vs, uint8_t *out)
{
    const Geom &g = h->g;
    const size_t bytes = (size_t)h->width * g.ny * 4;
    WT_TRY(ensure_stage(h, bytes));
    FieldParams<T> fp;
    fp.U0 = (T)u0; fp.maxS = (T)max_s; fp.cpMin = (T)cp_min; fp.cpMax = (T)cp_max; fp.vortScale = (T)vs; fp.mode = mode;
    dim3 blk(32, 8), grd((g.ny + 31) / 32, (h->width + 31) / 32);
    hipLaunchKernelGGL(k_render<T>, grd, blk, 0, h->s_compute, reinterpret_cast<const T *>(h->macro), h->mask, g, h->gl,
                       h->width, fp, reinterpret_cast<uchar4 *>(h->stage));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, h->stage, bytes, hipMemcpyDeviceToHost, h->s_compute));
    HIP_TRY(hipStreamSynchronize(h->s_compute));
    return check_stuck(h);
}

extern "C" int wt_render_rgba(wt_handle *h, int mode, double u0, double max_s, double cp_min, double cp_max,
                              double vort_scale, uint8_t *rgba_out)
{
    WT_TRY(check_handle(h));
    if (!rgba_out) return fail(WT_ERR_ARG, "rgba_out is null");
    if (mode < 0 || mode > 2) return fail(WT_ERR_ARG, "mode must be 0 (speed), 1 (cp) or 2 (vort)");
    if (!h->inited || !h->mask_set) return fail(WT_ERR_STATE, "state or mask missing");
    if (h->macro_stale) return fail(WT_ERR_STATE, "wt_write_f replaced the populations: (rho,ux,uy) are emitted by the next wt_step");
    HIP_TRY(hipSetDevice(h->device));
    if (mode == WT_FIELD_VORT && h->nranks > 1) WT_TRY(refresh_macro_ghosts(h));
    return h->dtype == WT_F32 ? render_impl<float>(h, mode, u0, max_s, cp_min, cp_max, vort_scale, rgba_out)
                              : render_impl<double>(h, mode, u0, max_s, cp_min, cp_max, vort_scale, rgba_out);
}

// ------------------------------------------------------------------------------------------
// tracers
// ------------------------------------------------------------------------------------------
extern "C" int wt_advect_tracers(wt_handle *h, int n, const double *x, const double *y, double dt_frame, double u0,
                                 double dx0, double dx1, double dy0, double dy1,
                                 double *x_new, double *y_new, double *speed, uint8_t *ok)
{
    WT_TRY(check_handle(h));
    if (n < 0 || (n > 0 && (!x || !y || !x_new || !y_new || !speed || !ok))) return fail(WT_ERR_ARG, "bad tracer arrays");
    if (!h->inited || !h->mask_set) return fail(WT_ERR_STATE, "state or mask missing");
    if (h->macro_stale) return fail(WT_ERR_STATE, "wt_write_f replaced the populations: (rho,ux,uy) are emitted by the next wt_step");
    if (h->nranks > 1) return fail(WT_ERR_STATE, "tracers need the whole lattice on one handle");
    if (!(dx1 > dx0) || !(dy1 > dy0) || !(u0 != 0.0)) return fail(WT_ERR_ARG, "bad window or u0");
    if (n == 0) return WT_OK;
    HIP_TRY(hipSetDevice(h->device));
    const size_t nd = (size_t)n * sizeof(double);
    WT_TRY(ensure_stage(h, 5 * nd + (size_t)n + 64));
    char *base = reinterpret_cast<char *>(h->stage);
    double *dx = reinterpret_cast<double *>(base), *dy = dx + n, *ox = dy + n, *oy = ox + n, *os = oy + n;
    unsigned char *dok = reinterpret_cast<unsigned char *>(os + n);
    HIP_TRY(hipMemcpyAsync(dx, x, nd, hipMemcpyHostToDevice, h->s_compute));
    HIP_TRY(hipMemcpyAsync(dy, y, nd, hipMemcpyHostToDevice, h->s_compute));
    const Window w{dx0, dx1, dy0, dy1};
    const int nb = (n + 255) / 256;
    if (h->dtype == WT_F32)
        hipLaunchKernelGGL(k_advect<float>, dim3(nb), dim3(256), 0, h->s_compute, (const float *)h->macro, h->mask, h->g, h->gl, u0, w,
                           dt_frame, n, dx, dy, ox, oy, os, dok);
    else
        hipLaunchKernelGGL(k_advect<double>, dim3(nb), dim3(256), 0, h->s_compute, (const double *)h->macro, h->mask, h->g, h->gl, u0, w,
                           dt_frame, n, dx, dy, ox, oy, os, dok);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(x_new, ox, nd, hipMemcpyDeviceToHost, h->s_compute));
    HIP_TRY(hipMemcpyAsync(y_new, oy, nd, hipMemcpyDeviceToHost, h->s_compute));
    HIP_TRY(hipMemcpyAsync(speed, os, nd, hipMemcpyDeviceToHost, h->s_compute));
    HIP_TRY(hipMemcpyAsync(ok, dok, (size_t)n, hipMemcpyDeviceToHost, h->s_compute));
    HIP_TRY(hipStreamSynchronize(h->s_compute));
    return check_stuck(h);
}

// ------------------------------------------------------------------------------------------
// the page's canvas (canvas.hpp)
// ------------------------------------------------------------------------------------------
static const int CV_MAX_POLY = 4096;
static const int CV_MAX_SCALE = 8;

static int canvas_ensure(wt_handle *h, int scale)
{
    if (scale < 1 || scale > CV_MAX_SCALE) return fail(WT_ERR_ARG, "canvas scale must be 1 .. %d", CV_MAX_SCALE);
    if (h->nranks > 1) return fail(WT_ERR_STATE, "the canvas needs the whole lattice on one handle");
    if (h->cv_scale == scale) return WT_OK;
    // a handle holds ONE canvas at one scale; re-allocating it at another scale frees the particle layer and the text map.  With strokes on the
    // layer that would silently wipe a caller's trails: refused until the layer has been cleared (fade = 2) at its own scale.
    if (h->cv_scale != 0 && h->cv_layer_live)
        return fail(WT_ERR_STATE, "this handle's canvas lives at scale %d with tracer strokes on its particle layer; a canvas at scale %d would wipe them: "
                                  "clear the layer first (wt_canvas_stroke(h, %d, 2, 0, NULL)) or compose at scale %d", h->cv_scale, scale, h->cv_scale, h->cv_scale);
    const CanvasDims d = canvas_dims(scale);
    const size_t npx = (size_t)d.w * d.h;
    if (h->cv_layer) { HIP_TRY(hipFree(h->cv_layer)); h->cv_layer = nullptr; }
    if (h->cv_text) { HIP_TRY(hipFree(h->cv_text)); h->cv_text = nullptr; }
    if (h->cv_out) { HIP_TRY(hipFree(h->cv_out)); h->cv_out = nullptr; }
    h->cv_scale = 0;
    HIP_TRY(hipMalloc((void **)&h->cv_layer, npx * sizeof(double4)));
    HIP_TRY(hipMalloc((void **)&h->cv_text, npx * sizeof(float)));
    HIP_TRY(hipMalloc((void **)&h->cv_out, npx * sizeof(uchar4)));
    if (!h->cv_small) HIP_TRY(hipMalloc((void **)&h->cv_small, ((size_t)2 * CV_MAX_POLY + (size_t)3 * 308 * CV_MAX_SCALE / 8 + 64) * sizeof(double)));
    HIP_TRY(hipMemsetAsync(h->cv_layer, 0, npx * sizeof(double4), h->s_compute));
    HIP_TRY(hipMemsetAsync(h->cv_text, 0, npx * sizeof(float), h->s_compute));
    h->cv_text_set = false;
    h->cv_layer_live = false;
    h->cv_scale = scale;
    return WT_OK;
}

extern "C" int wt_canvas_stroke(wt_handle *h, int scale, int fade, int n, const double *seg)
{
    WT_TRY(check_handle(h));
    if (fade < 0 || fade > 2 || n < 0 || (n > 0 && !seg)) return fail(WT_ERR_ARG, "bad stroke arguments");
    HIP_TRY(hipSetDevice(h->device));
    WT_TRY(canvas_ensure(h, scale));
    const CanvasDims d = canvas_dims(scale);
    if (n > 0) {
        const size_t bytes = (size_t)n * 8 * sizeof(double);
        if ((size_t)n > h->cv_seg_cap) {
            if (h->cv_seg) { HIP_TRY(hipFree(h->cv_seg)); h->cv_seg = nullptr; h->cv_seg_cap = 0; }
            const size_t cap = (size_t)n + (size_t)n / 2 + 256;
            HIP_TRY(hipMalloc((void **)&h->cv_seg, cap * 8 * sizeof(double)));
            h->cv_seg_cap = cap;
        }
        for (int i = 0; i < n; i++)
            if (!(seg[8 * i + 4] >= 2.0 && seg[8 * i + 4] <= 65536.0)) return fail(WT_ERR_ARG, "segment %d: sample count out of range", i);
        HIP_TRY(hipMemcpyAsync(h->cv_seg, seg, bytes, hipMemcpyHostToDevice, h->s_compute));
    }
    if (fade == 0 && n == 0) return WT_OK;
    if (fade == 2) h->cv_layer_live = false;                      // cleared ...
    if (n > 0) h->cv_layer_live = true;                           // ... and drawn on
    hipLaunchKernelGGL(k_canvas_stroke, dim3((unsigned)((d.w + 7) / 8), (unsigned)((d.h + 7) / 8)), dim3(64), 0, h->s_compute, h->cv_layer, d, fade, n,
                       (const double *)h->cv_seg);
    HIP_TRY(hipGetLastError());
    if (n > 0) HIP_TRY(hipStreamSynchronize(h->s_compute));       // the caller's segment array may go away
    return WT_OK;
}

template <typename T>
static void canvas_launch(wt_handle *h, const CanvasArgs &a, int mode, double u0, double max_s, double cp_min, double cp_max, double vs)
{
    FieldParams<T> fp;
    fp.U0 = (T)u0; fp.maxS = (T)max_s; fp.cpMin = (T)cp_min; fp.cpMax = (T)cp_max; fp.vortScale = (T)vs; fp.mode = mode;
    hipLaunchKernelGGL(k_canvas_compose<T>, dim3((unsigned)((a.d.w + 15) / 16), (unsigned)((a.d.h + 15) / 16)), dim3(256), 0, h->s_compute,
                       reinterpret_cast<const T *>(h->macro), (const uint8_t *)h->mask, h->g, fp, a, h->cv_out);
}

extern "C" int wt_canvas_compose(wt_handle *h, int scale, int mode, double u0, double max_s, double cp_min, double cp_max, double vort_scale,
                                 const double *poly_xy, int npoly, const uint8_t *bar_rgb, const float *text_alpha, int use_trails,
                                 uint8_t *rgba_out)
{
    WT_TRY(check_handle(h));
    if (!rgba_out || !bar_rgb) return fail(WT_ERR_ARG, "null argument");
    if (mode < 0 || mode > 2) return fail(WT_ERR_ARG, "mode must be 0 (speed), 1 (cp) or 2 (vort)");
    if (npoly < 0 || npoly > CV_MAX_POLY || (npoly > 0 && !poly_xy)) return fail(WT_ERR_ARG, "polygon of 0 .. %d points expected", CV_MAX_POLY);
    if (!h->inited || !h->mask_set) return fail(WT_ERR_STATE, "state or mask missing");
    if (h->macro_stale) return fail(WT_ERR_STATE, "wt_write_f replaced the populations: (rho,ux,uy) are emitted by the next wt_step");
    HIP_TRY(hipSetDevice(h->device));
    WT_TRY(canvas_ensure(h, scale));
    const CanvasDims d = canvas_dims(scale);
    const size_t npx = (size_t)d.w * d.h;
    CanvasArgs a;
    a.d = d;
    a.layer = use_trails ? h->cv_layer : nullptr;
    a.npoly = npoly;
    a.poly = h->cv_small;
    a.bar = reinterpret_cast<const uint8_t *>(h->cv_small + 2 * CV_MAX_POLY);
    a.foil_r = 1.4 * scale / 2.0 + 0.5;
    a.pbx0 = a.pby0 = 1e300; a.pbx1 = a.pby1 = -1e300;
    for (int i = 0; i < npoly; i++) {
        a.pbx0 = std::min(a.pbx0, poly_xy[2 * i]); a.pbx1 = std::max(a.pbx1, poly_xy[2 * i]);
        a.pby0 = std::min(a.pby0, poly_xy[2 * i + 1]); a.pby1 = std::max(a.pby1, poly_xy[2 * i + 1]);
    }
    if (npoly > 0) HIP_TRY(hipMemcpyAsync(h->cv_small, poly_xy, (size_t)npoly * 2 * sizeof(double), hipMemcpyHostToDevice, h->s_compute));
    HIP_TRY(hipMemcpyAsync(h->cv_small + 2 * CV_MAX_POLY, bar_rgb, (size_t)d.ph * 3, hipMemcpyHostToDevice, h->s_compute));
    if (text_alpha) {
        HIP_TRY(hipMemcpyAsync(h->cv_text, text_alpha, npx * sizeof(float), hipMemcpyHostToDevice, h->s_compute));
        h->cv_text_set = true;
    }
    a.text = h->cv_text_set ? h->cv_text : nullptr;
    if (h->dtype == WT_F32) canvas_launch<float>(h, a, mode, u0, max_s, cp_min, cp_max, vort_scale);
    else canvas_launch<double>(h, a, mode, u0, max_s, cp_min, cp_max, vort_scale);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(rgba_out, h->cv_out, npx * 4, hipMemcpyDeviceToHost, h->s_compute));
    HIP_TRY(hipStreamSynchronize(h->s_compute));
    return check_stuck(h);
}
