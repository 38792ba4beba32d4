/*
 * windtunnel.h — C-ABI of libwindtunnel.so: the MI355X-native D2Q9 lattice-
 * Boltzmann wind tunnel that replaces the WebGL2 component of
 * 583phoenix-hue/Airfoil-CFD-Tool, pages/airfoil_flow_lbm_aerolab.html (cited
 * below as html:LINE).  The reference exports nothing (one IIFE, html:61); the
 * entry points below are its de-facto internal interface, one export per
 * reference function, so that a ctypes/cffi binding can stand where the page's
 * JS runtime stood.
 *
 * Conventions
 *   - every function returns 0 (WT_OK) or a negative wt_status; the message for
 *     the calling thread's last failure is wt_last_error().  No exceptions and
 *     no aborts cross this boundary.
 *   - host arrays are C-contiguous, shape [NY][W] (row 0 = bottom of the tunnel,
 *     x fastest), W = the handle's owned column count (W = NX for a whole-lattice
 *     handle; see wt_get_info).  Population arrays are [9][NY][W] in the
 *     reference's direction order (html:238-248).  Element type = the handle's
 *     dtype (float for WT_F32, double for WT_F64).  The caller owns every host
 *     buffer; the library owns all device memory, streams and communicators.
 *   - a handle is driven by one host thread at a time.  wt_step only enqueues
 *     work on the handle's HIP streams and returns; every read-back and wt_sync
 *     block until the device has finished.
 *   - there is NO CPU fallback: without a usable HIP device wt_create fails
 *     with WT_ERR_HIP.
 */
#ifndef WINDTUNNEL_H
#define WINDTUNNEL_H

#include <stdint.h>
#include <stddef.h>

/* The library is built with -fvisibility=hidden: the entry points below are its ONLY dynamic symbols. */
#if defined(__GNUC__)
#define WT_API __attribute__((visibility("default")))
#else
#define WT_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wt_handle wt_handle;

typedef enum wt_status {
    WT_OK = 0,
    WT_ERR_ARG = -1,     /* bad argument / shape */
    WT_ERR_HIP = -2,     /* HIP runtime failure (no device, launch error, ...) */
    WT_ERR_RCCL = -3,    /* RCCL failure */
    WT_ERR_OOM = -4,     /* device or host allocation failed */
    WT_ERR_STATE = -5    /* call not valid in the handle's current state */
} wt_status;

typedef enum wt_dtype { WT_F32 = 0, WT_F64 = 1 } wt_dtype;

/* html:32-36, 527, 953: the page's field selector */
typedef enum wt_field_mode { WT_FIELD_SPEED = 0, WT_FIELD_CP = 1, WT_FIELD_VORT = 2 } wt_field_mode;

typedef struct wt_info {
    int32_t nx_global;   /* lattice columns of the whole tunnel                   */
    int32_t ny;          /* lattice rows                                           */
    int32_t dtype;       /* wt_dtype                                               */
    int32_t device;      /* HIP device ordinal                                     */
    int32_t rank;        /* slab index, 0 = inlet side                             */
    int32_t nranks;      /* number of column slabs                                 */
    int32_t x0;          /* first owned global column                              */
    int32_t width;       /* owned columns W                                        */
    int32_t halo;        /* ghost columns kept on each interior side               */
    int32_t reserved;
    int64_t steps_done;  /* simStep count since wt_init_equilibrium / wt_write_f   */
    int64_t device_bytes;/* device memory held by the handle                       */
} wt_info;

/* ---- life cycle ------------------------------------------------------------ */

/* Replaces the WebGL context + texture/FBO creation (html:87-95, 438-469, 492-500).
 * One handle = the whole NX x NY lattice on one GPU. */
WT_API int wt_create(int nx, int ny, int dtype, int device, wt_handle **out);

/* One column slab of a lattice that is split over `nranks` handles (one GPU
 * each): rank r owns global columns [r*NX/nranks, (r+1)*NX/nranks) plus `halo`
 * ghost columns on each interior side, refreshed every `halo` steps.  New design
 * (the reference is single-context); SURVEY.md §8e. */
WT_API int wt_create_slab(int nx_global, int ny, int dtype, int device,
                   int rank, int nranks, int halo, wt_handle **out);

/* The same with the split given by the caller: edges[0] = 0 < edges[1] < ... <
 * edges[nranks] = nx_global, rank r owns [edges[r], edges[r+1]).  Every rank
 * passes the SAME array (the narrowest slab decides the steps per pass of all
 * of them).  For tunnels whose body makes some columns dearer than others:
 * the host cuts the slabs by measured cost (distributed.balanced_edges). */
WT_API int wt_create_slab_at(int nx_global, int ny, int dtype, int device,
                      int rank, int nranks, int halo, const int *edges, wt_handle **out);

WT_API int wt_destroy(wt_handle *h);
WT_API int wt_get_info(const wt_handle *h, wt_info *info);
WT_API const char *wt_last_error(void);
WT_API const char *wt_version(void);

/* Tuning knobs (no counterpart in the reference).  Every setting but "fast_math" gives bit-identical results.
 *   "fuse_steps" (0 / 1 = where it pays / 2 = always): advance SEVERAL steps per pass over the lattice — marching kernels that keep
 *       the intermediate steps in registers, body / inlet / outlet included (csrc/step_march.hpp: two steps, csrc/step_march3.hpp:
 *       three and four, csrc/step_chain.hpp: the same with workgroups whose units share their edge columns).  fp32 (even NY) and
 *       fp64 handles, whole lattices and slabs, with at least 8 local columns and a lattice below 4 GiB.  Default 1 (environment
 *       WT_FUSE2=0|1|2 overrides at wt_create); handles that are not eligible, or too small for it to pay, stay on k_step.
 *   "fuse_depth" (0 = automatic / 2 / 3 / 4): steps per pass.  Automatic, by columns per resident unit: fp32 3 from one, 4 from eight
 *       (the slabs of a split: from five); fp64 3 from four, 4 from 12; single steps below (measured: windtunnel.hip rebuild_fuse_plan).
 *       A step count that is not a multiple is finished with SHORTER FUSED passes on the
 *       same tables (5 = 3 + 2, 4 = 2 + 2 — a remainder of one step is never left behind where two passes fit); single steps only
 *       where not even two exact ghost columns / steps are left.  An fp32 tau for which the fast division is not proved takes
 *       three-step passes on a four-step plan (the IEEE four-step kernel is not built).
 *   "chain" (default 1): plain-fluid workgroups march their four units in alternating directions and hand the edge columns over
 *       through LDS instead of recomputing them (csrc/step_chain.hpp).
 *   "tune" (default 1): before the first pass on a new plan (new mask, changed option) time its units — a few trial passes into the
 *       lattice nobody reads, the populations are not touched — and cut the columns again by the measured times; the plan whose
 *       marching kernel ran fastest is kept, the modelled one included (windtunnel.hip tune_fuse_plan; once per mask: 14 ms on 4096^2, 3 ms
 *       on 1024x512; a mask that follows one which lived fewer than 16 passes - a slider being dragged - is timed only once it has
 *       lived that long itself).
 *       0 keeps the modelled cut.  Reported afterwards: "tune_rounds" (plans measured), "tune_gain" (modelled / kept kernel time).
 *   "plan_columns" (default 0 = this handle's): choose the steps per pass as for a lattice of that many local columns — a stand-alone
 *       handle that stands in for one slab of a split plans like that split's narrowest slab (distributed.measure_slab_cost).
 *   "fuse_chunk": cost limit of one marching unit in columns (0 = the default: units cut by TIME into one resident round).
 *   "fuse_sites": kept for callers of round 2; the sites per lane are fixed by the element type (fp32 2, fp64 1).
 *   "fast_div" (default 1): divide by tau through a reciprocal and fused multiply-adds instead of the IEEE division (csrc/d2q9.hpp); 0 keeps the
 *       IEEE division everywhere.  fp32: three operations (q0 = x r, e = fma(-q0, tau, x), q = fma(e, r, q0)) where an exhaustive on-device
 *       check over all 2^23 significands has PROVED the sequence equal to the IEEE quotient for this tau.  fp64 (round 5): four operations
 *       (p = x rlo, q1 = fma(x, rhi, p), e = fma(-q1, tau, x), q = fma(e, rhi, q1)) whose equality with the IEEE quotient is a theorem for
 *       every tau (a faithful quotient corrected once with a correctly rounded reciprocal: Markstein), guarded like the fp32 one against
 *       non-finite and huge populations (those waves divide in IEEE arithmetic).
 *   "fast_div_two_op" (default 1, fp32, four steps per pass): where the same exhaustive check has ALSO proved the two-operation form
 *       (p = x rlo, q = fma(x, r, p) with r + rlo = 1/tau to 2^-48) for this tau, the four-step kernel uses it; about 1 tau in 120 fails it
 *       and keeps the three-operation form.  Rank-local (not part of the cross-rank check): the same bits either way.
 *       "fast_div_two_op_active" reports the choice for the tau of the last stepping call.
 *   "selftest_tau" + the read-only "selftest_fastdiv32_3" / "selftest_fastdiv32_2" / "selftest_fastdiv64": mismatches of the three fast
 *       divisions against the IEEE quotient for that tau, counted on the device — the binary32 forms over all 2^23 significands and both
 *       signs, the binary64 form on 2^28 pseudo-random and boundary-hugging numerators (tests/test_gpu_fastdiv.py).
 *   "fast_math" (default 0, fp32, OPT-IN, NOT bit-identical): the marching kernels collide with contracted arithmetic — fused
 *       multiply-adds, v_rcp / v_rsq for the divisions and the square root (csrc/d2q9.hpp collide_contracted).  Held to BASELINE.md's
 *       tolerance against the oracle (|d rho| <= 1e-5, |d u| <= 5e-6; tests/test_gpu_fast_math.py), +19 % on 4096^2, +36 % on a
 *       544-column slab.  k_step (single steps, HBM-bound) keeps the reference arithmetic.
 *   "trim_ghosts" (default 1, slab handles, three / four steps per pass): a pass that starts with gv exact ghost columns and advances k steps
 *       leaves gv - k of them exact; the columns beyond are not marched at all (their results could never be read) — one unit list per
 *       remaining depth, cut from the kept plan's column costs the first time it is needed.  Between two refreshes a slab with halo 17 marches
 *       12, 8, 4 and 0 ghost columns per side instead of 14 four times.  "trimmed_passes" counts the passes that ran on a trimmed list.
 *   "window_overlap" (default -1 = automatic; fp32, three / four steps per pass): the layout of the marching kernels' 128-row windows.  0: windows
 *       that TILE the column — the rows beyond a window's seams come, level by level, out of halo lines a halo kernel builds before every pass
 *       from a seam buffer the pass before wrote.  1: OVERLAPPING windows — a window owns the 120 rows in its middle and carries four margin rows
 *       on either side, which lose one row of validity per level and are never stored: no halo lines, no halo kernel, no seam buffer, 128 / 120
 *       of the arithmetic, loads and stores that straddle cache lines; the workgroups of such a plan are dealt to the XCDs in contiguous runs
 *       (vertical neighbours share an L2).  The same bits either way.  Automatic: lattices — whole ones and the slabs of a split alike, by their
 *       LOCAL size — of up to 7.5 M sites overlap: the halo kernel is a fixed 10 of the 58 us of a pass on a 600-column slab (slowest real slab of the
 *       8-way split of 4096^2 19.9 -> 17.5 us per step, profiles/r05_t_slab_costs_cfg2_overlap.txt), and 1024x512 goes 11.8 -> 9.3 us per step,
 *       2048x1024 19.5 -> 14.5, 3584x2048 37.2 -> 35.4 (profiles/r05_zy_overlap_sizes.txt); larger ones tile (4096x2048: 42.5 against 46.9, 4096^2:
 *       77 against 90-97, a 2110-column slab of 4096 rows 44.1 against 46.4-48.3: half bound by their line traffic).  fp64 handles and "fast_math"
 *       always tile.  Rank-local: the steps per pass are chosen from the tiling windows' count either
 *       way, so the sequence of passes does not depend on it.  wt_get_option reports the present plan's layout.
 *   "refresh" (default 0, slab handles): how the ghost columns are renewed.  0: by a SINGLE step (k_step) whose interior columns run beside the
 *       exchange, the two edge strips (the ghost columns + one owned column: a few microseconds) after it — the step runs at the un-fused rate
 *       and the pass after it builds its halo lines by the gather path.  1: by an exchange at a PASS BOUNDARY — nothing runs beside it, and every
 *       step of the cycle is a fused one ("single_steps" stays 0, "boundary_exchanges" counts them; wt_plan_steps reports such an exchange as -2).
 *       2 (round 5): INSIDE a fused pass of k steps — the exchange runs beside the marching of the interior columns [gl + k, gl + width - k), the
 *       two edge strips (the k owned columns next to each edge and the ghost columns that stay exact) are marched once the ghosts have landed;
 *       no single step, no exchange outside a pass ("fused_renewals" counts them; wt_plan_steps reports 100 + k), and the window an exchange
 *       can hide in is a whole pass instead of one k_step.  Measured on the real 8-way split of the 4096^2 tunnel (one slab at a time, linked
 *       to itself: profiles/r05_e_slab_costs_cfg2.txt): 0 = 21.5 us per step with a copy kernel as the exchange, 22.2 with the modelled
 *       exchange over one xGMI link (67.7 us per refresh, 30 of them exposed); 2 = 23.2 either way (nothing exposed, but the strips of a
 *       four-step pass cost more behind the exchange than the one-column strips of a single step): 0 stays the default.  All slabs of a
 *       tunnel must choose alike (checked).
 *   "agree_check" (default 1, slab handles): every slab of a tunnel must take the SAME sequence of fused passes, single steps and ghost
 *       refreshes — over RCCL each rank decides alone and the exchange is collective, so a rank that decides differently is a hang.  The
 *       library therefore compares, across the slabs, everything that decides that sequence (lattice, split, halo, dtype, the options above,
 *       the documented WT_* environment presets, the device's resident wave slots, the planner's outcome): by an all-reduce inside
 *       wt_comm_init_rank and again at the first stepping call after wt_set_option / wt_set_mask / wt_init_equilibrium / wt_write_f (which
 *       are collective on slab handles: every rank makes the same call), by a host comparison in wt_step_group; a mismatch fails with
 *       WT_ERR_STATE naming the field and both values.  The fast-division verdict for a tau is agreed the same way (all-reduce(min)).
 *       A refused check stays due: the next stepping call checks again (and is refused again until the ranks agree); a handle refused inside
 *       wt_comm_init_rank has its communicator destroyed and is back to "no transport".
 *       Fingerprinted: nx, ny, nranks, halo, dtype, the split, fuse_steps, fuse_depth, fuse_chunk, chain, fast_div, fast_math, plan_columns,
 *       refresh, the device's resident wave slots, eligibility, mask set, the planner's outcome (plan ready, steps per pass, pass cap), steps
 *       done, exact ghost columns.  Deliberately rank-local, i.e. NOT compared: "tune" / WT_TUNE and the cut it measures (its trial passes
 *       exchange nothing), "trim_ghosts" (which ghost columns a rank bothers to march), "fast_div_two_op", "window_overlap", "exchange_timing".
 *       0 is for tests that mix plans on purpose.  "agree_checks" counts the checks made.
 *   "exchange_timing" (default 0, slab handles): HIP events around every ghost exchange (comm stream) and around the interior kernel and the
 *       wait that follows it (compute stream); summed over the refresh steps since the option was set by "exchange_ms", "interior_ms",
 *       "exchange_exposed_ms" (what the compute stream still waited after its interior kernel) and counted by "exchanges".
 * wt_get_option also reports "fuse_active", "fuse_units", "chain_units", "fuse_depth" (steps per pass the plan's tables are built for) /
 * "pass_depth" (steps a full pass takes for the tau of the last stepping call: 3 on a four-step fp32 plan whose tau has no proved fast
 * division) / "fuse_sites" (in use), "fuse_tiles_general", "fast_div_active", "tune_rounds", "tune_gain", "chain_downgrades" (groups of four
 * units whose chain flags failed the library's plan check and were downgraded to solo units: 0 by construction), "comm_ranks" (ncclCommCount of
 * the handle's communicator), "wave_slots", "passes" and "single_steps" (fused passes / whole k_step steps since the last
 * wt_init_equilibrium or wt_write_f).
 *
 * Environment.  Five variables preset a handle's options at wt_create and are part of the interface; four of them enter the cross-rank check
 * above through the option they set, WT_TUNE does not (the measured cut is a rank's own affair):
 *   WT_FUSE2=0|1|2 ("fuse_steps"), WT_FUSE_CHUNK=n ("fuse_chunk"), WT_FAST_DIV=0|1 ("fast_div"), WT_CHAIN=0|1 ("chain"), WT_TUNE=0|1 ("tune").
 * Nothing else in the environment reaches a production library.  The planner constants and launch orders the experiments under tools/ vary
 * (WT_PLAN_TIMED, WT_ALPHA, WT_ALPHA_SOLID, WT_BETA, WT_MAX_CHAIN, WT_MARCH_WAVES, WT_MARCH_ROUNDS, WT_MARCH_REV, WT_DEPTH3_MIN, WT_DEPTH4_MIN,
 * WT_TUNE_ROUNDS, WT_TUNE_DAMP, WT_TUNE_TRACE) are read only by a library built with -DWT_EXPERIMENT_KNOBS (`make lib EXPERIMENT=1`). */
WT_API int wt_set_option(wt_handle *h, const char *name, double value);
WT_API int wt_get_option(const wt_handle *h, const char *name, double *value);

/* ---- ghost-column transport for slab handles -------------------------------- */

/* RCCL over xGMI, one process per GPU: rank 0 calls wt_comm_unique_id, the host
 * broadcasts the WT_COMM_ID_BYTES bytes (e.g. torch.distributed), then EVERY
 * rank calls wt_comm_init_rank (collective).
 * WHICH RCCL: the library has no link-time dependency on one.  The first wt_comm_* call of the process binds the ncclXxx entry points to the
 * ONE RCCL the process holds — the symbols already in the global scope (an LD_PRELOADed stand-in), else the single librccl.so* in
 * /proc/self/maps (PyTorch's copy once torch is imported), else dlopen("librccl.so.1") through the library's run path (/opt/rocm/lib) — and
 * wt_version() then names its version and path.  Two different librccl.so* files mapped at that moment make every wt_comm_* call return
 * WT_ERR_RCCL naming both (csrc/rccl_bind.hpp, tests/test_rccl_binding.py): a process must not talk to two RCCL instances by accident. */
#define WT_COMM_ID_BYTES 128
WT_API int wt_comm_unique_id(void *id_out);
WT_API int wt_comm_init_rank(wt_handle *h, const void *id);
/* One-GPU check of the RCCL plumbing: one-rank communicator + the grouped send/recv pattern of the
 * ghost exchange (to self).  0 on success. */
WT_API int wt_comm_selftest(int device, int ny);

/* In-process transport: all slabs of one tunnel live in the calling process
 * (any mix of devices); ghost columns move by peer copies.  `hs` are the
 * nranks slab handles ordered by rank.  Stepping then goes through
 * wt_step_group, which advances every slab in lock-step.
 * n = 1 with a slab handle links that slab to ITSELF: its ghost columns are refreshed from its own owned edge columns (a tunnel periodic in x
 * over this slab; needs width >= 2 halo).  The timing stand-in for one slab of a split — trimmed ghost marching, the refresh mode, the exchange
 * beside the interior all run, on one handle alone on the GPU (distributed.measure_slab_real, tools/r5_slab_costs.py). */
WT_API int wt_link_local(wt_handle **hs, int n);
WT_API int wt_step_group(wt_handle **hs, int n, int nsteps, double tau, double u0);
/* As wt_step_group, every slab's share bracketed by HIP events on that slab's compute stream; blocks, and returns the
 * elapsed device time of each slab in milliseconds (elapsed_ms[n]).  Measurement only (bench.py --local-slabs). */
WT_API int wt_step_group_timed(wt_handle **hs, int n, int nsteps, double tau, double u0, float *elapsed_ms);

/* ---- state ------------------------------------------------------------------ */

/* Replaces makeMaskTex/applyGeometry's upload (html:448-458, 579-586).
 * `mask` is the WHOLE tunnel's mask, [NY][NX_global] uint8, non-zero = solid;
 * a slab handle takes the columns it needs (owned + ghost).  The flow state is
 * NOT re-initialised (html:579-586). */
WT_API int wt_set_mask(wt_handle *h, const uint8_t *mask);

/* Replaces initSim/equilibriumInitData (html:474-500): every site, solids
 * included, gets feq(rho=1,u=(u0,0)) evaluated in double and stored in the
 * handle's dtype; macro = (1,u0,0). */
WT_API int wt_init_equilibrium(wt_handle *h, double u0);

/* Replaces `nsteps` x simStep (html:510-525) = STEP_FS main() (html:283-360) per
 * site: pull-stream, half-way bounce-back, far-field/outlet BC, moments, clamp,
 * BGK.  tau and u0 are the shader uniforms (html:521-522), converted to the
 * handle's dtype.  The LAST step of the call also stores the macroscopic
 * fields (rho,ux,uy) that the reference writes into texC (html:357-359). */
WT_API int wt_step(wt_handle *h, int nsteps, double tau, double u0);

/* How `nsteps` steps would be taken from the handle's present state, without taking them: seq[i] = +k for a fused pass of k steps, 1 for a
 * single step, -1 for a single step that refreshes the ghost columns first; returns the length of the sequence (at most `cap` entries are
 * written).  Slab handles of one tunnel answer alike — the property the collective exchange over RCCL rests on. */
WT_API int wt_plan_steps(wt_handle *h, int nsteps, double tau, int *seq, int cap);

/* As wt_step, bracketed by HIP events on the stream the step kernels run on;
 * blocks, and returns the elapsed device time in milliseconds. */
WT_API int wt_step_timed(wt_handle *h, int nsteps, double tau, double u0, float *elapsed_ms);

/* Test / checkpoint access to the populations, [9][NY][W].
 * wt_write_f replaces the populations only: the (rho,ux,uy) planes the reference keeps in texC are the pre-collision
 * moments of the step that PRODUCED a state (html:357-359) and cannot be derived from a restored state alone, so after
 * wt_write_f the calls that read them (wt_read_macro, wt_reduce_ranges, wt_forces, wt_field, wt_render_rgba,
 * wt_advect_tracers) return WT_ERR_STATE until a wt_step / wt_step_group with nsteps >= 1 has emitted them again.
 * Slab handles: ghost columns are refreshed from the neighbours before the next step. */
WT_API int wt_read_f(wt_handle *h, void *f_out);
WT_API int wt_write_f(wt_handle *h, const void *f_in);

/* ---- read-backs and reductions ---------------------------------------------- */

/* Replaces readMacro (html:547-552): rho, ux, uy of the last step, each [NY][W]. */
WT_API int wt_read_macro(wt_handle *h, void *rho, void *ux, void *uy);

/* Replaces the range scan of updateFieldsFromMacro (html:596-614), in doubles
 * over fluid sites: max_s = max hypot(ux,uy)/u0 over values < 4 (0 if none);
 * cp range over -4 < cp < 1.2 with cp=(rho-1)/(1.5 u0^2) (+inf/-inf if none).
 * Partial result of the handle's owned columns; the "keep the previous value"
 * rule (html:611-613) stays with the caller, as in the page. */
WT_API int wt_reduce_ranges(wt_handle *h, double u0, double *max_s, double *cp_min, double *cp_max);

/* Replaces computeForces' two scans (html:650-698), raw sums over the fluid/
 * solid faces whose FLUID cell is owned: fx,fy = sum of (rho_fluid/3) * (unit
 * vector from the fluid cell into the solid); surf = face count; rev = faces
 * whose fluid cell has ux < 0.  Coefficients and smoothing stay with the caller
 * (html:676-679, 699). */
WT_API int wt_forces(wt_handle *h, double *fx, double *fy, int64_t *surf, int64_t *rev);

/* Diagnostics of the stability net STEP_FS applies silently (html:344-350; SURVEY §5 "clamp-event counter"): the number
 * of owned fluid sites whose last emitted state sits at a density bound (rho = 0.5 or 2.0) and at the speed bound
 * (|u| = 0.35).  Both are 0 in a healthy run.  Per-slab partials, like wt_forces. */
WT_API int wt_clamp_events(wt_handle *h, int64_t *rho_events, int64_t *u_events);

/* Replaces RENDER_FS main()'s field math (html:395-420): the scalar t handed to
 * the colour map, [NY][W], NaN on solid sites.  max_s/cp_min/cp_max are the
 * uniforms of html:540-542, vort_scale html:528/543. */
WT_API int wt_field(wt_handle *h, int mode, double u0, double max_s, double cp_min, double cp_max,
             double vort_scale, void *t_out);

/* Replaces RENDER_FS's colour maps (html:371-393, 397): RGBA8 image, [NY][W][4],
 * row 0 = bottom. */
WT_API int wt_render_rgba(wt_handle *h, int mode, double u0, double max_s, double cp_min, double cp_max,
                   double vort_scale, uint8_t *rgba_out);

/* Replaces advect() (html:758-771) over sampleUV/sampleScalar (html:616-639) for n tracer
 * particles at world positions (x,y): midpoint step through the velocity field normalised by u0
 * (bilinear over fluid cells only).  dt_frame is the frame time in ms (html:903); the window is
 * html:73's DX0,DX1,DY0,DY1.  ok[i]=0 where the reference returns null (outside the window or no
 * fluid sample): x_new/y_new then repeat the input.  Whole-lattice handles only. */
WT_API int wt_advect_tracers(wt_handle *h, int n, const double *x, const double *y, double dt_frame, double u0,
                      double dx0, double dx1, double dy0, double dy1,
                      double *x_new, double *y_new, double *speed, uint8_t *ok);

/* ---- the page's 2-D canvas on the device (csrc/canvas.hpp) ---------------------------------
 * Replaces the canvas drawing of frame() (html:919-927) for a whole-lattice handle: 680*scale x 360*scale pixels, top row first.
 * Round 3 composited the frame with NumPy on the host (compose.py, which stays as the host path and as this one's test reference); the
 * page's loop then ran at 9 frames per second at any lattice size, all of it host time (profiles/r04_a_frame_loop_host_canvas.txt).
 *
 * wt_canvas_stroke: the particle layer `pcv` (html:780-808), kept in device memory per handle.  fade: 0 none, 1 the frame's
 *   destination-out fade (x 0.945), 2 clear.  Then n segments are stroked in order: seg[i] = {x0, y0, x1, y1 in canvas pixels, ns = points
 *   sampled along the segment (>= 2), r, g, b = the stroke's colour 0..255}; a round brush of radius 1.1 scale / 2 + 0.5, alpha 0.75.
 * wt_canvas_compose: one frame — background, the field image of wt_render_rgba resampled into the plot rectangle (drawImage, html:923),
 *   the particle layer when use_trails != 0, the foil polygon poly_xy[npoly][2] (canvas pixels) filled (#0d1018) and outlined (html:815-828),
 *   the colour bar bar_rgb[308*scale][3] (html:830-848) and the labels: text_alpha[360*scale][680*scale] = alpha of the white text at every
 *   pixel (0 = none), or NULL to keep the map of the previous call (it changes with the angle and the field only).  mode .. vort_scale as
 *   in wt_render_rgba.  rgba_out: [360*scale][680*scale][4]. */
WT_API int wt_canvas_stroke(wt_handle *h, int scale, int fade, int n, const double *seg);
WT_API int wt_canvas_compose(wt_handle *h, int scale, int mode, double u0, double max_s, double cp_min, double cp_max, double vort_scale,
                             const double *poly_xy, int npoly, const uint8_t *bar_rgb, const float *text_alpha, int use_trails,
                             uint8_t *rgba_out);

WT_API int wt_sync(wt_handle *h);

#ifdef __cplusplus
}
#endif
#endif /* WINDTUNNEL_H */
