/*
 * CPU ORACLE (C restatement) — test infrastructure only; never linked into or
 * called from the product path (libwindtunnel.so).  Only tests/, smoke() and
 * bench.py's cpu_baseline leg may load it.
 *
 * Restates STEP_FS main() of the reference,
 *   pages/airfoil_flow_lbm_aerolab.html:283-360 (helpers html:234-281),
 * and equilibriumInitData (html:474-490), with the same arithmetic contract as
 * oracle/lbm_numpy.py: IEEE binary32 / binary64, literal left-to-right
 * evaluation, one rounding per operation, NO FMA contraction (build with
 * -ffp-contract=off, no -ffast-math), division by tau.  It exists because the
 * NumPy transcription takes minutes on the 1024x512x2000 configuration.
 *
 * Pinning: bit-identical to lbm_numpy.py and to the reference shader text
 * executed by oracle/ref_js (tests/test_oracle_golden.py).
 *
 * Layout: f[9][NY][NX], solid[NY][NX] (non-zero = solid), row 0 = bottom.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

static const int EX[9] = {0, 1, 0, -1, 0, 1, -1, -1, 1};   /* html:238-248 */
static const int EY[9] = {0, 0, 1, 0, -1, 1, 1, -1, -1};
static const int OPP[9] = {0, 3, 4, 1, 2, 7, 8, 5, 6};     /* html:254-264 */

#define REAL float
#define SQRT sqrtf
#define NAME(x) x##_f32
#include "lbm_oracle_body.inc"
#undef REAL
#undef SQRT
#undef NAME

#define REAL double
#define SQRT sqrt
#define NAME(x) x##_f64
#include "lbm_oracle_body.inc"
#undef REAL
#undef SQRT
#undef NAME
