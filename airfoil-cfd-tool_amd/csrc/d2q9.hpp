// d2q9.hpp — per-site D2Q9 arithmetic shared by every step kernel.
//
// Follows STEP_FS of the reference (pages/airfoil_flow_lbm_aerolab.html:234-281,
// 335-359).  Arithmetic contract = the oracle's: IEEE, literal evaluation order,
// one rounding per operation, no FMA contraction (the library is compiled with
// -ffp-contract=off), IEEE-correct division and sqrt.  The only liberties taken
// are exact identities: x*1 = x, x+0 = x, -(a)+(-b) = -(a+b), (c*(-e))*(-e) =
// (c*e)*e — each preserves every bit of the result.
#pragma once
#include <hip/hip_runtime.h>

namespace wt {

// html:238-248 (dir) and html:254-264 (opp)
__host__ __device__ constexpr int ex_of(int k) { return (k == 1 || k == 5 || k == 8) ? 1 : ((k == 3 || k == 6 || k == 7) ? -1 : 0); }
__host__ __device__ constexpr int ey_of(int k) { return (k == 2 || k == 5 || k == 6) ? 1 : ((k == 4 || k == 7 || k == 8) ? -1 : 0); }
__host__ __device__ constexpr int opp_of(int k) { return k == 0 ? 0 : (k <= 4 ? ((k + 1) % 4) + 1 : ((k - 3) % 4) + 5); }
static_assert(opp_of(1) == 3 && opp_of(2) == 4 && opp_of(3) == 1 && opp_of(4) == 2, "opp");
static_assert(opp_of(5) == 7 && opp_of(6) == 8 && opp_of(7) == 5 && opp_of(8) == 6, "opp");

template <typename T> __device__ __forceinline__ T wt_sqrt(T x);
template <> __device__ __forceinline__ float wt_sqrt<float>(float x) { return sqrtf(x); }
template <> __device__ __forceinline__ double wt_sqrt<double>(double x) { return sqrt(x); }

// html:276-281 for all nine directions at once:
//   feq_k = wt(k)*rho*(1.0+3.0*eu+4.5*eu*eu-1.5*uu),  eu = e_k.u
template <typename T>
__device__ __forceinline__ void feq_all(T rho, T ux, T uy, T (&eq)[9])
{
    const T w0 = T(4.0) / T(9.0), ws = T(1.0) / T(9.0), wd = T(1.0) / T(36.0);   // html:234-236
    const T uu = ux * ux + uy * uy;
    const T c = T(1.5) * uu;
    const T w0r = w0 * rho, wsr = ws * rho, wdr = wd * rho;
    const T one = T(1.0);
    // k=0: eu = 0
    eq[0] = w0r * (one - c);
    // k=1/3: eu = +-ux
    {
        const T a = T(3.0) * ux, b = (T(4.5) * ux) * ux;
        eq[1] = wsr * (((one + a) + b) - c);
        eq[3] = wsr * (((one - a) + b) - c);
    }
    // k=2/4: eu = +-uy
    {
        const T a = T(3.0) * uy, b = (T(4.5) * uy) * uy;
        eq[2] = wsr * (((one + a) + b) - c);
        eq[4] = wsr * (((one - a) + b) - c);
    }
    // k=5/7: eu = +-(ux+uy)
    {
        const T e = ux + uy;
        const T a = T(3.0) * e, b = (T(4.5) * e) * e;
        eq[5] = wdr * (((one + a) + b) - c);
        eq[7] = wdr * (((one - a) + b) - c);
    }
    // k=6/8: eu = -+(ux-uy);  k=6: -ux+uy, k=8: ux-uy
    {
        const T e = ux - uy;
        const T a = T(3.0) * e, b = (T(4.5) * e) * e;
        eq[8] = wdr * (((one + a) + b) - c);
        eq[6] = wdr * (((one - a) + b) - c);
    }
}

// html:335-338: moments of nine populations (rho summed sequentially from 0.0)
template <typename T>
__device__ __forceinline__ void moments(const T (&f)[9], T &rho, T &ux, T &uy)
{
    T r = T(0.0);
#pragma unroll
    for (int k = 0; k < 9; k++) r += f[k];
    rho = r;
    ux = (f[1] + f[5] + f[8] - f[3] - f[6] - f[7]) / r;
    uy = (f[2] + f[5] + f[6] - f[4] - f[7] - f[8]) / r;
}

// --------------------------------------------------------------------------------------
// Division by the relaxation time (html:352-356: fin - (fin - feq)/tau, nine per site).
//
// tau is one value for the whole launch, so the generic IEEE expansion (v_div_scale x2,
// v_rcp, five fma, v_div_fmas, v_div_fixup: 11 VALU instructions and one transcendental
// issue slot, 44 % of the step's arithmetic) can be replaced by a short sequence on
// r = RN(1/tau) and rlo = RN(1/tau - r):
//   three operations (rounds 1-4):  q0 = RN(x r);  e = fma(-q0, tau, x)  (exact residual);  q = fma(e, r, q0)
//   two operations (round 5):       p = RN(x rlo);  q = fma(x, r, p)     (r + rlo = 1/tau to 2^-48: q = RN(x/tau (1 + ~2^-48)))
// Either returns the correctly rounded x/tau for every x of one binade iff it does for all
// 2^23 significands (every operation commutes with an exact scaling by a power of two as
// long as nothing over- or underflows).  The library PROVES this per tau and per form before
// using it: k_verify_fastdiv below compares both sequences with the IEEE quotient for all 2^23
// significands on the device (a few microseconds, cached per tau); a tau that fails the
// two-operation form (about 1 in 120: tools/fastdiv_check.c, profiles/r05_a_fastdiv_forms.txt)
// keeps the three-operation one, a tau that fails both keeps the IEEE division (none seen in
// 3 400 random tau).  Range argument for the scaling: after the clamp (html:344-350)
// feq >= (1/36)*0.5*0.43 > 2^-8, so x = fin - feq is 0 or |x| >= 2^-32 (a multiple of
// ulp(2^-9)), the residual is a multiple of 2^-80 and x rlo >= 2^-58 — far from the subnormal
// range — and x = +-0 gives +-0 in all forms (x = -0 cannot occur: fin - feq = -0 needs feq = +0).
// Sites whose populations are not all below 2^100 in magnitude (blown-up or non-finite states)
// take the IEEE division.
//
// binary64 (round 5): 2^52 significands cannot be enumerated, so the fp64 sequence is one whose
// correctness is a THEOREM for every tau and x (no over- / underflow: same guard):
//     p = RN(x rlo);  q1 = fma(x, rhi, p);  e = fma(-q1, tau, x);  q = fma(e, rhi, q1)       rhi = RN(1/tau), rlo = RN((1 - rhi tau)/tau)
// q1 is RN of a value within 2^-104 |x/tau| of x/tau, hence a FAITHFUL quotient (one of the two doubles around x/tau); rhi approximates 1/tau with
// relative error < 2^-53; then e is exact and q = RN(x/tau) by Markstein's theorem (P. Markstein, "Computation of elementary functions on the IBM
// RISC System/6000 processor", IBM J. Res. Dev. 34, 1990; Muller et al., Handbook of Floating-Point Arithmetic, "Newton-Raphson-based division
// with an FMA": a faithful quotient corrected once with a reciprocal good to half an ulp is the correctly rounded quotient).  Cross-checked on the
// CPU (tools/fastdiv_check.c: 10^9 (x, tau) pairs, half of them built next to rounding boundaries, and the binary32 analogue of the same
// four operations exhaustively for 3 400 tau: no mismatch) and on the device (tests/test_gpu_fastdiv.py).
// --------------------------------------------------------------------------------------
struct FastDiv {
    float tau, rtau, rlo;                // binary32: tau, RN(1/tau), RN(1/tau - rtau)
    int on64;                            // binary64: use the four-operation sequence (option "fast_div"); 0 = IEEE division
    double tau64, rhi64, rlo64;          // binary64: tau, RN(1/tau), RN((1 - rhi tau)/tau)
};
static inline FastDiv make_fastdiv(double tau)
{
    FastDiv f;
    f.tau = (float)tau;
    f.rtau = 1.0f / f.tau;
    f.rlo = (float)(1.0 / (double)f.tau - (double)f.rtau);
    f.on64 = 0;
    f.tau64 = tau;
    f.rhi64 = 1.0 / tau;
    f.rlo64 = __builtin_fma(-f.rhi64, tau, 1.0) / tau;
    return f;
}

template <bool TWO_OP = false>
__device__ __forceinline__ float div_by_tau_fast(float x, const FastDiv &fd)
{
    if constexpr (TWO_OP) {
        const float p = x * fd.rlo;
        return __builtin_fmaf(x, fd.rtau, p);
    } else {
        const float q0 = x * fd.rtau;
        const float e = __builtin_fmaf(-q0, fd.tau, x);
        return __builtin_fmaf(e, fd.rtau, q0);
    }
}
__device__ __forceinline__ double div_by_tau_fast64(double x, const FastDiv &fd)
{
    const double p = x * fd.rlo64;
    const double q1 = __builtin_fma(x, fd.rhi64, p);
    const double e = __builtin_fma(-q1, fd.tau64, x);
    return __builtin_fma(e, fd.rhi64, q1);
}

// exhaustive proof for one tau: counts significands whose fast quotient differs from x/tau — nbad[0]: the three-operation form, nbad[1]: the two-operation form
__global__ void k_verify_fastdiv(FastDiv fd, unsigned int *__restrict__ nbad)
{
    unsigned int bad3 = 0, bad2 = 0;
    const float tau = fd.tau;
    for (unsigned int m = blockIdx.x * blockDim.x + threadIdx.x; m < (1u << 23); m += gridDim.x * blockDim.x) {
        const float x = __uint_as_float(0x3f800000u | m);            // [1, 2)
        const float a = x / tau, b = div_by_tau_fast<false>(x, fd), c = div_by_tau_fast<true>(x, fd);
        bad3 += (__float_as_uint(a) != __float_as_uint(b));
        bad2 += (__float_as_uint(a) != __float_as_uint(c));
        const float xn = -x;                                         // the sequences are odd in x; checked anyway
        const float an = xn / tau, bn = div_by_tau_fast<false>(xn, fd), cn = div_by_tau_fast<true>(xn, fd);
        bad3 += (__float_as_uint(an) != __float_as_uint(bn));
        bad2 += (__float_as_uint(an) != __float_as_uint(cn));
    }
    if (bad3) atomicAdd(nbad, bad3);
    if (bad2) atomicAdd(nbad + 1, bad2);
}
// the binary64 sequence against the IEEE quotient on `n` pseudo-random significands per thread and on values built next to rounding boundaries
// (x = RN(m tau) +- a few ulps for midpoints m): a TEST of the theorem's preconditions as implemented (tests/test_gpu_fastdiv.py), not the proof
__global__ void k_check_fastdiv64(FastDiv fd, unsigned long long seed, int n, unsigned int *__restrict__ nbad)
{
    unsigned long long s = seed + 0x9e3779b97f4a7c15ULL * (unsigned long long)(blockIdx.x * blockDim.x + threadIdx.x + 1);
    unsigned int bad = 0;
    for (int i = 0; i < n; i++) {
        s += 0x9e3779b97f4a7c15ULL;
        unsigned long long z = s;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; z ^= z >> 31;
        double x = __longlong_as_double((long long)(0x3ff0000000000000ULL | (z & 0xfffffffffffffULL)));
        if (i & 1) {
            const double mlo = x;                                               // a double of [1, 2); the midpoint above it is mlo + 2^-53
            const double xm = __builtin_fma(mlo, fd.tau64, 0x1p-53 * fd.tau64);
            long long xb = __double_as_longlong(xm) + (long long)((z >> 60) & 7ULL) - 3;      // RN(m tau) and its neighbours
            x = __longlong_as_double(xb);
        }
        const double a = x / fd.tau64, b = div_by_tau_fast64(x, fd);
        bad += (__double_as_longlong(a) != __double_as_longlong(b));
        const double an = (-x) / fd.tau64, bn = div_by_tau_fast64(-x, fd);
        bad += (__double_as_longlong(an) != __double_as_longlong(bn));
    }
    if (bad) atomicAdd(nbad, bad);
}

// html:335-359: moments, stability clamp, BGK relaxation.  `fin` are the
// post-stream populations; returns the post-collision populations and the
// clamped pre-collision (rho,ux,uy) the reference stores in texC.
// FD = 1: fp32 only, division by tau through div_by_tau_fast (proved per tau, see above); 0: IEEE division;
// 2: fast division without the magnitude guard (experiments only).  TWO_OP: the two-operation form (proved per tau as well).
template <int FD, bool TWO_OP = false>
__device__ __forceinline__ void collide_fd(const float (&fin)[9], const FastDiv &fd, float (&fo)[9], float &rho, float &ux, float &uy)
{
    float r, u, v;
    moments(fin, r, u, v);
    const float uMax = 0.35f, rhoMin = 0.5f, rhoMax = 2.0f;        // html:344
    r = (r < rhoMin) ? rhoMin : r;
    r = (rhoMax < r) ? rhoMax : r;
    const float spd2 = u * u + v * v;
    if (spd2 > uMax * uMax) {
        const float k = uMax / wt_sqrt<float>(spd2);
        u *= k;
        v *= k;
    }
    float eq[9];
    feq_all(r, u, v, eq);
    bool fast = FD != 0;
    if (FD == 1) {
        // all |fin| < 2^100 (NaN compares false -> IEEE path)
        const float m0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(fin[0]), __builtin_fabsf(fin[1])), __builtin_fabsf(fin[2]));
        const float m1 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(fin[3]), __builtin_fabsf(fin[4])), __builtin_fabsf(fin[5]));
        const float m2 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(fin[6]), __builtin_fabsf(fin[7])), __builtin_fabsf(fin[8]));
        const float m = __builtin_fmaxf(__builtin_fmaxf(m0, m1), m2);
        fast = (m < 0x1p100f) && (r == r) && (spd2 == spd2);
    }
    if (fast) {
#pragma unroll
        for (int k = 0; k < 9; k++) fo[k] = fin[k] - div_by_tau_fast<TWO_OP>(fin[k] - eq[k], fd);
    } else {
#pragma unroll
        for (int k = 0; k < 9; k++) fo[k] = fin[k] - (fin[k] - eq[k]) / fd.tau;   // html:352-356
    }
    rho = r;
    ux = u;
    uy = v;
}

// The same collision in two stages, for kernels that collide several sites per lane and want ONE decision
// "fast or IEEE division" for all of them (a branch per site costs more than the test saves):
//   collide_head: moments and density clamp (html:335-346); `safe` = this site may use the fast division
//   collide_tail<FAST>: velocity clamp (html:347-350), equilibrium, relaxation (html:352-356)
// Operation for operation the sequence of collide_fd.
__device__ __forceinline__ void collide_head(const float (&fin)[9], float &r, float &u, float &v, float &spd2, bool &safe)
{
    moments(fin, r, u, v);
    const float rhoMin = 0.5f, rhoMax = 2.0f;
    r = (r < rhoMin) ? rhoMin : r;
    r = (rhoMax < r) ? rhoMax : r;
    spd2 = u * u + v * v;
    const float m0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(fin[0]), __builtin_fabsf(fin[1])), __builtin_fabsf(fin[2]));
    const float m1 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(fin[3]), __builtin_fabsf(fin[4])), __builtin_fabsf(fin[5]));
    const float m2 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(fin[6]), __builtin_fabsf(fin[7])), __builtin_fabsf(fin[8]));
    const float m = __builtin_fmaxf(__builtin_fmaxf(m0, m1), m2);
    safe = (m < 0x1p100f) && (r == r) && (spd2 == spd2);
}

template <bool FAST, bool TWO_OP = false>
__device__ __forceinline__ void collide_tail(const float (&fin)[9], const FastDiv &fd, float r, float &u, float &v, float spd2, float (&fo)[9])
{
    const float uMax = 0.35f;
    if (spd2 > uMax * uMax) {
        const float k = uMax / wt_sqrt<float>(spd2);
        u *= k;
        v *= k;
    }
    float eq[9];
    feq_all(r, u, v, eq);
#pragma unroll
    for (int k = 0; k < 9; k++) fo[k] = fin[k] - (FAST ? div_by_tau_fast<TWO_OP>(fin[k] - eq[k], fd) : (fin[k] - eq[k]) / fd.tau);
}

// --------------------------------------------------------------------------------------
// OPT-IN, never the default (option "fast_math"): the same collision with contracted arithmetic — fused multiply-adds in the
// equilibrium and the relaxation (fo = fin + omega (feq - fin), omega = RN(1/tau)), v_rcp_f32 / v_rsq_f32 in place of the IEEE
// divisions and the square root.  About half the vector instructions of collide_fd; results differ from the reference arithmetic
// in the last bits of every operation, so this path is held to BASELINE.md's tolerance (|d rho| <= 1e-5, |d u| <= 5e-6 against the
// oracle, tests/test_gpu_fast_math.py), not to bit-equality.  Same operations in the same order as html:335-356 otherwise.
// --------------------------------------------------------------------------------------
__device__ __forceinline__ void collide_contracted(const float (&fin)[9], float omega, float (&fo)[9], float &rho, float &ux, float &uy)
{
    float r = 0.0f;
#pragma unroll
    for (int k = 0; k < 9; k++) r += fin[k];
    const float inv = __builtin_amdgcn_rcpf(r);
    float u = (fin[1] + fin[5] + fin[8] - fin[3] - fin[6] - fin[7]) * inv;
    float v = (fin[2] + fin[5] + fin[6] - fin[4] - fin[7] - fin[8]) * inv;
    const float uMax = 0.35f, rhoMin = 0.5f, rhoMax = 2.0f;        // html:344
    r = (r < rhoMin) ? rhoMin : r;
    r = (rhoMax < r) ? rhoMax : r;
    const float spd2 = __builtin_fmaf(u, u, v * v);
    if (spd2 > uMax * uMax) {
        const float k = uMax * __builtin_amdgcn_rsqf(spd2);
        u *= k;
        v *= k;
    }
    const float w0r = (4.0f / 9.0f) * r, wsr = (1.0f / 9.0f) * r, wdr = (1.0f / 36.0f) * r;
    const float uu = __builtin_fmaf(u, u, v * v);
    const float c = __builtin_fmaf(-1.5f, uu, 1.0f);              // 1 - 1.5 uu
    float eq[9];
    eq[0] = w0r * c;
    auto pair = [&](float e, float w, float &plus, float &minus) {
        const float t = __builtin_fmaf(4.5f * e, e, c);           // 1 + 4.5 e^2 - 1.5 uu
        plus = w * __builtin_fmaf(3.0f, e, t);
        minus = w * __builtin_fmaf(-3.0f, e, t);
    };
    pair(u, wsr, eq[1], eq[3]);
    pair(v, wsr, eq[2], eq[4]);
    pair(u + v, wdr, eq[5], eq[7]);
    pair(u - v, wdr, eq[8], eq[6]);
#pragma unroll
    for (int k = 0; k < 9; k++) fo[k] = __builtin_fmaf(omega, eq[k] - fin[k], fin[k]);
    rho = r;
    ux = u;
    uy = v;
}

template <typename T>
__device__ __forceinline__ void collide(const T (&fin)[9], T tau, T (&fo)[9], T &rho, T &ux, T &uy)
{
    T r, u, v;
    moments(fin, r, u, v);
    const T uMax = T(0.35), rhoMin = T(0.5), rhoMax = T(2.0);   // html:344
    r = (r < rhoMin) ? rhoMin : r;                              // clamp = min(max(x,lo),hi)
    r = (rhoMax < r) ? rhoMax : r;
    const T spd2 = u * u + v * v;
    if (spd2 > uMax * uMax) {
        const T k = uMax / wt_sqrt<T>(spd2);
        u *= k;
        v *= k;
    }
    T eq[9];
    feq_all(r, u, v, eq);
#pragma unroll
    for (int k = 0; k < 9; k++) fo[k] = fin[k] - (fin[k] - eq[k]) / tau;   // html:352-356
    rho = r;
    ux = u;
    uy = v;
}

// binary64 with the four-operation division by tau (see "Division by the relaxation time"): operation for operation collide<double> up to the
// relaxation; the division is the fast one iff every lane of the wave holds finite populations below 2^100 (else, and with fd.on64 = 0, IEEE).
__device__ __forceinline__ void collide_fd64(const double (&fin)[9], const FastDiv &fd, double (&fo)[9], double &rho, double &ux, double &uy)
{
    double r, u, v;
    moments(fin, r, u, v);
    const double uMax = 0.35, rhoMin = 0.5, rhoMax = 2.0;       // html:344
    r = (r < rhoMin) ? rhoMin : r;
    r = (rhoMax < r) ? rhoMax : r;
    const double spd2 = u * u + v * v;
    if (spd2 > uMax * uMax) {
        const double k = uMax / wt_sqrt<double>(spd2);
        u *= k;
        v *= k;
    }
    double eq[9];
    feq_all(r, u, v, eq);
    const double m0 = __builtin_fmax(__builtin_fmax(__builtin_fabs(fin[0]), __builtin_fabs(fin[1])), __builtin_fabs(fin[2]));
    const double m1 = __builtin_fmax(__builtin_fmax(__builtin_fabs(fin[3]), __builtin_fabs(fin[4])), __builtin_fabs(fin[5]));
    const double m2 = __builtin_fmax(__builtin_fmax(__builtin_fabs(fin[6]), __builtin_fabs(fin[7])), __builtin_fabs(fin[8]));
    const double m = __builtin_fmax(__builtin_fmax(m0, m1), m2);
    const bool safe = (m < 0x1p100) && (r == r) && (spd2 == spd2);         // (NaN compares false -> IEEE path)
    if (fd.on64 != 0 && __ballot(!safe) == 0ULL) {
#pragma unroll
        for (int k = 0; k < 9; k++) fo[k] = fin[k] - div_by_tau_fast64(fin[k] - eq[k], fd);
    } else {
#pragma unroll
        for (int k = 0; k < 9; k++) fo[k] = fin[k] - (fin[k] - eq[k]) / fd.tau64;   // html:352-356
    }
    rho = r;
    ux = u;
    uy = v;
}

}  // namespace wt
