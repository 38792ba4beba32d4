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

// html:335-359: moments, stability clamp, BGK relaxation.  `fin` are the
// post-stream populations; returns the post-collision populations and the
// clamped pre-collision (rho,ux,uy) the reference stores in texC.
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

}  // namespace wt
