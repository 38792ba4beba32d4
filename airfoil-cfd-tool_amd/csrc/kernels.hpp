// kernels.hpp — HIP kernels of libwindtunnel.so (gfx950 / MI355X).
//
// DEVICE LAYOUT ("column-major SoA").  The lattice is stored with y (iy, here
// `j`) as the FAST axis and x (ix, here `i`, the streamwise direction) as the
// slow axis:  f[k][i][j], one plane per population k, `pitch` elements per
// column (pitch >= NY, a multiple of 256), one pad column before column 0 and
// one after the last column of every plane.  Why: the tunnel is sharded over
// GPUs as COLUMN slabs, so with y fastest a ghost column is one contiguous run
// of NY elements per population — RCCL send/recv operate straight on the
// lattice (no pack/unpack kernels) and the slab-edge strips are ordinary
// coalesced launches of the same kernel.  Inlet and outlet columns become whole
// memory rows (wave-uniform branches).  Host-side arrays keep the reference's
// [NY][NX] layout; transposition happens in the read-back/upload kernels below.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "d2q9.hpp"

namespace wt {

struct Geom {
    int nxl;        // local columns (ghosts included)
    int ny;         // rows
    int gi0;        // global column index of local column 0
    int nx_g;       // global columns
    long pitch;     // elements per column
    long plane;     // elements per population plane = (nxl+2)*pitch
};

// tile classes (one per wave-tile of TILE_J consecutive j in one column), built by k_classify
enum : uint8_t { TILE_GENERAL = 0, TILE_FAST = 1, TILE_SOLID = 2, TILE_INLET = 3, TILE_OUTLET = 4 };

// --------------------------------------------------------------------------------------
// site_general: one lattice site, every branch of STEP_FS main() (html:283-360) taken per
// lane.  Correct for any site; used by the tiles that touch the body surface and by ragged
// tiles.  (A 4-sites-per-lane vector form of this path was measured SLOWER on the 4096^2
// body case, 213 vs 207 us per step: its 111 VGPRs cost the whole kernel two waves per SIMD.)
// --------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void site_general(const T *__restrict__ s, T *__restrict__ d, T *__restrict__ macro,
                                             const uint8_t *__restrict__ m, const Geom &g, int i, int j,
                                             T tau, T U0, bool emit)
{
    const long c = (long)i * g.pitch + j;
    const int gi = i + g.gi0;
    T out[9], rho, ux, uy;
    if (m[c]) {                                                    // html:287-294 solid
#pragma unroll
        for (int k = 0; k < 9; k++) out[k] = s[opp_of(k) * g.plane + c];
        rho = T(1.0); ux = T(0.0); uy = T(0.0);
    } else if (gi == g.nx_g - 1) {                                 // html:301-312 outlet
        T q[9];
#pragma unroll
        for (int k = 0; k < 9; k++) q[k] = s[k * g.plane + c - g.pitch];
        rho = q[0] + q[1] + q[2] + q[3] + q[4] + q[5] + q[6] + q[7] + q[8];
        ux = (q[1] + q[5] + q[8] - q[3] - q[6] - q[7]) / rho;
        uy = (q[2] + q[5] + q[6] - q[4] - q[7] - q[8]) / rho;
#pragma unroll
        for (int k = 0; k < 9; k++) out[k] = q[k];
    } else if (gi == 0 || j == g.ny - 1 || j == 0) {               // html:314-322 far field
        rho = T(1.0); ux = U0; uy = T(0.0);
        feq_all(rho, ux, uy, out);
    } else {                                                       // html:324-359 interior fluid
        T fin[9];
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const long src = c - (long)ex_of(k) * g.pitch - ey_of(k);
            fin[k] = m[src] ? s[opp_of(k) * g.plane + c] : s[k * g.plane + src];
        }
        collide(fin, tau, out, rho, ux, uy);
    }
#pragma unroll
    for (int k = 0; k < 9; k++) d[k * g.plane + c] = out[k];
    if (emit) {
        const long mp = (long)g.nxl * g.pitch;
        macro[c] = rho; macro[mp + c] = ux; macro[2 * mp + c] = uy;
    }
}

// --------------------------------------------------------------------------------------
// init: equilibriumInitData (html:474-490).  v[k] are evaluated on the host in double.
// --------------------------------------------------------------------------------------
template <typename T>
struct Init9 { T v[9]; T u0; };

template <typename T>
__global__ void k_fill_equilibrium(T *__restrict__ f0, T *__restrict__ f1, T *__restrict__ macro,
                                   Geom g, Init9<T> iv)
{
    const long n = g.plane;
    const long mp = (long)g.nxl * g.pitch;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
#pragma unroll
        for (int k = 0; k < 9; k++) { f0[k * n + t] = iv.v[k]; f1[k * n + t] = iv.v[k]; }
        if (t < mp) { macro[t] = T(1.0); macro[mp + t] = iv.u0; macro[2 * mp + t] = T(0.0); }
    }
}

// --------------------------------------------------------------------------------------
// layout converters between the host's [NY][W] rows and the device's [i][j] columns
// --------------------------------------------------------------------------------------
// dst[j*W + x] = src[(i0+x)*pitch + j]   (device columns -> host rows), x in [0,W)
template <typename T>
__global__ void k_cols_to_rows(const T *__restrict__ src, T *__restrict__ dst, int i0, int W, int ny, long pitch)
{
    __shared__ T tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;   // bx: j block, by: x block
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int x = by + r, j = bx + threadIdx.x;
        if (x < W && j < ny) tile[r][threadIdx.x] = src[(long)(i0 + x) * pitch + j];
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int j = bx + r, x = by + threadIdx.x;
        if (x < W && j < ny) dst[(long)j * W + x] = tile[threadIdx.x][r];
    }
}

// dst[(i0+x)*pitch + j] = src[j*ld + x]  (host rows -> device columns), x in [0,W)
template <typename T>
__global__ void k_rows_to_cols(const T *__restrict__ src, T *__restrict__ dst, int i0, int W, int ny, long pitch, long ld)
{
    __shared__ T tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;   // bx: x block, by: j block
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int j = by + r, x = bx + threadIdx.x;
        if (x < W && j < ny) tile[r][threadIdx.x] = src[(long)j * ld + x];
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int x = bx + r, j = by + threadIdx.x;
        if (x < W && j < ny) dst[(long)(i0 + x) * pitch + j] = tile[threadIdx.x][r];
    }
}

// --------------------------------------------------------------------------------------
// reductions (on demand, never inside the step loop)
// --------------------------------------------------------------------------------------
struct RangePartial { double max_s, cp_min, cp_max; };
struct ForcePartial { double fx, fy; long long surf, rev; };

__device__ __forceinline__ double wave_max(double v) { for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ double wave_min(double v) { for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ double wave_sum(double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ __forceinline__ long long wave_sum(long long v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }

// updateFieldsFromMacro's scan (html:596-614), doubles, fluid sites of owned columns
template <typename T>
__global__ __launch_bounds__(256) void k_ranges(const T *__restrict__ macro, const uint8_t *__restrict__ mask, Geom g,
                                                int i_own0, int W, double u0, RangePartial *__restrict__ part)
{
    const uint8_t *m = mask + g.pitch;
    const long mp = (long)g.nxl * g.pitch;
    double mx = 0.0, cmin = __builtin_inf(), cmax = -__builtin_inf();
    const double cpden = 1.5 * u0 * u0;
    const long total = (long)W * g.ny;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int x = (int)(t / g.ny), j = (int)(t % g.ny);
        const long c = (long)(i_own0 + x) * g.pitch + j;
        if (m[c]) continue;
        const double rho = (double)macro[c], ux = (double)macro[mp + c], uy = (double)macro[2 * mp + c];
        const double u = ux / u0, v = uy / u0;
        const double cp = (rho - 1.0) / cpden;
        const double s = hypot(u, v);
        if (s > mx && s < 4.0) mx = s;
        if (cp > -4.0 && cp < 1.2) { if (cp < cmin) cmin = cp; if (cp > cmax) cmax = cp; }
    }
    __shared__ double sh[3][4];
    mx = wave_max(mx); cmin = wave_min(cmin); cmax = wave_max(cmax);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][w] = mx; sh[1][w] = cmin; sh[2][w] = cmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        RangePartial r;
        r.max_s = fmax(fmax(sh[0][0], sh[0][1]), fmax(sh[0][2], sh[0][3]));
        r.cp_min = fmin(fmin(sh[1][0], sh[1][1]), fmin(sh[1][2], sh[1][3]));
        r.cp_max = fmax(fmax(sh[2][0], sh[2][1]), fmax(sh[2][2], sh[2][3]));
        part[blockIdx.x] = r;
    }
}

// computeForces (html:650-698) seen from the fluid side: every owned fluid cell adds, for each
// of its 4 face neighbours that is inside the grid and solid, p = rho/3 along the unit vector
// from the fluid cell into the solid.
template <typename T>
__global__ __launch_bounds__(256) void k_forces(const T *__restrict__ macro, const uint8_t *__restrict__ mask, Geom g,
                                                int i_own0, int W, ForcePartial *__restrict__ part)
{
    const uint8_t *m = mask + g.pitch;
    const long mp = (long)g.nxl * g.pitch;
    double fx = 0.0, fy = 0.0;
    long long surf = 0, rev = 0;
    const long total = (long)W * g.ny;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int x = (int)(t / g.ny), j = (int)(t % g.ny);
        const int i = i_own0 + x;
        const int gi = i + g.gi0;
        const long c = (long)i * g.pitch + j;
        if (m[c]) continue;
        // solid neighbour at (gi+dx, j+dy) inside the grid
        const int sxp = (gi + 1 < g.nx_g) && m[c + g.pitch];
        const int sxm = (gi - 1 >= 0) && m[c - g.pitch];
        const int syp = (j + 1 < g.ny) && m[c + 1];
        const int sym = (j - 1 >= 0) && m[c - 1];
        const int nf = sxp + sxm + syp + sym;
        if (nf == 0) continue;
        const double p = (double)macro[c] / 3.0;
        fx += p * (double)(sxp - sxm);
        fy += p * (double)(syp - sym);
        surf += nf;
        if (macro[mp + c] < T(0.0)) rev += nf;
    }
    __shared__ double shd[2][4];
    __shared__ long long shl[2][4];
    fx = wave_sum(fx); fy = wave_sum(fy); surf = wave_sum(surf); rev = wave_sum(rev);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { shd[0][w] = fx; shd[1][w] = fy; shl[0][w] = surf; shl[1][w] = rev; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ForcePartial r;
        r.fx = shd[0][0] + shd[0][1] + shd[0][2] + shd[0][3];
        r.fy = shd[1][0] + shd[1][1] + shd[1][2] + shd[1][3];
        r.surf = shl[0][0] + shl[0][1] + shl[0][2] + shl[0][3];
        r.rev = shl[1][0] + shl[1][1] + shl[1][2] + shl[1][3];
        part[blockIdx.x] = r;
    }
}

// Diagnostics for the stability net (html:344-350): how many fluid sites of the owned columns sat AT a clamp in the
// last emitted state — stored rho equal to a density bound, stored |u| at the speed bound (the stored velocity of a
// clamped site is u * (0.35 / |u|), i.e. 0.35 up to rounding).  On demand, never in the step loop.
struct ClampPartial { long long rho_events, u_events; };

template <typename T>
__global__ __launch_bounds__(256) void k_clamp_events(const T *__restrict__ macro, const uint8_t *__restrict__ mask, Geom g,
                                                      int i_own0, int W, ClampPartial *__restrict__ part)
{
    const uint8_t *m = mask + g.pitch;
    const long mp = (long)g.nxl * g.pitch;
    long long nr = 0, nu = 0;
    const long total = (long)W * g.ny;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int x = (int)(t / g.ny), j = (int)(t % g.ny);
        const long c = (long)(i_own0 + x) * g.pitch + j;
        if (m[c]) continue;
        const T rho = macro[c], ux = macro[mp + c], uy = macro[2 * mp + c];
        nr += (rho == T(0.5) || rho == T(2.0));
        nu += ((double)ux * (double)ux + (double)uy * (double)uy >= 0.35 * 0.35 * (1.0 - 1e-6));
    }
    __shared__ long long sh[2][4];
    nr = wave_sum(nr); nu = wave_sum(nu);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][w] = nr; sh[1][w] = nu; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ClampPartial r;
        r.rho_events = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
        r.u_events = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
        part[blockIdx.x] = r;
    }
}

// RENDER_FS main() field math (html:395-420): scalar t per owned site, written in HOST layout
// out[j*W + x]; NaN on solids.  Neighbour columns beyond the slab come from the ghost columns
// (kept fresh by the caller), beyond the tunnel from CLAMP_TO_EDGE.
template <typename T>
struct FieldParams { T U0, maxS, cpMin, cpMax, vortScale; int mode; };

template <typename T>
__device__ __forceinline__ T field_value(const T *__restrict__ macro, const Geom &g, int i, int j, const FieldParams<T> &fp)
{
    const long mp = (long)g.nxl * g.pitch;
    const long c = (long)i * g.pitch + j;
    if (fp.mode == 0) {
        const T ux = macro[mp + c], uy = macro[2 * mp + c];
        const T s = wt_sqrt<T>(ux * ux + uy * uy) / fp.U0;
        const T den = fp.maxS * T(0.92);
        return s / (den < T(1e-6) ? T(1e-6) : den);
    } else if (fp.mode == 1) {
        const T cp = (macro[c] - T(1.0)) / (T(1.5) * fp.U0 * fp.U0);
        const T r = fp.cpMax - fp.cpMin;
        return (cp - fp.cpMin) / (r < T(1e-6) ? T(1e-6) : r);
    }
    const int gi = i + g.gi0;
    const long cR = (gi + 1 < g.nx_g) ? c + g.pitch : c;
    const long cL = (gi - 1 >= 0) ? c - g.pitch : c;
    const long cU = (j + 1 < g.ny) ? c + 1 : c;
    const long cD = (j - 1 >= 0) ? c - 1 : c;
    const T dvydx = (macro[2 * mp + cR] - macro[2 * mp + cL]) * T(0.5);
    const T duxdy = (macro[mp + cU] - macro[mp + cD]) * T(0.5);
    const T vort = dvydx - duxdy;
    const T den = fp.U0 * fp.vortScale;
    return vort / (den < T(1e-6) ? T(1e-6) : den);
}

template <typename T>
__global__ void k_field(const T *__restrict__ macro, const uint8_t *__restrict__ mask, Geom g,
                        int i_own0, int W, FieldParams<T> fp, T *__restrict__ out)
{
    // 32x32 tiles: read along j (coalesced in device layout), write along x (coalesced in host layout)
    __shared__ T tile[32][33];
    const uint8_t *m = mask + g.pitch;
    const int bj = blockIdx.x * 32, bx = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int x = bx + r, j = bj + threadIdx.x;
        if (x < W && j < g.ny) {
            const int i = i_own0 + x;
            const long c = (long)i * g.pitch + j;
            tile[r][threadIdx.x] = m[c] ? T(__builtin_nanf("")) : field_value<T>(macro, g, i, j, fp);
        }
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int j = bj + r, x = bx + threadIdx.x;
        if (x < W && j < g.ny) out[(long)j * W + x] = tile[threadIdx.x][r];
    }
}


// --------------------------------------------------------------------------------------
// RENDER_FS colour maps (html:371-393) and solid colour (html:397), evaluated in T like the
// shader (mix(a,b,u) = a*(1-u)+b*u, stops divided by 255.0), then quantised to RGBA8 the way a
// GL framebuffer does (round to nearest).
// --------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void lerp_stops(T t, const unsigned char (*stops)[3], int nseg, T (&rgb)[3])
{
    t = t < T(0.0) ? T(0.0) : t;            // clamp(t,0,1) = min(max(t,0),1)
    t = T(1.0) < t ? T(1.0) : t;
    const T f = t * T(nseg);
    int i = (int)floor((double)f);
    if (i > nseg - 1) i = nseg - 1;
    if (i < 0) i = 0;
    const T u = f - T(i);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const T a = T(stops[i][c]) / T(255.0), b = T(stops[i + 1][c]) / T(255.0);
        rgb[c] = a * (T(1.0) - u) + b * u;
    }
}

__device__ const unsigned char kSpeedStops[10][3] = {{5, 5, 20}, {0, 20, 120}, {0, 60, 200}, {0, 140, 220}, {0, 220, 220},
                                                     {0, 210, 140}, {80, 200, 0}, {220, 210, 0}, {255, 120, 0}, {220, 20, 0}};
__device__ const unsigned char kCpStops[8][3] = {{20, 50, 160}, {40, 110, 210}, {100, 175, 235}, {190, 220, 245},
                                                 {248, 248, 248}, {248, 214, 140}, {240, 150, 60}, {205, 50, 25}};

template <typename T>
__device__ __forceinline__ void colour_of(int mode, T t, T (&rgb)[3])
{
    if (mode == 0) { lerp_stops<T>(t, kSpeedStops, 9, rgb); return; }
    if (mode == 1) { lerp_stops<T>(t, kCpStops, 7, rgb); return; }
    t = t < T(-1.0) ? T(-1.0) : t;          // vortColor, html:389-393
    t = T(1.0) < t ? T(1.0) : t;
    const T base[3] = {T(0.06), T(0.07), T(0.11)};
    const T neg[3] = {T(0.15), T(0.5), T(0.98)}, pos[3] = {T(0.98), T(0.28), T(0.18)};
    const bool isneg = t < T(0.0);
    const T a = isneg ? -t : t;
#pragma unroll
    for (int c = 0; c < 3; c++) rgb[c] = base[c] * (T(1.0) - a) + (isneg ? neg[c] : pos[c]) * a;
}

__device__ __forceinline__ unsigned char to_unorm8(double c)
{
    c = c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c);
    return (unsigned char)(int)(c * 255.0 + 0.5);
}

template <typename T>
__global__ void k_render(const T *__restrict__ macro, const uint8_t *__restrict__ mask, Geom g,
                         int i_own0, int W, FieldParams<T> fp, uchar4 *__restrict__ out)
{
    __shared__ uchar4 tile[32][33];
    const uint8_t *m = mask + g.pitch;
    const int bj = blockIdx.x * 32, bx = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int x = bx + r, j = bj + threadIdx.x;
        if (x < W && j < g.ny) {
            const int i = i_own0 + x;
            const long c = (long)i * g.pitch + j;
            T rgb[3];
            if (m[c]) { rgb[0] = T(0.039); rgb[1] = T(0.043); rgb[2] = T(0.078); }
            else colour_of<T>(fp.mode, field_value<T>(macro, g, i, j, fp), rgb);
            tile[r][threadIdx.x] = make_uchar4(to_unorm8((double)rgb[0]), to_unorm8((double)rgb[1]), to_unorm8((double)rgb[2]), 255);
        }
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int j = bj + r, x = bx + threadIdx.x;
        if (x < W && j < g.ny) out[(long)j * W + x] = tile[threadIdx.x][r];
    }
}


// --------------------------------------------------------------------------------------
// tracer particles: sampleScalar / sampleUV (html:616-639) and advect (html:758-771).
// JS doubles on top of Ufield/Vfield = Float32Array(ux/U0, uy/U0) (html:603-604), solids excluded.
// --------------------------------------------------------------------------------------
struct Window { double dx0, dx1, dy0, dy1; };

template <typename T>
__device__ __forceinline__ bool sample_uv(const T *__restrict__ macro, const uint8_t *__restrict__ m, const Geom &g, int i_own0,
                                          double U0, const Window &w, double wx, double wy, double &u, double &v)
{
    if (wx < w.dx0 || wx > w.dx1 || wy < w.dy0 || wy > w.dy1) return false;
    const int NX = g.nx_g, NY = g.ny;
    const double fx = (wx - w.dx0) / (w.dx1 - w.dx0) * NX - 0.5;
    const double fy = (wy - w.dy0) / (w.dy1 - w.dy0) * NY - 0.5;
    int ix = (int)floor(fx), iy = (int)floor(fy);
    ix = ix > NX - 2 ? NX - 2 : ix; ix = ix < 0 ? 0 : ix;
    iy = iy > NY - 2 ? NY - 2 : iy; iy = iy < 0 ? 0 : iy;
    const double tx = fx - ix, ty = fy - iy;
    const double ws[4] = {(1 - tx) * (1 - ty), tx * (1 - ty), (1 - tx) * ty, tx * ty};
    const int cx[4] = {ix, ix + 1, ix, ix + 1}, cy[4] = {iy, iy, iy + 1, iy + 1};
    const long mp = (long)g.nxl * g.pitch;
    double su = 0, wu = 0, sv = 0, wv = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const long c = (long)(i_own0 + cx[k]) * g.pitch + cy[k];
        if (m[c]) continue;
        const double U = (double)(float)((double)macro[mp + c] / U0);
        const double V = (double)(float)((double)macro[2 * mp + c] / U0);
        if (isfinite(U)) { su += U * ws[k]; wu += ws[k]; }
        if (isfinite(V)) { sv += V * ws[k]; wv += ws[k]; }
    }
    if (!(wu > 0) || !(wv > 0)) return false;
    u = su / wu; v = sv / wv;
    return true;
}

template <typename T>
__global__ void k_advect(const T *__restrict__ macro, const uint8_t *__restrict__ mask, Geom g, int i_own0, double U0, Window w,
                         double dt_frame, int n, const double *__restrict__ px, const double *__restrict__ py,
                         double *__restrict__ ox, double *__restrict__ oy, double *__restrict__ ospeed, unsigned char *__restrict__ ok)
{
    const uint8_t *m = mask + g.pitch;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        const double x = px[t], y = py[t];
        double u1, v1;
        if (!sample_uv<T>(macro, m, g, i_own0, U0, w, x, y, u1, v1)) { ok[t] = 0; ox[t] = x; oy[t] = y; ospeed[t] = 0.0; continue; }
        const double kBase = 0.00105 * dt_frame;
        const double speed1 = hypot(u1, v1);
        double dtEff = kBase;
        const double maxDisp = 0.05;
        if (speed1 * dtEff > maxDisp) dtEff = maxDisp / fmax(speed1, 1e-6);
        const double midx = x + u1 * dtEff * 0.5, midy = y + v1 * dtEff * 0.5;
        double u2, v2;
        if (!sample_uv<T>(macro, m, g, i_own0, U0, w, midx, midy, u2, v2)) { u2 = u1; v2 = v1; }
        ox[t] = x + u2 * dtEff; oy[t] = y + v2 * dtEff; ospeed[t] = hypot(u2, v2); ok[t] = 1;
    }
}

}  // namespace wt
