// canvas.hpp — the page's 2-D canvas on the device.
//
// The reference draws every frame onto a 680 x 360 canvas (pages/airfoil_flow_lbm_aerolab.html:919-927): background, the WebGL field scaled
// into the plot rectangle (drawImage, html:923), the particle layer `pcv` (fading strokes of the tracers, html:780-808), the filled and
// outlined foil (html:815-828), the colour bar (html:830-848) and the labels (html:850-860).  Round 3 composited that with NumPy on the host
// (airfoil-cfd-tool_amd/compose.py) — and the page's loop then ran at 9 frames per second whatever the lattice, 99.7 % of a frame in the
// host's strokes and blends (profiles/r04_a_frame_loop_host_canvas.txt).  Here the same drawing rules run per canvas pixel on the GPU:
//   k_canvas_stroke  — the particle layer, kept in device memory: fade (destination-out 0.055), then the frame's segments in particle order;
//   k_canvas_compose — one thread per canvas pixel: field colour (the arithmetic of k_render at the four texels around the sample point,
//                      quantised like the GL framebuffer, then the drawImage bilinear blend), particle layer, foil fill and outline, bar, text.
// Arithmetic in doubles, operation for operation the NumPy compositor's (which stays as the host path and as this one's test reference).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.hpp"

namespace wt {

struct CanvasDims { int s, w, h, px, py, pw, ph; };      // scale; canvas size; plot rectangle (html:69-72)
static inline CanvasDims canvas_dims(int s)
{
    CanvasDims d;
    d.s = s; d.w = 680 * s; d.h = 360 * s; d.px = 54 * s; d.py = 26 * s; d.pw = (680 - 54 - 42) * s; d.ph = (360 - 2 * 26) * s;
    return d;
}

// Particle layer: per pixel {r, g, b (premultiplied), a}.  One 8 x 8 pixel tile per wave; every pixel walks the frame's segments IN ORDER
// (strokes blend over each other), a segment whose bounding box misses the tile is skipped by a wave-uniform test.  A segment is splat the way
// compose.TrailLayer.stroke does it: `ns` points along it, a round brush of radius 1.1 s / 2 + 0.5, coverage clamp(r - distance to the nearest
// point, 0, 1) x 0.75.   seg[i] = {x0, y0, x1, y1 (canvas pixels), ns, r, g, b}.
__global__ __launch_bounds__(64) void k_canvas_stroke(double4 *__restrict__ layer, CanvasDims d, int fade, int n, const double *__restrict__ seg)
{
    const int tx0 = blockIdx.x * 8, ty0 = blockIdx.y * 8;
    const int x = tx0 + (threadIdx.x & 7), y = ty0 + (threadIdx.x >> 3);
    const bool inside = x < d.w && y < d.h;
    double4 v = inside ? layer[(long)y * d.w + x] : double4{0, 0, 0, 0};
    if (fade == 2) v = double4{0, 0, 0, 0};
    else if (fade == 1) { const double k = 1.0 - 0.055; v.x *= k; v.y *= k; v.z *= k; v.w *= k; }
    const double r = 1.1 * d.s / 2.0 + 0.5;
    for (int i = 0; i < n; i++) {
        const double *q = seg + (long)i * 8;
        const double x0 = q[0], y0 = q[1], dx = q[2] - x0, dy = q[3] - y0;
        const int ns = (int)q[4];
        const double xe = x0 + dx, ye = y0 + dy;                     // the last sample point (linspace ends at exactly 1.0)
        int xa = (int)floor(fmin(x0, xe) - r), xb = (int)ceil(fmax(x0, xe) + r);
        int ya = (int)floor(fmin(y0, ye) - r), yb = (int)ceil(fmax(y0, ye) + r);
        xa = xa < 0 ? 0 : xa; ya = ya < 0 ? 0 : ya; xb = xb > d.w - 1 ? d.w - 1 : xb; yb = yb > d.h - 1 ? d.h - 1 : yb;
        if (xb < tx0 || xa > tx0 + 7 || yb < ty0 || ya > ty0 + 7) continue;      // wave-uniform: the tile is clear of this segment
        if (x < xa || x > xb || y < ya || y > yb) continue;
        const double step = 1.0 / (double)(ns - 1);
        double dmin = 1e300;
        for (int k = 0; k < ns; k++) {
            const double t = (k == ns - 1) ? 1.0 : (double)k * step;
            const double sx = x0 + t * dx, sy = y0 + t * dy;
            const double dd = hypot((double)x + 0.5 - sx, (double)y + 0.5 - sy);
            dmin = dd < dmin ? dd : dmin;
        }
        double a = r - dmin;
        a = a < 0.0 ? 0.0 : (a > 1.0 ? 1.0 : a);
        a *= 0.75;
        v.x = v.x * (1.0 - a) + q[5] * a;
        v.y = v.y * (1.0 - a) + q[6] * a;
        v.z = v.z * (1.0 - a) + q[7] * a;
        v.w = v.w * (1.0 - a) + a;
    }
    if (inside) layer[(long)y * d.w + x] = v;
}

// the RGBA8 texel k_render writes for lattice site (i, j), as doubles 0 .. 255
template <typename T>
__device__ __forceinline__ void canvas_texel(const T *__restrict__ macro, const uint8_t *__restrict__ m, const Geom &g, int i, int j,
                                             const FieldParams<T> &fp, double (&c)[3])
{
    T rgb[3];
    if (m[(long)i * g.pitch + j]) { rgb[0] = T(0.039); rgb[1] = T(0.043); rgb[2] = T(0.078); }
    else colour_of<T>(fp.mode, field_value<T>(macro, g, i, j, fp), rgb);
#pragma unroll
    for (int k = 0; k < 3; k++) c[k] = (double)to_unorm8((double)rgb[k]);
}

struct CanvasArgs {
    CanvasDims d;
    const double4 *layer;        // particle layer or null
    const double *poly;          // [npoly][2] canvas coordinates of the foil outline (closed by its first point)
    int npoly;
    const uint8_t *bar;          // [ph][3] colours of the bar's rows
    const float *text;           // [h][w] alpha of the (white) text, 0 where there is none
    double foil_r;               // 1.4 s / 2 + 0.5: half the outline's width + the anti-aliasing margin
    double pbx0, pby0, pbx1, pby1;   // bounding box of the polygon
};

template <typename T>
__global__ __launch_bounds__(256) void k_canvas_compose(const T *__restrict__ macro, const uint8_t *__restrict__ mask, Geom g, FieldParams<T> fp,
                                                        CanvasArgs a, uchar4 *__restrict__ out)
{
    const CanvasDims &d = a.d;
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= d.w || y >= d.h) return;
    double c[3] = {(double)0x0a, (double)0x0d, (double)0x18};                        // '#0a0d18', html:919
    // ---- drawImage(glcv, PX, PY, PW, PH), html:923: bilinear resampling of the lattice image (top row first) into the plot rectangle
    if (x >= d.px && x < d.px + d.pw && y >= d.py && y < d.py + d.ph) {
        const uint8_t *m = mask + g.pitch;
        const int nx = g.nxl, ny = g.ny;
        const double fx = ((double)(x - d.px) + 0.5) / (double)d.pw * (double)nx - 0.5;
        const double fy = ((double)(y - d.py) + 0.5) / (double)d.ph * (double)ny - 0.5;
        int x0 = (int)floor(fx), y0 = (int)floor(fy);
        x0 = x0 < 0 ? 0 : (x0 > nx - 1 ? nx - 1 : x0); y0 = y0 < 0 ? 0 : (y0 > ny - 1 ? ny - 1 : y0);
        const int x1 = x0 + 1 > nx - 1 ? nx - 1 : x0 + 1, y1 = y0 + 1 > ny - 1 ? ny - 1 : y0 + 1;
        double tx = fx - (double)x0, ty = fy - (double)y0;
        tx = tx < 0.0 ? 0.0 : (tx > 1.0 ? 1.0 : tx); ty = ty < 0.0 ? 0.0 : (ty > 1.0 ? 1.0 : ty);
        double c00[3], c01[3], c10[3], c11[3];
        canvas_texel<T>(macro, m, g, x0, ny - 1 - y0, fp, c00);
        canvas_texel<T>(macro, m, g, x1, ny - 1 - y0, fp, c01);
        canvas_texel<T>(macro, m, g, x0, ny - 1 - y1, fp, c10);
        canvas_texel<T>(macro, m, g, x1, ny - 1 - y1, fp, c11);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const double top = c00[k] * (1.0 - tx) + c01[k] * tx, bot = c10[k] * (1.0 - tx) + c11[k] * tx;
            c[k] = top * (1.0 - ty) + bot * ty;
        }
    }
    // ---- drawImage(pcv, 0, 0), html:924
    if (a.layer) {
        const double4 v = a.layer[(long)y * d.w + x];
        c[0] = c[0] * (1.0 - v.w) + v.x; c[1] = c[1] * (1.0 - v.w) + v.y; c[2] = c[2] * (1.0 - v.w) + v.z;
    }
    // ---- drawFoil, html:815-828: even-odd fill at the pixel centre, then the outline (one path: the segments' coverages max-combine)
    const double pxc = (double)x + 0.5, pyc = (double)y + 0.5;
    if (a.npoly >= 3 && pxc >= a.pbx0 - a.foil_r - 1.0 && pxc <= a.pbx1 + a.foil_r + 1.0 && pyc >= a.pby0 - a.foil_r - 1.0 && pyc <= a.pby1 + a.foil_r + 1.0) {
        bool in = false;
        double cov = 0.0;
        const int yrow_lo = (int)floor(a.pby0), yrow_hi = (int)ceil(a.pby1);
        for (int i = 0; i < a.npoly; i++) {
            const int j = i + 1 == a.npoly ? 0 : i + 1;
            const double ax = a.poly[2 * i], ay = a.poly[2 * i + 1], bx = a.poly[2 * j], by = a.poly[2 * j + 1];
            if (((ay <= pyc) && (by > pyc)) || ((by <= pyc) && (ay > pyc))) {
                const double xc = ax + (pyc - ay) / (by - ay) * (bx - ax);
                if (xc <= pxc) in = !in;
            }
            const double ex = bx - ax, ey = by - ay, L2 = ex * ex + ey * ey;
            double t = 0.0;
            if (L2 > 0.0) { t = ((pxc - ax) * ex + (pyc - ay) * ey) / L2; t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t); }
            const double dd = hypot(pxc - (ax + t * ex), pyc - (ay + t * ey));
            double cv = a.foil_r - dd;
            cv = cv < 0.0 ? 0.0 : (cv > 1.0 ? 1.0 : cv);
            cov = cv > cov ? cv : cov;
        }
        if (in && y >= yrow_lo && y <= yrow_hi) { c[0] = (double)0x0d; c[1] = (double)0x10; c[2] = (double)0x18; }      // '#0d1018', html:824
        const double al = cov * 0.85;                                                                                     // rgba(200,215,255,0.85), html:826
        c[0] = c[0] * (1.0 - al) + 200.0 * al; c[1] = c[1] * (1.0 - al) + 215.0 * al; c[2] = c[2] * (1.0 - al) + 255.0 * al;
    }
    // ---- drawBar, html:830-848
    const int bx0 = d.w - 32 * d.s, bw = 10 * d.s;
    if (x >= bx0 && x < bx0 + bw) {
        if (y >= d.py && y < d.py + d.ph) {
            const uint8_t *row = a.bar + 3 * (y - d.py);
            c[0] = (double)row[0]; c[1] = (double)row[1]; c[2] = (double)row[2];
        } else if (y == d.py + d.ph && y < d.h) {      // the last rect's lower half pixel
            const uint8_t *row = a.bar + 3 * (d.ph - 1);
            c[0] = c[0] * 0.5 + (double)row[0] * 0.5; c[1] = c[1] * 0.5 + (double)row[1] * 0.5; c[2] = c[2] * 0.5 + (double)row[2] * 0.5;
        }
    }
    // ---- drawLabels, html:850-860 (white, the strings' alphas from the host's 5 x 7 glyphs)
    if (a.text) {
        const double al = (double)a.text[(long)y * d.w + x];
        if (al > 0.0) { c[0] = c[0] * (1.0 - al) + 255.0 * al; c[1] = c[1] * (1.0 - al) + 255.0 * al; c[2] = c[2] * (1.0 - al) + 255.0 * al; }
    }
    uchar4 o;
    double q;
    q = rint(c[0]); q = q < 0.0 ? 0.0 : (q > 255.0 ? 255.0 : q); o.x = (unsigned char)q;
    q = rint(c[1]); q = q < 0.0 ? 0.0 : (q > 255.0 ? 255.0 : q); o.y = (unsigned char)q;
    q = rint(c[2]); q = q < 0.0 ? 0.0 : (q > 255.0 ? 255.0 : q); o.z = (unsigned char)q;
    o.w = 255;
    out[(long)y * d.w + x] = o;
}

}  // namespace wt
