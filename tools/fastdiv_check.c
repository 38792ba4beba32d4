// tools/fastdiv_check.c — CPU evidence for the two division-by-tau sequences of csrc/d2q9.hpp (developer experiment; the library itself PROVES the
// fp32 form per tau on the device, k_verify_fastdiv, and the fp64 form rests on Markstein's theorem — this program is the empirical cross-check).
//
//   fp32, two operations:   p = RN(x * rlo);  q = fma(x, rhi, p)                        rhi = RN(1/tau), rlo = RN(1/tau - rhi)
//   fp64 / fp32, four ops:  p = RN(x * rlo);  q1 = fma(x, rhi, p);  e = fma(-q1, tau, x);  q = fma(e, rhi, q1)
//
// Mode "f32 N seed": N random binary32 tau in [0.5, 2): both forms against x / tau for ALL 2^23 significands x in [1, 2) (the sequences commute with
// scaling by powers of two and are odd in x).  Mode "f64 N seed": N random (x, tau) pairs per thread in binary64, the four-operation form against x / tau,
// plus pairs constructed to sit next to rounding boundaries (x = RN(m * tau) for midpoints m).
//   gcc -O2 -march=native -fopenmp -ffp-contract=off tools/fastdiv_check.c -lm -o /tmp/fastdiv_check
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static uint64_t rng_next(uint64_t *s) { uint64_t z = (*s += 0x9e3779b97f4a7c15ULL); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); }

int main(int argc, char **argv)
{
    const char *mode = argc > 1 ? argv[1] : "f32";
    const long n = argc > 2 ? atol(argv[2]) : 200;
    uint64_t seed = argc > 3 ? strtoull(argv[3], 0, 10) : 1;
    if (!strcmp(mode, "f32")) {
        long fail2 = 0, fail4 = 0, fail3 = 0;
        for (long t = 0; t < n; t++) {
            uint32_t tb = 0x3f000000u + (uint32_t)(rng_next(&seed) & 0xffffffu);      // [0.5, 2)
            float tau; memcpy(&tau, &tb, 4);
            if (t == 0) tau = 0.58f;
            if (t == 1) tau = (float)(0.5 + 3.0 * 0.06 * (4096.0 / 1.84) / 1e6);
            const float rhi = 1.0f / tau;
            const float rlo = (float)(1.0 / (double)tau - (double)rhi);
            long b2 = 0, b3 = 0, b4 = 0;
#pragma omp parallel for reduction(+ : b2, b3, b4)
            for (uint32_t m = 0; m < (1u << 23); m++) {
                const uint32_t xb = 0x3f800000u | m;
                float x; memcpy(&x, &xb, 4);
                const float ref = x / tau;
                const float p = x * rlo;
                const float q2 = fmaf(x, rhi, p);
                const float q0 = x * rhi;
                const float q3 = fmaf(fmaf(-q0, tau, x), rhi, q0);
                const float q4 = fmaf(fmaf(-q2, tau, x), rhi, q2);
                b2 += q2 != ref; b3 += q3 != ref; b4 += q4 != ref;
            }
            if (b2 || b3 || b4) printf("tau %.9g (0x%08x): two-op %ld  three-op %ld  four-op %ld mismatches\n", tau, tb, b2, b3, b4);
            fail2 += b2 != 0; fail3 += b3 != 0; fail4 += b4 != 0;
        }
        printf("f32: %ld tau, all 2^23 significands each: tau with a mismatch: two-op %ld, three-op (round 1-4's form) %ld, four-op %ld\n", n, fail2, fail3, fail4);
        return 0;
    }
    long bad = 0, total = 0;
#pragma omp parallel reduction(+ : bad, total)
    {
        uint64_t s = seed * 1000003ULL + 77ULL * (uint64_t)
#ifdef _OPENMP
            omp_get_thread_num();
#else
            0;
#endif
        for (long i = 0; i < n; i++) {
            uint64_t tb = 0x3fe0000000000000ULL + (rng_next(&s) & 0x1fffffffffffffULL);        // [0.5, 2)
            double tau; memcpy(&tau, &tb, 8);
            const double rhi = 1.0 / tau, rlo = fma(-rhi, tau, 1.0) / tau;
            for (int k = 0; k < 64; k++) {
                double x;
                if (k < 32) { uint64_t xb = 0x3ff0000000000000ULL | (rng_next(&s) & 0xfffffffffffffULL); memcpy(&x, &xb, 8); }
                else {
                    // next to a rounding boundary: a midpoint m = (2 j + 1) 2^-53 of [1, 2) or [0.5, 1); x = RN(m tau) +- a few ulps
                    const uint64_t j = rng_next(&s) & 0xfffffffffffffULL;
                    const double mlo = 1.0 + (double)j * 0x1p-52;                     // a double; the midpoint above it is mlo + 2^-53
                    const double xm = fma(mlo, tau, 0x1p-53 * tau);
                    x = nextafter(xm, (k & 1) ? 4.0 : 0.0);
                    for (int q = 0; q < (k >> 1) % 3; q++) x = nextafter(x, (k & 1) ? 4.0 : 0.0);
                }
                const double ref = x / tau;
                const double p = x * rlo, q1 = fma(x, rhi, p), e = fma(-q1, tau, x), q = fma(e, rhi, q1);
                total++;
                if (q != ref) { bad++; if (bad < 5) printf("f64 mismatch: x %a tau %a ref %a got %a\n", x, tau, ref, q); }
            }
        }
    }
    printf("f64: %ld (x, tau) pairs (half of them built next to rounding boundaries): %ld mismatches of the four-operation form against x / tau\n", total, bad);
    return bad != 0;
}
