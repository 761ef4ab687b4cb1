// Bit-faithful fp32 sinf / cosf of glibc 2.35 (sysdeps/ieee754/flt-32/s_sincosf.h, s_sinf.c, s_cosf.c,
// s_sincosf_data.c -- the ARM "optimized routines" algorithm), usable from HIP device code and host C/C++.
//
// WHY: the reference's Costas loop evaluates cosf(-phase) / sinf(-phase) per sample
// (/root/reference/src/urh/cythonext/signal_functions.pyx:301; compiled as C++, GCC merges the pair into
// sincosf, which computes the same two polynomials), and its output feeds back into the loop, so a 1-ulp
// difference anywhere changes every later sample.  glibc's routines work in DOUBLE precision (range reduction by
// one multiply-subtract with pi/2, degree-7/8 polynomials) and round once to float; they are not correctly
// rounded.  x86-64 glibc selects at run time (ifunc) between a build without FMA (`__sinf_sse2`) and one compiled
// with -mfma (`__sinf_fma`, chosen on every CPU with FMA+AVX2), in which GCC contracts every a + b*c into an fma:
// URH_SINCOSF_FMA selects that evaluation (default 1: all hosts of interest have FMA).  The two differ in the
// rounded float for about one input in 10^9.
// |x| < 120 takes the fast reduction (the Costas phase is kept within +-2*pi); larger finite arguments -- the carrier
// argument 2*pi*f*t of modulate_c (signal_functions.pyx:159-166) reaches 10^5 rad -- take glibc's reduce_large: a
// 32x96-bit fixed-point multiply with 4/pi (table __inv_pio4: 4/pi in 8-bit steps, recomputed here from pi with integer
// arithmetic and checked against the host libm by tests/test_sincosf_port.py).  inf / NaN return NaN.
// Constants are the published ones of ARM optimized-routines (MIT licence) as shipped in glibc.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define URH_SC_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#define URH_SC_HD static inline
#endif

#ifndef URH_SINCOSF_FMA
#define URH_SINCOSF_FMA 1
#endif

URH_SC_HD double urh_sc_madd(double a, double b, double c) {   // a + b * c as the selected libm build evaluates it
#if URH_SINCOSF_FMA
    return __builtin_fma(b, c, a);
#else
    return a + b * c;
#endif
}

URH_SC_HD uint32_t urh_sc_abstop12(float x) {
    union { float f; uint32_t u; } v; v.f = x;
    return (v.u >> 20) & 0x7ff;
}

// polynomial table entry 0 / 1 (entry 1: cosine polynomial negated)
URH_SC_HD float urh_sinf_poly(double x, double x2, int neg_cos, int n) {
    const double s1c = -0x1.555545995a603p-3, s2c = 0x1.1107605230bc4p-7, s3c = -0x1.994eb3774cf24p-13;
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double s1 = urh_sc_madd(s2c, x2, s3c);
        const double x7 = x3 * x2;
        const double s = urh_sc_madd(x, x3, s1c);
        return (float)urh_sc_madd(s, x7, s1);
    } else {
        const double sg = neg_cos ? -1.0 : 1.0;
        const double c0 = sg * 0x1p0, c1c = sg * -0x1.ffffffd0c621cp-2, c2c = sg * 0x1.55553e1068f19p-5,
                     c3c = sg * -0x1.6c087e89a359dp-10, c4c = sg * 0x1.99343027bf8c3p-16;
        const double x4 = x2 * x2;
        const double c2 = urh_sc_madd(c3c, x2, c4c);
        const double c1 = urh_sc_madd(c0, x2, c1c);
        const double x6 = x4 * x2;
        const double c = urh_sc_madd(c1, x4, c2c);
        return (float)urh_sc_madd(c, x6, c2);
    }
}

// reduce_fast (!TOINT_INTRINSICS, x86-64): quadrant from a 2^24-scaled float->int conversion
URH_SC_HD double urh_sc_reduce_fast(double x, int *np) {
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    const double r = x * hpi_inv;
    const int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
#if URH_SINCOSF_FMA
    return __builtin_fma(-(double)n, hpi, x);
#else
    return x - (double)n * hpi;
#endif
}

URH_SC_HD float urh_sincosf_invalid(void) {
    union { uint32_t u; float f; } v; v.u = 0x7fc00000u; return v.f;
}

// reduce_large: xi = bits of a float with |x| >= 2; returns x mod pi/2 in [-pi/4, pi/4] and the quadrant
URH_SC_HD double urh_sc_reduce_large(uint32_t xi, int *np) {
    const uint32_t inv_pio4[24] = {0xa2, 0xa2f9, 0xa2f983, 0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529,
                                   0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1, 0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0,
                                   0x34ddc0db, 0xddc0db62, 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041};
    const double pi63 = 0x1.921FB54442D18p-62;             // 2 pi * 2^-64
    const uint32_t *arr = &inv_pio4[(xi >> 26) & 15];
    const int shift = (xi >> 23) & 7;
    uint64_t n, res0, res1, res2;
    xi = (xi & 0xffffff) | 0x800000;
    xi <<= shift;
    res0 = (uint32_t)(xi * arr[0]);                        // 32-bit product (the reference multiplies two uint32_t)
    res1 = (uint64_t)xi * arr[4];
    res2 = (uint64_t)xi * arr[8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    n = (res0 + (1ULL << 61)) >> 62;
    res0 -= n << 62;
    *np = (int)n;
    return (double)(int64_t)res0 * pi63;
}

URH_SC_HD float urh_sinf(float y) {
    double x = y;
    if (urh_sc_abstop12(y) < 0x3f4) {                       // |y| < pi/4 (abstop12(0x1.921FB6p-1f))
        const double s = x * x;
        if (urh_sc_abstop12(y) < 0x398) return y;           // |y| < 2^-12
        return urh_sinf_poly(x, s, 0, 0);
    }
    if (urh_sc_abstop12(y) < 0x42f) {                       // |y| < 120
        int n;
        x = urh_sc_reduce_fast(x, &n);
        const double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;     // sign[] = {1, -1, -1, 1}
        return urh_sinf_poly(x * s, x * x, (n & 2) != 0, n);
    }
    if (urh_sc_abstop12(y) < 0x7f8) {                       // finite
        union { float f; uint32_t u; } v; v.f = y;
        const int sign = (int)(v.u >> 31);
        int n;
        x = urh_sc_reduce_large(v.u, &n);
        const int k = (n + sign) & 3;
        const double s = (k == 1 || k == 2) ? -1.0 : 1.0;
        return urh_sinf_poly(x * s, x * x, (k & 2) != 0, n);
    }
    return urh_sincosf_invalid();
}

URH_SC_HD float urh_cosf(float y) {
    double x = y;
    if (urh_sc_abstop12(y) < 0x3f4) {
        const double x2 = x * x;
        if (urh_sc_abstop12(y) < 0x398) return 1.0f;
        return urh_sinf_poly(x, x2, 0, 1);
    }
    if (urh_sc_abstop12(y) < 0x42f) {
        int n;
        x = urh_sc_reduce_fast(x, &n);
        const double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
        return urh_sinf_poly(x * s, x * x, (n & 2) != 0, n ^ 1);
    }
    if (urh_sc_abstop12(y) < 0x7f8) {
        union { float f; uint32_t u; } v; v.f = y;
        int n;
        x = urh_sc_reduce_large(v.u, &n);                   // cosine is even: the sign of y is ignored
        const double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
        return urh_sinf_poly(x * s, x * x, (n & 2) != 0, n ^ 1);
    }
    return urh_sincosf_invalid();
}

// sinf(y) and cosf(y) together, without branches, for |y| < 120 -- bit for bit what urh_sinf / urh_cosf return:
//   * |y| < 0.75 (their first branch) is the fast reduction with quadrant 0: r = y * (2/pi) 2^24 stays below 2^23 in magnitude, so
//     n = 0 and y - 0 * (pi/2) = y exactly; the sign is +1: the same polynomial on the same argument;
//   * |y| < 2^-12, where they return y and 1.0f without evaluating anything: the sine polynomial is y (1 - y^2/6 + ...) with
//     y^2/6 < 2^-26.5 -- less than half an ulp even just below a power of two -- and rounds to y; the cosine polynomial is
//     1 - y^2/2 + ... > 1 - 2^-25, closer to 1 than to the float below it; y = -0 is returned as it is.  (tests/test_gpu_parity.py
//     compares EVERY float below 120.)
//   * both results are the sine polynomial A of (x s) and the cosine polynomial B (negated in quadrants 2, 3 -- negating every
//     coefficient negates the result exactly): sin = n even ? A : B, cos = n even ? B : A.
// The Costas loop evaluates the pair once per sample and lane; with the branches every wavefront walked through all four of them.
URH_SC_HD void urh_sincosf_fast(float y, float *sn, float *cs) {
    const double s1c = -0x1.555545995a603p-3, s2c = 0x1.1107605230bc4p-7, s3c = -0x1.994eb3774cf24p-13;
    const double c0 = 0x1p0, c1c = -0x1.ffffffd0c621cp-2, c2c = 0x1.55553e1068f19p-5, c3c = -0x1.6c087e89a359dp-10, c4c = 0x1.99343027bf8c3p-16;
    int n;
    const double xr = urh_sc_reduce_fast((double)y, &n);
    const double sg = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    const double x = xr * sg, x2 = xr * xr;
    // sine polynomial (urh_sinf_poly, n even)
    const double x3 = x * x2;
    const double s1 = urh_sc_madd(s2c, x2, s3c);
    const double x7 = x3 * x2;
    const double sp = urh_sc_madd(x, x3, s1c);
    const float A = (float)urh_sc_madd(sp, x7, s1);
    // cosine polynomial (n odd), entry 1 of the table = every coefficient negated
    const double x4 = x2 * x2;
    const double c2 = urh_sc_madd(c3c, x2, c4c);
    const double c1 = urh_sc_madd(c0, x2, c1c);
    const double x6 = x4 * x2;
    const double cc = urh_sc_madd(c1, x4, c2c);
    const float B0 = (float)urh_sc_madd(cc, x6, c2);
    const float B = (n & 2) ? -B0 : B0;
    *sn = (y == 0.0f) ? y : ((n & 1) ? B : A);             // sinf(-0) = -0 (the polynomial's x + x^3 c turns it into +0)
    *cs = (n & 1) ? A : B;
}
