// Bit-faithful fp32 atanf / atan2f, usable from HIP device code and from host C/C++.
//
// WHY: the reference's FSK demodulator computes `atan2(tmp.imag, tmp.real)` on C `float`
// operands inside a module that is compiled as C++
// (/root/reference/src/urh/cythonext/signal_functions.pyx:376, setup.py:112), which binds to
// glibc's `atan2f`.  glibc 2.35's atan2f/atanf (sysdeps/ieee754/flt-32/e_atan2f.c, s_atanf.c) is
// the classic Sun fdlibm float algorithm: pure fp32 adds/muls/divs, NOT correctly rounded.
// ROCm's ocml atan2f is a different algorithm (differs by 1 ulp on ~16 % of inputs), so to get a
// demodulated signal (`Signal.qad`) that is bit-identical to the reference we restate the
// published fdlibm algorithm here.  Constants are the public fdlibm ones (Sun Microsystems,
// "Permission to use, copy, modify, and distribute this software is freely granted").
//
// REQUIREMENTS for bit-exactness: compile with -ffp-contract=off (no FMA contraction), IEEE
// correctly rounded fp32 division (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt), and
// fp32 denormals enabled (hipcc default on gfx9).
// tests/test_atan2f_port.py checks this file against the host libm on >1e8 inputs.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define URH_HD __host__ __device__ __forceinline__
#else
#define URH_HD static inline
#endif

URH_HD uint32_t urh_f2u(float f) {
    union { float f; uint32_t u; } v; v.f = f; return v.u;
}
URH_HD float urh_u2f(uint32_t u) {
    union { float f; uint32_t u; } v; v.u = u; return v.f;
}

// ---- atanf (fdlibm s_atanf.c, 11-term polynomial, glibc thresholds) --------------------------
// Polynomial tail shared by all branches: returns x*(s1+s2) with z=x*x, w=z*z.
URH_HD float urh_atanf_poly(float x) {
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f,
                aT3 = -1.1111110449e-01f, aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f,
                aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f, aT8 = 4.9768779427e-02f,
                aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    float z = x * x;
    float w = z * z;
    float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    return x * (s1 + s2);
}

URH_HD float urh_atanf(float x) {
    const float atanhi0 = 4.6364760399e-01f, atanhi1 = 7.8539812565e-01f,
                atanhi2 = 9.8279368877e-01f, atanhi3 = 1.5707962513e+00f;
    const float atanlo0 = 5.0121582440e-09f, atanlo1 = 3.7748947079e-08f,
                atanlo2 = 3.4473217170e-08f, atanlo3 = 7.5497894159e-08f;
    uint32_t hx = urh_f2u(x);
    uint32_t ix = hx & 0x7fffffffu;
    if (ix >= 0x4c000000u) {                      // |x| >= 2^25
        if (ix > 0x7f800000u) return x + x;       // NaN
        float r = atanhi3 + atanlo3;
        return (hx >> 31) ? -r : r;
    }
    if (ix < 0x3ee00000u) {                       // |x| < 0.4375
        if (ix < 0x31000000u) return x;           // |x| < 2^-29
        return x - urh_atanf_poly(x);
    }
    float ax = urh_u2f(ix);
    float hi, lo, t;
    if (ix < 0x3f980000u) {                       // |x| < 1.1875
        if (ix < 0x3f300000u) {                   // 7/16 <= |x| < 11/16
            hi = atanhi0; lo = atanlo0; t = (2.0f * ax - 1.0f) / (2.0f + ax);
        } else {                                  // 11/16 <= |x| < 19/16
            hi = atanhi1; lo = atanlo1; t = (ax - 1.0f) / (ax + 1.0f);
        }
    } else {
        if (ix < 0x401c0000u) {                   // |x| < 2.4375
            hi = atanhi2; lo = atanlo2; t = (ax - 1.5f) / (1.0f + 1.5f * ax);
        } else {                                  // 2.4375 <= |x| < 2^25
            hi = atanhi3; lo = atanlo3; t = -1.0f / ax;
        }
    }
    float z = hi - ((urh_atanf_poly(t) - lo) - t);
    return (hx >> 31) ? -z : z;
}

// ---- atan2f (fdlibm e_atan2f.c) ----------------------------------------------------------------
URH_HD float urh_atan2f(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f,
                pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    uint32_t hx = urh_f2u(x), hy = urh_f2u(y);
    uint32_t ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
    if (ix > 0x7f800000u || iy > 0x7f800000u) return x + y;   // NaN
    if (hx == 0x3f800000u) return urh_atanf(y);               // x == 1.0
    uint32_t m = (hy >> 31) | ((hx >> 30) & 2u);              // 2*sign(x)+sign(y)
    if (iy == 0) {
        switch (m) {
            case 0: case 1: return y;
            case 2: return pi + tiny;
            default: return -pi - tiny;
        }
    }
    if (ix == 0) return (hy >> 31) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000u) {
        if (iy == 0x7f800000u) {
            switch (m) {
                case 0: return pi_o_4 + tiny;
                case 1: return -pi_o_4 - tiny;
                case 2: return 3.0f * pi_o_4 + tiny;
                default: return -3.0f * pi_o_4 - tiny;
            }
        } else {
            switch (m) {
                case 0: return 0.0f;
                case 1: return -0.0f;
                case 2: return pi + tiny;
                default: return -pi - tiny;
            }
        }
    }
    if (iy == 0x7f800000u) return (hy >> 31) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    int32_t k = ((int32_t)iy - (int32_t)ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;                    // |y/x| > 2^60
    else if ((hx >> 31) && k < -60) z = 0.0f;                 // |y|/x < -2^60
    else z = urh_atanf(urh_u2f(urh_f2u(y / x) & 0x7fffffffu));
    switch (m) {
        case 0: return z;
        case 1: return urh_u2f(urh_f2u(z) ^ 0x80000000u);
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}
