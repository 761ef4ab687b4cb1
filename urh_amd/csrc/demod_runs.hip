// demod_runs.hip -- the hot kernel of the IQ->bits path for gfx950 (MI355X).
//
// ONE pass over the interleaved IQ stream does
//   (1) afp_demod           /root/reference/src/urh/cythonext/signal_functions.pyx:333-378
//   (2) the per-sample state classification and the tolerance hysteresis of grab_pulse_lens
//                            /root/reference/src/urh/cythonext/signal_functions.pyx:392-495
// and emits the demodulated signal (Signal.qad, optional) plus a compact list of "accepted run
// starts" per chunk.  The pulse table is then a handful of tiny kernels (pulse_table.hip).
//
// Roofline: HBM.  Algorithmic traffic 8 B (complex64 read) + 4 B (qad write) per sample; the run
// records are ~0.05 B/sample.  No MFMA: every stage is a map, a stencil, a ballot or a scan.
//
// Bit-exactness rules (tests/ compare against the oracle / the real reference):
//   * compiled with -ffp-contract=off; fp32 division and sqrt are the correctly rounded forms;
//   * FSK uses the fdlibm atan2f restated in fdlibm_atan2f.h (== glibc 2.35 atan2f);
//   * the conj(prev)*cur product follows the exact operation sequence the reference's generated
//     C++ performs (including the 0*x terms that decide signed zeros, see conj_mul()).
//
// Layout.  A row = 128 consecutive samples = one 1 KiB wavefront-wide load: lane t owns samples 2t, 2t+1 (one 16-byte load,
// lane-contiguous).  The previous sample of a lane's first sample comes from lane t-1 by DPP wave_shr:1; lane 0 takes the last
// sample of the previous row, carried in SGPRs (v_readlane) from row to row.
//   k_demod_runs_bp  (modulation orders 2 and 4, tolerance <= 64: the 1 GiB benchmark): a chunk is up to 64 rows shared by FOUR
//                    wavefronts (ASK: eight) of one workgroup, each streaming its own 16 rows with no barrier; the states are BIT
//                    PLANES -- the classification IS the v_cmp: 64-bit wavefront masks parked in lane r of a few registers for row
//                    r -- which meet in wavefront 0 through 2 KiB of LDS, where lane r analyses row r with 64-bit logic.
//   k_demod_runs     (every other order, tolerance > 64, the partial last tile): one wavefront per chunk walks tiles of 2048
//                    samples, state bytes in LDS in sample order, lane t owning 32 consecutive samples in the run phase.
#include <atomic>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>

#include "common.hpp"
#include "fdlibm_atan2f.h"
#include "launchers.hpp"
#include "runs.hpp"

namespace urh {

enum { SRC_IQ = 0, SRC_QAD = 1 };

// Tuning knobs (tools/kbench A/B builds override them): rows per load batch, __launch_bounds__ waves per SIMD.
#ifndef URH_KBATCH
#define URH_KBATCH 2
#endif
#ifndef URH_MINWAVES
#define URH_MINWAVES 1
#endif
#ifndef URH_SPEC
#ifndef URH_WIDE_INT
#define URH_WIDE_INT 0      // A/B builds: 1 gives the integer instantiations the wide loop too (with URH_HOT_WAVES_INT=7: at 64 VGPRs their fast loop spills
                            // beside it).  Measured, int8: wide deviations 0.452-0.469 -> 0.343-0.349 ms per pass, but the narrow capture 0.2568 ->
                            // 0.2687-0.2718 ms per pipelined step (seven wavefronts per SIMD, spills at the loop header): not the default; what it
                            // wants is an instantiation of its own that the launcher picks from what the stream's last pass met
#endif
#ifndef URH_NO_WIDE
#define URH_NO_WIDE 0     // A/B builds (tools/quick_tag.sh): 1 leaves the batch-level wide loop (fsk_wide) out
#endif
#define URH_SPEC 1        // branch-free speculative fast path per batch of rows (see spec_pair)
#endif
#ifndef URH_BITPLANE
#define URH_BITPLANE 1     // order-2 work goes through k_demod_runs_bp (0: tools/kbench A/B builds)
#endif
#ifndef URH_WPB
#define URH_WPB 4          // k_demod_runs_bp: wavefronts per chunk (1, 2, 4 or 8)
#endif
#ifndef URH_ASK_WPB
#define URH_ASK_WPB 8      // ... when it demodulates ASK (magnitudes: little per-wavefront start-up work, measured faster with 8)
#endif
#ifndef URH_NT
#define URH_NT 1          // non-temporal IQ loads / qad stores (streamed once): +8 % on the copy ceiling, tools/kbench
#endif

// ---- small device helpers ---------------------------------------------------------------------
__device__ __forceinline__ float dpp_wave_shr1(float x, float lane0) {
    // value of lane-1; lane 0 (no source lane) receives `lane0`
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(lane0), __float_as_int(x), 0x138 /*wave_shr:1*/, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane63(float x) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}

template <int DT> struct Iq;
template <> struct Iq<URHGPU_DT_F32> {
    static constexpr int kBytes = 8;
    static __device__ __forceinline__ void load2(const void *b, int64_t i, float &c0, float &d0, float &c1, float &d1) {
#if URH_NT
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f v = __builtin_nontemporal_load((const v4f *)b + (i >> 1)); c0 = v.x; d0 = v.y; c1 = v.z; d1 = v.w;
#else
        float4 v = ((const float4 *)b)[i >> 1]; c0 = v.x; d0 = v.y; c1 = v.z; d1 = v.w;
#endif
    }
    static __device__ __forceinline__ void load1(const void *b, int64_t i, float &c, float &d) {
        float2 v = ((const float2 *)b)[i]; c = v.x; d = v.y;
    }
    // a lane's two samples at `ptr` (16-byte aligned)
    static __device__ __forceinline__ void ld(const void *ptr, float &c0, float &d0, float &c1, float &d1) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f v = __builtin_nontemporal_load((const v4f *)ptr); c0 = v.x; d0 = v.y; c1 = v.z; d1 = v.w;
    }
};
template <> struct Iq<URHGPU_DT_I8> {
    static constexpr int kBytes = 2;
    static __device__ __forceinline__ void load2(const void *b, int64_t i, float &c0, float &d0, float &c1, float &d1) {
        char4 v = ((const char4 *)b)[i >> 1]; c0 = (float)v.x; d0 = (float)v.y; c1 = (float)v.z; d1 = (float)v.w;
    }
    static __device__ __forceinline__ void load1(const void *b, int64_t i, float &c, float &d) {
        char2 v = ((const char2 *)b)[i]; c = (float)v.x; d = (float)v.y;
    }
    static __device__ __forceinline__ void ld(const void *ptr, float &c0, float &d0, float &c1, float &d1) {
        char4 v = *(const char4 *)ptr; c0 = (float)v.x; d0 = (float)v.y; c1 = (float)v.z; d1 = (float)v.w;
    }
};
template <> struct Iq<URHGPU_DT_U8> {
    static constexpr int kBytes = 2;
    static __device__ __forceinline__ void load2(const void *b, int64_t i, float &c0, float &d0, float &c1, float &d1) {
        uchar4 v = ((const uchar4 *)b)[i >> 1]; c0 = (float)v.x; d0 = (float)v.y; c1 = (float)v.z; d1 = (float)v.w;
    }
    static __device__ __forceinline__ void load1(const void *b, int64_t i, float &c, float &d) {
        uchar2 v = ((const uchar2 *)b)[i]; c = (float)v.x; d = (float)v.y;
    }
    static __device__ __forceinline__ void ld(const void *ptr, float &c0, float &d0, float &c1, float &d1) {
        uchar4 v = *(const uchar4 *)ptr; c0 = (float)v.x; d0 = (float)v.y; c1 = (float)v.z; d1 = (float)v.w;
    }
};
template <> struct Iq<URHGPU_DT_I16> {
    static constexpr int kBytes = 4;
    static __device__ __forceinline__ void load2(const void *b, int64_t i, float &c0, float &d0, float &c1, float &d1) {
        short4 v = ((const short4 *)b)[i >> 1]; c0 = (float)v.x; d0 = (float)v.y; c1 = (float)v.z; d1 = (float)v.w;
    }
    static __device__ __forceinline__ void load1(const void *b, int64_t i, float &c, float &d) {
        short2 v = ((const short2 *)b)[i]; c = (float)v.x; d = (float)v.y;
    }
    static __device__ __forceinline__ void ld(const void *ptr, float &c0, float &d0, float &c1, float &d1) {
        short4 v = *(const short4 *)ptr; c0 = (float)v.x; d0 = (float)v.y; c1 = (float)v.z; d1 = (float)v.w;
    }
};
template <> struct Iq<URHGPU_DT_U16> {
    static constexpr int kBytes = 4;
    static __device__ __forceinline__ void load2(const void *b, int64_t i, float &c0, float &d0, float &c1, float &d1) {
        ushort4 v = ((const ushort4 *)b)[i >> 1]; c0 = (float)v.x; d0 = (float)v.y; c1 = (float)v.z; d1 = (float)v.w;
    }
    static __device__ __forceinline__ void load1(const void *b, int64_t i, float &c, float &d) {
        ushort2 v = ((const ushort2 *)b)[i]; c = (float)v.x; d = (float)v.y;
    }
    static __device__ __forceinline__ void ld(const void *ptr, float &c0, float &d0, float &c1, float &d1) {
        ushort4 v = *(const ushort4 *)ptr; c0 = (float)v.x; d0 = (float)v.y; c1 = (float)v.z; d1 = (float)v.w;
    }
};

// conj(a+jb) * (c+jd) exactly as the reference's generated C++ evaluates
//   (s[i-1,0] - imag_unit*s[i-1,1]) * (real + imag_unit*imag)           signal_functions.pyx:375
// with std::complex<float> operands: imag_unit*x is a full complex product (0*x - 1*0, 0*0 + 1*x),
// so signed zeros come out as on the CPU.  For finite non-zero inputs this is
// re = fl(fl(ac)+fl(bd)), im = fl(fl(ad)-fl(bc)).
__device__ __forceinline__ void conj_mul(float a, float b, float c, float d, float &re, float &im) {
    float A = a - 0.0f * b;          // a - (0*b - 1*0)
    float B = 0.0f - (0.0f + b);     // 0 - (0*0 + 1*b)
    float C = c + 0.0f * d;          // c + (0*d - 1*0)
    float D = 0.0f + d;              // 0 + (0*0 + 1*d)
    re = A * C - B * D;
    im = A * D + B * C;
}

// Message segmentation (seg_mode) on integer captures: util.get_magnitudes computes I*I + Q*Q in C `int` (wrapping) and
// takes the DOUBLE square root (util.pyx:128-136), and segment_messages_from_magnitudes compares that double with the
// float threshold (auto_interpretation.pyx:82).  The classification only needs "above" / "not above": return a value on
// the right side of every threshold.  (c, d) hold the integer sample exactly (|value| < 2^16).
__device__ __forceinline__ float seg_value_int(float c, float d, float threshold) {
    const int re = (int)c, im = (int)d;
    const int s = (int)((unsigned)(re * re) + (unsigned)(im * im));
    const double m = __builtin_sqrt((double)s);
    return (m > (double)threshold) ? 3.0e38f : -3.0e38f;
}

// atan2f for the common case (both operands finite, non-zero, exponents within 2^60 of each
// other); everything else goes through the literal port in fdlibm_atan2f.h.
__device__ __forceinline__ float atan2f_dev(float y, float x) {
    const uint32_t hx = __float_as_uint(x), hy = __float_as_uint(y);
    const uint32_t ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
    const int k = ((int)iy - (int)ix) >> 23;
    const bool special = (ix - 1u >= 0x7f7fffffu) | (iy - 1u >= 0x7f7fffffu) | (k > 60) | (k < -60);
    if (__builtin_expect(special, 0)) return urh_atan2f(y, x);
    const float pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const float r = __uint_as_float(__float_as_uint(y / x) & 0x7fffffffu);
    const uint32_t ir = __float_as_uint(r);
    float z;
    if (ir < 0x3ee00000u) {                       // |y/x| < 0.4375: no argument reduction
        z = (ir < 0x31000000u) ? r : r - urh_atanf_poly(r);
    } else if (ir >= 0x4c000000u) {               // >= 2^25
        z = 1.5707962513e+00f + 7.5497894159e-08f;
    } else {
        float hi, lo, num, den;
        if (ir < 0x3f300000u) { hi = 4.6364760399e-01f; lo = 5.0121582440e-09f; num = 2.0f * r - 1.0f; den = 2.0f + r; }
        else if (ir < 0x3f980000u) { hi = 7.8539812565e-01f; lo = 3.7748947079e-08f; num = r - 1.0f; den = r + 1.0f; }
        else if (ir < 0x401c0000u) { hi = 9.8279368877e-01f; lo = 3.4473217170e-08f; num = r - 1.5f; den = 1.0f + 1.5f * r; }
        else { hi = 1.5707962513e+00f; lo = 7.5497894159e-08f; num = -1.0f; den = r; }
        const float t = num / den;
        z = hi - ((urh_atanf_poly(t) - lo) - t);
    }
    const uint32_t m = (hy >> 31) | ((hx >> 30) & 2u);
    if (m == 0) return z;
    if (m == 1) return -z;
    if (m == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}

// ---- FSK fast path --------------------------------------------------------------------------------
// For the overwhelmingly common sample -- conj(prev)*cur has finite non-zero parts and
// 2^-29 <= |im/re| < 0.4375 (|phase step| < 0.412 rad or within 0.412 rad of pi) -- fdlibm's atan2f
// is straight-line code: one division, the 11-term polynomial without argument reduction and the
// quadrant fix-up.  kAtanLo/kAtanSpan express that range as one unsigned compare on |im/re|'s bits
// (NaN, inf, 0 and everything that needs argument reduction fall outside it).  A wavefront whose
// every lane is in range (or noise-gated) takes this path; otherwise the whole wavefront runs the
// general code (conj_mul + atan2f_dev above).  In range, the plain product below equals conj_mul():
// the two only differ in the sign of exact zeros, and a signed zero cannot change a non-zero sum.
constexpr uint32_t kAtanLo = 0x31000000u, kAtanSpan = 0x3ee00000u - 0x31000000u;

// Fast-path division.  For a denominator d in [2^-40, 2^40] and a quotient below 1 the IEEE-correct fp32
// division the compiler emits (v_div_scale / v_rcp / Newton / two residual corrections / v_div_fmas / v_div_fixup)
// never scales and never fixes anything up, so the bare Newton + residual chain below returns the same bits with
// three instructions less per quotient -- and, being plain fma/mul, it packs two quotients per v_pk_* instruction.
// kReLo/kReSpan express "d is a positive float in [2^-40, 2^40)" as one unsigned compare on d's bits.
#ifndef URH_FASTDIV
#define URH_FASTDIV 1
#endif
constexpr uint32_t kReLo = 0x2b800000u /* 2^-40 */, kReSpan = 0x53800000u /* 2^40 */ - 0x2b800000u;
__device__ __forceinline__ float div_fast(float n, float d) {
#if URH_FASTDIV
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = n * r;
    const float e2 = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(e2, r, q);
    const float e3 = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(e3, r, q);
#else
    return n / d;
#endif
}
// fdlibm atanf's first two argument reductions (s_atanf.c): 7/16 <= ax < 11/16: atan(1/2) + atan((2 ax - 1) / (2 + ax));
// 11/16 <= ax < 19/16 (used below 1 only): atan(1) + atan((ax - 1) / (ax + 1)).  Lanes that are not `mid` get (t, 1): the
// division then returns t itself.  kAtanMid*: "0.4375 <= ax < 1" as one unsigned compare.
constexpr uint32_t kAtanMidLo = 0x3ee00000u, kAtanMidSpan = 0x3f800000u - 0x3ee00000u;
__device__ __forceinline__ void atan_reduce(float ax, bool mid, float t, float &num, float &den, float &hi, float &lo) {
    const bool low = ax < 0.6875f;
    const float n_lo = 2.0f * ax - 1.0f, d_lo = 2.0f + ax, n_hi = ax - 1.0f, d_hi = ax + 1.0f;
    num = mid ? (low ? n_lo : n_hi) : t;
    den = mid ? (low ? d_lo : d_hi) : 1.0f;
    hi = low ? 4.6364760399e-01f : 7.8539812565e-01f;
    lo = low ? 5.0121582440e-09f : 3.7748947079e-08f;
}

// two quotients at once: the fma chain as 2-vectors (v_pk_fma_f32 / v_pk_mul_f32)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void div_fast2(float n0, float d0, float n1, float d1, float &q0, float &q1) {
    const v2f n = {n0, n1}, d = {d0, d1}, one = {1.0f, 1.0f};
    v2f r = {__builtin_amdgcn_rcpf(d0), __builtin_amdgcn_rcpf(d1)};
    const v2f e = __builtin_elementwise_fma(-d, r, one);
    r = __builtin_elementwise_fma(e, r, r);
    v2f q = n * r;
    const v2f e2 = __builtin_elementwise_fma(-d, q, n);
    q = __builtin_elementwise_fma(e2, r, q);
    const v2f e3 = __builtin_elementwise_fma(-d, q, n);
    q = __builtin_elementwise_fma(e3, r, q);
    q0 = q.x; q1 = q.y;
}


// The FSK pair of a lane when some lane of the row needs an argument reduction (mdX: 0.4375 <= |tX| < 1) or has an exactly
// zero cross product (zrX); every lane's denominator is in the fast division's window.  Lanes in the plain fast range come
// out as t - poly(t) like everywhere else.
__device__ __forceinline__ void fsk_reduced(float pc, float pd, float c0, float d0, float c1, float d1, float t0, float t1,
                                            bool md0, bool md1, bool zr0, bool zr1, float &q0, float &q1) {
    float n0v, d0v, h0, l0, n1v, d1v, h1, l1;
    atan_reduce(__uint_as_float(__float_as_uint(t0) & 0x7fffffffu), md0, t0, n0v, d0v, h0, l0);
    atan_reduce(__uint_as_float(__float_as_uint(t1) & 0x7fffffffu), md1, t1, n1v, d1v, h1, l1);
    float u0, u1;
    div_fast2(n0v, d0v, n1v, d1v, u0, u1);           // lanes in the fast range: t / 1 = t exactly
    const float p0 = urh_atanf_poly(u0), p1 = urh_atanf_poly(u1);
    const float z0 = h0 - ((p0 - l0) - u0), z1 = h1 - ((p1 - l1) - u1);
    q0 = md0 ? __uint_as_float(__float_as_uint(z0) | (__float_as_uint(t0) & 0x80000000u)) : t0 - p0;
    q1 = md1 ? __uint_as_float(__float_as_uint(z1) | (__float_as_uint(t1) & 0x80000000u)) : t1 - p1;
    if (__builtin_amdgcn_ballot_w64(zr0 | zr1) != 0) {
        // atan2f(+-0, re > 0) = +-0 (fdlibm: "atan(+-0, +anything) = +-0"), the sign being that of the reference's product
        // (conj_mul: its zeros are signed differently from the plain product's)
        float re_r, im_r;
        conj_mul(pc, pd, c0, d0, re_r, im_r); q0 = zr0 ? im_r : q0;
        conj_mul(c0, d0, c1, d1, re_r, im_r); q1 = zr1 ? im_r : q1;
    }
}

// The FSK pair of a lane in a row where some lane's phase step lies beyond pi/4 (|im/re| >= 1: wide deviations, low sample rates)
// or in the backward half-plane (re < 0: beyond pi/2) -- every lane regular: |re| in the fast division's window, 2^-29 <= |im/re|
// < 2^25, nothing gated.  fdlibm's atan2f without a branch: t = |im| / |re|, one of atanf's four argument reductions (or none)
// chosen by selects, the division of the reduced argument, the polynomial, hi - ((p - lo) - u), then the quadrant: pi - (z - pi_lo) for re < 0, the sign
// of im on top (a - b == -(b - a) exactly: cases 1 and 3 of e_atan2f.c are the negations of 0 and 2).  Before, such a row went
// through the rolled-up general code, sample by sample: 0.62 - 0.68 ms per GiB at deviations of +-100 kHz and more (1 MS/s).
// atanf's argument reduction as a table: range k of |t| (none, 7/16.., 11/16.., 19/16.., 39/16..) -> reduced argument
// (a t + b) / (c t + d) and the constants hi, lo of  atan(t) = hi - ((p - lo) - u).  Every entry reproduces fdlibm's expression
// bit for bit: 2 t - 1 = fl(fl(2 t) + -1), 2 + t = fl(fl(1 t) + 2), 1 + 1.5 t = fl(fl(1.5 t) + 1), -1 / t = (0 t + -1) / (1 t + 0);
// "none" is t / 1 with hi = lo = 0: 0 - ((p - 0) - u) = u - p exactly.  Six floats per range in LDS (a lane's range is its own):
// 15 instructions per sample instead of the 29 of nested selects.
struct AtanRow { float a, b, c, d, hi, lo, pad0, pad1; };
__device__ __forceinline__ AtanRow *atan_table() {
    __shared__ __attribute__((aligned(32))) AtanRow s_atan[5];
    return s_atan;
}
__device__ __forceinline__ AtanRow atan_row_of(int k) {
    AtanRow r;
    r.a = (k == 1) ? 2.0f : ((k == 4) ? 0.0f : 1.0f);
    r.b = (k == 0) ? 0.0f : ((k == 3) ? -1.5f : -1.0f);
    r.c = (k == 0) ? 0.0f : ((k == 3) ? 1.5f : 1.0f);
    r.d = (k == 0) ? 1.0f : ((k == 1) ? 2.0f : ((k == 4) ? 0.0f : 1.0f));
    r.hi = (k == 0) ? 0.0f : ((k == 1) ? 4.6364760399e-01f : ((k == 2) ? 7.8539812565e-01f : ((k == 3) ? 9.8279368877e-01f : 1.5707962513e+00f)));
    r.lo = (k == 0) ? 0.0f : ((k == 1) ? 5.0121582440e-09f : ((k == 2) ? 3.7748947079e-08f : ((k == 3) ? 3.4473217170e-08f : 7.5497894159e-08f)));
    r.pad0 = r.pad1 = 0.0f;
    return r;
}
// once per workgroup, before the first use (every kernel that reaches fsk_extended calls it, then a barrier)
__device__ __forceinline__ void atan_table_init() {
    if (threadIdx.x < 5) atan_table()[threadIdx.x] = atan_row_of((int)threadIdx.x);
}
// The same table indexed WITHOUT the four compares (the batch-level wide loop, fsk_wide): all four range bounds are multiples of 2^18 as
// bit patterns (7/16 = 0xFB8 << 18, 11/16 = 0xFCC << 18, 19/16 = 0xFE6 << 18, 39/16 = 0x1007 << 18), so bits >> 18, clamped to
// [0xFB7, 0x1007], names the range: 81 rows of (a, c, b, d, hi, lo) -- (a, c) and (b, d) as the pairs the packed multiply / add take.
constexpr uint32_t kLutLo = 0xFB7u, kLutRows = 0x1007u - 0xFB7u + 1u;
struct AtanLutRow { float a, c, b, d, hi, lo, pad0, pad1; };
__device__ __forceinline__ AtanLutRow *atan_lut() {
    __shared__ __attribute__((aligned(32))) AtanLutRow s_lut[kLutRows];
    return s_lut;
}
// Filled by the WAVEFRONT that enters the wide loop, when it first does (every wavefront writes the same values: no barrier, and a
// wavefront's own LDS operations execute in order) -- a workgroup that never leaves the fast loop pays nothing for it (filled by every
// workgroup up front it cost the headline capture 1.3 %: tools/ab_variants.py against -DURH_NO_WIDE=1).
__device__ __forceinline__ void atan_lut_fill_wave(int lane) {       // (behind atan_table_init() and its barrier)
    for (uint32_t i = (uint32_t)lane; i < kLutRows; i += 64u) {
        const uint32_t b = kLutLo + i;
        const AtanRow r = atan_table()[(int)(b >= 0xFB8u) + (int)(b >= 0xFCCu) + (int)(b >= 0xFE6u) + (int)(b >= 0x1007u)];
        AtanLutRow o;
        o.a = r.a; o.c = r.c; o.b = r.b; o.d = r.d; o.hi = r.hi; o.lo = r.lo; o.pad0 = o.pad1 = 0.0f;
        atan_lut()[i] = o;
    }
}
__device__ __forceinline__ void atan_reduce_full(float ax, float &num, float &den, float &hi, float &lo) {
    const uint32_t ir = __float_as_uint(ax);
    const int k = (int)(ir >= 0x3ee00000u) + (int)(ir >= 0x3f300000u) + (int)(ir >= 0x3f980000u) + (int)(ir >= 0x401c0000u);
    const AtanRow r = atan_table()[k];
    num = r.a * ax + r.b;
    den = r.c * ax + r.d;
    hi = r.hi; lo = r.lo;
}
constexpr uint32_t kAtanExtSpan = 0x4c000000u - 0x31000000u;          // 2^-29 <= t < 2^25 as one unsigned compare (with kAtanLo)
// t0 / t1 = |im / re| (the caller's signed quotients with the sign bit masked off: every step of the division is sign-symmetric)
__device__ __forceinline__ void fsk_extended(float re0, float im0, float re1, float im1, float t0, float t1, float &q0, float &q1) {
    const float pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    float n0, d0, h0, l0, n1, d1, h1, l1;
    atan_reduce_full(t0, n0, d0, h0, l0);
    atan_reduce_full(t1, n1, d1, h1, l1);
    float u0, u1;
    div_fast2(n0, d0, n1, d1, u0, u1);                                 // denominators in [1, 2^25), |quotients| < 1; no reduction: t / 1 = t
    const float p0 = urh_atanf_poly(u0), p1 = urh_atanf_poly(u1);
    const float z0 = h0 - ((p0 - l0) - u0), z1 = h1 - ((p1 - l1) - u1);
    const float b0 = (__float_as_int(re0) < 0) ? pi - (z0 - pi_lo) : z0, b1 = (__float_as_int(re1) < 0) ? pi - (z1 - pi_lo) : z1;
    q0 = __uint_as_float((__float_as_uint(b0) & 0x7fffffffu) | (__float_as_uint(im0) & 0x80000000u));
    q1 = __uint_as_float((__float_as_uint(b1) & 0x7fffffffu) | (__float_as_uint(im1) & 0x80000000u));
}

__device__ __forceinline__ float atan2f_small(float r, float y, float x) {
    // r = |y/x| in [2^-29, 0.4375)
    const float pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const float z = r - urh_atanf_poly(r);
    const float zx = pi - (z - pi_lo);                       // x < 0: pi - (z - pi_lo); y's sign negates exactly
    const float base = (__float_as_int(x) < 0) ? zx : z;
    return __uint_as_float((__float_as_uint(base) & 0x7fffffffu) | (__float_as_uint(y) & 0x80000000u));
}

template <bool ORDER2>
__device__ __forceinline__ uint32_t classify(float q, const RunArgs &p, bool check_noise = true) {
    if (check_noise && q == p.noise_val) return kStPause;
    if (ORDER2) return (q <= p.thr[0]) ? 1u : 2u;
    int st = p.order - 1;
    for (int k = 0; k < p.order - 1; ++k)
        if (q <= p.thr[k]) { st = k; break; }
    return (uint32_t)st + 1u;
}

// Demodulate one sample.  (pc,pd) = previous IQ sample, (c,d) = this one.
template <int MOD, int DT = URHGPU_DT_F32>
__device__ __forceinline__ float demod_one(float pc, float pd, float c, float d, const RunArgs &p) {
    if (MOD == URHGPU_MOD_ASK && DT != URHGPU_DT_F32 && p.seg_mode) return seg_value_int(c, d, p.thr[0]);
    const float mag = c * c + d * d;
    if (mag <= p.noise_sqrd) return p.noise_val;
    if (MOD == URHGPU_MOD_ASK) return __builtin_sqrtf(mag) / p.max_magnitude;   // (double)sqrtf/(double) == fp32 div
    if (MOD == URHGPU_MOD_FSK) {
        float re, im;
        conj_mul(pc, pd, c, d, re, im);
        return atan2f_dev(im, re);
    }
    return 0.0f;   // MOD_OTHER: np.zeros stays
}

// One 16-byte row slice of a lane: its two consecutive samples (SRC_QAD: c0/d0 = two demodulated samples).
struct RowIn { float c0, d0, c1, d1; };

// Issue the loads of rows [rb, rb+NB) of the tile at sample ta.
// FULL: the whole tile lies inside the capture, nothing is bounds-checked (the hot path).  Otherwise
// (the single partial tile at the end of a capture) samples at or beyond a1 read as 0.
template <int SRC, int DT, bool FULL, int NB>
__device__ __forceinline__ void load_rows(const RunArgs &p, int64_t ta, int rb, int t, int64_t a1, RowIn (&r)[NB]) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int64_t i0 = ta + (rb + j) * kRowSamples + 2 * t;
        r[j].c0 = r[j].d0 = r[j].c1 = r[j].d1 = 0.f;
        if (SRC == SRC_QAD) {
            const float *q = (const float *)p.in;
            if (FULL || i0 + 1 < a1) { const float2 v = *(const float2 *)(q + i0); r[j].c0 = v.x; r[j].d0 = v.y; }
            else if (i0 < a1) r[j].c0 = q[i0];
        } else {
            if (FULL || i0 + 1 < a1) Iq<DT>::load2(p.in, i0, r[j].c0, r[j].d0, r[j].c1, r[j].d1);
            else if (i0 < a1) Iq<DT>::load1(p.in, i0, r[j].c0, r[j].d0);
        }
    }
}

// The bit-plane kernel's loader: whole rows only; row rb + j of the chunk at sample a0 is a wavefront-uniform base (scalar arithmetic)
// plus the lane's 32-bit byte offset -- the global_load's SGPR-base + VGPR-offset form, no 64-bit VALU address arithmetic per row.
template <int SRC, int DT, int NB>
__device__ __forceinline__ void load_rows_bp(const RunArgs &p, int64_t a0, int rb, int lane, RowIn (&r)[NB]) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int64_t i0 = a0 + (int64_t)(rb + j) * kRowSamples;            // first sample of the row (uniform)
        if (SRC == SRC_QAD) {
            const char *row = (const char *)((const float *)p.in + i0);
            const float2 v = *(const float2 *)(row + (uint32_t)lane * 8u);
            r[j].c0 = v.x; r[j].d0 = v.y; r[j].c1 = r[j].d1 = 0.f;
        } else {
            const char *row = (const char *)p.in + i0 * Iq<DT>::kBytes;
            Iq<DT>::ld(row + (uint32_t)lane * (2u * Iq<DT>::kBytes), r[j].c0, r[j].d0, r[j].c1, r[j].d1);
        }
    }
}

// ... the same rows as whole 4-vectors (the complex64 FSK fast loop keeps them as the load delivers them)
template <int DT, int NB>
__device__ __forceinline__ void load_rows_v4(const RunArgs &p, int64_t a0, int rb, int lane, v4f (&r)[NB]) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const char *row = (const char *)p.in + (a0 + (int64_t)(rb + j) * kRowSamples) * Iq<DT>::kBytes;
        float c0, d0, c1, d1;
        Iq<DT>::ld(row + (uint32_t)lane * (2u * Iq<DT>::kBytes), c0, d0, c1, d1);
        r[j].x = c0; r[j].y = d0; r[j].z = c1; r[j].w = d1;
    }
}

// Demodulate a lane's two consecutive samples (q0, q1); all 64 lanes are active.  (prev_c, prev_d)
// is the sample before lane 0's first sample (wavefront-uniform).
// Returns 0 (or 3: the same, computed by the extended any-angle form) when q0/q1 are final and NO sample of the row is
// noise-gated (the fast path: states follow from the thresholds alone), 1 when q0/q1 are final but some sample is gated,
// 2 (FSK only) when some lane needs the general code: q0/q1 are then NOT valid and the caller redoes the row with
// fsk_row_general().
template <int MOD>
__device__ __forceinline__ int demod_pair(const RowIn &r, float prev_c, float prev_d, const RunArgs &p, float &q0, float &q1) {
    const float c0 = r.c0, d0 = r.d0, c1 = r.c1, d1 = r.d1;
    const float mag0 = c0 * c0 + d0 * d0, mag1 = c1 * c1 + d1 * d1;
    const bool n0 = mag0 <= p.noise_sqrd, n1 = mag1 <= p.noise_sqrd;
    const bool any_noise = __builtin_amdgcn_ballot_w64(n0 | n1) != 0;
    if (any_noise && __builtin_amdgcn_ballot_w64(n0 & n1) == ~0ull) {     // a row inside a pause: all NOISE
        q0 = q1 = p.noise_val;
        return 1;
    }
    if (MOD == URHGPU_MOD_FSK) {
        // Forward-quadrant fast path: re > 0 and 2^-29 <= |im/re| < 0.4375.  There atan2f(im, re) is
        // atanf(im/re) without argument reduction, and atanf is odd, so the signed quotient goes
        // straight through the polynomial (a negated operand negates every product and sum exactly).
        const float pc = dpp_wave_shr1(c1, prev_c), pd = dpp_wave_shr1(d1, prev_d);   // lane-1's second sample
        const float re0 = pc * c0 + pd * d0, im0 = pc * d0 - pd * c0;
        const float re1 = c0 * c1 + d0 * d1, im1 = c0 * d1 - d0 * c1;
#if URH_FASTDIV
        float t0, t1;
        div_fast2(im0, re0, im1, re1, t0, t1);
        // This function re-does the rows the speculative pass flagged.  Besides the fast range it accepts, still without a
        // branch per lane, 0.4375 <= |im/re| < 1 (phase steps up to pi/4: wide deviations, noisy samples) -- fdlibm's first
        // two argument reductions, one more division -- and exact zeros (see spec_pair).
        const uint32_t a0 = __float_as_uint(t0) & 0x7fffffffu, a1 = __float_as_uint(t1) & 0x7fffffffu;
        const bool rw0 = (__float_as_uint(re0) - kReLo < kReSpan), rw1 = (__float_as_uint(re1) - kReLo < kReSpan);
        const bool md0 = (a0 - kAtanMidLo < kAtanMidSpan), md1 = (a1 - kAtanMidLo < kAtanMidSpan);
        const bool zr0 = (im0 == 0.0f), zr1 = (im1 == 0.0f);
        const bool ok0 = (int)((a0 - kAtanLo < kAtanSpan) | md0 | zr0) & (int)rw0;
        const bool ok1 = (int)((a1 - kAtanLo < kAtanSpan) | md1 | zr1) & (int)rw1;
        if (!any_noise && __builtin_amdgcn_ballot_w64(!(ok0 & ok1)) == 0 && __builtin_amdgcn_ballot_w64(md0 | md1 | zr0 | zr1) != 0) {
            fsk_reduced(pc, pd, c0, d0, c1, d1, t0, t1, md0, md1, zr0, zr1, q0, q1);
            return 0;
        }
        {
            // any angle: |re| in the division's window (either sign), 2^-29 <= |im/re| < 2^25 (t's sign bit is masked off: a0 / a1)
            const bool ew0 = ((__float_as_uint(re0) & 0x7fffffffu) - kReLo < kReSpan), ew1 = ((__float_as_uint(re1) & 0x7fffffffu) - kReLo < kReSpan);
            const bool ex0 = (int)(a0 - kAtanLo < kAtanExtSpan) & (int)ew0, ex1 = (int)(a1 - kAtanLo < kAtanExtSpan) & (int)ew1;
            if (!any_noise && __builtin_amdgcn_ballot_w64(!(ex0 & ex1)) == 0) {
                fsk_extended(re0, im0, re1, im1, __uint_as_float(a0), __uint_as_float(a1), q0, q1);
                return 3;                                    // as 0, through the extended form (demod_batch counts these)
            }
        }
#else
        const float t0 = im0 / re0, t1 = im1 / re1;
        const bool ok0 = ((__float_as_uint(t0) & 0x7fffffffu) - kAtanLo < kAtanSpan) & (re0 > 0.0f);
        const bool ok1 = ((__float_as_uint(t1) & 0x7fffffffu) - kAtanLo < kAtanSpan) & (re1 > 0.0f);
#endif
        if (any_noise || __builtin_amdgcn_ballot_w64(!(ok0 & ok1)) != 0) return 2;
        q0 = t0 - urh_atanf_poly(t0);
        q1 = t1 - urh_atanf_poly(t1);
        return 0;
    }
    if (MOD == URHGPU_MOD_ASK) {
        q0 = __builtin_sqrtf(mag0) / p.max_magnitude;    // (double)sqrtf / (double)max == fp32 division
        q1 = __builtin_sqrtf(mag1) / p.max_magnitude;
    } else {
        q0 = q1 = 0.0f;                                  // MOD_OTHER: np.zeros stays
    }
    if (!any_noise && MOD == URHGPU_MOD_ASK) return 0;   // (MOD_OTHER: 0.0 may equal the sentinel -0.0 of "QAM")
    q0 = n0 ? p.noise_val : q0;
    q1 = n1 ? p.noise_val : q1;
    return 1;
}

// Branch-free speculative form of demod_pair's common case: computes the fast-path result of a lane's two
// samples unconditionally and returns true when the lane needs anything else (a noise-gated sample, an FSK
// quotient outside the fast range).  The caller ballots the flags of a whole batch of rows once and sends only
// the flagged rows through demod_pair: the hot loop has no per-row branches and the rows' dependent chains interleave.
template <int MOD, int DT = URHGPU_DT_F32>
__device__ __forceinline__ bool spec_pair(const RowIn &r, float prev_c, float prev_d, const RunArgs &p, float &q0, float &q1) {
    const float c0 = r.c0, d0 = r.d0, c1 = r.c1, d1 = r.d1;
    const float mag0 = c0 * c0 + d0 * d0, mag1 = c1 * c1 + d1 * d1;
    const bool n0 = mag0 <= p.noise_sqrd, n1 = mag1 <= p.noise_sqrd;
    if (MOD == URHGPU_MOD_FSK) {
        const float pc = dpp_wave_shr1(c1, prev_c), pd = dpp_wave_shr1(d1, prev_d);
        const float re0 = pc * c0 + pd * d0, im0 = pc * d0 - pd * c0;
        const float re1 = c0 * c1 + d0 * d1, im1 = c0 * d1 - d0 * c1;
        float t0, t1;
        div_fast2(im0, re0, im1, re1, t0, t1);
        const bool rw0 = (__float_as_uint(re0) - kReLo < kReSpan), rw1 = (__float_as_uint(re1) - kReLo < kReSpan);
        bool ok0 = (int)((__float_as_uint(t0) & 0x7fffffffu) - kAtanLo < kAtanSpan) & (int)rw0;
        bool ok1 = (int)((__float_as_uint(t1) & 0x7fffffffu) - kAtanLo < kAtanSpan) & (int)rw1;
        q0 = t0 - urh_atanf_poly(t0);
        q1 = t1 - urh_atanf_poly(t1);
        {
            // Integer captures (and float captures recorded from 8-bit receivers: multiples of 2^-7): the cross product of
            // two such samples is exact and EXACTLY zero about once in 700 samples at 8 bits -- one row in six would leave
            // the fast path.  atan2f(+-0, re > 0) = +-0 (fdlibm: "atan(+-0, +anything) = +-0"), the sign being that of the
            // reference's product (conj_mul: its zeros are signed differently from the plain product's).
            // (Lanes that need an argument reduction are NOT settled here but when the flagged row is re-done, demod_pair:
            // testing for them in this pass costs the rows that need nothing 1 % (complex64) to 9 % (int8).)
            const bool z0 = (im0 == 0.0f) & rw0, z1 = (im1 == 0.0f) & rw1;
            if (__builtin_amdgcn_ballot_w64(z0 | z1) != 0) {           // wavefront-uniform
                float re_r, im_r;
                conj_mul(pc, pd, c0, d0, re_r, im_r); q0 = z0 ? im_r : q0;
                conj_mul(c0, d0, c1, d1, re_r, im_r); q1 = z1 ? im_r : q1;
            }
            ok0 |= z0; ok1 |= z1;
        }
        return n0 | n1 | !ok0 | !ok1;
    }
    if (DT != URHGPU_DT_F32 && p.seg_mode) {
        q0 = seg_value_int(c0, d0, p.thr[0]);
        q1 = seg_value_int(c1, d1, p.thr[0]);
        return false;
    }
    q0 = __builtin_sqrtf(mag0) / p.max_magnitude;
    q1 = __builtin_sqrtf(mag1) / p.max_magnitude;
    return n0 | n1;
}

// The wide-deviation form of spec_pair: the branch-free extended atan2f (fsk_extended) for a lane's two samples, whatever the angle;
// true when the lane needs something else (a gated sample, |re| outside the division's window, |im/re| outside [2^-29, 2^25) --
// which includes an exactly zero cross product).  For lanes inside spec_pair's range the result is the same bits: there the
// reduction is "none", u = |t|, and t - poly(t) is odd in t.
__device__ __forceinline__ bool ext_pair(const RowIn &r, float prev_c, float prev_d, const RunArgs &p, float &q0, float &q1) {
    const float c0 = r.c0, d0 = r.d0, c1 = r.c1, d1 = r.d1;
    const float mag0 = c0 * c0 + d0 * d0, mag1 = c1 * c1 + d1 * d1;
    const bool n0 = mag0 <= p.noise_sqrd, n1 = mag1 <= p.noise_sqrd;
    const float pc = dpp_wave_shr1(c1, prev_c), pd = dpp_wave_shr1(d1, prev_d);
    const float re0 = pc * c0 + pd * d0, im0 = pc * d0 - pd * c0;
    const float re1 = c0 * c1 + d0 * d1, im1 = c0 * d1 - d0 * c1;
    float t0, t1;
    div_fast2(im0, re0, im1, re1, t0, t1);
    const uint32_t a0 = __float_as_uint(t0) & 0x7fffffffu, a1 = __float_as_uint(t1) & 0x7fffffffu;
    const bool ew0 = ((__float_as_uint(re0) & 0x7fffffffu) - kReLo < kReSpan), ew1 = ((__float_as_uint(re1) & 0x7fffffffu) - kReLo < kReSpan);
    const bool ex0 = (int)(a0 - kAtanLo < kAtanExtSpan) & (int)ew0, ex1 = (int)(a1 - kAtanLo < kAtanExtSpan) & (int)ew1;
    fsk_extended(re0, im0, re1, im1, __uint_as_float(a0), __uint_as_float(a1), q0, q1);
    return n0 | n1 | !ex0 | !ex1;
}

// The general FSK row (any operand class, any angle, noise gating).  Deliberately rolled up (one
// copy of the general atan2f per kernel) so that the hot loop stays small in the instruction cache.
__device__ __forceinline__ void fsk_row_general(const RowIn &r, float prev_c, float prev_d, const RunArgs &p, float &q0, float &q1) {
    const float pc = dpp_wave_shr1(r.c1, prev_c), pd = dpp_wave_shr1(r.d1, prev_d);
    float out[2];
#pragma unroll 1
    for (int s = 0; s < 2; ++s) {
        const float a = s ? r.c0 : pc, b = s ? r.d0 : pd, c = s ? r.c1 : r.c0, d = s ? r.d1 : r.d0;
        float re, im, q = p.noise_val;
        if (!(c * c + d * d <= p.noise_sqrd)) {
            conj_mul(a, b, c, d, re, im);
            q = atan2f_dev(im, re);
        }
        if (s) out[1] = q; else out[0] = q;
    }
    q0 = out[0]; q1 = out[1];
}

// ---- the batch-level FSK fast path (round 6: the VALU diet of the hot loop) ---------------------------------------------------
// spec_pair's arithmetic for a whole batch of rows with ONE flag for the batch, in two halves so that the rows' registers are free for
// the next batch's loads after the first:
//   fsk_front   the products, on the register pairs the 16-byte load delivers -- X0 = (c0, d0), X1 = (c1, d1): every v_pk_* reads
//               aligned pairs as they are (no v_mov marshalling):
//                 A = P * X0 = (pc c0, pd d0)    re0 = A.x + A.y       C = X0 * X1    = (c0 c1, d0 d1)   re1 = C.x + C.y
//                 B = P * X0.yx = (pc d0, pd c0) im0 = B.x - B.y       D = X0 * X1.yx = (c0 d1, d0 c1)   im1 = D.x - D.y
//               (the same products, sums and differences, each rounded once, as spec_pair's), the smallest |sample|^2 of the batch,
//               and the seam operand for the batch after it;
//   fsk_divide  t = im / re (div_fast), z = t t, and the batch's flag.  The range tests are taken over the batch with integer
//               min / max on the BIT PATTERNS (a NaN or a negative float is a large unsigned number: nothing hides, unlike v_max_f32):
//                 re in [2^-40, 2^40)                 the fast division's window;
//                 z = fl(t t) in [2^-58, 0.4375^2)    <=> 2^-29 <= |t| < 0.4375: both bounds are exactly representable squares and
//                                                     rounding is monotonic (the largest float below 0.4375 squares to 1.75 ulp below
//                                                     0.19140625); z is never negative;
//                 min |sample|^2 > noise_sqrd         nothing gated (v_min_f32 skips a NaN operand, but a NaN sample makes re a NaN).
//               An exactly zero cross product has z = 0: flagged.
//   fsk_finish  q = t - poly(t) for the batches that were not flagged.
// A flagged batch goes through demod_batch (the per-row pass) on its rows loaded again.
constexpr uint32_t kZLo = 0x22800000u /* 2^-58 */, kZHi = 0x3e440000u /* 0.19140625 = 0.4375^2 */;
__device__ __forceinline__ v2f atanf_poly2(v2f x, v2f z) {          // urh_atanf_poly on two samples: x (s1 + s2), z = x x given
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f,
                aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f,
                aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    const v2f w = z * z;
    const v2f s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const v2f s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    return x * (s1 + s2);
}
__device__ __forceinline__ v2f div_fast2v(v2f n, v2f d) {
    const v2f one = {1.0f, 1.0f};
    v2f r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const v2f e = __builtin_elementwise_fma(-d, r, one);
    r = __builtin_elementwise_fma(e, r, r);
    v2f q = n * r;
    const v2f e2 = __builtin_elementwise_fma(-d, q, n);
    q = __builtin_elementwise_fma(e2, r, q);
    const v2f e3 = __builtin_elementwise_fma(-d, q, n);
    return __builtin_elementwise_fma(e3, r, q);
}
template <int NB> struct FskFront { v2f re[NB], im[NB]; float mag_min; };
// rows as the load delivers them: (c0, d0, c1, d1) in four consecutive registers
template <int NB>
__device__ __forceinline__ void fsk_front(const v4f (&cur)[NB], float &prev_c, float &prev_d, FskFront<NB> &f) {
    float mag_min = 0.f;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const v2f X0 = cur[j].xy, X1 = cur[j].zw;
        const v2f M0 = X0 * X0, M1 = X1 * X1;
        const float m = __builtin_fminf(M0.x + M0.y, M1.x + M1.y);
        mag_min = (j == 0) ? m : __builtin_fminf(mag_min, m);
        const v2f P = {dpp_wave_shr1(X1.x, prev_c), dpp_wave_shr1(X1.y, prev_d)};
        const v2f A = P * X0, B = P * X0.yx, C = X0 * X1, D = X0 * X1.yx;
        // the four horizontal sums as scalar instructions into the halves of re / im (left alone, the compiler gathers their operands
        // into pairs with three v_mov per v_pk_add)
        float re0 = A.x + A.y, re1 = C.x + C.y, im0 = B.x - B.y, im1 = D.x - D.y;
        __asm__("" : "+v"(re0)); __asm__("" : "+v"(re1)); __asm__("" : "+v"(im0)); __asm__("" : "+v"(im1));
        f.re[j] = v2f{re0, re1}; f.im[j] = v2f{im0, im1};
        prev_c = lane63(X1.x); prev_d = lane63(X1.y);
    }
    f.mag_min = mag_min;
}
// ZEROS_OK (integer captures): an exactly zero cross product -- one sample in 700 at 8 bits -- does not flag the batch (z - 1 wraps to the
// largest unsigned number for z = 0: it passes the lower bound and z itself the upper one); the caller settles those samples in place.
template <int NB, bool ZEROS_OK>
__device__ __forceinline__ bool fsk_divide(const FskFront<NB> &f, const RunArgs &p, v2f (&t)[NB], v2f (&z)[NB]) {
    uint32_t re_max = 0u, re_min = 0u, z_max = 0u, z_min = 0u;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        t[j] = div_fast2v(f.im[j], f.re[j]);
        z[j] = t[j] * t[j];
        const uint32_t r0b = __float_as_uint(f.re[j].x), r1b = __float_as_uint(f.re[j].y), z0b = __float_as_uint(z[j].x), z1b = __float_as_uint(z[j].y);
        re_max = (j == 0) ? max(r0b, r1b) : max(re_max, max(r0b, r1b)); re_min = (j == 0) ? min(r0b, r1b) : min(re_min, min(r0b, r1b));
        z_max = (j == 0) ? max(z0b, z1b) : max(z_max, max(z0b, z1b));
        const uint32_t l0 = ZEROS_OK ? z0b - 1u : z0b, l1 = ZEROS_OK ? z1b - 1u : z1b;
        z_min = (j == 0) ? min(l0, l1) : min(z_min, min(l0, l1));
    }
    // (one ballot per compare: the masks are OR-ed on the scalar unit; a ballot of the OR-ed condition costs a v_cndmask + v_cmp)
    const uint64_t any = __builtin_amdgcn_ballot_w64(re_max >= kReLo + kReSpan) | __builtin_amdgcn_ballot_w64(re_min < kReLo) |
                         __builtin_amdgcn_ballot_w64(z_max >= kZHi) | __builtin_amdgcn_ballot_w64(z_min < (ZEROS_OK ? kZLo - 1u : kZLo)) |
                         __builtin_amdgcn_ballot_w64(f.mag_min <= p.noise_sqrd);
    return any != 0;
}

// ---- the batch-level WIDE form (round 6, late): fsk_extended for a whole batch on the same register pairs -------------------------------
// A capture whose phase steps reach beyond atan(7/16) = 0.41 rad per sample -- deviations from about 50 kHz at 1 MS/s with some noise, few
// samples per symbol, a filtered capture -- flags EVERY batch of the fast loop; the per-row forms it then went through (spec_pair ->
// demod_pair -> ext_pair, scalar polynomials, four compares per range) cost 203-217 VALU wave-instructions per row against 89
// (profiles/r06s_deviation_pmc.txt).  The wide loop is the fast loop with fdlibm's argument reduction in it:
//   t = im / re (div_fast, signed), ax = |t|;  range row = lut[clamp(bits(ax) >> 18)]: (num, den) = (a, c) ax + (b, d) as ONE packed
//   multiply and ONE packed add per sample;  u = num / den (div_fast);  z = hi - ((poly(u) - lo) - u);  re < 0: pi - (z - pi_lo);
//   the sign of im on top -- statement for statement fsk_extended (whose results fdlibm's atan2f pins), the polynomial packed.
// Window (one flag per batch, integer min / max over bit patterns as in fsk_divide): |re| in [2^-40, 2^40), 2^-29 <= ax < 2^25, nothing
// gated.  Returns 1 when the batch lies outside it (q invalid), else 0, or 2 when it also lies inside the fast loop's window (the caller
// counts those to find its way back).  ZEROS_OK (integer captures): exact zeros of im pass; the caller settles them.
template <int NB, bool ZEROS_OK>
__device__ __forceinline__ int fsk_wide(const FskFront<NB> &f, const RunArgs &p, float (&q0)[NB], float (&q1)[NB]) {
    const float pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    v2f t[NB];
    uint32_t ax[NB][2];
    uint32_t re_max = 0u, re_min = 0u, a_max = 0u, a_min = 0u, raw_max = 0u;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        t[j] = div_fast2v(f.im[j], f.re[j]);
        ax[j][0] = __float_as_uint(t[j].x) & 0x7fffffffu; ax[j][1] = __float_as_uint(t[j].y) & 0x7fffffffu;
        const uint32_t w0 = __float_as_uint(f.re[j].x), w1 = __float_as_uint(f.re[j].y), r0b = w0 & 0x7fffffffu, r1b = w1 & 0x7fffffffu;
        const uint32_t l0 = ZEROS_OK ? ax[j][0] - 1u : ax[j][0], l1 = ZEROS_OK ? ax[j][1] - 1u : ax[j][1];
        re_max = (j == 0) ? max(r0b, r1b) : max(re_max, max(r0b, r1b)); re_min = (j == 0) ? min(r0b, r1b) : min(re_min, min(r0b, r1b));
        a_max = (j == 0) ? max(ax[j][0], ax[j][1]) : max(a_max, max(ax[j][0], ax[j][1]));
        a_min = (j == 0) ? min(l0, l1) : min(a_min, min(l0, l1));
        raw_max = (j == 0) ? max(w0, w1) : max(raw_max, max(w0, w1));
    }
    const uint64_t out = __builtin_amdgcn_ballot_w64(re_max >= kReLo + kReSpan) | __builtin_amdgcn_ballot_w64(re_min < kReLo) |
                         __builtin_amdgcn_ballot_w64(a_max >= kAtanLo + kAtanExtSpan) | __builtin_amdgcn_ballot_w64(a_min < (ZEROS_OK ? kAtanLo - 1u : kAtanLo)) |
                         __builtin_amdgcn_ballot_w64(f.mag_min <= p.noise_sqrd);
    if (out != 0) return 1;
    const uint64_t beyond = __builtin_amdgcn_ballot_w64(a_max >= kAtanLo + kAtanSpan) | __builtin_amdgcn_ballot_w64(raw_max >= 0x80000000u);
    const AtanLutRow *const lut = atan_lut();
    // (stage by stage over the batch's rows: the rows' chains interleave, as in the fast loop)
    v2f n[NB], d[NB], hi[NB], lo[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const uint32_t i0 = min(max(ax[j][0] >> 18, kLutLo), kLutLo + kLutRows - 1u) - kLutLo, i1 = min(max(ax[j][1] >> 18, kLutLo), kLutLo + kLutRows - 1u) - kLutLo;
        const v4f c0 = *(const v4f *)&lut[i0].a, c1 = *(const v4f *)&lut[i1].a;
        const v2f h0 = *(const v2f *)&lut[i0].hi, h1 = *(const v2f *)&lut[i1].hi;
        const float a0 = __uint_as_float(ax[j][0]), a1 = __uint_as_float(ax[j][1]);
        const v2f nd0 = c0.xy * v2f{a0, a0} + c0.zw, nd1 = c1.xy * v2f{a1, a1} + c1.zw;        // (num, den) of each sample: multiply and add rounded separately
        n[j] = v2f{nd0.x, nd1.x}; d[j] = v2f{nd0.y, nd1.y};
        hi[j] = v2f{h0.x, h1.x}; lo[j] = v2f{h0.y, h1.y};
    }
    v2f u[NB], pp[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) u[j] = div_fast2v(n[j], d[j]);
#pragma unroll
    for (int j = 0; j < NB; ++j) pp[j] = atanf_poly2(u[j], u[j] * u[j]);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const v2f z = hi[j] - ((pp[j] - lo[j]) - u[j]);
        const v2f zb = v2f{pi, pi} - (z - v2f{pi_lo, pi_lo});
        const float b0 = (__float_as_int(f.re[j].x) < 0) ? zb.x : z.x, b1 = (__float_as_int(f.re[j].y) < 0) ? zb.y : z.y;
        q0[j] = __uint_as_float((__float_as_uint(b0) & 0x7fffffffu) | (__float_as_uint(f.im[j].x) & 0x80000000u));
        q1[j] = __uint_as_float((__float_as_uint(b1) & 0x7fffffffu) | (__float_as_uint(f.im[j].y) & 0x80000000u));
    }
    return beyond != 0 ? 0 : 2;
}

// Demodulate one batch of NB rows (cur[j] = a lane's two samples of row j).  (prev_c, prev_d) is the IQ sample before
// the batch (wavefront-uniform) and is advanced to the batch's last sample.  Returns the per-row "gated" flags (bit j:
// some sample of row j may equal the NOISE sentinel, so the classification has to test for it).
template <int SRC, int DT, int MOD, int NB>
__device__ __forceinline__ uint32_t demod_batch(const RowIn (&cur)[NB], float &prev_c, float &prev_d, const RunArgs &p,
                                                float (&q0)[NB], float (&q1)[NB], uint32_t &hint) {
    constexpr int kBatch = NB;
    uint32_t general = 0, gated = 0;                        // per-row flags, wavefront-uniform
    float pcs[kBatch], pds[kBatch];                         // seam operand of each row (uniform)
#if URH_SPEC
    if (SRC == SRC_IQ && MOD != URHGPU_MOD_OTHER) {
        uint32_t bad = 0;
        uint32_t form = 0;
        if (MOD == URHGPU_MOD_FSK && hint != 0) {
            // The narrow speculative pass has just been failing for every row (wide deviation, noise): the next batches skip it.
            // Form 1: every row goes straight to the re-do below (demod_pair: narrow / middle / extended form, row by row, rolled
            // up); when most rows of a batch needed the extended form there -- phase steps beyond pi/4 -- form 2: the extended form
            // for every row, unrolled like the narrow pass.  hint = form << 4 | batches left until the next narrow attempt
            // (wavefront-uniform); form 2 steps down to one batch of form 1.
            form = hint >> 4;
            hint = ((hint & 15u) > 1u) ? hint - 1u : ((form == 2u) ? ((1u << 4) | 1u) : 0u);
            if (form == 1u) {
                bad = (1u << kBatch) - 1u;
#pragma unroll
                for (int j = 0; j < kBatch; ++j) {
                    pcs[j] = prev_c; pds[j] = prev_d;
                    prev_c = lane63(cur[j].c1); prev_d = lane63(cur[j].d1);
                }
            } else {
                bool flag[kBatch];
#pragma unroll
                for (int j = 0; j < kBatch; ++j) {
                    pcs[j] = prev_c; pds[j] = prev_d;
                    flag[j] = ext_pair(cur[j], prev_c, prev_d, p, q0[j], q1[j]);
                    prev_c = lane63(cur[j].c1); prev_d = lane63(cur[j].d1);
                }
#pragma unroll
                for (int j = 0; j < kBatch; ++j) if (__builtin_amdgcn_ballot_w64(flag[j]) != 0) bad |= 1u << j;
            }
        } else {
            bool flag[kBatch];
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                pcs[j] = prev_c; pds[j] = prev_d;
                flag[j] = spec_pair<MOD, DT>(cur[j], prev_c, prev_d, p, q0[j], q1[j]);
                if (MOD == URHGPU_MOD_FSK) { prev_c = lane63(cur[j].c1); prev_d = lane63(cur[j].d1); }
            }
#pragma unroll
            for (int j = 0; j < kBatch; ++j) if (__builtin_amdgcn_ballot_w64(flag[j]) != 0) bad |= 1u << j;
            // every row of the batch: likely to go on.  (Switching as soon as a quarter of the rows are flagged was measured: deviations
            // of 50 - 70 kHz, where a minority of rows holds a sample beyond 0.4375, went from 0.45 to 0.49 ms -- the extended form
            // costs every row about three times the narrow one.)
            if (MOD == URHGPU_MOD_FSK && bad == (1u << kBatch) - 1u) hint = (1u << 4) | 7u;
        }
        if (__builtin_expect(bad != 0, 0)) {
            int n_ext = 0;
#pragma unroll 1
            for (int j = 0; j < kBatch; ++j) {
                if (!((bad >> j) & 1u)) continue;
                RowIn r = cur[0]; float pc = pcs[0], pd = pds[0];
#pragma unroll
                for (int k = 1; k < kBatch; ++k) if (j == k) { r = cur[k]; pc = pcs[k]; pd = pds[k]; }
                float g0 = 0.f, g1 = 0.f;
                int kind = demod_pair<MOD>(r, pc, pd, p, g0, g1);
                if (kind == 3) { kind = 0; ++n_ext; }
                if (kind == 2) general |= 1u << j;
                else {
#pragma unroll
                    for (int k = 0; k < kBatch; ++k) if (j == k) { q0[k] = g0; q1[k] = g1; }
                }
                if (kind != 0) gated |= 1u << j;
            }
            if (MOD == URHGPU_MOD_FSK && form == 1u && 2 * n_ext >= kBatch) hint = (2u << 4) | 7u;
        }
    } else
#endif
    {
#pragma unroll
    for (int j = 0; j < kBatch; ++j) {
        if (SRC == SRC_QAD) { q0[j] = cur[j].c0; q1[j] = cur[j].d0; gated |= 1u << j; continue; }
        pcs[j] = prev_c; pds[j] = prev_d;
        const int k = demod_pair<MOD>(cur[j], prev_c, prev_d, p, q0[j], q1[j]);
        if (k == 2) general |= 1u << j;
        if (k != 0 && k != 3) gated |= 1u << j;
        if (MOD == URHGPU_MOD_FSK) { prev_c = lane63(cur[j].c1); prev_d = lane63(cur[j].d1); }
    }
    }
    if (MOD == URHGPU_MOD_FSK && SRC == SRC_IQ && general) {
#pragma unroll 1
        for (int j = 0; j < kBatch; ++j) {
            if (!((general >> j) & 1u)) continue;
            RowIn r = cur[0]; float pc = pcs[0], pd = pds[0];
#pragma unroll
            for (int k = 1; k < kBatch; ++k) if (j == k) { r = cur[k]; pc = pcs[k]; pd = pds[k]; }
            float g0, g1;
            fsk_row_general(r, pc, pd, p, g0, g1);
#pragma unroll
            for (int k = 0; k < kBatch; ++k) if (j == k) { q0[k] = g0; q1[k] = g1; }
        }
    }
    return gated;
}

// Chunk prologue: the state of sample a0-1 (kStNone when the capture starts at a0) and, for FSK, the IQ sample a0-1
// itself (the seam operand of the chunk's first sample).  Wavefront-uniform loads.
template <int SRC, int DT, int MOD, bool ORDER2>
__device__ __forceinline__ uint32_t chunk_prologue(const RunArgs &p, int64_t a0, bool global_start, float &prev_c, float &prev_d) {
    uint32_t st = kStNone;
    if (SRC == SRC_QAD) {
        const float *q = (const float *)p.in;
        if (a0 > 0) st = classify<ORDER2>(q[a0 - 1], p);
        else if (!global_start) st = classify<ORDER2>(((const float *)p.left_halo)[0], p);
    } else {
        // qad[a0-1] needs IQ[a0-1] and (FSK) IQ[a0-2]
        float pc = 0, pd = 0;
        bool have = false, is_global0 = false;
        if (a0 >= 1) {
            Iq<DT>::load1(p.in, a0 - 1, prev_c, prev_d);
            have = true;
            if (a0 >= 2) Iq<DT>::load1(p.in, a0 - 2, pc, pd);
            else if (!global_start) Iq<DT>::load1(p.left_halo, 1, pc, pd);
            else is_global0 = true;           // sample a0-1 is global sample 0 -> NOISE
        } else if (!global_start) {
            Iq<DT>::load1(p.left_halo, 1, prev_c, prev_d);
            Iq<DT>::load1(p.left_halo, 0, pc, pd);
            have = true;
        }
        if (have) {
            const float q = is_global0 ? p.noise_val : demod_one<MOD, DT>(pc, pd, prev_c, prev_d, p);
            st = classify<ORDER2>(q, p);
        }
    }
    return st;
}

// initial cur_state of the reference state machine (signal_functions.pyx:421-429), kept in chunk 0's ChunkInfo:
// PAUSE if samples[0] == NOISE else the state of the literal 0.0
template <int SRC, int DT, int MOD, bool ORDER2>
__device__ __forceinline__ uint32_t chunk_init_state(const RunArgs &p, int64_t chunk) {
    uint32_t init = 0;
    if (chunk == 0) {
        bool first_is_noise;
        if (SRC == SRC_QAD) first_is_noise = (((const float *)p.in)[0] == p.noise_val);
        else first_is_noise = true;                            // afp_demod: result[0] = NOISE
        init = first_is_noise ? kStPause : classify<ORDER2>(0.0f, p, false);   // literal 0.0: thresholds only
        if (SRC == SRC_QAD && p.seg_mode) init = classify<ORDER2>(((const float *)p.in)[0], p);   // segmentation: the state of sample 0 itself
        if (SRC == SRC_IQ && p.seg_mode) {
            float c = 0.f, d = 0.f;
            Iq<DT>::load1(p.in, 0, c, d);
            init = classify<ORDER2>(demod_one<MOD, DT>(0.f, 0.f, c, d, p), p);
        }
    }
    return init;
}

// Block-wide helpers --------------------------------------------------------------------------------
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int u = __shfl_up(v, o);
        if (lane >= o) v += u;
    }
    return v;
}

// Nibble (4 bits, sample order) of "byte j differs from the byte before it" for one LDS word.
__device__ __forceinline__ uint32_t diff_nibble(uint32_t w, uint32_t pw) {
    const uint32_t sh = __builtin_amdgcn_alignbyte(w, pw, 3);   // (w << 8) | (pw >> 24)
    const uint32_t x = w ^ sh;
    uint32_t y = ((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x;          // bit 7 of each byte = byte != 0
    y = (y >> 7) & 0x01010101u;
    return (y * 0x01020408u) >> 24;
}

// -----------------------------------------------------------------------------------------------------
// k_demod_runs<SRC, DT, MOD, ORDER2, WRITE_QAD, FULL>
//   one workgroup per chunk; workgroup b handles chunk p.chunk_base + b = samples
//   [p.range_begin + b*chunk_len, min(.. + chunk_len, p.range_end)).  FULL: the range consists of whole
//   tiles only (the hot launch); the partial tile at the end of a capture is a chunk of its own,
//   handled by a one-workgroup launch of the bounds-checked instantiation.
// -----------------------------------------------------------------------------------------------------
template <int SRC, int DT, int MOD, bool ORDER2, bool WRITE_QAD, bool FULL>
__global__ __launch_bounds__(kBlock, URH_MINWAVES) void k_demod_runs(const RunArgs p) {
    __shared__ __attribute__((aligned(16))) uint8_t s_state[16 + kTile];   // [15] = state of the sample before the tile
    __shared__ uint32_t s_bm[kBlock + 1];
    __shared__ int s_first, s_last;            // first / last boundary offset in the tile (or kTile / -1)
    constexpr int kWaves = kBlock / 64;
    __shared__ int s_wave_cnt[kWaves];
    __shared__ uint32_t s_wave_last[kWaves];
    // chunk-level carries
    __shared__ int64_t s_pend_pos;             // unresolved run start (absolute) or -1
    __shared__ uint32_t s_pend_state;
    __shared__ int64_t s_lead;                 // -1 until the chunk's first boundary is seen
    __shared__ uint32_t s_carry_last;          // state of the chunk's last stable run so far, 0xFFFF = none
    __shared__ uint32_t s_first_state;
    __shared__ int s_count;                    // records written so far
    __shared__ uint32_t s_prev_state8;         // state byte of the sample before the next tile
    __shared__ int s_newpend;                  // offset (in tile) of this tile's unresolved last run or -1
    __shared__ unsigned long long s_last_pos;  // position of the last record written (records are position-ordered)

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t chunk = p.chunk_base + blockIdx.x;
    const int64_t a0 = p.range_begin + (int64_t)blockIdx.x * p.chunk_len;
    const int64_t a1 = (a0 + p.chunk_len < p.range_end) ? a0 + p.chunk_len : p.range_end;
    uint64_t *slab = p.slab + chunk * p.slab_stride;
    const bool global_start = (p.left_halo == nullptr);
    if (SRC == SRC_IQ && MOD == URHGPU_MOD_FSK) { atan_table_init(); __syncthreads(); }

    // ---- chunk prologue: the two samples before the chunk (wavefront-uniform loads), the state of
    // sample a0-1, initial carries ------------------------------------------------------------------
    float prev_c = 0.f, prev_d = 0.f;          // IQ sample a0-1: seam operand of the chunk's first sample
    {
        const uint32_t st = chunk_prologue<SRC, DT, MOD, ORDER2>(p, a0, global_start, prev_c, prev_d);
        if (t == 0) {
            s_prev_state8 = st;
            s_pend_pos = -1; s_pend_state = 0; s_lead = -1; s_carry_last = 0xFFFFu; s_first_state = 0xFFFFu; s_count = 0; s_last_pos = 0;
        }
    }
    __syncthreads();

    constexpr int kBatch = URH_KBATCH;   // rows (16-byte loads per thread) in flight per batch
    RowIn cur[kBatch], nxt[kBatch];
    bool have_cur = false;         // cur[] already holds rows 0..kBatch-1 of the tile about to start (uniform)
    uint32_t spec_hint = 0;        // demod_batch: batches left that skip the speculative pass
    for (int64_t ta = a0; ta < a1; ta += kTile) {
        const int tv = (int)((a1 - ta < kTile) ? (a1 - ta) : kTile);     // valid samples in this tile
        if (t == 0) {
            s_state[15] = (uint8_t)s_prev_state8;
            s_first = kTile; s_last = -1; s_newpend = -1;
        }
        // ================= phase 1: demodulate + classify, 2 samples per lane per row ====================
        // Loads are software-pipelined one batch ahead; the last batch of a tile prefetches the first
        // batch of the next tile, so the run phase below overlaps the memory latency.
        {
            const bool first_row = (ta == 0) && global_start && (t == 0);   // sample 0 of the capture is mine
            if (!have_cur) load_rows<SRC, DT, FULL>(p, ta, 0, t, a1, cur);
#pragma unroll 1
            for (int rb = 0; rb < kRows; rb += kBatch) {
                if (rb + kBatch < kRows) {
                    load_rows<SRC, DT, FULL>(p, ta, rb + kBatch, t, a1, nxt);
                } else {
                    have_cur = (ta + kTile < a1);
                    if (have_cur) load_rows<SRC, DT, FULL>(p, ta + kTile, 0, t, a1, nxt);
                }
                float q0[kBatch], q1[kBatch];
                const uint32_t gated = demod_batch<SRC, DT, MOD, kBatch>(cur, prev_c, prev_d, p, q0, q1, spec_hint);
#pragma unroll
                for (int j = 0; j < kBatch; ++j) {
                    const int off = (rb + j) * kRowSamples + 2 * t;
                    if (SRC == SRC_IQ) {
                        if (j == 0 && rb == 0 && first_row && !p.seg_mode) q0[0] = p.noise_val;   // result[0] = NOISE (:361)
                        if (WRITE_QAD) {
                            float w0 = q0[j], w1 = q1[j];
                            if (MOD == URHGPU_MOD_ASK && DT == URHGPU_DT_F32 && p.seg_mode) {     // see k_demod_runs_bp
                                const float m0 = cur[j].c0 * cur[j].c0 + cur[j].d0 * cur[j].d0, m1 = cur[j].c1 * cur[j].c1 + cur[j].d1 * cur[j].d1;
                                w0 = (m0 <= p.dm_noise_sqrd) ? p.dm_noise_val : __builtin_sqrtf(m0) / p.dm_max_magnitude;
                                w1 = (m1 <= p.dm_noise_sqrd) ? p.dm_noise_val : __builtin_sqrtf(m1) / p.dm_max_magnitude;
                                if (j == 0 && rb == 0 && first_row) w0 = p.dm_noise_val;
                            }
#if URH_NT
                            typedef float v2s __attribute__((ext_vector_type(2)));
                            if (FULL || ta + off + 1 < a1) { const v2s qq = {w0, w1}; __builtin_nontemporal_store(qq, (v2s *)(p.qad + ta + off)); }
#else
                            if (FULL || ta + off + 1 < a1) *(float2 *)(p.qad + ta + off) = make_float2(w0, w1);
#endif
                            else if (ta + off < a1) p.qad[ta + off] = w0;
                        }
                    }
                    uint32_t st0, st1;
                    if (((gated >> j) & 1u) || (j == 0 && rb == 0 && ta == 0 && global_start)) {
                        st0 = classify<ORDER2>(q0[j], p); st1 = classify<ORDER2>(q1[j], p);
                    } else {                                            // nothing gated: thresholds only
                        st0 = classify<ORDER2>(q0[j], p, false); st1 = classify<ORDER2>(q1[j], p, false);
                    }
                    *(uint16_t *)(s_state + 16 + off) = (uint16_t)(st0 | (st1 << 8));   // bytes beyond tv are masked in phase 2
                }
#pragma unroll
                for (int j = 0; j < kBatch; ++j) cur[j] = nxt[j];
            }
        }
        __syncthreads();   // A: states complete

        // ================= phase 2: runs.  thread t owns samples [32t, 32t+32) of the tile ==============
        uint32_t bm = 0;
        {
            const uint4 wa = *(const uint4 *)(s_state + 16 + kSpan * t);
            const uint4 wb = *(const uint4 *)(s_state + 16 + kSpan * t + 16);
            const uint32_t pw = *(const uint32_t *)(s_state + 16 + kSpan * t - 4);
            bm |= diff_nibble(wa.x, pw);
            bm |= diff_nibble(wa.y, wa.x) << 4;
            bm |= diff_nibble(wa.z, wa.y) << 8;
            bm |= diff_nibble(wa.w, wa.z) << 12;
            bm |= diff_nibble(wb.x, wa.w) << 16;
            bm |= diff_nibble(wb.y, wb.x) << 20;
            bm |= diff_nibble(wb.z, wb.y) << 24;
            bm |= diff_nibble(wb.w, wb.z) << 28;
            const int vc = tv - kSpan * t;                      // valid samples in my span
            if (vc < kSpan) bm = (vc <= 0) ? 0u : (bm & ((1u << vc) - 1u));
        }
        s_bm[t] = bm;
        if (bm) {
            atomicMin(&s_first, kSpan * t + __builtin_ctz(bm));
            atomicMax(&s_last, kSpan * t + 31 - __builtin_clz(bm));
        }
        __syncthreads();   // B: boundary masks complete

        // stable mask: run starting at boundary p is stable iff (next boundary - p) > tol
        uint32_t stable = 0;
        if (bm) {
            // next boundary after my span (bounded look-ahead: beyond tol it does not matter)
            int q = tv;
            {
                const int limit = kSpan * t + 31 + p.tol + 1;     // boundaries at or beyond this never matter
                for (int u = t + 1; u < kBlock && kSpan * u < limit && kSpan * u < tv; ++u) {
                    const uint32_t m = s_bm[u];
                    if (m) { q = kSpan * u + __builtin_ctz(m); break; }
                }
            }
            uint32_t m = bm;
            while (m) {
                const int hi = 31 - __builtin_clz(m);
                const int pos = kSpan * t + hi;
                if (q - pos > p.tol) stable |= 1u << hi;
                else if (pos == s_last) s_newpend = pos;          // the tile's last run, still short: carry on
                q = pos;
                m &= ~(1u << hi);
            }
        }
        // thread 0: settle the run carried over from earlier tiles, note the chunk's lead
        if (t == 0) {
            const int first = s_first;
            if (s_lead < 0 && first < kTile) s_lead = (ta - a0) + first;
            if (s_pend_pos >= 0) {
                const int64_t end = ta + ((first < kTile) ? first : tv);
                const bool decided = (first < kTile) || (end - s_pend_pos > p.tol);
                if (decided) {
                    if (end - s_pend_pos > p.tol) {               // stable
                        if (s_pend_state != s_carry_last) {       // accepted (or the chunk's tentative first)
                            if (s_count == 0) s_first_state = s_pend_state;
                            slab[s_count] = rec_make(s_pend_pos + p.pos_base, s_pend_state);
                            s_last_pos = (unsigned long long)(s_pend_pos + p.pos_base);
                            s_count += 1;
                        }
                        s_carry_last = s_pend_state;
                    }
                    s_pend_pos = -1;
                }
            }
        }
        // per-thread: state of my last stable run
        const bool has = stable != 0;
        uint32_t my_last = 0xFFFFu;
        if (has) my_last = s_state[16 + kSpan * t + 31 - __builtin_clz(stable)];
        const unsigned long long hm = __ballot(has);
        if (hm && lane == 63 - __builtin_clzll(hm)) s_wave_last[wave] = my_last;
        if (hm == 0 && lane == 0) s_wave_last[wave] = 0xFFFFu;
        __syncthreads();   // C: s_wave_last, carries settled

        // state of the stable run preceding my first one
        uint32_t carry;
        {
            const unsigned long long lower = hm & ((1ull << lane) - 1ull);
            const int src = lower ? 63 - __builtin_clzll(lower) : 0;
            const uint32_t from_lane = __shfl(my_last, src);
            if (lower) carry = from_lane;
            else {
                carry = s_carry_last;
                for (int w = 0; w < wave; ++w) if (s_wave_last[w] != 0xFFFFu) carry = s_wave_last[w];
            }
        }
        uint32_t accmask = 0;
        int cnt = 0;
        {
            uint32_t m = stable, prev = carry;
            while (m) {
                const int lo = __builtin_ctz(m);
                const uint32_t st = s_state[16 + kSpan * t + lo];
                if (st != prev) { accmask |= 1u << lo; ++cnt; }
                prev = st;
                m &= m - 1;
            }
        }
        const int incl = wave_incl_scan(cnt, lane);
        if (lane == 63) s_wave_cnt[wave] = incl;
        __syncthreads();   // D: per-wave counts
        {
            int base = s_count;
            for (int w = 0; w < wave; ++w) base += s_wave_cnt[w];
            int o = base + incl - cnt;
            uint32_t m = accmask;
            while (m) {
                const int lo = __builtin_ctz(m);
                const uint32_t st = s_state[16 + kSpan * t + lo];
                if (o == 0) s_first_state = st;
                slab[o++] = rec_make(p.pos_base + ta + kSpan * t + lo, st);
                m &= m - 1;
            }
            if (accmask) atomicMax(&s_last_pos, (unsigned long long)(p.pos_base + ta + kSpan * t + 31 - __builtin_clz(accmask)));
        }
        // thread 0 picks up what it needs from this tile's LDS state before the tile is released
        uint32_t t0_prev8 = 0, t0_pend_state = 0;
        int t0_newpend = -1;
        if (t == 0) {
            t0_prev8 = s_state[16 + tv - 1];
            t0_newpend = s_newpend;
            if (t0_newpend >= 0) t0_pend_state = s_state[16 + t0_newpend];
        }
        __syncthreads();   // E: all reads of this tile's LDS state done; next tile may overwrite it
        if (t == 0) {
            for (int w = 0; w < kWaves; ++w) { s_count += s_wave_cnt[w]; if (s_wave_last[w] != 0xFFFFu) s_carry_last = s_wave_last[w]; }
            s_prev_state8 = t0_prev8;
            if (t0_newpend >= 0) { s_pend_pos = ta + t0_newpend; s_pend_state = t0_pend_state; }
        }
    }

    if (t == 0) {
        ChunkInfo ci;
        ci.pend_pos = (s_pend_pos >= 0) ? s_pend_pos + p.pos_base : -1;
        ci.start = p.pos_base + a0;
        ci.len = a1 - a0;
        ci.lead = (s_lead < 0) ? (a1 - a0) : s_lead;
        ci.cnt = s_count;
        ci.first_state = (uint16_t)s_first_state;
        ci.last_state = (uint16_t)s_carry_last;
        ci.pend_state = (uint16_t)s_pend_state;
        ci.last_pos = (int64_t)s_last_pos;
        const uint32_t init = chunk_init_state<SRC, DT, MOD, ORDER2>(p, chunk);
        ci.init_state = (uint16_t)init;
        ci.first_acc = 0; ci.pend_acc = 0; ci.pend_stable = 0; ci.pad = 0;
        p.chunks[chunk] = ci;
    }
}

// -----------------------------------------------------------------------------------------------------
// k_demod_runs_bp<SRC, DT, MOD, WRITE_QAD> -- the hot kernel for modulation order 2 (2-FSK, OOK, message
// segmentation: the states are {PAUSE, 1, 2}), whole rows only, tolerance <= kBpMaxTol, chunks of at most 64 rows.
//
// Same contract as k_demod_runs (qad, slab records, ChunkInfo) with the run phase done on BIT PLANES instead of state
// bytes in LDS: the per-sample states of a row are two compare masks per sample parity (v_cmp writes them as 64-bit
// wavefront masks: B = "q <= threshold", P = "q == NOISE", lane t <-> samples 2t / 2t+1), parked in lane r of four
// registers for row r.  After the chunk's last row, lane r analyses row r with 64-bit logic:
//   boundary masks  D_even = B_e ^ (B_o << 1 | carry), D_odd = B_e ^ B_o            (either plane)
//   stable runs     a boundary is the start of a stable run iff no boundary follows within `tol` samples: OR of
//                   shifted copies of the 128-bit (this row : next row) boundary masks, log2(tol) steps
//   accepted runs   walk the (few) stable bits of the row, compare with the state of the stable run before it
//                   (ballot + shuffle over rows), wave prefix sum of the counts, write the records
// No LDS, no per-boundary loops over the noise-induced (unstable) boundaries, ~6 VALU instructions per row instead
// of ~45, and the state classification is the v_cmp itself.
// -----------------------------------------------------------------------------------------------------
constexpr int kBpMaxTol = 64;
constexpr int kBpMaxRows = 64;

struct M128 { uint64_t lo, hi; };
__device__ __forceinline__ M128 m_shr(M128 x, int s) { return M128{(x.lo >> s) | (x.hi << (64 - s)), x.hi >> s}; }   // 0 < s < 64
__device__ __forceinline__ M128 m_or(M128 a, M128 b) { return M128{a.lo | b.lo, a.hi | b.hi}; }
// OR of (x >> j) for j in [0, k), 1 <= k <= 64 (wavefront-uniform k)
__device__ __forceinline__ M128 m_smear(M128 x, int k) {
    int have = 1;
    while (2 * have <= k) { x = m_or(x, m_shr(x, have)); have *= 2; }
    if (have < k) x = m_or(x, m_shr(x, k - have));
    return x;
}
__device__ __forceinline__ int bp_first(uint64_t e, uint64_t o) {     // lowest sample offset set in (even, odd) masks; 256 if none
    const int pe = e ? 2 * __builtin_ctzll(e) : 256, po = o ? 2 * __builtin_ctzll(o) + 1 : 256;
    return pe < po ? pe : po;
}
__device__ __forceinline__ int bp_last(uint64_t e, uint64_t o) {      // highest sample offset set; -1 if none
    const int pe = e ? 2 * (63 - __builtin_clzll(e)) : -1, po = o ? 2 * (63 - __builtin_clzll(o)) + 1 : -1;
    return pe > po ? pe : po;
}

// value -> lane `row` of a register that holds one word per row (value and row are wavefront-uniform): v_writelane_b32.
// This clang has no __builtin_amdgcn_writelane; the LLVM intrinsic is reached through its assembler name.
extern "C" __device__ int urh_llvm_writelane_i32(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ uint32_t put_lane(uint32_t value, int row, uint32_t old) {
    return (uint32_t)urh_llvm_writelane_i32((int)value, row, (int)old);
}

// RUNS = false: demodulation only (urhgpu_afp_demod[_dev], Signal.qad): the same streaming structure without the planes
// NPL: bit planes of the state besides PAUSE: 1 for modulation order 2 (plane = "q <= threshold": state 1, else 2), 2 for
// order 4 (planes = the two bits of state - 1, from the three threshold masks).
template <int SRC, int MOD> constexpr int bp_waves() { return (SRC == SRC_IQ && MOD == URHGPU_MOD_ASK) ? URH_ASK_WPB : URH_WPB; }

// The integer instantiations of the 2-FSK kernel come out at 76-77 VGPRs (the conversions' temporaries): six wavefronts per SIMD where the
// complex64 kernel has seven.  Held to seven they spill two registers in the rolled-up general atan2f path only (tools/kstats.sh + objdump:
// one scratch store / load pair inside fsk_row_general's loop, none in the streaming loop).  -DURH_INT_WAVES7=0: the compiler's choice.
#ifndef URH_HOT_WAVES
#define URH_HOT_WAVES 7       // wavefronts per SIMD the 2-FSK bit-plane kernel is held to (A/B: 8 = 64 VGPRs)
#endif
#ifndef URH_HOT_WAVES_INT
#define URH_HOT_WAVES_INT 8   // ... and its integer instantiations (64 VGPRs: the fast loop does not spill; measured 0.264 -> 0.259 ms per pipelined int8 step)
#endif
// (WIDEI: the integer instantiation WITH the wide loop -- seven wavefronts per SIMD as the float32 one: at eight its fast loop spills beside it)
template <int DT, bool WIDEI = false> constexpr int bp_hot_waves() { return (DT == URHGPU_DT_F32 || WIDEI) ? URH_HOT_WAVES : URH_HOT_WAVES_INT; }
#ifndef URH_INT_WAVES7
#define URH_INT_WAVES7 1
#endif
template <int SRC, int DT, int MOD, bool RUNS, int NPL> constexpr bool bp_seven() {
    return URH_INT_WAVES7 && SRC == SRC_IQ && MOD == URHGPU_MOD_FSK && RUNS && NPL == 1;
}
template <int SRC, int DT, int MOD, bool WRITE_QAD, bool RUNS = true, int NPL = 1, bool STAMPS = false, bool WIDEI = false>
__global__ __launch_bounds__((kBlock * bp_waves<SRC, MOD>())) __attribute__((amdgpu_waves_per_eu((STAMPS || bp_seven<SRC, DT, MOD, RUNS, NPL>()) ? bp_hot_waves<DT, WIDEI>() : 1, (STAMPS || bp_seven<SRC, DT, MOD, RUNS, NPL>()) ? bp_hot_waves<DT, WIDEI>() : 8)))
void k_demod_runs_bp(const RunArgs p) {
    // One workgroup per chunk, URH_WPB wavefronts: wavefront w streams the w-th share of the chunk's rows on its own
    // (no barrier inside the streaming phase), so that the wavefronts resident on the chip cover a NARROW window of
    // the capture (DRAM page locality: a wavefront per 16 KiB measured 8 % faster than a wavefront per 64 KiB on a
    // pure copy of this shape) while the per-chunk work (prologue, run phase, ChunkInfo) is paid once per 64 rows.
    constexpr int W = bp_waves<SRC, MOD>();
    __shared__ uint32_t s_planes[W > 1 ? (NPL + 1) * 4 : 1][W > 1 ? 64 : 1];
    int lane = threadIdx.x & 63;
    const int w = (W > 1) ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;     // wavefront-uniform
    const int64_t chunk = p.chunk_base + blockIdx.x;
    int64_t a0 = p.range_begin + (int64_t)blockIdx.x * p.chunk_len;
    int64_t a1 = (a0 + p.chunk_len < p.range_end) ? a0 + p.chunk_len : p.range_end;
    if (p.graded_from > 0 && chunk >= p.graded_from) {        // graded tail (RunArgs::graded_from): the launch's last chunks are short
        a0 = p.graded_from * p.chunk_len + (chunk - p.graded_from) * p.graded_len;
        a1 = (a0 + p.graded_len < p.range_end) ? a0 + p.graded_len : p.range_end;
    }
    const int nr = (int)((a1 - a0) / kRowSamples);            // whole rows in this chunk: a multiple of W, at most 64
    const int R = nr / W;                                     // rows of this wavefront: [w R, w R + R)
    const int r0 = w * R;
    uint64_t *slab = RUNS ? p.slab + chunk * p.slab_stride : nullptr;
    // STAMPS (probe only, tools/boundary_probe.py): wavefront 0 leaves three s_memrealtime stamps (10 ns units, low 32 bits: entry, streaming
    // phase over, ChunkInfo written) and its hardware ids in the ChunkInfo fields the resolve kernels fill in later
    // (stored at once, re-read for the ChunkInfo store: a value kept in a register across the streaming loop costs this kernel its seventh wavefront)
    if (STAMPS && w == 0 && lane == 0) p.chunks[chunk].first_acc = (int32_t)(uint32_t)wall_clock64();
    const bool global_start = (p.left_halo == nullptr);
    const bool first_row = (a0 == 0) && global_start && (w == 0);   // this wavefront holds sample 0 of the capture

    // the first batch of rows is requested before the prologue's (dependent, wavefront-uniform) loads: their latency overlaps
    constexpr int kBatch = URH_KBATCH;
    constexpr bool kFskFast = URH_SPEC && SRC == SRC_IQ && MOD == URHGPU_MOD_FSK;
    constexpr bool kIntCapture = DT != URHGPU_DT_F32;
    // the wide loop (fsk_wide) for float32 captures, and for integer captures in an instantiation of its own (WIDEI: seven wavefronts per SIMD)
    // that the launcher takes when RunArgs::wide_int says so (capture streams probe their captures: k_wide_probe): the default integer
    // instantiations are held to 64 VGPRs (eight wavefronts per SIMD), where the FAST loop reloads spilled registers in every iteration with
    // the wide loop beside it (int8 step 0.269 -> 0.294 ms), and at seven the narrow capture loses 5 % (profiles/r06s_deviation_pmc.txt)
    constexpr bool kWideLoop = (!kIntCapture || WIDEI || URH_WIDE_INT) && !URH_NO_WIDE;
    RowIn cur[kBatch] = {}, nxt[kBatch];
    v4f cv[kBatch], nv[kBatch];                               // kFskFast: the rows as 4-vectors
    if (kFskFast) load_rows_v4<DT>(p, a0, r0, lane, cv);
    else load_rows_bp<SRC, DT>(p, a0, r0, lane, cur);
    // (Filling this table lazily, by the wavefront that needs it and without the barrier, was measured: 0.295 against 0.284 ms per K = 20 step --
    // the build spilled nine registers in the prologue, and the barrier is also what starts a chunk's four wavefronts together.)
    if (SRC == SRC_IQ && MOD == URHGPU_MOD_FSK) { atan_table_init(); __syncthreads(); }      // (the loads above are in flight)

    float prev_c = 0.f, prev_d = 0.f;                         // IQ sample before my first row (FSK seam operand)
    uint32_t st_before = kStNone;                             // state of sample a0-1: wavefront 0 (it runs phase 2)
    if (w == 0 && RUNS) st_before = chunk_prologue<SRC, DT, MOD, NPL == 1>(p, a0, global_start, prev_c, prev_d);
    else if (SRC == SRC_IQ && MOD == URHGPU_MOD_FSK) {
        const int64_t before = a0 + (int64_t)r0 * kRowSamples - 1;
        if (before >= 0) Iq<DT>::load1(p.in, before, prev_c, prev_d);
        else if (!global_start) Iq<DT>::load1(p.left_halo, 1, prev_c, prev_d);
    }

    // ================= phase 1: demodulate, one compare mask per plane and parity, parked in lane `row` ==============
    uint32_t spec_hint = 0;                                    // demod_batch: batches left that skip the speculative pass
    uint32_t pl[NPL + 1][2][2] = {};                          // [state planes ..., PAUSE][even, odd samples][low, high word]: lane r <- row r
    // what follows the demodulation of a batch: qad stores (wavefront-uniform row base + the lane's 32-bit offset: no VALU address
    // arithmetic), the compare masks, their parking
    auto emit = [&](const RowIn (&rows)[kBatch], float (&q0)[kBatch], float (&q1)[kBatch], const uint32_t gated, const int rb, const bool may_row0) {
        float *const qrow = WRITE_QAD ? p.qad + a0 + (int64_t)rb * kRowSamples : nullptr;     // wavefront-uniform
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const bool row0 = may_row0 && (j == 0) && (rb == 0) && first_row;
            if (SRC == SRC_IQ) {
                if (row0 && lane == 0 && !p.seg_mode) q0[0] = p.noise_val;            // result[0] = NOISE (:361)
                if (WRITE_QAD) {
                    float w0 = q0[j], w1 = q1[j];
                    if (MOD == URHGPU_MOD_ASK && DT == URHGPU_DT_F32 && p.seg_mode) {
                        // the segmentation pass leaves the ASK demodulation of the same samples (signal_functions.pyx:343-378, ASK branch,
                        // with the demodulation's own constants; result[0] = NOISE): the estimator needs both and this one has the samples
                        const float m0 = rows[j].c0 * rows[j].c0 + rows[j].d0 * rows[j].d0, m1 = rows[j].c1 * rows[j].c1 + rows[j].d1 * rows[j].d1;
                        w0 = (m0 <= p.dm_noise_sqrd) ? p.dm_noise_val : __builtin_sqrtf(m0) / p.dm_max_magnitude;
                        w1 = (m1 <= p.dm_noise_sqrd) ? p.dm_noise_val : __builtin_sqrtf(m1) / p.dm_max_magnitude;
                        if (row0 && lane == 0) w0 = p.dm_noise_val;
                    }
                    typedef float v2s __attribute__((ext_vector_type(2)));
                    const v2s qq = {w0, w1};
                    v2s *const dst = (v2s *)((char *)(qrow + j * kRowSamples) + (uint32_t)lane * 8u);
#if URH_NT
                    __builtin_nontemporal_store(qq, dst);
#else
                    *dst = qq;
#endif
                }
            }
            if (!RUNS) continue;
            uint64_t X[NPL][2];                                // [plane][parity]
            if (NPL == 1) {
                X[0][0] = __builtin_amdgcn_ballot_w64(q0[j] <= p.thr[0]); X[0][1] = __builtin_amdgcn_ballot_w64(q1[j] <= p.thr[0]);
            } else {
                // order 4: state - 1 = the first k with q <= thr[k], else 3 (signal_functions.pyx:438-442), as two bits
                const float qq[2] = {q0[j], q1[j]};
#pragma unroll
                for (int par = 0; par < 2; ++par) {
                    const uint64_t m0 = __builtin_amdgcn_ballot_w64(qq[par] <= p.thr[0]), m1 = __builtin_amdgcn_ballot_w64(qq[par] <= p.thr[1]),
                                   m2 = __builtin_amdgcn_ballot_w64(qq[par] <= p.thr[2]);
                    X[NPL - 1][par] = ~m0 & ~m1;               // bit 1
                    X[0][par] = ~m0 & (m1 | ~m2);              // bit 0
                }
            }
            const int row = rb + j;
            if (((gated >> j) & 1u) || row0) {                 // wavefront-uniform: some sample may be the NOISE sentinel
                const uint64_t Pe = __builtin_amdgcn_ballot_w64(q0[j] == p.noise_val), Po = __builtin_amdgcn_ballot_w64(q1[j] == p.noise_val);
#pragma unroll
                for (int k = 0; k < NPL; ++k) { X[k][0] &= ~Pe; X[k][1] &= ~Po; }
                pl[NPL][0][0] = put_lane((uint32_t)Pe, row, pl[NPL][0][0]); pl[NPL][0][1] = put_lane((uint32_t)(Pe >> 32), row, pl[NPL][0][1]);
                pl[NPL][1][0] = put_lane((uint32_t)Po, row, pl[NPL][1][0]); pl[NPL][1][1] = put_lane((uint32_t)(Po >> 32), row, pl[NPL][1][1]);
            }
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
#pragma unroll
                for (int par = 0; par < 2; ++par) {
                    pl[k][par][0] = put_lane((uint32_t)X[k][par], row, pl[k][par][0]);
                    pl[k][par][1] = put_lane((uint32_t)(X[k][par] >> 32), row, pl[k][par][1]);
                }
            }
        }
    };
    // complex64 FSK: the batch-level fast path (fsk_front / fsk_divide / atanf_poly2) as a tight inner loop of its own -- nothing of the
    // per-row machinery (hint forms, re-dos, the general atan2f) shares its registers or its instruction-cache lines; a flagged batch
    // -- rare -- leaves it for ONE step of the generic body (demod_batch), which may set the hint that keeps the following batches there.
    // The capture's very first batch (result[0] = NOISE) is a generic step too.
    const int r_end = r0 + R;
    if (kFskFast) {
        int rb = r0;
        bool wide = false, have_nv = false, lut_ready = false;      // wavefront-uniform: the wide loop is on; nv holds batch rb + kBatch; this wavefront has filled the range table
        // exactly zero cross products of integer samples (products and their difference are exact): atan2f(+-0, re > 0) = +-0 and
        // atan2f(+-0, re < 0) = +-pi (e_atan2f.c: y = 0 -> y, pi + tiny, -pi - tiny), the sign being that of the reference's product
        // (conj_mul: its zeros are signed differently from the plain product's); backward: only the wide loop meets re < 0
        auto settle_zeros = [&](const FskFront<kBatch> &f, float (&q0)[kBatch], float (&q1)[kBatch], const bool backward) {
            uint64_t zany = 0;
#pragma unroll
            for (int j = 0; j < kBatch; ++j) zany |= __builtin_amdgcn_ballot_w64(f.im[j].x == 0.0f) | __builtin_amdgcn_ballot_w64(f.im[j].y == 0.0f);
            if (zany == 0) return;
            float sc = prev_c, sd = prev_d;
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                const float pc = dpp_wave_shr1(cv[j].z, sc), pd = dpp_wave_shr1(cv[j].w, sd);
                float re_r, im_r;
                conj_mul(pc, pd, cv[j].x, cv[j].y, re_r, im_r);
                if (backward && f.re[j].x < 0.0f) im_r = __uint_as_float((__float_as_uint(im_r) & 0x80000000u) | 0x40490fdbu);      // +-pi
                q0[j] = (f.im[j].x == 0.0f) ? im_r : q0[j];
                conj_mul(cv[j].x, cv[j].y, cv[j].z, cv[j].w, re_r, im_r);
                if (backward && f.re[j].y < 0.0f) im_r = __uint_as_float((__float_as_uint(im_r) & 0x80000000u) | 0x40490fdbu);
                q1[j] = (f.im[j].y == 0.0f) ? im_r : q1[j];
                sc = lane63(cv[j].z); sd = lane63(cv[j].w);
            }
        };
        for (;;) {
#pragma unroll 1
            while (rb < r_end && spec_hint == 0 && !wide && !(first_row && rb == 0)) {
                if (rb + kBatch < r_end) load_rows_v4<DT>(p, a0, rb + kBatch, lane, nv);      // (have_nv is false on every way into this loop)
                have_nv = true;
                FskFront<kBatch> f;
                float nc = prev_c, nd = prev_d;
                fsk_front<kBatch>(cv, nc, nd, f);
                v2f t[kBatch], z[kBatch];
                if (__builtin_expect((fsk_divide<kBatch, kIntCapture>(f, p, t, z)), 0)) break;          // cv still holds batch rb
                float q0[kBatch], q1[kBatch];
#pragma unroll
                for (int j = 0; j < kBatch; ++j) { const v2f q = t[j] - atanf_poly2(t[j], z[j]); q0[j] = q.x; q1[j] = q.y; }
                if (kIntCapture) settle_zeros(f, q0, q1, false);
                prev_c = nc; prev_d = nd;
                emit(cur, q0, q1, 0u, rb, false);          // (`cur`: read by the float32 segmentation pass only, not this instantiation)
#pragma unroll
                for (int j = 0; j < kBatch; ++j) cv[j] = nv[j];
                have_nv = false;
                rb += kBatch;
            }
            if (rb >= r_end) break;
            // The wide loop (fsk_wide): entered by a batch the fast loop flagged; left for the fast loop by the FIRST batch that one would
            // have taken -- an isolated outlier (the headline capture has one batch in 120) costs its own batch and the next one here,
            // about what the generic step it used to take cost (staying for four such batches cost that capture 2.5 %: the wavefronts that
            // met an outlier ran the rest of their rows at 104 instead of 68 instructions and held their workgroups up); left for ONE
            // generic step (and whatever its hint says) by a batch outside the wide window.
            // (The fast loop's flagged exit carries no statement of its own: with one, the compiler splits the loop's body at the flag
            // and the division's tail no longer interleaves with the polynomial.  The state is kept in plain bools and the loop entered
            // unconditionally on a flag: the variations tried on this -- a counter of flags, one packed state word, a hint shared by the
            // context's passes through memory so that wavefronts start in the right loop -- made the register allocator spill the bit
            // planes, and the build with the shared hint stepped at 0.53 ms instead of 0.28: profiles/r06s_deviation_pmc.txt.)
            wide = kWideLoop && spec_hint == 0 && !(first_row && rb == 0);      // i.e. the fast loop ended on a flagged batch
            if (kWideLoop && wide) {
                bool outside = false;
                if (!lut_ready) { atan_lut_fill_wave(lane); lut_ready = true; }
#pragma unroll 1
                while (rb < r_end) {
                    if (!have_nv && rb + kBatch < r_end) load_rows_v4<DT>(p, a0, rb + kBatch, lane, nv);
                    have_nv = true;
                    FskFront<kBatch> f;
                    float nc = prev_c, nd = prev_d;
                    fsk_front<kBatch>(cv, nc, nd, f);
                    float q0[kBatch], q1[kBatch];
                    const int kind = fsk_wide<kBatch, kIntCapture>(f, p, q0, q1);
                    if (__builtin_expect(kind == 1, 0)) { outside = true; break; }
                    if (kIntCapture) settle_zeros(f, q0, q1, true);
                    prev_c = nc; prev_d = nd;
                    emit(cur, q0, q1, 0u, rb, false);
#pragma unroll
                    for (int j = 0; j < kBatch; ++j) cv[j] = nv[j];
                    have_nv = false;
                    rb += kBatch;
                    if (kind == 2) break;
                }
                wide = false;
                if (rb >= r_end) break;
                if (!outside) continue;                       // back to the fast loop
            }
            float q0[kBatch], q1[kBatch];
#pragma unroll
            for (int j = 0; j < kBatch; ++j) { cur[j].c0 = cv[j].x; cur[j].d0 = cv[j].y; cur[j].c1 = cv[j].z; cur[j].d1 = cv[j].w; }
            if (!have_nv && rb + kBatch < r_end) load_rows_v4<DT>(p, a0, rb + kBatch, lane, nv);
            const uint32_t gated = demod_batch<SRC, DT, MOD, kBatch>(cur, prev_c, prev_d, p, q0, q1, spec_hint);
            emit(cur, q0, q1, gated, rb, true);
#pragma unroll
            for (int j = 0; j < kBatch; ++j) cv[j] = nv[j];
            have_nv = false;
            rb += kBatch;
        }
    } else {
#pragma unroll 1
        for (int rb = r0; rb < r_end; rb += kBatch) {
            float q0[kBatch], q1[kBatch];
            if (rb + kBatch < r_end) load_rows_bp<SRC, DT>(p, a0, rb + kBatch, lane, nxt);
            const uint32_t gated = demod_batch<SRC, DT, MOD, kBatch>(cur, prev_c, prev_d, p, q0, q1, spec_hint);
            emit(cur, q0, q1, gated, rb, true);
#pragma unroll
            for (int j = 0; j < kBatch; ++j) cur[j] = nxt[j];
        }
    }

    if (!RUNS) return;
    // (the lane number again, from the execution mask's bit count: kept across the streaming loops it is one register too many -- spilled)
    lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if (STAMPS && w == 0 && lane == 0) p.chunks[chunk].pend_stable = (int32_t)(uint32_t)wall_clock64();
    // the chunk's planes come together in wavefront 0 (lane r <- row r)
    if (W > 1) {
        if (w != 0) {
            if (lane >= r0 && lane < r0 + R) {
#pragma unroll
                for (int k = 0; k <= NPL; ++k) {
                    s_planes[4 * k + 0][lane] = pl[k][0][0]; s_planes[4 * k + 1][lane] = pl[k][0][1];
                    s_planes[4 * k + 2][lane] = pl[k][1][0]; s_planes[4 * k + 3][lane] = pl[k][1][1];
                }
            }
        }
        __syncthreads();
        if (w != 0) return;
        if (lane >= R) {
#pragma unroll
            for (int k = 0; k <= NPL; ++k) {
                pl[k][0][0] = s_planes[4 * k + 0][lane]; pl[k][0][1] = s_planes[4 * k + 1][lane];
                pl[k][1][0] = s_planes[4 * k + 2][lane]; pl[k][1][1] = s_planes[4 * k + 3][lane];
            }
        }
    }

    // ================= phase 2: lane r owns row r (samples [128 r, 128 r + 128) of the chunk) =====================
    uint64_t XE[NPL + 1], XO[NPL + 1];                         // planes of my row: even / odd samples
#pragma unroll
    for (int k = 0; k <= NPL; ++k) { XE[k] = ((uint64_t)pl[k][0][1] << 32) | pl[k][0][0]; XO[k] = ((uint64_t)pl[k][1][1] << 32) | pl[k][1][0]; }
    auto state_at = [&](int pos) -> uint32_t {                 // state byte of sample `pos` of my row
        const int idx = pos >> 1;
        if ((((pos & 1) ? XO[NPL] : XE[NPL]) >> idx) & 1) return kStPause;
        const uint32_t b0 = (uint32_t)((((pos & 1) ? XO[0] : XE[0]) >> idx) & 1);
        if (NPL == 1) return b0 ? 1u : 2u;
        const uint32_t b1 = (uint32_t)((((pos & 1) ? XO[NPL - 1] : XE[NPL - 1]) >> idx) & 1);
        return 1u + b0 + 2u * b1;
    };
    // boundaries: sample differs from the one before it (in any plane)
    uint64_t De = 0, Do = 0;
    {
#pragma unroll
        for (int k = 0; k <= NPL; ++k) {
            uint32_t up = __shfl_up(pl[k][1][1], 1) >> 31;     // last sample of the row above
            if (lane == 0) {
                if (k == NPL) up = (st_before == kStPause);
                else if (NPL == 1) up = (st_before == 1u);
                else up = (st_before >= 1u && st_before <= 4u) ? (((st_before - 1u) >> k) & 1u) : 0u;
            }
            De |= XE[k] ^ ((XO[k] << 1) | up);
            Do |= XE[k] ^ XO[k];
        }
        if (lane == 0 && st_before == kStNone) De |= 1ull;     // the capture starts here: its first sample starts a run
        if (lane >= nr) { De = 0; Do = 0; }
    }
    // stable runs: no boundary within the next `tol` samples (the chunk end counts as one: what it cuts short is
    // either truly unstable or the chunk's pending run)
    uint64_t Se = De, So = Do;
    if (p.tol > 0) {
        uint64_t Ne = __shfl_down((unsigned long long)De, 1), No = __shfl_down((unsigned long long)Do, 1);
        if (lane >= nr - 1) { Ne = (lane == nr - 1) ? 1ull : 0ull; No = 0; }
        const int ke = p.tol >> 1, ko = (p.tol + 1) >> 1;     // samples p+1..p+tol: ke / ko of each parity
        const M128 Xe{De, Ne}, Xo{Do, No};
        M128 e_ke{0, 0}, o_ke{0, 0}, e_ko = Xe, o_ko = Xo;
        if (ke > 0) {
            e_ke = m_smear(Xe, ke); o_ke = m_smear(Xo, ke);
            e_ko = e_ke; o_ko = o_ke;
            if (ko > ke) { e_ko = m_or(e_ke, m_shr(Xe, ke)); o_ko = m_or(o_ke, m_shr(Xo, ke)); }
        }
        const uint64_t near_e = o_ko.lo | m_shr(e_ke, 1).lo;           // after an even sample: odd idx t..t+ko-1, even idx t+1..t+ke
        const uint64_t near_o = m_shr(e_ko, 1).lo | m_shr(o_ke, 1).lo; // after an odd sample: even idx t+1..t+ko, odd idx t+1..t+ke
        Se = De & ~near_e; So = Do & ~near_o;
    }
    // chunk-level facts: lead (offset of the first boundary), the pending run (last boundary within tol of the end)
    int64_t lead = a1 - a0;
    {
        const uint64_t bm = __builtin_amdgcn_ballot_w64((De | Do) != 0);
        if (bm) {
            const int fl = __builtin_ctzll(bm);
            lead = (int64_t)fl * kRowSamples + __builtin_amdgcn_readlane(bp_first(De, Do), fl);
        }
    }
    int64_t pend_pos = -1; uint32_t pend_state = 0;
    {
        const int lp = bp_last(De, Do);
        const uint32_t ls = state_at(lp < 0 ? 0 : lp);
        const int lpu = __builtin_amdgcn_readlane(lp, nr - 1);
        const uint32_t lsu = __builtin_amdgcn_readlane(ls, nr - 1);
        if (lpu >= 0 && kRowSamples - lpu <= p.tol) { pend_pos = a0 + (int64_t)(nr - 1) * kRowSamples + lpu; pend_state = lsu; }
    }
    // accepted runs: stable runs whose state differs from the stable run before them
    const bool has = (Se | So) != 0;
    const int hp = bp_last(Se, So);
    const uint32_t my_last = has ? state_at(hp) : 0xFFFFu;
    const uint64_t hm = __builtin_amdgcn_ballot_w64(has);
    uint32_t prev;
    {
        const uint64_t lower = hm & ((1ull << lane) - 1ull);
        const int src = lower ? 63 - __builtin_clzll(lower) : 0;
        const uint32_t from_lane = __shfl(my_last, src);
        prev = lower ? from_lane : 0xFFFFu;
    }
    uint64_t ae = 0, ao = 0;
    int cnt = 0, last_acc = -1;
    uint32_t first_acc_state = 0;
    {
        uint64_t se = Se, so = So;
        while (se | so) {
            const int pos = bp_first(se, so);
            const uint32_t st = state_at(pos);
            if (st != prev) {
                if (cnt == 0) first_acc_state = st;
                if (pos & 1) ao |= 1ull << (pos >> 1); else ae |= 1ull << (pos >> 1);
                ++cnt; last_acc = pos;
            }
            prev = st;
            if (pos & 1) so &= so - 1; else se &= se - 1;
        }
    }
    const int incl = wave_incl_scan(cnt, lane);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    // streamed pass (RunArgs::progress): the records go THROUGH to memory (agent-scope stores: no dirty line stays in this XCD's L2), so
    // that an acknowledged store is visible to the tail kernels of this chunk's segment, which start while this kernel is still running
    const bool through = (p.progress != nullptr);
    {
        int o = incl - cnt;
        const int64_t base = p.pos_base + a0 + (int64_t)lane * kRowSamples;
        while (ae | ao) {
            const int pos = bp_first(ae, ao);
            const uint64_t rec = rec_make(base + pos, state_at(pos));
            if (through) __hip_atomic_store(slab + o, rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else slab[o] = rec;
            ++o;
            if (pos & 1) ao &= ao - 1; else ae &= ae - 1;
        }
    }
    const uint64_t am = __builtin_amdgcn_ballot_w64(cnt > 0);
    uint32_t first_state = 0xFFFFu; int64_t last_pos = 0;
    if (am) {
        const int fl = __builtin_ctzll(am), ll = 63 - __builtin_clzll(am);
        first_state = __builtin_amdgcn_readlane(first_acc_state, fl);
        last_pos = p.pos_base + a0 + (int64_t)ll * kRowSamples + __builtin_amdgcn_readlane(last_acc, ll);
    }
    const uint32_t last_state = hm ? (uint32_t)__builtin_amdgcn_readlane(my_last, 63 - __builtin_clzll(hm)) : 0xFFFFu;

    if (lane == 0) {
        ChunkInfo ci;
        ci.pend_pos = (pend_pos >= 0) ? pend_pos + p.pos_base : -1;
        ci.start = p.pos_base + a0;
        ci.len = a1 - a0;
        ci.lead = lead;
        ci.cnt = total;
        ci.first_state = (uint16_t)first_state;
        ci.last_state = (uint16_t)last_state;
        ci.pend_state = (uint16_t)pend_state;
        ci.last_pos = last_pos;
        ci.init_state = (uint16_t)chunk_init_state<SRC, DT, MOD, NPL == 1>(p, chunk);
        ci.first_acc = 0; ci.pend_acc = 0; ci.pend_stable = 0; ci.pad = 0;
        if (STAMPS) {
            // gfx9 HW_REG_HW_ID: wave / SIMD / CU / SE ids; the XCC id is its own register (HW_REG_XCC_ID = 20)
            const unsigned hw = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
            const unsigned xcc = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (3 << 11));
            ci.first_acc = p.chunks[chunk].first_acc; ci.pend_stable = p.chunks[chunk].pend_stable; ci.pad = (int32_t)(uint32_t)wall_clock64();
            ci.pend_acc = (int32_t)((hw & 0xFFFFu) | ((xcc & 0xFu) << 16));
        }
        if (through) {
            static_assert(sizeof(ChunkInfo) % 8 == 0, "ChunkInfo is written as 64-bit words");
            uint64_t w[sizeof(ChunkInfo) / 8];
            __builtin_memcpy(w, &ci, sizeof(ChunkInfo));
            uint64_t *dst = (uint64_t *)(p.chunks + chunk);
#pragma unroll
            for (int k = 0; k < (int)(sizeof(ChunkInfo) / 8); ++k) __hip_atomic_store(dst + k, w[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            p.chunks[chunk] = ci;
        }
    }
    if (through) {
        // every lane's write-through stores acknowledged (gfx9: vmcnt counts stores too), THEN the chunk counts itself into its segment:
        // what the LLVM memory model does for a release at agent scope, minus the L2 write-back that ordinary (cached) stores would need
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            int k = 0;
            while (k < p.n_seg - 1 && chunk >= (int64_t)p.seg_end[k]) ++k;
            __hip_atomic_fetch_add(p.progress + k * kProgressStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// -----------------------------------------------------------------------------------------------------
// Plain afp_demod kernel (no run segmentation): used by urhgpu_afp_demod[_dev] for MOD_OTHER and
// when only Signal.qad is wanted.
// -----------------------------------------------------------------------------------------------------
constexpr int kAfpBlock = 256;                  // k_afp_demod: 4 wavefronts, 512 samples per workgroup-wide load
constexpr int kAfpRow = kAfpBlock * 2;

template <int DT, int MOD>
__global__ __launch_bounds__(kAfpBlock) void k_afp_demod(const RunArgs p) {
    const int lane = threadIdx.x & 63;
    const bool global_start = (p.left_halo == nullptr);
    const int64_t stride = (int64_t)gridDim.x * kAfpRow;
    for (int64_t base = (int64_t)blockIdx.x * kAfpRow; base < p.n; base += stride) {
        const int64_t i0 = base + 2 * threadIdx.x;
        float c0 = 0, d0 = 0, c1 = 0, d1 = 0;
        const bool v1 = (i0 + 1 < p.n), v0 = (i0 < p.n);
        if (v1) Iq<DT>::load2(p.in, i0, c0, d0, c1, d1);
        else if (v0) Iq<DT>::load1(p.in, i0, c0, d0);
        float sc = 0, sd = 0;
        if (MOD == URHGPU_MOD_FSK) {
            const int64_t w0 = base + 2 * (threadIdx.x - lane);      // wavefront seam (uniform address)
            if (w0 >= 1) { if (w0 - 1 < p.n) Iq<DT>::load1(p.in, w0 - 1, sc, sd); }
            else if (!global_start) Iq<DT>::load1(p.left_halo, 1, sc, sd);
        }
        const float pc = dpp_wave_shr1(c1, sc), pd = dpp_wave_shr1(d1, sd);
        if (v0) {
            float q0 = demod_one<MOD>(pc, pd, c0, d0, p);
            if (i0 == 0 && global_start) q0 = p.noise_val;
            if (v1) {
                const float q1 = demod_one<MOD>(c0, d0, c1, d1, p);
                *(float2 *)(p.qad + i0) = make_float2(q0, q1);
            } else {
                p.qad[i0] = q0;
            }
        }
    }
}
// Test hook: out[i] = number of (n, d) pairs (d in the fast-path range, 2^-31 <= |n/d| < 2^26) on which div_fast differs from
// the IEEE division, over `reps` pseudo-random pairs per thread.
__global__ void k_test_div(uint64_t seed, int reps, unsigned long long *mismatches) {
    uint64_t s = seed + (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) * 0x9e3779b97f4a7c15ull;
    unsigned long long bad = 0;
    for (int i = 0; i < reps; ++i) {
        s += 0x9e3779b97f4a7c15ull;
        uint64_t z = s; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; z ^= z >> 31;
        const uint32_t a = (uint32_t)z, b = (uint32_t)(z >> 32);
        // d: positive, exponent in [-40, 40); n: |d| scaled by +-[0.5, 1) and a random power of two from 2^-30 to 2^26 (the extended
        // path divides by |re| whatever the quotient: phase steps beyond pi/4 have |im/re| >= 1)
        const float d = __uint_as_float(((b % 80u + 87u) << 23) | (b >> 9));
        const float f = __uint_as_float((a & 0x807fffffu) | 0x3f000000u);            // +-[0.5, 1)
        const float n = f * d * __uint_as_float((97u + (a >> 23) % 57u) << 23);
        float q2, q3;
        div_fast2(n, d, d * 0.25f, n == 0.f ? 1.0f : n * 8.0f, q2, q3);   // second slot: another in-range pair when |n*8| is
        const float q1 = n / d;
        if (__float_as_uint(q1) != __float_as_uint(q2)) ++bad;
        if (__float_as_uint(div_fast(n, d)) != __float_as_uint(q1)) ++bad;
        (void)q3;
    }
    if (bad) atomicAdd(mismatches, bad);
}

__global__ void k_test_atan2f(const float *y, const float *x, int64_t n, float *out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = atan2f_dev(y[i], x[i]);
}

// ---- host-side launchers ---------------------------------------------------------------------------
// test hook (urhgpu_test_force_state_bytes): route order-2 work through the state-byte kernel as well
bool g_force_state_bytes = false;
std::atomic<long long> g_wide_int_launches{0};   // urhgpu_test_wide_int_launches: hot launches that took the WIDEI instantiation
bool g_stamp_probe = false;          // test hook (urhgpu_test_hot_stamps): complex64 2-FSK passes run the STAMPS instantiation of the bit-plane kernel
thread_local HotEvents g_hot_events;

// `a` describes the whole capture (a.n samples, chunk table / slab for n_main + has_tail chunks):
// one launch over the whole tiles, one one-workgroup launch for the partial tile at the end.
template <int SRC, int DT, int MOD, bool O2, bool WQ>
static void launch_runs_4(RunArgs a, hipStream_t s) {
    // a.launch_part: 0 = every chunk; 1 = every chunk but the first (it alone needs the left halo of a sharded capture,
    // which may still be in flight); 2 = the first chunk only
    const int part = a.launch_part;
    if (g_stamp_probe) a.stamp_probe = 1;
    const int64_t n_full = (a.n / kTile) * kTile;
    const int64_t n_main = (n_full + a.chunk_len - 1) / a.chunk_len;
    int64_t c_lo = (part == 1) ? 1 : 0, c_hi = (part == 2) ? std::min<int64_t>(n_main, 1) : n_main;
    if (a.graded_from > 0 && part == 0 && a.launch_hi == 0) c_hi = n_main + (n_main - a.graded_from) * (a.chunk_len / a.graded_len - 1);   // graded tail: more, shorter chunks at the end
    const bool ranged = a.launch_hi > 0;                      // an explicit chunk range (streamed passes that upload piece by piece)
    if (ranged) { c_lo = std::min<int64_t>(a.launch_lo, n_main); c_hi = std::min<int64_t>(a.launch_hi, n_main); }
    if (c_hi > c_lo) {
        a.range_begin = c_lo * a.chunk_len; a.range_end = std::min<int64_t>(n_full, c_hi * a.chunk_len); a.chunk_base = c_lo;
        // orders 2 and 4: the bit-plane kernel; anything else: the state-byte kernel
        const bool planes_ok = URH_BITPLANE && a.tol <= kBpMaxTol && a.chunk_len <= (int64_t)kBpMaxRows * kRowSamples && !g_force_state_bytes;
        // (the STAMPS instantiation exists for complex64 2-FSK with qad only: a probe request on any other pass is an ordinary launch, with its
        // completion event -- ADVICE r5)
        constexpr bool stamps_ok = SRC == SRC_IQ && DT == URHGPU_DT_F32 && MOD == URHGPU_MOD_FSK && WQ;
        const bool stamps = a.stamp_probe && stamps_ok;
        // integer FSK captures with wide phase steps (RunArgs::wide_int): the instantiation with the wide loop
        constexpr bool widei_ok = SRC == SRC_IQ && (DT == URHGPU_DT_I8 || DT == URHGPU_DT_I16) && MOD == URHGPU_MOD_FSK;      // (unsigned samples are not centred: re > 0)
        if (widei_ok && a.wide_int && planes_ok && (O2 || a.order == 4)) {
            ++g_wide_int_launches;
            const bool ev = (g_hot_events.start || g_hot_events.stop) && !g_hot_events.used;
            if (O2 && ev) {
                hipExtLaunchKernelGGL((k_demod_runs_bp<SRC, DT, MOD, WQ, true, 1, false, widei_ok>), dim3((unsigned)(c_hi - c_lo)), dim3(kBlock * bp_waves<SRC, MOD>()),
                                      (size_t)a.lds_pad, s, g_hot_events.start, g_hot_events.stop, 0, a);
                g_hot_events.used = true;
            } else if (O2)
                hipLaunchKernelGGL((k_demod_runs_bp<SRC, DT, MOD, WQ, true, 1, false, widei_ok>), dim3((unsigned)(c_hi - c_lo)), dim3(kBlock * bp_waves<SRC, MOD>()), (size_t)a.lds_pad, s, a);
            else
                hipLaunchKernelGGL((k_demod_runs_bp<SRC, DT, MOD, WQ, true, 2, false, widei_ok>), dim3((unsigned)(c_hi - c_lo)), dim3(kBlock * bp_waves<SRC, MOD>()), (size_t)a.lds_pad, s, a);
        } else
        if (planes_ok && O2 && stamps && (g_hot_events.start || g_hot_events.stop) && !g_hot_events.used) {
            hipExtLaunchKernelGGL((k_demod_runs_bp<SRC_IQ, URHGPU_DT_F32, URHGPU_MOD_FSK, true, true, 1, true>), dim3((unsigned)(c_hi - c_lo)), dim3(kBlock * bp_waves<SRC, MOD>()),
                                  (size_t)a.lds_pad, s, g_hot_events.start, g_hot_events.stop, 0, a);
            g_hot_events.used = true;
        } else if (planes_ok && O2 && !stamps && (g_hot_events.start || g_hot_events.stop) && !g_hot_events.used) {
            hipExtLaunchKernelGGL((k_demod_runs_bp<SRC, DT, MOD, WQ>), dim3((unsigned)(c_hi - c_lo)), dim3(kBlock * bp_waves<SRC, MOD>()),
                                  (size_t)a.lds_pad, s, g_hot_events.start, g_hot_events.stop, 0, a);
            g_hot_events.used = true;
        } else if (planes_ok && O2 && stamps)
            hipLaunchKernelGGL((k_demod_runs_bp<SRC_IQ, URHGPU_DT_F32, URHGPU_MOD_FSK, true, true, 1, true>), dim3((unsigned)(c_hi - c_lo)), dim3(kBlock * bp_waves<SRC, MOD>()), (size_t)a.lds_pad, s, a);
        else if (planes_ok && O2)
            hipLaunchKernelGGL((k_demod_runs_bp<SRC, DT, MOD, WQ>), dim3((unsigned)(c_hi - c_lo)), dim3(kBlock * bp_waves<SRC, MOD>()), (size_t)a.lds_pad, s, a);
        else if (planes_ok && a.order == 4)
            hipLaunchKernelGGL((k_demod_runs_bp<SRC, DT, MOD, WQ, true, 2>), dim3((unsigned)(c_hi - c_lo)), dim3(kBlock * bp_waves<SRC, MOD>()), (size_t)a.lds_pad, s, a);
        else
            hipLaunchKernelGGL((k_demod_runs<SRC, DT, MOD, O2, WQ, true>), dim3((unsigned)(c_hi - c_lo)), dim3(kBlock), 0, s, a);
    }
    const bool tail_is_first = (n_main == 0);                 // a capture shorter than one tile: its only chunk
    if (ranged ? (n_full < a.n && a.launch_lo <= n_main && a.launch_hi > n_main)
               : (n_full < a.n && ((part == 0) || (part == 1 && !tail_is_first) || (part == 2 && tail_is_first)))) {
        a.range_begin = n_full; a.range_end = a.n; a.chunk_base = n_main;
        hipLaunchKernelGGL((k_demod_runs<SRC, DT, MOD, O2, WQ, false>), dim3(1), dim3(kBlock), 0, s, a);
    }
}

template <int SRC, int DT, int MOD>
static void launch_runs_3(const RunArgs &a, bool write_qad, hipStream_t s) {
    const bool o2 = (a.order == 2);
    if (o2) {
        if (write_qad) launch_runs_4<SRC, DT, MOD, true, true>(a, s);
        else launch_runs_4<SRC, DT, MOD, true, false>(a, s);
    } else {
        if (write_qad) launch_runs_4<SRC, DT, MOD, false, true>(a, s);
        else launch_runs_4<SRC, DT, MOD, false, false>(a, s);
    }
}

template <int DT>
static int launch_runs_2(const RunArgs &a, int mod, bool write_qad, hipStream_t s) {
    switch (mod) {
        case URHGPU_MOD_ASK: launch_runs_3<SRC_IQ, DT, URHGPU_MOD_ASK>(a, write_qad, s); return URHGPU_OK;
        case URHGPU_MOD_FSK: launch_runs_3<SRC_IQ, DT, URHGPU_MOD_FSK>(a, write_qad, s); return URHGPU_OK;
        case URHGPU_MOD_OTHER: launch_runs_3<SRC_IQ, DT, URHGPU_MOD_OTHER>(a, write_qad, s); return URHGPU_OK;
        default: return URHGPU_ERR_ARG;
    }
}

// Fused demod + run segmentation over IQ (ASK / FSK / OTHER).
int launch_demod_runs_iq(const RunArgs &a, int dtype, int mod, bool write_qad, hipStream_t s) {
    switch (dtype) {
        case URHGPU_DT_F32: return launch_runs_2<URHGPU_DT_F32>(a, mod, write_qad, s);
        case URHGPU_DT_I8: return launch_runs_2<URHGPU_DT_I8>(a, mod, write_qad, s);
        case URHGPU_DT_U8: return launch_runs_2<URHGPU_DT_U8>(a, mod, write_qad, s);
        case URHGPU_DT_I16: return launch_runs_2<URHGPU_DT_I16>(a, mod, write_qad, s);
        case URHGPU_DT_U16: return launch_runs_2<URHGPU_DT_U16>(a, mod, write_qad, s);
        default: return URHGPU_ERR_DTYPE;
    }
}

// How wide are an integer capture's phase steps?  ONE workgroup samples 4096 pairs of neighbouring samples spread over the capture and counts
// those the hot kernel's fast loop would flag -- re <= 0 or |im| >= 0.4375 re (a phase step of atan(7/16) = 0.41 rad and more) -- among the pairs
// above the noise gate; out[0] = per mille of them, out[1] = pairs counted (pinned host memory: the stream reads it at its next push and picks
// the instantiation, RunArgs::wide_int).  A statistic for a choice of code path only: no result depends on it.
template <int DT>
__global__ __launch_bounds__(1024) void k_wide_probe(const void *iq, int64_t n, float noise_sqrd, int32_t *out) {
    __shared__ int s_w[16], s_v[16];
    int wide = 0, valid = 0;
    const int64_t stride = (n - 2) / 4096 > 0 ? (n - 2) / 4096 : 1;
#pragma unroll
    for (int k = threadIdx.x; k < 4096; k += 1024) {
        const int64_t i = (int64_t)k * stride;
        if (i + 1 >= n) break;
        float a, b, c, d;
        Iq<DT>::load1(iq, i, a, b);
        Iq<DT>::load1(iq, i + 1, c, d);
        if (!(c * c + d * d > noise_sqrd)) continue;
        const float re = a * c + b * d, im = a * d - b * c;
        ++valid;
        if (!(re > 0.0f) || !(__builtin_fabsf(im) < 0.4375f * re)) ++wide;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { wide += __shfl_xor(wide, o); valid += __shfl_xor(valid, o); }
    if ((threadIdx.x & 63) == 0) { s_w[threadIdx.x >> 6] = wide; s_v[threadIdx.x >> 6] = valid; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int w = 0, v = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { w += s_w[k]; v += s_v[k]; }
        out[1] = v;
        out[0] = v > 0 ? (int32_t)((int64_t)w * 1000 / v) : 0;
    }
}
int launch_wide_probe(const void *d_iq, int dtype, int64_t n, float noise_sqrd, int32_t *h_out, hipStream_t s) {
    if (n < 2 || !d_iq || !h_out) return URHGPU_ERR_ARG;
    switch (dtype) {
        case URHGPU_DT_I8: hipLaunchKernelGGL(k_wide_probe<URHGPU_DT_I8>, dim3(1), dim3(1024), 0, s, d_iq, n, noise_sqrd, h_out); break;
        case URHGPU_DT_U8: hipLaunchKernelGGL(k_wide_probe<URHGPU_DT_U8>, dim3(1), dim3(1024), 0, s, d_iq, n, noise_sqrd, h_out); break;
        case URHGPU_DT_I16: hipLaunchKernelGGL(k_wide_probe<URHGPU_DT_I16>, dim3(1), dim3(1024), 0, s, d_iq, n, noise_sqrd, h_out); break;
        case URHGPU_DT_U16: hipLaunchKernelGGL(k_wide_probe<URHGPU_DT_U16>, dim3(1), dim3(1024), 0, s, d_iq, n, noise_sqrd, h_out); break;
        default: return URHGPU_ERR_DTYPE;
    }
    return URHGPU_OK;
}

// Can a pass with these arguments be streamed (RunArgs::progress)?  Only the bit-plane kernel counts its chunks: orders 2 and 4,
// tolerance within its range, whole tiles only (the partial tile at the end of a capture goes through the state-byte kernel).
bool runs_streamable(const RunArgs &a) {
    return URH_BITPLANE && (a.order == 2 || a.order == 4) && a.tol <= kBpMaxTol && a.chunk_len <= (int64_t)kBpMaxRows * kRowSamples &&
           !g_force_state_bytes && a.n >= kTile && a.n % kTile == 0;
}

// Run segmentation over an already demodulated float32 signal (grab_pulse_lens proper).
int launch_runs_qad(const RunArgs &a, hipStream_t s) {
    launch_runs_3<SRC_QAD, URHGPU_DT_F32, URHGPU_MOD_OTHER>(a, false, s);
    return URHGPU_OK;
}

template <int DT, int MOD>
static void launch_afp_3(const RunArgs &a0, int grid, hipStream_t s) {
    // whole chunks of 64 rows through the streaming structure of the hot kernel, the remainder through k_afp_demod
    RunArgs a = a0;
    const int64_t chunk = (int64_t)kBpMaxRows * kRowSamples;
    const int64_t n_main = (a.n / chunk) * chunk;
    if (n_main > 0) {
        a.chunk_len = chunk; a.range_begin = 0; a.range_end = n_main; a.chunk_base = 0;
        hipLaunchKernelGGL((k_demod_runs_bp<SRC_IQ, DT, MOD, true, false>), dim3((unsigned)(n_main / chunk)), dim3(kBlock * bp_waves<SRC_IQ, MOD>()), 0, s, a);
    }
    if (n_main < a.n) {
        RunArgs t = a0;
        if (n_main > 0) {
            t.in = (const char *)a0.in + (size_t)n_main * Iq<DT>::kBytes;
            t.left_halo = (const char *)a0.in + (size_t)(n_main - 2) * Iq<DT>::kBytes;
            t.qad = a0.qad + n_main;
            t.n = a0.n - n_main;
        }
        const int g = (int)std::min<int64_t>(grid, (t.n + kAfpRow - 1) / kAfpRow);
        hipLaunchKernelGGL((k_afp_demod<DT, MOD>), dim3(g), dim3(kAfpBlock), 0, s, t);
    }
}

template <int DT>
static int launch_afp_2(const RunArgs &a, int mod, int grid, hipStream_t s) {
    switch (mod) {
        case URHGPU_MOD_ASK: launch_afp_3<DT, URHGPU_MOD_ASK>(a, grid, s); return URHGPU_OK;
        case URHGPU_MOD_FSK: launch_afp_3<DT, URHGPU_MOD_FSK>(a, grid, s); return URHGPU_OK;
        case URHGPU_MOD_OTHER: launch_afp_3<DT, URHGPU_MOD_OTHER>(a, grid, s); return URHGPU_OK;
        default: return URHGPU_ERR_ARG;
    }
}

int launch_afp_demod(const RunArgs &a, int dtype, int mod, int grid, hipStream_t s) {
    switch (dtype) {
        case URHGPU_DT_F32: return launch_afp_2<URHGPU_DT_F32>(a, mod, grid, s);
        case URHGPU_DT_I8: return launch_afp_2<URHGPU_DT_I8>(a, mod, grid, s);
        case URHGPU_DT_U8: return launch_afp_2<URHGPU_DT_U8>(a, mod, grid, s);
        case URHGPU_DT_I16: return launch_afp_2<URHGPU_DT_I16>(a, mod, grid, s);
        case URHGPU_DT_U16: return launch_afp_2<URHGPU_DT_U16>(a, mod, grid, s);
        default: return URHGPU_ERR_DTYPE;
    }
}

void launch_test_div(uint64_t seed, int reps, unsigned long long *d_mismatches, hipStream_t s) {
    hipLaunchKernelGGL(k_test_div, dim3(4096), dim3(256), 0, s, seed, reps, d_mismatches);
}

void launch_test_atan2f(const float *y, const float *x, int64_t n, float *out, hipStream_t s) {
    hipLaunchKernelGGL(k_test_atan2f, dim3(1024), dim3(256), 0, s, y, x, n, out);
}

}  // namespace urh
