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
// Layout inside a tile (8192 samples, 256 threads): load row r covers 512 consecutive samples,
// thread t owns samples 2t, 2t+1 of the row (one 16-byte load, lane-contiguous => 1 KiB per
// wavefront instruction).  The previous sample of a thread's first sample comes from lane t-1 by
// DPP wave_shr:1; lane 0 of a wavefront re-reads it through the scalar cache.  State bytes go to
// LDS in sample order; in the run phase thread t owns 32 consecutive samples.
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "fdlibm_atan2f.h"
#include "launchers.hpp"
#include "runs.hpp"

namespace urh {

enum { SRC_IQ = 0, SRC_QAD = 1 };

// ---- small device helpers ---------------------------------------------------------------------
__device__ __forceinline__ float dpp_wave_shr1(float x) {
    // value of lane-1 (lane 0 keeps its own value; the caller overrides it)
    int xi = __float_as_int(x);
    return __int_as_float(__builtin_amdgcn_update_dpp(xi, xi, 0x138 /*wave_shr:1*/, 0xf, 0xf, false));
}

template <int DT> struct Iq;
template <> struct Iq<URHGPU_DT_F32> {
    static constexpr int kBytes = 8;
    static __device__ __forceinline__ void load2(const void *b, int64_t i, float &c0, float &d0, float &c1, float &d1) {
        float4 v = ((const float4 *)b)[i >> 1]; c0 = v.x; d0 = v.y; c1 = v.z; d1 = v.w;
    }
    static __device__ __forceinline__ void load1(const void *b, int64_t i, float &c, float &d) {
        float2 v = ((const float2 *)b)[i]; c = v.x; d = v.y;
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
};
template <> struct Iq<URHGPU_DT_U8> {
    static constexpr int kBytes = 2;
    static __device__ __forceinline__ void load2(const void *b, int64_t i, float &c0, float &d0, float &c1, float &d1) {
        uchar4 v = ((const uchar4 *)b)[i >> 1]; c0 = (float)v.x; d0 = (float)v.y; c1 = (float)v.z; d1 = (float)v.w;
    }
    static __device__ __forceinline__ void load1(const void *b, int64_t i, float &c, float &d) {
        uchar2 v = ((const uchar2 *)b)[i]; c = (float)v.x; d = (float)v.y;
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
};
template <> struct Iq<URHGPU_DT_U16> {
    static constexpr int kBytes = 4;
    static __device__ __forceinline__ void load2(const void *b, int64_t i, float &c0, float &d0, float &c1, float &d1) {
        ushort4 v = ((const ushort4 *)b)[i >> 1]; c0 = (float)v.x; d0 = (float)v.y; c1 = (float)v.z; d1 = (float)v.w;
    }
    static __device__ __forceinline__ void load1(const void *b, int64_t i, float &c, float &d) {
        ushort2 v = ((const ushort2 *)b)[i]; c = (float)v.x; d = (float)v.y;
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
template <int MOD>
__device__ __forceinline__ float demod_one(float pc, float pd, float c, float d, const RunArgs &p) {
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
// k_demod_runs<SRC, DT, MOD, ORDER2, WRITE_QAD>
// -----------------------------------------------------------------------------------------------------
template <int SRC, int DT, int MOD, bool ORDER2, bool WRITE_QAD>
__global__ __launch_bounds__(kBlock) void k_demod_runs(const RunArgs p) {
    __shared__ __attribute__((aligned(16))) uint8_t s_state[16 + kTile];   // [15] = state of the sample before the tile
    __shared__ uint32_t s_bm[kBlock + 1];
    __shared__ int s_first, s_last;            // first / last boundary offset in the tile (or kTile / -1)
    __shared__ int s_wave_cnt[4];
    __shared__ uint32_t s_wave_last[4];
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
    const int64_t chunk = blockIdx.x;
    const int64_t a0 = chunk * p.chunk_len;
    const int64_t a1 = (a0 + p.chunk_len < p.n) ? a0 + p.chunk_len : p.n;
    uint64_t *slab = p.slab + chunk * p.slab_stride;
    const bool global_start = (p.left_halo == nullptr);

    // ---- chunk prologue: state of sample a0-1 (halo), init carries --------------------------------
    if (t == 0) {
        uint32_t st = kStNone;
        if (SRC == SRC_QAD) {
            const float *q = (const float *)p.in;
            if (a0 > 0) st = classify<ORDER2>(q[a0 - 1], p);
            else if (!global_start) st = classify<ORDER2>(((const float *)p.left_halo)[0], p);
        } else {
            // qad[a0-1] needs IQ[a0-1] and (FSK) IQ[a0-2]
            float c = 0, d = 0, pc = 0, pd = 0;
            bool have = false, is_global0 = false;
            if (a0 >= 1) {
                Iq<DT>::load1(p.in, a0 - 1, c, d);
                have = true;
                if (a0 >= 2) Iq<DT>::load1(p.in, a0 - 2, pc, pd);
                else if (!global_start) Iq<DT>::load1(p.left_halo, 1, pc, pd);
                else is_global0 = true;           // sample a0-1 is global sample 0 -> NOISE
            } else if (!global_start) {
                Iq<DT>::load1(p.left_halo, 1, c, d);
                Iq<DT>::load1(p.left_halo, 0, pc, pd);
                have = true;
            }
            if (have) {
                const float q = is_global0 ? p.noise_val : demod_one<MOD>(pc, pd, c, d, p);
                st = classify<ORDER2>(q, p);
            }
        }
        s_prev_state8 = st;
        s_pend_pos = -1; s_pend_state = 0; s_lead = -1; s_carry_last = 0xFFFFu; s_first_state = 0xFFFFu; s_count = 0; s_last_pos = 0;
    }
    __syncthreads();

    for (int64_t ta = a0; ta < a1; ta += kTile) {
        const int tv = (int)((a1 - ta < kTile) ? (a1 - ta) : kTile);     // valid samples in this tile
        if (t == 0) {
            s_state[15] = (uint8_t)s_prev_state8;
            s_first = kTile; s_last = -1; s_newpend = -1;
        }
        // ================= phase 1: demodulate + classify, 2 samples per thread per row =================
        // Rows are processed in batches of kBatch: all loads of a batch are issued before any of its
        // arithmetic / stores so that kBatch 16-byte loads per thread are in flight.
        constexpr int kBatch = 4;
#pragma unroll 1
        for (int rb = 0; rb < kRows; rb += kBatch) {
            if (SRC == SRC_QAD) {
                const float *q = (const float *)p.in;
                float2 v[kBatch];
#pragma unroll
                for (int j = 0; j < kBatch; ++j) {
                    const int64_t i0 = ta + (rb + j) * kRowSamples + 2 * t;
                    v[j] = make_float2(0.f, 0.f);
                    if (i0 + 1 < a1) v[j] = *(const float2 *)(q + i0);
                    else if (i0 < a1) v[j].x = q[i0];
                }
#pragma unroll
                for (int j = 0; j < kBatch; ++j) {
                    const int off = (rb + j) * kRowSamples + 2 * t;
                    const int64_t i0 = ta + off;
                    const uint32_t st0 = (i0 < a1) ? classify<ORDER2>(v[j].x, p) : kStNone;
                    const uint32_t st1 = (i0 + 1 < a1) ? classify<ORDER2>(v[j].y, p) : kStNone;
                    *(uint16_t *)(s_state + 16 + off) = (uint16_t)(st0 | (st1 << 8));
                }
            } else {
                float c0[kBatch], d0[kBatch], c1[kBatch], d1[kBatch], sc[kBatch], sd[kBatch];
#pragma unroll
                for (int j = 0; j < kBatch; ++j) {
                    const int off = (rb + j) * kRowSamples + 2 * t;
                    const int64_t i0 = ta + off;
                    c0[j] = d0[j] = c1[j] = d1[j] = 0.f; sc[j] = sd[j] = 0.f;
                    if (i0 + 1 < a1) Iq<DT>::load2(p.in, i0, c0[j], d0[j], c1[j], d1[j]);
                    else if (i0 < a1) Iq<DT>::load1(p.in, i0, c0[j], d0[j]);
                    if (MOD == URHGPU_MOD_FSK) {
                        // wavefront seam: the sample before this wavefront's row segment (uniform address)
                        const int64_t w0 = ta + __builtin_amdgcn_readfirstlane(off - 2 * lane);
                        if (w0 >= 1) { if (w0 - 1 < a1) Iq<DT>::load1(p.in, w0 - 1, sc[j], sd[j]); }
                        else if (!global_start) Iq<DT>::load1(p.left_halo, 1, sc[j], sd[j]);
                    }
                }
#pragma unroll
                for (int j = 0; j < kBatch; ++j) {
                    const int off = (rb + j) * kRowSamples + 2 * t;
                    const int64_t i0 = ta + off;
                    const bool v1 = (i0 + 1 < a1), v0 = (i0 < a1);
                    uint32_t st0 = kStNone, st1 = kStNone;
                    // previous sample of (c0,d0): lane-1's second sample, or the seam sample for lane 0
                    float pc = dpp_wave_shr1(c1[j]), pd = dpp_wave_shr1(d1[j]);
                    if (lane == 0) { pc = sc[j]; pd = sd[j]; }
                    if (v0) {
                        float q0 = demod_one<MOD>(pc, pd, c0[j], d0[j], p);
                        if (i0 == 0 && global_start) q0 = p.noise_val;            // result[0] = NOISE  (:361)
                        st0 = classify<ORDER2>(q0, p);
                        if (v1) {
                            const float q1 = demod_one<MOD>(c0[j], d0[j], c1[j], d1[j], p);
                            st1 = classify<ORDER2>(q1, p);
                            if (WRITE_QAD) *(float2 *)(p.qad + i0) = make_float2(q0, q1);
                        } else if (WRITE_QAD) {
                            p.qad[i0] = q0;
                        }
                    }
                    *(uint16_t *)(s_state + 16 + off) = (uint16_t)(st0 | (st1 << 8));
                }
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
            s_count += s_wave_cnt[0] + s_wave_cnt[1] + s_wave_cnt[2] + s_wave_cnt[3];
            for (int w = 0; w < 4; ++w) if (s_wave_last[w] != 0xFFFFu) s_carry_last = s_wave_last[w];
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
        // initial cur_state of the reference state machine (signal_functions.pyx:421-429):
        // PAUSE if samples[0] == NOISE else the state of the literal 0.0
        uint32_t init = 0;
        if (chunk == 0) {
            bool first_is_noise;
            if (SRC == SRC_QAD) first_is_noise = (((const float *)p.in)[0] == p.noise_val);
            else first_is_noise = true;                            // afp_demod: result[0] = NOISE
            init = first_is_noise ? kStPause : classify<ORDER2>(0.0f, p, false);   // literal 0.0: thresholds only
        }
        ci.init_state = (uint16_t)init;
        ci.first_acc = 0; ci.pend_acc = 0; ci.out_off = 0; ci.prev_pos = -1; ci.prev_state = 0; ci.pad = 0;
        p.chunks[chunk] = ci;
    }
}

// -----------------------------------------------------------------------------------------------------
// Plain afp_demod kernel (no run segmentation): used by urhgpu_afp_demod[_dev] for MOD_OTHER and
// when only Signal.qad is wanted.
// -----------------------------------------------------------------------------------------------------
template <int DT, int MOD>
__global__ __launch_bounds__(kBlock) void k_afp_demod(const RunArgs p) {
    const int lane = threadIdx.x & 63;
    const bool global_start = (p.left_halo == nullptr);
    const int64_t stride = (int64_t)gridDim.x * kRowSamples;
    for (int64_t base = (int64_t)blockIdx.x * kRowSamples; base < p.n; base += stride) {
        const int64_t i0 = base + 2 * threadIdx.x;
        float c0 = 0, d0 = 0, c1 = 0, d1 = 0;
        const bool v1 = (i0 + 1 < p.n), v0 = (i0 < p.n);
        if (v1) Iq<DT>::load2(p.in, i0, c0, d0, c1, d1);
        else if (v0) Iq<DT>::load1(p.in, i0, c0, d0);
        float pc = dpp_wave_shr1(c1), pd = dpp_wave_shr1(d1);
        if (MOD == URHGPU_MOD_FSK) {
            const int64_t w0 = base + 2 * (threadIdx.x - lane);
            float sc = 0, sd = 0;
            if (w0 >= 1) { if (w0 - 1 < p.n) Iq<DT>::load1(p.in, w0 - 1, sc, sd); }
            else if (!global_start) Iq<DT>::load1(p.left_halo, 1, sc, sd);
            if (lane == 0) { pc = sc; pd = sd; }
        }
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

__global__ void k_test_atan2f(const float *y, const float *x, int64_t n, float *out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = atan2f_dev(y[i], x[i]);
}

// ---- host-side launchers ---------------------------------------------------------------------------
template <int SRC, int DT, int MOD>
static void launch_runs_3(const RunArgs &a, int64_t n_chunks, bool write_qad, hipStream_t s) {
    const bool o2 = (a.order == 2);
    dim3 g((unsigned)n_chunks), b(kBlock);
    if (o2) {
        if (write_qad) hipLaunchKernelGGL((k_demod_runs<SRC, DT, MOD, true, true>), g, b, 0, s, a);
        else hipLaunchKernelGGL((k_demod_runs<SRC, DT, MOD, true, false>), g, b, 0, s, a);
    } else {
        if (write_qad) hipLaunchKernelGGL((k_demod_runs<SRC, DT, MOD, false, true>), g, b, 0, s, a);
        else hipLaunchKernelGGL((k_demod_runs<SRC, DT, MOD, false, false>), g, b, 0, s, a);
    }
}

template <int DT>
static int launch_runs_2(const RunArgs &a, int mod, int64_t n_chunks, bool write_qad, hipStream_t s) {
    switch (mod) {
        case URHGPU_MOD_ASK: launch_runs_3<SRC_IQ, DT, URHGPU_MOD_ASK>(a, n_chunks, write_qad, s); return URHGPU_OK;
        case URHGPU_MOD_FSK: launch_runs_3<SRC_IQ, DT, URHGPU_MOD_FSK>(a, n_chunks, write_qad, s); return URHGPU_OK;
        case URHGPU_MOD_OTHER: launch_runs_3<SRC_IQ, DT, URHGPU_MOD_OTHER>(a, n_chunks, write_qad, s); return URHGPU_OK;
        default: return URHGPU_ERR_ARG;
    }
}

// Fused demod + run segmentation over IQ (ASK / FSK / OTHER).
int launch_demod_runs_iq(const RunArgs &a, int dtype, int mod, int64_t n_chunks, bool write_qad, hipStream_t s) {
    switch (dtype) {
        case URHGPU_DT_F32: return launch_runs_2<URHGPU_DT_F32>(a, mod, n_chunks, write_qad, s);
        case URHGPU_DT_I8: return launch_runs_2<URHGPU_DT_I8>(a, mod, n_chunks, write_qad, s);
        case URHGPU_DT_U8: return launch_runs_2<URHGPU_DT_U8>(a, mod, n_chunks, write_qad, s);
        case URHGPU_DT_I16: return launch_runs_2<URHGPU_DT_I16>(a, mod, n_chunks, write_qad, s);
        case URHGPU_DT_U16: return launch_runs_2<URHGPU_DT_U16>(a, mod, n_chunks, write_qad, s);
        default: return URHGPU_ERR_DTYPE;
    }
}

// Run segmentation over an already demodulated float32 signal (grab_pulse_lens proper).
int launch_runs_qad(const RunArgs &a, int64_t n_chunks, hipStream_t s) {
    launch_runs_3<SRC_QAD, URHGPU_DT_F32, URHGPU_MOD_OTHER>(a, n_chunks, false, s);
    return URHGPU_OK;
}

template <int DT>
static int launch_afp_2(const RunArgs &a, int mod, int grid, hipStream_t s) {
    dim3 g(grid), b(kBlock);
    switch (mod) {
        case URHGPU_MOD_ASK: hipLaunchKernelGGL((k_afp_demod<DT, URHGPU_MOD_ASK>), g, b, 0, s, a); return URHGPU_OK;
        case URHGPU_MOD_FSK: hipLaunchKernelGGL((k_afp_demod<DT, URHGPU_MOD_FSK>), g, b, 0, s, a); return URHGPU_OK;
        case URHGPU_MOD_OTHER: hipLaunchKernelGGL((k_afp_demod<DT, URHGPU_MOD_OTHER>), g, b, 0, s, a); return URHGPU_OK;
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

void launch_test_atan2f(const float *y, const float *x, int64_t n, float *out, hipStream_t s) {
    hipLaunchKernelGGL(k_test_atan2f, dim3(1024), dim3(256), 0, s, y, x, n, out);
}

}  // namespace urh
