// costas.hip -- the Costas-loop PSK demodulator (row 5 of SURVEY.md §8a) for gfx950.
//
//   costa_demod   /root/reference/src/urh/cythonext/signal_functions.pyx:252-330
//
// The loop is a nonlinear recurrence over the whole capture: state (freq, phase) in float32, per step a noise gate,
// sinf / cosf of the phase (glibc's, restated in glibc_sincosf.h), a complex product, a clamped error, two state
// updates, a wrap into +-2*pi evaluated in double.  Every output must equal the serial evaluation bit for bit.
//
// Exact parallel evaluation (k_costas_spec / k_costas_map / k_costas_stitch / k_costas_final):
//   * The capture is cut into chunks of kChunk samples.  Started from ANY state, the loop forgets that state: after
//     some hundred un-gated samples two trajectories that lock onto the same phase ambiguity become bit-identical
//     (measured on the reference).  The lock points are pi/2 (order 4) or pi (order 2) apart, each with a
//     twin 2*pi away inside the +-2*pi wrap range.
//   * k_costas_spec runs, for every chunk, one candidate per lock point: it starts `warm` un-gated samples before the
//     chunk from phase 1.5 + k * (pi/2 | pi), freq 0, and records the candidate's state at the chunk start (S), at
//     checkpoints inside the chunk (CP) and at its end (E).  Chunk 0's candidate 0 IS the true trajectory.
//   * k_costas_map: for every chunk c and candidate k of chunk c-1, which candidate of chunk c starts (bitwise) in
//     E[c-1][k]?  k_costas_stitch follows this map from chunk 0: as long as a candidate matches, the true state at
//     the next chunk start is known without touching a sample.  Where nothing matches (acquisition, long gated
//     stretches) it evaluates the chunk serially from the true state, leaving it as soon as the state equals a
//     candidate's checkpoint.  Either way the TRUE state at every chunk start comes out.
//   * k_costas_final re-evaluates every chunk from its true start state, in parallel, and writes the output.  (Tried in round 3:
//     the candidate pass keeps every distinct candidate's output -- collected 16 samples at a time, written as whole 64-byte lines into
//     a slot per candidate -- and a gather kernel copies the true candidate's slot per chunk, so that this pass only re-evaluates
//     chunks without one.  Bit-identical, but no faster: the candidate pass grew from 1.96 to 2.22 ms, the gather took 0.23, against
//     0.69 saved here: 3.36 instead of 3.39 ms per 2^27 samples, for 2 GB more scratch.  Not kept.)
// The result is exact by construction (equality is tested on the state bits, never assumed); only the speed depends on
// how quickly candidates converge.  Work: (K warm + D kChunk + kChunk) steps per chunk instead of kChunk (D <= K distinct candidates).
// Loop orders other than 2 and 4 leave the output unwritten in the reference; they use the serial kernel.
//
// Bound: dependent-instruction latency, hidden by running one chunk-candidate per lane over the whole machine.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <math.h>
#include <string.h>

#include "cmul.hpp"
#include "common.hpp"
#include "launchers.hpp"
#include "glibc_sincosf.h"

namespace urh {

struct CostasArgs {
    const void *iq; int64_t n; float *out;
    float noise_sqrd, alpha, beta, scale, shift;
    int loop_order;
    int warm;                // un-gated samples a candidate runs before its chunk (from the loop bandwidth, see launch_costas)
};
struct CostasState { float freq, phase; };

__device__ __forceinline__ float costas_clamp(float x) {      // :246-250 (NaN passes through: both comparisons are false)
    float r = (x > 1.0f) ? 1.0f : x;
    r = (x < -1.0f) ? -1.0f : r;
    return r;
}

template <int DT> struct CostasLoad;
template <> struct CostasLoad<URHGPU_DT_F32> { static __device__ __forceinline__ float2 at(const void *p, int64_t i) { return ((const float2 *)p)[i]; } };
template <> struct CostasLoad<URHGPU_DT_I8> { static __device__ __forceinline__ float2 at(const void *p, int64_t i) { const char2 v = ((const char2 *)p)[i]; return make_float2((float)v.x, (float)v.y); } };
template <> struct CostasLoad<URHGPU_DT_U8> { static __device__ __forceinline__ float2 at(const void *p, int64_t i) { const uchar2 v = ((const uchar2 *)p)[i]; return make_float2((float)v.x, (float)v.y); } };
template <> struct CostasLoad<URHGPU_DT_I16> { static __device__ __forceinline__ float2 at(const void *p, int64_t i) { const short2 v = ((const short2 *)p)[i]; return make_float2((float)v.x, (float)v.y); } };
template <> struct CostasLoad<URHGPU_DT_U16> { static __device__ __forceinline__ float2 at(const void *p, int64_t i) { const ushort2 v = ((const ushort2 *)p)[i]; return make_float2((float)v.x, (float)v.y); } };

__device__ __forceinline__ bool costas_gated(float2 sm, const CostasArgs &a) { return sm.x * sm.x + sm.y * sm.y <= a.noise_sqrd; }

// One sample of the loop (:291-328): returns the output, updates the state.  err is only carried for loop orders
// other than 2 / 4 (where it never changes); callers keep it.
// UNIT: float32 captures (shift 0, scale 1): (x + 0.0f) / 1.0f == x + 0.0f for every x, the two IEEE divisions are dropped.
// ORDER: 2 / 4 = the loop order as a compile-time constant (the speculative kernels: five wave-uniform branches less per step);
// 0 = read a.loop_order.
template <bool UNIT = false, int ORDER = 0>
__device__ __forceinline__ float costas_step(float2 sm, CostasState &st, float &err, const CostasArgs &a) {
    const int loop_order = ORDER ? ORDER : a.loop_order;
    if (costas_gated(sm, a)) return -4.0f;                              // NOISE_FSK_PSK, state frozen (:293-295)
    const double two_pi = 2 * 3.14159265358979323846;
    const float real_float = UNIT ? sm.x + 0.0f : (sm.x + a.shift) / a.scale, imag_float = UNIT ? sm.y + 0.0f : (sm.y + a.shift) / a.scale;
    const float2 cur = make_float2(real_float + 0.0f * imag_float, 1.0f * imag_float);   // re + imag_unit * im
    float sn, cs;
    if (urh_sc_abstop12(st.phase) < 0x42f) urh_sincosf_fast(-st.phase, &sn, &cs);        // |phase| < 120: always (the loop keeps it within +-2 pi)
    else { sn = urh_sinf(-st.phase); cs = urh_cosf(-st.phase); }
    const float2 nco = make_float2(cs + 0.0f * sn, 1.0f * sn);
    const float2 z = cmul(nco, cur);
    if (loop_order == 2) {
        err = z.y * z.x;
    } else if (loop_order == 4) {
        const float f1 = z.x > 0.0f ? 1.0f : -1.0f, f2 = z.y > 0.0f ? 1.0f : -1.0f;
        err = f1 * z.y - f2 * z.x;
    }
    err = costas_clamp(err);
    st.freq += a.beta * err;
    st.phase += st.freq + a.alpha * err;
    while ((double)st.phase > two_pi) st.phase = (float)((double)st.phase - two_pi);     // double compare / subtract (:318-321)
    while ((double)st.phase < -two_pi) st.phase = (float)((double)st.phase + two_pi);
    st.freq = costas_clamp(st.freq);
    if (loop_order == 2) return z.x;
    if (loop_order == 4) return (float)(2.0 * (double)z.x + (double)z.y);
    return 0.0f;
}

// The same step as straight-line code (the speculative kernels; loop order 2 or 4): the gate, the two wraps and the clamps are
// selects; what cannot happen in a healthy loop (|phase| beyond 2 pi on entry, a product with two NaN parts) is collected in ONE flag,
// tested once at the end, and sends the whole wavefront through costas_step from the saved state.  Measured effect: small (candidate
// pass 1.97 -> 1.91 ms): nearly every one of a step's ~94 instructions depends on the one before it, a dependent VALU operation
// issues every ~9.5 cycles whatever its type (tools/kbench/lbench.hip: fp32, fp64, conversions alike), and two trajectories per SIMD
// lane cannot fill the gaps -- the pass sits at 94 x 9.5 cycles x 4096 steps.  One wrap is enough when
// |phase| <= 2 pi on entry: |freq'| <= 1 + beta, |alpha err| <= alpha, alpha + beta < 4 for every bandwidth, so |phase'| < 2 pi + 5 < 4 pi.
template <bool UNIT, int ORDER>
__device__ __forceinline__ float costas_step_bf(float2 sm, CostasState &st, float &err, const CostasArgs &a) {
    static_assert(ORDER == 2 || ORDER == 4, "the serial kernel handles the other orders");
    const double two_pi = 2 * 3.14159265358979323846;
    const CostasState old = st;
    const float old_err = err;
    const bool gated = costas_gated(sm, a);
    const float real_float = UNIT ? sm.x + 0.0f : (sm.x + a.shift) / a.scale, imag_float = UNIT ? sm.y + 0.0f : (sm.y + a.shift) / a.scale;
    const float2 cur = make_float2(real_float + 0.0f * imag_float, 1.0f * imag_float);   // re + imag_unit * im
    const bool in_range = __builtin_fabsf(old.phase) <= __uint_as_float(0x40C90FDBu);   // the float next above 2 pi (NaN: false)
    float sn, cs;
    urh_sincosf_fast(-old.phase, &sn, &cs);
    const float2 nco = make_float2(cs + 0.0f * sn, 1.0f * sn);
    float2 z;
    z.x = nco.x * cur.x - nco.y * cur.y;                     // cmul() without its NaN-recovery branch
    z.y = nco.x * cur.y + nco.y * cur.x;
    const bool rare = !gated && (!in_range || ((z.x != z.x) & (z.y != z.y)));
    float e;
    if (ORDER == 2) e = z.y * z.x;
    else {
        const float f1 = z.x > 0.0f ? 1.0f : -1.0f, f2 = z.y > 0.0f ? 1.0f : -1.0f;
        e = f1 * z.y - f2 * z.x;
    }
    e = costas_clamp(e);
    float freq = old.freq + a.beta * e;
    float phase = old.phase + (freq + a.alpha * e);
    const double d = (double)phase;
    const float wrapped_down = (float)(d - two_pi), wrapped_up = (float)(d + two_pi);
    phase = (d > two_pi) ? wrapped_down : ((d < -two_pi) ? wrapped_up : phase);
    freq = costas_clamp(freq);
    const float out = (ORDER == 2) ? z.x : (float)(2.0 * (double)z.x + (double)z.y);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(rare) != 0, 0)) {      // wavefront-uniform; never in a healthy loop
        st = old; err = old_err;
        return costas_step<UNIT, ORDER>(sm, st, err, a);
    }
    st.freq = gated ? old.freq : freq;
    st.phase = gated ? old.phase : phase;
    err = gated ? old_err : e;
    return gated ? -4.0f : out;
}

// ---- serial evaluation: one wavefront, every lane the same recurrence, lane k keeps outputs k, k+64, ... of a tile ----
constexpr int kCostasTile = 1024;

template <int DT>
__global__ __launch_bounds__(64) void k_costas(const CostasArgs a) {
    __shared__ float2 s_x[kCostasTile];
    const int lane = threadIdx.x;
    CostasState st{0.0f, 1.5f};                                // :261
    float err = 0.0f;
    if (lane == 0 && a.n > 0) a.out[0] = -4.0f;                // reference: np.empty, never written (documented in urhgpu.h)
    for (int64_t base = 0; base < a.n; base += kCostasTile) {
        const int tv = (int)((a.n - base < kCostasTile) ? (a.n - base) : kCostasTile);
        __syncthreads();
        for (int u = lane; u < tv; u += 64) s_x[u] = CostasLoad<DT>::at(a.iq, base + u);
        __syncthreads();
        float mine[kCostasTile / 64];
#pragma unroll 1
        for (int g = 0; g < kCostasTile / 64; ++g) {
            float keep = 0.0f;
#pragma unroll 1
            for (int u = 0; u < 64; ++u) {
                const int k = g * 64 + u;
                if (k >= tv || (base + k) == 0) continue;       // the loop starts at sample 1 (:289)
                const float o = costas_step<DT == URHGPU_DT_F32>(s_x[k], st, err, a);
                if (u == lane) keep = o;
            }
            mine[g] = keep;
        }
#pragma unroll 1
        for (int g = 0; g < kCostasTile / 64; ++g) {
            const int k = g * 64 + lane;
            if (k < tv && base + k != 0) a.out[base + k] = mine[g];
        }
    }
}

// ---- speculative parallel evaluation ----------------------------------------------------------------------------------
// samples per chunk.  Round 3, config-5 capture (tools/costas_chunk_ab.py, libraries built with -DURH_COSTAS_CHUNK=...): 1024 / 2048 / 4096 /
// 8192 samples per chunk take 5.04 / 3.83 / 3.39 / 3.92 ms -- shorter chains, but as many warm-up steps per chunk as before.
#ifndef URH_COSTAS_CHUNK
#define URH_COSTAS_CHUNK 4096
#endif
constexpr int kChunk = URH_COSTAS_CHUNK;
constexpr int kWarmBackFactor = 16; // a candidate looks back at most 16 x its warm-up length for un-gated samples
constexpr int kCkpt = 256;          // checkpoint spacing inside a chunk
constexpr int kNumCkpt = kChunk / kCkpt;   // checkpoints at offsets kCkpt, 2 kCkpt, ... < kChunk  (index j = off / kCkpt - 1)
constexpr int kMaxCand = 8;

struct SpecBuffers {
    CostasState *S;        // [n_chunks][K]  state at the chunk start (before its first sample)
    CostasState *E;        // [n_chunks][K]  state after the chunk's last sample
    CostasState *CP;       // [n_chunks][kNumCkpt][K] state after offset (j+1)*kCkpt samples of the chunk
    uint32_t *map;         // [n_chunks] nibble k: candidate of this chunk that starts in E[c-1][k], 0xF none
    CostasState *T;        // [n_chunks] TRUE state at the chunk start (written by the stitch)
    int32_t *run_list;     // [n_chunks * K] chunk * K + candidate of every DISTINCT candidate (k_costas_spec appends, k_costas_run consumes)
    int32_t *run_count;    // [1] entries in run_list
    uint8_t *is_rep;       // [n_chunks][K] 1: the candidate runs (no lower-numbered candidate of the chunk starts in the same state)
    int32_t *gidx;         // [n_chunks] candidate whose trajectory IS the true one over the whole chunk, or -1 (written by the stitch)
    int32_t *ungated;      // [n_chunks] un-gated samples in the chunk
    CostasState *resume;   // [1] true state at the start of chunk stats[3] when the stitch hands back to the host
    int32_t *stats;        // [0] chunks resolved by the map, [1] by a checkpoint inside a serial run, [2] fully serial,
                           // [3] chunk the stitch stopped in front of (n_chunks: finished), [4] re-speculation rounds
};

__device__ __forceinline__ bool same_state(CostasState a, CostasState b) {
    return __float_as_uint(a.freq) == __float_as_uint(b.freq) && __float_as_uint(a.phase) == __float_as_uint(b.phase);
}
__device__ __forceinline__ int64_t chunk_begin(int64_t c) { return 1 + c * (int64_t)kChunk; }   // sample 0 is not part of the loop

// Candidates of the chunks c >= c_from.  use_seed: all candidates start with freq = seed_freq (the true loop's own
// frequency where the chain last broke) instead of the estimate from the data.
template <int DT, int ORDER>
__global__ __launch_bounds__(256) void k_costas_spec(const CostasArgs a, SpecBuffers b, int64_t n_chunks, int K, int64_t c_from,
                                                      int use_seed, float seed_freq) {
    const int64_t gid = blockIdx.x * 256ll + threadIdx.x;
    const int64_t c = c_from + gid / K;
    const int k = (int)(gid % K);
    if (c >= n_chunks) return;
    const int64_t s0 = chunk_begin(c);
    // candidate k: phase 1.5 + (k - K/2) * spacing, spaced pi/2 (order 4, K = 8) or pi (order 2, K = 4): every lock point and its
    // twin 2*pi away; candidate K/2 has the reference's own initial phase 1.5
    const float spacing = (a.loop_order == 4) ? 1.57079632679489661923f : 3.14159265358979323846f;
    CostasState st{0.0f, 1.5f + (float)(k - K / 2) * spacing};
    float err = 0.0f;
    int64_t p = s0;
    if (c == 0) {
        st = CostasState{0.0f, 1.5f};                       // every candidate of chunk 0 is the true trajectory
    } else {
        int ungated = 0;
        const int64_t back = (int64_t)kWarmBackFactor * a.warm;
        // walk back until `warm` un-gated samples lie between p and the chunk -- eight samples per round trip to memory
        while (p > 1 && ungated < a.warm && s0 - p < back) {
            float2 g[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = CostasLoad<DT>::at(a.iq, (p - 1 - j >= 1) ? p - 1 - j : 1);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (p > 1 && ungated < a.warm && s0 - p < back) {
                    --p;
                    if (!costas_gated(g[j], a)) ++ungated;
                }
            }
        }
        if (p == 1) {
            st = CostasState{0.0f, 1.5f};                   // reached the start of the capture: exact, not a guess
        } else if (use_seed) {
            st.freq = seed_freq;
        } else {
            // Seed the candidate's frequency with the carrier offset of the warm-up stretch (M-th power of the
            // differential phase: the PSK symbol steps are multiples of 2*pi/M and drop out).  A candidate that starts
            // at freq 0 against an offset beyond the loop bandwidth needs thousands of samples to pull in; seeded, it
            // locks within the warm-up.  Heuristic only: a wrong seed costs time (serial fallback), never exactness.
            float ax = 0.0f, ay = 0.0f;
            float2 prev = CostasLoad<DT>::at(a.iq, p);
            bool prev_ok = !costas_gated(prev, a);
            const int64_t wend = (p + 512 < s0) ? p + 512 : s0;
            for (int64_t i0 = p + 1; i0 < wend; i0 += 8) {
                float2 g[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) g[j] = CostasLoad<DT>::at(a.iq, (i0 + j < wend) ? i0 + j : wend - 1);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (i0 + j < wend) {
                        const float2 cur = g[j];
                        const bool ok = !costas_gated(cur, a);
                        if (ok && prev_ok) {
                            const float cr = (cur.x + a.shift) / a.scale, ci = (cur.y + a.shift) / a.scale;
                            const float pr = (prev.x + a.shift) / a.scale, pi = (prev.y + a.shift) / a.scale;
                            float dx = cr * pr + ci * pi, dy = ci * pr - cr * pi;          // cur * conj(prev)
                            float tx = dx * dx - dy * dy, ty = 2.0f * dx * dy;             // ^2
                            if (a.loop_order == 4) { const float ux = tx * tx - ty * ty, uy = 2.0f * tx * ty; tx = ux; ty = uy; }
                            ax += tx; ay += ty;
                        }
                        prev = cur; prev_ok = ok;
                    }
                }
            }
            if (ax != 0.0f || ay != 0.0f) st.freq = costas_clamp(atan2f(ay, ax) / (float)a.loop_order);
        }
        {
            constexpr int PF = 4;                               // as in k_costas_run: the next four samples are on their way
            float2 cur[PF];
#pragma unroll
            for (int j = 0; j < PF; ++j) cur[j] = CostasLoad<DT>::at(a.iq, (p + j < s0) ? p + j : s0 - 1);
            for (int64_t i = p; i < s0; i += PF) {
                float2 nxt[PF];
#pragma unroll
                for (int j = 0; j < PF; ++j) { const int64_t q = i + PF + j; nxt[j] = CostasLoad<DT>::at(a.iq, (q < s0) ? q : s0 - 1); }
#pragma unroll
                for (int j = 0; j < PF; ++j) if (i + j < s0) costas_step_bf<DT == URHGPU_DT_F32, ORDER>(cur[j], st, err, a);
#pragma unroll
                for (int j = 0; j < PF; ++j) cur[j] = nxt[j];
            }
        }
    }
    b.S[c * K + k] = st;
    // Candidates that have met during the warm-up -- the twins 2*pi apart do as soon as a carrier offset has wrapped the
    // phase once, every candidate of chunk 0 is the true trajectory -- would repeat each other sample for sample: only
    // the lowest-numbered one of every distinct start state runs the chunk (k_costas_run, lanes packed densely).
    const int lane = threadIdx.x & 63, base = lane - k;              // the chunk's K candidates are adjacent lanes of one wavefront
    bool dup = false;
    for (int j = 0; j < K; ++j) {
        CostasState o;
        o.freq = __shfl(st.freq, base + j); o.phase = __shfl(st.phase, base + j);
        if (j < k && same_state(o, st)) dup = true;
    }
    b.is_rep[c * K + k] = dup ? 0 : 1;
    const unsigned long long m = __ballot(!dup);
    int pos = 0;
    if (lane == __builtin_ctzll(m)) pos = atomicAdd(b.run_count, __builtin_popcountll(m));
    pos = __shfl(pos, __builtin_ctzll(m));
    if (!dup) b.run_list[pos + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = (int32_t)(c * K + k);
}

// The chunk itself, one lane per distinct candidate: checkpoints, end state, un-gated sample count.
template <int DT, int ORDER>
__global__ __launch_bounds__(256) void k_costas_run(const CostasArgs a, SpecBuffers b, int K) {
    const int64_t gid = blockIdx.x * 256ll + threadIdx.x;
    if (gid >= *b.run_count) return;
    const int64_t ck = b.run_list[gid];
    const int64_t c = ck / K;
    const int k = (int)(ck % K);
    const int64_t s0 = chunk_begin(c);
    const int64_t e0 = (s0 + kChunk < a.n) ? s0 + kChunk : a.n;
    CostasState st = b.S[ck];
    float err = 0.0f;
    int ung = 0;
    // The samples of the NEXT four steps are requested before the current four are evaluated: a step is ~130 dependent instructions,
    // and with two wavefronts per SIMD (one lane per distinct candidate) nothing else hides the latency of its own load -- half of
    // the wave cycles were spent in s_waitcnt before.
    constexpr int PF = 4;
    float2 cur[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) cur[j] = CostasLoad<DT>::at(a.iq, (s0 + j < e0) ? s0 + j : e0 - 1);
    for (int64_t i = s0; i < e0; i += PF) {
        float2 nxt[PF];
#pragma unroll
        for (int j = 0; j < PF; ++j) { const int64_t q = i + PF + j; nxt[j] = CostasLoad<DT>::at(a.iq, (q < e0) ? q : e0 - 1); }
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            if (i + j < e0) {
                if (!costas_gated(cur[j], a)) ++ung;
                costas_step_bf<DT == URHGPU_DT_F32, ORDER>(cur[j], st, err, a);
                const int off = (int)(i + j - s0) + 1;
                if (off % kCkpt == 0 && off < kChunk) b.CP[(c * kNumCkpt + (off / kCkpt - 1)) * K + k] = st;
            }
        }
#pragma unroll
        for (int j = 0; j < PF; ++j) cur[j] = nxt[j];
    }
    b.E[ck] = st;
    if (k == 0) b.ungated[c] = ung;
}

__global__ __launch_bounds__(256) void k_costas_map(SpecBuffers b, int64_t n_chunks, int K, int64_t c_from) {
    const int64_t c = c_from + blockIdx.x * 256ll + threadIdx.x;
    if (c >= n_chunks) return;
    uint32_t m = 0xFFFFFFFFu;
    if (c > 0) {
        for (int k = 0; k < K; ++k) {
            int hit = 0xF;
            if (b.is_rep[(c - 1) * K + k]) {                 // a duplicate did not run: no end state (it is never the carrying candidate)
                const CostasState e = b.E[(c - 1) * K + k];
                for (int q = 0; q < K; ++q) if (same_state(e, b.S[c * K + q])) { hit = q; break; }
            }
            m = (m & ~(0xFu << (4 * k))) | ((uint32_t)hit << (4 * k));
        }
    }
    b.map[c] = m;
}

// Composition of two chunk maps (8 nibbles: candidate of the previous chunk -> candidate of this chunk, 0xF = none):
// (later o earlier)[k] = later[earlier[k]], none stays none.
__device__ __forceinline__ uint32_t map_compose(uint32_t later, uint32_t earlier) {
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t mid = (earlier >> (4 * k)) & 0xFu;
        const uint32_t v = (mid == 0xFu) ? 0xFu : ((later >> (4 * mid)) & 0xFu);
        r |= v << (4 * k);
    }
    return r;
}

// One wavefront walks the chunks c_from .. in order.  While a candidate carries the true trajectory the walk is a
// composition of the chunk maps, evaluated 1024 chunks at a time (16 per lane) by a wavefront prefix "scan" with map_compose (the serial
// walk -- two dependent global loads per chunk -- took 10.9 ms for the 32 767 chunks of a 1 GiB capture).  Where the maps
// end (acquisition, long gated stretches) the chunk is handled as before: compare the true state with the chunk's
// candidates, else evaluate it serially from the true state until it meets a candidate at a checkpoint.
// Entry: the true state at the start of chunk c_from is *b.resume (c_from == 1: chunk 0's end, taken from E[0][0]).
// Exit: either all chunks are resolved, or -- when allow_break -- the chain broke for good (a chunk with plenty of
// un-gated samples was evaluated serially to its end and met no candidate): stats[3] = the next chunk, *b.resume = its true
// start state, and the host re-speculates the remaining chunks around that state's frequency.
constexpr int kStitchBlock = 1024;              // 16 wavefronts: the fast-forward composes 4096 chunk maps per round
template <int DT, int ORDER>
__global__ __launch_bounds__(kStitchBlock) void k_costas_stitch(const CostasArgs a, SpecBuffers b, int64_t n_chunks, int K, int64_t c_from,
                                                                 int allow_break) {
    __shared__ uint32_t s_wp[kStitchBlock / 64];    // per-wavefront composition of its chunk maps
    __shared__ int s_fail, s_cand, s_flags;         // first position where the chain ends; broadcast slots
    __shared__ CostasState s_T;
    __shared__ long long s_c;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int cand = -1;                                  // candidate of chunk c - 1 that is the true trajectory, or -1
    CostasState T = *b.resume;                      // true state at the start of chunk c (valid while cand < 0)
    if (c_from == 1) { cand = 0; if (tid == 0) { b.T[0] = CostasState{0.0f, 1.5f}; b.gidx[0] = 0; } }   // chunk 0: every candidate is exact
    int n_map = 0, n_ckpt = 0, n_serial = 0;        // (kept identically by every thread in the fast-forward, by wavefront 0 below)
    int64_t stop_at = n_chunks;
    int64_t c = c_from;
    bool map_failed = false;                        // chunk c: the map already said that no candidate starts in T
    bool stop = false;
    while (c < n_chunks && !stop) {
        while (cand >= 0 && c < n_chunks) {         // ---- fast-forward over up to 4096 chunks: 4 consecutive chunks per thread
            constexpr int Q = 4;
            if (tid == 0) s_fail = 0x7fffffff;
            const int64_t cc0 = c + (int64_t)Q * tid;
            uint32_t L[Q];                                            // L[q] = map[cc0 + q] o ... o map[cc0]
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const uint32_t m = (cc0 + q < n_chunks) ? b.map[cc0 + q] : 0xFFFFFFFFu;
                L[q] = q ? map_compose(m, L[q - 1]) : m;
            }
            uint32_t P = L[Q - 1];                                    // inclusive prefix composition over the lanes of the wavefront
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t u = __shfl_up(P, o);
                if (lane >= o) P = map_compose(P, u);
            }
            if (lane == 63) s_wp[wave] = P;
            __syncthreads();
            uint32_t X = 0x76543210u;                                 // everything before my first chunk: the wavefronts before mine ...
            for (int w = 0; w < wave; ++w) X = map_compose(s_wp[w], X);
            const uint32_t before = __shfl_up(P, 1);                  // ... then the lanes before me
            if (lane != 0) X = map_compose(before, X);
            uint32_t g[Q], gp[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) g[q] = (map_compose(L[q], X) >> (4 * cand)) & 0xFu;   // candidate of chunk cc0 + q on the true trajectory
            gp[0] = (X >> (4 * cand)) & 0xFu;
#pragma unroll
            for (int q = 1; q < Q; ++q) gp[q] = g[q - 1];
            CostasState Tl[Q];
            int firstq = Q;                                           // my first chunk where the chain ends
#pragma unroll
            for (int q = Q - 1; q >= 0; --q) {
                const int64_t cc = cc0 + q;
                const bool reach = (gp[q] != 0xFu) && (cc < n_chunks);   // the true state at the start of chunk cc is E[cc - 1][gp]
                Tl[q] = T;
                if (reach) { Tl[q] = b.E[(cc - 1) * K + gp[q]]; b.T[cc] = Tl[q]; b.gidx[cc] = (g[q] == 0xFu) ? -1 : (int32_t)g[q]; }
                if (reach && g[q] == 0xFu) firstq = q;
            }
            if (firstq < Q) atomicMin(&s_fail, tid * Q + firstq);
            __syncthreads();
            const int fail = s_fail;
            const int64_t left = n_chunks - c;
            const int total = (int)(left < (int64_t)kStitchBlock * Q ? left : (int64_t)kStitchBlock * Q);
            const int pos = (fail == 0x7fffffff) ? total - 1 : fail;      // the chunk whose thread has the hand-over value
            if (tid == pos / Q) {
                const int qq = pos % Q;
                uint32_t gl = g[0];
                CostasState Ts = Tl[0];
#pragma unroll
                for (int q = 1; q < Q; ++q) if (qq == q) { gl = g[q]; Ts = Tl[q]; }
                s_cand = (int)gl; s_T = Ts;
            }
            __syncthreads();
            if (fail == 0x7fffffff) {
                cand = s_cand;
                n_map += total; c += total;
            } else {
                T = s_T;
                n_map += fail; c += fail; cand = -1; map_failed = true;
            }
            __syncthreads();                                          // (s_fail / s_cand / s_T are rewritten by the next round)
        }
        if (c >= n_chunks) break;
        // ---- chunk c without a carrying candidate: T is the true state at its start.  One wavefront's work; the others wait.
        if (wave == 0) {
            bool done = false;
            if (!map_failed) {
                if (lane == 0) { b.T[c] = T; b.gidx[c] = -1; }
                const bool hit = lane < K && same_state(T, b.S[c * K + lane]);
                const unsigned long long m = __ballot(hit);
                if (m) {
                    cand = __builtin_ctzll(m);
                    if (lane == 0) b.gidx[c] = cand;
                    ++n_map; ++c;
                    done = true;
                }
            }
            if (!done) {
                map_failed = false;
                if (b.ungated[c] == 0) { ++n_serial; ++c; done = true; }     // fully gated chunk: the state does not move
            }
            if (!done) {
                // no candidate starts in T: evaluate the chunk from T until the state meets a candidate at a checkpoint
                const int64_t s0 = chunk_begin(c);
                const int64_t e0 = (s0 + kChunk < a.n) ? s0 + kChunk : a.n;
                float err = 0.0f;
                CostasState st = T;
                for (int64_t i = s0; i < e0; ++i) {
                    costas_step_bf<DT == URHGPU_DT_F32, ORDER>(CostasLoad<DT>::at(a.iq, i), st, err, a);
                    const int off = (int)(i - s0) + 1;
                    if (off % kCkpt == 0 && off < kChunk) {
                        const bool hit = lane < K && b.is_rep[c * K + lane] && same_state(st, b.CP[(c * kNumCkpt + (off / kCkpt - 1)) * K + lane]);
                        const unsigned long long m = __ballot(hit);
                        if (m) { cand = __builtin_ctzll(m); break; }
                    }
                }
                if (cand >= 0) { ++n_ckpt; ++c; }       // gidx[c] stays -1: only part of the chunk lies on the candidate
                else {
                    ++n_serial;
                    T = st;                             // the state at the END of chunk c = start of chunk c + 1
                    if (allow_break && b.ungated[c] >= kChunk / 2 && c + 1 < n_chunks) { stop_at = c + 1; stop = true; }
                    else ++c;
                }
            }
            if (lane == 0) { s_c = c; s_cand = cand; s_T = T; s_flags = (map_failed ? 1 : 0) | (stop ? 2 : 0); }
        }
        __syncthreads();
        c = s_c; cand = s_cand; T = s_T; map_failed = (s_flags & 1) != 0; stop = (s_flags & 2) != 0;
        __syncthreads();
    }
    if (tid == 0) {
        b.stats[0] += n_map; b.stats[1] += n_ckpt; b.stats[2] += n_serial; b.stats[3] = (int32_t)stop_at;
        *b.resume = T;
    }
}

// Output pass: every chunk from its TRUE start state.  A chunk that lies on a candidate's trajectory from its first
// sample (gidx >= 0) is evaluated by kNumCkpt lanes, each from the candidate's checkpoint state (bitwise the true state
// there); any other chunk by one lane from T[c].
template <int DT, int ORDER>
__global__ __launch_bounds__(256) void k_costas_final(const CostasArgs a, SpecBuffers b, int64_t n_chunks, int K) {
    const int64_t gid = blockIdx.x * 256ll + threadIdx.x;
    const int64_t c = gid / kNumCkpt;
    const int j = (int)(gid % kNumCkpt);            // segment j = samples [j kCkpt, (j + 1) kCkpt) of the chunk
    if (c >= n_chunks) return;
    if (gid == 0 && a.n > 0) a.out[0] = -4.0f;      // reference: np.empty, never written (documented in urhgpu.h)
    const int64_t s0 = chunk_begin(c);
    const int64_t e0 = (s0 + kChunk < a.n) ? s0 + kChunk : a.n;
    const int g = b.gidx[c];
    int64_t i0 = s0, i1 = e0;
    CostasState st = b.T[c];
    if (g >= 0) {
        i0 = s0 + (int64_t)j * kCkpt;
        i1 = (i0 + kCkpt < e0) ? i0 + kCkpt : e0;
        if (j > 0) st = b.CP[(c * kNumCkpt + (j - 1)) * K + g];
    } else if (j != 0) return;
    float err = 0.0f;
    // 16 samples (one 128-byte line of complex64) per round: the lanes of a wavefront are 2 KiB apart, so every line a lane
    // touches is fetched for that lane alone -- ask for all of it at once instead of 16 times 8 bytes
    constexpr int kTileF = 16;
    int64_t i = i0;
    for (; i + kTileF <= i1; i += kTileF) {
        float2 x[kTileF];
#pragma unroll
        for (int u = 0; u < kTileF; ++u) x[u] = CostasLoad<DT>::at(a.iq, i + u);
        float o[kTileF];
#pragma unroll 1
        for (int u = 0; u < kTileF; ++u) {
            float2 xv = x[0];
#pragma unroll
            for (int q = 1; q < kTileF; ++q) if (u == q) xv = x[q];
            const float ov = costas_step_bf<DT == URHGPU_DT_F32, ORDER>(xv, st, err, a);
#pragma unroll
            for (int q = 0; q < kTileF; ++q) if (u == q) o[q] = ov;
        }
#pragma unroll
        for (int u = 0; u < kTileF; ++u) a.out[i + u] = o[u];
    }
    for (; i < i1; ++i) a.out[i] = costas_step_bf<DT == URHGPU_DT_F32, ORDER>(CostasLoad<DT>::at(a.iq, i), st, err, a);
}

size_t costas_scratch_bytes(int64_t n) {
    const int64_t nc = (std::max<int64_t>(n - 1, 0) + kChunk - 1) / kChunk + 1;
    return (size_t)nc * kMaxCand * sizeof(CostasState) * 2 + (size_t)nc * kNumCkpt * kMaxCand * sizeof(CostasState) +
           (size_t)nc * 4 * 3 + (size_t)nc * sizeof(CostasState) + (size_t)nc * kMaxCand * 5 + 3 * 64 + 16 * 256;
}

constexpr int kMaxRounds = 24;     // re-speculation rounds before the stitch stops handing back (and runs serially)

// NOTE: synchronises the stream (at least once): the host has to learn whether the chunk chain closed.
template <int DT, int ORDER>
static int launch_costas_spec(const CostasArgs &a, void *scratch, urhgpu_ctx *ctx);

template <int DT>
static int launch_costas_dt(const CostasArgs &a, void *scratch, urhgpu_ctx *ctx) {
    const bool parallel = (a.loop_order == 2 || a.loop_order == 4) && a.n > 2 * kChunk && scratch != nullptr;
    if (!parallel) {
        hipLaunchKernelGGL(k_costas<DT>, dim3(1), dim3(64), 0, ctx->stream, a);
        ctx->h_counts[12] = ctx->h_counts[13] = ctx->h_counts[14] = 0;
        return URHGPU_OK;
    }
    return a.loop_order == 4 ? launch_costas_spec<DT, 4>(a, scratch, ctx) : launch_costas_spec<DT, 2>(a, scratch, ctx);
}

template <int DT, int ORDER>
static int launch_costas_spec(const CostasArgs &a, void *scratch, urhgpu_ctx *ctx) {
    hipStream_t s = ctx->stream;
    const int K = (a.loop_order == 4) ? 8 : 4;
    const int64_t nc = (a.n - 1 + kChunk - 1) / kChunk;
    char *p = (char *)scratch;
    auto take = [&](size_t bytes) { char *r = p; p += (bytes + 255) & ~size_t(255); return r; };
    SpecBuffers b;
    b.S = (CostasState *)take((size_t)nc * K * sizeof(CostasState));
    b.E = (CostasState *)take((size_t)nc * K * sizeof(CostasState));
    b.CP = (CostasState *)take((size_t)nc * kNumCkpt * K * sizeof(CostasState));
    b.map = (uint32_t *)take((size_t)nc * 4);
    b.T = (CostasState *)take((size_t)nc * sizeof(CostasState));
    b.ungated = (int32_t *)take((size_t)nc * 4);
    b.gidx = (int32_t *)take((size_t)nc * 4);
    b.run_list = (int32_t *)take((size_t)nc * K * 4);
    b.is_rep = (uint8_t *)take((size_t)nc * K);
    b.run_count = (int32_t *)take(64);
    b.resume = (CostasState *)take(64);
    b.stats = (int32_t *)take(64);
    URH_HIP(hipMemsetAsync(b.stats, 0, 64, s));
    URH_HIP(hipMemsetAsync(b.resume, 0, 64, s));
    int64_t c_from = 0;
    int use_seed = 0;
    float seed_freq = 0.0f;
    int32_t *h = (int32_t *)(ctx->h_counts + 12);            // pinned: stats[0..4], then the resume state
    int rounds = 0;
    for (int round = 0;; ++round) {
        const int64_t todo = nc - c_from;
        URH_HIP(hipMemsetAsync(b.run_count, 0, 4, s));
        hipLaunchKernelGGL((k_costas_spec<DT, ORDER>), dim3((unsigned)((todo * K + 255) / 256)), dim3(256), 0, s, a, b, nc, K, c_from, use_seed,
                           seed_freq);
        hipLaunchKernelGGL((k_costas_run<DT, ORDER>), dim3((unsigned)((todo * K + 255) / 256)), dim3(256), 0, s, a, b, K);
        hipLaunchKernelGGL(k_costas_map, dim3((unsigned)((todo + 255) / 256)), dim3(256), 0, s, b, nc, K, std::max<int64_t>(c_from, 1));
        hipLaunchKernelGGL((k_costas_stitch<DT, ORDER>), dim3(1), dim3(kStitchBlock), 0, s, a, b, nc, K, std::max<int64_t>(c_from, 1),
                           round < kMaxRounds ? 1 : 0);
        URH_HIP(hipGetLastError());
        URH_HIP(hipMemcpyAsync(h, b.stats, 20, hipMemcpyDeviceToHost, s));
        URH_HIP(hipMemcpyAsync(h + 6, b.resume, 8, hipMemcpyDeviceToHost, s));
        URH_HIP(hipStreamSynchronize(s));
        const int64_t stop_at = h[3];
        if (stop_at >= nc) break;
        c_from = stop_at;
        use_seed = 1;
        memcpy(&seed_freq, h + 6, 4);                        // resume state: {freq, phase}
        rounds = round + 1;
    }
    h[4] = rounds;
    hipLaunchKernelGGL((k_costas_final<DT, ORDER>), dim3((unsigned)((nc * kNumCkpt + 255) / 256)), dim3(256), 0, s, a, b, nc, K);
    return URHGPU_OK;
}

int launch_costas(urhgpu_ctx *ctx, const void *d_iq, int64_t n, const urhgpu_params *p, float *d_qad, void *scratch) {
    CostasArgs a;
    a.iq = d_iq; a.n = n; a.out = d_qad;
    a.noise_sqrd = p->noise_threshold * p->noise_threshold;
    // :253-254 as the reference's generated code evaluates them (damping = (float)(sqrt(2)/2), bandwidth*bandwidth a float product)
    const float bandwidth = p->costas_loop_bandwidth;
    const float damping = (float)(sqrt(2.0) / 2.0);
    const double den = (1.0 + ((2.0 * (double)damping) * (double)bandwidth)) + (double)(bandwidth * bandwidth);
    a.alpha = (float)(((4.0 * (double)damping) * (double)bandwidth) / den);
    a.beta = (float)(((4.0 * (double)bandwidth) * (double)bandwidth) / den);
    switch (p->dtype) {                                        // :267-283
        case URHGPU_DT_I8: a.scale = 127.5f; a.shift = 0.5f; break;
        case URHGPU_DT_U8: a.scale = 127.5f; a.shift = -127.5f; break;
        case URHGPU_DT_I16: a.scale = 32767.5f; a.shift = 0.5f; break;
        case URHGPU_DT_U16: a.scale = 65535.0f; a.shift = -32767.5f; break;
        case URHGPU_DT_F32: a.scale = 1.0f; a.shift = 0.0f; break;
        default: return URHGPU_ERR_DTYPE;
    }
    // Warm-up: two trajectories in the same lock class contract by about (1 - alpha) per sample once the seeded frequency is close;
    // from a phase error of order 1 down to the last float bit takes ~ 18 / alpha samples (130 at the default bandwidth 0.1), the
    // pull-in before that a few loop time constants: 40 / bandwidth samples, rounded up to a power of two, covers both with margin
    // (512 at 0.1; it was a fixed 1024).  Too short a warm-up only costs time: unmatched chunks are re-speculated / run serially.
    {
        const double want = 40.0 / std::max(1e-3, std::min(1.0, (double)fabsf(bandwidth)));
        int w = 256;
        while (w < want && w < 8192) w *= 2;
        a.warm = w;
    }
    int order = p->mod_order > 0 ? p->mod_order : (1 << p->bits_per_symbol);
    if (order > 4) order = 4;                                  // :285-287
    a.loop_order = order;
    switch (p->dtype) {
        case URHGPU_DT_I8: return launch_costas_dt<URHGPU_DT_I8>(a, scratch, ctx);
        case URHGPU_DT_U8: return launch_costas_dt<URHGPU_DT_U8>(a, scratch, ctx);
        case URHGPU_DT_I16: return launch_costas_dt<URHGPU_DT_I16>(a, scratch, ctx);
        case URHGPU_DT_U16: return launch_costas_dt<URHGPU_DT_U16>(a, scratch, ctx);
        default: return launch_costas_dt<URHGPU_DT_F32>(a, scratch, ctx);
    }
}

// every float with |y| < 120 (bit patterns 0 .. 0x42f00000, both signs): urh_sincosf_fast against urh_sinf / urh_cosf
__global__ __launch_bounds__(256) void k_test_sincosf_fast(unsigned long long *mismatches) {
    unsigned long long bad = 0;
    for (uint64_t u = blockIdx.x * 256ull + threadIdx.x; u < 2ull * 0x42f00000ull; u += (uint64_t)gridDim.x * 256ull) {
        const uint32_t bits = (u < 0x42f00000ull) ? (uint32_t)u : ((uint32_t)(u - 0x42f00000ull) | 0x80000000u);
        const float y = __uint_as_float(bits);
        float sn, cs;
        urh_sincosf_fast(y, &sn, &cs);
        if (__float_as_uint(sn) != __float_as_uint(urh_sinf(y)) || __float_as_uint(cs) != __float_as_uint(urh_cosf(y))) ++bad;
    }
    if (bad) atomicAdd(mismatches, bad);
}
void launch_test_sincosf_fast(unsigned long long *d_mismatches, hipStream_t s) {
    hipLaunchKernelGGL(k_test_sincosf_fast, dim3(256 * 32), dim3(256), 0, s, d_mismatches);
}

}  // namespace urh
