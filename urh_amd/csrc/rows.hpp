// rows.hpp -- what the row stage needs wherever it runs: as kernels of the tail (pulse_table.hip) or inside the hot kernel (demod_runs.hip,
// FUSED instantiation).  The composition of chunks (ResElem), a row's contribution to _ppseq_to_bits (row_value), small wavefront helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "launchers.hpp"
#include "scan.hpp"

namespace urh {

__device__ __forceinline__ int64_t num_symbols_of(int64_t num_samples, int64_t sps) {
    // int(n / sps) (+1 when the fractional part exceeds 0.5), in double like the Python source (:353-358).
    // Below 2^32 the double quotient's floor and the sign of (fraction - 0.5) are those of the exact quotient (the nearest
    // quotients to an integer or to a half differ from it by >= 1 / (2 sps) > 2^-33, against a rounding error < 2^-52 * 2^32):
    // integer form q + (2 r > sps), one 32-bit division.
    if ((((uint64_t)num_samples | (uint64_t)sps) >> 32) == 0) {
        const uint32_t n32 = (uint32_t)num_samples, s32 = (uint32_t)sps;
        const uint32_t q = n32 / s32, r = n32 - q * s32;
        return (int64_t)q + ((2ull * r > s32) ? 1 : 0);
    }
    const double f = (double)num_samples / (double)sps;
    int64_t k = (int64_t)f;
    if (f - (double)k > 0.5) k += 1;
    return k;
}


struct HugeRow { int64_t kb, ob, op, ts, type; };   // a row that expands to more than kHugeBits bits
constexpr int64_t kHugeBits = 4096;
constexpr int kHugeCap = 8192;


struct ResElem {             // effect of a stretch of chunks on the reference's state machine, as a function of the entry state
    int64_t cnt;             // accepted runs, not counting the conditional first one
    int64_t first_pos;       // position of the first stable run (meaningful while nothing else has been accepted)
    int64_t la_pos;          // last run that is accepted whatever the entry state
    uint64_t meta;           // first_state | last_state << 16 | la_state << 32 | has << 48 | la_valid << 49
    __device__ __forceinline__ bool has() const { return (meta >> 48) & 1; }
    __device__ __forceinline__ bool la_valid() const { return (meta >> 49) & 1; }
    __device__ __forceinline__ uint32_t first_state() const { return (uint32_t)(meta & 0xFFFF); }
    __device__ __forceinline__ uint32_t last_state() const { return (uint32_t)((meta >> 16) & 0xFFFF); }
    __device__ __forceinline__ uint32_t la_state() const { return (uint32_t)((meta >> 32) & 0xFFFF); }
};
__device__ __forceinline__ ResElem res_identity() { ResElem e; e.cnt = 0; e.first_pos = -1; e.la_pos = -1; e.meta = 0; return e; }
__device__ __forceinline__ ResElem res_make(uint32_t first_state, int64_t first_pos, uint32_t last_state, int64_t cnt, bool la_valid,
                                            int64_t la_pos, uint32_t la_state) {
    ResElem e;
    e.cnt = cnt; e.first_pos = first_pos; e.la_pos = la_pos;
    e.meta = (uint64_t)(first_state & 0xFFFF) | ((uint64_t)(last_state & 0xFFFF) << 16) | ((uint64_t)(la_state & 0xFFFF) << 32) |
             (1ull << 48) | ((uint64_t)(la_valid ? 1 : 0) << 49);
    return e;
}
// a, then b.  Written as per-field selects: returning one of several structs makes the compiler build them in scratch memory.
__device__ __forceinline__ ResElem res_combine(const ResElem &a, const ResElem &b) {
    const bool ah = a.has(), bh = b.has();
    const bool acc = b.first_state() != a.last_state();        // b's first stable run switches the state machine
    const bool blv = b.la_valid();
    const int64_t both_cnt = a.cnt + b.cnt + (acc ? 1 : 0);
    const int64_t both_la_pos = blv ? b.la_pos : (acc ? b.first_pos : a.la_pos);
    const uint64_t both_la_state = blv ? b.la_state() : (acc ? b.first_state() : a.la_state());
    const uint64_t both_lav = (blv || acc || a.la_valid()) ? 1 : 0;
    const uint64_t both_meta = (uint64_t)a.first_state() | ((uint64_t)b.last_state() << 16) | (both_la_state << 32) | (1ull << 48) | (both_lav << 49);
    ResElem r;
    r.cnt = !ah ? b.cnt : (!bh ? a.cnt : both_cnt);
    r.first_pos = !ah ? b.first_pos : a.first_pos;
    r.la_pos = !ah ? b.la_pos : (!bh ? a.la_pos : both_la_pos);
    r.meta = !ah ? b.meta : (!bh ? a.meta : both_meta);
    return r;
}
__device__ __forceinline__ ResElem res_shfl_up(const ResElem &x, int o) {
    ResElem r;
    r.cnt = __shfl_up(x.cnt, o); r.first_pos = __shfl_up(x.first_pos, o); r.la_pos = __shfl_up(x.la_pos, o);
    r.meta = (uint64_t)__shfl_up((long long)x.meta, o);
    return r;
}
__device__ __forceinline__ ResElem res_shfl_down(const ResElem &x, int o) {
    ResElem r;
    r.cnt = __shfl_down(x.cnt, o); r.first_pos = __shfl_down(x.first_pos, o); r.la_pos = __shfl_down(x.la_pos, o);
    r.meta = (uint64_t)__shfl_down((long long)x.meta, o);
    return r;
}
__device__ __forceinline__ ResElem res_shfl(const ResElem &x, int src) {
    ResElem r;
    r.cnt = __shfl(x.cnt, src); r.first_pos = __shfl(x.first_pos, src); r.la_pos = __shfl(x.la_pos, src);
    r.meta = (uint64_t)__shfl((long long)x.meta, src);
    return r;
}
__device__ __forceinline__ ResElem res_wave_incl_scan(ResElem x, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const ResElem u = res_shfl_up(x, o);
        if (lane >= o) x = res_combine(u, x);
    }
    return x;
}
// the stable runs of one chunk as an element (after chunk_stable has settled its pending run)
__device__ __forceinline__ ResElem res_of_chunk(int cnt, uint32_t c_first_state, uint32_t c_last_state, uint32_t c_pend_state, int64_t c_last_pos,
                                                int64_t c_pend_pos, int pend_stable) {
    const bool pend_in = pend_stable && (cnt == 0 || c_pend_state != c_last_state);   // the pending run is a further stable run
    const int k = cnt + (pend_in ? 1 : 0);
    if (k == 0) return res_identity();
    const uint32_t first_state = cnt > 0 ? c_first_state : c_pend_state;
    const int64_t last_pos = pend_in ? c_pend_pos : c_last_pos;
    const uint32_t last_state = pend_in ? c_pend_state : c_last_state;
    // k == 1: the only stable run IS the first one (cnt == 1: last_pos is record 0's position)
    return res_make(first_state, k == 1 ? last_pos : -1, last_state, k - 1, k >= 2, last_pos, last_state);
}


// What one row contributes to _ppseq_to_bits (ProtocolAnalyzer.py:346-401): v[0] bits, v[1] long pause, v[2] samples, v[3] data row
__device__ __forceinline__ VecK<4> row_value(int64_t type, int64_t len, bool global_row0, const BitsParams &bp) {
    VecK<4> v; v.zero();
    v.v[2] = len;
    if (type == kRowAbsorbed) return v;
    if (global_row0 && type == -1) return v;             // "Starts with Pause" (:346-348): only seeds total_samples
    const int64_t ns = num_symbols_of(len, bp.sps);
    if (type == -1) {
        if (ns <= bp.pause_threshold || bp.pause_threshold == 0) v.v[0] = (ns > 0) ? ns * bp.bps : 0;
        else v.v[1] = 1;
    } else {
        v.v[0] = (ns > 0) ? ns * bp.bps : 0;
        v.v[3] = (ns > 0) ? 1 : 0;
    }
    return v;
}

struct HugeRef { int64_t tile, row; };                   // a row of more than kHugeBits bits, found while the rows were emitted

__device__ __forceinline__ int64_t lane_bcast(int64_t v, int src) {          // src wave-uniform
    const int lo = __builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)v, src), hi = __builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), src);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}
__device__ __forceinline__ int lane_bcast(int v, int src) { return __builtin_amdgcn_readlane(v, src); }


}  // namespace urh
