// pulse_table.hip -- from per-chunk accepted-run records to the reference's outputs:
//   * the pulse table ("ppseq", int64[P][2]) of signal_functions.grab_pulse_lens
//       /root/reference/src/urh/cythonext/signal_functions.pyx:392-495
//   * bits / pauses / bit_sample_pos of ProtocolAnalyzer._ppseq_to_bits
//       /root/reference/src/urh/signalprocessing/ProtocolAnalyzer.py:323-414
// Integer-only work on ~N/samples_per_symbol items: a few small kernels, all device resident,
// element counts passed through device memory so that nothing synchronises with the host.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "common.hpp"
#include "compact.hpp"
#include "launchers.hpp"
#include "runs.hpp"
#include "scan.hpp"

namespace urh {

// =====================================================================================================
// Resolve stage: settle everything a chunk could not decide alone.
//   - does a chunk's trailing short run grow past `tol` in the following chunks?  (ChunkInfo.lead)
//   - is a chunk's first stable run "accepted", i.e. different from the last stable run before it?
//   - global index of each chunk's first accepted run, and the accepted run preceding it
// and write the last pulse-table row (signal_functions.pyx:485-493).
// Two per-chunk kernels (one thread per chunk, any grid) separated by two single-workgroup scans
// over small int arrays: "last chunk before me that has X" is an exclusive max-scan of (c if X else -1).
// =====================================================================================================
#ifndef URH_RESOLVE_BLOCK
#define URH_RESOLVE_BLOCK 256
#endif
#ifndef URH_RESOLVE_ITEMS
#define URH_RESOLVE_ITEMS 4
#endif
constexpr int kResolveBlock = URH_RESOLVE_BLOCK;

__host__ __device__ static inline int64_t resolve_blocks(int64_t n_chunks) { return (n_chunks + kResolveBlock - 1) / kResolveBlock; }

size_t resolve_scratch_bytes(int64_t n_chunks) {
    return (size_t)n_chunks * (4 * 4 + 2 * 8) + (size_t)(resolve_blocks(n_chunks) + 1) * (4 + 4 + 8) + 10 * 256;
}

ResolveScratch resolve_scratch_carve(void *mem, int64_t n_chunks) {
    char *p = (char *)mem;
    auto take = [&](size_t bytes) { char *r = p; p += (bytes + 255) & ~size_t(255); return r; };
    const int64_t nb = resolve_blocks(n_chunks) + 1;
    ResolveScratch sc;
    sc.out_cnt = (int64_t *)take((size_t)n_chunks * 8);
    sc.out_off = (int64_t *)take((size_t)n_chunks * 8);
    sc.has_stable = (int32_t *)take((size_t)n_chunks * 4);
    sc.prev_stable = (int32_t *)take((size_t)n_chunks * 4);
    sc.has_acc = (int32_t *)take((size_t)n_chunks * 4);
    sc.prev_acc = (int32_t *)take((size_t)n_chunks * 4);
    sc.blk_stable = (int32_t *)take((size_t)nb * 4);
    sc.blk_acc = (int32_t *)take((size_t)nb * 4);
    sc.blk_cnt = (int64_t *)take((size_t)nb * 8);
    return sc;
}

// R1: per chunk.
__device__ __forceinline__ int32_t chunk_stable(const ResolveArgs &a, int64_t c) {
    ChunkInfo *ch = a.chunks;
    int ps = 0;
    const int64_t pend_pos = ch[c].pend_pos;
    if (pend_pos >= 0) {
        int64_t len = ch[c].start + ch[c].len - pend_pos;              // run length inside chunk c
        bool hit = false;                                              // ended by a run boundary
        for (int64_t u = c + 1; len <= a.tol && u < a.n_chunks; ++u) {
            const int64_t lead = ch[u].lead;
            len += lead;
            if (lead < ch[u].len) { hit = true; break; }
        }
        ps = len > a.tol;
        if (a.local_pass && !ps && !hit) a.aux->open_chunk = (int32_t)c;   // at most one chunk: everything after it is one run
    }
    ch[c].pend_stable = ps;
    const bool has = ps || ch[c].cnt > 0;
    if (a.local_pass) {                                     // only the shard summary needs them (resolve_finish)
        if (has) atomicMin(&a.aux->first_stable, (int32_t)c);
        if (ch[c].lead < ch[c].len) atomicMin(&a.aux->first_nonlead, (int32_t)c);
    }
    return has ? (int32_t)c : -1;
}

// Exclusive scans inside ONE workgroup, one element per thread: ex_m = max of the elements before mine (or -1),
// ex_s = their sum; tot_* = over the whole workgroup.  The scans over the whole table are two-level: every workgroup
// scans its own kResolveBlock chunks and publishes its totals (blk_*), and whoever needs a global value adds the
// totals of the workgroups before it (a handful: 16 for a 1 GiB capture) -- no workgroup ever walks the whole table.
template <bool WITH_SUM>
__device__ __forceinline__ void block_local_scan(int32_t m, int64_t v, int32_t &ex_m, int64_t &ex_s, int32_t &tot_m, int64_t &tot_s) {
    __shared__ int32_t s_m[kResolveBlock / 64];
    __shared__ int64_t s_s[kResolveBlock / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int32_t im = m; int64_t is = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int32_t um = __shfl_up(im, o); const int64_t us = WITH_SUM ? __shfl_up(is, o) : 0;
        if (lane >= o) { im = max(im, um); is += us; }
    }
    __syncthreads();                                       // s_m / s_s may still be read by a previous call
    if (lane == 63) { s_m[wave] = im; s_s[wave] = is; }
    __syncthreads();
    int32_t bm = -1, tm = -1; int64_t bs = 0, ts = 0;
    for (int w = 0; w < kResolveBlock / 64; ++w) {
        if (w < wave) { bm = max(bm, s_m[w]); bs += s_s[w]; }
        tm = max(tm, s_m[w]); ts += s_s[w];
    }
    int32_t em = __shfl_up(im, 1); int64_t es = WITH_SUM ? __shfl_up(is, 1) : 0;
    if (lane == 0) { em = -1; es = 0; }
    ex_m = max(em, bm); ex_s = es + bs; tot_m = tm; tot_s = ts;
}

// max / sum of the totals of the workgroups before workgroup b (every thread gets the result)
template <bool WITH_SUM>
__device__ __forceinline__ void blocks_before(const int32_t *blk_m, const int64_t *blk_s, int64_t b, int32_t &pm, int64_t &ps) {
    __shared__ int32_t s_pm;
    __shared__ int64_t s_ps;
    __syncthreads();                                       // a previous call's result may still be read
    if (threadIdx.x < 64) {
        int32_t m = -1; int64_t v = 0;
        for (int64_t u = threadIdx.x; u < b; u += 64) { m = max(m, blk_m[u]); if (WITH_SUM) v += blk_s[u]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { m = max(m, __shfl_down(m, o)); if (WITH_SUM) v += __shfl_down(v, o); }
        if (threadIdx.x == 0) { s_pm = m; s_ps = v; }
    }
    __syncthreads();
    pm = s_pm; ps = s_ps;
}

__device__ __forceinline__ uint32_t chunk_last_stable_state(const ChunkInfo &ci) { return ci.pend_stable ? ci.pend_state : ci.last_state; }

// R2: per chunk: acceptance of the tentative first record / the pending run, counts.
__device__ __forceinline__ void chunk_accept(const ResolveArgs &a, int64_t c, int32_t ip, int64_t &out_cnt, int32_t &has_acc) {
    ChunkInfo *ch = a.chunks;
    const uint32_t prev_state = (ip < 0) ? (a.local_pass ? 0xFFFFu : (uint32_t)ch[0].init_state) : chunk_last_stable_state(ch[ip]);
    const int cnt = ch[c].cnt;
    const int ps = ch[c].pend_stable;
    const int first_acc = (cnt > 0) && (ch[c].first_state != prev_state);
    const uint32_t before_pend = (cnt > 0) ? ch[c].last_state : prev_state;
    const int pend_acc = ps && (ch[c].pend_state != before_pend);
    ch[c].first_acc = first_acc;
    ch[c].pend_acc = pend_acc;
    out_cnt = (cnt > 0 ? cnt - 1 + first_acc : 0) + pend_acc;
    has_acc = (pend_acc || cnt >= 2 || (cnt == 1 && first_acc)) ? (int32_t)c : -1;
}

// position / state of the last accepted run of chunk ci (which contributes at least one)
__device__ __forceinline__ void chunk_last_acc(const ChunkInfo &ci, int64_t &pos, uint32_t &state) {
    if (ci.pend_acc) { pos = ci.pend_pos; state = ci.pend_state; }
    else { pos = ci.last_pos; state = ci.last_state; }
}

// Totals and the final row (signal_functions.pyx:485-493; skipped when the table already has n rows, :487): one thread,
// from the workgroup totals.  out_off values it needs are rebuilt from the workgroup-local offsets (sc.out_cnt).
struct ResolveTotals {       // sums / maxima over the workgroup totals, computed by the whole of workgroup 0 (k_resolve_c)
    int32_t last_c, last_stable;
    int64_t P;               // accepted runs in the whole table
    int64_t before_first;    // accepted runs in the resolve workgroups before the one that holds chunk_first
    int64_t before_end;      //   ... before the one that holds chunk_first + n_local
};

__device__ __forceinline__ void resolve_finish(const ResolveArgs &a, const ResolveTotals &t) {
    const int32_t last_c = t.last_c;
    const int64_t P = t.P;
    a.aux->last_stable = t.last_stable;
    auto global_off = [&](int64_t c, int64_t before) { return a.sc.out_cnt[c] + before; };   // out_off[c] from the local offset
    {
        *a.d_n_acc = P;
        if (a.local_pass) {
            // the ChunkInfo that stands for this whole shard in the other ranks' tables
            const ChunkInfo *ch = a.chunks;
            const ResolveAux ax = *a.aux;
            ChunkInfo s;
            s.start = ch[0].start; s.len = a.n_total;
            s.lead = (ax.first_nonlead >= a.n_chunks) ? a.n_total : ch[ax.first_nonlead].start - ch[0].start + ch[ax.first_nonlead].lead;
            s.pend_pos = -1; s.pend_state = 0;
            if (ax.open_chunk < a.n_chunks) { s.pend_pos = ch[ax.open_chunk].pend_pos; s.pend_state = ch[ax.open_chunk].pend_state; }
            s.cnt = (int32_t)P;
            s.first_state = 0xFFFFu;
            if (ax.first_stable < a.n_chunks) {
                const ChunkInfo &f = ch[ax.first_stable];
                s.first_state = f.cnt > 0 ? f.first_state : f.pend_state;
            }
            s.last_state = (ax.last_stable < 0) ? (uint16_t)0xFFFFu : (uint16_t)chunk_last_stable_state(ch[ax.last_stable]);
            int64_t lpos = 0; uint32_t lst = 0;
            if (last_c >= 0) chunk_last_acc(ch[last_c], lpos, lst);
            s.last_pos = lpos;
            s.init_state = ch[0].init_state;
            s.first_acc = 0; s.pend_acc = 0; s.pend_stable = 0; s.pad = 0;
            *a.summary_out = s;
            return;
        }
        // this GPU's rows are global rows [row_base, row_end) (+ the table's last row on the last GPU)
        const int64_t row_base = global_off(a.chunk_first, t.before_first);
        const int64_t row_end = (a.chunk_first + a.n_local < a.n_chunks) ? global_off(a.chunk_first + a.n_local, t.before_end) : P;
        int64_t n_rows = row_end - row_base;
        if (P < a.n_total && a.write_last_row) {
            const int64_t o = P - row_base;
            n_rows = o + 1;
            int64_t fpos = -1; uint32_t fstate = a.chunks[0].init_state;
            if (last_c >= 0) chunk_last_acc(a.chunks[last_c], fpos, fstate);
            if (a.rows != nullptr && o < a.cap_rows) {
                const int64_t len = (P == 0) ? (a.n_total - a.tol) : (a.n_total - 1 - fpos - a.tol);
                a.rows[2 * o] = (int64_t)fstate - 1;
                a.rows[2 * o + 1] = len;
            }
            if (o == 0 && a.d_ts_carry) *a.d_ts_carry = (P == 0) ? 0 : fpos + 1;
        }
        *a.d_n_rows_needed = n_rows;
        *a.d_n_rows = (a.rows != nullptr && n_rows > a.cap_rows) ? a.cap_rows : n_rows;
    }
}

// The resolve stage as three launches over the table, kResolveBlock chunks per workgroup, no workgroup waiting for another:
//   k_resolve_a  per chunk: does its trailing short run turn stable?  workgroup-local "last chunk with a stable run before me"
//   k_resolve_b  per chunk: acceptance of its first / pending run, counts; workgroup-local offsets and "last chunk that
//                contributes an accepted run before me"
//   k_resolve_c  adds the totals of the workgroups before: out_off, prev_acc; thread 0 of workgroup 0: totals, last row
// aux is persistent context memory that holds kAuxNone / -1 between passes (k_resolve_c restores it).
__global__ __launch_bounds__(kResolveBlock) void k_resolve_a(const ResolveArgs a) {
    const int64_t c = (int64_t)blockIdx.x * kResolveBlock + threadIdx.x;
    const int32_t hs = (c < a.n_chunks) ? chunk_stable(a, c) : -1;
    int32_t ex, tot; int64_t d0, d1;
    block_local_scan<false>(hs, 0, ex, d0, tot, d1);
    if (c < a.n_chunks) a.sc.prev_stable[c] = ex;
    if (threadIdx.x == 0) a.sc.blk_stable[blockIdx.x] = tot;
}
__global__ __launch_bounds__(kResolveBlock) void k_resolve_b(const ResolveArgs a) {
    const int64_t c = (int64_t)blockIdx.x * kResolveBlock + threadIdx.x;
    int32_t pm; int64_t dummy;
    blocks_before<false>(a.sc.blk_stable, nullptr, blockIdx.x, pm, dummy);
    int64_t cnt = 0; int32_t ha = -1;
    if (c < a.n_chunks) chunk_accept(a, c, max(a.sc.prev_stable[c], pm), cnt, ha);
    int32_t ex_m, tot_m; int64_t ex_s, tot_s;
    block_local_scan<true>(ha, cnt, ex_m, ex_s, tot_m, tot_s);
    if (c < a.n_chunks) { a.sc.out_cnt[c] = ex_s; a.sc.has_acc[c] = ex_m; }     // workgroup-local exclusive values
    if (threadIdx.x == 0) { a.sc.blk_acc[blockIdx.x] = tot_m; a.sc.blk_cnt[blockIdx.x] = tot_s; }
}
__global__ __launch_bounds__(kResolveBlock) void k_resolve_c(const ResolveArgs a) {
    const int64_t c = (int64_t)blockIdx.x * kResolveBlock + threadIdx.x;
    int32_t pm; int64_t ps;
    blocks_before<true>(a.sc.blk_acc, a.sc.blk_cnt, blockIdx.x, pm, ps);
    if (c < a.n_chunks) { a.sc.out_off[c] = a.sc.out_cnt[c] + ps; a.sc.prev_acc[c] = max(a.sc.has_acc[c], pm); }
    if (blockIdx.x == 0) {                                 // totals, last row: workgroup 0 sums the workgroup totals together
        const int64_t nb = resolve_blocks(a.n_chunks);
        ResolveTotals t;
        int64_t dummy;
        blocks_before<true>(a.sc.blk_acc, a.sc.blk_cnt, nb, t.last_c, t.P);
        blocks_before<false>(a.sc.blk_stable, nullptr, nb, t.last_stable, dummy);
        int32_t dm;
        blocks_before<true>(a.sc.blk_acc, a.sc.blk_cnt, a.chunk_first / kResolveBlock, dm, t.before_first);
        const int64_t ce = (a.chunk_first + a.n_local < a.n_chunks) ? a.chunk_first + a.n_local : 0;
        blocks_before<true>(a.sc.blk_acc, a.sc.blk_cnt, ce / kResolveBlock, dm, t.before_end);
        if (threadIdx.x == 0) {
            resolve_finish(a, t);
            a.aux->first_nonlead = kAuxNone; a.aux->open_chunk = kAuxNone; a.aux->first_stable = kAuxNone; a.aux->last_stable = -1;
        }
    }
}

// =====================================================================================================
// k_emit_rows: one wavefront per chunk turns accepted run starts into pulse-table rows
//   row g = (state of accepted run g-1, start_g - start_{g-1}); row 0 = (init state, start_0 + 1)
//   ASK: pause rows shorter than samples_per_symbol are relabelled 0 (signal_functions.pyx:471-473).
// =====================================================================================================
__device__ __forceinline__ void emit_chunk_rows(const EmitArgs &a, int64_t c, int64_t out_off, int64_t row_base, int32_t ip) {
    const ChunkInfo ci = a.chunks[c];
    const uint64_t *slab = a.slab + (c - a.chunk_first) * a.slab_stride;
    const int skip = (ci.cnt > 0 && !ci.first_acc) ? 1 : 0;
    const int64_t from_slab = (ci.cnt > 0) ? ci.cnt - skip : 0;
    const int64_t total = from_slab + ci.pend_acc;
    if (total == 0) return;
    int64_t prev_pos = -1; uint32_t prev_state = a.chunks[0].init_state;   // before the very first accepted run
    if (ip >= 0) chunk_last_acc(a.chunks[ip], prev_pos, prev_state);
    for (int64_t j = threadIdx.x; j < total; j += blockDim.x) {
        int64_t pos;
        if (j < from_slab) pos = rec_pos(slab[j + skip]);
        else pos = ci.pend_pos;
        int64_t ppos; uint32_t pst;
        if (j == 0) { ppos = prev_pos; pst = prev_state; }
        else { const uint64_t r = slab[j - 1 + skip]; ppos = rec_pos(r); pst = rec_state(r); }
        const int64_t g = out_off + j;
        const int64_t len = (g == 0) ? pos + 1 : pos - ppos;
        int64_t state = (int64_t)pst - 1;
        if (a.is_ask && state == -1 && len < a.sps) state = 0;
        const int64_t o = g - row_base;
        if (o >= 0 && o < a.cap_rows) { a.rows[2 * o] = state; a.rows[2 * o + 1] = len; }
        if (o == 0 && a.d_ts_carry) *a.d_ts_carry = (g == 0) ? 0 : ppos + 1;
    }
}

__global__ __launch_bounds__(64) void k_emit_rows(const EmitArgs a) {
    const int64_t c = a.chunk_first + blockIdx.x;
    emit_chunk_rows(a, c, a.sc.out_off[c], a.sc.out_off[a.chunk_first], a.sc.prev_acc[c]);
}

// Single-GPU captures (the table holds this GPU's chunks only, starting at row 0): k_resolve_c folded into the row pass --
// every chunk's wavefront adds up the totals of the resolve workgroups before its own (at most a few hundred values from L2),
// workgroup 0 also writes the totals and the table's last row.  One launch less on the latency chain of the tail.
__global__ __launch_bounds__(64) void k_emit_rows_fused(const EmitArgs a, const ResolveArgs r) {
    const int64_t c = blockIdx.x;
    const int lane = threadIdx.x;
    const int64_t b = c / kResolveBlock;
    int32_t pm = -1; int64_t ps = 0;
    for (int64_t u = lane; u < b; u += 64) { pm = max(pm, r.sc.blk_acc[u]); ps += r.sc.blk_cnt[u]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { pm = max(pm, __shfl_xor(pm, o)); ps += __shfl_xor(ps, o); }
    emit_chunk_rows(a, c, r.sc.out_cnt[c] + ps, 0, max(r.sc.has_acc[c], pm));
    if (c == 0) {
        const int64_t nb = resolve_blocks(r.n_chunks);
        ResolveTotals t;
        t.last_c = -1; t.last_stable = -1; t.P = 0; t.before_first = 0; t.before_end = 0;
        for (int64_t u = lane; u < nb; u += 64) { t.last_c = max(t.last_c, r.sc.blk_acc[u]); t.last_stable = max(t.last_stable, r.sc.blk_stable[u]); t.P += r.sc.blk_cnt[u]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            t.last_c = max(t.last_c, __shfl_xor(t.last_c, o)); t.last_stable = max(t.last_stable, __shfl_xor(t.last_stable, o));
            t.P += __shfl_xor(t.P, o);
        }
        if (lane == 0) {
            resolve_finish(r, t);
            r.aux->first_nonlead = kAuxNone; r.aux->open_chunk = kAuxNone; r.aux->first_stable = kAuxNone; r.aux->last_stable = -1;
        }
    }
}

// =====================================================================================================
// ASK only: merge adjacent rows of equal state (signal_functions.pyx:475-480, 488-489).
// =====================================================================================================
struct MergeLoad {
    const int64_t *rows;
    __device__ VecK<2> operator()(int64_t i) const {
        VecK<2> v;
        v.v[0] = (i == 0 || rows[2 * i] != rows[2 * (i - 1)]) ? 1 : 0;   // head of a group
        v.v[1] = rows[2 * i + 1];
        return v;
    }
};
struct MergeStore {
    const int64_t *rows;
    const int64_t *d_n;
    int64_t *grp_state;      // [groups]
    int64_t *grp_end_sum;    // [groups] inclusive length sum at the group's last row
    __device__ void operator()(int64_t i, const VecK<2> &val, const VecK<2> &ex) const {
        const int64_t n = *d_n;
        const int64_t g = ex.v[0] + val.v[0] - 1;                 // group index of row i
        if (val.v[0]) grp_state[g] = rows[2 * i];
        const bool last = (i + 1 == n) || (rows[2 * (i + 1)] != rows[2 * i]);
        if (last) grp_end_sum[g] = ex.v[1] + val.v[1];
    }
};
__global__ void k_merge_finish(const VecK<2> *grand_total, const int64_t *grp_state, const int64_t *grp_end_sum,
                               int64_t *rows_out, int64_t cap_rows, int64_t *d_n_rows_out) {
    const int64_t groups = grand_total->v[0];
    for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < groups; g += (int64_t)gridDim.x * blockDim.x) {
        if (g < cap_rows) {
            rows_out[2 * g] = grp_state[g];
            rows_out[2 * g + 1] = grp_end_sum[g] - (g ? grp_end_sum[g - 1] : 0);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *d_n_rows_out = groups;
}

// =====================================================================================================
// _ppseq_to_bits on device  (ProtocolAnalyzer.py:323-414)
//   row kinds: D data row with >= 1 symbol; S pause row with <= pause_threshold symbols (-> zero bits);
//              L long pause (closes a message when data was seen, else discards what was gathered)
//   groups   : stretches of rows between L rows; a group with at least one D row is a message.
// =====================================================================================================
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

struct RowInfo {             // per-row scan results kept for the expansion kernel
    int64_t bit_prefix;      // bits before this row (all rows, kept or not)
    int64_t ts_prefix;       // total_samples before this row
    int64_t kbits;           // bits this row contributes (0 for L rows); 64 bits: one sample per symbol on a multi-GiB capture
    int32_t group;           // long pauses before this row
    int32_t pad;
};

struct GroupInfo {
    int64_t bits_end;        // bit_prefix at the closing row (== bits before + inside the group)
    int64_t data_end;        // D rows before the closing row
    int64_t ts_close;        // total_samples before the closing L row (or at the very end)
    int64_t pause;           // pauses[] entry when this group is a message
    int32_t closed;          // 1: closed by an L row, 0: the trailing group
    int32_t pad;
};

struct BitsLoad {
    const int64_t *rows;
    const int64_t *d_n_rows;
    BitsParams bp;
    int64_t *d_n_groups0;    // single-GPU: the group count is written by BitsStore at the last row; zero rows -> zero groups
    __device__ void on_start(int64_t n) const { if (d_n_groups0 && n <= 0) *d_n_groups0 = 0; }
    // i counts rows AFTER the skipped leading pause row? no: all rows; row 0 is neutralised when it is a pause
    __device__ VecK<4> operator()(int64_t i) const {
        VecK<4> v; v.zero();
        const longlong2 row = *(const longlong2 *)(rows + 2 * i);          // one 16-byte load
        const int64_t type = row.x, len = row.y;
        v.v[2] = len;
        if (type == kRowAbsorbed) return v;              // merged into the previous GPU's last row: length only
        const bool global_row0 = (i == 0) && (bp.d_row_base == nullptr || *bp.d_row_base == 0);
        if (global_row0 && type == -1) return v;         // "Starts with Pause" (:346-348): only seeds total_samples
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
};

struct BitsStore {
    const int64_t *rows;
    const int64_t *d_n_rows;
    RowInfo *info;
    GroupInfo *groups;
    const int64_t *d_ts_carry;   // sharded captures: total_samples before this GPU's first row (nullptr: 0)
    const int64_t *d_absorbed;
    int64_t *d_n_groups;         // single-GPU: number of groups = long pauses + 1 (sharded: GroupCountFinal, with the flags)
    __device__ void operator()(int64_t i, const VecK<4> &val, const VecK<4> &ex) const {
        const int64_t n = *d_n_rows;
        const int64_t ts0 = d_ts_carry ? *d_ts_carry : 0;
        RowInfo ri;
        ri.bit_prefix = ex.v[0]; ri.ts_prefix = ts0 + ex.v[2]; ri.group = (int32_t)ex.v[1]; ri.kbits = val.v[0]; ri.pad = 0;
        info[i] = ri;
        if (val.v[1]) {                                   // L row closes group ex.v[1]
            GroupInfo g;
            g.bits_end = ex.v[0]; g.data_end = ex.v[3]; g.ts_close = ts0 + ex.v[2]; g.pause = val.v[2]; g.closed = 1; g.pad = 0;
            groups[ex.v[1]] = g;
        }
        if (i + 1 == n) {                                 // trailing group (index = total L count)
            GroupInfo g;
            g.bits_end = ex.v[0] + val.v[0]; g.data_end = ex.v[3] + val.v[3]; g.ts_close = ts0 + ex.v[2] + val.v[2];
            g.pause = (rows[2 * i] == -1) ? rows[2 * i + 1] : 0;      // :411
            if (rows[2 * i] == kRowAbsorbed && d_absorbed && *d_absorbed >= 0) g.pause = *d_absorbed;
            g.closed = 0; g.pad = 0;
            groups[ex.v[1] + val.v[1]] = g;
            if (d_n_groups) *d_n_groups = ex.v[1] + val.v[1] + 1;
        }
    }
};

// groups -> messages: exclusive scan over (messages closed here, kept bits, kept positions).
// A group is KEPT when it holds a data row -- on this GPU or, for the groups that continue across a shard
// boundary (the first one and the trailing one), on the neighbouring GPUs (d_extra) -- and is a MESSAGE of this
// GPU when it is kept and closes here (by a long pause, or by the end of the capture on the last GPU).
struct GroupLoad {
    const GroupInfo *groups;
    const int64_t *d_n_groups;
    const int32_t *d_extra;
    int is_last_rank;
    int write_pos;
    int tentative_open = 0;  // a segment of a streamed pass before the last one: the group that is still open counts as kept -- whether
                             // it holds data is only known when it closes; its rows are expanded where its bits WOULD go (see SegFinal)
    __device__ bool kept(int64_t g, const GroupInfo &gi) const {
        const int64_t d0 = g ? groups[g - 1].data_end : 0;
        if (gi.data_end - d0 > 0) return true;
        if (tentative_open && !gi.closed) return true;
        if (!d_extra) return false;
        return (g == 0 && d_extra[0]) || (!gi.closed && d_extra[1]);
    }
    __device__ VecK<3> operator()(int64_t g) const {
        VecK<3> v; v.zero();
        const GroupInfo gi = groups[g];
        if (kept(g, gi)) {
            const int64_t b0 = g ? groups[g - 1].bits_end : 0;
            const bool closes = gi.closed || is_last_rank;
            v.v[0] = closes ? 1 : 0;
            v.v[1] = gi.bits_end - b0;
            v.v[2] = write_pos ? (gi.bits_end - b0) + (gi.closed ? 2 : (is_last_rank ? 1 : 0)) : 0;
        }
        return v;
    }
};
struct GroupOut {            // per group, for the expansion kernel
    int64_t bits_start;      // bit_prefix at the group's first row
    int64_t out_bits;        // offset of the group's bits in bits[]
    int64_t out_pos;         // offset of the group's positions in pos[]
    int32_t is_msg;          // kept
    int32_t pad;
};
struct GroupStore {
    GroupLoad ld;
    GroupOut *gout;
    int64_t *msg_off, *pauses, *pos_off, *pos;
    int64_t cap_msg, cap_pos;
    uint32_t *h_pos = nullptr;   // direct passes that ship positions: the compact blob's pos32 section in pinned HOST memory (stored as they are written)
    __device__ void operator()(int64_t g, const VecK<3> &val, const VecK<3> &ex) const {
        const GroupInfo gi = ld.groups[g];
        const bool kept = ld.kept(g, gi);
        GroupOut o;
        o.bits_start = g ? ld.groups[g - 1].bits_end : 0;
        o.out_bits = ex.v[1]; o.out_pos = ex.v[2]; o.is_msg = kept ? 1 : 0; o.pad = 0;
        gout[g] = o;
        if (val.v[0]) {                                       // a message of this GPU: END offsets at [m + 1]
            const int64_t m = ex.v[0];
            if (m < cap_msg) {
                msg_off[m + 1] = ex.v[1] + val.v[1];
                pos_off[m + 1] = ex.v[2] + val.v[2];
                pauses[m] = gi.pause;
            }
        }
        if (kept && ld.write_pos) {
            const int64_t p0 = ex.v[2] + val.v[1];            // sentinels follow the per-bit positions
            if (gi.closed) {
                if (p0 + 1 < cap_pos) {
                    pos[p0] = gi.ts_close; pos[p0 + 1] = gi.ts_close + gi.pause;
                    if (h_pos) { h_pos[p0] = (uint32_t)gi.ts_close; h_pos[p0 + 1] = (uint32_t)(gi.ts_close + gi.pause); }
                }
            } else if (ld.is_last_rank) {
                if (p0 < cap_pos) { pos[p0] = gi.ts_close; if (h_pos) h_pos[p0] = (uint32_t)gi.ts_close; }
            }
        }
    }
};

// Runs once after the group scan: counts = {n_rows, n_msg, n_bits, n_pos}; msg_off[0] = pos_off[0] = 0
// (msg_off[m + 1] / pos_off[m + 1] = END of message m, written by GroupStore).
struct BitsCountsFinal {
    const int64_t *d_n_rows;
    int64_t *msg_off, *pos_off, *counts;
    int32_t *huge_count;
    const int64_t *d_rows_needed;      // rows the pulse table needed before clamping to cap_rows (nullptr: unknown)
    int64_t *h_counts;                 // pinned host memory that receives the counts as well (nullptr: none)
    __device__ void operator()(const VecK<3> &grand) const {
        *huge_count = 0;
        const int64_t needed = d_rows_needed ? *d_rows_needed : *d_n_rows;
        counts[4] = needed;
        const int64_t n_rows = *d_n_rows;
        VecK<3> g; g.zero();
        if (n_rows > 0) g = grand;
        counts[0] = n_rows; counts[1] = g.v[0]; counts[2] = g.v[1]; counts[3] = g.v[2];
        if (h_counts) { h_counts[0] = n_rows; h_counts[1] = g.v[0]; h_counts[2] = g.v[1]; h_counts[3] = g.v[2]; h_counts[4] = needed; }
        msg_off[0] = 0; pos_off[0] = 0;
    }
};

// bit k of a row that repeats the symbol `type` (bits_per_symbol bits, most significant first): number_to_bits(state) * n (:390-392).
// k % bps in 32 bits where k fits (a 64-bit remainder is a 50-instruction sequence per bit); bps == 1: always bit 0.
__device__ __forceinline__ uint8_t symbol_bit(int64_t type, int64_t k, int bps) {
    if (type < 0) return 0;
    if (bps == 1) return (uint8_t)(type & 1);
    const int r = (k < 0x7fffffff) ? (int)((uint32_t)k % (uint32_t)bps) : (int)(k % bps);
    return (uint8_t)((type >> (bps - 1 - r)) & 1);
}

struct HugeRow { int64_t kb, ob, op, ts, type; };   // a row that expands to more than kHugeBits bits
constexpr int64_t kHugeBits = 4096;
constexpr int kHugeCap = 8192;

struct ExpandArgs {
    const int64_t *rows;
    const int64_t *d_n_rows;
    const RowInfo *info;
    const GroupOut *gout;
    uint8_t *bits;
    int64_t cap_bits;
    int64_t *pos;
    int64_t cap_pos;
    BitsParams bp;
    HugeRow *huge;           // work list for k_expand_huge (long constant stretches: one row, thousands of bits)
    int32_t *huge_count;     // zero before k_expand_bits (BitsCountsFinal), consumed by k_expand_huge
    int32_t huge_cap;
};

// one thread per row; rows with many bits are expanded cooperatively by the whole wavefront
__global__ __launch_bounds__(256) void k_expand_bits(const ExpandArgs a) {
    const int64_t n = *a.d_n_rows;
    const int lane = threadIdx.x & 63;
    // the grid is sized for a typical row count, not for the table's capacity (empty workgroups are not free): stride loop
    for (int64_t wave_base = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) - lane; wave_base < n; wave_base += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = wave_base + lane;
    int64_t kb = 0, ob = 0, op = 0, ts = 0, type = 0;
    if (i < n) {
        const RowInfo ri = a.info[i];
        if (ri.kbits > 0) {
            const GroupOut go = a.gout[ri.group];
            if (go.is_msg) {
                kb = ri.kbits;
                ob = go.out_bits + (ri.bit_prefix - go.bits_start);
                op = go.out_pos + (ri.bit_prefix - go.bits_start);
                ts = ri.ts_prefix;
                type = a.rows[2 * i];
            }
        }
    }
    const int bps = (int)a.bp.bps;
    constexpr int kShort = 16;
    if (kb > 0 && kb <= kShort) {
        for (int64_t k = 0; k < kb; ++k) {
            const uint8_t b = symbol_bit(type, k, bps);
            if (ob + k < a.cap_bits) a.bits[ob + k] = b;
            if (a.bp.write_pos && op + k < a.cap_pos) a.pos[op + k] = ts + k * a.bp.samples_per_bit;
        }
    }
    unsigned long long big = __ballot(kb > kShort);
    while (big) {
        const int src = __builtin_ctzll(big);
        big &= big - 1;
        const int64_t kb_s = __shfl(kb, src), ob_s = __shfl(ob, src), op_s = __shfl(op, src), ts_s = __shfl(ts, src),
                      ty_s = __shfl(type, src);
        if (kb_s > kHugeBits) {                              // too long for one wavefront: hand it to the whole grid
            int slot = 0;
            if (lane == 0) slot = atomicAdd(a.huge_count, 1);
            slot = __shfl(slot, 0);
            if (slot < a.huge_cap) {
                if (lane == 0) a.huge[slot] = HugeRow{kb_s, ob_s, op_s, ts_s, ty_s};
                continue;
            }
        }
        for (int64_t k = lane; k < kb_s; k += 64) {
            const uint8_t b = symbol_bit(ty_s, k, bps);
            if (ob_s + k < a.cap_bits) a.bits[ob_s + k] = b;
            if (a.bp.write_pos && op_s + k < a.cap_pos) a.pos[op_s + k] = ts_s + k * a.bp.samples_per_bit;
        }
    }
    }
}

// rows of more than kHugeBits bits: blockIdx.y strides over the work list, the blocks along x share one row -- a capture
// that is one long constant stretch per message (an unmodulated carrier: one row per message, thousands of bits each) keeps
// every row's blocks busy at once instead of walking the list row by row
constexpr unsigned kHugeGridX = 64, kHugeGridY = 64;
constexpr int64_t kExpandMaxBlocks = 4096;     // k_expand_bits: 2^20 rows per sweep
__global__ __launch_bounds__(256) void k_expand_huge(const ExpandArgs a) {
    int cnt = *a.huge_count;
    if (cnt > a.huge_cap) cnt = a.huge_cap;
    const int bps = (int)a.bp.bps;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int w = blockIdx.y; w < cnt; w += gridDim.y) {
        const HugeRow r = a.huge[w];
        for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < r.kb; k += stride) {
            const int sh = (bps == 1) ? 0 : bps - 1 - (int)(k % bps);
            const uint8_t b = (r.type < 0) ? 0 : (uint8_t)((r.type >> sh) & 1);
            if (r.ob + k < a.cap_bits) a.bits[r.ob + k] = b;
            if (a.bp.write_pos && r.op + k < a.cap_pos) a.pos[r.op + k] = r.ts + k * a.bp.samples_per_bit;
        }
    }
}


// =====================================================================================================
// Single-GPU tail for everything but ASK ("tile" tail): five launches from chunk records to bits.
//
//   k_resolve_one      per chunk: does its trailing short run turn stable (chunk_stable)?  Then ONE scan instead of the three
//                      dependent ones above: what a chunk does to the pulse table is a function of the stable state it is entered
//                      with, and these functions compose associatively (ResElem), so a single prefix composition yields, per
//                      chunk, the stable state before it, the number of accepted runs before it and the last accepted run
//                      before it.  Workgroup-local exclusive prefixes + one total per workgroup (no workgroup waits for another).
//   k_emit_rows_tiles  one wavefront per chunk: folds the workgroup totals before its own (a few dozen 32-byte values), writes
//                      the chunk's rows and -- the rows being in registers anyway -- what they contribute to _ppseq_to_bits
//                      (bits, long pauses, samples, data rows: a VecK<4> per chunk, the chunk being a TILE of the row table);
//                      one more wavefront writes the totals, the table's last row and the last (one-row) tile.
//   tile scan          k_scan_lookback<4> over the 16 K tile aggregates (a handful of workgroups): exclusive prefix per tile; the
//                      few tiles that hold a long pause (or the table's last row) are walked for their GroupInfo.
//   group scan         k_scan_lookback<3> as in the generic path.
//   k_expand_tiles     one wavefront per tile: re-reads its rows (16 B each), a wave scan places every row's bits and positions;
//                      rows longer than kHugeBits bits were listed by k_emit_rows_tiles and are expanded by extra workgroups of
//                      the same launch.
// Against the generic path (resolve a / b, rows, row-scan reduce + apply, group scan, expand, expand-huge: 8 launches, two of them
// passes over a 24-byte-per-row side table with 1.3 wavefronts per SIMD) this drops the RowInfo table, three launches and every
// under-occupied pass.  ASK (rows merge across chunks after the short-pause rule) and sharded captures keep the generic path.
// =====================================================================================================
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

struct TileTail {            // scratch of the tile tail (carved from the context's arena)
    ResElem *ploc;           // [n_chunks] exclusive prefix inside the chunk's resolve workgroup
    ResElem *btot;           // [resolve workgroups] totals
    VecK<4> *agg;            // [n_chunks + 1] per tile: (bits, long pauses, samples, data rows) of its rows
    VecK<4> *excl;           // [n_chunks + 1] exclusive prefix of agg (tile scan)
    int64_t *tile_off;       // [n_chunks + 1] first row of the tile
    int32_t *tile_cnt;       // [n_chunks + 1] rows in the tile
    int64_t *d_n_tiles;      // = n_chunks + 1 (device copy for the scan kernel)
    int32_t *huge_count;     // [2] persistent, alternating by pass parity; zero between uses
    int parity;
    int want_bits;
    GranDesc *rdesc;         // look-back descriptors of k_resolve_one (persistent, tagged with the pass number)
    unsigned long long epoch;
    int64_t *d_row_base;     // sharded captures: receives the global index of this GPU's first row (nullptr on a single GPU)
    int64_t *esc;            // staged passes with 16-bit row lengths: the escape list, esc[0] = count (reset by k_resolve_one); else nullptr
};

// The kernels below are latency chains of a few memory round trips on a nearly idle chip, not bandwidth: every load whose
// address does not depend on loaded data is issued BEFORE the first use of any of them (measured: the same kernels written in
// natural order spent 9 us per wavefront in 5 serialised round trips).
// blk0: first resolve workgroup of this launch -- 0 for a whole capture; a SEGMENT of a streamed pass (see "segments" below) covers the
// workgroups [blk0, blk0 + gridDim.x), its look-ahead into the chunk behind the segment's last one reads a record the hot kernel has
// already finished (the segment's gate waits for that chunk too)
__global__ __launch_bounds__(kResolveBlock) void k_resolve_one(const ResolveArgs a, const TileTail ft, const int64_t blk0, const SegGate gate) {
    URH_TAIL_PRIO();
    __shared__ ResElem s_w[kResolveBlock / 64];
    const int64_t blk = blk0 + blockIdx.x;
    const int64_t c = blk * kResolveBlock + threadIdx.x;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (gate.fused) {
        // the gate of a short segment (k_seg_gate's loop, see "Segments" below) inside its only resolve workgroup: ONE thread polls the
        // segment's counter; the acquire fence behind it (every thread: L1 and the non-local lines of this XCD's L2 are dropped) orders the
        // loads below after the hot kernel's acknowledged write-through stores
        if (t == 0) {
            const uint32_t *ctr = gate.progress + gate.k * kProgressStride;
            const long long t0 = (long long)wall_clock64();
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gate.target) {
                __builtin_amdgcn_s_sleep(8);
                if ((long long)wall_clock64() - t0 > gate.max_ticks) { gate.seg->err = 1 + gate.k; break; }
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    if (gate.seg && gate.init && blockIdx.x == 0 && t == 0) {
        if (!gate.progress) gate.seg->err = 0;               // (no gate kernel in front of this segment: its part of the initialisation)
        // the first rows segment of a streamed pass clears what the bits segments carry along (they start behind this kernel)
        int64_t *w = (int64_t *)&gate.seg->in[0];
        for (int i = 0; i < (int)(2 * sizeof(SegState::In) / 8); ++i) w[i] = 0;
    }
    // this pass's (segment's) huge-row counter starts at zero whatever an earlier pass of the same parity left behind (one that failed
    // between its row and its expansion launches never reached the kernel that clears the counter for its successor)
    if (blockIdx.x == 0 && t == 0 && ft.want_bits) ft.huge_count[ft.parity] = 0;
    if (blockIdx.x == 0 && t == 0 && ft.esc) ft.esc[0] = 0;
    ResElem e = res_identity();
    if (c < a.n_chunks) {
        ChunkInfo *ch = a.chunks + c;
        // everything of this chunk, and -- speculatively -- the next chunk's leading stretch, in one round trip
        const int64_t pend_pos = ch->pend_pos, start = ch->start, len = ch->len, last_pos = ch->last_pos;
        const int cnt = ch->cnt;
        const uint32_t first_state = ch->first_state, last_state = ch->last_state, pend_state = ch->pend_state;
        const bool has_next = c + 1 < a.n_chunks;
        int64_t lead = has_next ? ch[1].lead : 0, nlen = has_next ? ch[1].len : 1;
        // does the trailing short run grow past `tol` in the following chunks (chunk_stable, single-GPU form)
        int ps = 0;
        if (pend_pos >= 0) {
            int64_t run = start + len - pend_pos;
            for (int64_t u = c + 1; run <= a.tol && u < a.n_chunks; ++u) {
                if (u > c + 1) { lead = a.chunks[u].lead; nlen = a.chunks[u].len; }
                run += lead;
                if (lead < nlen) break;
            }
            ps = run > a.tol;
        }
        ch->pend_stable = ps;
        e = res_of_chunk(cnt, first_state, last_state, pend_state, last_pos, pend_pos, ps);
    }
    const ResElem incl = res_wave_incl_scan(e, lane);
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    ResElem base = res_identity(), tot = res_identity();
#pragma unroll
    for (int w = 0; w < kResolveBlock / 64; ++w) {
        if (w < wave) base = res_combine(base, s_w[w]);
        tot = res_combine(tot, s_w[w]);
    }
    ResElem ex = res_shfl_up(incl, 1);
    if (lane == 0) ex = res_identity();
    ex = res_combine(base, ex);
    if (c < a.n_chunks) ft.ploc[c] = ex;                  // exclusive prefix inside this workgroup
    if (t == 0) ft.btot[blk] = tot;                       // consumers compose the totals of the workgroups before theirs
}

// composition of btot[0 .. count) by one wavefront, in order (every lane gets it); `first` = btot[lane] already loaded
__device__ __forceinline__ ResElem res_fold_blocks(const ResElem *btot, int64_t count, int lane, ResElem first) {
    ResElem carry = res_identity();
    for (int64_t u0 = 0; u0 < count; u0 += 64) {
        ResElem x = first;
        if (u0 > 0) { x = res_identity(); if (u0 + lane < count) x = btot[u0 + lane]; }
        x = res_wave_incl_scan(x, lane);
        carry = res_combine(carry, res_shfl(x, 63));
    }
    return carry;
}

// state machine before table entry c (c == n_chunks: behind the last one): the resolve workgroups before c's, then c's workgroup-local prefix
__device__ __forceinline__ ResElem res_before_entry(const TileTail &ft, const ResElem &init, int64_t c, int64_t n_chunks, int lane) {
    const int64_t nb = (c >= n_chunks) ? resolve_blocks(n_chunks) : c / kResolveBlock;
    ResElem first = res_identity();
    if (lane < nb) first = ft.btot[lane];
    ResElem pre = res_combine(init, res_fold_blocks(ft.btot, nb, lane, first));
    if (c < n_chunks) pre = res_combine(pre, ft.ploc[c]);
    return pre;
}

// ---- local pass of a sharded capture as ONE launch --------------------------------------------------------------------------
// The shard on its own -> the ChunkInfo that stands for it in the other ranks' tables (resolve_finish's local_pass branch spells the
// fields out).  The generic local pass was the three resolve launches above (~ 40 us in front of the summary exchange, i.e. in the
// tail chain that sets a sharded step's period); but every field of the summary is a function of the composition of the shard's
// chunks as ResElems entered with an unknown state -- P = cnt + 1, first / last stable state, the last accepted run -- plus two
// indices (first chunk with a run boundary, the chunk whose trailing short run reaches the shard end undecided): kResolveBlock
// chunks per workgroup, the workgroup that delivers the last partial composes them in order.  Leaves chunks, scratch and aux alone.
struct ShardPart { ResElem tot; int32_t first_nonlead, open_chunk; };
__global__ __launch_bounds__(kResolveBlock) void k_shard_summary(const ResolveArgs a, ShardPart *part, int32_t *ticket) {
    URH_TAIL_PRIO();
    __shared__ ResElem s_w[kResolveBlock / 64];
    __shared__ int32_t s_fn[kResolveBlock / 64], s_oc[kResolveBlock / 64];
    __shared__ bool s_last;
    const int64_t c = (int64_t)blockIdx.x * kResolveBlock + threadIdx.x;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    ResElem e = res_identity();
    int32_t fn = kAuxNone, oc = kAuxNone;
    if (c < a.n_chunks) {
        const ChunkInfo *ch = a.chunks + c;
        const int64_t pend_pos = ch->pend_pos, start = ch->start, len = ch->len, last_pos = ch->last_pos, own_lead = ch->lead;
        const int cnt = ch->cnt;
        const uint32_t first_state = ch->first_state, last_state = ch->last_state, pend_state = ch->pend_state;
        const bool has_next = c + 1 < a.n_chunks;
        int64_t lead = has_next ? ch[1].lead : 0, nlen = has_next ? ch[1].len : 1;
        int ps = 0;
        if (pend_pos >= 0) {                                 // chunk_stable, local form: a run that reaches the shard end undecided stays open
            int64_t run = start + len - pend_pos;
            bool hit = false;
            for (int64_t u = c + 1; run <= a.tol && u < a.n_chunks; ++u) {
                if (u > c + 1) { lead = a.chunks[u].lead; nlen = a.chunks[u].len; }
                run += lead;
                if (lead < nlen) { hit = true; break; }
            }
            ps = run > a.tol;
            if (!ps && !hit) oc = (int32_t)c;
        }
        e = res_of_chunk(cnt, first_state, last_state, pend_state, last_pos, pend_pos, ps);
        if (own_lead < len) fn = (int32_t)c;
    }
    const ResElem incl = res_wave_incl_scan(e, lane);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { fn = min(fn, __shfl_down(fn, o)); oc = min(oc, __shfl_down(oc, o)); }
    if (lane == 63) s_w[wave] = incl;
    if (lane == 0) { s_fn[wave] = fn; s_oc[wave] = oc; }
    __syncthreads();
    if (t == 0) {
        ResElem tot = res_identity();
        int32_t f = kAuxNone, o = kAuxNone;
#pragma unroll
        for (int w = 0; w < kResolveBlock / 64; ++w) { tot = res_combine(tot, s_w[w]); f = min(f, s_fn[w]); o = min(o, s_oc[w]); }
        ShardPart *p = part + blockIdx.x;
        __hip_atomic_store(&p->tot.cnt, tot.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&p->tot.first_pos, tot.first_pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&p->tot.la_pos, tot.la_pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&p->tot.meta, tot.meta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&p->first_nonlead, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&p->open_chunk, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        s_last = atomicAdd(ticket, 1) == (int32_t)gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last || wave != 0) return;
    __threadfence();
    const int64_t nb = gridDim.x;
    ResElem tot = res_identity();
    fn = kAuxNone; oc = kAuxNone;
    for (int64_t u0 = 0; u0 < nb; u0 += 64) {
        ResElem x = res_identity();
        if (u0 + lane < nb) {
            const ShardPart *p = part + u0 + lane;
            x.cnt = __hip_atomic_load(&p->tot.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            x.first_pos = __hip_atomic_load(&p->tot.first_pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            x.la_pos = __hip_atomic_load(&p->tot.la_pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            x.meta = __hip_atomic_load(&p->tot.meta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            fn = min(fn, __hip_atomic_load(&p->first_nonlead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            oc = min(oc, __hip_atomic_load(&p->open_chunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        x = res_wave_incl_scan(x, lane);
        tot = res_combine(tot, res_shfl(x, 63));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { fn = min(fn, __shfl_down(fn, o)); oc = min(oc, __shfl_down(oc, o)); }
    if (lane != 0) return;
    const ChunkInfo *ch = a.chunks;
    const bool has = tot.has();
    const int64_t P = has ? tot.cnt + 1 : 0;              // entered with an unknown state, the first stable run is accepted
    ChunkInfo s;
    s.start = ch[0].start; s.len = a.n_total;
    s.lead = (fn >= a.n_chunks) ? a.n_total : ch[fn].start - ch[0].start + ch[fn].lead;
    s.pend_pos = -1; s.pend_state = 0;
    if (oc < a.n_chunks) { s.pend_pos = ch[oc].pend_pos; s.pend_state = ch[oc].pend_state; }
    s.cnt = (int32_t)P;
    s.first_state = has ? (uint16_t)tot.first_state() : (uint16_t)0xFFFFu;
    s.last_state = has ? (uint16_t)tot.last_state() : (uint16_t)0xFFFFu;
    s.last_pos = !has ? 0 : (tot.la_valid() ? tot.la_pos : tot.first_pos);
    s.init_state = ch[0].init_state;
    s.first_acc = 0; s.pend_acc = 0; s.pend_stable = 0; s.pad = 0;
    *a.summary_out = s;
    *a.d_n_acc = P;
    *ticket = 0;
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
// Wavefronts per workgroup of k_emit_rows_tiles / k_expand_tiles.  Four, not eight: beside the next pass's hot kernel (pipelined passes)
// a workgroup needs one free wave slot per SIMD, which retiring hot workgroups (four wavefronts) leave; eight-wavefront workgroups were
// starved until the hot kernel had drained (the row kernel then took that kernel's whole 290 us).
constexpr int kEmitWaves = 4;
// Consecutive chunks (tiles) per wavefront.  Beside a hot kernel that fills every SIMD's register file, a workgroup of the tail
// starts only when a hot workgroup retires (rocprofv3 timeline of round 3: one chunk per wavefront = 4096 workgroups took 110 us
// beside the hot kernel against 18 us on an idle machine -- the kernel was bound by the number of workgroups that had to find a
// slot, not by its work).  kTailCPW chunks per wavefront: the per-chunk metadata of all of them arrives in ONE round trip (lane j holds
// chunk j's), the records of the next chunk are requested before the current one is processed.  Measured (round 3, profiles/HISTORY.md):
// 1 / 4 / 8 / 16 chunks per wavefront give 0.319 / 0.311 / 0.311 / 0.311 ms per pipelined step with D2H, but 0.402 / 0.402 / 0.406 /
// 0.444 ms for ONE capture on an idle machine (fewer wavefronts = less parallelism there): four.
#ifndef URH_TAIL_CPW
#define URH_TAIL_CPW 4
#endif
constexpr int kTailCPW = URH_TAIL_CPW;
#ifndef URH_EXPAND_PREFETCH
#define URH_EXPAND_PREFETCH 1
#endif
#ifndef URH_PACK_WORDS
#define URH_PACK_WORDS 1
#endif
#ifndef URH_EXPAND_EARLY_EXIT
#define URH_EXPAND_EARLY_EXIT 1
#endif
// -DURH_TAIL_CAP80: at most 80 VGPRs (six wavefronts per SIMD) for the tail kernels that need more -- what ONE retiring hot wavefront
// (72 + the 8 it never had) leaves free on a SIMD; costs spills (A/B knob)
#ifdef URH_TAIL_CAP80
#define URH_TAIL_OCC __attribute__((amdgpu_waves_per_eu(URH_TAIL_CAP80)))
#else
#define URH_TAIL_OCC
#endif
// the same per kernel (A/B builds): -DURH_OCC_EMIT=8 / -DURH_OCC_EXPAND=8 hold the row / expansion kernel to that many wavefronts per SIMD
#ifdef URH_OCC_EMIT
#define URH_EMIT_OCC __attribute__((amdgpu_waves_per_eu(URH_OCC_EMIT)))
#else
#define URH_EMIT_OCC
#endif
#ifdef URH_OCC_EXPAND
#define URH_EXPAND_OCC __attribute__((amdgpu_waves_per_eu(URH_OCC_EXPAND)))
#else
#define URH_EXPAND_OCC URH_TAIL_OCC
#endif
static_assert(kResolveBlock % kTailCPW == 0 && kTailCPW <= 64, "the chunks of one wavefront share their resolve workgroup");

__device__ __forceinline__ int64_t lane_bcast(int64_t v, int src) {          // src wave-uniform
    const int lo = __builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)v, src), hi = __builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), src);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}
__device__ __forceinline__ int lane_bcast(int v, int src) { return __builtin_amdgcn_readlane(v, src); }

struct EmitTileArgs {
    EmitArgs e;
    ResolveArgs r;
    TileTail ft;
    BitsParams bp;
    HugeRef *huge;
    int32_t huge_cap;
    // Wavefronts [w0, w_end) own chunks, wavefront w_end writes the totals.  A whole capture: w0 = 0, w_end = tail_waves(n_chunks),
    // final_seg = 1.  A segment of a streamed pass covers the chunks [w0, w_end) * kTailCPW; before the last one (final_seg = 0) the
    // totals wavefront only notes how many rows are final now (seg->n_rows: what the segment's later kernels take as the row count).
    int64_t w0, w_end;
    int final_seg;
    SegState *seg;
    int seg_k;               // rows segment index: seg->rows_at[seg_k] receives the rows that are final after it
    // a streamed pass ships its rows while it writes them: the compact blob's row_state / row_len sections in pinned HOST memory
    // (plain stores over PCIe, fire and forget; nullptr: not shipped)
    int8_t *h_state;
    int32_t *h_len;
    int len16;               // 1: h_len holds uint16 lengths, 2: uint16 state | length words (no h_state stores); see RowsSegment::len16
    int64_t *esc;
    int64_t esc_cap;
};

__host__ __device__ static inline int64_t tail_waves(int64_t n_items) { return (n_items + kTailCPW - 1) / kTailCPW; }

// a row's state and length into the shipped sections (EmitTileArgs::h_state / h_len): int32 lengths, or uint16 with the escape list
__device__ __forceinline__ void ship_row(const EmitTileArgs &g, int64_t idx, int64_t state, int64_t len) {
    if (g.len16 == 2) {
        // URHGPU_BLOB_ROW16: state and length in ONE uint16 -- (state + 1) << 13 | length, 0x1FFF = look the row up in the escape list
        const bool fits = len >= 0 && len < 0x1FFF;
        ((uint16_t *)g.h_len)[idx] = (uint16_t)((((uint32_t)(state + 1) & 7u) << 13) | (fits ? (uint32_t)len : 0x1FFFu));
        if (!fits) {
            const unsigned long long slot = atomicAdd((unsigned long long *)g.esc, 1ull);
            if ((int64_t)slot < g.esc_cap) g.esc[1 + slot] = (int64_t)(((uint64_t)(uint32_t)(int32_t)len << 32) | (uint64_t)(uint32_t)idx);
        }
        return;
    }
    g.h_state[idx] = (int8_t)state;
    if (!g.len16) { g.h_len[idx] = (int32_t)len; return; }
    const bool fits = len >= 0 && len < 0xFFFF;
    ((uint16_t *)g.h_len)[idx] = fits ? (uint16_t)len : (uint16_t)0xFFFF;
    if (!fits) {
        const unsigned long long slot = atomicAdd((unsigned long long *)g.esc, 1ull);
        if ((int64_t)slot < g.esc_cap) g.esc[1 + slot] = (int64_t)(((uint64_t)(uint32_t)(int32_t)len << 32) | (uint64_t)(uint32_t)idx);
    }
}

// Sharded captures run the same kernel over the TABLE of the generic resolve pass -- the other shards' summaries around this GPU's
// chunks (ResolveArgs::chunk_first / n_local).  A summary entry is a chunk whose records live elsewhere: it writes no rows, but what its
// rows span (positions telescope) is its tile's sample count, so that the exclusive tile scan hands this GPU's first tile the
// total_samples before its first row; bits, long pauses and data rows stay per GPU (their cross-shard part is the flags exchange).
// Rows are indexed from this GPU's first one (row_base, also left in *ft.d_row_base for the kernels that follow).
__global__ __launch_bounds__(64 * kEmitWaves) URH_EMIT_OCC void k_emit_rows_tiles(const EmitTileArgs g) {
    URH_TAIL_PRIO();
    const EmitArgs &a = g.e;
    const ResolveArgs &r = g.r;
    const int lane = threadIdx.x & 63;
    const int64_t w = g.w0 + (int64_t)blockIdx.x * kEmitWaves + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t n_cw = g.w_end;                         // wavefronts below it own chunks; wavefront n_cw: totals, last row, last tile
    if (w > n_cw) return;
    const bool totals = (w == n_cw);
    const bool table = (r.n_local != r.n_chunks);         // sharded: entries outside [lc0, lc1) are the other shards' summaries
    const int64_t lc0 = a.chunk_first, lc1 = a.chunk_first + r.n_local;
    const int64_t c0 = totals ? (g.final_seg ? r.n_chunks : n_cw * kTailCPW) : w * kTailCPW;
    // ---- one round trip: everything this wavefront reads that does not depend on loaded data ----
    // resolve workgroups composed on the left: the same for every chunk of this wavefront (kResolveBlock % kTailCPW == 0); the totals
    // wavefront takes them all
    const int64_t n_before = (totals && g.final_seg) ? resolve_blocks(r.n_chunks) : c0 / kResolveBlock;
    const uint32_t init_state = a.chunks[0].init_state;
    ResElem first = res_identity();
    if (lane < n_before) first = g.ft.btot[lane];
    const int n_mine = totals ? 0 : (int)((r.n_chunks - c0 < kTailCPW) ? r.n_chunks - c0 : kTailCPW);
    ResElem pl_l = res_identity();
    int cnt_l = 0, fl_l = 0, pp_l = 0;                    // fl: first_state | last_state << 16 (0xFFFF = none); pp: pend_state | pend_stable << 16
    int64_t pend_pos_l = -1, last_pos_l = 0;
    if (lane < n_mine) {                                  // lane j: chunk c0 + j
        pl_l = g.ft.ploc[c0 + lane];
        const ChunkInfo *ch = a.chunks + c0 + lane;
        cnt_l = ch->cnt; pend_pos_l = ch->pend_pos;
        if (table) last_pos_l = ch->last_pos;
        fl_l = (int)((uint32_t)ch->first_state | ((uint32_t)ch->last_state << 16));
        pp_l = (int)((uint32_t)ch->pend_state | ((uint32_t)(ch->pend_stable ? 1 : 0) << 16));
    }
    uint64_t rec = 0;
    if (n_mine > 0 && c0 >= lc0 && c0 < lc1 && lane < a.slab_stride)
        rec = (a.slab + (c0 - lc0) * a.slab_stride)[lane];     // speculative: the first 64 records (those beyond cnt are ignored)
    const ResElem init = res_make(init_state, -1, init_state, 0, false, -1, 0);
    const ResElem base = res_combine(init, res_fold_blocks(g.ft.btot, n_before, lane, first));   // state machine before this wavefront's resolve workgroup
    int64_t row_base = 0;                                 // global index of this GPU's first row
    if (table) row_base = res_before_entry(g.ft, init, lc0, r.n_chunks, lane).cnt;
    if (totals) {
        // totals, the table's last row (signal_functions.pyx:485-493; skipped when the table already has n rows, :487), last tile
        const ResElem pre = base;
        if (!g.final_seg) {
            // a segment of a streamed pass that is not the last: every accepted run of the chunks so far has its row (a row ends where
            // the next accepted run begins), so rows [0, pre.cnt) are final
            if (lane == 0) g.seg->rows_at[g.seg_k] = (r.rows != nullptr && pre.cnt > r.cap_rows) ? r.cap_rows : pre.cnt;
            return;
        }
        const int64_t c = r.n_chunks;
        int64_t row_end = pre.cnt;                         // this GPU's rows are global rows [row_base, row_end) (+ the last row on the last GPU)
        if (table && lc1 < r.n_chunks) row_end = res_before_entry(g.ft, init, lc1, r.n_chunks, lane).cnt;
        if (lane == 0) {
            const int64_t P = pre.cnt;
            *r.d_n_acc = P;
            int64_t n_rows = row_end - row_base;
            int64_t o = n_rows;
            VecK<4> v; v.zero();
            int32_t tcnt = 0;
            if (P < r.n_total && r.write_last_row) {
                o = P - row_base;
                n_rows = o + 1;
                tcnt = 1;
                const int64_t fpos = pre.la_valid() ? pre.la_pos : -1;
                const uint32_t fstate = pre.la_valid() ? pre.la_state() : init_state;
                const int64_t len = (P == 0) ? (r.n_total - r.tol) : (r.n_total - 1 - fpos - r.tol);
                if (r.rows != nullptr && o < r.cap_rows) {
                    r.rows[2 * o] = (int64_t)fstate - 1; r.rows[2 * o + 1] = len;
                    if (g.h_state) ship_row(g, o, (int64_t)fstate - 1, len);
                }
                if (g.ft.want_bits) {
                    v = row_value((int64_t)fstate - 1, len, P == 0, g.bp);
                    if (v.v[0] > kHugeBits) {
                        const int slot = atomicAdd(g.ft.huge_count + g.ft.parity, 1);
                        if (slot < g.huge_cap) { g.huge[slot].tile = c; g.huge[slot].row = o; }
                    }
                }
            }
            *r.d_n_rows_needed = n_rows;
            *r.d_n_rows = (r.rows != nullptr && n_rows > r.cap_rows) ? r.cap_rows : n_rows;
            if (g.ft.d_row_base) *g.ft.d_row_base = row_base;
            g.ft.agg[c] = v; g.ft.tile_off[c] = o; g.ft.tile_cnt[c] = tcnt;
        }
        return;
    }
    for (int jc = 0; jc < n_mine; ++jc) {
        const int64_t c = c0 + jc;
        const bool local = (c >= lc0 && c < lc1);
        const uint64_t *slab = a.slab + (c - lc0) * a.slab_stride;          // (local entries only)
        // the next chunk's records are on their way while this one is worked on
        uint64_t rec_next = 0;
        if (jc + 1 < n_mine && c + 1 >= lc0 && c + 1 < lc1 && lane < a.slab_stride) rec_next = (a.slab + (c + 1 - lc0) * a.slab_stride)[lane];
        ResElem pl;
        pl.cnt = lane_bcast(pl_l.cnt, jc); pl.first_pos = lane_bcast(pl_l.first_pos, jc); pl.la_pos = lane_bcast(pl_l.la_pos, jc);
        pl.meta = (uint64_t)lane_bcast((int64_t)pl_l.meta, jc);
        const int cnt = lane_bcast(cnt_l, jc);
        const uint32_t fl = (uint32_t)lane_bcast(fl_l, jc), pp = (uint32_t)lane_bcast(pp_l, jc);
        const int64_t pend_pos = lane_bcast(pend_pos_l, jc);
        const uint32_t c_first = fl & 0xFFFFu, c_last = fl >> 16, c_pend = pp & 0xFFFFu;
        const int pend_stable = (int)(pp >> 16);
        const ResElem pre = res_combine(base, pl);              // state machine before this chunk
        const uint32_t prev_state = pre.last_state();
        const int first_acc = (cnt > 0) && (c_first != prev_state);
        const uint32_t before_pend = (cnt > 0) ? c_last : prev_state;
        const int pend_acc = pend_stable && (c_pend != before_pend);
        const int skip = (cnt > 0 && !first_acc) ? 1 : 0;
        const int64_t from_slab = (cnt > 0) ? cnt - skip : 0;
        const int64_t total = from_slab + pend_acc;
        const int64_t out_off = pre.cnt;
        const int64_t prev_pos = pre.la_valid() ? pre.la_pos : -1;
        const uint32_t prev_st = pre.la_valid() ? pre.la_state() : init_state;
        if (!local) {
            // another shard's summary: no rows of this GPU, only the samples its rows span
            VecK<4> acc; acc.zero();
            if (g.ft.want_bits && total > 0) acc.v[2] = (pend_acc ? pend_pos : lane_bcast(last_pos_l, jc)) - prev_pos;
            if (lane == 0) { g.ft.agg[c] = acc; g.ft.tile_off[c] = 0; g.ft.tile_cnt[c] = 0; }
            rec = rec_next;
            continue;
        }
        // what the chunk's rows contribute to _ppseq_to_bits: bits (64-bit sum), long pauses and data rows (two 32-bit counts in one
        // word); the samples need no sum at all -- row lengths telescope: sum = (position of the chunk's last accepted run) - prev_pos
        int64_t acc_bits = 0;
        uint64_t acc_ld = 0;
        int64_t my_last_pos = 0;
        const bool in_regs = cnt <= 64;                      // every record this chunk has is in `rec`
        // record of row j's run (j + skip) and of the run before it, from the neighbours' registers
        const uint64_t rec_j = skip ? (uint64_t)__shfl_down((long long)rec, 1) : rec;
        const uint64_t rec_p = (uint64_t)__shfl_up((long long)rec_j, 1);
        for (int64_t j = lane; j < total; j += 64) {
            int64_t pos, ppos; uint32_t pst;
            if (in_regs && j < 64) {
                pos = (j < from_slab) ? rec_pos(rec_j) : pend_pos;
                if (j == 0) { ppos = prev_pos; pst = prev_st; }
                else { ppos = rec_pos(rec_p); pst = rec_state(rec_p); }
            } else {
                pos = (j < from_slab) ? rec_pos(slab[j + skip]) : pend_pos;
                if (j == 0) { ppos = prev_pos; pst = prev_st; }
                else { const uint64_t rr = slab[j - 1 + skip]; ppos = rec_pos(rr); pst = rec_state(rr); }
            }
            my_last_pos = pos;
            const int64_t gi = out_off + j;                  // global row; this GPU's row gi - row_base
            const int64_t len = (gi == 0) ? pos + 1 : pos - ppos;
            const int64_t state = (int64_t)pst - 1;
            if (gi - row_base < a.cap_rows) {
                if (a.rows) *(longlong2 *)(a.rows + 2 * (gi - row_base)) = longlong2{(long long)state, (long long)len};
                if (g.h_state) ship_row(g, gi - row_base, state, len);
            }
            if (g.ft.want_bits) {
                const VecK<4> v = row_value(state, len, gi == 0, g.bp);
                acc_bits += v.v[0];
                acc_ld += (uint64_t)v.v[1] | ((uint64_t)v.v[3] << 32);
                if (v.v[0] > kHugeBits) {
                    const int slot = atomicAdd(g.ft.huge_count + g.ft.parity, 1);
                    if (slot < g.huge_cap) { g.huge[slot].tile = c; g.huge[slot].row = gi - row_base; }
                }
            }
        }
        VecK<4> acc; acc.zero();
        if (g.ft.want_bits && total > 0) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { acc_bits += __shfl_xor(acc_bits, o); acc_ld += (uint64_t)__shfl_xor((long long)acc_ld, o); }
            const int64_t last_pos = __shfl(my_last_pos, (int)((total - 1) & 63));       // the lane that wrote the chunk's last row
            acc.v[0] = acc_bits; acc.v[1] = (int64_t)(acc_ld & 0xFFFFFFFFull); acc.v[3] = (int64_t)(acc_ld >> 32);
            acc.v[2] = last_pos - prev_pos;                  // (prev_pos = -1 before the table's first row: its length is position + 1)
        }
        if (lane == 0) { g.ft.agg[c] = acc; g.ft.tile_off[c] = out_off - row_base; g.ft.tile_cnt[c] = (int32_t)total; }
        rec = rec_next;
    }
}

// ---- tile scan: exclusive prefix per tile (one thread per tile, look-back across the few dozen workgroups); the tiles that hold a
// long pause or the table's last row are then walked, one wavefront each, for their GroupInfo -----------------------------------
struct TileScanArgs {
    const int64_t *rows;
    const int64_t *d_n_rows;
    const VecK<4> *agg;
    const int64_t *tile_off;
    const int32_t *tile_cnt;
    VecK<4> *excl;
    GroupInfo *groups;
    int64_t cap_groups;
    int64_t *d_n_groups;
    int64_t n_tiles;
    BitsParams bp;
    GranDesc *desc;
    unsigned long long epoch;
    // a segment of a streamed pass: workgroups [b0, b0 + gridDim.x) -- the look-back reaches into the descriptors the earlier segments'
    // launches published under the same epoch --, n_tiles = the segment's end, *d_n_rows = the rows that are final so far (the
    // "trailing group" the walk records is then the group that is still open: what the segment's group scan starts the next one from)
    int64_t b0;
    SegState *seg;
    int seg_parity;          // segment index & 1
};
__global__ __launch_bounds__(kScanBlock) URH_TAIL_OCC void k_tile_scan(const TileScanArgs a) {
    URH_TAIL_PRIO();
    __shared__ VecK<4> s_wave[kScanBlock / 64];
    __shared__ VecK<4> s_prefix;
    __shared__ VecK<4> s_ex[kScanBlock];                 // listed tiles: prefix, first row, end
    __shared__ int64_t s_off[kScanBlock], s_end[kScanBlock];
    __shared__ int s_count;
    const int64_t b = a.b0 + blockIdx.x, nb = a.b0 + gridDim.x;
    const int64_t t = b * kScanBlock + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // one round trip
    const int64_t n = *a.d_n_rows;
    const bool row0g = !a.bp.d_row_base || *a.bp.d_row_base == 0;     // this GPU's row 0 is the capture's row 0
    VecK<4> mine; mine.zero();
    int64_t off = 0; int32_t cnt = 0;
    if (t < a.n_tiles) { mine = a.agg[t]; off = a.tile_off[t]; cnt = a.tile_cnt[t]; }
    if (threadIdx.x == 0) s_count = 0;
    VecK<4> total;
    VecK<4> ex = block_excl_scan_vec<4>(mine, total, s_wave);
    if (threadIdx.x < 64) {                               // wavefront 0 publishes and looks back
        const uint32_t tag = (uint32_t)a.epoch;
        if (threadIdx.x == 0) gran_store(a.desc[b].agg, total, tag);
        VecK<4> zero; zero.zero();
        bool ok;
        const VecK<4> prefix = gran_look_back<VecK<4>>(a.desc, b, tag, zero,
                                                         [](const VecK<4> &l, const VecK<4> &r) { VecK<4> x = l; x.add(r); return x; },
                                                         [](const VecK<4> &x, int o) { VecK<4> r; for (int k = 0; k < 4; ++k) r.v[k] = __shfl_down(x.v[k], o); return r; },
                                                         [](const VecK<4> &x) { VecK<4> r; for (int k = 0; k < 4; ++k) r.v[k] = __shfl(x.v[k], 0); return r; },
                                                         lane, ok);
        if (threadIdx.x == 0) {
            VecK<4> incl = prefix; incl.add(total);
            gran_store(a.desc[b].incl, incl, tag);
            s_prefix = prefix;
            if (b == nb - 1) {                            // number of groups = long pauses + 1
                int64_t g = (n > 0) ? incl.v[1] + 1 : 0;
                if (g > a.cap_groups) g = a.cap_groups;
                *a.d_n_groups = g;
                if (a.seg) {                              // the segment's group scan covers the groups from the one that was open before it
                    const int64_t g0 = a.seg->in[a.seg_parity].g0;
                    a.seg->n_groups_local = (g > g0) ? g - g0 : 0;
                }
            }
        }
    }
    __syncthreads();
    ex.add(s_prefix);
    int64_t end = off + cnt;
    if (end > n) end = n;
    if (t < a.n_tiles) {
        a.excl[t] = ex;
        if (end > off && (mine.v[1] > 0 || end == n)) {
            const int slot = atomicAdd(&s_count, 1);
            s_ex[slot] = ex; s_off[slot] = off; s_end[slot] = end;
        }
    }
    __syncthreads();
    const int n_list = s_count;
    for (int w = wave; w < n_list; w += kScanBlock / 64) {
        // walk the listed tile's rows 64 at a time
        const int64_t o = s_off[w], e = s_end[w];
        VecK<4> run = s_ex[w];
        for (int64_t i0 = o; i0 < e; i0 += 64) {
            const int64_t i = i0 + lane;
            VecK<4> v; v.zero();
            int64_t type = 0, len = 0;
            if (i < e) {
                const longlong2 row = *(const longlong2 *)(a.rows + 2 * i);
                type = row.x; len = row.y;
                v = row_value(type, len, i == 0 && row0g, a.bp);
            }
            const VecK<4> incl = wave_incl_scan_vec<4>(v, lane);
            VecK<4> before = run;
#pragma unroll
            for (int k = 0; k < 4; ++k) before.v[k] += incl.v[k] - v.v[k];
            if (i < e && v.v[1] && before.v[1] < a.cap_groups) {          // long pause: closes group before.v[1]
                GroupInfo gi;
                gi.bits_end = before.v[0]; gi.data_end = before.v[3]; gi.ts_close = before.v[2]; gi.pause = len; gi.closed = 1; gi.pad = 0;
                a.groups[before.v[1]] = gi;
            }
            if (i < e && i + 1 == n && before.v[1] + v.v[1] < a.cap_groups) {   // trailing group
                GroupInfo gi;
                gi.bits_end = before.v[0] + v.v[0]; gi.data_end = before.v[3] + v.v[3]; gi.ts_close = before.v[2] + v.v[2];
                gi.pause = (type == -1) ? len : 0;          // :411
                gi.closed = 0; gi.pad = 0;
                a.groups[before.v[1] + v.v[1]] = gi;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) run.v[k] += __shfl(incl.v[k], 63);
        }
    }
}

struct ExpandTileArgs {
    const int64_t *rows;
    const int64_t *d_n_rows;
    const VecK<4> *excl;
    const int64_t *tile_off;
    const int32_t *tile_cnt;
    const GroupOut *gout;
    const int64_t *d_n_groups;
    uint8_t *bits; int64_t cap_bits;
    int64_t *pos; int64_t cap_pos;
    BitsParams bp;
    const HugeRef *huge;
    int32_t *huge_count;     // [2]
    int32_t huge_cap;
    int parity;
    int64_t n_tiles;
    int64_t t_base;          // first tile of this launch (0; a segment of a streamed pass: its first chunk)
    uint32_t *h_pos;         // see GroupStore::h_pos (nullptr: positions are not shipped by this kernel)
};
constexpr unsigned kHugeBlocksX = 16, kHugeBlocksY = 16;

// one wavefront per kTailCPW consecutive tiles (see kTailCPW); workgroups beyond the tiles expand the listed huge rows, kHugeBlocksX
// workgroups per row
__host__ __device__ static inline int64_t expand_tile_blocks(int64_t n_tiles) { return (tail_waves(n_tiles) + kEmitWaves - 1) / kEmitWaves; }

__global__ __launch_bounds__(64 * kEmitWaves) URH_EXPAND_OCC void k_expand_tiles(const ExpandTileArgs a) {
    URH_TAIL_PRIO();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int bps = (int)a.bp.bps;
    const int64_t tile_blocks = expand_tile_blocks(a.n_tiles - a.t_base);
    if ((int64_t)blockIdx.x >= tile_blocks) {
        // ---- huge rows ----
        const int64_t n = *a.d_n_rows;
        const bool row0g = !a.bp.d_row_base || *a.bp.d_row_base == 0;
        const int64_t hb = (int64_t)blockIdx.x - tile_blocks;
        const int hx = (int)(hb % kHugeBlocksX), hy = (int)(hb / kHugeBlocksX);
        if (hb == 0 && threadIdx.x == 0) a.huge_count[a.parity ^ 1] = 0;       // the other parity's counter: free for the next pass
        int cnt = a.huge_count[a.parity];
        if (cnt > a.huge_cap) cnt = 0;                      // list overflowed: the tiles' own wavefronts expand every row
        __shared__ int64_t s_h[5];
        for (int w = hy; w < cnt; w += kHugeBlocksY) {
            const HugeRef h = a.huge[w];
            __syncthreads();
            // a bits segment of a streamed pass expands the listed rows of ITS tiles: the rows kernels of later segments may be appending to
            // the list right now (an entry that is not the row of one of this launch's tiles -- another segment's, or one still being
            // written -- is skipped; whatever passes the test is expanded from the row itself, so a duplicate is harmless)
            if (h.tile < a.t_base || h.tile >= a.n_tiles || h.row < a.tile_off[h.tile] || h.row >= a.tile_off[h.tile] + a.tile_cnt[h.tile]) continue;
            if (wave == 0) {
                // prefix of the row inside its tile: walk the tile's rows before it, 64 at a time
                VecK<4> run = a.excl[h.tile];
                const int64_t off = a.tile_off[h.tile];
                for (int64_t i0 = off; i0 <= h.row; i0 += 64) {
                    const int64_t i = i0 + lane;
                    VecK<4> v; v.zero();
                    if (i < h.row && i < n) v = row_value(a.rows[2 * i], a.rows[2 * i + 1], i == 0 && row0g, a.bp);
                    run.add(wave_sum_vec<4>(v));
                }
                if (lane == 0) {
                    int64_t kb = 0, ob = 0, op = 0, ty = 0;
                    if (h.row < n && run.v[1] < *a.d_n_groups) {
                        const GroupOut go = a.gout[run.v[1]];
                        if (go.is_msg) {
                            ty = a.rows[2 * h.row];
                            kb = row_value(ty, a.rows[2 * h.row + 1], h.row == 0 && row0g, a.bp).v[0];
                            ob = go.out_bits + (run.v[0] - go.bits_start);
                            op = go.out_pos + (run.v[0] - go.bits_start);
                        }
                    }
                    s_h[0] = kb; s_h[1] = ob; s_h[2] = op; s_h[3] = run.v[2]; s_h[4] = ty;
                }
            }
            __syncthreads();
            const int64_t kb = s_h[0], ob = s_h[1], op = s_h[2], ts = s_h[3], ty = s_h[4];
            for (int64_t k = (int64_t)hx * blockDim.x + threadIdx.x; k < kb; k += (int64_t)kHugeBlocksX * blockDim.x) {
                const int sh = (bps == 1) ? 0 : bps - 1 - (int)(k % bps);
                const uint8_t b = (ty < 0) ? 0 : (uint8_t)((ty >> sh) & 1);
                if (ob + k < a.cap_bits) a.bits[ob + k] = b;
                if (a.bp.write_pos && op + k < a.cap_pos) {
                    a.pos[op + k] = ts + k * a.bp.samples_per_bit;
                    if (a.h_pos) a.h_pos[op + k] = (uint32_t)(ts + k * a.bp.samples_per_bit);
                }
            }
        }
        return;
    }
    const int64_t t0 = a.t_base + ((int64_t)blockIdx.x * kEmitWaves + wave) * kTailCPW;
    if (t0 >= a.n_tiles) return;
    const int n_mine = (int)((a.n_tiles - t0 < kTailCPW) ? a.n_tiles - t0 : kTailCPW);
    // ---- round trip 1: everything that does not depend on loaded data; lane j holds the metadata of tile t0 + j ----
    const int64_t n = *a.d_n_rows;
    const int64_t n_groups = *a.d_n_groups;
    const bool row0g = !a.bp.d_row_base || *a.bp.d_row_base == 0;
    const int hc = a.huge_count[a.parity];
    int64_t off_l = 0; int32_t tcnt_l = 0;
    VecK<4> run_l; run_l.zero();
    if (lane < n_mine) { off_l = a.tile_off[t0 + lane]; tcnt_l = a.tile_cnt[t0 + lane]; run_l = a.excl[t0 + lane]; }
    const int64_t own_limit = (hc > a.huge_cap) ? INT64_MAX : kHugeBits;   // longer rows are on the list
    // ---- round trip 2: the group every tile starts in (nearly every row of a tile belongs to it), and the first tile's rows ----
    GroupOut go_l; go_l.bits_start = 0; go_l.out_bits = 0; go_l.out_pos = 0; go_l.is_msg = 0; go_l.pad = 0;
    if (lane < n_mine && run_l.v[1] < n_groups) go_l = a.gout[run_l.v[1]];
    const int64_t off_first = lane_bcast(off_l, 0);
    int64_t end_first = off_first + lane_bcast((int)tcnt_l, 0);
    if (end_first > n) end_first = n;
    longlong2 row0 = longlong2{0, 0};
    if (off_first + lane < end_first) row0 = *(const longlong2 *)(a.rows + 2 * (off_first + lane));
    for (int jt = 0; jt < n_mine; ++jt) {
        const int64_t off = lane_bcast(off_l, jt);
        const int32_t tcnt = lane_bcast((int)tcnt_l, jt);
        VecK<4> run;
        run.v[0] = lane_bcast(run_l.v[0], jt); run.v[1] = lane_bcast(run_l.v[1], jt); run.v[2] = lane_bcast(run_l.v[2], jt); run.v[3] = 0;
        GroupOut go0;
        go0.bits_start = lane_bcast(go_l.bits_start, jt); go0.out_bits = lane_bcast(go_l.out_bits, jt); go0.out_pos = lane_bcast(go_l.out_pos, jt);
        go0.is_msg = lane_bcast((int)go_l.is_msg, jt); go0.pad = 0;
        int64_t end = off + tcnt;
        if (end > n) end = n;
        // the next tile's first 64 rows are on their way while this tile is expanded
        longlong2 row_next = longlong2{0, 0};
        if (jt + 1 < n_mine) {
            const int64_t off_n = lane_bcast(off_l, jt + 1);
            int64_t end_n = off_n + lane_bcast((int)tcnt_l, jt + 1);
            if (end_n > n) end_n = n;
            if (off_n + lane < end_n) row_next = *(const longlong2 *)(a.rows + 2 * (off_n + lane));
        }
        const int64_t g0 = run.v[1];
        longlong2 row_cur = row0;
        for (int64_t i0 = off; i0 < end; i0 += 64) {
            const int64_t i = i0 + lane;
            VecK<4> v; v.zero();
            int64_t type = 0;
#if URH_EXPAND_PREFETCH
            // the tile's next 64 rows are requested before these are expanded (dense pulse tables -- 10 samples per symbol: 400 rows per tile --
            // spent a memory round trip per 64 rows here, beside a hot kernel that saturates the HBM: profiles/r06fin_sps10_skips.txt)
            longlong2 row_pf = longlong2{0, 0};
            if (i + 64 < end) row_pf = *(const longlong2 *)(a.rows + 2 * (i + 64));
            if (i < end) {
                type = row_cur.x;
                v = row_value(type, row_cur.y, i == 0 && row0g, a.bp);
            }
            row_cur = row_pf;
#else
            if (i < end) {
                const longlong2 row = (i0 == off) ? row0 : *(const longlong2 *)(a.rows + 2 * i);
                type = row.x;
                v = row_value(type, row.y, i == 0 && row0g, a.bp);
            }
#endif
            // inclusive wave scans of what the expansion needs: bits and samples (64-bit), long pauses (32-bit: a tile has few rows)
            int64_t in_bits = v.v[0], in_ts = v.v[2];
            int in_l = (int)v.v[1];
#if URH_DPP_SCAN
            in_bits = wave_incl_sum_dpp(in_bits); in_ts = wave_incl_sum_dpp(in_ts); in_l = wave_incl_sum_dpp(in_l);
#else
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int64_t ub = __shfl_up(in_bits, o), ut = __shfl_up(in_ts, o);
                const int ul = __shfl_up(in_l, o);
                if (lane >= o) { in_bits += ub; in_ts += ut; in_l += ul; }
            }
#endif
            int64_t kb = 0, ob = 0, op = 0, ts = 0;
            if (i < end && v.v[0] > 0 && v.v[0] <= own_limit) {
                const int64_t grp = run.v[1] + in_l - v.v[1];
                if (grp < n_groups) {
                    GroupOut go = go0;
                    if (grp != g0) go = a.gout[grp];
                    if (go.is_msg) {
                        const int64_t bit_prefix = run.v[0] + in_bits - v.v[0];
                        kb = v.v[0];
                        ob = go.out_bits + (bit_prefix - go.bits_start);
                        op = go.out_pos + (bit_prefix - go.bits_start);
                        ts = run.v[2] + in_ts - v.v[2];
                    }
                }
            }
            constexpr int kShort = 16;
#if URH_EXPAND_EARLY_EXIT
            {
                // a few bits per row (the common case): capacity checks hoisted out of the loop, positions by repeated addition.  The loop ends
                // when NO lane has a bit left (wave-uniform test): written as `for k < kb` the compiler unrolled it into sixteen predicated steps
                // that every 64 rows paid in full -- 71 VALU and 32 memory instructions where a dense table's rows hold one or two bits each
                const int kbs = (kb > 0 && kb <= kShort) ? (int)kb : 0;
                const int nb = (ob + kbs <= a.cap_bits) ? kbs : (int)((a.cap_bits > ob) ? a.cap_bits - ob : 0);
                const int np = !a.bp.write_pos ? 0 : ((op + kbs <= a.cap_pos) ? kbs : (int)((a.cap_pos > op) ? a.cap_pos - op : 0));
                uint8_t *bp8 = a.bits + ob;
                int64_t *pp = a.pos + op;
                uint32_t *hp = a.h_pos ? a.h_pos + op : nullptr;
                int64_t tsk = ts;
                int sh = bps - 1;
#pragma unroll 1
                for (int k = 0; k < kShort; ++k) {
                    if (__ballot(k < kbs) == 0ull) break;
                    const uint8_t b = (type < 0) ? 0 : (uint8_t)((type >> sh) & 1);
                    sh = (sh == 0) ? bps - 1 : sh - 1;
                    if (k < nb) bp8[k] = b;
                    if (k < np) { pp[k] = tsk; if (hp) hp[k] = (uint32_t)tsk; }
                    tsk += a.bp.samples_per_bit;
                }
            }
#else
            if (kb > 0 && kb <= kShort) {
                // a few bits per row (the common case): capacity checks hoisted out of the loop, positions by repeated addition
                const int nb = (ob + kb <= a.cap_bits) ? (int)kb : (int)((a.cap_bits > ob) ? a.cap_bits - ob : 0);
                const int np = !a.bp.write_pos ? 0 : ((op + kb <= a.cap_pos) ? (int)kb : (int)((a.cap_pos > op) ? a.cap_pos - op : 0));
                uint8_t *bp8 = a.bits + ob;
                int64_t *pp = a.pos + op;
                uint32_t *hp = a.h_pos ? a.h_pos + op : nullptr;
                int64_t tsk = ts;
                int sh = bps - 1;
                for (int k = 0; k < (int)kb; ++k) {
                    const uint8_t b = (type < 0) ? 0 : (uint8_t)((type >> sh) & 1);
                    sh = (sh == 0) ? bps - 1 : sh - 1;
                    if (k < nb) bp8[k] = b;
                    if (k < np) { pp[k] = tsk; if (hp) hp[k] = (uint32_t)tsk; }
                    tsk += a.bp.samples_per_bit;
                }
            }
#endif
            unsigned long long big = __ballot(kb > kShort);
            while (big) {
                const int src = __builtin_ctzll(big);
                big &= big - 1;
                const int64_t kb_s = __shfl(kb, src), ob_s = __shfl(ob, src), op_s = __shfl(op, src), ts_s = __shfl(ts, src),
                              ty_s = __shfl(type, src);
                for (int64_t k = lane; k < kb_s; k += 64) {
                    const uint8_t b = symbol_bit(ty_s, k, bps);
                    if (ob_s + k < a.cap_bits) a.bits[ob_s + k] = b;
                    if (a.bp.write_pos && op_s + k < a.cap_pos) {
                        a.pos[op_s + k] = ts_s + k * a.bp.samples_per_bit;
                        if (a.h_pos) a.h_pos[op_s + k] = (uint32_t)(ts_s + k * a.bp.samples_per_bit);
                    }
                }
            }
            run.v[0] += lane_bcast(in_bits, 63); run.v[1] += lane_bcast(in_l, 63); run.v[2] += lane_bcast(in_ts, 63);     // wave-uniform: scalar registers
        }
        row0 = row_next;
    }
}

// ---- host-side launchers ---------------------------------------------------------------------------
int launch_resolve(const ResolveArgs &a, int32_t *tickets, hipStream_t s) {
    if (a.n_chunks <= 0) return URHGPU_ERR_ARG;
    const unsigned g = (unsigned)((a.n_chunks + kResolveBlock - 1) / kResolveBlock);
    (void)tickets;
    hipLaunchKernelGGL(k_resolve_a, dim3(g), dim3(kResolveBlock), 0, s, a);
    hipLaunchKernelGGL(k_resolve_b, dim3(g), dim3(kResolveBlock), 0, s, a);
    hipLaunchKernelGGL(k_resolve_c, dim3(g), dim3(kResolveBlock), 0, s, a);
    return URHGPU_OK;
}

// local pass of a sharded capture: the shard's summary in one launch (partials in the resolve scratch the table pass overwrites)
int launch_shard_summary(const ResolveArgs &a, int32_t *ticket, hipStream_t s) {
    if (a.n_chunks <= 0 || !a.local_pass || !a.summary_out || a.n_chunks >= kAuxNone) return URHGPU_ERR_ARG;
    static_assert(sizeof(ShardPart) <= 256 && sizeof(ShardPart) <= (size_t)kResolveBlock * 8, "one partial per kResolveBlock chunks fits the out_cnt section (8 bytes per chunk, 256-byte granules)");
    const unsigned g = (unsigned)resolve_blocks(a.n_chunks);
    hipLaunchKernelGGL(k_shard_summary, dim3(g), dim3(kResolveBlock), 0, s, a, (ShardPart *)a.sc.out_cnt, ticket);
    return URHGPU_OK;
}

// resolve + rows of a single-GPU capture: k_resolve_a, k_resolve_b, k_emit_rows_fused
int launch_resolve_emit_single(const ResolveArgs &r, const EmitArgs &e, hipStream_t s) {
    if (r.n_chunks <= 0 || r.local_pass || r.chunk_first != 0 || r.n_local != r.n_chunks || e.chunk_first != 0) return URHGPU_ERR_ARG;
    const unsigned g = (unsigned)((r.n_chunks + kResolveBlock - 1) / kResolveBlock);
    hipLaunchKernelGGL(k_resolve_a, dim3(g), dim3(kResolveBlock), 0, s, r);
    hipLaunchKernelGGL(k_resolve_b, dim3(g), dim3(kResolveBlock), 0, s, r);
    hipLaunchKernelGGL(k_emit_rows_fused, dim3((unsigned)r.n_chunks), dim3(64), 0, s, e, r);
    return URHGPU_OK;
}

int launch_emit_rows(const EmitArgs &a, int64_t n_local_chunks, hipStream_t s) {
    if (n_local_chunks > 0) hipLaunchKernelGGL(k_emit_rows, dim3((unsigned)n_local_chunks), dim3(64), 0, s, a);
    return URHGPU_OK;
}

static inline int64_t scan_blocks(int64_t cap) { return (cap + kScanTile - 1) / kScanTile; }
// the group scan runs one element per thread (k_scan_lookback<..., kGroupItems>)
constexpr int kGroupItems = 1;
static inline int64_t group_scan_blocks(int64_t cap) { return (cap + kScanBlock * kGroupItems - 1) / (kScanBlock * kGroupItems); }

// scratch bytes needed by merge_rows_ask for up to cap rows
size_t merge_scratch_bytes(int64_t cap) {
    return (size_t)(scan_blocks(cap) + 1) * sizeof(VecK<2>) + 256 + (size_t)cap * 16 + 512;
}

// rows_in (d_n_in rows) -> rows_out (d_n_out rows); rows_in and rows_out must differ.
int launch_merge_rows_ask(const int64_t *rows_in, const int64_t *d_n_in, int64_t cap, int64_t *rows_out,
                          int64_t cap_out, int64_t *d_n_out, void *scratch, int32_t *tickets, hipStream_t s) {
    if (cap <= 0) return URHGPU_OK;
    const int64_t nb = scan_blocks(cap);
    char *p = (char *)scratch;
    VecK<2> *partials = (VecK<2> *)p; p += ((size_t)(nb + 1) * sizeof(VecK<2>) + 255) & ~size_t(255);
    int64_t *grp_state = (int64_t *)p; p += (size_t)cap * 8;
    int64_t *grp_end = (int64_t *)p;
    MergeLoad ld{rows_in};
    MergeStore st{rows_in, d_n_in, grp_state, grp_end};
    (void)tickets;
    hipLaunchKernelGGL((k_scan_reduce<2, MergeLoad>), dim3(scan_grid(nb)), dim3(kScanBlock), 0, s, d_n_in, ld, partials, nb);
    hipLaunchKernelGGL((k_scan_apply<2, MergeLoad, MergeStore>), dim3(scan_grid(nb)), dim3(kScanBlock), 0, s, d_n_in, ld, partials, nb, st,
                       kScanAlwaysDirect);
    hipLaunchKernelGGL((k_scan_finish<2, NoFinal>), dim3(1), dim3(kScanBlock), 0, s, d_n_in, partials, nb, NoFinal(), kScanAlwaysDirect);
    int64_t fin_blocks = (cap + 255) / 256; if (fin_blocks > 4096) fin_blocks = 4096;
    hipLaunchKernelGGL(k_merge_finish, dim3((unsigned)fin_blocks), dim3(256), 0, s, partials + nb, grp_state, grp_end,
                       rows_out, cap_out, d_n_out);
    return URHGPU_OK;
}

size_t bits_scratch_bytes(int64_t cap_rows) {
    const int64_t nb = scan_blocks(cap_rows);
    const int64_t cap_groups = cap_rows + 1;
    const int64_t nbg = group_scan_blocks(cap_groups);
    return (size_t)(nb + 1) * sizeof(VecK<4>) + (size_t)(nbg + 1) * sizeof(VecK<3>) + (size_t)cap_rows * sizeof(RowInfo) +
           (size_t)cap_groups * (sizeof(GroupInfo) + sizeof(GroupOut)) + 64 + 8 * 256 + 64 + 256 +
           (size_t)kHugeCap * sizeof(HugeRow) + 256;
}

namespace {
struct BitsScratch {
    VecK<4> *part4; VecK<3> *part3; RowInfo *info; GroupInfo *groups; GroupOut *gout; int64_t *d_n_groups;
    HugeRow *huge; int32_t *huge_count;
    int64_t nb, nbg;
};
BitsScratch carve_bits(void *scratch, int64_t cap_rows) {
    BitsScratch b;
    b.nb = scan_blocks(cap_rows);
    const int64_t cap_groups = cap_rows + 1;
    b.nbg = group_scan_blocks(cap_groups);
    char *p = (char *)scratch;
    auto take = [&](size_t bytes) { char *r = p; p += (bytes + 255) & ~size_t(255); return r; };
    b.part4 = (VecK<4> *)take((size_t)(b.nb + 1) * sizeof(VecK<4>));
    b.part3 = (VecK<3> *)take((size_t)(b.nbg + 1) * sizeof(VecK<3>));
    b.info = (RowInfo *)take((size_t)cap_rows * sizeof(RowInfo));
    b.groups = (GroupInfo *)take((size_t)cap_groups * sizeof(GroupInfo));
    b.gout = (GroupOut *)take((size_t)cap_groups * sizeof(GroupOut));
    b.d_n_groups = (int64_t *)take(64);
    b.huge_count = (int32_t *)take(64);
    b.huge = (HugeRow *)take((size_t)kHugeCap * sizeof(HugeRow));
    return b;
}
}  // namespace

// Runs once after the row scan: number of groups = total L rows + 1;
// flags = {long pause present, data before the first one, data after the last one}
struct GroupCountFinal {
    const int64_t *d_n_rows;
    const GroupInfo *groups;
    int64_t *d_n_groups;
    int64_t *d_flags;
    __device__ void operator()(const VecK<4> &grand) const {
        const int64_t n = *d_n_rows;
        const int64_t n_l = (n > 0) ? grand.v[1] : 0;
        *d_n_groups = (n > 0) ? n_l + 1 : 0;
        if (d_flags) {
            d_flags[0] = n_l > 0;
            d_flags[1] = (n > 0) && groups[0].data_end > 0;
            d_flags[2] = (n > 0) && (groups[n_l].data_end - (n_l ? groups[n_l - 1].data_end : 0) > 0);
        }
    }
};

// descriptor memory of the two single-pass scans (rows: K = 4, groups: K = 3)
size_t bits_desc_bytes(int64_t cap_rows) {
    if (cap_rows <= 0) cap_rows = 1;
    return (((size_t)(scan_blocks(cap_rows) + 1) * sizeof(ScanDesc<4>) + 255) & ~size_t(255)) + (size_t)(group_scan_blocks(cap_rows + 1) + 1) * sizeof(ScanDesc<3>) + 512;
}

int launch_bits_prepare(const int64_t *rows, const int64_t *d_n_rows, int64_t cap_rows, const BitsParams &bp,
                        void *scratch, int64_t *d_flags, const ScanState &ss, hipStream_t s) {
    if (cap_rows <= 0) cap_rows = 1;
    if (ss.desc_bytes < bits_desc_bytes(cap_rows)) return URHGPU_ERR_ARG;
    const BitsScratch b = carve_bits(scratch, cap_rows);
    // a capture on one GPU needs no flags for its neighbours: the group count comes out of the row scan itself (last row) and
    // the one-workgroup epilogue launch is dropped
    const bool single = (d_flags == nullptr);
    BitsLoad ld{rows, d_n_rows, bp, single ? b.d_n_groups : nullptr};
    BitsStore st{rows, d_n_rows, b.info, b.groups, bp.d_ts_carry, bp.d_absorbed, single ? b.d_n_groups : nullptr};
    GroupCountFinal fin{d_n_rows, b.groups, b.d_n_groups, d_flags};
    // Two passes for the rows: a single-pass look-back scan over hundreds of workgroups measured 54 us against 48 us for
    // this pair -- every look-back hop is a round trip through memory between XCDs (their L2s are not coherent).
    hipLaunchKernelGGL((k_scan_reduce<4, BitsLoad>), dim3(scan_grid(b.nb)), dim3(kScanBlock), 0, s, d_n_rows, ld, b.part4, b.nb);
    hipLaunchKernelGGL((k_scan_apply<4, BitsLoad, BitsStore>), dim3(scan_grid(b.nb)), dim3(kScanBlock), 0, s, d_n_rows, ld, b.part4, b.nb, st,
                       kScanAlwaysDirect);
    if (!single)
        hipLaunchKernelGGL((k_scan_finish<4, GroupCountFinal>), dim3(1), dim3(kScanBlock), 0, s, d_n_rows, b.part4, b.nb, fin, kScanAlwaysDirect);
    return URHGPU_OK;
}

int launch_bits_finish(const int64_t *rows, const int64_t *d_n_rows, int64_t cap_rows, const BitsParams &bp,
                       const BitsOut &o, void *scratch, const ScanState &ss, hipStream_t s) {
    if (cap_rows <= 0) cap_rows = 1;
    if (ss.desc_bytes < bits_desc_bytes(cap_rows)) return URHGPU_ERR_ARG;
    const BitsScratch b = carve_bits(scratch, cap_rows);
    ScanDesc<3> *desc3 = (ScanDesc<3> *)((char *)ss.desc + (((size_t)(b.nb + 1) * sizeof(ScanDesc<4>) + 255) & ~size_t(255)));
    GroupLoad gl{b.groups, b.d_n_groups, bp.d_extra, bp.is_last_rank, bp.write_pos};
    GroupStore gs{gl, b.gout, o.msg_off, o.pauses, o.pos_off, o.pos, o.cap_msg, o.cap_pos};
    BitsCountsFinal fin{d_n_rows, o.msg_off, o.pos_off, o.counts, b.huge_count, bp.d_rows_needed, o.h_counts};
    // the groups (usually a handful: one workgroup) go through the single-pass look-back scan: one launch instead of two
    if (!(g_tail_skip & 8))
        hipLaunchKernelGGL((k_scan_lookback<3, GroupLoad, GroupStore, BitsCountsFinal, kGroupItems>), dim3(scan_grid(b.nbg)), dim3(kScanBlock), 0, s,
                           b.d_n_groups, gl, desc3, b.nbg, gs, fin, ++*ss.epoch, ss.tickets + 2, 0);
    ExpandArgs ea{rows, d_n_rows, b.info, b.gout, o.bits, o.cap_bits, o.pos, o.cap_pos, bp, b.huge, b.huge_count, kHugeCap};
    const int64_t eb = std::min<int64_t>((cap_rows + 255) / 256, kExpandMaxBlocks);
    hipLaunchKernelGGL(k_expand_bits, dim3((unsigned)eb), dim3(256), 0, s, ea);
    hipLaunchKernelGGL(k_expand_huge, dim3(kHugeGridX, kHugeGridY), dim3(256), 0, s, ea);
    return URHGPU_OK;
}

int launch_ppseq_to_bits(const int64_t *rows, const int64_t *d_n_rows, int64_t cap_rows, const BitsParams &bp,
                         const BitsOut &o, void *scratch, const ScanState &ss, hipStream_t s) {
    URH_TRY(launch_bits_prepare(rows, d_n_rows, cap_rows, bp, scratch, nullptr, ss, s));
    return launch_bits_finish(rows, d_n_rows, cap_rows, bp, o, scratch, ss, s);
}


// ---- tile tail: host side ----------------------------------------------------------------------------------------------
// Measurement hook (urhgpu_test_tail_skip, tools/inrun_anatomy.py): bits 0 .. 5 leave out k_resolve_one, k_emit_rows_tiles, k_tile_scan, the
// group scan, k_expand_tiles, k_pack_seg of the tile tail -- what each of them costs the hot kernel it runs beside.  Results are only
// meaningful when every pass processes the same capture (the buffers then still hold the previous pass's identical outputs).
int g_tail_skip = 0;
namespace {
constexpr int kTileHugeCap = 8192;
struct TileCarve { TileTail ft; HugeRef *huge; };
TileCarve carve_tile(const TileTailMem &m) {
    TileCarve tc;
    char *p = (char *)m.mem;
    auto take = [&](size_t bytes) { char *r = p; p += (bytes + 255) & ~size_t(255); return r; };
    const int64_t nt = m.n_chunks + 1;
    tc.ft.ploc = (ResElem *)take((size_t)m.n_chunks * sizeof(ResElem));
    tc.ft.btot = (ResElem *)take((size_t)(resolve_blocks(m.n_chunks) + 1) * sizeof(ResElem));
    tc.ft.agg = (VecK<4> *)take((size_t)nt * sizeof(VecK<4>));
    tc.ft.excl = (VecK<4> *)take((size_t)nt * sizeof(VecK<4>));
    tc.ft.tile_off = (int64_t *)take((size_t)nt * 8);
    tc.ft.tile_cnt = (int32_t *)take((size_t)nt * 4);
    tc.ft.d_n_tiles = (int64_t *)take(64);
    tc.huge = (HugeRef *)take((size_t)kTileHugeCap * sizeof(HugeRef));
    tc.ft.huge_count = m.huge_count;
    tc.ft.parity = m.parity;
    tc.ft.want_bits = 0;
    tc.ft.rdesc = (GranDesc *)m.rdesc;
    tc.ft.epoch = m.epoch;
    tc.ft.d_row_base = m.d_row_base;
    tc.ft.esc = nullptr;
    return tc;
}
}  // namespace

// the tile scan runs one workgroup per 256 tiles: descriptor memory sized (bits_desc_bytes) for this many "rows" holds them
int64_t tile_desc_cap(int64_t cap_rows, int64_t n_chunks) { return std::max<int64_t>(cap_rows, 8 * (n_chunks + 2) + kScanTile); }

size_t tile_rdesc_bytes(int64_t n_chunks) {        // resolve scan + tile scan descriptors
    return (size_t)(resolve_blocks(n_chunks) + 2 + (n_chunks + 1 + kScanBlock - 1) / kScanBlock + 2) * sizeof(GranDesc);
}

size_t tile_tail_bytes(int64_t n_chunks) {
    const int64_t nt = n_chunks + 1;
    return (size_t)n_chunks * sizeof(ResElem) + (size_t)(resolve_blocks(n_chunks) + 1) * sizeof(ResElem) + (size_t)nt * (2 * sizeof(VecK<4>) + 12) +
           64 + (size_t)kTileHugeCap * sizeof(HugeRef) + 10 * 256;
}

// resolve + rows (+ per-tile bit aggregates when bp != nullptr) of a non-ASK capture: two launches.  Sharded captures pass the table of
// the generic resolve pass (the other shards' summaries around this GPU's chunks: r.chunk_first, r.n_local) and m.d_row_base.
int launch_tile_rows(const ResolveArgs &r, const EmitArgs &e, const TileTailMem &m, const BitsParams *bp, hipStream_t s) {
    if (r.n_chunks <= 0 || r.local_pass || r.chunk_first != e.chunk_first || r.chunk_first < 0 || r.chunk_first + r.n_local > r.n_chunks ||
        m.n_chunks != r.n_chunks || e.is_ask)
        return URHGPU_ERR_ARG;
    if ((r.n_local != r.n_chunks) != (m.d_row_base != nullptr)) return URHGPU_ERR_ARG;
    TileCarve tc = carve_tile(m);
    EmitTileArgs g;
    g.e = e; g.r = r; g.ft = tc.ft; g.huge = tc.huge; g.huge_cap = kTileHugeCap;
    memset(&g.bp, 0, sizeof(g.bp));
    if (bp) { g.bp = *bp; g.ft.want_bits = 1; }
    const unsigned gb = (unsigned)resolve_blocks(r.n_chunks);
    g.w0 = 0; g.w_end = tail_waves(r.n_chunks); g.final_seg = 1; g.seg = nullptr; g.seg_k = 0; g.h_state = nullptr; g.h_len = nullptr; g.len16 = 0; g.esc = nullptr; g.esc_cap = 0; g.ft.esc = nullptr;
    SegGate no_gate;
    memset(&no_gate, 0, sizeof(no_gate));
    if (!(g_tail_skip & 1)) hipLaunchKernelGGL(k_resolve_one, dim3(gb), dim3(kResolveBlock), 0, s, r, g.ft, (int64_t)0, no_gate);
    if (!(g_tail_skip & 2)) hipLaunchKernelGGL(k_emit_rows_tiles, dim3((unsigned)((tail_waves(r.n_chunks) + 1 + kEmitWaves - 1) / kEmitWaves)), dim3(64 * kEmitWaves), 0, s, g);
    return URHGPU_OK;
}

// bits / pauses / bit_sample_pos from the tiles launch_tile_rows left behind: tile scan, group scan, expansion.
// Sharded captures stop after the tile scan for the flags exchange (d_flags: what GroupCountFinal gives the generic path).
namespace {
__global__ void k_tile_flags(const int64_t *d_n_rows, const GroupInfo *groups, const int64_t *d_n_groups, int64_t *d_flags) {
    const int64_t n = *d_n_rows, ng = *d_n_groups;
    const int64_t n_l = (n > 0 && ng > 0) ? ng - 1 : 0;
    d_flags[0] = n_l > 0;
    d_flags[1] = (n > 0) && groups[0].data_end > 0;
    d_flags[2] = (n > 0) && (groups[n_l].data_end - (n_l ? groups[n_l - 1].data_end : 0) > 0);
}
}  // namespace

int launch_tile_bits_prepare(const TileTailMem &m, const int64_t *rows, const int64_t *d_n_rows, int64_t cap_rows, const BitsParams &bp,
                             void *scratch, int64_t *d_flags, const ScanState &ss, hipStream_t s) {
    if (cap_rows <= 0) cap_rows = 1;
    const int64_t nt = m.n_chunks + 1;
    const int64_t cap_desc = tile_desc_cap(cap_rows, m.n_chunks);
    if (ss.desc_bytes < bits_desc_bytes(cap_desc)) return URHGPU_ERR_ARG;
    const TileCarve tc = carve_tile(m);
    const BitsScratch b = carve_bits(scratch, cap_rows);
    const int64_t cap_groups = cap_rows + 1;
    GranDesc *tdesc = (GranDesc *)m.rdesc + resolve_blocks(m.n_chunks) + 2;
    TileScanArgs ta{rows, d_n_rows, tc.ft.agg, tc.ft.tile_off, tc.ft.tile_cnt, tc.ft.excl, b.groups, cap_groups, b.d_n_groups, nt, bp, tdesc,
                    m.epoch, 0, nullptr, 0};
    const unsigned nb_ts = (unsigned)((nt + kScanBlock - 1) / kScanBlock);
    if (!(g_tail_skip & 4)) hipLaunchKernelGGL(k_tile_scan, dim3(nb_ts), dim3(kScanBlock), 0, s, ta);
    if (d_flags) hipLaunchKernelGGL(k_tile_flags, dim3(1), dim3(1), 0, s, d_n_rows, b.groups, b.d_n_groups, d_flags);
    return URHGPU_OK;
}

int launch_tile_bits_finish(const TileTailMem &m, const int64_t *rows, const int64_t *d_n_rows, int64_t cap_rows, const BitsParams &bp,
                            const BitsOut &o, void *scratch, const ScanState &ss, hipStream_t s) {
    if (cap_rows <= 0) cap_rows = 1;
    const int64_t nt = m.n_chunks + 1;
    const int64_t cap_desc = tile_desc_cap(cap_rows, m.n_chunks);
    if (ss.desc_bytes < bits_desc_bytes(cap_desc)) return URHGPU_ERR_ARG;
    const TileCarve tc = carve_tile(m);
    const BitsScratch b = carve_bits(scratch, cap_rows);
    ScanDesc<3> *desc3 = (ScanDesc<3> *)((char *)ss.desc + (((size_t)(scan_blocks(cap_desc) + 1) * sizeof(ScanDesc<4>) + 255) & ~size_t(255)));
    GroupLoad gl{b.groups, b.d_n_groups, bp.d_extra, bp.is_last_rank, bp.write_pos};
    GroupStore gs{gl, b.gout, o.msg_off, o.pauses, o.pos_off, o.pos, o.cap_msg, o.cap_pos};
    BitsCountsFinal fin{d_n_rows, o.msg_off, o.pos_off, o.counts, b.huge_count, bp.d_rows_needed, o.h_counts};
    if (!(g_tail_skip & 8))
        hipLaunchKernelGGL((k_scan_lookback<3, GroupLoad, GroupStore, BitsCountsFinal, kGroupItems>), dim3(scan_grid(b.nbg)), dim3(kScanBlock), 0, s,
                           b.d_n_groups, gl, desc3, b.nbg, gs, fin, ++*ss.epoch, ss.tickets + 2, 0);
    ExpandTileArgs ea{rows, d_n_rows, tc.ft.excl, tc.ft.tile_off, tc.ft.tile_cnt, b.gout, b.d_n_groups, o.bits, o.cap_bits, o.pos, o.cap_pos,
                      bp, tc.huge, m.huge_count, kTileHugeCap, m.parity, nt, 0, nullptr};
    const unsigned tile_blocks = (unsigned)expand_tile_blocks(nt);
    if (!(g_tail_skip & 16)) hipLaunchKernelGGL(k_expand_tiles, dim3(tile_blocks + kHugeBlocksX * kHugeBlocksY), dim3(64 * kEmitWaves), 0, s, ea);
    return URHGPU_OK;
}

int launch_tile_bits(const TileTailMem &m, const int64_t *rows, const int64_t *d_n_rows, int64_t cap_rows, const BitsParams &bp,
                     const BitsOut &o, void *scratch, const ScanState &ss, hipStream_t s) {
    URH_TRY(launch_tile_bits_prepare(m, rows, d_n_rows, cap_rows, bp, scratch, nullptr, ss, s));
    return launch_tile_bits_finish(m, rows, d_n_rows, cap_rows, bp, o, scratch, ss, s);
}

// =====================================================================================================
// Segments: the tile tail of a STREAMED pass (urhgpu_stream_*).
//
// One capture on an otherwise idle GPU used to be hot kernel, then tail, then pack, then copy -- 0.45 ms for a 1 GiB capture whose
// hot kernel takes 0.28 (SURVEY 8(d)'s window ends with the compact outputs on the host).  Everything behind the hot kernel is
// CAUSAL: the state machine before a chunk, its row offset, the bits / pauses / samples before a tile and the messages before a group
// are prefix compositions.  So the capture's chunks are cut into a few SEGMENTS (boundaries on kSegAlign chunks), and segment k's tail
// -- the same five kernels over its range of chunks / tiles, then a pack kernel that stores the segment's share of the compact blob
// straight into pinned host memory -- runs while the hot kernel is still working on the chunks behind it:
//   * the hot kernel (ONE launch: kernel boundaries cost 10 us each) writes its records through to memory and counts every finished
//     chunk into the counter of its segment (demod_runs.hip: RunArgs::progress); k_seg_gate, first on the tail stream, waits for it.
//     A segment's counter also covers the FIRST chunk of the next segment: the resolve kernel looks one chunk ahead (does a trailing
//     short run continue?);
//   * resolve / rows: workgroup totals (btot) and per-tile aggregates of the earlier segments are in memory: the later segments'
//     wavefronts fold / look back into them exactly as the later workgroups of a single launch do;
//   * rows are final as soon as they are written; a GROUP (message candidate) may still be open at the end of a segment: its rows are
//     expanded where its bits would go (GroupLoad::tentative_open), and the pack kernel of the next segment ships again from the
//     start of a group that had no data row yet -- should it be dropped after all, the next kept group overwrites its bits (SegFinal);
//   * per segment the host receives: its rows, the bytes of packed bits that are complete, the positions, and the messages closed.
// Results are bit-identical to the one-launch tail (tests/test_stream_segments.py compares them on random captures and boundaries).
// =====================================================================================================
// The gate of a rows segment: ONE wavefront, first on the tail stream, polls the segment's progress counter (its own 128-byte line) a few
// times per microsecond until the hot kernel -- still running -- has counted the segment's chunks in; the chunks' records were written
// through to memory and acknowledged before they counted (demod_runs.hip), and the kernels behind the gate start with an empty cache.
// (Polling from every workgroup of the resolve kernel instead was measured: 8 to 32 workgroups hammering the counters' line slowed the
// hot kernel from 0.28 to 0.5 - 0.9 ms.)
__global__ void k_seg_gate(const SegGate gate) {
    if (threadIdx.x != 0) return;
    if (gate.init) gate.seg->err = 0;
    if (!gate.progress) return;
    const uint32_t *ctr = gate.progress + gate.k * kProgressStride;
    const long long t0 = (long long)wall_clock64();
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gate.target) {
        __builtin_amdgcn_s_sleep(16);
        if ((long long)wall_clock64() - t0 > gate.max_ticks) { gate.seg->err = 1 + gate.k; break; }       // (the hot kernel never came: a bug, not a hang)
    }
}

struct SegGroupLoad {
    GroupLoad ld;
    const SegState *seg;
    int parity;
    __device__ VecK<3> operator()(int64_t i) const { return ld(seg->in[parity].g0 + i); }
};
struct SegGroupStore {
    GroupStore st;
    const SegState *seg;
    int parity;
    __device__ void operator()(int64_t i, const VecK<3> &val, const VecK<3> &ex) const {
        const int64_t g0 = seg->in[parity].g0;
        VecK<3> e = ex;
#pragma unroll
        for (int k = 0; k < 3; ++k) e.v[k] += seg->in[parity].carry[k];
        st(g0 + i, val, e);
    }
};
// after a bits segment's group scan: what the next one starts from, what the pack kernel ships; the last segment: the pass's counts
struct SegFinal {
    SegGroupLoad ld;
    SegState *seg;
    const int64_t *d_n_rows;
    int parity;
    int final;
    BitsCountsFinal counts;
    __device__ void operator()(const VecK<3> &grand) const {
        const SegState::In in = seg->in[parity];
        const int64_t nl = seg->n_groups_local;
        VecK<3> tot;
#pragma unroll
        for (int k = 0; k < 3; ++k) tot.v[k] = in.carry[k] + grand.v[k];
        if (final) {
            counts(tot);
            const bool any = *d_n_rows > 0;
            seg->end_msgs = any ? tot.v[0] : 0; seg->end_bits = any ? tot.v[1] : 0; seg->end_pos = any ? tot.v[2] : 0;
            return;
        }
        SegState::In nx = in;
        if (nl <= 0) {                                        // no row yet
            seg->end_msgs = 0; seg->end_bits = 0; seg->end_pos = 0;
        } else {
            const int64_t g_open = in.g0 + nl - 1;
            const VecK<3> vo = ld(nl - 1);
            VecK<3> exo;
#pragma unroll
            for (int k = 0; k < 3; ++k) exo.v[k] = tot.v[k] - vo.v[k];
            const int64_t d0 = g_open ? ld.ld.groups[g_open - 1].data_end : 0;
            const bool has_data = ld.ld.groups[g_open].data_end - d0 > 0;
            nx.g0 = g_open;
#pragma unroll
            for (int k = 0; k < 3; ++k) nx.carry[k] = exo.v[k];
            seg->end_msgs = exo.v[0]; seg->end_bits = exo.v[1] + vo.v[1]; seg->end_pos = exo.v[2] + vo.v[2];
            // an open group without a data row may still be dropped: the next segment ships from its start again
            nx.ship_msgs = exo.v[0];
            nx.ship_bits = has_data ? exo.v[1] + vo.v[1] : exo.v[1];
            nx.ship_pos = has_data ? exo.v[2] + vo.v[2] : exo.v[2];
        }
        seg->in[parity ^ 1] = nx;
    }
};

// A bits segment's share of the compact blob (include/urhgpu.h), stored straight into the pinned host blob: the sections sit at the
// offsets the CAPACITIES give (blob_layout of the capacities), so every segment knows where its elements go and the host needs no
// assembly (the rows went there as they were written: k_emit_rows_tiles); the last segment writes the header.
struct SegPack {
    const uint8_t *bits; const int64_t *msg_off, *pauses, *pos_off, *pos; const int64_t *counts;
    int64_t cap_rows, cap_bits, cap_msg, cap_pos;
    int has_pos;
    char *host;
    BlobLayout L;            // blob_layout(capacities)
    const SegState *seg;
    const int64_t *d_n_rows;
    int parity, final;
    uint32_t *progress;      // the last segment zeroes the pass's counters for the next pass on this arena
    int pos_shipped;         // the positions have been stored into the host blob by the kernels that wrote them (GroupStore::h_pos)
    int split;               // staged pass (one segment): the small sections go tight behind the header, L's row / position offsets are the split layout's
    char *head;              // where header and small sections go (split: the pinned host blob; else == host)
    const int64_t *esc;      // 16-bit row lengths: the escape list to append to the head (behind the packed bits); nullptr: int32 lengths
    int64_t esc_cap;
    int64_t row16_esc_off;   // > 0: URHGPU_BLOB_ROW16 -- the escape list goes to this offset of the host blob instead (header[11] names it)
};
__global__ __launch_bounds__(256) void k_pack_seg(const SegPack a) {
    URH_TAIL_PRIO();
    const int64_t gtid = blockIdx.x * 256ll + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    const SegState::In in = a.seg->in[a.parity];
    auto lim = [](int64_t x, int64_t cap) { return x < cap ? (x < 0 ? 0 : x) : cap; };
    const int64_t nbits = lim(a.seg->end_bits, a.cap_bits);
    const int64_t j0 = lim(in.ship_bits, a.cap_bits) / 8, j1 = a.final ? (nbits + 7) / 8 : nbits / 8;
    const int64_t p0 = a.has_pos ? lim(in.ship_pos, a.cap_pos) : 0, p1 = a.has_pos ? lim(a.seg->end_pos, a.cap_pos) : 0;
    const int64_t m0 = lim(in.ship_msgs, a.cap_msg), m1 = lim(a.seg->end_msgs, a.cap_msg);
    BlobLayout L = a.L;
    if (a.split) {                                         // (one segment: every count is final here)
        const int64_t c5[5] = {0, m1, nbits, 0, 0};
        const BlobLayout T = blob_layout(c5, a.cap_rows, a.cap_bits, a.cap_msg, a.cap_pos, 0);
        L.off_pauses = T.off_pauses; L.off_msg_off = T.off_msg_off; L.off_pos_off = T.off_pos_off; L.off_bits = T.off_bits;
    }
    {
        uint8_t *out = (uint8_t *)(a.head + L.off_bits);
        const unsigned long long *in8 = (const unsigned long long *)a.bits;
        auto pack8 = [&](int64_t j) -> unsigned long long {   // bits 8 j .. 8 j + 7 as one byte
            unsigned long long w;
            if (8 * j + 8 <= nbits) w = in8[j];
            else { w = 0; for (int k = 0; k < 8 && 8 * j + k < nbits; ++k) w |= (unsigned long long)a.bits[8 * j + k] << (8 * k); }
            return ((w & 0x0101010101010101ull) * 0x8040201008040201ull) >> 56;
        };
#if URH_PACK_WORDS
        // The destination is pinned HOST memory: one byte per thread made every wavefront's store a 64-byte PCIe write -- 1.7 MB of them for a
        // dense pulse table (10 samples per symbol: 13 M bits), 100 us beside the hot kernel.  Eight bytes per thread instead: the bytes up to
        // the destination's first 8-byte boundary, whole 64-bit words (512 contiguous bytes per wavefront store), the rest.
        int64_t ja = j0 + (int64_t)((8 - ((uintptr_t)(out + j0) & 7)) & 7);
        if (ja > j1) ja = j1;
        int64_t nw = (j1 - ja) / 8;
        const int64_t full = nbits / 8;                     // bytes whose eight bits all exist
        if (ja + 8 * nw > full) nw = full > ja ? (full - ja) / 8 : 0;
        for (int64_t k = gtid; k < nw; k += stride) {
            const int64_t j = ja + 8 * k;
            unsigned long long r = 0;
#pragma unroll
            for (int b = 0; b < 8; ++b) r |= (((in8[j + b] & 0x0101010101010101ull) * 0x8040201008040201ull) >> 56) << (8 * b);
            *(unsigned long long *)(out + j) = r;
        }
        for (int64_t j = j0 + gtid; j < ja; j += stride) out[j] = (uint8_t)pack8(j);
        for (int64_t j = ja + 8 * nw + gtid; j < j1; j += stride) out[j] = (uint8_t)pack8(j);
#else
        for (int64_t j = j0 + gtid; j < j1; j += stride) out[j] = (uint8_t)pack8(j);
#endif
    }
    if (a.has_pos && !a.pos_shipped) {
        uint32_t *p32 = (uint32_t *)(a.host + L.off_pos32);
        for (int64_t i = p0 + gtid; i < p1; i += stride) p32[i] = (uint32_t)a.pos[i];
    }
    {
        int64_t *pa = (int64_t *)(a.head + L.off_pauses), *mo = (int64_t *)(a.head + L.off_msg_off), *po = (int64_t *)(a.head + L.off_pos_off);
        if (gtid == 0 && m0 == 0) { mo[0] = 0; po[0] = 0; }
        for (int64_t i = m0 + gtid; i < m1; i += stride) { pa[i] = a.pauses[i]; mo[i + 1] = a.msg_off[i + 1]; po[i + 1] = a.pos_off[i + 1]; }
    }
    int64_t n_esc = 0;
    if (a.esc && a.final) {
        // 16-bit row lengths: the escape list goes behind the packed bits -- int64 count, then the {row, length} pairs
        n_esc = a.esc[0];
        const bool over = n_esc > a.esc_cap;
        if (over) n_esc = a.esc_cap;
        int64_t *dst = (int64_t *)(a.head + (a.row16_esc_off > 0 ? a.row16_esc_off : ((L.off_bits + (nbits + 7) / 8 + 15) & ~int64_t(15))));
        if (gtid == 0) dst[0] = over ? -n_esc : n_esc;            // (negative: the list overflowed -- cannot happen for esc_cap = n / 65535 + 2)
        for (int64_t i = gtid; i < n_esc; i += stride) dst[1 + i] = a.esc[1 + i];
    }
    if (a.final && gtid == 0) {
        int64_t *hdr = (int64_t *)a.head;
        const int64_t *c = a.counts;
        const int64_t n_rows = lim(*a.d_n_rows, a.cap_rows);
        hdr[1] = n_rows; hdr[2] = m1; hdr[3] = nbits; hdr[4] = p1; hdr[5] = c[4];
        const bool row16 = a.esc && a.row16_esc_off > 0;
        hdr[6] = URHGPU_BLOB_HEADER_BYTES + n_rows * (row16 ? 2 : (a.esc ? 3 : 5)) + (a.esc ? 8 + n_esc * 8 : 0) + (nbits + 7) / 8 + p1 * 4 + m1 * 24 + 16;     // bytes that crossed PCIe for this pass
        hdr[7] = a.has_pos | (row16 ? URHGPU_BLOB_ROW16 : (a.esc ? URHGPU_BLOB_LEN16 : 0));
        hdr[8] = L.off_pauses; hdr[9] = L.off_msg_off; hdr[10] = L.off_pos_off; hdr[11] = row16 ? a.row16_esc_off : L.off_row_state; hdr[12] = L.off_bits;
        hdr[13] = L.off_row_len; hdr[14] = L.off_pos32;
        hdr[15] = ((c[1] > a.cap_msg || c[2] > a.cap_bits || (a.has_pos && c[3] > a.cap_pos) || c[4] > a.cap_rows) ? 1 : 0) | (a.seg->err ? 2 : 0);
        hdr[0] = URHGPU_BLOB_MAGIC;
    }
    if (a.final && a.progress && gtid < kMaxSegments) a.progress[gtid * kProgressStride] = 0;
}

// A rows segment's two kernels on stream s: resolve (with the gate) and rows (shipped to the host as they are written).
int launch_rows_segment(const ResolveArgs &r, const EmitArgs &e, const TileTailMem &m, const BitsParams &bp, SegState *state, const RowsSegment &sg,
                        hipStream_t s) {
    if (r.n_chunks <= 0 || r.local_pass || r.chunk_first != 0 || e.chunk_first != 0 || r.n_local != r.n_chunks || m.n_chunks != r.n_chunks ||
        e.is_ask || m.d_row_base != nullptr || !state || sg.index < 0 || sg.index >= kMaxSegments)
        return URHGPU_ERR_ARG;
    if (sg.c0 < 0 || sg.c0 % kSegAlign || sg.c1 <= sg.c0 || sg.c1 > r.n_chunks || (sg.final ? sg.c1 != r.n_chunks : (sg.c1 % kSegAlign != 0 || sg.c1 >= r.n_chunks)))
        return URHGPU_ERR_ARG;
    static_assert(kSegAlign % kResolveBlock == 0 && kSegAlign % kScanBlock == 0 && kSegAlign % (kTailCPW * kEmitWaves) == 0, "segment alignment");
    if (sg.final && (r.d_n_rows != &state->rows_at[sg.index] || r.d_n_rows_needed != &state->rows_needed)) return URHGPU_ERR_ARG;
    if ((sg.h_state == nullptr) != (sg.h_len == nullptr) || sg.gate.seg != state) return URHGPU_ERR_ARG;
    const TileCarve tc = carve_tile(m);
    EmitTileArgs g;
    g.e = e; g.r = r; g.ft = tc.ft; g.huge = tc.huge; g.huge_cap = kTileHugeCap;
    g.bp = bp; g.ft.want_bits = 1;
    g.w0 = sg.c0 / kTailCPW; g.w_end = sg.final ? tail_waves(r.n_chunks) : sg.c1 / kTailCPW; g.final_seg = sg.final; g.seg = state; g.seg_k = sg.index;
    g.h_state = sg.h_state; g.h_len = sg.h_len;
    g.len16 = (sg.len16 && sg.esc && sg.h_len) ? sg.len16 : 0; g.esc = g.len16 ? sg.esc : nullptr; g.esc_cap = sg.esc_cap;
    g.ft.esc = (g.len16 && sg.index == 0) ? sg.esc : nullptr;          // (the pass's first rows segment resets the list)
    if (g_tail_skip & 64) { g.h_state = nullptr; g.h_len = nullptr; }      // measurement: the row kernel without its host stores
    if (g_tail_skip & 128) g.e.rows = nullptr;                              // measurement: ... without the int64 table
    // the huge-row list is consumed by the bits segment that covers these tiles, which may cover several rows segments: only the pass's
    // first rows segment starts the list (k_resolve_one clears the counter when want_bits is set)
    TileTail ft_resolve = g.ft;
    if (sg.index != 0) ft_resolve.want_bits = 0;
    SegGate gate = sg.gate;
    gate.fused = (sg.fuse_gate && gate.progress && !gate.init && resolve_blocks(sg.c1 - sg.c0) == 1) ? 1 : 0;
    if (!gate.fused && gate.progress) hipLaunchKernelGGL(k_seg_gate, dim3(1), dim3(64), 0, s, gate);
    if (!(g_tail_skip & 1)) hipLaunchKernelGGL(k_resolve_one, dim3((unsigned)resolve_blocks(sg.c1 - sg.c0)), dim3(kResolveBlock), 0, s, r, ft_resolve, sg.c0 / kResolveBlock, gate);
    if (!(g_tail_skip & 2)) hipLaunchKernelGGL(k_emit_rows_tiles, dim3((unsigned)((g.w_end - g.w0 + 1 + kEmitWaves - 1) / kEmitWaves)), dim3(64 * kEmitWaves), 0, s, g);
    return URHGPU_OK;
}

// A bits segment's kernels on stream s (which has waited for the rows of the chunks below sg.c1): tile scan, group scan, expansion, pack.
int launch_bits_segment(const TileTailMem &m, const BitsParams &bp, const BitsOut &o, void *scratch, const ScanState &ss, int64_t *rows,
                        int64_t cap_rows, const BitsSegment &sg, const SegPackDst *dst, hipStream_t s) {
    if (m.n_chunks <= 0 || m.d_row_base != nullptr || !sg.state || sg.rows_index < 0 || sg.rows_index >= kMaxSegments) return URHGPU_ERR_ARG;
    if (sg.c0 < 0 || sg.c0 % kSegAlign || sg.c1 <= sg.c0 || sg.c1 > m.n_chunks || (sg.final ? sg.c1 != m.n_chunks : (sg.c1 % kSegAlign != 0 || sg.c1 >= m.n_chunks)))
        return URHGPU_ERR_ARG;
    if (cap_rows <= 0) cap_rows = 1;
    const int64_t cap_desc = tile_desc_cap(cap_rows, m.n_chunks);
    if (ss.desc_bytes < bits_desc_bytes(cap_desc)) return URHGPU_ERR_ARG;
    const TileCarve tc = carve_tile(m);
    const BitsScratch b = carve_bits(scratch, cap_rows);
    SegState *st = sg.state;
    const int64_t *d_n_rows = &st->rows_at[sg.rows_index];
    const int parity = sg.index & 1;
    // tile scan over the segment's tiles (the last segment: + the tile of the table's last row)
    const int64_t t1 = sg.final ? m.n_chunks + 1 : sg.c1;
    const int64_t cap_groups = cap_rows + 1;
    GranDesc *tdesc = (GranDesc *)m.rdesc + resolve_blocks(m.n_chunks) + 2;
    TileScanArgs ta{rows, d_n_rows, tc.ft.agg, tc.ft.tile_off, tc.ft.tile_cnt, tc.ft.excl, b.groups, cap_groups, &st->n_groups, t1, bp, tdesc,
                    m.epoch, sg.c0 / kScanBlock, st, parity};
    if (!(g_tail_skip & 4)) hipLaunchKernelGGL(k_tile_scan, dim3((unsigned)((t1 - sg.c0 + kScanBlock - 1) / kScanBlock)), dim3(kScanBlock), 0, s, ta);
    // groups [the one that was open before this segment, the one that is open now]
    ScanDesc<3> *desc3 = (ScanDesc<3> *)((char *)ss.desc + (((size_t)(scan_blocks(cap_desc) + 1) * sizeof(ScanDesc<4>) + 255) & ~size_t(255)));
    GroupLoad gl{b.groups, &st->n_groups, nullptr, sg.final ? 1 : 0, bp.write_pos, sg.final ? 0 : 1};
    // A pass of ONE segment (direct passes) that ships positions: the kernels that write them -- group scan (the sentinels behind a
    // message's bits) and expansion -- also store them as uint32 into the pinned host blob's pos32 section, spread over their run time like
    // the rows; the pack kernel then has no positions to move.  (With several segments an open group is expanded tentatively and may be
    // shipped again: those passes' positions go through the pack kernel.)
    const int has_pos = (bp.write_pos && o.pos) ? 1 : 0;
    const int64_t caps[5] = {cap_rows, o.cap_msg, o.cap_bits, o.cap_pos, cap_rows};
    BlobLayout L = blob_layout(caps, cap_rows, o.cap_bits, o.cap_msg, o.cap_pos, has_pos);
    const bool split = dst && dst->host && dst->split;
    if (split) {
        if (!sg.final || sg.c0 != 0) return URHGPU_ERR_ARG;             // the split layout's head is written by ONE segment
        const StagedLayout SL = staged_layout(cap_rows, o.cap_bits, o.cap_msg, o.cap_pos, has_pos);
        L.off_row_state = SL.off_row_state; L.off_row_len = SL.off_row_len; L.off_pos32 = SL.off_pos32;
    }
    const bool pos_direct = dst && dst->host && has_pos && sg.final && sg.c0 == 0 && dst->pos_direct;
    uint32_t *h_pos = pos_direct ? (uint32_t *)((char *)dst->host + L.off_pos32) : nullptr;
    GroupStore gs{gl, b.gout, o.msg_off, o.pauses, o.pos_off, o.pos, o.cap_msg, o.cap_pos, h_pos};
    SegGroupLoad sl{gl, st, parity};
    SegGroupStore sst{gs, st, parity};
    SegFinal fin{sl, st, d_n_rows, parity, sg.final ? 1 : 0, BitsCountsFinal{d_n_rows, o.msg_off, o.pos_off, o.counts, b.huge_count, &st->rows_needed, o.h_counts}};
    if (!(g_tail_skip & 8))
        hipLaunchKernelGGL((k_scan_lookback<3, SegGroupLoad, SegGroupStore, SegFinal, kGroupItems>), dim3((unsigned)std::min<int64_t>(scan_grid(b.nbg), 32)),
                           dim3(kScanBlock), 0, s, &st->n_groups_local, sl, desc3, b.nbg, sst, fin, ++*ss.epoch, ss.tickets + 2, 0);
    ExpandTileArgs ea{rows, d_n_rows, tc.ft.excl, tc.ft.tile_off, tc.ft.tile_cnt, b.gout, &st->n_groups, o.bits, o.cap_bits, o.pos, o.cap_pos,
                      bp, tc.huge, m.huge_count, kTileHugeCap, m.parity, t1, sg.c0, h_pos};
    const unsigned tile_blocks = (unsigned)expand_tile_blocks(t1 - sg.c0);
    if (g_tail_skip & 2048) ea.cap_bits = 0;                 // measurement: the expansion without its byte stores
    if (!(g_tail_skip & 16)) hipLaunchKernelGGL(k_expand_tiles, dim3(tile_blocks + kHugeBlocksX * kHugeBlocksY), dim3(64 * kEmitWaves), 0, s, ea);
    if (dst && dst->host) {
        if (((uintptr_t)o.bits & 7) || ((uintptr_t)dst->host & 15)) return URHGPU_ERR_ARG;
        SegPack pk{o.bits, o.msg_off, o.pauses, o.pos_off, o.pos, o.counts, cap_rows, o.cap_bits, o.cap_msg, o.cap_pos, has_pos, (char *)dst->host,
                   L, st, d_n_rows, parity, sg.final ? 1 : 0, sg.final ? dst->progress_reset : nullptr, pos_direct ? 1 : 0, split ? 1 : 0,
                   (split && dst->host_head) ? (char *)dst->host_head : (char *)dst->host, split ? dst->esc : nullptr, dst->esc_cap,
                   (split && dst->esc) ? dst->row16_esc_off : 0};
        if (dst->cap_host < pk.L.total) return URHGPU_ERR_CAPACITY;
        if (!(g_tail_skip & 32)) hipLaunchKernelGGL(k_pack_seg, dim3(dst->blocks > 0 ? dst->blocks : 32), dim3(256), 0, s, pk);
    }
    return URHGPU_OK;
}

// ---- sharded captures: the tiny cross-shard fix-ups (one thread each; world <= a few dozen) ---------------
__global__ void k_merge_summary(const int64_t *rows, const int64_t *d_n_rows, int64_t *out5) {
    const int64_t n = *d_n_rows;
    out5[0] = n;
    out5[1] = n ? rows[0] : 0; out5[2] = n ? rows[1] : 0;
    out5[3] = n ? rows[2 * (n - 1)] : 0; out5[4] = n ? rows[2 * (n - 1) + 1] : 0;
}
void launch_merge_summary(const int64_t *rows, const int64_t *d_n_rows, int64_t *d_out5, hipStream_t s) {
    hipLaunchKernelGGL(k_merge_summary, dim3(1), dim3(1), 0, s, rows, d_n_rows, d_out5);
}

// ASK: rows of equal state merge across shard boundaries (signal_functions.pyx:475-480).  The merged row belongs
// to the rank where it starts: my last row absorbs the first rows of the following ranks while their state
// matches; my first row is marked absorbed when it continues the previous rank's last row.
__global__ void k_merge_fix(int64_t *rows, const int64_t *d_n_rows, const int64_t *all, int rank, int world, int64_t *d_absorbed) {
    const int64_t n = *d_n_rows;
    *d_absorbed = -1;
    if (n == 0) return;
    auto M = [&](int q, int k) { return all[5 * q + k]; };
    int prev = -1;
    for (int q = rank - 1; q >= 0; --q) if (M(q, 0) > 0) { prev = q; break; }
    const bool absorbed = prev >= 0 && M(prev, 3) == rows[0];
    if (!(absorbed && n == 1)) {
        const int64_t last_state = rows[2 * (n - 1)];
        for (int q = rank + 1; q < world; ++q) {
            if (M(q, 0) == 0) continue;
            if (M(q, 1) != last_state) break;
            rows[2 * (n - 1) + 1] += M(q, 2);
            if (M(q, 0) > 1) break;
        }
    }
    if (absorbed) {
        if (n == 1 && rows[0] == -1) {
            // my only row continues a pause owned by an earlier rank: the reference's last-row pause
            // (ProtocolAnalyzer.py:411) is the owner's last row plus everything absorbed into it
            int64_t tot = rows[1];
            for (int q = rank + 1; q < world; ++q) tot += (M(q, 0) > 0) ? M(q, 2) : 0;
            for (int q = rank - 1; q >= 0; --q) {
                if (M(q, 0) == 0) continue;
                tot += M(q, 4);
                if (M(q, 0) > 1) break;
                int pq = -1;
                for (int u = q - 1; u >= 0; --u) if (M(u, 0) > 0) { pq = u; break; }
                if (!(pq >= 0 && M(pq, 3) == M(q, 1))) break;
            }
            *d_absorbed = tot;
        }
        rows[0] = kRowAbsorbed;
    }
}
void launch_merge_fix(int64_t *rows, const int64_t *d_n_rows, const int64_t *d_all, int rank, int world,
                      int64_t *d_absorbed, hipStream_t s) {
    hipLaunchKernelGGL(k_merge_fix, dim3(1), dim3(1), 0, s, rows, d_n_rows, d_all, rank, world, d_absorbed);
}

__global__ void k_bits_extra(const int64_t *all, int rank, int world, int32_t *extra) {
    int head = 0, tail = 0;
    for (int q = rank - 1; q >= 0; --q) { head |= (int)all[3 * q + 2]; if (all[3 * q]) break; }
    for (int q = rank + 1; q < world; ++q) { tail |= (int)all[3 * q + 1]; if (all[3 * q]) break; }
    extra[0] = head; extra[1] = tail;
}
void launch_bits_extra(const int64_t *d_all, int rank, int world, int32_t *d_extra, hipStream_t s) {
    hipLaunchKernelGGL(k_bits_extra, dim3(1), dim3(1), 0, s, d_all, rank, world, d_extra);
}

}  // namespace urh
