// Device-wide inclusive prefix sums over K int64 channels at once (reduce / scan-partials / apply).
// Small utility for the pulse-table and bit-expansion stages (arrays of ~1e5..1e7 elements); the
// element count lives in DEVICE memory (`const int64_t* d_n`), grids are sized by a host-side
// upper bound and surplus workgroups exit immediately.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace urh {

constexpr int kScanBlock = 256;
#ifndef URH_SCAN_ITEMS
#define URH_SCAN_ITEMS 4
#endif
constexpr int kScanItems = URH_SCAN_ITEMS;
constexpr int kScanTile = kScanBlock * kScanItems;   // 2048 elements per workgroup

template <int K> struct VecK {
    int64_t v[K];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = 0;
    }
    __device__ __forceinline__ void add(const VecK &o) {
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] += o.v[k];
    }
};

// Inclusive wavefront prefix SUM by DPP (gfx9 row_shr / row_bcast controls): a Kogge-Stone scan inside each row of 16 lanes (row_shr 1, 2,
// 4, 8: a lane whose source lies before its row keeps the 0 it was given), then row 0's / row 2's total into rows 1 / 3 (row_bcast:15, rows
// 0xA) and lane 31 into rows 2 and 3 (row_bcast:31, rows 0xC).  Six steps of plain VALU moves and adds -- __shfl_up goes through the LDS
// crossbar (ds_bpermute_b32: five of them and an lgkmcnt wait per step for the expansion's three channels; the tail's kernels are chains of
// such latencies beside a hot kernel).  Every lane of the wavefront must be active, as for __shfl_up.
// Measured at the round's end (profiles/r06last_dpp_scan_ab.txt: exact -- the whole gpu suite passes on it -- and no faster: 0.471-0.479 ms per
// step either way at 10 samples per symbol, 0.2752-0.2760 against 0.2755-0.2768 on the headline capture): not the default; -DURH_DPP_SCAN=1 builds it.
#ifndef URH_DPP_SCAN
#define URH_DPP_SCAN 0
#endif
template <int CTRL, int ROWS> __device__ __forceinline__ int dpp_or_zero(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, ROWS, 0xF, false); }
template <int CTRL, int ROWS> __device__ __forceinline__ int64_t dpp_or_zero(int64_t x) {
    const uint32_t lo = (uint32_t)dpp_or_zero<CTRL, ROWS>((int)(uint32_t)(uint64_t)x), hi = (uint32_t)dpp_or_zero<CTRL, ROWS>((int)(uint32_t)((uint64_t)x >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
template <class T> __device__ __forceinline__ T wave_incl_sum_dpp(T x) {
    x += dpp_or_zero<0x111, 0xF>(x);                        // row_shr:1
    x += dpp_or_zero<0x112, 0xF>(x);                        // row_shr:2
    x += dpp_or_zero<0x114, 0xF>(x);                        // row_shr:4
    x += dpp_or_zero<0x118, 0xF>(x);                        // row_shr:8
    x += dpp_or_zero<0x142, 0xA>(x);                        // row_bcast:15 into rows 1 and 3
    x += dpp_or_zero<0x143, 0xC>(x);                        // row_bcast:31 into rows 2 and 3
    return x;
}

template <int K>
__device__ __forceinline__ VecK<K> wave_incl_scan_vec(VecK<K> x, int lane) {
#if URH_DPP_SCAN
    (void)lane;
#pragma unroll
    for (int k = 0; k < K; ++k) x.v[k] = wave_incl_sum_dpp(x.v[k]);
#else
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int64_t u = __shfl_up(x.v[k], o);
            if (lane >= o) x.v[k] += u;
        }
    }
#endif
    return x;
}

// Workgroup-wide exclusive scan of one VecK per thread; returns the exclusive prefix of the
// calling thread and the workgroup total in `total`.
template <int K>
__device__ __forceinline__ VecK<K> block_excl_scan_vec(const VecK<K> &mine, VecK<K> &total, VecK<K> *s_wave /*[kScanBlock/64]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = kScanBlock / 64;
    VecK<K> incl = wave_incl_scan_vec<K>(mine, lane);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    VecK<K> base; base.zero(); total.zero();
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        if (w < wave) base.add(s_wave[w]);
        total.add(s_wave[w]);
    }
    __syncthreads();
    VecK<K> ex = base;
#pragma unroll
    for (int k = 0; k < K; ++k) ex.v[k] += incl.v[k] - mine.v[k];
    return ex;
}

// Participating workgroups of a pass over n elements: the ones that own elements, and always workgroup 0 (so
// that the "last workgroup" epilogues below run even for n == 0).
__device__ __forceinline__ int64_t scan_active_blocks(int64_t n, int64_t nblocks_max) {
    int64_t nb = (n + kScanTile - 1) / kScanTile;
    if (nb > nblocks_max) nb = nblocks_max;
    return nb < 1 ? 1 : nb;
}

// Launch grid for a scan whose element count is usually far below the host-side bound (pulse tables: rows <= samples / (tol + 1),
// in practice a few percent of that): at most kScanGridMax workgroups, each looping over the active tiles with stride gridDim.x.
// Surplus workgroups still cost a dispatch each -- beside a running hot kernel (pipelined passes) thousands of them wait for wave
// slots one retiring workgroup at a time: the group scan "ran" for 60-240 us there.
constexpr int64_t kScanGridMax = 512;
inline unsigned scan_grid(int64_t nblocks_max) { return (unsigned)(nblocks_max < kScanGridMax ? (nblocks_max < 1 ? 1 : nblocks_max) : kScanGridMax); }

// "Last workgroup done" election: every participating workgroup calls this after its global writes; exactly one
// call (the last to arrive) returns true, with all the other workgroups' writes visible.  *ticket must be 0 before
// the launch and is 0 again afterwards.
__device__ __forceinline__ bool scan_last_block(int32_t *ticket, int64_t nb) {
    __shared__ int s_is_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const int t = atomicAdd(ticket, 1);
        s_is_last = (t == (int)nb - 1);
        if (s_is_last) *ticket = 0;
    }
    __syncthreads();
    const bool last = s_is_last != 0;
    if (last) __threadfence();
    return last;
}

// Three small kernels make a device-wide scan; none of them elects a "last workgroup" (an election costs every workgroup a
// device-scope fence and an atomic on ONE address: 131 072 workgroups compacting a 1 GiB capture spent 10 ms in it):
//   k_scan_reduce    per-workgroup totals -> partials[b]
//   k_scan_partials  ONE workgroup, only when more than `direct_max` workgroups are active: partials -> exclusive prefixes,
//                    grand total -> partials[nblocks_max]
//   k_scan_apply     the scan proper; with at most `direct_max` active workgroups every workgroup adds up the totals of the
//                    workgroups before it itself (the pulse-table stages: a few hundred workgroups, direct_max = "always")
//   k_scan_finish    ONE workgroup: grand total, Final
constexpr int64_t kScanAlwaysDirect = INT64_MAX;
constexpr int64_t kScanDirect = 1024;      // stream compaction over a whole capture: beyond this many workgroups use k_scan_partials
constexpr int kScanPartialsBlock = 1024;

// Sum of partials[0 .. count) by the whole workgroup (every thread gets it).
template <int K>
__device__ __forceinline__ VecK<K> block_sum_partials(const VecK<K> *partials, int64_t count, VecK<K> *s_wave) {
    VecK<K> acc; acc.zero();
    for (int64_t u = threadIdx.x; u < count; u += kScanBlock) acc.add(partials[u]);
    VecK<K> total;
    block_excl_scan_vec<K>(acc, total, s_wave);
    return total;
}

// Optional Load::on_start(n): called once (workgroup 0, thread 0) before anything else, whatever n is.
template <class Load> __device__ __forceinline__ auto scan_on_start(const Load &l, int64_t n, int) -> decltype(l.on_start(n), void()) { l.on_start(n); }
template <class Load> __device__ __forceinline__ void scan_on_start(const Load &, int64_t, long) {}

// Load::operator()(int64 i) -> VecK<K> (only called for i < n).
template <int K, class Load>
__global__ __launch_bounds__(kScanBlock) void k_scan_reduce(const int64_t *d_n, Load load, VecK<K> *partials, int64_t nblocks_max) {
    __shared__ VecK<K> s_wave[kScanBlock / 64];
    const int64_t n = *d_n;
    if (blockIdx.x == 0 && threadIdx.x == 0) scan_on_start(load, n, 0);
    const int64_t nb = scan_active_blocks(n, nblocks_max);
    for (int64_t b = blockIdx.x; b < nb; b += gridDim.x) {             // the grid may be smaller than nblocks_max (scan_grid)
        const int64_t base = b * kScanTile;
        VecK<K> acc; acc.zero();
        const int64_t i0 = base + (int64_t)threadIdx.x * kScanItems;
#pragma unroll
        for (int j = 0; j < kScanItems; ++j)
            if (i0 + j < n) acc.add(load(i0 + j));
        VecK<K> total;
        block_excl_scan_vec<K>(acc, total, s_wave);
        if (threadIdx.x == 0) partials[b] = total;
    }
}

// ONE workgroup of kScanPartialsBlock threads, 8 totals per thread and round.
template <int K>
__global__ __launch_bounds__(kScanPartialsBlock) void k_scan_partials(const int64_t *d_n, VecK<K> *partials, int64_t nblocks_max, int64_t direct_max) {
    constexpr int kPer = 8, kW = kScanPartialsBlock / 64;
    __shared__ VecK<K> s_w[kW];
    __shared__ VecK<K> s_carry;
    const int64_t nb = scan_active_blocks(*d_n, nblocks_max);
    if (nb <= direct_max) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry.zero();
    __syncthreads();
    for (int64_t b0 = 0; b0 < nb; b0 += (int64_t)kScanPartialsBlock * kPer) {
        const int64_t i0 = b0 + (int64_t)threadIdx.x * kPer;
        VecK<K> item[kPer], acc; acc.zero();
#pragma unroll
        for (int j = 0; j < kPer; ++j) { item[j].zero(); if (i0 + j < nb) item[j] = partials[i0 + j]; acc.add(item[j]); }
        VecK<K> incl = wave_incl_scan_vec<K>(acc, lane);
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        VecK<K> ex = s_carry, total; total.zero();
        for (int w = 0; w < kW; ++w) { if (w < wave) ex.add(s_w[w]); total.add(s_w[w]); }
#pragma unroll
        for (int k = 0; k < K; ++k) ex.v[k] += incl.v[k] - acc.v[k];
#pragma unroll
        for (int j = 0; j < kPer; ++j) { if (i0 + j < nb) partials[i0 + j] = ex; ex.add(item[j]); }
        __syncthreads();
        if (threadIdx.x == 0) s_carry.add(total);
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[nblocks_max] = s_carry;
}

// Store::operator()(int64 i, const VecK<K>& value, const VecK<K>& excl_prefix)
template <int K, class Load, class Store>
__global__ __launch_bounds__(kScanBlock) void k_scan_apply(const int64_t *d_n, Load load, const VecK<K> *partials, int64_t nblocks_max,
                                                            Store store, int64_t direct_max) {
    __shared__ VecK<K> s_wave[kScanBlock / 64];
    const int64_t n = *d_n;
    const int64_t nb = scan_active_blocks(n, nblocks_max);
    for (int64_t b = blockIdx.x; b < nb; b += gridDim.x) {
        const int64_t base = b * kScanTile;
        VecK<K> item[kScanItems];
        VecK<K> acc; acc.zero();
        const int64_t i0 = base + (int64_t)threadIdx.x * kScanItems;
#pragma unroll
        for (int j = 0; j < kScanItems; ++j) {
            item[j].zero();
            if (i0 + j < n) item[j] = load(i0 + j);
            acc.add(item[j]);
        }
        VecK<K> before;
        if (nb <= direct_max) before = block_sum_partials<K>(partials, b, s_wave);
        else before = partials[b];
        VecK<K> total;
        VecK<K> ex = block_excl_scan_vec<K>(acc, total, s_wave);
        ex.add(before);
#pragma unroll
        for (int j = 0; j < kScanItems; ++j) {
            if (i0 + j < n) store(i0 + j, item[j], ex);
            ex.add(item[j]);
        }
    }
}

// Final::operator()(const VecK<K>& grand_total) runs once, on one thread, after every element has been stored; the grand
// total is also left in partials[nblocks_max].  Launch with ONE workgroup right after k_scan_apply.
template <int K, class Final>
__global__ __launch_bounds__(kScanBlock) void k_scan_finish(const int64_t *d_n, VecK<K> *partials, int64_t nblocks_max, Final fin,
                                                             int64_t direct_max) {
    __shared__ VecK<K> s_wave[kScanBlock / 64];
    const int64_t nb = scan_active_blocks(*d_n, nblocks_max);
    VecK<K> grand;
    if (nb <= direct_max) grand = block_sum_partials<K>(partials, nb, s_wave);
    else grand = partials[nblocks_max];
    if (threadIdx.x == 0) { partials[nblocks_max] = grand; fin(grand); }
}

// ---- single-pass scan (decoupled look-back) -----------------------------------------------------------------------------
// One launch instead of two: every workgroup publishes the total of its tile (AGGREGATE), then a wavefront looks back over
// its predecessors -- 64 descriptors per step -- adding aggregates until it meets one that already carries its inclusive
// prefix, publishes its own inclusive prefix (PREFIX) and finishes its tile.  Workgroups are dispatched in index order, so
// a workgroup only ever waits for workgroups that are already running.  Descriptors are never reset: the flag carries the
// pass number (`epoch`, a per-context counter), anything older reads as "not yet published".
template <int K> struct ScanDesc {
    VecK<K> agg;
    VecK<K> incl;
    unsigned long long flag;     // epoch << 2 | state (1 = aggregate valid, 2 = inclusive prefix valid)
    unsigned long long pad;
};
constexpr unsigned long long kScanAgg = 1, kScanIncl = 2;
constexpr int kScanSpinLimit = 1 << 24;    // a lost predecessor would otherwise hang the GPU: give up and poison the result

template <int K>
__device__ __forceinline__ VecK<K> wave_sum_vec(VecK<K> x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int k = 0; k < K; ++k) x.v[k] += __shfl_xor(x.v[k], o);
    return x;
}

// exclusive prefix of workgroup b (the sum of the totals of workgroups 0 .. b-1); executed by wavefront 0 of the workgroup
template <int K>
__device__ __forceinline__ VecK<K> scan_look_back(ScanDesc<K> *desc, int64_t b, unsigned long long epoch) {
    const int lane = threadIdx.x & 63;
    VecK<K> prefix; prefix.zero();
    int64_t top = b - 1;                                  // nearest predecessor not yet accounted for
    int spins = 0;
    while (top >= 0) {
        const int64_t j = top - lane;
        unsigned long long f = 0;
        if (j >= 0) f = __hip_atomic_load(&desc[j].flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool ready = (j < 0) || ((f >> 2) == epoch);
        const bool has_incl = (j >= 0) && ready && ((f & 3) == kScanIncl);
        const unsigned long long m_incl = __ballot(has_incl);
        const unsigned long long m_not_ready = __ballot(!ready);
        // lanes below the first inclusive prefix must all be published
        const int first_incl = m_incl ? __builtin_ctzll(m_incl) : 64;
        const unsigned long long need = (first_incl >= 63) ? ~0ull : ((2ull << first_incl) - 1ull);
        if (m_not_ready & need) {
            if (++spins > kScanSpinLimit) { prefix.v[0] = -(1ll << 60); return prefix; }
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        __threadfence();                                  // acquire: the values behind the flags
        VecK<K> mine; mine.zero();
        if (j >= 0 && lane <= first_incl) mine = (lane == first_incl) ? desc[j].incl : desc[j].agg;
        prefix.add(wave_sum_vec<K>(mine));
        if (m_incl) break;
        top -= 64;
    }
    return prefix;
}

// Load::operator()(int64 i) -> VecK<K>; Store::operator()(int64 i, const VecK<K>& value, const VecK<K>& excl_prefix);
// Final::operator()(const VecK<K>& grand_total) runs once on one thread: by the last workgroup of the scan when it does not
// depend on what the OTHER workgroups stored (elect == 0), else by the workgroup that finishes last (ticket election).
// ITEMS: elements per thread (tile = kScanBlock * ITEMS elements).  Scans over a handful of elements whose Load / Store are heavy (the
// group scan: 105 VGPRs at four items) take one: the kernel then fits beside the hot kernel of the next pass (see scan_grid).
// At least six waves per SIMD = at most 80 VGPRs: what one retiring hot workgroup (72 + the 8 it never had) leaves free on a SIMD.
template <int K, class Load, class Store, class Final, int ITEMS = kScanItems>
__global__ __launch_bounds__(kScanBlock) __attribute__((amdgpu_waves_per_eu(6))) void k_scan_lookback(const int64_t *d_n, Load load, ScanDesc<K> *desc, int64_t nblocks_max,
                                                               Store store, Final fin, unsigned long long epoch, int32_t *ticket,
                                                               int elect) {
    URH_TAIL_PRIO();
    __shared__ VecK<K> s_wave[kScanBlock / 64];
    __shared__ VecK<K> s_prefix;
    const int64_t n = *d_n;
    constexpr int64_t kTileElems = (int64_t)kScanBlock * ITEMS;
    int64_t nb = (n + kTileElems - 1) / kTileElems;          // scan_active_blocks for this tile size
    if (nb > nblocks_max) nb = nblocks_max;
    if (nb < 1) nb = 1;
    // The grid may be smaller than nblocks_max (scan_grid): workgroup w then takes tiles w, w + G, ...  A tile still only waits
    // for tiles below it, and those belong to workgroups that are running or will be dispatched whatever this one does.
    for (int64_t b = blockIdx.x; b < nb; b += gridDim.x) {
        const int64_t base = b * kTileElems;
        VecK<K> item[ITEMS];
        VecK<K> acc; acc.zero();
        const int64_t i0 = base + (int64_t)threadIdx.x * ITEMS;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            item[j].zero();
            if (i0 + j < n) item[j] = load(i0 + j);
            acc.add(item[j]);
        }
        VecK<K> total;
        VecK<K> ex = block_excl_scan_vec<K>(acc, total, s_wave);
        if (threadIdx.x < 64) {                               // wavefront 0 publishes and looks back
            if (b > 0 && threadIdx.x == 0) {
                desc[b].agg = total;
                __threadfence();
                __hip_atomic_store(&desc[b].flag, (epoch << 2) | kScanAgg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const VecK<K> prefix = scan_look_back<K>(desc, b, epoch);
            if (threadIdx.x == 0) {
                VecK<K> incl = prefix; incl.add(total);
                desc[b].incl = incl;
                __threadfence();
                __hip_atomic_store(&desc[b].flag, (epoch << 2) | kScanIncl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_prefix = prefix;
            }
        }
        __syncthreads();
        const VecK<K> prefix = s_prefix;
        ex.add(prefix);
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            if (i0 + j < n) store(i0 + j, item[j], ex);
            ex.add(item[j]);
        }
        if (elect) {
            if (scan_last_block(ticket, nb) && threadIdx.x == 0) fin(desc[nb - 1].incl);
        } else if (b == nb - 1 && threadIdx.x == 0) {
            VecK<K> grand = prefix; grand.add(total);
            fin(grand);
        }
        __syncthreads();                                      // s_prefix / s_wave are reused by the next tile
    }
}


// ---- look-back on data-tagged granules ----------------------------------------------------------------------------------
// The descriptor of a workgroup is 2 x 8 granules of 8 bytes {32 bits of payload, 32-bit pass tag}, each written by ONE
// write-through (agent-scope relaxed atomic = sc1) store and polled with sc1 loads: a granule whose tag is the current pass's
// carries valid payload -- no flag, no fence, no ordering between the granules (MI355X_MICROARCH.md, persistent-kernel price
// list: a fenced flag hand-off costs 2 x 3.5 us of __threadfence, a granule hand-off about 1 us).  Memory must be zero
// (or hold older tags) before first use; tags never repeat on a context within 2^32 passes.
struct GranDesc { unsigned long long agg[8]; unsigned long long incl[8]; };
static_assert(sizeof(GranDesc) == 128, "GranDesc");

template <class T> __device__ __forceinline__ void gran_store(unsigned long long *g, const T &x, uint32_t tag) {
    static_assert(sizeof(T) == 32, "granule payload is 8 x 32 bits");
    uint32_t w[8];
    __builtin_memcpy(w, &x, 32);
#pragma unroll
    for (int k = 0; k < 8; ++k)
        __hip_atomic_store(g + k, ((unsigned long long)tag << 32) | w[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class T> __device__ __forceinline__ bool gran_load(const unsigned long long *g, T &x, uint32_t tag) {
    uint32_t w[8];
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const unsigned long long v = __hip_atomic_load(g + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        w[k] = (uint32_t)v;
        ok &= (uint32_t)(v >> 32) == tag;
    }
    __builtin_memcpy(&x, w, 32);
    return ok;
}

// Ordered composition of the totals of workgroups 0 .. b-1, executed by ONE wavefront (every lane gets the result).
// comb(left, right) may be non-commutative; shfl_down(x, o) moves a T across lanes.  Lane l of a window looks at workgroup
// top - l; a descending-lane suffix composition keeps the workgroups in order.  `ok` is cleared when a predecessor never shows up.
template <class T, class Comb, class ShflDown, class Shfl0>
__device__ __forceinline__ T gran_look_back(GranDesc *desc, int64_t b, uint32_t tag, const T &identity, Comb comb, ShflDown shfl_down,
                                            Shfl0 shfl0, int lane, bool &ok) {
    T carry = identity;
    int64_t top = b - 1;
    int spins = 0;
    ok = true;
    while (top >= 0) {
        const int64_t j = top - lane;
        T vi = identity, va = identity;
        bool has_incl = false, has_agg = false;
        if (j >= 0) {
            has_incl = gran_load(desc[j].incl, vi, tag);
            if (!has_incl) has_agg = gran_load(desc[j].agg, va, tag);
        }
        const bool ready = (j < 0) || has_incl || has_agg;
        const unsigned long long m_incl = __ballot(has_incl);
        const unsigned long long m_not_ready = __ballot(!ready);
        const int first_incl = m_incl ? __builtin_ctzll(m_incl) : 64;
        const unsigned long long need = (first_incl >= 63) ? ~0ull : ((2ull << first_incl) - 1ull);
        if (m_not_ready & need) {
            if (++spins > kScanSpinLimit) { ok = false; break; }
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        T mine = identity;
        if (j >= 0 && lane <= first_incl) mine = (lane == first_incl) ? vi : va;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const T u = shfl_down(mine, o);                   // workgroups further to the left
            if (lane + o < 64) mine = comb(u, mine);
        }
        carry = comb(shfl0(mine), carry);
        if (m_incl) break;
        top -= 64;
    }
    return carry;
}

struct NoFinal {
    template <class V> __device__ void operator()(const V &) const {}
};

}  // namespace urh
