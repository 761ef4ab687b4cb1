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
#define URH_SCAN_ITEMS 8
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

template <int K>
__device__ __forceinline__ VecK<K> wave_incl_scan_vec(VecK<K> x, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int64_t u = __shfl_up(x.v[k], o);
            if (lane >= o) x.v[k] += u;
        }
    }
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

// Exclusive scan of the workgroup totals by ONE workgroup; writes the grand total to partials[nblocks_max].
template <int K>
__device__ __forceinline__ void scan_partials_body(VecK<K> *partials, int64_t nb, int64_t nblocks_max, VecK<K> *s_wave, VecK<K> *s_carry) {
    if (threadIdx.x == 0) s_carry->zero();
    __syncthreads();
    for (int64_t b0 = 0; b0 < nb; b0 += kScanBlock) {
        const int64_t b = b0 + threadIdx.x;
        VecK<K> mine; mine.zero();
        if (b < nb) mine = partials[b];
        VecK<K> total;
        VecK<K> ex = block_excl_scan_vec<K>(mine, total, s_wave);
        ex.add(*s_carry);
        if (b < nb) partials[b] = ex;
        __syncthreads();
        if (threadIdx.x == 0) s_carry->add(total);
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[nblocks_max] = *s_carry;
}

// Pass 1: per-workgroup totals; the last workgroup to finish turns them into exclusive prefixes.
// Load::operator()(int64 i) -> VecK<K> (only called for i < n).
template <int K, class Load>
__global__ __launch_bounds__(kScanBlock) void k_scan_reduce(const int64_t *d_n, Load load, VecK<K> *partials, int64_t nblocks_max,
                                                             int32_t *ticket) {
    __shared__ VecK<K> s_wave[kScanBlock / 64];
    __shared__ VecK<K> s_carry;
    const int64_t n = *d_n;
    const int64_t nb = scan_active_blocks(n, nblocks_max);
    if ((int64_t)blockIdx.x >= nb) return;
    const int64_t base = (int64_t)blockIdx.x * kScanTile;
    VecK<K> acc; acc.zero();
    const int64_t i0 = base + (int64_t)threadIdx.x * kScanItems;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j)
        if (i0 + j < n) acc.add(load(i0 + j));
    VecK<K> total;
    block_excl_scan_vec<K>(acc, total, s_wave);
    if (threadIdx.x == 0) partials[blockIdx.x] = total;
    if (!scan_last_block(ticket, nb)) return;
    scan_partials_body<K>(partials, nb, nblocks_max, s_wave, &s_carry);
}

// Pass 2: the scan proper.  Store::operator()(int64 i, const VecK<K>& value, const VecK<K>& excl_prefix);
// Final::operator()(const VecK<K>& grand_total) runs once, on one thread, after every element has been stored.
template <int K, class Load, class Store, class Final>
__global__ __launch_bounds__(kScanBlock) void k_scan_apply(const int64_t *d_n, Load load, const VecK<K> *partials, int64_t nblocks_max,
                                                            Store store, Final fin, int32_t *ticket) {
    __shared__ VecK<K> s_wave[kScanBlock / 64];
    const int64_t n = *d_n;
    const int64_t nb = scan_active_blocks(n, nblocks_max);
    if ((int64_t)blockIdx.x >= nb) return;
    const int64_t base = (int64_t)blockIdx.x * kScanTile;
    VecK<K> item[kScanItems];
    VecK<K> acc; acc.zero();
    const int64_t i0 = base + (int64_t)threadIdx.x * kScanItems;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
        item[j].zero();
        if (i0 + j < n) item[j] = load(i0 + j);
        acc.add(item[j]);
    }
    VecK<K> total;
    VecK<K> ex = block_excl_scan_vec<K>(acc, total, s_wave);
    ex.add(partials[blockIdx.x]);
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
        if (i0 + j < n) store(i0 + j, item[j], ex);
        ex.add(item[j]);
    }
    if (!scan_last_block(ticket, nb)) return;
    if (threadIdx.x == 0) fin(partials[nblocks_max]);
}

struct NoFinal {
    template <class V> __device__ void operator()(const V &) const {}
};

}  // namespace urh
