// msg_estimators.hip -- AutoInterpretation.estimate's per-message statistics for ALL messages of a capture at once.
//
// The reference (src/urh/ainterpretation/AutoInterpretation.py:373-471) walks the messages one by one: detect_center
// (:226-277: noise removal, 5 % trim, min / max, np.var, np.histogram) and get_plateau_lengths
// (cythonext/auto_interpretation.pyx:179-208) per message.  Doing the same on the GPU message by message costs a dozen
// launches and several read-backs per message (seconds for a hundred messages).  Here every stage is ONE launch over a tile
// table that covers all messages, results stay in device memory between the stages, and the host reads back twice per capture:
//   urhgpu_msg_center_stats   kept count, trimmed length, min, max, mean, variance and the histogram of every message
//   urhgpu_msg_plateaus       plateau lengths of every message for the centers the host picked from those histograms
// The arithmetic is the reference's: float32 sums in numpy's pairwise order (see estimators.hip), float64 bin edges
// first + i * delta as np.arange fills them, np.histogram's half-open bins with the last one closed.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <math.h>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <unistd.h>
#include <vector>

#include "common.hpp"
#include "launchers.hpp"

namespace urh {

constexpr int kMeTile = 4096;                 // samples per tile: 256 threads x 16 consecutive samples
constexpr int kMeBlock = 256;
constexpr int kMePer = kMeTile / kMeBlock;
constexpr int kPwChunkM = 8192, kPwLeafM = 128, kLeavesPerTile = kMeTile / kPwLeafM;    // numpy's pairwise summation geometry

struct MsgTile { int32_t msg; int32_t idx; };           // tile `idx` of message `msg`

struct MsgState {            // one per message, device resident between the stages (host mirror: same layout)
    int64_t start, end;      // sample range in the demodulated signal
    int64_t first_tile;      // index of its first tile in the tile table
    int64_t kept;            // samples > -4 (AutoInterpretation.py:227)
    int64_t a, L;            // trimmed range [a, a + L) of the kept samples (:231)
    float mn, mx;            // util.minmax of the trimmed samples (:236)
    float mean, var;         // float32 np.mean / np.var (:240)
    double e0, delta;        // bin edges e0 + i * delta (np.arange(min, max + step, step), :243)
    int64_t n_edges;         // 0: no histogram (empty, zero / NaN variance, fewer than 2 edges); > max_bins + 1: too many for the pool
    int64_t n_plateaus;      // urhgpu_msg_plateaus: plateaus found, -1: the window did not reach the 25 % mark
    int64_t edge_cnt;        // boundaries found in the window
    int64_t window;          // samples of the message searched for boundaries
    double center;           // center handed to urhgpu_msg_plateaus (NaN: none)
    double peak_center;      // urhgpu_msg_center_stats: center picked from the histogram (k_me_peaks)
    int64_t peak_flag;       // 0 none, 1 peak_center valid, 2 more bins than the pool holds, 3 equally populated peaks compete for a slot
    int64_t skip;            // 1: the message's first sample is filtered (x <= -4: afp_demod's result[0] = NOISE of FSK / PSK) -- k_me_first
    int64_t pairs_base;      // urhgpu_msg_plateau_decisions: first (value, count) pair of the message's plateau lengths in the pool ...
    int64_t pairs_n;         // ... and how many; -1: more distinct lengths than the table holds / the pool is full (the sequence decides)
};
// nothing filtered but (possibly) the first sample: the kept samples ARE x[start + skip : end], no compaction needed
__host__ __device__ inline bool me_clean(const MsgState &m) { return m.kept == (m.end - m.start) - m.skip; }

// Tile geometry: thread `tid` of a tile looks at samples j * 256 + tid (j = 0..15): every load instruction of a wavefront reads 256
// consecutive bytes.  Order-preserving work (the compaction, the boundary lists) ranks a sample by (row j, wavefront, lane) from
// ballots.  (16 consecutive samples per thread -- the first version -- made every load touch 32 cache lines: 1.9 TB/s.)
__device__ __forceinline__ unsigned long long me_lanes_below() { return (1ull << (threadIdx.x & 63)) - 1ull; }
// exclusive prefix of the 64 cells (row j, wavefront w) of a tile in s_cell[j * 4 + w] (in place), total returned to all threads
__device__ __forceinline__ int me_cell_scan(int *s_cell, int *s_total) {
    __syncthreads();
    if (threadIdx.x < 64) {
        const int v = s_cell[threadIdx.x];
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o); if ((int)threadIdx.x >= o) incl += u; }
        s_cell[threadIdx.x] = incl - v;
        if (threadIdx.x == 63) *s_total = incl;
    }
    __syncthreads();
    return *s_total;
}
// samples of a message after the x > -4 filter: the capture itself when nothing was filtered (ASK magnitudes: always)
__device__ __forceinline__ const float *me_src(const float *x, const float *kept, const MsgState &m) {
    return me_clean(m) ? x + m.start + m.skip : kept + m.start;
}

// A tile's 32 leaf sums (in lane 0 of every group of 8 lanes, leaf 8 w + g in group g of wavefront w) are half of a chunk's perfect
// binary tree (s[i] = s[2 i] + s[2 i + 1], level by level): three levels inside the wavefront; its lane 0 leaves the wavefront's
// subtree in s_h[w], and behind a barrier the tile's half is (s_h[0] + s_h[1]) + (s_h[2] + s_h[3]).  The chunk's sum is half 2 c + half
// 2 c + 1 (k_me_sum_fin).  (Rounds 3-5: the leaf sums went to device memory and a kernel of its own built the trees, k_me_chunk_trees.)
__device__ __forceinline__ float me_half_tree(float leaf_sum, float *s_h) {
    float h = leaf_sum;
    h = h + __shfl_down(h, 8);
    h = h + __shfl_down(h, 16);
    h = h + __shfl_down(h, 32);
    if ((threadIdx.x & 63) == 0) s_h[threadIdx.x >> 6] = h;
    return h;
}

// ---- stage 1: count x > -4 per natural tile, stable compaction per message into kept[start + j] -----------------------------
// ---- stage 1, speculative form: ONE pass that counts AND -- assuming nothing will be filtered (ASK magnitudes never are; an FSK / PSK
// message without a single noise sample is not either) -- takes min / max and the first-round leaf sums (np.mean) of the trimmed range.
// With k = len the trim is known up front (a = int(0.05 len), L = int(0.95 len) - a), so tile t reads the window [a + 4096 t, a + 4096
// (t + 1)) of its message in the leaf geometry of k_me_leaves (8 threads per 128-element leaf) and counts x > -4 there; the head [0, a)
// of the message is counted in slices, one per tile.  When the count confirms k == len the first leaf
// pass is skipped (k_me_leaves mode 0 returns at once); otherwise the general path runs on the counts this pass has left per natural
// tile (compaction, then the leaf passes).  Three passes over a clean message instead of five; five reads and one write instead of
// six and one over a message with filtered samples.
__global__ __launch_bounds__(kMeBlock) void k_me_first(const float *x, MsgState *st, const MsgTile *tiles, int32_t *tile_cnt, float2 *tile_mm,
                                                        float *half_sums) {
    __shared__ int s_c[4 * (kMeBlock / 64)];
    __shared__ float s_mn[kMeBlock / 64], s_mx[kMeBlock / 64], s_h[kMeBlock / 64];
    const MsgTile t = tiles[blockIdx.x];
    const MsgState m = st[t.msg];
    // The demodulated signal of an FSK / PSK capture STARTS with a filtered sample (result[0] = NOISE = -4, signal_functions.pyx:361): the
    // speculation is "nothing is filtered but, possibly, the first sample" -- the kept samples are then x[start + skip : end], contiguous.
    const int64_t skip = (m.end > m.start && !(x[m.start] > -4.0f)) ? 1 : 0;
    if (t.idx == 0 && threadIdx.x == 0) st[t.msg].skip = skip;
    const int64_t len = m.end - m.start - skip;
    const int64_t a = (int64_t)(0.05 * (double)len), b = (int64_t)(0.95 * (double)len);       // k_me_trim's arithmetic with k = len
    const int64_t L = b > a ? b - a : 0;
    const int64_t full = m.end - m.start;
    const int64_t nt = ((full > 1 ? full : 1) + kMeTile - 1) / kMeTile;       // tiles of the message (build_batch)
    const float *src = x + m.start + skip;
    const int lf = threadIdx.x >> 3, j = threadIdx.x & 7;
    // Everything per element is 32-bit arithmetic relative to wave-uniform 64-bit bases (the first version indexed every element in 64
    // bits: 35 VALU instructions per 64 samples, 116 us of instruction issue for a pass whose bytes take 95 -- profiles/r03d_estimator_pmc.txt)
    const int64_t wbase = a + (int64_t)t.idx * kMeTile;                      // first sample of this tile's window, in src
    const float *win = src + wbase;
    const int rel0 = lf * kPwLeafM + j;                                      // this thread's first element of the window
    const int64_t left_len = len - wbase, left_L = L - (int64_t)t.idx * kMeTile;
    const int lim_len = (int)(left_len < 0 ? 0 : (left_len > kMeTile ? kMeTile : left_len));   // samples of the window that exist ...
    const int lim_L = (int)(left_L < 0 ? 0 : (left_L > kMeTile ? kMeTile : left_L));           // ... and that lie in the trimmed range
    // The counts are kept per NATURAL tile of the message (samples [4096 u, 4096 (u + 1)) from its start) although the reads are not
    // aligned to them: the compaction of a message that does have filtered samples needs those, and this way nobody reads the message
    // a second time just to count.  A window meets two natural tiles, a head slice (at most 5 % of 4096 samples) two as well.
    static_assert((kMeTile & (kMeTile - 1)) == 0, "natural tiles by shift and mask");
    const int64_t wa = (skip + wbase) / kMeTile;                             // natural tile of the window's first sample
    const int woff = (int)((skip + wbase) & (kMeTile - 1));                  // ... and where in it the window starts
    const bool any = L > 0 && (int64_t)t.idx * kMeTile < L;
    const float first = L > 0 ? src[a] : 0.f;
    float v[kPwLeafM / 8];
    int cw[4] = {0, 0, 0, 0};                              // per WAVEFRONT: window in tiles wa, wa + 1; head slice in tiles ha, ha + 1
    float mn = first, mx = first;                          // min / max over the elements inside the trimmed range, seeded with its first element (util.minmax: a NaN there stays)
    if (lim_len == kMeTile && lim_L == kMeTile) {
        // Round 6 (late): a window that lies wholly inside the message and the trimmed range -- all but two or three per message.  Loads
        // without a bound, the count of a row of 64 samples from the compare's own lane mask (scalar popcounts), no range test in front
        // of min / max: 7 VALU instructions per sample instead of 31 -- the pass was bound by them (PMC: VALU busy for 113 of its 106 us
        // of wall time on the slowest SIMD; profiles/r06q_first_pmc.txt), now by its bytes.
#pragma unroll
        for (int i = 0; i < kPwLeafM / 8; ++i) v[i] = win[rel0 + 8 * i];
        const int thr = kMeTile - woff;                    // window positions from thr on lie in the second natural tile
#pragma unroll
        for (int i = 0; i < kPwLeafM / 8; ++i) {
            const unsigned long long hit = __ballot(v[i] > -4.0f), sec = __ballot(rel0 + 8 * i >= thr);
            cw[0] += __popcll(hit & ~sec); cw[1] += __popcll(hit & sec);
            if (v[i] < mn) mn = v[i];
            if (v[i] > mx) mx = v[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < kPwLeafM / 8; ++i) v[i] = (rel0 + 8 * i < lim_len) ? win[rel0 + 8 * i] : -5.0f;
        int c0 = 0, c1 = 0;
#pragma unroll
        for (int i = 0; i < kPwLeafM / 8; ++i) {
            const int hit = (v[i] > -4.0f) ? 1 : 0;
            const bool second = woff + rel0 + 8 * i >= kMeTile;
            c0 += second ? 0 : hit; c1 += second ? hit : 0;
            const float w = (rel0 + 8 * i < lim_L) ? v[i] : first;
            if (w < mn) mn = w;
            if (w > mx) mx = w;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { c0 += __shfl_xor(c0, o); c1 += __shfl_xor(c1, o); }
        cw[0] = c0; cw[1] = c1;
    }
    int64_t ha = 0;
    if (a > 0) {                                           // the head [0, a): slice t of nt
        const int64_t h = (a + nt - 1) / nt, h0 = (int64_t)t.idx * h, h1 = (h0 + h < a) ? h0 + h : a;
        ha = (skip + h0) / kMeTile;
        const int hoff = (int)((skip + h0) & (kMeTile - 1));
        const int hn = (int)(h1 > h0 ? h1 - h0 : 0);        // (a slice holds at most a / nt + 1 <= 0.05 * 4096 + 1 samples)
        const float *head = src + h0;
        int c2 = 0, c3 = 0;
        for (int k = threadIdx.x; k < hn; k += kMeBlock) {
            const int hit = (head[k] > -4.0f) ? 1 : 0;
            const bool second = hoff + k >= kMeTile;
            c2 += second ? 0 : hit; c3 += second ? hit : 0;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { c2 += __shfl_xor(c2, o); c3 += __shfl_xor(c3, o); }
        cw[2] = c2; cw[3] = c3;
    }
    // leaf sums of the FULL chunks (k_me_leaves, mode 0): accumulator j of the leaf, then ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7))
    const int64_t n_full_leaves = (L / kPwChunkM) * (kPwChunkM / kPwLeafM);
    float acc = v[0];
#pragma unroll
    for (int i = 1; i < kPwLeafM / 8; ++i) acc += v[i];
    acc = acc + __shfl_down(acc, 1);
    acc = acc + __shfl_down(acc, 2);
    acc = acc + __shfl_down(acc, 4);
    me_half_tree(acc, s_h);                                   // (s_h is read behind the barrier below)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float u = __shfl_xor(mn, o), w = __shfl_xor(mx, o);
        if (u < mn) mn = u;
        if (w > mx) mx = w;
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) s_c[(threadIdx.x >> 6) * 4 + k] = cw[k];
        s_mn[threadIdx.x >> 6] = mn; s_mx[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x < 4) {                                 // thread k: counter k of the workgroup -> its natural tile (tile_cnt is zeroed before the launch)
        const int k = threadIdx.x;
        const int total = s_c[k] + s_c[4 + k] + s_c[8 + k] + s_c[12 + k];
        const int64_t u = (k < 2 ? wa : ha) + (k & 1);
        if (total > 0 && u < nt) atomicAdd(&tile_cnt[m.first_tile + u], total);
    }
    if (threadIdx.x == 0) {
        for (int w = 1; w < kMeBlock / 64; ++w) { if (s_mn[w] < mn) mn = s_mn[w]; if (s_mx[w] > mx) mx = s_mx[w]; }
        if (any) tile_mm[blockIdx.x] = float2{mn, mx};
        if ((int64_t)t.idx * kLeavesPerTile < n_full_leaves) half_sums[m.first_tile + t.idx] = (s_h[0] + s_h[1]) + (s_h[2] + s_h[3]);
    }
}
// exclusive prefix of the per-tile counts over the whole tile table (ONE workgroup: the table has n / 4096 entries); a tile's
// offset inside its message is pre[tile] - pre[message's first tile] -- no workgroup re-adds the counts of the tiles before it
// (a capture that is ONE message of 2^27 samples has 32 768 tiles)
constexpr int kMeScanBlock = 1024;
// st != nullptr (the center statistics): the kept count and the trimmed range [a, a + L) of every message are this kernel's last step --
// they need nothing but the prefix (rounds 3-5: two launches of their own, k_me_spec and k_me_trim)
// *any_dirty (cleared by the caller): some message has filtered samples -- the compaction and its trees have work to do
__global__ __launch_bounds__(kMeScanBlock) void k_me_tile_scan(const int32_t *cnt, int64_t n_tiles, int64_t *pre, MsgState *st, int n_msgs,
                                                                unsigned int *any_dirty) {
    // every wavefront owns a contiguous stretch of the table and reads it 64 entries at a time (coalesced): first its sum, then --
    // after ONE exchange of the 16 sums -- a running wavefront scan over the stretch.  (The first version walked the whole table
    // 1024 entries at a time: 32 rounds of three barriers each for a 1 GiB capture, 31 us.)
    __shared__ int64_t s_w[kMeScanBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int kW = kMeScanBlock / 64;
    const int64_t per = ((n_tiles + kW - 1) / kW + 63) / 64 * 64;
    const int64_t lo = (int64_t)wave * per, hi = (lo + per < n_tiles) ? lo + per : n_tiles;
    int64_t mine = 0;
#pragma unroll 4
    for (int64_t i = lo + lane; i < hi; i += 64) mine += cnt[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if (lane == 0) s_w[wave] = mine;
    __syncthreads();
    int64_t run = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kW; ++w) { if (w < wave) run += s_w[w]; total += s_w[w]; }
    for (int64_t i0 = lo; i0 < hi; i0 += 4 * 64) {           // four loads in flight per round
        int v[4];                                             // (a tile holds at most 4096 samples: 64 of them sum in 32 bits)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int64_t i = i0 + j * 64 + lane; v[j] = (i < hi) ? cnt[i] : 0; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t i = i0 + j * 64 + lane;
            int incl = v[j];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o); if (lane >= o) incl += u; }
            if (i < hi) pre[i] = run + (int64_t)(incl - v[j]);
            run += (int64_t)__shfl(incl, 63);
        }
    }
    if (threadIdx.x == 0) pre[n_tiles] = total;
    if (!st) return;
    __syncthreads();                                          // (the prefix is visible to the whole workgroup)
    for (int m = threadIdx.x; m < n_msgs; m += kMeScanBlock) {
        const int64_t len = st[m].end - st[m].start, nt = ((len > 1 ? len : 1) + kMeTile - 1) / kMeTile;
        const int64_t k = pre[st[m].first_tile + nt] - pre[st[m].first_tile];
        st[m].kept = k;
        const int64_t a = (int64_t)(0.05 * (double)k), b = (int64_t)(0.95 * (double)k);     // int(0.05 * len), int(0.95 * len) (:231)
        st[m].a = a;
        st[m].L = b > a ? b - a : 0;
        if (k != len - st[m].skip) *any_dirty = 1u;
    }
}

// Round 6 (late): the compaction also delivers what the first leaf pass (np.mean) and the min / max pass read the compacted samples
// for.  A tile's kept samples are consecutive ranks [before, before + cnt) of the message; the trimmed range starts at rank a, leaf l covers
// the ranks a + 128 l .. a + 128 l + 127.  The leaves that lie WHOLLY inside the tile's ranks are summed from LDS, where the tile stages its
// kept samples by rank (a leaf per row of 136 floats: the 8 leaves a wavefront sums at a time hit 64 different banks) -- in k_me_leaves'
// order, 8 threads per leaf.  The one leaf that starts in the tile and ends in a later one is marked kLeafPending and summed from the
// compacted samples by k_me_dirty_trees, which turns the leaf sums into the half-chunk sums k_me_sum_fin reads.  (A kept sample is never
// a NaN -- the filter is x > -4 -- so no sum of samples carries that NaN's payload.)  Min / max partials: per natural tile, seeded with
// +-inf (k_me_sum_fin seeds the fold with the first trimmed sample, as util.minmax does).
constexpr int kMeStagePitch = kPwLeafM + 8;
constexpr unsigned int kLeafPending = 0x7fc0deadu;
constexpr int kMeCompactGroup = 4;             // consecutive tiles per workgroup (a batch without filtered samples: a quarter of the workgroups to retire)
__global__ __launch_bounds__(kMeBlock) void k_me_compact(const float *x, const MsgState *st, const MsgTile *tiles, int64_t n_tiles_all, const int64_t *tile_pre,
                                                          float *kept, float *leaf_sums, float2 *tile_mm, const unsigned int *any_dirty) {
    __shared__ int s_cell[kMePer * (kMeBlock / 64)];
    __shared__ int s_total;
    __shared__ float s_v[kLeavesPerTile * kMeStagePitch];
    __shared__ float s_mn[kMeBlock / 64], s_mx[kMeBlock / 64];
    if (*any_dirty == 0u) return;
    const int wave = threadIdx.x >> 6;
    for (int64_t tix = (int64_t)blockIdx.x * kMeCompactGroup; tix < n_tiles_all && tix < ((int64_t)blockIdx.x + 1) * kMeCompactGroup; ++tix) {
        const MsgTile t = tiles[tix];
        const MsgState m = st[t.msg];
        if (me_clean(m)) continue;                           // nothing filtered (but the first sample): the later stages read the capture (me_src)
        const int64_t before = tile_pre[tix] - tile_pre[m.first_tile];             // kept samples in the message's earlier tiles
        const int cnt = (int)(tile_pre[tix + 1] - tile_pre[tix]);
        const int64_t base = m.start + (int64_t)t.idx * kMeTile;
        float v[kMePer];
        unsigned long long bal[kMePer];
#pragma unroll
        for (int j = 0; j < kMePer; ++j) { const int64_t i = base + j * kMeBlock + threadIdx.x; v[j] = (i < m.end) ? x[i] : -5.0f; }
#pragma unroll
        for (int j = 0; j < kMePer; ++j) {
            bal[j] = __ballot(v[j] > -4.0f);
            if ((threadIdx.x & 63) == 0) s_cell[j * (kMeBlock / 64) + wave] = __popcll(bal[j]);
        }
        me_cell_scan(s_cell, &s_total);
        // trimmed-relative position of the kept sample of rank r (inside the tile): p = off + r
        const int64_t off = before - m.a, p_hi = off + cnt;
        const int64_t n_full_leaves = (m.L / kPwChunkM) * (kPwChunkM / kPwLeafM);
        const int64_t l0 = off <= 0 ? 0 : (off + kPwLeafM - 1) / kPwLeafM;                          // first leaf that starts at or behind the tile's first kept sample
        int64_t l1 = p_hi <= 0 ? 0 : p_hi / kPwLeafM;                                                // leaves that end inside the tile
        const bool pending = l1 < n_full_leaves && l1 * kPwLeafM >= off && l1 * kPwLeafM < p_hi;     // leaf l1 starts here and ends later
        if (l1 > n_full_leaves) l1 = n_full_leaves;
        const int n_stage = l1 > l0 ? (int)(l1 - l0) : 0;                                            // whole leaves: at most 32
        const int64_t sh64 = l0 * kPwLeafM - off;                                                    // rank of leaf l0's first sample
        const int sh = (int)(sh64 > kMeTile ? kMeTile : sh64);
        const int r_lo = (int)(off >= 0 ? 0 : (-off > kMeTile ? kMeTile : -off));                    // ranks inside the trimmed range: [r_lo, r_hi)
        const int64_t r_hi64 = m.L - off;
        const int r_hi = (int)(r_hi64 < 0 ? 0 : (r_hi64 > kMeTile ? kMeTile : r_hi64));
        const unsigned long long below = me_lanes_below();
        float mn = __builtin_inff(), mx = -__builtin_inff();
        float *dst = kept + m.start + before;
#pragma unroll
        for (int j = 0; j < kMePer; ++j) {
            if (v[j] > -4.0f) {
                const int r = s_cell[j * (kMeBlock / 64) + wave] + __popcll(bal[j] & below);
                dst[r] = v[j];
                if (r >= r_lo && r < r_hi) { if (v[j] < mn) mn = v[j]; if (v[j] > mx) mx = v[j]; }
                const int q = r - sh;
                if (q >= 0 && q < n_stage * kPwLeafM) s_v[(q >> 7) * kMeStagePitch + (q & (kPwLeafM - 1))] = v[j];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float u = __shfl_xor(mn, o), w = __shfl_xor(mx, o);
            if (u < mn) mn = u;
            if (w > mx) mx = w;
        }
        if ((threadIdx.x & 63) == 0) { s_mn[wave] = mn; s_mx[wave] = mx; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < kMeBlock / 64; ++w) { if (s_mn[w] < mn) mn = s_mn[w]; if (s_mx[w] > mx) mx = s_mx[w]; }
            tile_mm[tix] = float2{mn, mx};
            if (pending) ((unsigned int *)leaf_sums)[m.first_tile * kLeavesPerTile + l1] = kLeafPending;
        }
        // the whole leaves: accumulator j of leaf lf, then ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)) (k_me_leaves)
        const int lf = threadIdx.x >> 3, jj = threadIdx.x & 7;
        const float *row = s_v + lf * kMeStagePitch + jj;
        float acc = 0.f;
        if (lf < n_stage) {
            acc = row[0];
#pragma unroll
            for (int i = 8; i < kPwLeafM; i += 8) acc += row[i];
        }
        acc = acc + __shfl_down(acc, 1);
        acc = acc + __shfl_down(acc, 2);
        acc = acc + __shfl_down(acc, 4);
        if (lf < n_stage && jj == 0) leaf_sums[m.first_tile * kLeavesPerTile + l0 + lf] = acc;
        __syncthreads();                                     // (s_v, s_cell, s_mn are the next tile's)
    }
}
// the leaf sums of a message with filtered samples -> half-chunk sums (me_half_tree's): one wavefront per chunk, lane = leaf; the
// leaves k_me_compact has left pending (two or three per chunk) first, 8 lanes per leaf
__global__ __launch_bounds__(kMeBlock) void k_me_dirty_trees(const float *kept, const MsgState *st, const MsgTile *tiles, int64_t n_tiles_all,
                                                              const float *leaf_sums, float *half_sums, const unsigned int *any_dirty) {
    if (*any_dirty == 0u) return;
    const int lane = threadIdx.x & 63;
    const int64_t tix = (int64_t)blockIdx.x * (kMeBlock / 64) + (threadIdx.x >> 6);                 // a wavefront per tile: the chunk's even one works
    if (tix >= n_tiles_all) return;
    const MsgTile t = tiles[tix];
    if (t.idx & 1) return;
    const MsgState m = st[t.msg];
    if (me_clean(m)) return;
    const int64_t c = t.idx >> 1;
    if (c >= m.L / kPwChunkM) return;
    float v = leaf_sums[m.first_tile * kLeavesPerTile + c * 64 + lane];
    unsigned long long pm = __ballot(__float_as_uint(v) == kLeafPending);
    const float *r = kept + m.start + m.a + c * kPwChunkM;   // the chunk's trimmed samples
    while (pm) {                                             // up to 8 pending leaves per round: group g of 8 lanes takes the g-th
        int leaf_of[8];
        unsigned long long tmp = pm;
#pragma unroll
        for (int g = 0; g < 8; ++g) { leaf_of[g] = tmp ? __builtin_ctzll(tmp) : -1; if (tmp) tmp &= tmp - 1ull; }
        int mine = -1;
#pragma unroll
        for (int g = 0; g < 8; ++g) if ((lane >> 3) == g) mine = leaf_of[g];
        float acc = 0.f;
        if (mine >= 0) {
            const float *q = r + mine * kPwLeafM + (lane & 7);
            acc = q[0];
#pragma unroll
            for (int i = 8; i < kPwLeafM; i += 8) acc += q[i];
        }
        acc = acc + __shfl_down(acc, 1);
        acc = acc + __shfl_down(acc, 2);
        acc = acc + __shfl_down(acc, 4);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const float sv = __shfl(acc, g * 8);
            if (leaf_of[g] == lane) v = sv;
        }
        pm = tmp;
    }
    v = v + __shfl_down(v, 1);                // s[i] = s[2i] + s[2i+1], level by level: lanes 0 and 32 end with the two halves
    v = v + __shfl_down(v, 2);
    v = v + __shfl_down(v, 4);
    v = v + __shfl_down(v, 8);
    v = v + __shfl_down(v, 16);
    if ((lane & 31) == 0) half_sums[m.first_tile + t.idx + (lane >> 5)] = v;
}

// ---- stage 4: numpy's float32 pairwise sums (np.mean, np.var) ---------------------------------------------------------------
// see estimators.hip for the order: chunks of 8192 accumulated left to right, a full chunk = perfect binary tree over 64 leaves
// of 128 elements, a leaf = 8 strided accumulators combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)); the irregular last chunk is
// split recursively (n2 = n/2 rounded down to a multiple of 8).
__device__ __forceinline__ float me_elem(const float *r, int64_t i, int mode, float mean) {
    const float v = r[i];
    if (mode == 0) return v;
    const float d = v - mean;
    return d * d;
}
// leaf sums of the FULL chunks of the squared deviations (np.var's second sum; the first one -- np.mean -- comes out of k_me_first or
// k_me_compact): 8 threads per leaf (one per accumulator), 32 leaves per tile, the tile's half-chunk tree on top
__global__ __launch_bounds__(kMeBlock) void k_me_leaves(const float *x, const float *kept, const MsgState *st, const MsgTile *tiles, float *half_sums) {
    __shared__ float s_h[kMeBlock / 64];
    const MsgTile t = tiles[blockIdx.x];
    const MsgState m = st[t.msg];
    const int64_t n_full_leaves = (m.L / kPwChunkM) * (kPwChunkM / kPwLeafM);
    // (32 leaves per tile, 64 per chunk: a tile's leaves are all full-chunk leaves or none is)
    if ((int64_t)t.idx * kLeavesPerTile >= n_full_leaves) return;
    const int64_t leaf = (int64_t)t.idx * kLeavesPerTile + (threadIdx.x >> 3);
    const int j = threadIdx.x & 7;
    const float *r = me_src(x, kept, m) + m.a + leaf * kPwLeafM;
    float acc = me_elem(r, j, 1, m.mean);
#pragma unroll
    for (int i = 8; i < kPwLeafM; i += 8) acc += me_elem(r, i + j, 1, m.mean);
    // ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)) across the 8 lanes of the leaf
    acc = acc + __shfl_down(acc, 1);
    acc = acc + __shfl_down(acc, 2);
    acc = acc + __shfl_down(acc, 4);
    me_half_tree(acc, s_h);
    __syncthreads();
    if (threadIdx.x == 0) half_sums[m.first_tile + t.idx] = (s_h[0] + s_h[1]) + (s_h[2] + s_h[3]);
}
// one wavefront per message: the left-to-right accumulation of the chunk sums, the irregular rest, the result
__device__ __forceinline__ float me_leaf_sum(const float *a, int len, int mode, float mean) {      // pw(a, len) for len <= 128
    float res;
    if (len < 8) {
        res = 0.f;
        for (int i = 0; i < len; ++i) res += me_elem(a, i, mode, mean);
    } else {
        float q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = me_elem(a, j, mode, mean);
        int i;
        for (i = 8; i < len - (len % 8); i += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) q[j] += me_elem(a, i + j, mode, mean);
        }
        res = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
        for (; i < len; ++i) res += me_elem(a, i, mode, mean);
    }
    return res;
}
// The irregular rest (< 8192 elements) is a tree whose leaves hold 64..128 elements each, so every leaf contains a multiple of 64
// within its first 64 elements: the thread that owns that multiple owns the leaf.  It finds the leaf by walking down from the
// root (registers only), sums it, and the tree is combined level by level from the deepest one up -- the node values live in LDS at
// the slot of the node's leftmost leaf, and the owner of that leaf adds the right child (left + right, as pw() does).
constexpr int kMeRestSlots = kPwChunkM / 64, kMeRestDepth = 9;
constexpr int kMeSumBlock = 1024;            // one message can be the whole capture: 32 768 tiles of min / max partials
// mode 0 also settles min / max (util.minmax: seeded with element 0, `<` / `>` folds -- NaN never replaces a value) from the per-tile
// partials of the pass that read the samples anyway: k_me_first for messages without filtered samples (tile t = window [4096 t, 4096 (t +
// 1)) of the trimmed range), k_me_compact for the others (its natural tiles).  The wavefronts that have nothing to do while wavefront 0 walks the chain of chunk sums do it.  (Rounds 3-5:
// a kernel of its own, k_me_minmax_fin.)
__global__ __launch_bounds__(kMeSumBlock) void k_me_sum_fin(const float *x, const float *kept, MsgState *st, const float *half_sums, const float2 *tile_mm,
                                                             int mode) {
    __shared__ float s_val[kMeRestSlots + 1];
    __shared__ float s_mn[kMeSumBlock / 64], s_mx[kMeSumBlock / 64];
    __shared__ __attribute__((aligned(16))) float s_cs[8192];
    const int m = blockIdx.x, tid = threadIdx.x;
    const MsgState s = st[m];
    if (s.L <= 0) return;
    const int64_t n_chunks = s.L / kPwChunkM;
    const int rest = (int)(s.L % kPwChunkM);
    const float *hs = half_sums + s.first_tile;              // half 2 c and half 2 c + 1 of chunk c (me_half_tree)
    const float *src = me_src(x, kept, s) + s.a;             // the trimmed samples
    int mm_from = -1;                                        // min / max: done by threads [mm_from, kMeSumBlock)
    auto minmax = [&](int from) {
        // partials by k_me_first (a tile per 4096 samples of the trimmed range) or by k_me_compact (every natural tile of the message)
        const int64_t len = s.end - s.start;
        const int64_t nt = me_clean(s) ? (s.L + kMeTile - 1) / kMeTile : ((len > 1 ? len : 1) + kMeTile - 1) / kMeTile;
        float mn = src[0], mx = mn;
        const int step = kMeSumBlock - from;
        for (int64_t u = tid - from; u < nt; u += step) {
            const float2 p = tile_mm[s.first_tile + u];
            if (p.x < mn) mn = p.x;
            if (p.y > mx) mx = p.y;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float a = __shfl_xor(mn, o), b = __shfl_xor(mx, o);
            if (a < mn) mn = a;
            if (b > mx) mx = b;
        }
        if ((tid & 63) == 0) { s_mn[tid >> 6] = mn; s_mx[tid >> 6] = mx; }
    };
    // total = ((0 + c0) + c1) + ...: strictly sequential float adds (16 384 of them when the capture is one message).  Wavefront 0
    // holds 64 chunk sums in its lanes and adds them through v_readlane (the next 64 are already loaded): a dependent add every few
    // cycles instead of an LDS round trip per term.  (Running the chain through the lanes instead -- acc[s] = acc[s - 1] + v[s] as one
    // v_add_f32_dpp wave_shr:1 per term -- is one instruction per term instead of two but was slower, 160 against 101 us for 16 384
    // terms: the DPP operand's wait states weigh more than the second instruction, which is off the dependent chain here.)
    // Round 6: the terms come out of LDS.  The chain is ONE v_add_f32 per term -- a wavefront issues one VALU instruction every four cycles,
    // and the v_readlane that fetched each term from the lanes (rounds 3-5) was a second VALU instruction per term: 9.5 cycles per
    // term, 101 us for the 16 384 chunk sums of a 1 GiB message.  Staged in LDS (32 KiB per piece, copied by the whole workgroup) every
    // lane reads the SAME four terms with one ds_read_b128 (a broadcast: the LDS pipe's instruction, off the VALU), the reads run ahead
    // of the adds (unrolled: the next 32 terms are in registers), and the chain is the adds alone: 67 us (a dependent add every ~8 cycles).
    float total = 0.f;
    constexpr int kPiece = 8192;                             // chunk sums per LDS piece (32 KiB)
    for (int64_t c0 = 0; c0 < n_chunks; c0 += kPiece) {
        const int nc = (int)((n_chunks - c0 < kPiece) ? n_chunks - c0 : kPiece);
        __syncthreads();                                     // (the piece before has been consumed)
        for (int c = tid * 4; c < nc; c += kMeSumBlock * 4) {
            const float *q = hs + 2 * (c0 + c);              // chunk = left half + right half: the root of its tree
            if (c + 3 < nc && (((uintptr_t)q) & 15) == 0) {
                const float4 u = *(const float4 *)q, w = *(const float4 *)(q + 4);
                *(float4 *)(s_cs + c) = float4{u.x + u.y, u.z + u.w, w.x + w.y, w.z + w.w};
            } else { for (int e = 0; e < 4 && c + e < nc; ++e) s_cs[c + e] = q[2 * e] + q[2 * e + 1]; }
        }
        __syncthreads();
        if (tid < 64) {
            int c = 0;
            for (; c + 32 <= nc; c += 32) {
                float4 q[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) q[e] = *(const float4 *)(s_cs + c + 4 * e);
#pragma unroll
                for (int e = 0; e < 8; ++e) { total = total + q[e].x; total = total + q[e].y; total = total + q[e].z; total = total + q[e].w; }
            }
            for (; c < nc; ++c) total = total + s_cs[c];
        } else if (mode == 0 && mm_from < 0) minmax(64);     // (beside the chain)
        if (mode == 0 && mm_from < 0) mm_from = 64;
    }
    if (mode == 0 && mm_from < 0) { minmax(0); mm_from = 0; }       // (no full chunk: nobody was waiting for a chain)
    // pw(a, n) = pw(a, n2) + pw(a + n2, n - n2), n2 = n / 2 rounded down to a multiple of 8, down to leaves of <= 128 elements
    const int e = tid * 64;                                  // this thread's element
    int off[kMeRestDepth], len[kMeRestDepth];
    int depth = 0, leaf_o = -1;                              // depth / offset of the leaf that holds e
    bool owner = false;
    if (e < rest) {
        int o = 0, l = rest;
#pragma unroll
        for (int d = 0; d < kMeRestDepth; ++d) {
            off[d] = o; len[d] = l;
            if (l > kPwLeafM) {
                int n2 = l / 2;
                n2 -= n2 % 8;
                if (e < o + n2) l = n2; else { o += n2; l -= n2; }
                depth = d + 1;
            }
        }
        owner = (e - o) < 64;                                // the first multiple of 64 inside the leaf [o, o + l)
        leaf_o = o;
        if (owner) s_val[tid] = me_leaf_sum(src + n_chunks * kPwChunkM + o, l, mode, s.mean);
        // (off / len at depths beyond the leaf repeat the leaf)
    }
    __syncthreads();
#pragma unroll
    for (int d = kMeRestDepth - 2; d >= 0; --d) {
        // the node at depth d on this thread's path is internal when the leaf lies deeper; this thread combines it when its leaf is
        // the node's leftmost one (same offset)
        float right = 0.f;
        bool act = false;
        if (owner && d < depth) {
            const int o = off[d], l = len[d];
            int n2 = l / 2;
            n2 -= n2 % 8;
            if (leaf_o == o) { act = true; right = s_val[(o + n2 + 63) / 64]; }
        }
        __syncthreads();
        if (act) s_val[tid] = s_val[tid] + right;
        __syncthreads();
    }
    if (tid == 0) {
        if (rest) total = total + s_val[0];
        const float res = total / (float)s.L;                   // float32 sum / float32 count
        if (mode == 0) {
            st[m].mean = res;
            float mn = src[0], mx = mn;
            for (int w = mm_from / 64; w < kMeSumBlock / 64; ++w) { if (s_mn[w] < mn) mn = s_mn[w]; if (s_mx[w] > mx) mx = s_mx[w]; }
            st[m].mn = mn; st[m].mx = mx;
        } else st[m].var = res;
    }
}

// ---- stage 5: bin edges (np.arange(min, max + step, step) in float64) and histogram ---------------------------------------
constexpr int kMeHistSmallBins = 512;        // see k_me_hist
__global__ void k_me_bins(MsgState *st, int n_msgs, int64_t max_bins, unsigned int *n_wide, int *wide_list) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_msgs) return;
    MsgState s = st[m];
    int64_t ne = 0;
    double e0 = 0, delta = 0;
    if (s.L > 0) {
        const double step = (double)s.var, start = (double)s.mn, stop = (double)s.mx + step;
        // np.arange: length = ceil((stop - start) / step); step == 0 -> ZeroDivisionError, NaN -> ValueError: no histogram (:246-248)
        if (step == step && step != 0.0 && start == start && stop == stop) {
            const double q = (stop - start) / step;
            if (q == q && q < 9.0e15) {
                double len = ceil(q);
                if (len < 0) len = 0;
                ne = (int64_t)len;
                e0 = start;
                delta = (start + step) - start;               // np.arange fills first + i * (second - first)
            }
        }
        if (ne < 2) ne = 0;                                    // np.histogram needs at least two edges (ValueError -> None)
    }
    st[m].n_edges = ne; st[m].e0 = e0; st[m].delta = delta;
    // a message that needs the wide histogram kernel (k_me_hist<kMeHistLds>): on its list (rare: a nearly constant message; that kernel
    // walks the listed messages' tiles instead of looking at every tile of the batch for one -- 17 us on config 3's capture)
    if (ne - 1 > kMeHistSmallBins) wide_list[atomicAdd(n_wide, 1u)] = m;
    (void)max_bins;
}
__device__ __forceinline__ double me_edge(const MsgState &s, int64_t i) { return s.e0 + (double)i * s.delta; }

constexpr int kMeHistLds = 4096;             // bins kept in LDS per workgroup (the pool's max_bins is at most this)
// A message's row of the histogram pool holds max_bins counters and a demodulated message uses a handful: the workgroups of k_me_hist
// add their counts to one of up to 64 REPLICAS of the histogram inside the row (replica r at r * stride, a multiple of 128 bytes), and
// k_me_peaks folds them into replica 0 first.  (Device-wide atomic adds to ONE word are settled one after the other on the memory side:
// the 12 288 adds of a capture that is one message -- 4096 workgroups, three bins -- held the histogram pass at 150-180 us whatever
// the workgroups did before them; profiles/r06q_hist_ab.txt.)
__host__ __device__ inline void me_replicas(int64_t nb, int64_t max_bins, int &R, int64_t &stride) {
    stride = (nb + 31) & ~(int64_t)31;
    const int64_t r = stride > 0 ? max_bins / stride : 1;
    R = (int)(r > 64 ? 64 : r);
    if (R < 1) { R = 1; stride = nb; }
}
__device__ __forceinline__ double me_edge32(const MsgState &s, int i) { return s.e0 + (double)i * s.delta; }
// smallest float32 that is >= the float64 edge / largest float32 that is <= it: a float32 sample compares with the float64 edge exactly
// as it compares with this float32 (np.histogram casts the samples to float64 and searches the float64 edges)
__device__ __forceinline__ float me_f32_at_or_above(double e) {
    float f = (float)e;
    if ((double)f < e) f = __uint_as_float(f >= 0.f ? (f == 0.f ? 1u : __float_as_uint(f) + 1u) : __float_as_uint(f) - 1u);
    return f;
}
__device__ __forceinline__ float me_f32_at_or_below(double e) {
    float f = (float)e;
    if ((double)f > e) f = __uint_as_float(f > 0.f ? __float_as_uint(f) - 1u : (f == 0.f ? 0x80000001u : __float_as_uint(f) + 1u));
    return f;
}
#ifndef URH_HIST_GROUP
#define URH_HIST_GROUP 8
#endif
constexpr int kMeHistGroup = URH_HIST_GROUP;              // consecutive tiles per workgroup: one flush of the LDS counters per message and group
// Two instantiations share the tiles: BINS = kMeHistSmall serves the messages with at most that many bins (a demodulated message
// has a few dozen) out of 4 KiB of LDS -- eight workgroups per CU instead of four: the pass is bound by instruction issue and by the
// latency of each tile's loads, and twice the wavefronts hide twice as much of it -- BINS = kMeHistLds serves the others; a workgroup
// skips the messages of the other class.
#ifndef URH_HIST_WORDS
#define URH_HIST_WORDS 2048    // LDS counter words of the small instantiation of k_me_hist (A/B: 1024, 4096)
#endif
#ifndef URH_HIST_LOGC_MAX
#define URH_HIST_LOGC_MAX 6    // at most 2^this copies of a counter: one per lane (one per thread, 8: no faster, profiles/r06q_hist_ab.txt)
#endif
#ifndef URH_HIST_ROUNDS
#define URH_HIST_ROUNDS 2      // ballot rounds per row of k_me_hist before what is left goes through LDS atomics (A/B: 0, 1)
#endif
constexpr int kMeHistSmall = kMeHistSmallBins;
#ifdef URH_OCC_HIST
#define URH_HIST_OCC __attribute__((amdgpu_waves_per_eu(URH_OCC_HIST)))
#else
#define URH_HIST_OCC
#endif
template <int BINS>
__global__ __launch_bounds__(kMeBlock) URH_HIST_OCC void k_me_hist(const float *x, const float *kept, const MsgState *st, const MsgTile *tiles, int64_t n_tiles,
                                                       int64_t max_bins, unsigned int *counts, const unsigned int *n_wide, const int *wide_list) {
    if (BINS != kMeHistSmall && *n_wide == 0u) return;       // no message of this class in the batch (k_me_bins)
    // demodulated signals sit on two or four levels: nearly every sample of a message lands in a handful of bins.  Counting
    // straight into device memory serialises the whole pass on those few addresses (40 ms for a 1 GiB capture): the workgroup counts in
    // LDS, and only the non-empty bins of its tiles reach device memory -- when the message changes or at the end of its kMeHistGroup
    // consecutive tiles, into one of the message's replicas (me_replicas).  The bin is guessed in float32 and checked against a float32
    // table of the edges in LDS that decides exactly as the float64 edges do (me_f32_at_or_above).
    // The wide instantiation (more than kMeHistSmall bins: rare) counts the lanes that share a bin with one ballot per distinct bin for the
    // first two bins it meets in a row of 64 samples and sends what is left of the row through LDS atomics (rounds 3-6a: both did).
    // Round 6 (late): the small instantiation keeps 2^logC COPIES of every counter, copy (lane mod 2^logC) at word (bin << logC) + copy:
    // with 64 copies (up to 32 bins -- a demodulated message has a handful) every lane of a wavefront owns its bank and the count is ONE
    // ds_add_u32 per sample without a conflict, whatever the samples are; the ballot rounds below -- three VALU instructions and a
    // scalar popcount per row and distinct bin, in dependent chains across the two pipes -- were most of the kernel's time.  (Plain LDS
    // atomics on ONE copy serialise on the few populated addresses: +250 us, profiles/r04e_ab_hist_rounds.txt.)
    constexpr bool kPriv = BINS == kMeHistSmall;
    constexpr int kWords = kPriv ? URH_HIST_WORDS : BINS;      // counter words: bins << logC <= kWords
    __shared__ unsigned int s_c[kWords + (kPriv ? 64 : 0)];    // (+ 64 words nobody reads: where the lanes without a sample add)
    __shared__ float s_e[BINS + 2];                         // s_e[k] = first float32 inside bin k or above; s_e[nb] = first float32 beyond the last bin
    int logC = 0;
    int cur_msg = -1, nb = 0;
    bool valid = false, big = false;
    MsgState m = st[0];
    unsigned int *out = counts;
    const int lane = threadIdx.x & 63;
    // the tiles [tile0, tile1) -- consecutive in the table -- with one flush per message met and one at the end
    auto run = [&](const int64_t tile0, const int64_t tile1) {
    cur_msg = -1; valid = false; big = false;
    for (int64_t tix = tile0; tix <= tile1; ++tix) {
    const bool last = tix == tile1;
    MsgTile t = tiles[last ? tile1 - 1 : tix];
    if (last || t.msg != cur_msg) {                            // (workgroup-uniform)
        if (valid) {                                           // flush the message's counters
            __syncthreads();
            if (kPriv) {
                // the copies of a bin are 2^logC <= 64 consecutive words = consecutive lanes (256 is a multiple): fold them with shuffles
                const int C = 1 << logC;
                for (int w = threadIdx.x; w < (nb << logC); w += kMeBlock) {
                    unsigned int v = s_c[w];
                    for (int o = 1; o < C; o <<= 1) v += (unsigned int)__shfl_xor((int)v, o);
                    if ((w & (C - 1)) == 0 && v) atomicAdd(&out[w >> logC], v);
                }
            } else {
                for (int k = threadIdx.x; k < nb; k += kMeBlock) if (s_c[k]) atomicAdd(&out[k], s_c[k]);
            }
            __syncthreads();
        }
        if (last) break;
        cur_msg = t.msg;
        m = st[t.msg];
        big = false;
        valid = !(m.L <= 0 || m.n_edges < 2 || m.n_edges - 1 > max_bins || m.n_edges - 1 > INT32_MAX - 1);
        if (valid && m.n_edges - 1 > kMeHistLds) { valid = false; big = (BINS == kMeHistLds); }   // more bins than the LDS holds: the plain path below
        if (valid && ((BINS == kMeHistSmall) != (m.n_edges - 1 <= kMeHistSmall))) valid = false;  // the other instantiation's message
        nb = (valid || big) ? (int)(m.n_edges - 1) : 0;
        out = counts + (int64_t)t.msg * max_bins;
        if (valid) {                                               // this workgroup's replica of the message's histogram
            int R; int64_t stride;
            me_replicas(nb, max_bins, R, stride);
            out += (int64_t)(blockIdx.x % (unsigned)R) * stride;
        }
        if (valid) {
            if (kPriv) { logC = 0; while (logC < URH_HIST_LOGC_MAX && (nb << (logC + 1)) <= kWords) ++logC; }
            for (int k = threadIdx.x; k < (nb << logC); k += kMeBlock) s_c[k] = 0u;
            for (int k = threadIdx.x; k <= nb; k += kMeBlock) {
                // the last bin is closed on the right: "beyond it" starts at the first float32 above the last edge
                float e = me_f32_at_or_above(me_edge32(m, k));
                if (k == nb) { const float top = me_f32_at_or_below(me_edge32(m, nb)); e = __uint_as_float(top >= 0.f ? (top == 0.f ? 1u : __float_as_uint(top) + 1u) : __float_as_uint(top) - 1u); }
                s_e[k] = e;
            }
            __syncthreads();
        }
    }
    if (big && (int64_t)t.idx * kMeTile < m.L) {
        // a histogram of more than kMeHistLds bins (a caller with a larger pool): float64 edges per sample, counters in device memory
        const double e0 = m.e0, eN = me_edge32(m, nb);
        const float e0f = (float)e0, invf = (m.delta > 0.0) ? (float)(1.0 / m.delta) : 0.f;
        const float *r = me_src(x, kept, m) + m.a;
        for (int j = 0; j < kMePer; ++j) {
            const int64_t i = (int64_t)t.idx * kMeTile + threadIdx.x + (int64_t)j * kMeBlock;
            if (i >= m.L) continue;
            const double v = (double)r[i];
            if (!(v >= e0 && v <= eN)) continue;
            const float g = (r[i] - e0f) * invf;
            int k = (g >= 0.f) ? ((g < (float)(nb - 1)) ? (int)g : nb - 1) : 0;
            while (k > 0 && me_edge32(m, k) > v) --k;
            while (k < nb - 1 && me_edge32(m, k + 1) <= v) ++k;      // last bin closed on the right
            atomicAdd(&out[k], 1u);
        }
        continue;
    }
    if (!valid || (int64_t)t.idx * kMeTile >= m.L) continue;
    const float lo_all = s_e[0], hi_all = s_e[nb];             // inside: lo_all <= v < hi_all (NaN: neither)
    const float e0f = (float)m.e0, invf = (m.delta > 0.0) ? (float)(1.0 / m.delta) : 0.f;
    // (32-bit offsets from a wave-uniform base, as in k_me_first)
    const int64_t tbase = (int64_t)t.idx * kMeTile, left = m.L - tbase;
    const float *r = me_src(x, kept, m) + m.a + tbase;
    const int lim = (int)(left < 0 ? 0 : (left > kMeTile ? kMeTile : left));
    // Round 6: a histogram does not care which lane holds which sample -- thread t takes four runs of four CONSECUTIVE samples (16-byte
    // loads, 4-byte aligned: a message starts anywhere), a quarter of the load instructions; and the bin of a sample is the float32 guess
    // CHECKED ONCE against the table (one ds_read2: s_e[k] <= v < s_e[k + 1]) -- the trimmed range's minimum and maximum are the first and
    // the last edge, so every sample lies inside, and the guess is off only within a rounding error of an edge.  A wavefront in which any
    // check fails (that, a NaN, a sample outside) takes the exact path for its sixteen rows: guess, one step either way, check, search.
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    float val[kMePer];
#pragma unroll
    for (int j = 0; j < kMePer / 4; ++j) {
        const int i = 4 * ((int)threadIdx.x + j * kMeBlock);
        if (i + 3 < lim) { const f4u q = *(const f4u *)(r + i); val[4 * j] = q.x; val[4 * j + 1] = q.y; val[4 * j + 2] = q.z; val[4 * j + 3] = q.w; }
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) val[4 * j + e] = (i + e < lim) ? r[i + e] : __builtin_nanf("");
        }
    }
    if (kPriv) {
        // four samples at a time: guess, check (the table reads of the four are in flight together), count; a lane whose check fails (a
        // rounding error next to an edge, a NaN, padding) searches the table on its own -- nothing here needs the wavefront to agree
        const int sh = logC + 2;                                               // counter addresses in bytes: (bin << sh) + mine
        const unsigned int mine = (unsigned int)(lane & ((1 << logC) - 1)) * 4u, dump = (unsigned int)((nb << logC) + lane) * 4u;
#pragma unroll
        for (int j0 = 0; j0 < kMePer; j0 += 4) {
            int k[4];
            float lo[4], hi[4];
            unsigned int at[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) k[e] = min(max((int)((val[j0 + e] - e0f) * invf), 0), nb - 1);
#pragma unroll
            for (int e = 0; e < 4; ++e) { lo[e] = s_e[k[e]]; hi[e] = s_e[k[e] + 1]; at[e] = ((unsigned int)k[e] << sh) + mine; }
            bool ok[4], all = true;
#pragma unroll
            for (int e = 0; e < 4; ++e) { ok[e] = (val[j0 + e] >= lo[e]) & (val[j0 + e] < hi[e]); all = all & ok[e]; }
            if (!all) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = val[j0 + e];
                    if (!ok[e]) {
                        int q = k[e];
                        if ((v >= lo_all) & (v < hi_all)) {
                            while (q > 0 && s_e[q] > v) --q;
                            while (q < nb - 1 && s_e[q + 1] <= v) ++q;
                            at[e] = ((unsigned int)q << sh) + mine;
                        } else at[e] = dump;
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd((unsigned int *)((char *)s_c + at[e]), 1u);
        }
        continue;
    }
    int bin[kMePer];
    {
        bool bad = false;
#pragma unroll
        for (int j = 0; j < kMePer; ++j) {
            const float v = val[j];
            int k = (int)((v - e0f) * invf);
            k = (k < 0) ? 0 : ((k > nb - 1) ? nb - 1 : k);
            const float lo = s_e[k], hi = s_e[k + 1];
            const bool ok = v >= lo && v < hi;
            bad |= !ok;                                            // any failed check (NaN included) sends the wavefront to the exact path
            bin[j] = ok ? k : -1;
        }
        // (the padding beyond the tile's last sample is NaN: a wavefront of the last tile takes the exact path, which leaves padding out)
        if (__any(bad)) {
#pragma unroll
            for (int j = 0; j < kMePer; ++j) {
                const float v = val[j];
                const bool in = v >= lo_all && v < hi_all;         // inside (and not NaN)
                const float g = (v - e0f) * invf;
                int k = (g >= 0.f) ? ((g < (float)(nb - 1)) ? (int)g : nb - 1) : 0;
                k = in ? k : 0;
                const float lo = s_e[k], hi = s_e[k + 1];
                k += (v >= hi && k < nb - 1) ? 1 : ((v < lo && k > 0) ? -1 : 0);
                if (in) {
                    while (k > 0 && s_e[k] > v) --k;
                    while (k < nb - 1 && s_e[k + 1] <= v) ++k;
                }
                bin[j] = in ? k : -1;
            }
        }
    }
    // count: a bin first seen in row j is counted over rows j..15 at once (a demodulated signal has two or three bins per wavefront);
    // after two such rounds what is left of the row goes through LDS atomics, one per sample
#pragma unroll
    for (int j = 0; j < kMePer; ++j) {
        unsigned long long todo = __ballot(bin[j] >= 0);
        for (int round = 0; todo && round < URH_HIST_ROUNDS; ++round) {
            const int leader = __ffsll((long long)todo) - 1;
            const int kl = __shfl(bin[j], leader);
            unsigned int cnt = 0;
#pragma unroll
            for (int q = j; q < kMePer; ++q) {
                const bool same = bin[q] == kl;
                cnt += (unsigned)__popcll(__ballot(same));
                bin[q] = same ? -1 : bin[q];
            }
            if (lane == leader) atomicAdd(&s_c[kl], cnt);
            todo = __ballot(bin[j] >= 0);
        }
        if (todo) {
            if (bin[j] >= 0) atomicAdd(&s_c[bin[j]], 1u);
            bin[j] = -1;
        }
    }
    }
    };
    if (kPriv) {
        const int64_t tile0 = (int64_t)blockIdx.x * kMeHistGroup;
        run(tile0, (tile0 + kMeHistGroup < n_tiles) ? tile0 + kMeHistGroup : n_tiles);
    } else {
        // the listed messages' tiles, in groups of kMeHistGroup dealt round robin to the workgroups
        const unsigned int nw = *n_wide;
        for (unsigned int wi = 0; wi < nw; ++wi) {
            const MsgState w = st[wide_list[wi]];
            const int64_t len = w.end - w.start, nt = ((len > 1 ? len : 1) + kMeTile - 1) / kMeTile;
            for (int64_t g = blockIdx.x; g * kMeHistGroup < nt; g += gridDim.x) {
                const int64_t tile0 = w.first_tile + g * kMeHistGroup, tile_end = w.first_tile + nt;
                run(tile0, (tile0 + kMeHistGroup < tile_end) ? tile0 + kMeHistGroup : tile_end);
            }
        }
    }
}

// ---- stage 6: the peak picking of detect_center (AutoInterpretation.py:250-277) ---------------------------------------------
// Up to two bins, most populated first, that are strict maxima over +-(window - 1) bins (bins outside the histogram count as 0);
// the center is the mean of their left edges.  Two strict maxima are at least `window` bins apart: at most 21 candidates.  The
// reference walks the bins in np.argsort order, which leaves the order of equally populated bins to the sort implementation: that
// matters only when the second and third candidate tie -- reported as flag 3, decided by the caller with numpy itself.
__global__ __launch_bounds__(kMeBlock) void k_me_peaks(MsgState *st, unsigned int *counts, int64_t max_bins) {
    __shared__ int s_n;
    __shared__ int s_idx[64];
    __shared__ unsigned int s_cnt[64];
    const int m = blockIdx.x;
    const MsgState s = st[m];
    if (s.n_edges < 2) { if (threadIdx.x == 0) { st[m].peak_flag = 0; st[m].peak_center = 0.0; } return; }
    if (s.n_edges - 1 > max_bins) { if (threadIdx.x == 0) { st[m].peak_flag = 2; st[m].peak_center = 0.0; } return; }
    const int nb = (int)(s.n_edges - 1);
    unsigned int *y = counts + (int64_t)m * max_bins;
    if (nb <= kMeHistLds) {                                      // fold the replicas (k_me_hist; the plain path beyond kMeHistLds bins has one)
        int R; int64_t stride;
        me_replicas(nb, max_bins, R, stride);
        if (R > 1 && nb <= kMeBlock) {                            // a handful of bins, 64 replicas: every (replica, bin) is one thread's load
            __shared__ unsigned int s_fold[kMeBlock];
            s_fold[threadIdx.x] = 0u;
            __syncthreads();
            for (int q = threadIdx.x; q < R * nb; q += kMeBlock) {
                const unsigned int v = y[(int64_t)(q / nb) * stride + (q % nb)];
                if (v) atomicAdd(&s_fold[q % nb], v);
            }
            __syncthreads();
            if ((int)threadIdx.x < nb) y[threadIdx.x] = s_fold[threadIdx.x];
            __syncthreads();
        } else if (R > 1) {                                       // (more than 256 bins: at most 16 replicas)
            for (int i = threadIdx.x; i < nb; i += kMeBlock) {
                unsigned int v = y[i];
                for (int r = 1; r < R; ++r) v += y[(int64_t)r * stride + i];
                y[i] = v;
            }
            __syncthreads();
        }
    }
    int w = (int)(0.05 * (double)nb) + 1;
    if (w < 2) w = 2;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += kMeBlock) {
        const unsigned int yi = y[i];
        bool ok = yi > 0u;
        for (int d = 1; d < w && ok; ++d) {
            const unsigned int l = (i - d >= 0) ? y[i - d] : 0u, r = (i + d < nb) ? y[i + d] : 0u;
            ok = yi > l && yi > r;
        }
        if (ok) { const int q = atomicAdd(&s_n, 1); if (q < 64) { s_idx[q] = i; s_cnt[q] = yi; } }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nc = s_n < 64 ? s_n : 64;
        int best[3] = {-1, -1, -1};                            // the three most populated candidates (ties: lower bin first -- only
        for (int q = 0; q < nc; ++q) {                         // used to see whether a tie matters)
            int c = q;
            for (int r = 0; r < 3; ++r) {
                if (c < 0) break;
                if (best[r] < 0 || s_cnt[c] > s_cnt[best[r]] || (s_cnt[c] == s_cnt[best[r]] && s_idx[c] < s_idx[best[r]])) { const int tmp = best[r]; best[r] = c; c = tmp; }
            }
        }
        int64_t flag = 0;
        double center = 0.0;
        if (nc == 1) { flag = 1; center = me_edge(s, s_idx[best[0]]) / 1.0; }
        else if (nc >= 2) {
            if (nc >= 3 && s_cnt[best[1]] == s_cnt[best[2]]) flag = 3;
            else { flag = 1; center = (me_edge(s, s_idx[best[0]]) + me_edge(s, s_idx[best[1]])) / 2.0; }
        }
        st[m].peak_flag = flag; st[m].peak_center = center;
    }
}

// ---- plateaus: boundaries of (x <= center) per message, first 25 % (get_plateau_lengths) -----------------------------------
__device__ __forceinline__ void me_edge_flags(const float *x, const MsgState &m, const MsgTile &t, bool (&e)[kMePer]) {
    const float cen = (float)m.center;
    const float *r = x + m.start;
    float cur[kMePer], prev[kMePer];
#pragma unroll
    for (int j = 0; j < kMePer; ++j) {
        const int64_t i = (int64_t)t.idx * kMeTile + j * kMeBlock + threadIdx.x;          // position inside the message
        const bool in = i >= 1 && i < m.window;
        cur[j] = in ? r[i] : 0.f;
        prev[j] = in ? r[i - 1] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < kMePer; ++j) e[j] = (cur[j] <= cen) != (prev[j] <= cen);
}
__global__ __launch_bounds__(kMeBlock) void k_me_edge_count(const float *x, const MsgState *st, const MsgTile *tiles, int32_t *tile_cnt) {
    __shared__ int s_w[kMeBlock / 64];
    const MsgTile t = tiles[blockIdx.x];
    const MsgState m = st[t.msg];
    int c = 0;
    if (m.center == m.center) {
        bool e[kMePer];
        me_edge_flags(x, m, t, e);
#pragma unroll
        for (int j = 0; j < kMePer; ++j) c += e[j] ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ __launch_bounds__(kMeBlock) void k_me_edge_compact(const float *x, MsgState *st, const MsgTile *tiles, const int64_t *tile_pre,
                                                               int32_t *edges /* positions inside the message, region = message start */) {
    __shared__ int s_cell[kMePer * (kMeBlock / 64)];
    __shared__ int s_total;
    const MsgTile t = tiles[blockIdx.x];
    const MsgState m = st[t.msg];
    if (!(m.center == m.center)) { if (threadIdx.x == 0 && t.idx == 0) st[t.msg].edge_cnt = 0; return; }
    const int64_t before = tile_pre[blockIdx.x] - tile_pre[m.first_tile];
    const int wave = threadIdx.x >> 6;
    bool e[kMePer];
    me_edge_flags(x, m, t, e);
    unsigned long long bal[kMePer];
#pragma unroll
    for (int j = 0; j < kMePer; ++j) {
        bal[j] = __ballot(e[j]);
        if ((threadIdx.x & 63) == 0) s_cell[j * (kMeBlock / 64) + wave] = __popcll(bal[j]);
    }
    const int total = me_cell_scan(s_cell, &s_total);
    const unsigned long long below = me_lanes_below();
#pragma unroll
    for (int j = 0; j < kMePer; ++j)
        if (e[j]) edges[m.start + before + s_cell[j * (kMeBlock / 64) + wave] + __popcll(bal[j] & below)] =
                      (int32_t)((int64_t)t.idx * kMeTile + j * kMeBlock + threadIdx.x);
    const int64_t last_tile = (m.window + kMeTile - 1) / kMeTile - 1;
    if (threadIdx.x == 0 && t.idx == last_tile) st[t.msg].edge_cnt = before + total;
}
// plateau k = [B_{k-1}, B_k) (B_{-1} = 0) counts while the plateaus appended before it sum to less than limit = 25 * len / 100
// (C integer division): lengths[k] = B_k - B_{k-1} for B_{k-1} < limit.  In place: edges[] becomes lengths[].
__global__ __launch_bounds__(64) void k_me_plateaus(MsgState *st, int32_t *edges, int percentage) {
    const int m = blockIdx.x, lane = threadIdx.x;
    const MsgState s = st[m];
    if (!(s.center == s.center)) { if (lane == 0) st[m].n_plateaus = 0; return; }
    const int64_t len = s.end - s.start, limit = ((int64_t)percentage * len) / 100;
    int32_t *b = edges + s.start;
    const int64_t found = s.edge_cnt;
    // is the window enough?  it is when it is the whole message, or holds a boundary at or beyond the limit
    const bool enough = (s.window >= len) || (found > 0 && (int64_t)b[found - 1] >= limit);
    // number of plateaus kept: those whose start B_{k-1} < limit, i.e. k = 0 (start 0, kept iff limit > 0 ... the reference's
    // loop appends plateau 0 at the first boundary whenever it gets there: current_sum = 0 < limit unless limit == 0)
    int64_t keep = 0;
    for (int64_t k0 = 0; k0 < found; k0 += 64) {
        const int64_t k = k0 + lane;
        bool kept = false;
        if (k < found) { const int64_t startk = k ? (int64_t)b[k - 1] : 0; kept = startk < limit; }
        keep += __popcll(__ballot(kept));
    }
    // lengths in place, back to front would overwrite what is still needed: go front to back keeping the previous boundary
    int64_t prev_last = 0;
    for (int64_t k0 = 0; k0 < keep; k0 += 64) {
        const int64_t k = k0 + lane;
        int32_t cur = 0, prev = 0;
        if (k < keep) { cur = b[k]; prev = k ? ((k == k0) ? (int32_t)prev_last : b[k - 1]) : 0; }
        const int32_t last = __shfl(cur, 63);
        __syncthreads();
        if (k < keep) b[k] = cur - prev;
        prev_last = last;
        __syncthreads();
    }
    if (lane == 0) st[m].n_plateaus = enough ? keep : -1;
}

// lengths of every message, back to back, as uint64 (what get_plateau_lengths returns)
__global__ __launch_bounds__(64) void k_me_gather(const MsgState *st, const int32_t *lengths, const int64_t *out_begin, uint64_t *out) {
    const int m = blockIdx.x;
    const int64_t k = st[m].n_plateaus, o = out_begin[m];
    const int32_t *src = lengths + st[m].start;
    for (int64_t i = threadIdx.x; i < k; i += 64) out[o + i] = (uint64_t)(uint32_t)src[i];
}

// the selected messages only (out_begin[m] < 0: not wanted)
__global__ __launch_bounds__(64) void k_me_gather_some(const MsgState *st, const int32_t *lengths, const int64_t *out_begin, uint64_t *out) {
    const int m = blockIdx.x;
    const int64_t k = st[m].n_plateaus, o = out_begin[m];
    if (o < 0) return;
    const int32_t *src = lengths + st[m].start;
    for (int64_t i = threadIdx.x; i < k; i += 64) out[o + i] = (uint64_t)(uint32_t)src[i];
}

// What AutoInterpretation.estimate does with a message's plateau lengths (AutoInterpretation.py:416-433) depends, for a message
// without glitches, on the MULTISET of lengths only: the tolerance on the distinct values, the rounding on the digit counts, the divisor
// histogram on (value, multiplicity).  A demodulated message has thousands of plateaus and a few dozen distinct lengths: one
// workgroup per message counts them in an LDS hash table and appends its (value, count) pairs to a pool -- kilobytes cross PCIe instead
// of 8 bytes per plateau.  (Messages with glitches need merge_plateaus, which walks the sequence: the host asks for those.)
constexpr int kLcSlots = 4096, kLcMaxDistinct = 3072;
// ---- the chained call's (urhgpu_msg_estimate) traffic with the host: message ranges in, one record per message out ----------------
// Rounds 5-6a moved both stages' MsgState arrays both ways: 352 bytes per message built and copied on the host, uploaded, read back and
// copied again -- 0.5 MB each way for config 3's 1500 messages, 60-70 us of host time around the kernels.  The states are built ON the
// device from the uploaded ranges (16 bytes per message), and what the host needs of them comes back as one EstRec per message.
struct EstRec {
    double stats[8];         // urhgpu_msg_center_stats' row: kept, L, min, max, mean, var, n_edges, e0
    double center;           // peak_center
    int64_t flag;            // peak_flag
    int64_t n_plateaus, pairs_base, pairs_n;     // stage 2 (k_me_plateaus, k_me_len_counts)
};
__device__ __forceinline__ void me_write_rec(EstRec *rec, const MsgState &s1, const MsgState &s2) {
    EstRec r;
    r.stats[0] = (double)s1.kept; r.stats[1] = (double)s1.L; r.stats[2] = (double)s1.mn; r.stats[3] = (double)s1.mx; r.stats[4] = (double)s1.mean;
    r.stats[5] = (double)s1.var; r.stats[6] = (double)s1.n_edges; r.stats[7] = s1.e0;
    r.center = s1.peak_center; r.flag = s1.peak_flag;
    r.n_plateaus = s2.n_plateaus; r.pairs_base = s2.pairs_base; r.pairs_n = s2.pairs_n;
    *rec = r;
}
__global__ __launch_bounds__(kMeBlock) void k_me_len_counts(MsgState *st, const int32_t *lengths, unsigned long long *pool_count, uint64_t *pool,
                                                             int64_t cap_pairs, const MsgState *st1, EstRec *rec) {
    __shared__ int s_key[kLcSlots];                       // length + 1, 0 = empty
    __shared__ unsigned int s_cnt[kLcSlots];
    __shared__ int s_distinct, s_over, s_pos;
    __shared__ long long s_base;
    const int m = blockIdx.x;
    const int64_t k = st[m].n_plateaus;
    for (int i = threadIdx.x; i < kLcSlots; i += kMeBlock) { s_key[i] = 0; s_cnt[i] = 0u; }
    if (threadIdx.x == 0) { s_distinct = 0; s_over = 0; s_pos = 0; s_base = 0; }
    __syncthreads();
    const int32_t *src = lengths + st[m].start;
    for (int64_t i = threadIdx.x; i < k; i += kMeBlock) {
        const int32_t v = src[i];
        if (v < 0 || v == INT32_MAX) { s_over = 1; break; }
        const int key = v + 1;
        unsigned h = ((unsigned)v * 2654435761u) >> 20;    // 12 bits
        for (int probe = 0; probe < kLcSlots; ++probe) {
            if (s_over) break;
            const int old = atomicCAS(&s_key[h], 0, key);
            if (old == 0 && atomicAdd(&s_distinct, 1) >= kLcMaxDistinct) s_over = 1;
            if (old == 0 || old == key) { atomicAdd(&s_cnt[h], 1u); break; }
            h = (h + 1) & (kLcSlots - 1);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long base = 0;
        if (!s_over && s_distinct > 0) {
            base = (long long)atomicAdd(pool_count, (unsigned long long)s_distinct);
            if (base + s_distinct > cap_pairs) s_over = 1;
        }
        s_base = base;
        st[m].pairs_base = base;
        st[m].pairs_n = s_over ? -1 : (k > 0 ? s_distinct : 0);
        if (rec) me_write_rec(rec + m, st1[m], st[m]);       // a chained call: the message's record for the host (urhgpu_msg_estimate)
    }
    __syncthreads();
    if (s_over || k <= 0) return;
    for (int i = threadIdx.x; i < kLcSlots; i += kMeBlock) {
        if (s_key[i]) {
            const int pidx = atomicAdd(&s_pos, 1);
            pool[2 * (s_base + pidx)] = (uint64_t)(s_key[i] - 1);
            pool[2 * (s_base + pidx) + 1] = (uint64_t)s_cnt[i];
        }
    }
}

// the tile table from the message table: tile t belongs to the message whose [first_tile, next first_tile) holds it (the host uploads
// one MsgState per message, not one entry per 4096 samples)
__global__ __launch_bounds__(256) void k_me_fill_tiles(const MsgState *st, int n_msgs, MsgTile *tiles, int64_t n_tiles) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_tiles) return;
    int lo = 0, hi = n_msgs - 1;                            // last message with first_tile <= t
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (st[mid].first_tile <= t) lo = mid; else hi = mid - 1;
    }
    tiles[t] = MsgTile{lo, (int32_t)(t - st[lo].first_tile)};
}

// one workgroup: both stages' states with their first_tile (an exclusive scan of the tile counts over the messages); stage 2 searches
// [0, percentage % + extra_window) of every message for boundaries
constexpr int kMeInitBlock = 1024;
__device__ __forceinline__ int64_t me_tiles_of(int64_t span) { return ((span > 1 ? span : 1) + kMeTile - 1) / kMeTile; }
__global__ __launch_bounds__(kMeInitBlock) void k_me_init_states(const int64_t *ranges, int n_msgs, int percentage, int64_t extra_window, MsgState *st1,
                                                                  MsgState *st2) {
    __shared__ int64_t s_a[kMeInitBlock / 64], s_b[kMeInitBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (n_msgs + kMeInitBlock - 1) / kMeInitBlock;
    const int lo = tid * per, hi = (lo + per < n_msgs) ? lo + per : n_msgs;
    auto window_of = [&](int64_t len) { const int64_t limit = ((int64_t)percentage * len) / 100; return (len < limit + extra_window) ? len : limit + extra_window; };
    int64_t a = 0, b = 0;                                    // tiles of this thread's messages: stage 1, stage 2
    for (int m = lo; m < hi; ++m) {
        const int64_t len = ranges[2 * m + 1] - ranges[2 * m];
        a += me_tiles_of(len);
        b += me_tiles_of(window_of(len));
    }
    int64_t ia = a, ib = b;                                  // inclusive scan over the threads
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int64_t ua = __shfl_up(ia, o), ub = __shfl_up(ib, o);
        if (lane >= o) { ia += ua; ib += ub; }
    }
    if (lane == 63) { s_a[wave] = ia; s_b[wave] = ib; }
    __syncthreads();
    int64_t fa = ia - a, fb = ib - b;
    for (int w = 0; w < wave; ++w) { fa += s_a[w]; fb += s_b[w]; }
    for (int m = lo; m < hi; ++m) {
        const int64_t start = ranges[2 * m], end = ranges[2 * m + 1], len = end - start;
        MsgState z;
        memset(&z, 0, sizeof(z));
        z.start = start; z.end = end; z.center = __builtin_nan("");
        z.first_tile = fa; z.window = len;
        st1[m] = z;
        z.first_tile = fb; z.window = window_of(len);
        st2[m] = z;
        fa += me_tiles_of(len);
        fb += me_tiles_of(z.window);
    }
}

// stage 2's center of a chained call: what the host would hand over -- the picked center as a C float (get_plateau_lengths takes `float
// center`, auto_interpretation.pyx:179), NaN where stage 1 has none to give (no histogram, more bins than the pool, a tie numpy decides)
__global__ void k_me_chain_center(const MsgState *st1, MsgState *st2, int n_msgs) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_msgs) return;
    st2[m].center = (st1[m].peak_flag == 1) ? (double)(float)st1[m].peak_center : __builtin_nan("");
}

}  // namespace urh

using namespace urh;

namespace {

struct MsgBatch {
    std::vector<MsgState> host;
    MsgState *d_state = nullptr;
    MsgTile *d_tiles = nullptr;
    int64_t n_tiles = 0;
};

// urhgpu_msg_estimate: center statistics and plateau decisions back to back -- the centers stay on the device (k_me_chain_center), the
// scratch of both stages comes from ONE reservation, the states of both land in the pinned zone behind ONE synchronisation
// Round 6 (late): message ranges up in one small copy, the states built on the device; the records, the pool fill and the first pairs of
// the pool are ONE block of the arena, fetched with one copy behind stage 2 (rounds 5-6a: an upload and a read-back of the states per
// stage, in the stream between the kernels, and three copies at the end: each with its ~9 us of idle stream around it).
struct EstChain {
    MsgBatch b1, b2;         // the stages' batches: tile counts only (the states are built on the device, k_me_init_states)
    MsgState *d_st1 = nullptr, *d_st2 = nullptr;
    EstRec *d_rec = nullptr; // [records | pool fill (256 bytes) | pool]: one block, fetched with one copy
    unsigned long long *d_pool_count = nullptr;
    uint64_t *d_pool = nullptr;
    int64_t cap_pairs = 0;
    size_t rec_pad = 0;      // bytes of the records in the block (and where they land)
    std::vector<char> land;  // landing zone when the context's pinned one is too small
    const char *landed = nullptr;       // where the block has landed (behind stage 2's synchronisation)
};

// tile table over [start, start + span_m) of every message; span = whole message (window = nullptr) or the given windows
int build_batch(urhgpu_ctx *ctx, const int64_t *ranges, int n_msgs, int64_t n, const int64_t *windows, MsgBatch &b) {
    b.host.resize((size_t)n_msgs);
    int64_t n_tiles = 0;
    for (int m = 0; m < n_msgs; ++m) {
        const int64_t s = ranges[2 * m], e = ranges[2 * m + 1];
        if (s < 0 || e < s || e > n) return URHGPU_ERR_ARG;
        // per-message scratch (compacted samples, boundary lists) lives at [start, end) of capture-sized arrays: messages must be
        // ascending and disjoint, as segmentation delivers them -- overlapping ranges would race on it silently
        if (m > 0 && s < ranges[2 * m - 1]) return URHGPU_ERR_ARG;
        MsgState st;
        memset(&st, 0, sizeof(st));
        st.start = s; st.end = e; st.first_tile = n_tiles;
        st.center = __builtin_nan("");
        const int64_t span = windows ? windows[m] : e - s;
        st.window = span;
        const int64_t nt = (std::max<int64_t>(span, 1) + kMeTile - 1) / kMeTile;
        if (nt > INT32_MAX) return URHGPU_ERR_UNSUPPORTED;
        n_tiles += nt;
        b.host[(size_t)m] = st;
    }
    b.n_tiles = n_tiles;
    return URHGPU_OK;
}

// the batch of the plateau stage: boundaries are searched in [0, percentage % + extra_window) of every message
int plateau_batch(urhgpu_ctx *ctx, const int64_t *ranges, int n_msgs, int64_t n, int percentage, int64_t extra_window, MsgBatch &b) {
    std::vector<int64_t> windows((size_t)n_msgs);
    for (int m = 0; m < n_msgs; ++m) {
        const int64_t len = ranges[2 * m + 1] - ranges[2 * m];
        if (len > INT32_MAX) return URHGPU_ERR_UNSUPPORTED;       // positions inside a message are 32-bit
        const int64_t limit = ((int64_t)percentage * len) / 100;
        windows[(size_t)m] = std::min<int64_t>(len, limit + extra_window);
    }
    return build_batch(ctx, ranges, n_msgs, n, windows.data(), b);
}
// (value, count) pairs the pool holds: a message beyond its share of the pool is decided from its sequence
int64_t plateau_cap_pairs(int n_msgs) { return std::max<int64_t>(4096, (int64_t)n_msgs * 256); }

}  // namespace

extern "C" {

// one batch of urhgpu_msg_center_stats: at most kCenterBatchBytes of histogram pool (see below)
static void center_stats_collect(const MsgBatch &b, int n_msgs, int64_t max_bins, const std::vector<unsigned int> &hist, double *out_stats,
                                 int64_t *out_hist, double *out_center, int32_t *out_flag);

static int center_stats_batch(urhgpu_ctx *ctx, const float *d_x, int64_t n, const int64_t *ranges, int n_msgs, int64_t max_bins,
                              double *out_stats, int64_t *out_hist, double *out_center, int32_t *out_flag, EstChain *chain = nullptr) {
    MsgBatch local;
    MsgBatch &b = chain ? chain->b1 : local;
    if (!chain) { URH_TRY(build_batch(ctx, ranges, n_msgs, n, nullptr, b)); }      // (a chained call: built, placed and uploaded by urhgpu_msg_estimate)
    // scratch: state, tiles, per-tile counts / min-max, leaf sums, the compacted samples, the histogram pool
    const size_t need = (size_t)n_msgs * sizeof(MsgState) + (size_t)(b.n_tiles + 1) * (sizeof(MsgTile) + 4 + 8 + 8 + 4 + kLeavesPerTile * 4) +
                        (size_t)n * 4 + (size_t)n_msgs * (size_t)max_bins * 4 + (size_t)n_msgs * 4 + 18 * 256;
    if (!chain) { URH_TRY(ctx->arena.reserve(need)); ctx->arena.reset(); }       // (a chained call has reserved both stages' scratch)
    MsgState *d_st = chain ? chain->d_st1 : (MsgState *)ctx->arena.take((size_t)n_msgs * sizeof(MsgState));
    MsgTile *d_tiles = (MsgTile *)ctx->arena.take((size_t)b.n_tiles * sizeof(MsgTile));
    int64_t *d_pre = (int64_t *)ctx->arena.take((size_t)(b.n_tiles + 1) * 8);
    float2 *d_mm = (float2 *)ctx->arena.take((size_t)b.n_tiles * 8);
    float *d_half = (float *)ctx->arena.take((size_t)(b.n_tiles + 1) * 4);
    float *d_leaf = (float *)ctx->arena.take((size_t)b.n_tiles * kLeavesPerTile * 4);
    float *d_kept = (float *)ctx->arena.take((size_t)std::max<int64_t>(n, 1) * 4);
    int *d_wide = (int *)ctx->arena.take((size_t)n_msgs * 4);
    // the histogram pool, the any_wide flag behind it and the per-tile counts: what ONE fill clears before the passes
    const size_t pool_bytes = ((size_t)n_msgs * (size_t)max_bins * 4 + 256 + 255) & ~size_t(255);
    unsigned int *d_hist = (unsigned int *)ctx->arena.take(pool_bytes + (size_t)b.n_tiles * 4);
    unsigned int *d_any_wide = d_hist ? d_hist + (size_t)n_msgs * (size_t)max_bins : nullptr;
    unsigned int *d_any_dirty = d_any_wide ? d_any_wide + 1 : nullptr;
    int32_t *d_cnt = d_hist ? (int32_t *)((char *)d_hist + pool_bytes) : nullptr;
    if (!d_st || !d_tiles || !d_cnt || !d_pre || !d_mm || !d_half || !d_leaf || !d_kept || !d_wide || !d_hist) return URHGPU_ERR_ARG;
    hipStream_t s = ctx->stream;
    if (!chain) URH_HIP(hipMemcpyAsync(d_st, b.host.data(), (size_t)n_msgs * sizeof(MsgState), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_me_fill_tiles, dim3((unsigned)((b.n_tiles + 255) / 256)), dim3(256), 0, s, d_st, n_msgs, d_tiles, b.n_tiles);
    URH_HIP(hipMemsetAsync(d_hist, 0, pool_bytes + (size_t)b.n_tiles * 4, s));
    const unsigned gt = (unsigned)b.n_tiles, gm = (unsigned)((n_msgs + 63) / 64);
    hipLaunchKernelGGL(k_me_first, dim3(gt), dim3(kMeBlock), 0, s, d_x, d_st, d_tiles, d_cnt, d_mm, d_half);
    hipLaunchKernelGGL(k_me_tile_scan, dim3(1), dim3(kMeScanBlock), 0, s, d_cnt, b.n_tiles, d_pre, d_st, n_msgs, d_any_dirty);
    // messages with filtered samples (the two kernels retire at once when there is none): compaction with the first leaf sums and min / max
    hipLaunchKernelGGL(k_me_compact, dim3((unsigned)((b.n_tiles + kMeCompactGroup - 1) / kMeCompactGroup)), dim3(kMeBlock), 0, s, d_x, d_st, d_tiles, b.n_tiles,
                       d_pre, d_kept, d_leaf, d_mm, d_any_dirty);
    hipLaunchKernelGGL(k_me_dirty_trees, dim3((unsigned)((b.n_tiles + kMeBlock / 64 - 1) / (kMeBlock / 64))), dim3(kMeBlock), 0, s, d_kept, d_st, d_tiles,
                       b.n_tiles, d_leaf, d_half, d_any_dirty);
    hipLaunchKernelGGL(k_me_sum_fin, dim3((unsigned)n_msgs), dim3(kMeSumBlock), 0, s, d_x, d_kept, d_st, d_half, d_mm, 0);
    hipLaunchKernelGGL(k_me_leaves, dim3(gt), dim3(kMeBlock), 0, s, d_x, d_kept, d_st, d_tiles, d_half);
    hipLaunchKernelGGL(k_me_sum_fin, dim3((unsigned)n_msgs), dim3(kMeSumBlock), 0, s, d_x, d_kept, d_st, d_half, d_mm, 1);
    hipLaunchKernelGGL(k_me_bins, dim3(gm), dim3(64), 0, s, d_st, n_msgs, max_bins, d_any_wide, d_wide);
    hipLaunchKernelGGL((k_me_hist<kMeHistSmall>), dim3((unsigned)((b.n_tiles + kMeHistGroup - 1) / kMeHistGroup)), dim3(kMeBlock), 0, s, d_x, d_kept, d_st,
                       d_tiles, b.n_tiles, max_bins, d_hist, d_any_wide, d_wide);
    hipLaunchKernelGGL((k_me_hist<kMeHistLds>), dim3((unsigned)std::min<int64_t>((b.n_tiles + kMeHistGroup - 1) / kMeHistGroup, 1024)), dim3(kMeBlock), 0, s,
                       d_x, d_kept, d_st, d_tiles, b.n_tiles, max_bins, d_hist, d_any_wide, d_wide);
    hipLaunchKernelGGL(k_me_peaks, dim3((unsigned)n_msgs), dim3(kMeBlock), 0, s, d_st, d_hist, max_bins);
    URH_HIP(hipGetLastError());
    if (chain) return URHGPU_OK;                             // stage 2 goes on from here; the states come back with its own
    std::vector<unsigned int> hist;
    const size_t st_bytes = (size_t)n_msgs * sizeof(MsgState);
    const bool st_pinned = ctx->h_small && st_bytes <= kSmallPinned;     // (a truly asynchronous copy; pageable memory otherwise)
    URH_HIP(hipMemcpyAsync(st_pinned ? (void *)ctx->h_small : (void *)b.host.data(), d_st, st_bytes, hipMemcpyDeviceToHost, s));
    if (out_hist) {                                          // the histograms themselves: only a caller that has to break a tie wants them
        hist.resize((size_t)n_msgs * (size_t)max_bins);
        URH_HIP(hipMemcpyAsync(hist.data(), d_hist, hist.size() * 4, hipMemcpyDeviceToHost, s));
    }
    URH_HIP(wait_stream(ctx, s));
    if (st_pinned) memcpy(b.host.data(), ctx->h_small, st_bytes);
    center_stats_collect(b, n_msgs, max_bins, hist, out_stats, out_hist, out_center, out_flag);
    return URHGPU_OK;
}

static void center_stats_collect(const MsgBatch &b, int n_msgs, int64_t max_bins, const std::vector<unsigned int> &hist, double *out_stats,
                                 int64_t *out_hist, double *out_center, int32_t *out_flag) {
    for (int m = 0; m < n_msgs; ++m) {
        const MsgState &st = b.host[(size_t)m];
        double *o = out_stats + 8 * (size_t)m;
        o[0] = (double)st.kept; o[1] = (double)st.L; o[2] = (double)st.mn; o[3] = (double)st.mx; o[4] = (double)st.mean; o[5] = (double)st.var;
        o[6] = (double)st.n_edges; o[7] = st.e0;
        if (out_center) out_center[m] = st.peak_center;
        if (out_flag) out_flag[m] = (int32_t)st.peak_flag;
        if (out_hist) {
            int64_t *h = out_hist + (size_t)m * (size_t)max_bins;
            const int64_t nb = (st.n_edges >= 2 && st.n_edges - 1 <= max_bins) ? st.n_edges - 1 : 0;
            for (int64_t k = 0; k < max_bins; ++k) h[k] = (k < nb) ? (int64_t)hist[(size_t)m * (size_t)max_bins + (size_t)k] : 0;
        }
    }
}

// The histogram pool holds max_bins counters per message and is cleared for every call: a capture cut into 10^5 .. 10^6 segments (a
// bursty capture whose modulation is not OOK, so nothing was merged) would ask for gigabytes of it at once.  The messages are
// therefore taken in batches whose pool stays below kCenterBatchBytes; the compacted samples (indexed by message start) are shared.
constexpr size_t kCenterBatchBytes = size_t(64) << 20;

int urhgpu_msg_center_stats(urhgpu_ctx *ctx, const float *d_x, int64_t n, const int64_t *ranges, int n_msgs, int64_t max_bins,
                            double *out_stats, int64_t *out_hist, double *out_center, int32_t *out_flag) {
    if (!ctx || n < 0 || n_msgs < 0 || max_bins < 1 || (n_msgs > 0 && (!ranges || !out_stats || !d_x))) return URHGPU_ERR_ARG;
    if (n_msgs == 0) return URHGPU_OK;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(join_tail(ctx));                               // pipelined context: the arena below may still serve the last pass's tail
    const int64_t per = std::max<int64_t>(1, (int64_t)(kCenterBatchBytes / ((size_t)max_bins * 4)));
    for (int64_t m0 = 0; m0 < n_msgs; m0 += per) {
        const int nb = (int)std::min<int64_t>(per, n_msgs - m0);
        URH_TRY(center_stats_batch(ctx, d_x, n, ranges + 2 * m0, nb, max_bins, out_stats + 8 * m0,
                                   out_hist ? out_hist + (size_t)m0 * (size_t)max_bins : nullptr, out_center ? out_center + m0 : nullptr,
                                   out_flag ? out_flag + m0 : nullptr));
    }
    return URHGPU_OK;
}

int urhgpu_msg_plateaus(urhgpu_ctx *ctx, const float *d_x, int64_t n, const int64_t *ranges, const double *centers, int n_msgs,
                        int percentage, int64_t extra_window, int64_t *out_off, uint64_t *out_len, int64_t cap_total) {
    if (!ctx || n < 0 || n_msgs < 0 || percentage < 0 || extra_window < 0 || cap_total < 0 ||
        (n_msgs > 0 && (!ranges || !centers || !out_off || !d_x)))
        return URHGPU_ERR_ARG;
    if (out_off) out_off[0] = 0;
    if (n_msgs == 0) return URHGPU_OK;
    URH_HIP(hipSetDevice(ctx->device));
    // boundaries are searched in [0, 25 % + extra_window) of every message; a message whose window holds no boundary at or
    // beyond the 25 % mark comes back with count -1 (the caller repeats it with a larger window)
    std::vector<int64_t> windows((size_t)n_msgs);
    for (int m = 0; m < n_msgs; ++m) {
        const int64_t len = ranges[2 * m + 1] - ranges[2 * m];
        if (len > INT32_MAX) return URHGPU_ERR_UNSUPPORTED;       // positions inside a message are 32-bit
        const int64_t limit = ((int64_t)percentage * len) / 100;
        windows[(size_t)m] = std::min<int64_t>(len, limit + extra_window);
    }
    URH_TRY(join_tail(ctx));                               // pipelined context: the arena below may still serve the last pass's tail
    MsgBatch b;
    URH_TRY(build_batch(ctx, ranges, n_msgs, n, windows.data(), b));
    for (int m = 0; m < n_msgs; ++m) b.host[(size_t)m].center = centers[m];
    const size_t need = (size_t)n_msgs * sizeof(MsgState) + (size_t)(b.n_tiles + 1) * (sizeof(MsgTile) + 4 + 8) + (size_t)std::max<int64_t>(n, 1) * 4 + 8 * 256;
    URH_TRY(ctx->arena.reserve(need));
    ctx->arena.reset();
    MsgState *d_st = (MsgState *)ctx->arena.take((size_t)n_msgs * sizeof(MsgState));
    MsgTile *d_tiles = (MsgTile *)ctx->arena.take((size_t)b.n_tiles * sizeof(MsgTile));
    int32_t *d_cnt = (int32_t *)ctx->arena.take((size_t)b.n_tiles * 4);
    int64_t *d_pre = (int64_t *)ctx->arena.take((size_t)(b.n_tiles + 1) * 8);
    int32_t *d_edges = (int32_t *)ctx->arena.take((size_t)std::max<int64_t>(n, 1) * 4);
    if (!d_st || !d_tiles || !d_cnt || !d_pre || !d_edges) return URHGPU_ERR_ARG;
    hipStream_t s = ctx->stream;
    URH_HIP(hipMemcpyAsync(d_st, b.host.data(), (size_t)n_msgs * sizeof(MsgState), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_me_fill_tiles, dim3((unsigned)((b.n_tiles + 255) / 256)), dim3(256), 0, s, d_st, n_msgs, d_tiles, b.n_tiles);
    const unsigned gt = (unsigned)b.n_tiles;
    hipLaunchKernelGGL(k_me_edge_count, dim3(gt), dim3(kMeBlock), 0, s, d_x, d_st, d_tiles, d_cnt);
    hipLaunchKernelGGL(k_me_tile_scan, dim3(1), dim3(kMeScanBlock), 0, s, d_cnt, b.n_tiles, d_pre, (MsgState *)nullptr, 0, (unsigned int *)nullptr);
    hipLaunchKernelGGL(k_me_edge_compact, dim3(gt), dim3(kMeBlock), 0, s, d_x, d_st, d_tiles, d_pre, d_edges);
    hipLaunchKernelGGL(k_me_plateaus, dim3((unsigned)n_msgs), dim3(64), 0, s, d_st, d_edges, percentage);
    URH_HIP(hipGetLastError());
    URH_HIP(hipMemcpyAsync(b.host.data(), d_st, (size_t)n_msgs * sizeof(MsgState), hipMemcpyDeviceToHost, s));
    URH_HIP(wait_stream(ctx, s));
    // end offsets; a message whose window was too small has no plateaus here and is marked -(end + 1)
    std::vector<int64_t> begin((size_t)n_msgs);
    int64_t total = 0;
    for (int m = 0; m < n_msgs; ++m) {
        const int64_t k = b.host[(size_t)m].n_plateaus;
        begin[(size_t)m] = total;
        total += (k > 0) ? k : 0;
        out_off[m + 1] = (k < 0) ? -(total + 1) : total;
    }
    if (total > cap_total) { out_off[n_msgs] = total; return URHGPU_ERR_CAPACITY; }
    if (total > 0) {
        URH_TRY(ctx->staging.reserve((size_t)n_msgs * 8 + (size_t)total * 8 + 1024));
        ctx->staging.reset();
        int64_t *d_begin = (int64_t *)ctx->staging.take((size_t)n_msgs * 8);
        uint64_t *d_out = (uint64_t *)ctx->staging.take((size_t)total * 8);
        if (!d_begin || !d_out) return URHGPU_ERR_ARG;
        URH_HIP(hipMemcpyAsync(d_begin, begin.data(), (size_t)n_msgs * 8, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_me_gather, dim3((unsigned)n_msgs), dim3(64), 0, s, d_st, d_edges, d_begin, d_out);
        URH_HIP(hipGetLastError());
        URH_HIP(hipMemcpyAsync(out_len, d_out, (size_t)total * 8, hipMemcpyDeviceToHost, s));
        URH_HIP(wait_stream(ctx, s));
    }
    return URHGPU_OK;
}

// auto_interpretation.merge_plateaus (auto_interpretation.pyx:145-176) on the host: plateaus of at most `tolerance` samples are
// glitches and merge with their neighbours; a run of alternating glitches (67, 1, 10, 1, 21) merges as a whole.  Sequential by
// nature (what a merge swallows decides where the next one starts) and a few thousand elements long: native host code, as the
// reference's is.  out: n entries of room; *n_out = merged plateaus (at most max_count + 1).
int urhgpu_merge_plateaus(const uint64_t *plateaus, int64_t n, uint64_t tolerance, uint64_t max_count, uint64_t *out, int64_t *n_out) {
    if (n < 0 || !n_out || (n > 0 && (!plateaus || !out))) return URHGPU_ERR_ARG;
    if (n == 0) { *n_out = 0; return URHGPU_OK; }
    uint64_t cur = 0;
    out[0] = plateaus[0] <= tolerance ? 0 : plateaus[0];
    int64_t i = 1;
    while (i < n && cur < max_count) {
        if (plateaus[i] <= tolerance) {
            int64_t span = 2;                                   // the glitch and the plateau after it ...
            while (i + span < n && plateaus[i + span] <= tolerance) span += 2;     // ... and further alternating glitches
            const int64_t stop = std::min<int64_t>(n, i + span);
            uint64_t sum = 0;
            for (int64_t j = i - 1; j < stop; ++j) sum += plateaus[j];
            out[cur] = sum;
            i += span;
        } else {
            out[++cur] = plateaus[i++];
        }
    }
    *n_out = (int64_t)cur + 1;
    return URHGPU_OK;
}

}  // extern "C"

// ---- per-message decisions on plateau lengths: host arithmetic on a few thousand integers, all messages in one call -------------
namespace {

// numpy's float64 summation (np.add.reduce): chunks of 8192 accumulated left to right, pairwise inside a chunk
double np_pairwise_f64(const double *a, int64_t n) {
    if (n < 8) { double r = 0.0; for (int64_t i = 0; i < n; ++i) r += a[i]; return r; }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8) for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    const double x = np_pairwise_f64(a, n2), y = np_pairwise_f64(a + n2, n - n2);
    return x + y;
}
double np_sum_f64(const double *a, int64_t n) {
    double total = 0.0;
    for (int64_t c = 0; c < n; c += 8192) total += np_pairwise_f64(a + c, std::min<int64_t>(8192, n - c));
    return total;
}

// AutoInterpretation.estimate_tolerance_from_plateau_lengths (AutoInterpretation.py:280-298); -1: None, -2: undefined in the reference.
// u: the distinct lengths, ascending; total: how many plateaus there are
int64_t tolerance_of_unique(const std::vector<uint64_t> &u, size_t total) {
    if (total <= 1) return -1;
    const int64_t n = (int64_t)u.size();
    std::vector<double> d((size_t)n);
    for (int64_t i = 0; i < n; ++i) d[(size_t)i] = (double)u[(size_t)i];
    const double mean = np_sum_f64(d.data(), n) / (double)n;
    std::vector<double> sq((size_t)n);
    for (int64_t i = 0; i < n; ++i) { const double x = d[(size_t)i] - mean; sq[(size_t)i] = x * x; }
    const double sd = sqrt(np_sum_f64(sq.data(), n) / (double)n);
    bool any = false; uint64_t maximum = 0;                     // max_without_outliers(unique, z=2) (:14-18)
    for (int64_t i = 0; i < n; ++i)
        if (fabs(d[(size_t)i] - mean) <= 2.0 * sd) { any = true; if (u[(size_t)i] > maximum) maximum = u[(size_t)i]; }
    if (!any) return -2;
    const double limit = 0.05 * (double)maximum;
    if (u[0] > 1 && (double)u[0] >= limit) return 0;
    uint64_t result = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (u[(size_t)i] > 1 && (double)u[(size_t)i] >= limit) break;
        result = u[(size_t)i];
    }
    return (int64_t)result;
}
int64_t tolerance_of(const std::vector<uint64_t> &p) {
    if (p.size() <= 1) return -1;
    std::vector<uint64_t> u(p);
    std::sort(u.begin(), u.end());
    u.erase(std::unique(u.begin(), u.end()), u.end());
    return tolerance_of_unique(u, p.size());
}

int digits_of(uint64_t v) { int d = 1; while (v >= 10) { v /= 10; ++d; } return d; }

// round_plateau_lengths (:313-326, in place): keep the median number of digits (at most 3); int(round(p / f)) * f, round = half to even
void round_lengths(std::vector<uint64_t> &m) {
    size_t dhist[24] = {0};
    for (size_t i = 0; i < m.size(); ++i) ++dhist[digits_of(m[i])];
    auto kth = [&](size_t k) { size_t c = 0; for (int d = 0; d < 24; ++d) { c += dhist[d]; if (k < c) return d; } return 23; };   // k-th smallest digit count
    const double idx = 0.5 * (double)(m.size() - 1);
    const size_t lo = (size_t)floor(idx), hi = (size_t)ceil(idx);
    const double med = (double)kth(lo) + ((double)kth(hi) - (double)kth(lo)) * (idx - (double)lo);
    const int n_digits = std::min(3, (int)med);
    double f = 1.0;
    for (int k = 1; k < n_digits; ++k) f *= 10.0;
    for (auto &v : m) v = (uint64_t)nearbyint((double)v / f) * (uint64_t)f;
}

// get_threshold_divisor_histogram (auto_interpretation.pyx:113-143) from the multiset of values: the non-zero entries as (count, index).
// vc: (value, multiplicity), value != 0, ascending
void divisor_histogram_vc(const std::vector<std::pair<uint64_t, uint64_t>> &vc, std::vector<std::pair<uint64_t, uint64_t>> &hist, float threshold) {
    const double thr = (double)threshold;
    hist.clear();
    for (size_t a = 0; a < vc.size(); ++a) {
        uint64_t c = (0.0 < thr) ? vc[a].second * (vc[a].second - 1) / 2 : 0;      // pairs of equal values: ratio 1, fraction 0
        for (size_t b = a + 1; b < vc.size(); ++b) {
            const uint64_t mn = vc[a].first, mx = vc[b].first;
            if ((double)mx / (double)mn - (double)(mx / mn) < thr) c += vc[a].second * vc[b].second;
        }
        if (c) hist.push_back({c, vc[a].first});
    }
}
void divisor_histogram(const std::vector<uint64_t> &m, std::vector<std::pair<uint64_t, uint64_t>> &hist, float threshold = 0.2f) {
    std::vector<uint64_t> vals(m);
    std::sort(vals.begin(), vals.end());
    std::vector<std::pair<uint64_t, uint64_t>> vc;              // (value, count), value != 0, ascending
    for (size_t i = 0; i < vals.size();) {
        size_t j = i;
        while (j < vals.size() && vals[j] == vals[i]) ++j;
        if (vals[i] != 0) vc.push_back({vals[i], (uint64_t)(j - i)});
        i = j;
    }
    divisor_histogram_vc(vc, hist, threshold);
}

// the selection loop of get_bit_length_from_plateau_lengths (:358-370) on the histogram's non-zero entries; -2: the outcome depends on how
// np.argsort orders equal counts (the caller lets numpy order the histogram: urhgpu_msg_divisor_histogram / urhgpu_bit_length_from_order)
int64_t select_bit_length(std::vector<std::pair<uint64_t, uint64_t>> &hist) {
    if (hist.empty()) return -2;                                // all counts zero: argsort's order of equal elements decides
    std::sort(hist.begin(), hist.end(), [](const std::pair<uint64_t, uint64_t> &x, const std::pair<uint64_t, uint64_t> &y) { return x.first > y.first; });
    const uint64_t max_count = hist[0].first;
    int64_t result = (int64_t)hist[0].second;
    for (size_t i = 1; i < hist.size(); ++i) {
        if ((double)hist[i].first < 0.25 * (double)max_count) break;
        if (hist[i].first == hist[i - 1].first) return -2;      // equal counts among the candidates
        if ((double)hist[i].second <= 0.5 * (double)result) result = (int64_t)hist[i].second;
    }
    if (hist.size() > 1 && hist[1].first == max_count) return -2;
    return result;
}

// get_bit_length_from_plateau_lengths (:344-370) on the merged lengths (rounded in place)
int64_t bit_length_of(std::vector<uint64_t> &m) {
    if (m.empty()) return 0;
    if (m.size() == 1) return (int64_t)m[0];
    round_lengths(m);
    std::vector<std::pair<uint64_t, uint64_t>> hist;
    divisor_histogram(m, hist);
    return select_bit_length(hist);
}

// the same from the multiset of lengths: vc = (value, multiplicity), ascending by value, at least two plateaus in all
int64_t bit_length_of_counts(const std::vector<std::pair<uint64_t, uint64_t>> &vc) {
    // round_plateau_lengths: the median number of digits over ALL plateaus (np.percentile(..., 50): linear interpolation)
    size_t dhist[24] = {0};
    size_t total = 0;
    for (const auto &e : vc) { dhist[digits_of(e.first)] += (size_t)e.second; total += (size_t)e.second; }
    auto kth = [&](size_t k) { size_t c = 0; for (int d = 0; d < 24; ++d) { c += dhist[d]; if (k < c) return d; } return 23; };
    const double idx = 0.5 * (double)(total - 1);
    const size_t lo = (size_t)floor(idx), hi = (size_t)ceil(idx);
    const double med = (double)kth(lo) + ((double)kth(hi) - (double)kth(lo)) * (idx - (double)lo);
    const int n_digits = std::min(3, (int)med);
    double f = 1.0;
    for (int k = 1; k < n_digits; ++k) f *= 10.0;
    std::vector<std::pair<uint64_t, uint64_t>> r;               // rounded values (rounding is monotone: still ascending), equal ones joined
    for (const auto &e : vc) {
        const uint64_t v = (uint64_t)nearbyint((double)e.first / f) * (uint64_t)f;
        if (!r.empty() && r.back().first == v) r.back().second += e.second;
        else r.push_back({v, e.second});
    }
    if (!r.empty() && r.front().first == 0) r.erase(r.begin());
    std::vector<std::pair<uint64_t, uint64_t>> hist;
    divisor_histogram_vc(r, hist, 0.2f);
    return select_bit_length(hist);
}

// tolerance + merged plateaus of one message (AutoInterpretation.py:416-420); false: the tolerance is undefined
bool merged_lengths(const uint64_t *lens, int64_t n, int64_t *tol_out, std::vector<uint64_t> &merged) {
    std::vector<uint64_t> p(lens, lens + n);
    const int64_t tol = tolerance_of(p);
    *tol_out = tol;
    if (tol == -2) return false;
    if (tol > 0) {
        merged.resize(p.size());
        int64_t k = 0;
        if (!p.empty()) (void)urhgpu_merge_plateaus(p.data(), (int64_t)p.size(), (uint64_t)tol, 10000, merged.data(), &k);
        merged.resize((size_t)k);
    } else {
        merged.swap(p);
    }
    return true;
}

}  // namespace

// A small pool of host threads that lives as long as the process (creating 16 threads per call cost more than the work they did).
// Jobs are indices 0 .. n - 1 handed out by an atomic counter; the caller works too and returns when all are done.
namespace {
struct HostPool {
    std::mutex mu;
    std::condition_variable wake, done;
    std::vector<std::thread> threads;
    const std::function<void(int)> *fn = nullptr;
    std::atomic<int> next{0};
    int n = 0, generation = 0, active = 0;
    pid_t owner = 0;                          // a forked child has none of the threads: it builds its own pool
};
HostPool *g_pool = nullptr;                   // never destroyed: its threads may outlive static destruction
std::mutex g_pool_call;                       // one batch at a time

void host_pool_worker(HostPool *p) {
    int seen = 0;
    for (;;) {
        const std::function<void(int)> *fn;
        int n;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->wake.wait(lk, [&] { return p->generation != seen; });
            seen = p->generation;
            fn = p->fn; n = p->n;
        }
        for (int i; (i = p->next.fetch_add(1)) < n;) (*fn)(i);
        {
            std::lock_guard<std::mutex> lk(p->mu);
            if (--p->active == 0) p->done.notify_one();
        }
    }
}

void host_pool_run(int n, int want_threads, const std::function<void(int)> &fn) {
    const int hw = (int)std::max(1u, std::thread::hardware_concurrency());
    const int helpers = std::min(std::min(want_threads, hw) - 1, n - 1);
    if (helpers <= 0) { for (int i = 0; i < n; ++i) fn(i); return; }
    std::lock_guard<std::mutex> call(g_pool_call);
    if (!g_pool || g_pool->owner != getpid()) { g_pool = new HostPool(); g_pool->owner = getpid(); }
    HostPool *p = g_pool;
    while ((int)p->threads.size() < helpers) { p->threads.emplace_back(host_pool_worker, p); p->threads.back().detach(); }
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->fn = &fn; p->n = n; p->next.store(0); p->active = (int)p->threads.size(); ++p->generation;
    }
    p->wake.notify_all();
    for (int i; (i = p->next.fetch_add(1)) < n;) fn(i);
    std::unique_lock<std::mutex> lk(p->mu);
    p->done.wait(lk, [&] { return p->active == 0; });
}
}  // namespace

extern "C" {

// The per-message part of AutoInterpretation.estimate after get_plateau_lengths (AutoInterpretation.py:416-433) for every message:
// lens[|off[m]| .. |off[m + 1]|) are message m's plateau lengths (off as urhgpu_msg_plateaus returns it).  tol_out[m]: the estimated
// tolerance, -1 = None; bitlen_out[m]: the bit length, -1 = fewer than two merged plateaus (the message does not vote), -2 = the
// reference's result depends on numpy's ordering of equal histogram counts (decide that message with numpy).  Host arithmetic.
int urhgpu_msg_bit_lengths(const uint64_t *lens, const int64_t *off, int n_msgs, int64_t *tol_out, int64_t *bitlen_out) {
    if (n_msgs < 0 || (n_msgs > 0 && (!off || !tol_out || !bitlen_out))) return URHGPU_ERR_ARG;
    for (int m = 0; m < n_msgs; ++m) {
        const int64_t a = off[m] < 0 ? -off[m] - 1 : off[m], b = off[m + 1] < 0 ? -off[m + 1] - 1 : off[m + 1];
        if (b < a || (b > a && !lens)) return URHGPU_ERR_ARG;
    }
    auto one = [&](int m) {
        const int64_t a = off[m] < 0 ? -off[m] - 1 : off[m], b = off[m + 1] < 0 ? -off[m + 1] - 1 : off[m + 1];
        std::vector<uint64_t> merged;
        if (!merged_lengths(lens + a, b - a, &tol_out[m], merged)) { bitlen_out[m] = -2; return; }
        bitlen_out[m] = merged.size() < 2 ? -1 : bit_length_of(merged);
    };
    // the messages are independent: a few host threads when there are many of them (a sort of a few thousand values each)
    host_pool_run(n_msgs, n_msgs >= 16 ? 24 : 1, one);
    return URHGPU_OK;
}

// urhgpu_msg_plateaus + urhgpu_msg_bit_lengths in one call that moves (value, count) pairs instead of plateau sequences
// (k_me_len_counts): tol_out / bitlen_out as urhgpu_msg_bit_lengths gives them, -3 in both for a message whose search window did not
// reach the percentage mark (the caller takes that message through urhgpu_msg_plateaus with a larger window).  Messages whose
// tolerance is positive (glitches: merge_plateaus walks the sequence) or whose lengths overflow the table are decided from their
// sequences, fetched in a second copy -- the same arithmetic as urhgpu_msg_bit_lengths.
static int plateau_decisions_impl(urhgpu_ctx *ctx, const float *d_x, int64_t n, const int64_t *ranges, const double *centers, int n_msgs,
                                  int percentage, int64_t extra_window, int64_t *tol_out, int64_t *bitlen_out, EstChain *chain) {
    if (!ctx || n < 0 || n_msgs < 0 || percentage < 0 || extra_window < 0 || (n_msgs > 0 && (!ranges || (!centers && !chain) || !tol_out || !bitlen_out || !d_x)))
        return URHGPU_ERR_ARG;
    if (n_msgs == 0) return URHGPU_OK;
    URH_HIP(hipSetDevice(ctx->device));
    MsgBatch local;
    MsgBatch &b = chain ? chain->b2 : local;
    if (!chain) {
        URH_TRY(plateau_batch(ctx, ranges, n_msgs, n, percentage, extra_window, b));
        for (int m = 0; m < n_msgs; ++m) b.host[(size_t)m].center = centers[m];
        URH_TRY(join_tail(ctx));
    }
    const int64_t cap_pairs = chain ? chain->cap_pairs : plateau_cap_pairs(n_msgs);
    const size_t need = (size_t)n_msgs * sizeof(MsgState) + (size_t)(b.n_tiles + 1) * (sizeof(MsgTile) + 4 + 8) + (size_t)std::max<int64_t>(n, 1) * 4 +
                        (size_t)cap_pairs * 16 + 10 * 256;
    if (!chain) {                                            // (a chained call: behind stage 1's scratch, in the reservation made for both)
        URH_TRY(ctx->arena.reserve(need));
        ctx->arena.reset();
    }
    MsgState *d_st = chain ? chain->d_st2 : (MsgState *)ctx->arena.take((size_t)n_msgs * sizeof(MsgState));
    MsgTile *d_tiles = (MsgTile *)ctx->arena.take((size_t)b.n_tiles * sizeof(MsgTile));
    int32_t *d_cnt = (int32_t *)ctx->arena.take((size_t)b.n_tiles * 4);
    int64_t *d_pre = (int64_t *)ctx->arena.take((size_t)(b.n_tiles + 1) * 8);
    int32_t *d_edges = (int32_t *)ctx->arena.take((size_t)std::max<int64_t>(n, 1) * 4);
    unsigned long long *d_pool_count = chain ? chain->d_pool_count : (unsigned long long *)ctx->arena.take(64);
    uint64_t *d_pool = chain ? chain->d_pool : (uint64_t *)ctx->arena.take((size_t)cap_pairs * 16);
    if (!d_st || !d_tiles || !d_cnt || !d_pre || !d_edges || !d_pool_count || !d_pool) return URHGPU_ERR_ARG;
    hipStream_t s = ctx->stream;
    if (!chain) URH_HIP(hipMemcpyAsync(d_st, b.host.data(), (size_t)n_msgs * sizeof(MsgState), hipMemcpyHostToDevice, s));
    if (chain) hipLaunchKernelGGL(k_me_chain_center, dim3((unsigned)((n_msgs + 63) / 64)), dim3(64), 0, s, chain->d_st1, d_st, n_msgs);
    hipLaunchKernelGGL(k_me_fill_tiles, dim3((unsigned)((b.n_tiles + 255) / 256)), dim3(256), 0, s, d_st, n_msgs, d_tiles, b.n_tiles);
    URH_HIP(hipMemsetAsync(d_pool_count, 0, 8, s));
    const unsigned gt = (unsigned)b.n_tiles;
    hipLaunchKernelGGL(k_me_edge_count, dim3(gt), dim3(kMeBlock), 0, s, d_x, d_st, d_tiles, d_cnt);
    hipLaunchKernelGGL(k_me_tile_scan, dim3(1), dim3(kMeScanBlock), 0, s, d_cnt, b.n_tiles, d_pre, (MsgState *)nullptr, 0, (unsigned int *)nullptr);
    hipLaunchKernelGGL(k_me_edge_compact, dim3(gt), dim3(kMeBlock), 0, s, d_x, d_st, d_tiles, d_pre, d_edges);
    hipLaunchKernelGGL(k_me_plateaus, dim3((unsigned)n_msgs), dim3(64), 0, s, d_st, d_edges, percentage);
    hipLaunchKernelGGL(k_me_len_counts, dim3((unsigned)n_msgs), dim3(kMeBlock), 0, s, d_st, d_edges, d_pool_count, d_pool, cap_pairs,
                       chain ? (const MsgState *)chain->d_st1 : (const MsgState *)nullptr, chain ? chain->d_rec : (EstRec *)nullptr);
    URH_HIP(hipGetLastError());
    // states (a chained call: records), pool fill and (speculatively) the first pairs of the pool land in the context's pinned zone in ONE
    // round trip when they fit
    struct PlateauRes { int64_t n_plateaus, pairs_base, pairs_n; };
    std::vector<PlateauRes> res((size_t)n_msgs);
    unsigned long long pool_used = 0;
    const char *l_pairs = nullptr;
    int64_t spec_pairs = 0;
    if (chain) {
        const size_t head = chain->rec_pad + 256;
        const bool pinned = ctx->h_small && head + 4096 <= kSmallPinned;
        spec_pairs = std::min<int64_t>(cap_pairs, 8192);
        if (pinned) spec_pairs = std::min<int64_t>(spec_pairs, (int64_t)((kSmallPinned - head) / 16));
        else chain->land.resize(head + (size_t)spec_pairs * 16);
        char *land = pinned ? ctx->h_small : chain->land.data();
        URH_HIP(hipMemcpyAsync(land, chain->d_rec, head + (size_t)spec_pairs * 16, hipMemcpyDeviceToHost, s));
        URH_HIP(wait_stream(ctx, s));
        chain->landed = land;
        const EstRec *rec = (const EstRec *)land;
        for (int m = 0; m < n_msgs; ++m) res[(size_t)m] = PlateauRes{rec[m].n_plateaus, rec[m].pairs_base, rec[m].pairs_n};
        memcpy(&pool_used, land + chain->rec_pad, 8);
        l_pairs = land + head;
    } else {
        const size_t st_bytes = (size_t)n_msgs * sizeof(MsgState), st_pad = (st_bytes + 255) & ~size_t(255);
        const bool pinned = ctx->h_small && st_pad + 256 + 4096 <= kSmallPinned;
        char *zone = pinned ? ctx->h_small : nullptr;
        spec_pairs = pinned ? std::min<int64_t>(std::min<int64_t>(cap_pairs, 8192), (int64_t)((kSmallPinned - st_pad - 256) / 16)) : 0;
        URH_HIP(hipMemcpyAsync(pinned ? (void *)zone : (void *)b.host.data(), d_st, st_bytes, hipMemcpyDeviceToHost, s));
        URH_HIP(hipMemcpyAsync(pinned ? (void *)(zone + st_pad) : (void *)&pool_used, d_pool_count, 8, hipMemcpyDeviceToHost, s));
        if (spec_pairs > 0) URH_HIP(hipMemcpyAsync(zone + st_pad + 256, d_pool, (size_t)spec_pairs * 16, hipMemcpyDeviceToHost, s));
        URH_HIP(wait_stream(ctx, s));
        if (pinned) { memcpy(b.host.data(), zone, st_bytes); memcpy(&pool_used, zone + st_pad, 8); l_pairs = zone + st_pad + 256; }
        for (int m = 0; m < n_msgs; ++m) res[(size_t)m] = PlateauRes{b.host[(size_t)m].n_plateaus, b.host[(size_t)m].pairs_base, b.host[(size_t)m].pairs_n};
    }
    std::vector<uint64_t> pool((size_t)std::min<unsigned long long>(pool_used, (unsigned long long)cap_pairs) * 2);
    if (!pool.empty()) {
        const size_t have = std::min<size_t>(pool.size() / 2, (size_t)spec_pairs);
        if (have > 0) memcpy(pool.data(), l_pairs, have * 16);
        if (pool.size() / 2 > have) {
            URH_HIP(hipMemcpyAsync(pool.data() + 2 * have, d_pool + 2 * have, (pool.size() / 2 - have) * 16, hipMemcpyDeviceToHost, s));
            URH_HIP(wait_stream(ctx, s));
        }
    }
    // messages decided from their multiset; the rest need their sequences
    std::vector<int64_t> begin((size_t)n_msgs, -1);
    int64_t total = 0;
    for (int m = 0; m < n_msgs; ++m) {
        const PlateauRes &st = res[(size_t)m];
        const int64_t k = st.n_plateaus;
        if (k < 0) { tol_out[m] = -3; bitlen_out[m] = -3; continue; }
        bool need_seq = st.pairs_n < 0;
        if (!need_seq) {
            std::vector<std::pair<uint64_t, uint64_t>> vc((size_t)st.pairs_n);
            for (int64_t i = 0; i < st.pairs_n; ++i) vc[(size_t)i] = {pool[2 * (size_t)(st.pairs_base + i)], pool[2 * (size_t)(st.pairs_base + i) + 1]};
            std::sort(vc.begin(), vc.end());
            std::vector<uint64_t> u(vc.size());
            for (size_t i = 0; i < vc.size(); ++i) u[i] = vc[i].first;
            const int64_t tol = tolerance_of_unique(u, (size_t)k);
            if (tol > 0) need_seq = true;                      // glitches: merge_plateaus needs the order
            else {
                tol_out[m] = tol;
                bitlen_out[m] = (tol == -2) ? -2 : (k < 2 ? -1 : bit_length_of_counts(vc));
            }
        }
        if (need_seq) { begin[(size_t)m] = total; total += k; }
    }
    if (total > 0) {
        URH_TRY(ctx->staging.reserve((size_t)n_msgs * 8 + (size_t)total * 8 + 1024));
        ctx->staging.reset();
        int64_t *d_begin = (int64_t *)ctx->staging.take((size_t)n_msgs * 8);
        uint64_t *d_out = (uint64_t *)ctx->staging.take((size_t)total * 8);
        if (!d_begin || !d_out) return URHGPU_ERR_ARG;
        std::vector<uint64_t> lens((size_t)total);
        URH_HIP(hipMemcpyAsync(d_begin, begin.data(), (size_t)n_msgs * 8, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_me_gather_some, dim3((unsigned)n_msgs), dim3(64), 0, s, d_st, d_edges, d_begin, d_out);
        URH_HIP(hipGetLastError());
        URH_HIP(hipMemcpyAsync(lens.data(), d_out, (size_t)total * 8, hipMemcpyDeviceToHost, s));
        URH_HIP(wait_stream(ctx, s));
        std::vector<int> todo;
        for (int m = 0; m < n_msgs; ++m) if (begin[(size_t)m] >= 0) todo.push_back(m);
        auto one = [&](int j) {
            const int m = todo[(size_t)j];
            std::vector<uint64_t> merged;
            if (!merged_lengths(lens.data() + begin[(size_t)m], res[(size_t)m].n_plateaus, &tol_out[m], merged)) { bitlen_out[m] = -2; return; }
            bitlen_out[m] = merged.size() < 2 ? -1 : bit_length_of(merged);
        };
        host_pool_run((int)todo.size(), todo.size() >= 16 ? 24 : 1, one);
    }
    return URHGPU_OK;
}

int urhgpu_msg_plateau_decisions(urhgpu_ctx *ctx, const float *d_x, int64_t n, const int64_t *ranges, const double *centers, int n_msgs,
                                 int percentage, int64_t extra_window, int64_t *tol_out, int64_t *bitlen_out) {
    if (!centers) return URHGPU_ERR_ARG;
    return plateau_decisions_impl(ctx, d_x, n, ranges, centers, n_msgs, percentage, extra_window, tol_out, bitlen_out, nullptr);
}

// urhgpu_msg_center_stats followed by urhgpu_msg_plateau_decisions with the picked centers, in ONE native call: no host round trip between
// the stages (AutoInterpretation.py:397-433 per message: detect_center, get_plateau_lengths(center), tolerance / merge / bit length).  A
// message whose center needs the host (peak_flag 2: more bins than the pool holds; 3: a tie np.argsort decides) comes back with its flag
// and tol_out = bitlen_out = -4: the caller settles its center and repeats the second stage for it.  URHGPU_ERR_UNSUPPORTED: the
// messages' histogram pool would not fit one batch (take the two calls).
int urhgpu_msg_estimate(urhgpu_ctx *ctx, const float *d_x, int64_t n, const int64_t *ranges, int n_msgs, int64_t max_bins, int percentage,
                        int64_t extra_window, double *out_stats, double *out_center, int32_t *out_flag, int64_t *tol_out, int64_t *bitlen_out) {
    if (!ctx || n < 0 || n_msgs < 0 || max_bins < 1 || percentage < 0 || extra_window < 0 ||
        (n_msgs > 0 && (!ranges || !out_stats || !out_center || !out_flag || !tol_out || !bitlen_out || !d_x)))
        return URHGPU_ERR_ARG;
    if (n_msgs == 0) return URHGPU_OK;
    if ((size_t)n_msgs * (size_t)max_bins * 4 > kCenterBatchBytes) return URHGPU_ERR_UNSUPPORTED;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(join_tail(ctx));
    // one reservation for both stages (a reallocation between them would take stage 1's states away): bounds of their tile tables
    const size_t tiles_max = (size_t)(n / kMeTile + n_msgs + 2);
    const int64_t cap_pairs = std::max<int64_t>(4096, (int64_t)n_msgs * 256);
    const size_t need1 = (size_t)n_msgs * sizeof(MsgState) + (tiles_max + 1) * (sizeof(MsgTile) + 4 + 8 + 8 + 4 + kLeavesPerTile * 4) + (size_t)n * 4 +
                         (size_t)n_msgs * (size_t)max_bins * 4 + (size_t)n_msgs * 4 + 18 * 256;
    const size_t need2 = (size_t)n_msgs * sizeof(MsgState) + (tiles_max + 1) * (sizeof(MsgTile) + 4 + 8) + (size_t)std::max<int64_t>(n, 1) * 4 + (size_t)cap_pairs * 16 + 10 * 256;
    URH_TRY(ctx->arena.reserve(need1 + need2 + (size_t)n_msgs * (sizeof(EstRec) + 16) + 8192));
    ctx->arena.reset();
    EstChain chain;
    // the host's part of the batches: the ranges checked (build_batch's rules), the tile counts of both stages
    int64_t nt1 = 0, nt2 = 0;
    for (int m = 0; m < n_msgs; ++m) {
        const int64_t s0 = ranges[2 * m], e0 = ranges[2 * m + 1], len = e0 - s0;
        if (s0 < 0 || e0 < s0 || e0 > n || (m > 0 && s0 < ranges[2 * m - 1])) return URHGPU_ERR_ARG;
        if (len > INT32_MAX) return URHGPU_ERR_UNSUPPORTED;       // positions inside a message are 32-bit
        const int64_t limit = ((int64_t)percentage * len) / 100, window = std::min<int64_t>(len, limit + extra_window);
        nt1 += (std::max<int64_t>(len, 1) + kMeTile - 1) / kMeTile;
        nt2 += (std::max<int64_t>(window, 1) + kMeTile - 1) / kMeTile;
    }
    chain.b1.n_tiles = nt1; chain.b2.n_tiles = nt2;
    chain.cap_pairs = cap_pairs;
    chain.rec_pad = ((size_t)n_msgs * sizeof(EstRec) + 255) & ~size_t(255);
    char *blk = (char *)ctx->arena.take(chain.rec_pad + 256 + (size_t)cap_pairs * 16);
    chain.d_st1 = (MsgState *)ctx->arena.take((size_t)n_msgs * sizeof(MsgState));
    chain.d_st2 = (MsgState *)ctx->arena.take((size_t)n_msgs * sizeof(MsgState));
    int64_t *d_ranges = (int64_t *)ctx->arena.take((size_t)n_msgs * 16);
    if (!blk || !chain.d_st1 || !chain.d_st2 || !d_ranges) return URHGPU_ERR_ARG;
    chain.d_rec = (EstRec *)blk;
    chain.d_pool_count = (unsigned long long *)(blk + chain.rec_pad); chain.d_pool = (uint64_t *)(blk + chain.rec_pad + 256);
    const void *up = ranges;
    if (ctx->h_small && (size_t)n_msgs * 16 <= kSmallPinned) { memcpy(ctx->h_small, ranges, (size_t)n_msgs * 16); up = ctx->h_small; }     // (a truly asynchronous copy)
    URH_HIP(hipMemcpyAsync(d_ranges, up, (size_t)n_msgs * 16, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_me_init_states, dim3(1), dim3(kMeInitBlock), 0, ctx->stream, d_ranges, n_msgs, percentage, extra_window, chain.d_st1, chain.d_st2);
    URH_TRY(center_stats_batch(ctx, d_x, n, ranges, n_msgs, max_bins, out_stats, nullptr, out_center, out_flag, &chain));
    URH_TRY(plateau_decisions_impl(ctx, d_x, n, ranges, nullptr, n_msgs, percentage, extra_window, tol_out, bitlen_out, &chain));
    const EstRec *rec = (const EstRec *)chain.landed;
    for (int m = 0; m < n_msgs; ++m) {
        memcpy(out_stats + 8 * (size_t)m, rec[m].stats, 64);
        out_center[m] = rec[m].center;
        out_flag[m] = (int32_t)rec[m].flag;
    }
    for (int m = 0; m < n_msgs; ++m)
        if (out_flag[m] == 2 || out_flag[m] == 3) { tol_out[m] = -4; bitlen_out[m] = -4; }
    return URHGPU_OK;
}

// Test hook (host arithmetic): the multiset form of the decision on ONE message's plateau lengths, as urhgpu_msg_plateau_decisions
// takes it from the device's (value, count) pairs; *tol_out = *bitlen_out = -3 where that call would ask for the sequence.
int urhgpu_test_bit_length_from_counts(const uint64_t *lens, int64_t n, int64_t *tol_out, int64_t *bitlen_out) {
    if (n < 0 || !tol_out || !bitlen_out || (n > 0 && !lens)) return URHGPU_ERR_ARG;
    std::vector<uint64_t> v(lens, lens + n);
    std::sort(v.begin(), v.end());
    std::vector<std::pair<uint64_t, uint64_t>> vc;
    for (size_t i = 0; i < v.size();) { size_t j = i; while (j < v.size() && v[j] == v[i]) ++j; vc.push_back({v[i], (uint64_t)(j - i)}); i = j; }
    std::vector<uint64_t> u(vc.size());
    for (size_t i = 0; i < vc.size(); ++i) u[i] = vc[i].first;
    const int64_t tol = tolerance_of_unique(u, (size_t)n);
    if (tol > 0) { *tol_out = -3; *bitlen_out = -3; return URHGPU_OK; }
    *tol_out = tol;
    *bitlen_out = (tol == -2) ? -2 : (n < 2 ? -1 : bit_length_of_counts(vc));
    return URHGPU_OK;
}

// The two halves around np.argsort for a message whose bit length urhgpu_msg_bit_lengths could not decide (-2: equal counts in the
// divisor histogram, where the reference's result is whatever order np.argsort gives equal keys).
//   urhgpu_msg_divisor_histogram: tolerance, merged and rounded plateaus, then the DENSE histogram the reference sorts
//     (uint64[max value + 1], auto_interpretation.pyx:113-143).  *hist_len = its length (call with cap = 0 to ask), -1 = fewer than
//     two merged plateaus (no histogram in the reference either).
//   urhgpu_bit_length_from_order: the selection loop of get_bit_length_from_plateau_lengths (AutoInterpretation.py:358-370) for the
//     caller's order (= np.argsort(hist)[::-1]).
int urhgpu_msg_divisor_histogram(const uint64_t *lens, int64_t n, uint64_t *hist_out, int64_t cap, int64_t *hist_len, int64_t *tol_out) {
    if (n < 0 || !hist_len || !tol_out || (n > 0 && !lens) || cap < 0 || (cap > 0 && !hist_out)) return URHGPU_ERR_ARG;
    std::vector<uint64_t> merged;
    if (!merged_lengths(lens, n, tol_out, merged)) return URHGPU_ERR_UNSUPPORTED;
    if (merged.size() < 2) { *hist_len = -1; return URHGPU_OK; }
    round_lengths(merged);
    const uint64_t mx = *std::max_element(merged.begin(), merged.end());
    *hist_len = (int64_t)mx + 1;
    if (cap < *hist_len) return cap == 0 ? URHGPU_OK : URHGPU_ERR_CAPACITY;
    std::vector<std::pair<uint64_t, uint64_t>> hist;
    divisor_histogram(merged, hist);
    std::fill(hist_out, hist_out + *hist_len, 0ull);
    for (const auto &e : hist) hist_out[e.second] = e.first;
    return URHGPU_OK;
}

// auto_interpretation.get_threshold_divisor_histogram (auto_interpretation.pyx:113-143) itself: the DENSE histogram uint64[max + 1]
// (the P^2 / 2 pair test evaluated on the multiset of values).  cap = 0 only asks for *hist_len.  Host arithmetic.
int urhgpu_threshold_divisor_histogram(const uint64_t *lens, int64_t n, float threshold, uint64_t *hist_out, int64_t cap, int64_t *hist_len) {
    if (n <= 0 || !lens || !hist_len || cap < 0 || (cap > 0 && !hist_out)) return URHGPU_ERR_ARG;       // np.max of an empty array raises in the reference
    const uint64_t mx = *std::max_element(lens, lens + n);
    if (mx >= (uint64_t)1 << 40) return URHGPU_ERR_UNSUPPORTED;
    *hist_len = (int64_t)mx + 1;
    if (cap < *hist_len) return cap == 0 ? URHGPU_OK : URHGPU_ERR_CAPACITY;
    std::vector<uint64_t> m(lens, lens + n);
    std::vector<std::pair<uint64_t, uint64_t>> hist;
    divisor_histogram(m, hist, threshold);
    std::fill(hist_out, hist_out + *hist_len, 0ull);
    for (const auto &e : hist) hist_out[e.second] = e.first;
    return URHGPU_OK;
}

int urhgpu_bit_length_from_order(const uint64_t *hist, const int64_t *order_desc, int64_t len, int64_t *bitlen_out) {
    if (len < 0 || !bitlen_out || (len > 0 && (!hist || !order_desc))) return URHGPU_ERR_ARG;
    if (len == 0) { *bitlen_out = 0; return URHGPU_OK; }
    for (int64_t i = 0; i < len; ++i) if (order_desc[i] < 0 || order_desc[i] >= len) return URHGPU_ERR_ARG;
    const uint64_t max_count = hist[order_desc[0]];
    int64_t result = order_desc[0];
    for (int64_t i = 1; i < len; ++i) {
        if ((double)hist[order_desc[i]] < 0.25 * (double)max_count) break;
        if ((double)order_desc[i] <= 0.5 * (double)result) result = order_desc[i];
    }
    *bitlen_out = result;
    return URHGPU_OK;
}

}  // extern "C"
