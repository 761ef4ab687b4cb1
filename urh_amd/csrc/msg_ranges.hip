// msg_ranges.hip -- from the above / below-noise state table to message ranges, on the device
//   auto_interpretation.segment_messages_from_magnitudes  /root/reference/src/urh/cythonext/auto_interpretation.pyx:55-111
//   AutoInterpretation.merge_message_segments_for_ook     /root/reference/src/urh/ainterpretation/AutoInterpretation.py:107-148
//
// The hot kernel in seg_mode leaves one row per state change (state before the change, length) -- for an OOK capture one row
// per pulse edge: 1.4 M rows per GiB.  Reading them back and pairing them up in numpy cost 11 ms of a 22 ms estimate; here the
// pairing is index arithmetic on a prefix sum, the OOK merge three reductions and one compaction, and only the message ranges
// (a hundred pairs) cross PCIe.
//   change k >= 1 happens at pos_k = len_0 + ... + len_{k-1} - 1; a change to "above" opens a segment at pos_k - 1 (:101-104), a
//   change to "below" closes the open one at pos_k - 1 (:95-99); states alternate, so with s0 = state of sample 0:
//   s0 above: segment 0 opens at 0, open_j at change 2j, close_j at change 2j + 1;  s0 below: open_j at 2j + 1, close_j at 2j + 2.
//   A capture that ends above the noise closes its last segment where its trailing below-run starts (:107-109).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "common.hpp"
#include "launchers.hpp"
#include "scan.hpp"

namespace urh {

bool g_force_merge_ambiguous = false;        // test hook (urhgpu_test_force_merge_ambiguous): report every OOK merge as borderline

struct SegCtl {              // device control block
    int64_t n_seg;           // complete segments
    int64_t n_msgs;          // after the OOK merge (or n_seg)
    double sum, sq;          // reductions over the pulse lengths
    unsigned long long min_pulse;
    int ambiguous;           // a pulse sits within rounding of the outlier bound: the host repeats the decision in numpy's summation order
    int pad;
    unsigned int tickets[2]; // k_seg_moments, per pass: workgroups that have delivered their partial sum (the last one adds them up)
};

struct SegLoad {
    const int64_t *rows;
    __device__ VecK<1> operator()(int64_t k) const { VecK<1> v; v.v[0] = rows[2 * k + 1]; return v; }
};
struct SegStore {
    const int64_t *rows;
    int64_t *seg;            // [cap][2]
    int64_t cap;
    __device__ void operator()(int64_t k, const VecK<1> &, const VecK<1> &ex) const {
        const int s0 = (int)rows[0];
        if (k == 0) { if (s0 == 1 && cap > 0) seg[0] = 0; return; }
        const int64_t pos = ex.v[0] - 1;
        const bool to_above = rows[2 * k] == 1;
        int64_t j;
        if (to_above) j = (s0 == 1) ? k / 2 : (k - 1) / 2;
        else j = (s0 == 1) ? (k - 1) / 2 : (k - 2) / 2;
        if (j < cap) seg[2 * j + (to_above ? 0 : 1)] = pos - 1;
    }
};

// magnitude of sample i compared with the threshold as segment_messages_from_magnitudes does (double magnitude > float threshold)
template <int DT> __device__ bool seg_above(const void *iq, int64_t i, float thr);
template <> __device__ bool seg_above<URHGPU_DT_F32>(const void *iq, int64_t i, float thr) {
    const float2 v = ((const float2 *)iq)[i];
    return (double)__builtin_sqrtf(v.x * v.x + v.y * v.y) > (double)thr;
}
__device__ __forceinline__ bool seg_above_int(int re, int im, float thr) {
    const int s = (int)((unsigned)(re * re) + (unsigned)(im * im));
    return __builtin_sqrt((double)s) > (double)thr;
}
template <> __device__ bool seg_above<URHGPU_DT_I8>(const void *iq, int64_t i, float thr) { const char2 v = ((const char2 *)iq)[i]; return seg_above_int(v.x, v.y, thr); }
template <> __device__ bool seg_above<URHGPU_DT_U8>(const void *iq, int64_t i, float thr) { const uchar2 v = ((const uchar2 *)iq)[i]; return seg_above_int(v.x, v.y, thr); }
template <> __device__ bool seg_above<URHGPU_DT_I16>(const void *iq, int64_t i, float thr) { const short2 v = ((const short2 *)iq)[i]; return seg_above_int(v.x, v.y, thr); }
template <> __device__ bool seg_above<URHGPU_DT_U16>(const void *iq, int64_t i, float thr) {
    const ushort2 v = ((const ushort2 *)iq)[i];
    const int s = (int)((unsigned)v.x * (unsigned)v.x + (unsigned)v.y * (unsigned)v.y);
    return __builtin_sqrt((double)s) > (double)thr;
}

template <> __device__ bool seg_above<kDtAboveFlags>(const void *iq, int64_t i, float) { return ((const float *)iq)[i] > 0.5f; }

// number of complete segments; the trailing one of a capture that ends above the noise
template <int DT>
__global__ void k_seg_finish(const int64_t *rows, const int64_t *d_n_rows, const void *iq, int64_t n, float thr, int64_t *seg, int64_t cap,
                             SegCtl *ctl) {
    const int64_t n_rows = *d_n_rows;
    int64_t n_seg = 0;
    if (n_rows > 0) {
        const int s0 = (int)rows[0];
        const int64_t n_changes = n_rows - 1;
        const int64_t n_closes = (s0 == 1) ? (n_changes + 1) / 2 : n_changes / 2;
        const int64_t n_opens = (s0 == 1) ? 1 + n_changes / 2 : (n_changes + 1) / 2;
        n_seg = n_closes;
        if (n_opens > n_closes) {                             // still above at the end
            int64_t below = 0;
            for (int64_t i = n - 1; i >= 0 && i >= n - 10; --i) { if (seg_above<DT>(iq, i, thr)) break; ++below; }
            const int64_t start = (n_closes < cap) ? seg[2 * n_closes] : 0;
            if (start < n - below) { if (n_closes < cap) seg[2 * n_closes + 1] = n - below; n_seg = n_closes + 1; }
        }
    }
    ctl->n_seg = n_seg; ctl->n_msgs = n_seg; ctl->ambiguous = 0; ctl->sum = 0.0; ctl->sq = 0.0; ctl->min_pulse = ~0ull;
    ctl->tickets[0] = 0u; ctl->tickets[1] = 0u;
}

// ---- OOK merge: outlier-free minimum pulse, cuts at pauses >= 8 x that ---------------------------------------------------
// mean / standard deviation of the pulse lengths by two-level sums (fixed order); min over |p - mean| <= std.  The workgroup that
// delivers the LAST partial sum of a pass adds the partial sums up (in block order, as a tree: the same result whichever workgroup it is)
// -- rounds 3-6a had a one-workgroup kernel per pass for that, and one more that copied the segment count for the scan behind.
constexpr int kSegParts = 256;
__global__ __launch_bounds__(256) void k_seg_moments(const int64_t *seg, SegCtl *ctl, int pass, double *part) {
    __shared__ double s_p[4];
    __shared__ bool s_last;
    const int64_t n = ctl->n_seg;
    const double mean = (pass == 1 && n > 0) ? ctl->sum / (double)n : 0.0;
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = (lo + per < n) ? lo + per : n;
    double acc = 0.0;
    for (int64_t j = lo + threadIdx.x; j < hi; j += 256) {
        const double p = (double)(unsigned long long)(seg[2 * j + 1] - seg[2 * j]);
        acc += pass == 0 ? p : (p - mean) * (p - mean);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63) == 0) s_p[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(&part[blockIdx.x], (s_p[0] + s_p[1]) + (s_p[2] + s_p[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        s_last = atomicAdd(&ctl->tickets[pass], 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    double v = (threadIdx.x < gridDim.x) ? __hip_atomic_load(&part[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    __syncthreads();                                          // (s_p has been read)
    if ((threadIdx.x & 63) == 0) s_p[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) { const double t = (s_p[0] + s_p[1]) + (s_p[2] + s_p[3]); if (pass == 0) ctl->sum = t; else ctl->sq = t; }
}
__global__ __launch_bounds__(256) void k_seg_min_pulse(const int64_t *seg, SegCtl *ctl, int64_t *d_n) {
    __shared__ unsigned long long s_b[4];
    const int64_t n = ctl->n_seg;
    if (blockIdx.x == 0 && threadIdx.x == 0) *d_n = n;      // the segment count where the scan behind this kernel reads its length
    if (n <= 1) return;
    const double mean = ctl->sum / (double)n, sd = sqrt(ctl->sq / (double)n);
    unsigned long long best = ~0ull;
    for (int64_t j = blockIdx.x * 256ll + threadIdx.x; j < n; j += (int64_t)gridDim.x * 256) {
        const unsigned long long p = (unsigned long long)(seg[2 * j + 1] - seg[2 * j]);
        const double dev = fabs((double)p - mean);
        if (dev <= 1.0 * sd && p < best) best = p;           // min_without_outliers(pulses, z=1) (AutoInterpretation.py:21-25)
        // the sum of the pulses is exact in any order (integers below 2^53), the sum of the squared deviations is not: numpy's pairwise
        // order and the two-level order here agree to ~1e-14; a pulse closer than that to the bound is not decided here
        if (sd > 0.0 && fabs(dev - sd) <= 1e-9 * sd) ctl->ambiguous = 1;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long u = (unsigned long long)__shfl_down((long long)best, o); if (u < best) best = u; }
    if ((threadIdx.x & 63) == 0) s_b[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) if (s_b[w] < best) best = s_b[w];
        // one atomic per workgroup, and only where it can lower the minimum (it only ever falls: a stale read costs an add, never a miss)
        if (best != ~0ull && best < __hip_atomic_load(&ctl->min_pulse, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&ctl->min_pulse, best);
    }
}
// message i starts after the i-th long pause: flag[j] = pause after segment j is long (j < n_seg - 1); messages by a scan of the flags
struct CutLoad {
    const int64_t *seg; const SegCtl *ctl;
    __device__ VecK<1> operator()(int64_t j) const {
        VecK<1> v; v.v[0] = 0;
        if (j + 1 < ctl->n_seg) {
            const unsigned long long pause = (unsigned long long)(seg[2 * (j + 1)] - seg[2 * j + 1]);
            v.v[0] = (pause >= 8ull * ctl->min_pulse) ? 1 : 0;
        }
        return v;
    }
};
struct CutStore {
    const int64_t *seg; const SegCtl *ctl; int64_t *msgs; int64_t cap;
    __device__ void operator()(int64_t j, const VecK<1> &val, const VecK<1> &ex) const {
        const int64_t m = ex.v[0];                            // message segment j belongs to
        const bool first = (j == 0) || (CutLoad{seg, ctl}(j - 1).v[0] != 0);
        const bool last = (j + 1 == ctl->n_seg) || val.v[0] != 0;
        if (m < cap) {
            if (first) msgs[2 * m] = seg[2 * j];
            if (last) msgs[2 * m + 1] = seg[2 * j + 1];       // telescoped: a merged message ends where its last pulse ends
        }
    }
};
struct CutFinal {
    SegCtl *ctl;
    __device__ void operator()(const VecK<1> &grand) const { ctl->n_msgs = (ctl->n_seg > 0) ? grand.v[0] + 1 : 0; }
};

template <int DT>
static void launch_seg_finish(const int64_t *rows, const int64_t *d_n_rows, const void *iq, int64_t n, float thr, int64_t *seg, int64_t cap,
                              SegCtl *ctl, hipStream_t s) {
    hipLaunchKernelGGL(k_seg_finish<DT>, dim3(1), dim3(1), 0, s, rows, d_n_rows, iq, n, thr, seg, cap, ctl);
}

// rows (device, d_n_rows of them) -> message ranges in d_msgs (device, cap pairs), *ctl filled; scratch: see seg_scratch_bytes
int launch_message_ranges(const int64_t *d_rows, const int64_t *d_n_rows, int64_t cap_rows, const void *d_iq, int dtype, int64_t n, float thr,
                          int ook_merge, int64_t *d_seg, int64_t *d_msgs, int64_t cap, SegCtl *d_ctl, void *scratch, hipStream_t s) {
    const int64_t nb = std::max<int64_t>((cap_rows + kScanTile - 1) / kScanTile, 1);
    VecK<1> *part = (VecK<1> *)scratch;
    double *dpart = (double *)((char *)scratch + (((size_t)(nb + 2) * sizeof(VecK<1>) + 255) & ~size_t(255)));
    int64_t *d_nseg = (int64_t *)(dpart + 1024);
    SegLoad ld{d_rows};
    hipLaunchKernelGGL((k_scan_reduce<1, SegLoad>), dim3(scan_grid(nb)), dim3(kScanBlock), 0, s, d_n_rows, ld, part, nb);
    if (nb > kScanDirect) hipLaunchKernelGGL((k_scan_partials<1>), dim3(1), dim3(kScanPartialsBlock), 0, s, d_n_rows, part, nb, kScanDirect);
    hipLaunchKernelGGL((k_scan_apply<1, SegLoad, SegStore>), dim3(scan_grid(nb)), dim3(kScanBlock), 0, s, d_n_rows, ld, part, nb, SegStore{d_rows, d_seg, cap},
                       kScanDirect);
    switch (dtype) {
        case URHGPU_DT_F32: launch_seg_finish<URHGPU_DT_F32>(d_rows, d_n_rows, d_iq, n, thr, d_seg, cap, d_ctl, s); break;
        case URHGPU_DT_I8: launch_seg_finish<URHGPU_DT_I8>(d_rows, d_n_rows, d_iq, n, thr, d_seg, cap, d_ctl, s); break;
        case URHGPU_DT_U8: launch_seg_finish<URHGPU_DT_U8>(d_rows, d_n_rows, d_iq, n, thr, d_seg, cap, d_ctl, s); break;
        case URHGPU_DT_I16: launch_seg_finish<URHGPU_DT_I16>(d_rows, d_n_rows, d_iq, n, thr, d_seg, cap, d_ctl, s); break;
        case URHGPU_DT_U16: launch_seg_finish<URHGPU_DT_U16>(d_rows, d_n_rows, d_iq, n, thr, d_seg, cap, d_ctl, s); break;
        case kDtAboveFlags: launch_seg_finish<kDtAboveFlags>(d_rows, d_n_rows, d_iq, n, thr, d_seg, cap, d_ctl, s); break;
        default: return URHGPU_ERR_DTYPE;
    }
    if (!ook_merge) return URHGPU_OK;
    const int gp = kSegParts;
    for (int pass = 0; pass < 2; ++pass) hipLaunchKernelGGL(k_seg_moments, dim3(gp), dim3(256), 0, s, d_seg, d_ctl, pass, dpart + pass * kSegParts);
    hipLaunchKernelGGL(k_seg_min_pulse, dim3(gp), dim3(256), 0, s, d_seg, d_ctl, d_nseg);
    const int64_t nbs = std::max<int64_t>((cap + kScanTile - 1) / kScanTile, 1);
    VecK<1> *part2 = part;                                   // the first scan is done
    CutLoad cl{d_seg, d_ctl};
    hipLaunchKernelGGL((k_scan_reduce<1, CutLoad>), dim3(scan_grid(nbs)), dim3(kScanBlock), 0, s, d_nseg, cl, part2, nbs);
    if (nbs > kScanDirect) hipLaunchKernelGGL((k_scan_partials<1>), dim3(1), dim3(kScanPartialsBlock), 0, s, d_nseg, part2, nbs, kScanDirect);
    hipLaunchKernelGGL((k_scan_apply<1, CutLoad, CutStore>), dim3(scan_grid(nbs)), dim3(kScanBlock), 0, s, d_nseg, cl, part2, nbs,
                       CutStore{d_seg, d_ctl, d_msgs, cap}, kScanDirect);
    hipLaunchKernelGGL((k_scan_finish<1, CutFinal>), dim3(1), dim3(kScanBlock), 0, s, d_nseg, part2, nbs, CutFinal{d_ctl}, kScanDirect);
    return URHGPU_OK;
}

size_t seg_scratch_bytes(int64_t cap_rows, int64_t cap) {
    const int64_t nb = std::max<int64_t>((std::max(cap_rows, cap) + kScanTile - 1) / kScanTile, 1);
    return (((size_t)(nb + 2) * sizeof(VecK<1>) + 255) & ~size_t(255)) + 1024 * 8 + 256;
}
size_t seg_ctl_bytes() { return sizeof(SegCtl); }
void seg_ctl_read(const void *host_copy, int64_t *n_seg, int64_t *n_msgs, int *ambiguous) {
    const SegCtl *c = (const SegCtl *)host_copy;
    *n_seg = c->n_seg; *n_msgs = c->n_msgs; *ambiguous = c->ambiguous | (g_force_merge_ambiguous ? 1 : 0);
}

}  // namespace urh
