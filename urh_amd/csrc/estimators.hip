// estimators.hip -- the O(N) passes of the parameter estimators (rows 11-12 of SURVEY.md §8a) for gfx950.
//
//   detect_center          /root/reference/src/urh/ainterpretation/AutoInterpretation.py:226-277
//       rect[rect > -4] (stable compaction), 5 % trim, min / max, np.var (float32, numpy's pairwise summation),
//       np.histogram over explicit float64 bin edges
//   get_plateau_lengths    /root/reference/src/urh/cythonext/auto_interpretation.pyx:179-208
//       positions where (rect[i] <= center) changes
//
// The decisions on the few hundred histogram bins / plateau lengths stay on the host (urh_amd/estimators.py).
// All passes are HBM streaming (4 B/sample read, compaction: + <= 4 B/sample written); none is on the per-sample
// demodulation path, they run once per message when parameters are estimated.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "common.hpp"
#include "launchers.hpp"
#include "scan.hpp"

namespace urh {

// ---- stable compaction: values x[i] > thr, or indices i where (x[i] <= c) differs from (x[i-1] <= c) ----------
struct GtLoad {
    const float *x; float thr;
    __device__ VecK<1> operator()(int64_t i) const { VecK<1> v; v.v[0] = (x[i] > thr) ? 1 : 0; return v; }
};
struct GtStore {
    const float *x; float *out;
    __device__ void operator()(int64_t i, const VecK<1> &val, const VecK<1> &ex) const { if (val.v[0]) out[ex.v[0]] = x[i]; }
};
struct EdgeLoad {
    const float *x; float c;
    __device__ VecK<1> operator()(int64_t i) const {
        VecK<1> v; v.v[0] = (i > 0 && ((x[i] <= c) != (x[i - 1] <= c))) ? 1 : 0; return v;
    }
};
struct EdgeStore {
    int64_t *out; int64_t cap;
    __device__ void operator()(int64_t i, const VecK<1> &val, const VecK<1> &ex) const { if (val.v[0] && ex.v[0] < cap) out[ex.v[0]] = i; }
};
struct CountFinal {
    int64_t *d_count;
    __device__ void operator()(const VecK<1> &grand) const { *d_count = grand.v[0]; }
};

size_t compact_scratch_bytes(int64_t n) { return (size_t)((n + kScanTile - 1) / kScanTile + 2) * sizeof(VecK<1>) + 256; }

// d_n: device copy of n (the scan kernels take their element count from device memory)
int launch_compact_gt(const float *x, int64_t n, const int64_t *d_n, float thr, float *out, int64_t *d_count, void *scratch,
                      int32_t *tickets, hipStream_t s) {
    const int64_t nb = std::max<int64_t>((n + kScanTile - 1) / kScanTile, 1);
    VecK<1> *part = (VecK<1> *)scratch;
    GtLoad ld{x, thr};
    hipLaunchKernelGGL((k_scan_reduce<1, GtLoad>), dim3((unsigned)nb), dim3(kScanBlock), 0, s, d_n, ld, part, nb, tickets);
    hipLaunchKernelGGL((k_scan_apply<1, GtLoad, GtStore, CountFinal>), dim3((unsigned)nb), dim3(kScanBlock), 0, s, d_n, ld, part, nb,
                       GtStore{x, out}, CountFinal{d_count}, tickets + 1);
    return URHGPU_OK;
}

int launch_compact_edges(const float *x, int64_t n, const int64_t *d_n, float center, int64_t *out, int64_t cap, int64_t *d_count,
                         void *scratch, int32_t *tickets, hipStream_t s) {
    const int64_t nb = std::max<int64_t>((n + kScanTile - 1) / kScanTile, 1);
    VecK<1> *part = (VecK<1> *)scratch;
    EdgeLoad ld{x, center};
    hipLaunchKernelGGL((k_scan_reduce<1, EdgeLoad>), dim3((unsigned)nb), dim3(kScanBlock), 0, s, d_n, ld, part, nb, tickets);
    hipLaunchKernelGGL((k_scan_apply<1, EdgeLoad, EdgeStore, CountFinal>), dim3((unsigned)nb), dim3(kScanBlock), 0, s, d_n, ld, part, nb,
                       EdgeStore{out, cap}, CountFinal{d_count}, tickets + 1);
    return URHGPU_OK;
}

// ---- min / max of a float32 array (util.minmax, util.pyx:20-36) ---------------------------------------------------
// NaN handling as the reference's loop: comparisons with NaN are false, so a NaN never replaces min / max unless it
// is element 0 -- reproduced by seeding every partial with x[0] and folding with the same comparisons.
__global__ __launch_bounds__(256) void k_minmax_partials(const float *x, int64_t n, float *part /*[2*grid]*/) {
    __shared__ float s_min[4], s_max[4];
    float mn = x[0], mx = x[0];
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = x[i];
        if (v < mn) mn = v;
        if (v > mx) mx = v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float a = __shfl_down(mn, o), b = __shfl_down(mx, o);
        if (a < mn) mn = a;
        if (b > mx) mx = b;
    }
    if ((threadIdx.x & 63) == 0) { s_min[threadIdx.x >> 6] = mn; s_max[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { if (s_min[w] < mn) mn = s_min[w]; if (s_max[w] > mx) mx = s_max[w]; }
        part[2 * blockIdx.x] = mn; part[2 * blockIdx.x + 1] = mx;
    }
}
__global__ void k_minmax_finish(const float *part, int nblocks, float *out2) {
    float mn = part[0], mx = part[1];
    for (int b = 1; b < nblocks; ++b) { if (part[2 * b] < mn) mn = part[2 * b]; if (part[2 * b + 1] > mx) mx = part[2 * b + 1]; }
    out2[0] = mn; out2[1] = mx;
}
constexpr int kMinmaxBlocks = 1024;
size_t minmax_scratch_bytes() { return (size_t)kMinmaxBlocks * 8 + 64; }
int launch_minmax(const float *x, int64_t n, float *d_out2, void *scratch, hipStream_t s) {
    if (n <= 0) return URHGPU_ERR_ARG;
    const int grid = (int)std::min<int64_t>((n + 255) / 256, kMinmaxBlocks);
    hipLaunchKernelGGL(k_minmax_partials, dim3(grid), dim3(256), 0, s, x, n, (float *)scratch);
    hipLaunchKernelGGL(k_minmax_finish, dim3(1), dim3(1), 0, s, (const float *)scratch, grid, d_out2);
    return URHGPU_OK;
}

// ---- numpy's pairwise float32 summation (numpy/core/src/umath/loops_utils.h.src, *_pairwise_sum) ---------------
//   n < 8            : res = 0; res += a[i] in order
//   n <= 128         : 8 accumulators r[j] = a[j]; r[j] += a[i + j] for i = 8, 16, ... < n - n % 8;
//                      res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)); then the n % 8 tail in order
//   n > 128          : n2 = n / 2; n2 -= n2 % 8; pairwise(a, n2) + pairwise(a + n2, n - n2)
// np.mean / np.var of a float32 array reduce with exactly this routine (float32 accumulators), so reproducing
// detect_center's bin width (float(np.var(rect))) bit for bit needs the same tree.  The leaves (<= 128 elements)
// are evaluated one per thread on the GPU from a host-built leaf table; the tree above them (~n/100 adds) is
// combined on the host in the same order.  mode 0: a[i] = x[i]; mode 1: a[i] = (x[i] - mean)^2 in float32.
struct Leaf { int64_t off; int32_t len; int32_t pad; };

__device__ __forceinline__ float pw_elem(const float *x, int64_t i, int mode, float mean) {
    const float v = x[i];
    if (mode == 0) return v;
    const float d = v - mean;
    return d * d;
}

__global__ __launch_bounds__(256) void k_pairwise_leaves(const float *x, const Leaf *leaves, int64_t n_leaves, int mode, float mean,
                                                          float *sums) {
    const int64_t k = blockIdx.x * 256ll + threadIdx.x;
    if (k >= n_leaves) return;
    const int64_t off = leaves[k].off;
    const int n = leaves[k].len;
    float res;
    if (n < 8) {
        res = 0.f;
        for (int i = 0; i < n; ++i) res += pw_elem(x, off + i, mode, mean);
    } else {
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = pw_elem(x, off + j, mode, mean);
        int i;
        for (i = 8; i < n - (n % 8); i += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] += pw_elem(x, off + i + j, mode, mean);
        }
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += pw_elem(x, off + i, mode, mean);
    }
    sums[k] = res;
}

static void build_leaves(int64_t off, int64_t n, std::vector<Leaf> &out) {
    if (n <= 128) { out.push_back(Leaf{off, (int32_t)n, 0}); return; }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    build_leaves(off, n2, out);
    build_leaves(off + n2, n - n2, out);
}
static float combine_leaves(int64_t n, const float *sums, int64_t &next) {
    if (n <= 128) return sums[next++];
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    const float a = combine_leaves(n2, sums, next);
    const float b = combine_leaves(n - n2, sums, next);
    return a + b;
}

// Synchronous: builds the leaf table for n, runs the leaf kernel, combines on the host.  *out = 0 + pairwise(...)
// (np.add.reduce starts from the identity 0).
int pairwise_sum_f32(urhgpu_ctx *ctx, const float *d_x, int64_t n, int mode, float mean, float *out) {
    if (n <= 0) { *out = 0.f; return URHGPU_OK; }
    std::vector<Leaf> leaves;
    leaves.reserve((size_t)(n / 64 + 8));
    build_leaves(0, n, leaves);
    const int64_t nl = (int64_t)leaves.size();
    const size_t need = ((size_t)nl * sizeof(Leaf) + 255) / 256 * 256 + (size_t)nl * 4 + 512;
    URH_TRY(ctx->staging.reserve(need));
    ctx->staging.reset();
    Leaf *d_leaves = (Leaf *)ctx->staging.take((size_t)nl * sizeof(Leaf));
    float *d_sums = (float *)ctx->staging.take((size_t)nl * 4);
    if (!d_leaves || !d_sums) return URHGPU_ERR_ARG;
    hipStream_t s = ctx->stream;
    URH_HIP(hipMemcpyAsync(d_leaves, leaves.data(), (size_t)nl * sizeof(Leaf), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_pairwise_leaves, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, s, d_x, d_leaves, nl, mode, mean, d_sums);
    URH_HIP(hipGetLastError());
    std::vector<float> sums((size_t)nl);
    URH_HIP(hipMemcpyAsync(sums.data(), d_sums, (size_t)nl * 4, hipMemcpyDeviceToHost, s));
    URH_HIP(hipStreamSynchronize(s));
    int64_t next = 0;
    volatile float total = combine_leaves(n, sums.data(), next);      // float32 adds, no excess precision
    *out = 0.0f + total;
    return URHGPU_OK;
}

// ---- np.histogram over explicit (float64, ascending) bin edges --------------------------------------------------------
// count[k] = #{x : e[k] <= x < e[k+1]}, the last bin closed on the right; x compared as float64 (numpy casts the data to
// the common type).  Edges are staged in LDS when they fit (<= 8192 edges), found by binary search; integer counts
// accumulate with atomics (order independent => exact).
constexpr int kHistEdgesLds = 8192;

__global__ __launch_bounds__(256) void k_hist_edges(const float *x, int64_t n, const double *edges, int n_edges,
                                                     unsigned long long *counts) {
    __shared__ double s_e[kHistEdgesLds];
    const bool in_lds = n_edges <= kHistEdgesLds;
    if (in_lds) for (int k = threadIdx.x; k < n_edges; k += 256) s_e[k] = edges[k];
    __syncthreads();
    const double *e = in_lds ? s_e : edges;
    const double e0 = e[0], eN = e[n_edges - 1];
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double v = (double)x[i];
        if (!(v >= e0) || !(v <= eN)) continue;          // outside (or NaN)
        // largest k with e[k] <= v, clamped to the last bin
        int lo = 0, hi = n_edges - 1;                    // invariant: e[lo] <= v, and (hi == n_edges-1 or e[hi] > v)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (e[mid] <= v) lo = mid; else hi = mid;
        }
        if (lo == n_edges - 1) lo = n_edges - 2;        // v == last edge: closed last bin
        atomicAdd(&counts[lo], 1ull);
    }
}

int launch_hist_edges(const float *x, int64_t n, const double *d_edges, int n_edges, int64_t *d_counts, hipStream_t s) {
    if (n_edges < 2) return URHGPU_ERR_ARG;
    if (hipMemsetAsync(d_counts, 0, (size_t)(n_edges - 1) * 8, s) != hipSuccess) return URHGPU_ERR_HIP;
    if (n <= 0) return URHGPU_OK;
    const int grid = (int)std::min<int64_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(k_hist_edges, dim3(grid), dim3(256), 0, s, x, n, d_edges, n_edges, (unsigned long long *)d_counts);
    return URHGPU_OK;
}

}  // namespace urh
