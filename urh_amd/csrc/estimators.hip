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
    (void)tickets;
    hipLaunchKernelGGL((k_scan_reduce<1, GtLoad>), dim3((unsigned)nb), dim3(kScanBlock), 0, s, d_n, ld, part, nb);
    if (nb > kScanDirect) hipLaunchKernelGGL((k_scan_partials<1>), dim3(1), dim3(kScanPartialsBlock), 0, s, d_n, part, nb, kScanDirect);
    hipLaunchKernelGGL((k_scan_apply<1, GtLoad, GtStore>), dim3((unsigned)nb), dim3(kScanBlock), 0, s, d_n, ld, part, nb, GtStore{x, out},
                       kScanDirect);
    hipLaunchKernelGGL((k_scan_finish<1, CountFinal>), dim3(1), dim3(kScanBlock), 0, s, d_n, part, nb, CountFinal{d_count}, kScanDirect);
    return URHGPU_OK;
}

int launch_compact_edges(const float *x, int64_t n, const int64_t *d_n, float center, int64_t *out, int64_t cap, int64_t *d_count,
                         void *scratch, int32_t *tickets, hipStream_t s) {
    const int64_t nb = std::max<int64_t>((n + kScanTile - 1) / kScanTile, 1);
    VecK<1> *part = (VecK<1> *)scratch;
    EdgeLoad ld{x, center};
    (void)tickets;
    hipLaunchKernelGGL((k_scan_reduce<1, EdgeLoad>), dim3((unsigned)nb), dim3(kScanBlock), 0, s, d_n, ld, part, nb);
    if (nb > kScanDirect) hipLaunchKernelGGL((k_scan_partials<1>), dim3(1), dim3(kScanPartialsBlock), 0, s, d_n, part, nb, kScanDirect);
    hipLaunchKernelGGL((k_scan_apply<1, EdgeLoad, EdgeStore>), dim3((unsigned)nb), dim3(kScanBlock), 0, s, d_n, ld, part, nb,
                       EdgeStore{out, cap}, kScanDirect);
    hipLaunchKernelGGL((k_scan_finish<1, CountFinal>), dim3(1), dim3(kScanBlock), 0, s, d_n, part, nb, CountFinal{d_count}, kScanDirect);
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
__global__ __launch_bounds__(64) void k_minmax_finish(const float *part, int nblocks, float *out2) {
    float mn = part[0], mx = part[1];
    for (int b = threadIdx.x; b < nblocks; b += 64) { if (part[2 * b] < mn) mn = part[2 * b]; if (part[2 * b + 1] > mx) mx = part[2 * b + 1]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float a = __shfl_down(mn, o), b = __shfl_down(mx, o);
        if (a < mn) mn = a;
        if (b > mx) mx = b;
    }
    if (threadIdx.x == 0) { out2[0] = mn; out2[1] = mx; }
}
// ---- util.minmax for every element type of the fused `iq` type (util.pyx:20-36), same comparisons and NaN behaviour ----------
template <typename T>
__global__ __launch_bounds__(256) void k_minmax_partials_t(const T *x, int64_t n, T *part /*[2*grid]*/) {
    __shared__ T s_min[4], s_max[4];
    T mn = x[0], mx = x[0];
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const T v = x[i];
        if (v < mn) mn = v;
        if (v > mx) mx = v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const T a = (T)__shfl_down((float)mn, o), b = (T)__shfl_down((float)mx, o);       // every element type is exact in float32
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
template <typename T>
__global__ __launch_bounds__(64) void k_minmax_finish_t(const T *part, int nblocks, T *out2) {
    T mn = part[0], mx = part[1];
    for (int b = threadIdx.x; b < nblocks; b += 64) { if (part[2 * b] < mn) mn = part[2 * b]; if (part[2 * b + 1] > mx) mx = part[2 * b + 1]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const T a = (T)__shfl_down((float)mn, o), b = (T)__shfl_down((float)mx, o);
        if (a < mn) mn = a;
        if (b > mx) mx = b;
    }
    if (threadIdx.x == 0) { out2[0] = mn; out2[1] = mx; }
}
constexpr int kMinmaxBlocks = 1024;
template <typename T>
static int minmax_launch_t(const void *x, int64_t n, void *d_out2, void *scratch, hipStream_t s) {
    const int grid = (int)std::min<int64_t>((n + 255) / 256, kMinmaxBlocks);
    hipLaunchKernelGGL(k_minmax_partials_t<T>, dim3(grid), dim3(256), 0, s, (const T *)x, n, (T *)scratch);
    hipLaunchKernelGGL(k_minmax_finish_t<T>, dim3(1), dim3(64), 0, s, (const T *)scratch, grid, (T *)d_out2);
    return URHGPU_OK;
}
// d_out2: {min, max} in the element type
int launch_minmax_any(const void *x, int dtype, int64_t n, void *d_out2, void *scratch, hipStream_t s) {
    if (n <= 0) return URHGPU_ERR_ARG;
    switch (dtype) {
        case URHGPU_DT_I8: return minmax_launch_t<int8_t>(x, n, d_out2, scratch, s);
        case URHGPU_DT_U8: return minmax_launch_t<uint8_t>(x, n, d_out2, scratch, s);
        case URHGPU_DT_I16: return minmax_launch_t<int16_t>(x, n, d_out2, scratch, s);
        case URHGPU_DT_U16: return minmax_launch_t<uint16_t>(x, n, d_out2, scratch, s);
        case URHGPU_DT_F32: return minmax_launch_t<float>(x, n, d_out2, scratch, s);
        default: return URHGPU_ERR_DTYPE;
    }
}

// ---- segment_messages_from_magnitudes on caller-supplied magnitudes (auto_interpretation.pyx:55-111): the comparison
// `magnitudes[i] > noise_threshold` (float or double magnitude against a C float) as a 0 / 1 float, which the run segmentation
// then slices at 0.5 ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_above_flags(const T *mag, int64_t n, float thr, float *flags) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) flags[i] = ((double)mag[i] > (double)thr) ? 1.0f : 0.0f;
}
int launch_above_flags(const void *mag, int is_f64, int64_t n, float thr, float *flags, hipStream_t s) {
    if (n <= 0) return URHGPU_OK;
    const unsigned g = (unsigned)std::min<int64_t>((n + 255) / 256, 65536);
    if (is_f64) hipLaunchKernelGGL(k_above_flags<double>, dim3(g), dim3(256), 0, s, (const double *)mag, n, thr, flags);
    else hipLaunchKernelGGL(k_above_flags<float>, dim3(g), dim3(256), 0, s, (const float *)mag, n, thr, flags);
    return URHGPU_OK;
}

// ---- auto_interpretation.median_filter (auto_interpretation.pyx:213-240): window data[i : i + k] cut at the end of the array, the
// values rounded to float, sorted, element k' / 2 ---------------------------------------------------------------------------------
constexpr int kMedianMaxK = 64;
__global__ __launch_bounds__(256) void k_median_filter(const double *data, int64_t n, int k, float *out) {
    const int64_t i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= n) return;
    int kk = k;
    if (i + kk > n) kk = (int)(n - i);
    float buf[kMedianMaxK];
    for (int j = 0; j < kk; ++j) buf[j] = (float)data[i + j];
    for (int a = 1; a < kk; ++a) {                          // insertion sort (std::sort's result on floats; NaN-free data)
        const float v = buf[a];
        int b = a - 1;
        while (b >= 0 && buf[b] > v) { buf[b + 1] = buf[b]; --b; }
        buf[b + 1] = v;
    }
    out[i] = buf[kk / 2];
}
int launch_median_filter(const double *data, int64_t n, int k, float *out, hipStream_t s) {
    if (k < 1 || k > kMedianMaxK) return URHGPU_ERR_UNSUPPORTED;
    if (n <= 0) return URHGPU_OK;
    hipLaunchKernelGGL(k_median_filter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, data, n, k, out);
    return URHGPU_OK;
}

size_t minmax_scratch_bytes() { return (size_t)kMinmaxBlocks * 8 + 64; }
int launch_minmax(const float *x, int64_t n, float *d_out2, void *scratch, hipStream_t s) {
    if (n <= 0) return URHGPU_ERR_ARG;
    const int grid = (int)std::min<int64_t>((n + 255) / 256, kMinmaxBlocks);
    hipLaunchKernelGGL(k_minmax_partials, dim3(grid), dim3(256), 0, s, x, n, (float *)scratch);
    hipLaunchKernelGGL(k_minmax_finish, dim3(1), dim3(64), 0, s, (const float *)scratch, grid, d_out2);
    return URHGPU_OK;
}

// ---- numpy's float32 summation (np.add.reduce, what np.mean / np.var use) -------------------------------------------
// np.add.reduce walks a contiguous array in chunks of the ufunc buffer size (8192 elements) and accumulates
//     total = (((0 + pw(chunk 0)) + pw(chunk 1)) + ...)                       [verified against numpy 2.2 on this host]
// where pw is the pairwise routine of numpy/core/src/umath/loops_utils.h.src (float32 accumulators):
//   n < 8            : res = 0; res += a[i] in order
//   n <= 128         : 8 accumulators r[j] = a[j]; r[j] += a[i + j] for i = 8, 16, ... < n - n % 8;
//                      res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)); then the n % 8 tail in order
//   n > 128          : n2 = n / 2; n2 -= n2 % 8; pw(a, n2) + pw(a + n2, n - n2)
// A full chunk is therefore a perfect binary tree over 64 leaves of 128 elements.  Reproducing detect_center's bin
// width (float(np.var(rect))) bit for bit needs exactly this order: leaves are evaluated one per thread, each full
// chunk's tree by one thread (k_pairwise_chunks), and the O(n / 8192) chunk totals plus the irregular last chunk
// are accumulated on the host.  mode 0: a[i] = x[i]; mode 1: a[i] = (x[i] - mean)^2 in float32.
constexpr int kPwChunk = 8192, kPwLeaf = 128, kPwLeavesPerChunk = kPwChunk / kPwLeaf;
struct Leaf { int64_t off; int32_t len; int32_t pad; };

__device__ __forceinline__ float pw_elem(const float *x, int64_t i, int mode, float mean) {
    const float v = x[i];
    if (mode == 0) return v;
    const float d = v - mean;
    return d * d;
}

// leaves [0, n_regular) are the 128-element leaves of the full chunks; leaves beyond come from `extra`
__global__ __launch_bounds__(256) void k_pairwise_leaves(const float *x, int64_t n_regular, const Leaf *extra, int64_t n_extra,
                                                          int mode, float mean, float *sums) {
    const int64_t k = blockIdx.x * 256ll + threadIdx.x;
    if (k >= n_regular + n_extra) return;
    int64_t off = k * kPwLeaf;
    int n = kPwLeaf;
    if (k >= n_regular) { off = extra[k - n_regular].off; n = extra[k - n_regular].len; }
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

// one thread per full chunk: perfect binary tree over its 64 leaf sums, in place order
__global__ __launch_bounds__(64) void k_pairwise_chunks(const float *leaf_sums, int64_t n_chunks, float *chunk_sums) {
    const int64_t c = blockIdx.x * 64ll + threadIdx.x;
    if (c >= n_chunks) return;
    float s[kPwLeavesPerChunk];
#pragma unroll
    for (int i = 0; i < kPwLeavesPerChunk; ++i) s[i] = leaf_sums[c * kPwLeavesPerChunk + i];
#pragma unroll
    for (int w = kPwLeavesPerChunk / 2; w >= 1; w >>= 1) {
#pragma unroll
        for (int i = 0; i < w; ++i) s[i] = s[2 * i] + s[2 * i + 1];
    }
    chunk_sums[c] = s[0];
}

static void build_leaves(int64_t off, int64_t n, std::vector<Leaf> &out) {
    if (n <= kPwLeaf) { out.push_back(Leaf{off, (int32_t)n, 0}); return; }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    build_leaves(off, n2, out);
    build_leaves(off + n2, n - n2, out);
}
static float combine_leaves(int64_t n, const float *sums, int64_t &next) {
    if (n <= kPwLeaf) return sums[next++];
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    volatile float a = combine_leaves(n2, sums, next);
    volatile float b = combine_leaves(n - n2, sums, next);
    return a + b;
}

// Synchronous.  *out = np.add.reduce(a) for the float32 sequence a described above.
int pairwise_sum_f32(urhgpu_ctx *ctx, const float *d_x, int64_t n, int mode, float mean, float *out) {
    if (n <= 0) { *out = 0.f; return URHGPU_OK; }
    const int64_t n_chunks = n / kPwChunk, rest = n % kPwChunk;
    const int64_t n_regular = n_chunks * kPwLeavesPerChunk;
    std::vector<Leaf> extra;
    if (rest) build_leaves(n_chunks * kPwChunk, rest, extra);
    const int64_t n_extra = (int64_t)extra.size(), nl = n_regular + n_extra;
    const size_t need = (size_t)(n_extra + 1) * sizeof(Leaf) + (size_t)nl * 4 + (size_t)(n_chunks + 1) * 4 + 2048;
    URH_TRY(ctx->staging.reserve(need));
    ctx->staging.reset();
    Leaf *d_extra = (Leaf *)ctx->staging.take((size_t)(n_extra + 1) * sizeof(Leaf));
    float *d_sums = (float *)ctx->staging.take((size_t)nl * 4);
    float *d_chunk = (float *)ctx->staging.take((size_t)(n_chunks + 1) * 4);
    if (!d_extra || !d_sums || !d_chunk) return URHGPU_ERR_ARG;
    hipStream_t s = ctx->stream;
    if (n_extra) URH_HIP(hipMemcpyAsync(d_extra, extra.data(), (size_t)n_extra * sizeof(Leaf), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_pairwise_leaves, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, s, d_x, n_regular, d_extra, n_extra, mode,
                       mean, d_sums);
    if (n_chunks) hipLaunchKernelGGL(k_pairwise_chunks, dim3((unsigned)((n_chunks + 63) / 64)), dim3(64), 0, s, d_sums, n_chunks, d_chunk);
    URH_HIP(hipGetLastError());
    std::vector<float> chunk((size_t)n_chunks), tail((size_t)n_extra);
    if (n_chunks) URH_HIP(hipMemcpyAsync(chunk.data(), d_chunk, (size_t)n_chunks * 4, hipMemcpyDeviceToHost, s));
    if (n_extra) URH_HIP(hipMemcpyAsync(tail.data(), d_sums + n_regular, (size_t)n_extra * 4, hipMemcpyDeviceToHost, s));
    URH_HIP(wait_stream(ctx, s));
    volatile float total = 0.0f;                                   // float32 adds, no excess precision
    for (int64_t c = 0; c < n_chunks; ++c) total = total + chunk[(size_t)c];
    if (rest) { int64_t next = 0; const float t = combine_leaves(rest, tail.data(), next); total = total + t; }
    *out = total;
    return URHGPU_OK;
}

// ---- np.histogram over explicit (float64, ascending) bin edges --------------------------------------------------------
// count[k] = #{x : e[k] <= x < e[k+1]}, the last bin closed on the right; x compared as float64 (numpy casts the data to
// the common type).  Edges are staged in LDS when they fit (<= 8192 edges), found by binary search; integer counts
// accumulate with atomics (order independent => exact).
constexpr int kHistEdgesLds = 4096;

__global__ __launch_bounds__(256) void k_hist_edges(const float *x, int64_t n, const double *edges, int n_edges,
                                                     unsigned long long *counts) {
    // edges and a private copy of the counts live in LDS when they fit: detect_center's histograms are sharply peaked
    // (two symbol levels), so device-wide atomics on the few hot bins would serialise the whole pass
    __shared__ double s_e[kHistEdgesLds];
    __shared__ unsigned int s_c[kHistEdgesLds];
    const bool in_lds = n_edges <= kHistEdgesLds;
    if (in_lds) for (int k = threadIdx.x; k < n_edges; k += 256) { s_e[k] = edges[k]; s_c[k] = 0u; }
    __syncthreads();
    const double *e = in_lds ? s_e : edges;
    const double e0 = e[0], eN = e[n_edges - 1];
    // a workgroup owns a contiguous slice (so that its private counters cannot overflow 32 bits: slice < 2^32 samples)
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t a0 = blockIdx.x * per, a1 = (a0 + per < n) ? a0 + per : n;
    for (int64_t i = a0 + threadIdx.x; i < a1; i += 256) {
        const double v = (double)x[i];
        if (!(v >= e0) || !(v <= eN)) continue;          // outside (or NaN)
        int lo = 0, hi = n_edges - 1;                    // invariant: e[lo] <= v; hi == n_edges-1 or e[hi] > v
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (e[mid] <= v) lo = mid; else hi = mid;
        }
        // lo <= n_edges - 2: v == last edge lands in the (closed) last bin
        if (in_lds) atomicAdd(&s_c[lo], 1u); else atomicAdd(&counts[lo], 1ull);
    }
    if (in_lds) {
        __syncthreads();
        for (int k = threadIdx.x; k < n_edges - 1; k += 256)
            if (s_c[k]) atomicAdd(&counts[k], (unsigned long long)s_c[k]);
    }
}

int launch_hist_edges(const float *x, int64_t n, const double *d_edges, int n_edges, int64_t *d_counts, hipStream_t s) {
    if (n_edges < 2) return URHGPU_ERR_ARG;
    if (hipMemsetAsync(d_counts, 0, (size_t)(n_edges - 1) * 8, s) != hipSuccess) return URHGPU_ERR_HIP;
    if (n <= 0) return URHGPU_OK;
    const int grid = (int)std::min<int64_t>((n + 4095) / 4096, 2048);
    hipLaunchKernelGGL(k_hist_edges, dim3(grid), dim3(256), 0, s, x, n, d_edges, n_edges, (unsigned long long *)d_counts);
    return URHGPU_OK;
}

}  // namespace urh
