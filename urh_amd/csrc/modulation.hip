// modulation.hip -- AutoInterpretation.detect_modulation on the GPU
//   /root/reference/src/urh/ainterpretation/AutoInterpretation.py:150-223   (decision tree, first 100 messages)
//   /root/reference/src/urh/ainterpretation/Wavelet.py:7-43                 (continuous Haar wavelet transform through the FFT)
//   /root/reference/src/urh/cythonext/auto_interpretation.pyx:213-240       (median filter: forward window of k, cut at the end)
//
// The reference classifies a message from four variances -- of |CWT| of the normalised samples, of |CWT| of the samples scaled to
// unit magnitude, and of both after an 11-tap median filter -- and, for FSK against a single OOK pulse, from the ten largest bins of
// the spectrum.  It is a floating-point classifier: the reference evaluates the forward FFT in single precision (numpy >= 2.0,
// complex64 input) with pocketfft's mixed radices; here every transform is a radix-2 Stockham FFT in double precision, so the
// variances agree to about 1e-6 relative and the LABEL is the parity criterion (tests compare labels with the host restatement and,
// where the reference is staged, with the reference itself; the variances are returned for inspection).
//
// Per message: compaction of the non-zero samples + their (lexicographic, as np.max orders complex numbers) maximum; two inputs
// x / |max| and x / |x| in numpy's complex64 arithmetic; FFT, multiplication by the wavelet's transform evaluated on the fly, inverse
// FFT; magnitudes, median filter, four variances by deterministic two-level sums; ten largest spectrum bins per slice for the host's
// final look.  Transforms of up to 2048 points run in LDS in one launch, longer ones pass by pass through HBM.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <math.h>
#include <vector>

#include "common.hpp"
#include "launchers.hpp"

namespace urh {

constexpr int kMdBlock = 256;
constexpr int kMdPer = 16;
constexpr int kMdTile = kMdBlock * kMdPer;
constexpr int kMdSmallFft = 2048;              // LDS ping-pong: 2 x 2048 x 16 B = 64 KB
constexpr int kMdTopK = 10;
constexpr int kMdTopBlocks = 64;

__device__ __forceinline__ bool md_nonzero(float2 v) { return hypotf(v.x, v.y) > 0.0f; }      // np.abs(data) > 0
__device__ __forceinline__ bool md_cgreater(float2 a, float2 b) { return a.x > b.x || (a.x == b.x && a.y > b.y); }   // numpy's complex order

// per tile: number of non-zero samples, and the largest of them
__global__ __launch_bounds__(kMdBlock) void k_md_count(const float2 *x, int64_t n, int32_t *tile_cnt, float2 *tile_max, int32_t *tile_has) {
    __shared__ int s_c[kMdBlock / 64];
    __shared__ float2 s_m[kMdBlock / 64];
    __shared__ int s_h[kMdBlock / 64];
    const int64_t i0 = (int64_t)blockIdx.x * kMdTile + (int64_t)threadIdx.x * kMdPer;
    int c = 0, has = 0;
    float2 mx = make_float2(0.f, 0.f);
#pragma unroll
    for (int j = 0; j < kMdPer; ++j) {
        if (i0 + j < n) {
            const float2 v = x[i0 + j];
            if (md_nonzero(v)) { ++c; if (!has || md_cgreater(v, mx)) mx = v; has = 1; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        c += __shfl_xor(c, o);
        const float ox = __shfl_xor(mx.x, o), oy = __shfl_xor(mx.y, o);
        const int oh = __shfl_xor(has, o);
        if (oh && (!has || md_cgreater(make_float2(ox, oy), mx))) { mx = make_float2(ox, oy); has = 1; }
    }
    if ((threadIdx.x & 63) == 0) { s_c[threadIdx.x >> 6] = c; s_m[threadIdx.x >> 6] = mx; s_h[threadIdx.x >> 6] = has; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int tc = 0, th = 0; float2 tm = make_float2(0.f, 0.f);
        for (int w = 0; w < kMdBlock / 64; ++w) {
            tc += s_c[w];
            if (s_h[w] && (!th || md_cgreater(s_m[w], tm))) { tm = s_m[w]; th = 1; }
        }
        tile_cnt[blockIdx.x] = tc; tile_max[blockIdx.x] = tm; tile_has[blockIdx.x] = th;
    }
}
// exclusive prefix of the tile counts (one workgroup), total in pre[n_tiles]
__global__ __launch_bounds__(1024) void k_md_scan(const int32_t *cnt, int64_t n_tiles, int64_t *pre) {
    __shared__ int64_t s_w[16];
    __shared__ int64_t s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int64_t b0 = 0; b0 < n_tiles; b0 += 1024) {
        const int64_t i = b0 + threadIdx.x;
        const int64_t v = (i < n_tiles) ? cnt[i] : 0;
        int64_t incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int64_t u = __shfl_up(incl, o); if (lane >= o) incl += u; }
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        int64_t base = s_carry, total = 0;
        for (int w = 0; w < 16; ++w) { if (w < wave) base += s_w[w]; total += s_w[w]; }
        if (i < n_tiles) pre[i] = base + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) s_carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) pre[n_tiles] = s_carry;
}
__global__ __launch_bounds__(kMdBlock) void k_md_compact(const float2 *x, int64_t n, const int64_t *pre, float2 *kept) {
    __shared__ int s_w[kMdBlock / 64];
    const int64_t i0 = (int64_t)blockIdx.x * kMdTile + (int64_t)threadIdx.x * kMdPer;
    float2 v[kMdPer];
    bool k[kMdPer];
    int c = 0;
#pragma unroll
    for (int j = 0; j < kMdPer; ++j) { v[j] = (i0 + j < n) ? x[i0 + j] : make_float2(0.f, 0.f); k[j] = md_nonzero(v[j]); c += k[j]; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o); if (lane >= o) incl += u; }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    int wbase = 0;
#pragma unroll
    for (int w = 0; w < kMdBlock / 64; ++w) if (w < wave) wbase += s_w[w];
    int64_t o = pre[blockIdx.x] + wbase + incl - c;
#pragma unroll
    for (int j = 0; j < kMdPer; ++j) if (k[j]) kept[o++] = v[j];
}

// the two transform inputs, complex64 arithmetic as numpy evaluates `data / np.abs(np.max(data))` and `data / np.abs(data)`
// (complex / real through the complex quotient: (a.re + a.im * 0) * (1 / b), (a.im - a.re * 0) * (1 / b) in float32), then widened
__global__ void k_md_prepare(const float2 *kept, int64_t n2, float scale, double2 *x1, double2 *x2) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
        const float2 v = kept[i];
        const float r = 1.0f / scale;
        const float2 d = make_float2((v.x + v.y * 0.0f) * r, (v.y - v.x * 0.0f) * r);
        const float a = hypotf(d.x, d.y), ra = 1.0f / a;
        const float2 u = make_float2((d.x + d.y * 0.0f) * ra, (d.y - d.x * 0.0f) * ra);
        x1[i] = make_double2((double)d.x, (double)d.y);
        x2[i] = make_double2((double)u.x, (double)u.y);
    }
}

__device__ __forceinline__ double2 md_cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// Stockham radix-2: one pass  y[q + s (2p)] = a + b,  y[q + s (2p + 1)] = (a - b) w_p  with a = x[q + s p], b = x[q + s (p + m)],
// m = n_cur / 2, w_p = exp(sign 2 pi i p / n_cur); passes n_cur = N, N/2, ..., 2 with s = 1, 2, ..., N/2 leave natural order.
__global__ __launch_bounds__(kMdBlock) void k_md_fft_pass(const double2 *x, double2 *y, int64_t N, int64_t n_cur, int64_t s, double sign) {
    const int64_t half = N >> 1, m = n_cur >> 1;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < half; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = t / s, q = t - p * s;
        double sn, cs;
        sincospi(sign * 2.0 * (double)p / (double)n_cur, &sn, &cs);
        const double2 a = x[q + s * p], b = x[q + s * (p + m)];
        y[q + s * (2 * p)] = make_double2(a.x + b.x, a.y + b.y);
        y[q + s * (2 * p + 1)] = md_cmul(make_double2(a.x - b.x, a.y - b.y), make_double2(cs, sn));
    }
}
// the whole transform of N <= kMdSmallFft points in LDS
__global__ __launch_bounds__(kMdBlock) void k_md_fft_small(const double2 *x, double2 *y, int N, double sign) {
    extern __shared__ double2 s_buf[];
    double2 *a = s_buf, *b = s_buf + N;
    for (int t = threadIdx.x; t < N; t += kMdBlock) a[t] = x[t];
    __syncthreads();
    int s = 1;
    for (int n_cur = N; n_cur > 1; n_cur >>= 1, s <<= 1) {
        const int m = n_cur >> 1;
        for (int t = threadIdx.x; t < (N >> 1); t += kMdBlock) {
            const int p = t / s, q = t - p * s;
            double sn, cs;
            sincospi(sign * 2.0 * (double)p / (double)n_cur, &sn, &cs);
            const double2 u = a[q + s * p], v = a[q + s * (p + m)];
            b[q + s * (2 * p)] = make_double2(u.x + v.x, u.y + v.y);
            b[q + s * (2 * p + 1)] = md_cmul(make_double2(u.x - v.x, u.y - v.y), make_double2(cs, sn));
        }
        __syncthreads();
        double2 *tmp = a; a = b; b = tmp;
    }
    for (int t = threadIdx.x; t < N; t += kMdBlock) y[t] = a[t];
}

// x_hat * psi_hat (Wavelet.py:27-40) in place, for both transforms:
//   omega_j = f j (j < N/2), f (-j) (j >= N/2: the reference negates the index, it does not wrap it), f = 2 pi / N
//   psi_hat_j = sqrt(2 pi scale) * 1j * (-1 + exp(0.5j * scale * omega_j))^2 / (j == 0 ? 1 : omega_j)
__global__ void k_md_wavelet(double2 *a, double2 *b, int64_t N, double scale) {
    const double f = 2.0 * M_PI / (double)N, amp = sqrt(2.0 * M_PI * scale);
    for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < N; j += (int64_t)gridDim.x * blockDim.x) {
        const double omega = (j < N / 2) ? f * (double)j : f * ((double)j * -1.0);
        const double arg = scale * omega;
        double sn, cs;
        sincos(0.5 * arg, &sn, &cs);
        const double2 e = make_double2(-1.0 + cs, sn);                         // -1 + exp(0.5j arg)
        const double2 sq = md_cmul(e, e);
        const double2 num = make_double2(-sq.y, sq.x);                         // 1j * sq
        const double den = (j == 0) ? 1.0 : (arg / scale);                     // omega_cpy = (scale * omega) / scale, [0] = 1
        const double2 psi = make_double2(amp * (num.x / den), amp * (num.y / den));
        a[j] = md_cmul(a[j], psi);
        b[j] = md_cmul(b[j], psi);
    }
}

// magnitudes of W[2 scale : -2 scale] (the inverse transform's 1 / N applied here)
__global__ void k_md_mag(const double2 *w1, const double2 *w2, int64_t N, int64_t skip, double *m1, double *m2) {
    const int64_t L = N - 2 * skip;
    const double inv = 1.0 / (double)N;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < L; i += (int64_t)gridDim.x * blockDim.x) {
        const double2 a = w1[i + skip], b = w2[i + skip];
        m1[i] = hypot(a.x * inv, a.y * inv);
        m2[i] = hypot(b.x * inv, b.y * inv);
    }
}
// median filter (auto_interpretation.pyx:213-240): window data[i : i + k] cut at the end, values rounded to float, sorted, [k' / 2]
__global__ void k_md_median(const double *m1, const double *m2, int64_t L, int k, float *f1, float *f2) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < 2 * L; i += (int64_t)gridDim.x * blockDim.x) {
        const bool second = i >= L;
        const int64_t p = second ? i - L : i;
        const double *src = second ? m2 : m1;
        int kk = k;
        if (p + kk > L) kk = (int)(L - p);
        float buf[16];
        for (int j = 0; j < kk; ++j) buf[j] = (float)src[p + j];
        for (int a = 1; a < kk; ++a) {                                          // insertion sort of at most 11 values
            const float v = buf[a];
            int c = a - 1;
            while (c >= 0 && buf[c] > v) { buf[c + 1] = buf[c]; --c; }
            buf[c + 1] = v;
        }
        (second ? f2 : f1)[p] = buf[kk / 2];
    }
}

// deterministic two-level sums of up to four arrays at once: part[arr][block] = sum over the block's slice of (v - mean_arr)^power
struct MdSumArgs {
    const double *d[2];      // arrays 0, 1 (float64)
    const float *f[2];       // arrays 2, 3 (float32)
    int64_t L;
    double mean[4];
    int square;              // 0: sum of v, 1: sum of (v - mean)^2
    double *part;            // [4][gridDim.x]
};
__global__ __launch_bounds__(kMdBlock) void k_md_sum(const MdSumArgs a) {
    __shared__ double s_p[kMdBlock / 64][4];
    const int64_t per = (a.L + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = (lo + per < a.L) ? lo + per : a.L;
    double acc[4] = {0, 0, 0, 0};
    for (int64_t i = lo + threadIdx.x; i < hi; i += kMdBlock) {
        const double v[4] = {a.d[0][i], a.d[1][i], (double)a.f[0][i], (double)a.f[1][i]};
#pragma unroll
        for (int k = 0; k < 4; ++k) { const double t = a.square ? (v[k] - a.mean[k]) : v[k]; acc[k] += a.square ? t * t : t; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[k] += __shfl_down(acc[k], o);
    }
    if ((threadIdx.x & 63) == 0) for (int k = 0; k < 4; ++k) s_p[threadIdx.x >> 6][k] = acc[k];
    __syncthreads();
    if (threadIdx.x < 4) {
        double t = 0.0;
        for (int w = 0; w < kMdBlock / 64; ++w) t += s_p[w][threadIdx.x];
        a.part[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = t;
    }
}

// the ten largest |x_hat| of every slice of the fftshift-ed spectrum: (magnitude, shifted index) candidates for the host
__global__ __launch_bounds__(kMdBlock) void k_md_topk(const double2 *xhat, int64_t N, double *cand_val, int64_t *cand_idx) {
    __shared__ double s_v[kMdBlock];
    __shared__ int64_t s_i[kMdBlock];
    __shared__ int64_t s_taken[kMdTopK];
    const int64_t per = (N + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = (lo + per < N) ? lo + per : N;       // slice of SHIFTED indices
    for (int r = 0; r < kMdTopK; ++r) {
        double best = -1.0; int64_t bi = -1;
        for (int64_t i = lo + threadIdx.x; i < hi; i += kMdBlock) {
            bool taken = false;
            for (int u = 0; u < r; ++u) taken |= (s_taken[u] == i);
            if (taken) continue;
            const double2 v = xhat[(i + N / 2) % N];                            // fftshift: shifted[i] = x[(i + N/2) mod N] (N even)
            const double mg = hypot(v.x, v.y);
            if (mg > best) { best = mg; bi = i; }
        }
        s_v[threadIdx.x] = best; s_i[threadIdx.x] = bi;
        __syncthreads();
        for (int o = kMdBlock / 2; o > 0; o >>= 1) {
            if (threadIdx.x < o && (s_v[threadIdx.x + o] > s_v[threadIdx.x])) { s_v[threadIdx.x] = s_v[threadIdx.x + o]; s_i[threadIdx.x] = s_i[threadIdx.x + o]; }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            s_taken[r] = s_i[0];
            cand_val[(int64_t)blockIdx.x * kMdTopK + r] = s_v[0];
            cand_idx[(int64_t)blockIdx.x * kMdTopK + r] = s_i[0];
        }
        __syncthreads();
    }
}

}  // namespace urh

using namespace urh;

namespace {

int md_grid(int64_t n, int per_thread = 1) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + kMdBlock * per_thread - 1) / (kMdBlock * per_thread), 4096)); }

// forward (sign -1) / backward (+1, unscaled) transform of N points; result in *out (one of the two buffers)
int md_fft(hipStream_t s, double2 *buf_a, double2 *buf_b, int64_t N, double sign, double2 **out) {
    if (N <= kMdSmallFft) {
        hipLaunchKernelGGL(k_md_fft_small, dim3(1), dim3(kMdBlock), (size_t)N * 2 * sizeof(double2), s, buf_a, buf_b, (int)N, sign);
        *out = buf_b;
        return URHGPU_OK;
    }
    double2 *x = buf_a, *y = buf_b;
    int64_t st = 1;
    for (int64_t n_cur = N; n_cur > 1; n_cur >>= 1, st <<= 1) {
        hipLaunchKernelGGL(k_md_fft_pass, dim3(md_grid(N / 2)), dim3(kMdBlock), 0, s, x, y, N, n_cur, st, sign);
        std::swap(x, y);
    }
    *out = x;
    return URHGPU_OK;
}

enum { kLabNone = 0, kLabOok = 1, kLabAsk = 2, kLabFsk = 3, kLabPsk = 4 };

}  // namespace

extern "C" {

int urhgpu_detect_modulation_dev(urhgpu_ctx *ctx, const float *d_iq, int64_t n, const int64_t *ranges, int n_msgs, int wavelet_scale,
                                 int median_filter_order, int *labels_out, double *vars_out) {
    if (!ctx || n < 0 || n_msgs < 0 || wavelet_scale < 1 || median_filter_order < 1 || median_filter_order > 16 ||
        (n_msgs > 0 && (!ranges || !labels_out || !d_iq)))
        return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(join_tail(ctx));                               // pipelined context: the arena below may still serve the last pass's tail
    hipStream_t s = ctx->stream;
    int64_t max_len = 0;
    for (int m = 0; m < n_msgs; ++m) {
        const int64_t a = ranges[2 * m], b = ranges[2 * m + 1];
        if (a < 0 || b < a || b > n) return URHGPU_ERR_ARG;
        max_len = std::max(max_len, b - a);
    }
    if (n_msgs == 0) return URHGPU_OK;
    int64_t n2max = 1;
    while (n2max * 2 <= std::max<int64_t>(max_len, 1)) n2max *= 2;
    const int64_t tiles_max = (std::max<int64_t>(max_len, 1) + kMdTile - 1) / kMdTile;
    // scratch (the arena is free between the batched estimator calls)
    const size_t need = (size_t)tiles_max * (4 + 8 + 4) + (size_t)(tiles_max + 1) * 8 + (size_t)max_len * 8 + 4 * (size_t)n2max * 16 +
                        2 * (size_t)n2max * 8 + 2 * (size_t)n2max * 4 + 4 * 4096 * 8 + kMdTopBlocks * kMdTopK * 16 + 64 * 256;
    URH_TRY(ctx->arena.reserve(need));
    ctx->arena.reset();
    int32_t *d_cnt = (int32_t *)ctx->arena.take((size_t)tiles_max * 4);
    float2 *d_tmax = (float2 *)ctx->arena.take((size_t)tiles_max * 8);
    int32_t *d_thas = (int32_t *)ctx->arena.take((size_t)tiles_max * 4);
    int64_t *d_pre = (int64_t *)ctx->arena.take((size_t)(tiles_max + 1) * 8);
    float2 *d_kept = (float2 *)ctx->arena.take((size_t)std::max<int64_t>(max_len, 1) * 8);
    double2 *d_c[4];
    for (int k = 0; k < 4; ++k) d_c[k] = (double2 *)ctx->arena.take((size_t)n2max * 16);
    double *d_m1 = (double *)ctx->arena.take((size_t)n2max * 8), *d_m2 = (double *)ctx->arena.take((size_t)n2max * 8);
    float *d_f1 = (float *)ctx->arena.take((size_t)n2max * 4), *d_f2 = (float *)ctx->arena.take((size_t)n2max * 4);
    double *d_part = (double *)ctx->arena.take(4 * 4096 * 8);
    double *d_cv = (double *)ctx->arena.take(kMdTopBlocks * kMdTopK * 8);
    int64_t *d_ci = (int64_t *)ctx->arena.take(kMdTopBlocks * kMdTopK * 8);
    if (!d_cnt || !d_tmax || !d_thas || !d_pre || !d_kept || !d_c[3] || !d_m1 || !d_m2 || !d_f1 || !d_f2 || !d_part || !d_cv || !d_ci) return URHGPU_ERR_ARG;
    if (hipFuncSetAttribute((const void *)k_md_fft_small, hipFuncAttributeMaxDynamicSharedMemorySize, kMdSmallFft * 2 * (int)sizeof(double2)) != hipSuccess)
        return URHGPU_ERR_HIP;
    std::vector<int32_t> h_cnt; std::vector<float2> h_tmax; std::vector<int32_t> h_thas;
    std::vector<double> h_part((size_t)4 * 4096), h_cv(kMdTopBlocks * kMdTopK);
    std::vector<int64_t> h_ci(kMdTopBlocks * kMdTopK);
    for (int m = 0; m < n_msgs; ++m) {
        double *vo = vars_out ? vars_out + 4 * (size_t)m : nullptr;
        if (vo) vo[0] = vo[1] = vo[2] = vo[3] = __builtin_nan("");
        labels_out[m] = kLabNone;
        const int64_t len = ranges[2 * m + 1] - ranges[2 * m];
        if (len == 0) continue;
        const float2 *x = (const float2 *)d_iq + ranges[2 * m];
        const int64_t nt = (len + kMdTile - 1) / kMdTile;
        hipLaunchKernelGGL(k_md_count, dim3((unsigned)nt), dim3(kMdBlock), 0, s, x, len, d_cnt, d_tmax, d_thas);
        hipLaunchKernelGGL(k_md_scan, dim3(1), dim3(1024), 0, s, d_cnt, nt, d_pre);
        hipLaunchKernelGGL(k_md_compact, dim3((unsigned)nt), dim3(kMdBlock), 0, s, x, len, d_pre, d_kept);
        h_tmax.resize((size_t)nt); h_thas.resize((size_t)nt);
        int64_t n_kept = 0;
        URH_HIP(hipMemcpyAsync(&n_kept, d_pre + nt, 8, hipMemcpyDeviceToHost, s));
        URH_HIP(hipMemcpyAsync(h_tmax.data(), d_tmax, (size_t)nt * 8, hipMemcpyDeviceToHost, s));
        URH_HIP(hipMemcpyAsync(h_thas.data(), d_thas, (size_t)nt * 4, hipMemcpyDeviceToHost, s));
        URH_HIP(hipStreamSynchronize(s));
        if (n_kept == 0) continue;                                               // :153-154
        if (len - n_kept > 3) { labels_out[m] = kLabOok; continue; }             // :156-157
        float2 mx = make_float2(0.f, 0.f); bool has = false;
        for (int64_t t = 0; t < nt; ++t)
            if (h_thas[(size_t)t] && (!has || h_tmax[(size_t)t].x > mx.x || (h_tmax[(size_t)t].x == mx.x && h_tmax[(size_t)t].y > mx.y))) { mx = h_tmax[(size_t)t]; has = true; }
        const float scale = hypotf(mx.x, mx.y);                                  // np.abs(np.max(data)) as float32
        int64_t N = 1;
        while (N * 2 <= n_kept) N *= 2;                                          // 2 ** int(log2(len))
        const int64_t skip = 2 * (int64_t)wavelet_scale;
        if (N - 2 * skip <= 0) continue;                                         // empty transform: None (:161-162)
        const int64_t L = N - 2 * skip;
        hipLaunchKernelGGL(k_md_prepare, dim3(md_grid(N)), dim3(kMdBlock), 0, s, d_kept, N, scale, d_c[0], d_c[2]);
        double2 *xa, *xb;
        URH_TRY(md_fft(s, d_c[0], d_c[1], N, -1.0, &xa));
        URH_TRY(md_fft(s, d_c[2], d_c[3], N, -1.0, &xb));
        hipLaunchKernelGGL(k_md_topk, dim3(kMdTopBlocks), dim3(kMdBlock), 0, s, xa, N, d_cv, d_ci);      // spectrum of the normalised samples (:186-187)
        hipLaunchKernelGGL(k_md_wavelet, dim3(md_grid(N)), dim3(kMdBlock), 0, s, xa, xb, N, (double)wavelet_scale);
        double2 *wa, *wb;
        URH_TRY(md_fft(s, xa, xa == d_c[0] ? d_c[1] : d_c[0], N, 1.0, &wa));
        URH_TRY(md_fft(s, xb, xb == d_c[2] ? d_c[3] : d_c[2], N, 1.0, &wb));
        hipLaunchKernelGGL(k_md_mag, dim3(md_grid(L)), dim3(kMdBlock), 0, s, wa, wb, N, skip, d_m1, d_m2);
        hipLaunchKernelGGL(k_md_median, dim3(md_grid(2 * L)), dim3(kMdBlock), 0, s, d_m1, d_m2, L, median_filter_order, d_f1, d_f2);
        const int gb = (int)std::max<int64_t>(1, std::min<int64_t>((L + 4095) / 4096, 4096));
        MdSumArgs sa;
        sa.d[0] = d_m1; sa.d[1] = d_m2; sa.f[0] = d_f1; sa.f[1] = d_f2; sa.L = L; sa.square = 0; sa.part = d_part;
        for (int k = 0; k < 4; ++k) sa.mean[k] = 0.0;
        double var[4];
        for (int pass = 0; pass < 2; ++pass) {
            sa.square = pass;
            hipLaunchKernelGGL(k_md_sum, dim3(gb), dim3(kMdBlock), 0, s, sa);
            URH_HIP(hipMemcpyAsync(h_part.data(), d_part, (size_t)4 * gb * 8, hipMemcpyDeviceToHost, s));
            if (pass == 1) {
                URH_HIP(hipMemcpyAsync(h_cv.data(), d_cv, h_cv.size() * 8, hipMemcpyDeviceToHost, s));
                URH_HIP(hipMemcpyAsync(h_ci.data(), d_ci, h_ci.size() * 8, hipMemcpyDeviceToHost, s));
            }
            URH_HIP(hipStreamSynchronize(s));
            for (int k = 0; k < 4; ++k) {
                double t = 0.0;
                for (int b = 0; b < gb; ++b) t += h_part[(size_t)k * gb + b];
                if (pass == 0) sa.mean[k] = t / (double)L; else var[k] = t / (double)L;
            }
        }
        URH_HIP(hipGetLastError());
        const double var_mag = var[0], var_norm = var[1], var_fmag = var[2], var_fnorm = var[3];
        if (vo) { vo[0] = var_mag; vo[1] = var_norm; vo[2] = var_fmag; vo[3] = var_fnorm; }
        if (var_mag < 0.15 && var_norm < 0.15 && var_fmag < 0.15 && var_fnorm < 0.15) { labels_out[m] = kLabOok; continue; }   // :177-181
        if (var_mag > 1.5 * var_norm) { labels_out[m] = kLabAsk; continue; }     // :183-186
        if (var_mag > 10 * var_fmag) { labels_out[m] = kLabPsk; continue; }      // :189-190
        // FSK, or a single OOK pulse: an FSK spectrum has a second large peak away from the largest one (:192-205)
        std::vector<std::pair<double, int64_t>> cand;
        for (size_t c = 0; c < h_cv.size(); ++c) if (h_ci[c] >= 0) cand.push_back({h_cv[c], h_ci[c]});
        std::sort(cand.begin(), cand.end(), [](const std::pair<double, int64_t> &p, const std::pair<double, int64_t> &q) { return p.first > q.first; });
        bool fsk = false;
        const size_t top = std::min<size_t>(cand.size(), kMdTopK);
        for (size_t c = 0; c < top; ++c) {
            const int64_t dist = cand[c].second > cand[0].second ? cand[c].second - cand[0].second : cand[0].second - cand[c].second;
            if (dist >= 10 && cand[c].first >= 100.0) fsk = true;
        }
        labels_out[m] = fsk ? kLabFsk : kLabOok;
    }
    return URHGPU_OK;
}

}  // extern "C"
