// fft_peak.hip -- Signal.estimate_frequency (Signal.py:578-601) for a capture in HBM: the bin of the largest |FFT| over a power-of-two
// window of complex64 samples (the reference: np.fft.fft, np.argmax(np.abs(w)), fftfreq -- the modulation dialog's carrier guess).
// Windows run up to the whole capture (2^26 samples and more), so this is a general radix-2 FFT, not the window-sized ones of the
// spectrogram / band-pass:
//   n <= 8192        one workgroup, Stockham autosort stages in LDS (two buffers), twiddles from a 4096-entry table
//   larger n = n1 n2 four-step: transpose, n2 row FFTs of length n1 with the twiddle exp(-2 pi i j2 k1 / n) folded into their stores,
//                    transpose, n1 row FFTs of length n2; X[k1 + n1 k2] = E[k1][k2].  Two transposes + two FFT passes: 8 x 8 B per sample
//                    of HBM traffic, every pass coalesced; the FFT passes are LDS-bound (log2 L stages of 16 B per butterfly).
// Arithmetic: float32 butterflies (numpy's complex64 FFT is single precision too), twiddles rounded from float64 sincospi.  The result is
// the argmax bin, not the spectrum: equal to numpy's whenever the peak stands out by more than float32 rounding (first index on ties).
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "launchers.hpp"

namespace urh {

constexpr int kFpMaxLog2 = 13, kFpMaxLen = 1 << kFpMaxLog2, kFpBlock = 256;

__global__ void k_fp_twiddles(float2 *tw, int n_half, int n_full) {       // tw[j] = exp(-2 pi i j / n_full), j < n_half
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_half) return;
    double s, c;
    sincospi(-2.0 * (double)j / (double)n_full, &s, &c);
    tw[j] = float2{(float)c, (float)s};
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return float2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

// one workgroup per row of length L = 2^log2L (<= 8192): Stockham autosort, decimation in frequency -- stage with sub-length n, stride s:
// (a, b) = src[q + s p], src[q + s (p + n/2)] -> dst[q + 2 s p] = a + b, dst[q + s (2 p + 1)] = (a - b) w_n^p.  post_log2n > 0: the
// four-step twiddle exp(-2 pi i row col / 2^post_log2n) on the way out.
__global__ __launch_bounds__(kFpBlock) void k_fp_row_fft(const float2 *in, float2 *out, int log2L, const float2 *tw, int tw_log2, int post_log2n) {
    extern __shared__ float2 s_buf[];
    const int L = 1 << log2L;
    const int64_t row = blockIdx.x;
    const float2 *x = in + row * (int64_t)L;
    float2 *src = s_buf, *dst = s_buf + L;
    for (int i = threadIdx.x; i < L; i += kFpBlock) src[i] = x[i];
    __syncthreads();
    int log2s = 0;
    for (int log2n = log2L; log2n > 0; --log2n, ++log2s) {
        const int m = 1 << (log2n - 1), s = 1 << log2s;
        for (int idx = threadIdx.x; idx < (L >> 1); idx += kFpBlock) {
            const int p = idx >> log2s, q = idx & (s - 1);
            const float2 w = tw[(int64_t)p << (tw_log2 - log2n)];           // exp(-2 pi i p / n) from the table over 2^tw_log2
            const float2 a = src[q + s * p], b = src[q + s * (p + m)];
            dst[q + s * (2 * p)] = float2{a.x + b.x, a.y + b.y};
            dst[q + s * (2 * p + 1)] = cmul(float2{a.x - b.x, a.y - b.y}, w);
        }
        __syncthreads();
        float2 *t = src; src = dst; dst = t;
    }
    float2 *y = out + row * (int64_t)L;
    if (post_log2n > 0) {
        const int64_t mask = ((int64_t)1 << post_log2n) - 1;
        const double inv = 1.0 / (double)((int64_t)1 << post_log2n);
        for (int i = threadIdx.x; i < L; i += kFpBlock) {
            double sn, cs;
            sincospi(-2.0 * (double)((row * (int64_t)i) & mask) * inv, &sn, &cs);
            y[i] = cmul(src[i], float2{(float)cs, (float)sn});
        }
    } else {
        for (int i = threadIdx.x; i < L; i += kFpBlock) y[i] = src[i];
    }
}

// out[c][r] = in[r][c]; rows, cols multiples of 32
__global__ __launch_bounds__(256) void k_fp_transpose(const float2 *in, float2 *out, int64_t rows, int64_t cols) {
    __shared__ float2 tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
    for (int k = ty; k < 32; k += 8) tile[k][tx] = in[(r0 + k) * cols + c0 + tx];
    __syncthreads();
    for (int k = ty; k < 32; k += 8) out[(c0 + k) * rows + r0 + tx] = tile[tx][k];
}

// |E[e]| as numpy's float32 np.abs, the largest with the smallest frequency index k = k1 + n1 k2 (e = k1 n2 + k2)
struct FpBest { float mag; int64_t k; };
__device__ __forceinline__ bool fp_better(float m, int64_t k, float bm, int64_t bk) { return m > bm || (m == bm && k < bk); }

__global__ __launch_bounds__(256) void k_fp_argmax(const float2 *e, int64_t n, int log2n2, int log2n1, float *part_mag, int64_t *part_k) {
    __shared__ float s_m[256];
    __shared__ int64_t s_k[256];
    float bm = -1.f; int64_t bk = INT64_MAX;
    const int64_t mask2 = ((int64_t)1 << log2n2) - 1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float2 v = e[i];
        const float mag = (float)sqrt((double)v.x * (double)v.x + (double)v.y * (double)v.y);
        const int64_t k = (i >> log2n2) + ((i & mask2) << log2n1);
        if (fp_better(mag, k, bm, bk)) { bm = mag; bk = k; }
    }
    s_m[threadIdx.x] = bm; s_k[threadIdx.x] = bk;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o && fp_better(s_m[threadIdx.x + o], s_k[threadIdx.x + o], s_m[threadIdx.x], s_k[threadIdx.x])) {
            s_m[threadIdx.x] = s_m[threadIdx.x + o]; s_k[threadIdx.x] = s_k[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part_mag[blockIdx.x] = s_m[0]; part_k[blockIdx.x] = s_k[0]; }
}
__global__ __launch_bounds__(256) void k_fp_argmax_fin(const float *part_mag, const int64_t *part_k, int n_parts, int64_t *out) {
    __shared__ float s_m[256];
    __shared__ int64_t s_k[256];
    float bm = -1.f; int64_t bk = INT64_MAX;
    for (int i = threadIdx.x; i < n_parts; i += 256)
        if (fp_better(part_mag[i], part_k[i], bm, bk)) { bm = part_mag[i]; bk = part_k[i]; }
    s_m[threadIdx.x] = bm; s_k[threadIdx.x] = bk;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o && fp_better(s_m[threadIdx.x + o], s_k[threadIdx.x + o], s_m[threadIdx.x], s_k[threadIdx.x])) {
            s_m[threadIdx.x] = s_m[threadIdx.x + o]; s_k[threadIdx.x] = s_k[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = (s_k[0] == INT64_MAX) ? 0 : s_k[0];      // (every magnitude NaN: bin 0, as np.argmax of an all-NaN array)
}

size_t fft_peak_scratch_bytes(int64_t n) {
    return (size_t)n * 8 * 2 + (size_t)(kFpMaxLen / 2) * 8 + 1024 * (4 + 8) + 8 + 8 * 256;
}

// x: n = 2^log2n complex64 samples; scratch: fft_peak_scratch_bytes(n); d_peak: the bin (device)
int launch_fft_peak(const float2 *x, int log2n, void *scratch, int64_t *d_peak, hipStream_t s) {
    const int64_t n = (int64_t)1 << log2n;
    char *p = (char *)scratch;
    auto take = [&](size_t b) { char *r = p; p += (b + 255) & ~size_t(255); return r; };
    float2 *b0 = (float2 *)take((size_t)n * 8), *b1 = (float2 *)take((size_t)n * 8);
    float2 *tw = (float2 *)take((size_t)(kFpMaxLen / 2) * 8);
    float *part_mag = (float *)take(1024 * 4);
    int64_t *part_k = (int64_t *)take(1024 * 8);
    // (the attribute is per DEVICE: set before every use -- a process-wide "done" flag left a second GPU's launches without the opt-in)
    if (hipFuncSetAttribute((const void *)k_fp_row_fft, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kFpMaxLen * (int)sizeof(float2)) != hipSuccess)
        return URHGPU_ERR_HIP;
    hipLaunchKernelGGL(k_fp_twiddles, dim3(kFpMaxLen / 2 / 256), dim3(256), 0, s, tw, kFpMaxLen / 2, kFpMaxLen);
    const float2 *result;
    int log2n1 = 0, log2n2 = log2n;       // e = k1 n2 + k2 with n1 = 1: k = k2 = e
    if (log2n <= kFpMaxLog2) {
        hipLaunchKernelGGL(k_fp_row_fft, dim3(1), dim3(kFpBlock), (size_t)2 * n * sizeof(float2), s, x, b0, log2n, tw, kFpMaxLog2, 0);
        result = b0;
    } else {
        log2n1 = (log2n + 1) / 2; log2n2 = log2n - log2n1;
        const int64_t n1 = (int64_t)1 << log2n1, n2 = (int64_t)1 << log2n2;
        if (log2n1 > kFpMaxLog2) return URHGPU_ERR_UNSUPPORTED;
        // A[j1][j2] (n1 x n2) -> B[j2][j1]
        hipLaunchKernelGGL(k_fp_transpose, dim3((unsigned)(n2 / 32), (unsigned)(n1 / 32)), dim3(256), 0, s, x, b0, n1, n2);
        // C[j2][k1] = FFT over j1, times exp(-2 pi i j2 k1 / n)
        hipLaunchKernelGGL(k_fp_row_fft, dim3((unsigned)n2), dim3(kFpBlock), (size_t)2 * n1 * sizeof(float2), s, b0, b1, log2n1, tw, kFpMaxLog2, log2n);
        // D[k1][j2]
        hipLaunchKernelGGL(k_fp_transpose, dim3((unsigned)(n1 / 32), (unsigned)(n2 / 32)), dim3(256), 0, s, b1, b0, n2, n1);
        // E[k1][k2] = FFT over j2
        hipLaunchKernelGGL(k_fp_row_fft, dim3((unsigned)n1), dim3(kFpBlock), (size_t)2 * n2 * sizeof(float2), s, b0, b1, log2n2, tw, kFpMaxLog2, 0);
        result = b1;
    }
    int parts = (int)std::min<int64_t>(1024, (n + 255) / 256);
    hipLaunchKernelGGL(k_fp_argmax, dim3((unsigned)parts), dim3(256), 0, s, result, n, log2n2, log2n1, part_mag, part_k);
    hipLaunchKernelGGL(k_fp_argmax_fin, dim3(1), dim3(256), 0, s, part_mag, part_k, parts, d_peak);
    return URHGPU_OK;
}

}  // namespace urh
