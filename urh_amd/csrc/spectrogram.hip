// spectrogram.hip -- the Interpretation tab's spectrogram on the GPU
//   Spectrogram.stft                      /root/reference/src/urh/signalprocessing/Spectrogram.py:94-116
//   Spectrogram.__calculate_spectrogram   :158-164   (fftshift along frequency, complex64, util.arr2decibel, fliplr)
//   util.arr2decibel                      /root/reference/src/urh/cythonext/util.pyx:38-48
// One workgroup per frame: window_size (a power of two, 8 .. 4096) complex64 samples times the float64 window, a
// radix-2 decimation-in-time FFT in double precision in LDS (numpy's np.fft.fft of a complex128 array is double
// precision as well), divided by window_size, then either the complex128 spectrum itself (stft) or -- fused -- shifted,
// flipped, rounded to complex64 and turned into decibels (float32), so that the 16 bytes per bin of the reference's
// intermediate never reach HBM: 8 B read and 4 B written per (sample, frame) pair, i.e. 16 + 8 B per sample at the
// default 50 % overlap.  Floating point, not bit-exact against pocketfft's radix-4/8 butterflies: tests state the tolerance.
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "launchers.hpp"

namespace urh {

constexpr int kStftBlock = 256;

__device__ __forceinline__ double2 cmul_d(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// x: complex64 samples, n of them (frames lie inside [0, n); the caller pads as the reference does);
// tw[m] = exp(-2 pi i m / ws) for m < ws/2;  out_c128 (frames, ws) or out_db (frames, ws), exactly one non-null
__global__ __launch_bounds__(kStftBlock) void k_stft(const float2 *x, int64_t n, int ws, int log2ws, int64_t hop, const double *window,
                                                      const double2 *tw, double2 *out_c128, float *out_db) {
    extern __shared__ double2 s_x[];
    const int64_t frame = blockIdx.x;
    const float2 *src = x + frame * hop;
    for (int t = threadIdx.x; t < ws; t += kStftBlock) {
        float2 v = make_float2(0.f, 0.f);
        if (frame * hop + t < n) v = src[t];
        const double w = window[t];
        const int r = (int)(__brev((unsigned)t) >> (32 - log2ws));
        s_x[r] = make_double2((double)v.x * w, (double)v.y * w);
    }
    __syncthreads();
    for (int s = 0; s < log2ws; ++s) {
        const int half = 1 << s, tstep = ws >> (s + 1);
        for (int k = threadIdx.x; k < (ws >> 1); k += kStftBlock) {
            const int j = k & (half - 1);
            const int i0 = ((k >> s) << (s + 1)) + j, i1 = i0 + half;
            const double2 a = s_x[i0], b = cmul_d(s_x[i1], tw[j * tstep]);
            s_x[i0] = make_double2(a.x + b.x, a.y + b.y);
            s_x[i1] = make_double2(a.x - b.x, a.y - b.y);
        }
        __syncthreads();
    }
    const double inv = 1.0 / (double)ws;
    if (out_c128) {
        for (int t = threadIdx.x; t < ws; t += kStftBlock) out_c128[frame * ws + t] = make_double2(s_x[t].x * inv, s_x[t].y * inv);
    } else {
        for (int f = threadIdx.x; f < ws; f += kStftBlock) {
            const int shifted = ws - 1 - f;                         // np.fliplr
            const int k = (shifted + (ws >> 1)) & (ws - 1);         // np.fft.fftshift(axes=1)
            const float re = (float)(s_x[k].x * inv), im = (float)(s_x[k].y * inv);   // .astype(np.complex64)
            out_db[frame * ws + f] = 10.0f * log10f(re * re + im * im);               // util.pyx:47
        }
    }
}

int launch_stft(const float2 *x, int64_t n, int ws, int64_t hop, int64_t frames, const double *window, const double2 *tw,
                double2 *out_c128, float *out_db, hipStream_t s) {
    int l = 0;
    while ((1 << l) < ws) ++l;
    if ((1 << l) != ws || ws < 8 || ws > 4096 || hop < 1 || frames < 0 || frames > 0x7fffffff) return URHGPU_ERR_UNSUPPORTED;
    if (frames == 0) return URHGPU_OK;
    hipLaunchKernelGGL(k_stft, dim3((unsigned)frames), dim3(kStftBlock), (size_t)ws * sizeof(double2), s, x, n, ws, l, hop, window, tw,
                       out_c128, out_db);
    return URHGPU_OK;
}

// Spectrogram.apply_bgra_lookup (:196-210), normalize=True: index = int((len - 1) * ((data.T - min) / (max - min))) clipped
// (np.take mode="clip"), image[f][t] = colormap[index]; data is (frames, ws) float32, image (ws, frames) x 4 bytes.
__global__ void k_bgra_lookup(const float *data, int64_t frames, int ws, const uint32_t *colormap, int n_colors, float dmin, float dmax,
                              uint32_t *image) {
    const int64_t total = frames * ws;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = i / ws; const int f = (int)(i - t * ws);
        // numpy evaluates in float32 (data is float32, the Python scalars do not widen it)
        const float v = (float)(n_colors - 1) * ((data[i] - dmin) / (dmax - dmin));
        long long idx;
        if (v != v) idx = (long long)0x8000000000000000ull;        // astype(int) of NaN / +-inf: the x86-64 "indefinite" value
        else if (v >= 9.2233720368547758e18f || v <= -9.2233720368547758e18f) idx = (long long)0x8000000000000000ull;
        else idx = (long long)v;                                     // truncation toward zero
        const int c = idx < 0 ? 0 : (idx > n_colors - 1 ? n_colors - 1 : (int)idx);
        image[(int64_t)f * frames + t] = colormap[c];
    }
}

int launch_bgra_lookup(const float *data, int64_t frames, int ws, const uint32_t *colormap, int n_colors, float dmin, float dmax,
                       uint32_t *image, hipStream_t s) {
    if (frames <= 0 || ws <= 0) return URHGPU_OK;
    if (n_colors < 1) return URHGPU_ERR_ARG;
    int64_t g = (frames * ws + 255) / 256; if (g > 65536) g = 65536;
    hipLaunchKernelGGL(k_bgra_lookup, dim3((unsigned)g), dim3(256), 0, s, data, frames, ws, colormap, n_colors, dmin, dmax, image);
    return URHGPU_OK;
}

}  // namespace urh
