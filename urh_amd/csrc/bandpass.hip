// bandpass.hip -- the spectrogram band-pass (row 6b of SURVEY.md §8a) for gfx950.
//
//   k_bandpass     Filter.apply_bandpass_filter      /root/reference/src/urh/signalprocessing/Filter.py:84-101
//                  np.convolve(data, h, "same") / Filter.fft_convolve_1d (:70-82), both the centred linear convolution
//                  of the complex64 capture with complex128 taps, in complex128.
//
// The reference evaluates this with numpy (BLAS dot products per output, or pocketfft for long filters), whose
// summation order is not defined by the reference; this row is therefore floating point with a tolerance (stated in
// tests/test_gpu_parity.py), not bit-exact.  Here: direct form in fp64, taps ascending, fused multiply-adds.
//
// Roofline: 4 fp64 FMAs per tap per sample against 8 B read + 16 B (complex128) or 8 B (complex64) written per sample;
// the default filter (bw 0.08 -> 51 taps) is 1632 flop per 24 B = 68 flop/B, above the fp64 ridge (78.6 TFLOP/s /
// 8 TB/s = 9.8 flop/B): VALU-bound, so the layout serves the VALU exactly as k_fir does -- every lane owns R consecutive
// outputs and slides a register window over the LDS-staged input, taps arrive through scalar loads.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "common.hpp"
#include "launchers.hpp"

namespace urh {

constexpr int kBpBlock = 256;
constexpr int kBpR = 4;
constexpr int kBpTile = kBpBlock * kBpR;

struct BandpassArgs {
    const float2 *x;         // n samples
    const float2 *left;      // n_left samples preceding x[0] (sharded captures) or nullptr
    const float2 *right;     // n_right samples following x[n-1] or nullptr
    int64_t n_left, n_right;
    const double2 *taps;     // m complex128 taps
    double2 *out128;         // one of the two
    float2 *out64;
    int64_t n;
    int64_t n_out;           // outputs i in [0, n_out)
    int64_t shift;           // out[i] = sum_k taps[k] * X(i + shift - k)
    int m;
    int hist;
};

__device__ __forceinline__ float2 bp_sample(const BandpassArgs &a, int64_t j) {
    if (j >= 0) {
        if (j < a.n) return a.x[j];
        const int64_t r = j - a.n;
        if (a.right != nullptr && r < a.n_right) return a.right[r];
        return make_float2(0.f, 0.f);
    }
    const int64_t l = j + a.n_left;
    if (a.left != nullptr && l >= 0) return a.left[l];
    return make_float2(0.f, 0.f);
}

__global__ __launch_bounds__(kBpBlock) void k_bandpass(const BandpassArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    float2 *s_x = (float2 *)s_raw;                           // [hist + kBpTile], s_x[u] = X(base + shift - hist + u)
    constexpr int R = kBpR;
    const int t = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * kBpTile;
    const int total = a.hist + kBpTile;
    const int64_t j_first = base + a.shift - a.hist;
    for (int u = t; u < total; u += kBpBlock) s_x[u] = bp_sample(a, j_first + u);
    __syncthreads();
    const int64_t i0 = base + (int64_t)R * t;
    if (i0 >= a.n_out) return;
    double2 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_double2(0.0, 0.0);
    // W[u] = X(j0 - kb - (R-1) + u), u in [0, 2R-1); LDS index R*t + hist - kb - (R-1) + u
    float2 W[2 * R - 1];
    const int lds0 = R * t + a.hist - (R - 1);
#pragma unroll
    for (int u = 0; u < R - 1; ++u) W[u] = s_x[lds0 + R + u];   // becomes W[u + R] after the first shift
    for (int kb = 0; kb < a.m; kb += R) {
#pragma unroll
        for (int u = R - 2; u >= 0; --u) W[u + R] = W[u];
#pragma unroll
        for (int u = 0; u < R; ++u) W[u] = s_x[lds0 - kb + u];
#pragma unroll
        for (int tk = 0; tk < R; ++tk) {
            const int k = kb + tk;
            if (k < a.m) {                                   // wave-uniform
                const double2 h = a.taps[k];                 // wave-uniform address: scalar load
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float2 xf = W[r + (R - 1) - tk];
                    const double xr = (double)xf.x, xi = (double)xf.y;
                    acc[r].x = fma(h.x, xr, acc[r].x);
                    acc[r].x = fma(-h.y, xi, acc[r].x);
                    acc[r].y = fma(h.x, xi, acc[r].y);
                    acc[r].y = fma(h.y, xr, acc[r].y);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (i0 + r < a.n_out) {
            if (a.out128 != nullptr) a.out128[i0 + r] = acc[r];
            else a.out64[i0 + r] = make_float2((float)acc[r].x, (float)acc[r].y);
        }
    }
}

// ---- long filters: overlap-save through an FFT that lives in LDS (Filter.fft_convolve_1d, Filter.py:70-82) ----------------------
// The reference switches to ONE FFT over the whole capture when the filter has more taps than 8 ln(sqrt(N)) (75 for a 1 GiB capture;
// the default bandwidth 0.08 gives 51 taps, 0.01 gives 401, 0.001 gives 4001).  Here: blocks of F = 4096 or 8192 samples, one
// workgroup each -- forward FFT (decimation in frequency: natural order in, bit-reversed out), times the taps' spectrum (same layout:
// no permutation anywhere), inverse FFT (decimation in time: bit-reversed in, natural out), of which the F - m + 1 outputs that no
// wrap-around touches are kept.  All in complex128: errors of a few 1e-15 of sum|h| max|x| (the reference's transform is single
// precision for a complex64 capture).  Cost per output: ~2 log2(F) butterflies / (1 - m / F) against m multiply-adds.
constexpr int kBpFftBlock = 1024;
struct BandpassFftArgs {
    BandpassArgs a;
    const double2 *spectrum;  // FFT of the zero-padded taps, bit-reversed layout (nullptr: this launch computes it into spec_out)
    double2 *spec_out;
    const double2 *tw;        // exp(-2 pi i k / F), k < F / 2
    int F, logF;
    int64_t L;                // outputs per block = F - m + 1
};
__device__ __forceinline__ double2 bp_cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__global__ void k_bp_twiddle(double2 *tw, int F) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < F / 2) { double sn, cs; sincospi(-2.0 * (double)k / (double)F, &sn, &cs); tw[k] = make_double2(cs, sn); }
}
__global__ __launch_bounds__(kBpFftBlock) void k_bandpass_fft(const BandpassFftArgs g) {
    extern __shared__ double2 s_f[];
    const BandpassArgs &a = g.a;
    const int F = g.F, t = threadIdx.x;
    const bool taps_pass = g.spectrum == nullptr;
    const int64_t j0 = (int64_t)blockIdx.x * g.L + a.shift - (a.m - 1);       // first input sample of the block
    for (int u = t; u < F; u += kBpFftBlock) {
        if (taps_pass) s_f[u] = (u < a.m) ? a.taps[u] : make_double2(0.0, 0.0);
        else { const float2 v = bp_sample(a, j0 + u); s_f[u] = make_double2((double)v.x, (double)v.y); }
    }
    __syncthreads();
    for (int st = g.logF - 1; st >= 0; --st) {               // decimation in frequency
        const int half = 1 << st, tstep = F >> (st + 1);
        for (int k = t; k < (F >> 1); k += kBpFftBlock) {
            const int j = k & (half - 1);
            const int i0 = ((k >> st) << (st + 1)) + j, i1 = i0 + half;
            const double2 x0 = s_f[i0], x1 = s_f[i1];
            s_f[i0] = make_double2(x0.x + x1.x, x0.y + x1.y);
            s_f[i1] = bp_cmul(make_double2(x0.x - x1.x, x0.y - x1.y), g.tw[j * tstep]);
        }
        __syncthreads();
    }
    if (taps_pass) { for (int u = t; u < F; u += kBpFftBlock) g.spec_out[u] = s_f[u]; return; }
    for (int u = t; u < F; u += kBpFftBlock) s_f[u] = bp_cmul(s_f[u], g.spectrum[u]);
    __syncthreads();
    for (int st = 0; st < g.logF; ++st) {                    // decimation in time, conjugate twiddles: the inverse transform (x F)
        const int half = 1 << st, tstep = F >> (st + 1);
        for (int k = t; k < (F >> 1); k += kBpFftBlock) {
            const int j = k & (half - 1);
            const int i0 = ((k >> st) << (st + 1)) + j, i1 = i0 + half;
            const double2 w = g.tw[j * tstep];
            const double2 x0 = s_f[i0], x1 = bp_cmul(s_f[i1], make_double2(w.x, -w.y));
            s_f[i0] = make_double2(x0.x + x1.x, x0.y + x1.y);
            s_f[i1] = make_double2(x0.x - x1.x, x0.y - x1.y);
        }
        __syncthreads();
    }
    const double inv = 1.0 / (double)F;
    const int64_t i_first = (int64_t)blockIdx.x * g.L;
    for (int64_t u = t; u < g.L; u += kBpFftBlock) {
        const int64_t i = i_first + u;
        if (i >= a.n_out) break;
        const double2 y = s_f[u + a.m - 1];
        if (a.out128 != nullptr) a.out128[i] = make_double2(y.x * inv, y.y * inv);
        else a.out64[i] = make_float2((float)(y.x * inv), (float)(y.y * inv));
    }
}
size_t bandpass_fft_work_bytes() { return (size_t)8192 * 16 + (size_t)4096 * 16 + 512; }

// work: bandpass_fft_work_bytes() of device memory (nullptr: direct form whatever the length)
int launch_bandpass(const float2 *x, int64_t n, const float2 *left, int64_t n_left, const float2 *right, int64_t n_right,
                    const double2 *taps, int m, int64_t shift, int64_t n_out, double2 *out128, float2 *out64, hipStream_t s, void *work) {
    if (n_out <= 0) return URHGPU_OK;
    BandpassArgs a;
    a.x = x; a.left = left; a.right = right; a.n_left = n_left; a.n_right = n_right; a.taps = taps;
    a.out128 = out128; a.out64 = out64; a.n = n; a.n_out = n_out; a.shift = shift; a.m = m;
    if (work != nullptr && m >= 128 && m <= 4097 && n_out >= 4096) {
        // (measured on 2^27 samples: direct form 2.1 ms at 89 taps, this branch 2.8 ms at 101 and 2.9 ms at 401 taps: crossover near 128)
        BandpassFftArgs g;
        g.a = a;
        g.F = (m <= 1025) ? 4096 : 8192;
        g.logF = (g.F == 4096) ? 12 : 13;
        g.L = g.F - m + 1;
        double2 *spec = (double2 *)work, *tw = spec + 8192;
        const size_t lds = (size_t)g.F * sizeof(double2);
        if (hipFuncSetAttribute((const void *)k_bandpass_fft, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return URHGPU_ERR_HIP;
        hipLaunchKernelGGL(k_bp_twiddle, dim3((unsigned)((g.F / 2 + 255) / 256)), dim3(256), 0, s, tw, g.F);
        g.tw = tw; g.spectrum = nullptr; g.spec_out = spec;
        hipLaunchKernelGGL(k_bandpass_fft, dim3(1), dim3(kBpFftBlock), lds, s, g);
        g.spectrum = spec; g.spec_out = nullptr;
        const int64_t blocks = (n_out + g.L - 1) / g.L;
        if (blocks > 0x7fffffff) return URHGPU_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(k_bandpass_fft, dim3((unsigned)blocks), dim3(kBpFftBlock), lds, s, g);
        return URHGPU_OK;
    }
    a.hist = ((std::max(m, 1) - 1) / kBpR) * kBpR + kBpR - 1;
    const size_t lds = (size_t)(a.hist + kBpTile) * 8;
    if (lds > 150 * 1024) return URHGPU_ERR_UNSUPPORTED;       // m <= ~18000 taps (filter_bw >= 0.00023)
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void *)k_bandpass, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return URHGPU_ERR_HIP;
    const int64_t tiles = (n_out + kBpTile - 1) / kBpTile;
    hipLaunchKernelGGL(k_bandpass, dim3((unsigned)tiles), dim3(kBpBlock), lds, s, a);
    return URHGPU_OK;
}

}  // namespace urh
