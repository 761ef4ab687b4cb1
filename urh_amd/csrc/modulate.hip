// modulate.hip -- signal_functions.modulate_c on the GPU (the generator on the other side of the IQ->bits path)
//   /root/reference/src/urh/cythonext/signal_functions.pyx:56-177   (ASK, FSK, PSK, OQPSK), :196-228 (GFSK)
//
// Batched: URH modulates message by message (Modulator.modulate, ProtocolAnalyzerContainer.modulate), every message
// with its own bits / pause / start sample; one launch renders any number of messages back to back.
//
//   k_mod_phase   FSK only: the phase corrections that keep the carrier continuous across symbol boundaries
//                 (:121-137) are a serial recurrence with a float rounding and an fmod in every step -- one thread per
//                 MESSAGE walks its symbols;
//   k_modulate    one thread per sample: t = (float)(i + start) / sample_rate, carrier argument in double, rounded to
//                 float, glibc's sinf / cosf (glibc_sincosf.h: the argument reaches 10^5 rad, so the large-argument
//                 reduction is on the path), amplitude, C cast to the sample type.  Pauses and zero-amplitude ASK
//                 symbols are written as zeros (the reference starts from np.zeros).
//   k_gfsk_freq   GFSK: the per-sample symbol frequencies convolved with the Gaussian taps ("same" mode), every output an
//                 exactly accumulated (fp64) dot product rounded once to float32.  The reference gets this value from
//                 numpy's float32 BLAS dot, whose summation order is a property of the host CPU: callers who need the
//                 reference's bits on their host pass numpy's convolution in (freq_given) and this kernel is skipped;
//   k_gfsk_phase  GFSK: phases[i+1] = (float)(2 pi t[i] (f[i] - f[i+1]) + phases[i]) -- a float rounding in every step, one
//                 wavefront per message walks its samples (t = numpy's float32 arange, restated);
// Roofline: the sample kernel writes 8 B (complex64) per sample against ~100 fp64 operations: HBM write bandwidth and
// the fp64 vector rate are about level; the phase kernel is latency (one dependent fmod per symbol per message).
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "glibc_sincosf.h"
#include "launchers.hpp"

namespace urh {

__device__ __forceinline__ uint32_t mod_symbol_index(const uint8_t *bits, int64_t sym, int bps) {   // util.pyx:50-61
    uint32_t r = 0;
    const uint8_t *b = bits + sym * bps;
    for (int k = 0; k < bps; ++k) r = (r << 1) + b[k];
    return r;
}

// one thread per message
__global__ void k_mod_phase(const ModArgs a) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= a.n_msgs) return;
    const ModMsg g = a.msgs[m];
    const uint8_t *bits = a.bits + g.bit_off;
    float *pc = a.phase + g.sym_off;
    if (g.n_sym <= 0) return;
    const double two_pi = 2.0 * 3.14159265358979323846;
    float prev = 0.0f;
    pc[0] = 0.0f;
    float fp = a.params[mod_symbol_index(bits, 0, a.bps)];
    for (int64_t s = 1; s < g.n_sym; ++s) {
        const float f = a.params[mod_symbol_index(bits, s, a.bps)];
        if (f != fp) {
            const float t = ((float)(((s * (int64_t)a.sps) + (int64_t)g.start) - 1)) / a.sample_rate;
            prev = (float)fmod((double)prev + ((two_pi * (double)(fp - f)) * (double)t), two_pi);
        }
        pc[s] = prev;
        fp = f;
    }
}

// GFSK, :196-217: one thread per data sample, grid.y = message
__global__ __launch_bounds__(256) void k_gfsk_freq(const ModArgs a) {
    const ModMsg g = a.msgs[blockIdx.y];
    const uint8_t *bits = a.bits + g.bit_off;
    const int64_t n = g.n_sym * (int64_t)a.sps, m = a.n_taps;
    float *f = a.gf_freq + g.sym_off * (int64_t)a.sps;
    const int64_t off = (n >= m) ? (m - 1) / 2 : (n - 1) / 2;                  // "same": centred on the longer operand
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = i + off;
        const int64_t lo = k - m + 1 > 0 ? k - m + 1 : 0, hi = k < n - 1 ? k : n - 1;
        int64_t sym = lo / a.sps;
        uint32_t left = (uint32_t)((sym + 1) * (int64_t)a.sps - lo);           // samples of this symbol from lo on
        float val = a.params[mod_symbol_index(bits, sym, a.bps)];
        double acc = 0.0;
        for (int64_t j = lo; j <= hi; ++j) {
            acc += (double)val * (double)a.taps[k - j];
            if (--left == 0 && j < hi) { ++sym; left = a.sps; val = a.params[mod_symbol_index(bits, sym, a.bps)]; }
        }
        f[i] = (float)acc;
    }
}

// GFSK, :219-226: one wavefront per message.  Step i (phases[i] -> phases[i+1]) needs d_i = (2 pi t[i]) * (f[i] - f[i+1]) -- 64
// lanes evaluate 64 of them at once (division, products, coalesced loads) -- and then the one thing that is serial: the
// float rounding of the running sum, done uniformly in all lanes with d_i read from lane i (add, two conversions per step).
__device__ __forceinline__ double gfsk_readlane(double v, int k) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), k), hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
    return __hiloint2double(hi, lo);
}
__global__ __launch_bounds__(64) void k_gfsk_phase(const ModArgs a) {
    const ModMsg g = a.msgs[blockIdx.x];
    const int lane = threadIdx.x;
    const int64_t n = g.n_sym * (int64_t)a.sps;
    if (n <= 0) return;
    const float *__restrict__ f = a.gf_freq + g.sym_off * (int64_t)a.sps;
    float *__restrict__ ph = a.gf_phase + g.sym_off * (int64_t)a.sps;
    const double two_pi = 2.0 * 3.14159265358979323846;
    // np.arange(start, start + n, dtype=float32): buf[0] = start, buf[1] = start + 1, buf[i] = buf[0] + i * (buf[1] - buf[0])
    const float s0 = (float)(double)g.start, s1 = (float)((double)g.start + 1.0), delta = s1 - s0;
    float cur = a.carrier_phase;
    if (lane == 0) ph[0] = cur;
    for (int64_t base = 0; base + 1 < n; base += 64) {
        const int64_t i = base + lane;
        double d = 0.0;
        if (i + 1 < n) {
            const float prod = (float)i * delta;
            const float ti = (i == 0) ? s0 : ((i == 1) ? s1 : s0 + prod);
            const float t = ti / a.sample_rate;
            const float df = f[i] - f[i + 1];
            d = (two_pi * (double)t) * (double)df;
        }
        float mine = 0.0f;
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            cur = (float)(gfsk_readlane(d, k) + (double)cur);
            if (lane == k) mine = cur;
        }
        if (i + 1 < n) ph[i + 1] = mine;
    }
}

template <typename T> __device__ __forceinline__ T mod_cast(float v);
template <> __device__ __forceinline__ float mod_cast<float>(float v) { return v; }
// (char)float / (short)float as x86-64 evaluates them: truncating conversion to int32, low bits kept
template <> __device__ __forceinline__ int8_t mod_cast<int8_t>(float v) {
    const int i = (v >= -2147483648.0f && v < 2147483648.0f) ? (int)v : (int)0x80000000;
    return (int8_t)i;
}
template <> __device__ __forceinline__ int16_t mod_cast<int16_t>(float v) {
    const int i = (v >= -2147483648.0f && v < 2147483648.0f) ? (int)v : (int)0x80000000;
    return (int16_t)i;
}

template <typename T> struct Pair { T x, y; };

// blockIdx.y = message, grid-stride over its samples
template <typename T, int MOD>
__global__ __launch_bounds__(256) void k_modulate(const ModArgs a) {
    const ModMsg g = a.msgs[blockIdx.y];
    const uint8_t *bits = a.bits + g.bit_off;
    Pair<T> *out = (Pair<T> *)a.out + g.out_off;
    const int64_t n_data = g.n_sym * (int64_t)a.sps, n_total = n_data + g.pause;
    const double two_pi = 2.0 * 3.14159265358979323846;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_total; i += (int64_t)gridDim.x * blockDim.x) {
        Pair<T> o; o.x = (T)0; o.y = (T)0;
        if (i < n_data) {
            const int64_t s = i / a.sps;
            const uint32_t index = mod_symbol_index(bits, s, a.bps);
            float amp = a.carrier_amplitude, f = a.carrier_frequency, phi = a.carrier_phase, corr = 0.0f;
            if (MOD == URHGPU_MOD_ASK) amp = a.params[index];
            else if (MOD == URHGPU_MOD_FSK) { f = a.params[index]; corr = a.phase[g.sym_off + s]; }
            else if (MOD == kModGfsk) { f = a.gf_freq[g.sym_off * (int64_t)a.sps + i]; phi = a.gf_phase[g.sym_off * (int64_t)a.sps + i]; }
            else phi = a.params[index];
            if (!(MOD == URHGPU_MOD_ASK && amp == 0.0f)) {
                const float t = ((float)(i + (int64_t)g.start)) / a.sample_rate;
                const float arg = (float)(((((two_pi) * (double)f) * (double)t) + (double)phi) + (double)corr);
                o.x = mod_cast<T>(amp * urh_cosf(arg));
                o.y = mod_cast<T>(amp * urh_sinf(arg));
            }
            // OQPSK (:165-169): Q of the first symbol and I of the last one are blanked
            if (MOD == URHGPU_MOD_PSK && a.oqpsk) {
                if (i < (int64_t)a.sps) o.y = (T)0;
                if (i >= n_data - (int64_t)a.sps) o.x = (T)0;
            }
        }
        out[i] = o;
    }
}

template <typename T>
static int launch_mod_t(const ModArgs &a, int64_t max_samples, hipStream_t s) {
    int64_t gx = (max_samples + 255) / 256;
    if (gx > 8192) gx = 8192;
    if (gx < 1) gx = 1;
    const dim3 grid((unsigned)gx, (unsigned)a.n_msgs), block(256);
    switch (a.mod) {
        case URHGPU_MOD_ASK: hipLaunchKernelGGL((k_modulate<T, URHGPU_MOD_ASK>), grid, block, 0, s, a); return URHGPU_OK;
        case URHGPU_MOD_FSK: hipLaunchKernelGGL((k_modulate<T, URHGPU_MOD_FSK>), grid, block, 0, s, a); return URHGPU_OK;
        case URHGPU_MOD_PSK: hipLaunchKernelGGL((k_modulate<T, URHGPU_MOD_PSK>), grid, block, 0, s, a); return URHGPU_OK;
        case kModGfsk: hipLaunchKernelGGL((k_modulate<T, kModGfsk>), grid, block, 0, s, a); return URHGPU_OK;
        default: return URHGPU_ERR_UNSUPPORTED;
    }
}

int launch_modulate(const ModArgs &a, int64_t max_samples, hipStream_t s) {
    if (a.n_msgs <= 0) return URHGPU_OK;
    if (a.n_msgs > 65535) return URHGPU_ERR_ARG;
    if (a.mod == URHGPU_MOD_FSK) hipLaunchKernelGGL(k_mod_phase, dim3((unsigned)((a.n_msgs + 63) / 64)), dim3(64), 0, s, a);
    if (a.mod == kModGfsk) {
        if (!a.freq_given) {
            int64_t gx = (max_samples + 255) / 256;
            gx = gx > 16384 ? 16384 : (gx < 1 ? 1 : gx);
            hipLaunchKernelGGL(k_gfsk_freq, dim3((unsigned)gx, (unsigned)a.n_msgs), dim3(256), 0, s, a);
        }
        hipLaunchKernelGGL(k_gfsk_phase, dim3((unsigned)a.n_msgs), dim3(64), 0, s, a);
    }
    switch (a.dtype) {
        case URHGPU_DT_F32: return launch_mod_t<float>(a, max_samples, s);
        case URHGPU_DT_I8: return launch_mod_t<int8_t>(a, max_samples, s);
        case URHGPU_DT_I16: return launch_mod_t<int16_t>(a, max_samples, s);
        default: return URHGPU_ERR_DTYPE;
    }
}

}  // namespace urh
