// modulate.hip -- signal_functions.modulate_c on the GPU (the generator on the other side of the IQ->bits path)
//   /root/reference/src/urh/cythonext/signal_functions.pyx:56-177   (ASK, FSK, PSK, OQPSK)
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
        default: return URHGPU_ERR_UNSUPPORTED;
    }
}

int launch_modulate(const ModArgs &a, int64_t max_samples, hipStream_t s) {
    if (a.n_msgs <= 0) return URHGPU_OK;
    if (a.n_msgs > 65535) return URHGPU_ERR_ARG;
    if (a.mod == URHGPU_MOD_FSK) hipLaunchKernelGGL(k_mod_phase, dim3((unsigned)((a.n_msgs + 63) / 64)), dim3(64), 0, s, a);
    switch (a.dtype) {
        case URHGPU_DT_F32: return launch_mod_t<float>(a, max_samples, s);
        case URHGPU_DT_I8: return launch_mod_t<int8_t>(a, max_samples, s);
        case URHGPU_DT_I16: return launch_mod_t<int16_t>(a, max_samples, s);
        default: return URHGPU_ERR_DTYPE;
    }
}

}  // namespace urh
