// convert.hip -- IQArray.convert_to on the GPU: the sample-type conversions behind IQArray.from_file (.cu8 / .cu16 captures
// become signed on load), as_complex64 (spectrogram, filters, modulation detection) and the export paths
//   /root/reference/src/urh/signalprocessing/IQArray.py:127-203
// Elementwise, HBM bound (1-4 B read, 1-4 B written per value).  Integer conversions are the reference's wrapping
// numpy operations (np.add(..., dtype=, casting="unsafe"), shifts); float32 -> integer is numpy's astype = the C cast
// as x86-64 evaluates it (truncation toward zero through int32, low bits kept).
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "launchers.hpp"

namespace urh {

__device__ __forceinline__ int cvt_trunc_i32(float v) {
    return (v >= -2147483648.0f && v < 2147483648.0f) ? (int)v : (int)0x80000000;
}

template <typename S, typename D> struct Conv;
// ---- from uint8 (:131-143)
template <> struct Conv<uint8_t, int8_t> { static __device__ int8_t f(uint8_t v) { return (int8_t)(uint8_t)(v + 128u); } };
template <> struct Conv<uint8_t, int16_t> { static __device__ int16_t f(uint8_t v) { return (int16_t)((uint16_t)(int16_t)((int)v - 128) << 8); } };
template <> struct Conv<uint8_t, uint16_t> { static __device__ uint16_t f(uint8_t v) { return (uint16_t)((uint16_t)v << 8); } };
template <> struct Conv<uint8_t, float> { static __device__ float f(uint8_t v) { return (float)v * (1.0f / 128.0f) + -1.0f; } };
// ---- from int8 (:145-153)
template <> struct Conv<int8_t, uint8_t> { static __device__ uint8_t f(int8_t v) { return (uint8_t)((uint8_t)v + 128u); } };
template <> struct Conv<int8_t, int16_t> { static __device__ int16_t f(int8_t v) { return (int16_t)((uint16_t)(int16_t)v << 8); } };
template <> struct Conv<int8_t, uint16_t> { static __device__ uint16_t f(int8_t v) { return (uint16_t)((uint16_t)((uint16_t)(int16_t)v + 128u) << 8); } };
template <> struct Conv<int8_t, float> { static __device__ float f(int8_t v) { return (float)v * (1.0f / 128.0f); } };
// ---- from uint16 (:155-169)
template <> struct Conv<uint16_t, int8_t> { static __device__ int8_t f(uint16_t v) { return (int8_t)((int16_t)(uint16_t)(v + 32768u) >> 8); } };
template <> struct Conv<uint16_t, uint8_t> { static __device__ uint8_t f(uint16_t v) { return (uint8_t)(v >> 8); } };
template <> struct Conv<uint16_t, int16_t> { static __device__ int16_t f(uint16_t v) { return (int16_t)(uint16_t)(v + 32768u); } };
template <> struct Conv<uint16_t, float> { static __device__ float f(uint16_t v) { return (float)v * (1.0f / 32768.0f) + -1.0f; } };
// ---- from int16 (:171-181)
template <> struct Conv<int16_t, int8_t> { static __device__ int8_t f(int16_t v) { return (int8_t)(v >> 8); } };
template <> struct Conv<int16_t, uint8_t> { static __device__ uint8_t f(int16_t v) { return (uint8_t)((uint16_t)((uint16_t)v + 32768u) >> 8); } };
template <> struct Conv<int16_t, uint16_t> { static __device__ uint16_t f(int16_t v) { return (uint16_t)((uint16_t)v + 32768u); } };
template <> struct Conv<int16_t, float> { static __device__ float f(int16_t v) { return (float)v * (1.0f / 32768.0f); } };
// ---- from float32 (:183-196)
template <> struct Conv<float, int8_t> { static __device__ int8_t f(float v) { return (int8_t)cvt_trunc_i32(v * 127.0f); } };
template <> struct Conv<float, uint8_t> { static __device__ uint8_t f(float v) { return (uint8_t)cvt_trunc_i32((v + 1.0f) * 127.0f); } };
template <> struct Conv<float, int16_t> { static __device__ int16_t f(float v) { return (int16_t)cvt_trunc_i32(v * 32767.0f); } };
template <> struct Conv<float, uint16_t> { static __device__ uint16_t f(float v) { return (uint16_t)cvt_trunc_i32((v + 1.0f) * 32767.0f); } };

template <typename S, typename D>
__global__ __launch_bounds__(256) void k_convert(const S *src, D *dst, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = Conv<S, D>::f(src[i]);
}

template <typename S, typename D>
static int conv_launch(const void *src, void *dst, int64_t n, hipStream_t s) {
    int64_t g = (n + 255) / 256; if (g > 65536) g = 65536;
    hipLaunchKernelGGL((k_convert<S, D>), dim3((unsigned)g), dim3(256), 0, s, (const S *)src, (D *)dst, n);
    return URHGPU_OK;
}

template <typename S>
static int conv_from(const void *src, int dst_dtype, void *dst, int64_t n, hipStream_t s);
#define URH_CONV_CASE(S, code, D) case code: return conv_launch<S, D>(src, dst, n, s);
template <> int conv_from<uint8_t>(const void *src, int d, void *dst, int64_t n, hipStream_t s) {
    switch (d) { URH_CONV_CASE(uint8_t, URHGPU_DT_I8, int8_t) URH_CONV_CASE(uint8_t, URHGPU_DT_I16, int16_t) URH_CONV_CASE(uint8_t, URHGPU_DT_U16, uint16_t)
                 URH_CONV_CASE(uint8_t, URHGPU_DT_F32, float) default: return URHGPU_ERR_DTYPE; }
}
template <> int conv_from<int8_t>(const void *src, int d, void *dst, int64_t n, hipStream_t s) {
    switch (d) { URH_CONV_CASE(int8_t, URHGPU_DT_U8, uint8_t) URH_CONV_CASE(int8_t, URHGPU_DT_I16, int16_t) URH_CONV_CASE(int8_t, URHGPU_DT_U16, uint16_t)
                 URH_CONV_CASE(int8_t, URHGPU_DT_F32, float) default: return URHGPU_ERR_DTYPE; }
}
template <> int conv_from<uint16_t>(const void *src, int d, void *dst, int64_t n, hipStream_t s) {
    switch (d) { URH_CONV_CASE(uint16_t, URHGPU_DT_I8, int8_t) URH_CONV_CASE(uint16_t, URHGPU_DT_U8, uint8_t) URH_CONV_CASE(uint16_t, URHGPU_DT_I16, int16_t)
                 URH_CONV_CASE(uint16_t, URHGPU_DT_F32, float) default: return URHGPU_ERR_DTYPE; }
}
template <> int conv_from<int16_t>(const void *src, int d, void *dst, int64_t n, hipStream_t s) {
    switch (d) { URH_CONV_CASE(int16_t, URHGPU_DT_I8, int8_t) URH_CONV_CASE(int16_t, URHGPU_DT_U8, uint8_t) URH_CONV_CASE(int16_t, URHGPU_DT_U16, uint16_t)
                 URH_CONV_CASE(int16_t, URHGPU_DT_F32, float) default: return URHGPU_ERR_DTYPE; }
}
template <> int conv_from<float>(const void *src, int d, void *dst, int64_t n, hipStream_t s) {
    switch (d) { URH_CONV_CASE(float, URHGPU_DT_I8, int8_t) URH_CONV_CASE(float, URHGPU_DT_U8, uint8_t) URH_CONV_CASE(float, URHGPU_DT_I16, int16_t)
                 URH_CONV_CASE(float, URHGPU_DT_U16, uint16_t) default: return URHGPU_ERR_DTYPE; }
}

// ---- plain numpy casts (ndarray.astype / assignment into an array of another type, no IQArray scaling) -------------------
// integer -> float32 is exact; float32 -> integer is the C cast as x86-64 evaluates it (truncation toward zero through
// int32, low bits kept) -- what `IQArray.__setitem__` does with a filtered complex64 range (IQArray.py:31-33).
template <typename S, typename D> struct Cast { static __device__ D f(S v) { return (D)v; } };
template <typename D> struct Cast<float, D> { static __device__ D f(float v) { return (D)cvt_trunc_i32(v); } };
template <> struct Cast<float, float> { static __device__ float f(float v) { return v; } };

template <typename S, typename D>
__global__ __launch_bounds__(256) void k_astype(const S *src, D *dst, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = Cast<S, D>::f(src[i]);
}
template <typename S, typename D>
static int astype_launch(const void *src, void *dst, int64_t n, hipStream_t s) {
    int64_t g = (n + 255) / 256; if (g > 65536) g = 65536;
    hipLaunchKernelGGL((k_astype<S, D>), dim3((unsigned)g), dim3(256), 0, s, (const S *)src, (D *)dst, n);
    return URHGPU_OK;
}
// one side is float32 (the filters' working type), the other one of the four integer sample types
int launch_astype(const void *src, int src_dtype, void *dst, int dst_dtype, int64_t n, hipStream_t s) {
    if (n <= 0) return URHGPU_OK;
    if (dst_dtype == URHGPU_DT_F32) {
        switch (src_dtype) {
            case URHGPU_DT_I8: return astype_launch<int8_t, float>(src, dst, n, s);
            case URHGPU_DT_U8: return astype_launch<uint8_t, float>(src, dst, n, s);
            case URHGPU_DT_I16: return astype_launch<int16_t, float>(src, dst, n, s);
            case URHGPU_DT_U16: return astype_launch<uint16_t, float>(src, dst, n, s);
            default: return URHGPU_ERR_DTYPE;
        }
    }
    if (src_dtype == URHGPU_DT_F32) {
        switch (dst_dtype) {
            case URHGPU_DT_I8: return astype_launch<float, int8_t>(src, dst, n, s);
            case URHGPU_DT_U8: return astype_launch<float, uint8_t>(src, dst, n, s);
            case URHGPU_DT_I16: return astype_launch<float, int16_t>(src, dst, n, s);
            case URHGPU_DT_U16: return astype_launch<float, uint16_t>(src, dst, n, s);
            default: return URHGPU_ERR_DTYPE;
        }
    }
    return URHGPU_ERR_DTYPE;
}

// ---- PCM frames of a WAV file -> float32 IQ (Signal.__load_wav_file, Signal.py:114-173) ----------------------------------
// One thread per frame.  A sample of `width` bytes (1: unsigned, 2 / 3 / 4: signed little endian, three bytes sign-extended as the
// reference does with its fourth byte, :137-146) becomes np.multiply(1 / max, np.subtract(sample, center)) -- float64 arithmetic, rounded
// once into the float32 IQArray (:151-163) -- with max = 255 / 32767 / 8388607 / 2147483647 and center = (min + max) / 2 = 127.5 / -0.5.
// Mono: the imaginary part is 0 (the capture counts as already demodulated, :155-156); stereo: left -> real, right -> imag.
// 1-8 B read, 8 B written per frame: HBM-bound, byte loads (a 3-byte sample has no alignment).
__global__ __launch_bounds__(256) void k_pcm_to_iq(const uint8_t *raw, int64_t n_frames, int channels, int width, float *out) {
    const double mx = (width == 1) ? 255.0 : (width == 2) ? 32767.0 : (width == 3) ? 8388607.0 : 2147483647.0;
    const double center = (width == 1) ? 127.5 : -0.5, scale = 1.0 / mx;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_frames; i += (int64_t)gridDim.x * blockDim.x) {
        float v[2] = {0.f, 0.f};
        for (int c = 0; c < channels; ++c) {
            const uint8_t *q = raw + (i * channels + c) * (int64_t)width;
            int32_t x;
            if (width == 1) x = (int32_t)q[0];
            else if (width == 2) x = (int32_t)(int16_t)((uint16_t)q[0] | ((uint16_t)q[1] << 8));
            else if (width == 3) x = (int32_t)(((uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16)) << 8) >> 8;
            else x = (int32_t)((uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24));
            v[c] = (float)(scale * ((double)x - center));
        }
        *(float2 *)(out + 2 * i) = float2{v[0], v[1]};
    }
}
int launch_pcm_to_iq(const void *raw, int64_t n_frames, int channels, int width, float *out, hipStream_t s) {
    if (n_frames <= 0) return URHGPU_OK;
    int64_t g = (n_frames + 255) / 256; if (g > 65536) g = 65536;
    hipLaunchKernelGGL(k_pcm_to_iq, dim3((unsigned)g), dim3(256), 0, s, (const uint8_t *)raw, n_frames, channels, width, out);
    return URHGPU_OK;
}

// n = number of VALUES (2 per IQ sample); src_dtype != dst_dtype
int launch_convert(const void *src, int src_dtype, void *dst, int dst_dtype, int64_t n, hipStream_t s) {
    if (n <= 0) return URHGPU_OK;
    switch (src_dtype) {
        case URHGPU_DT_U8: return conv_from<uint8_t>(src, dst_dtype, dst, n, s);
        case URHGPU_DT_I8: return conv_from<int8_t>(src, dst_dtype, dst, n, s);
        case URHGPU_DT_U16: return conv_from<uint16_t>(src, dst_dtype, dst, n, s);
        case URHGPU_DT_I16: return conv_from<int16_t>(src, dst_dtype, dst, n, s);
        case URHGPU_DT_F32: return conv_from<float>(src, dst_dtype, dst, n, s);
        default: return URHGPU_ERR_DTYPE;
    }
}

}  // namespace urh
