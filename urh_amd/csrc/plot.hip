// plot.hip -- the O(N) pass behind the Interpretation tab's signal plot: per-pixel minimum / maximum
//   /root/reference/src/urh/cythonext/path_creator.pyx:46-66 (create_path)
// One workgroup per pixel (a stretch of samples_per_pixel samples), coalesced strided reads, HBM bound (every sample is
// read once, 2 values per pixel are written).  Results equal the reference's sequential scan bit for bit:
//   * a NaN never wins a comparison, so it is skipped -- unless it is the stretch's FIRST sample, which then stays
//     both minimum and maximum (:52-54);
//   * equal values (+0.0 / -0.0) keep the earliest one: the reduction carries the sample index as tie-break.
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "launchers.hpp"

namespace urh {

template <typename T> struct Ext { T v; int64_t i; };      // i < 0: none yet

template <typename T> __device__ __forceinline__ void ext_min(Ext<T> &a, const Ext<T> &b) {
    if (b.i < 0) return;
    if (a.i < 0 || b.v < a.v || (!(a.v < b.v) && b.i < a.i)) a = b;
}
template <typename T> __device__ __forceinline__ void ext_max(Ext<T> &a, const Ext<T> &b) {
    if (b.i < 0) return;
    if (a.i < 0 || b.v > a.v || (!(a.v > b.v) && b.i < a.i)) a = b;
}

template <typename T>
__global__ __launch_bounds__(256) void k_path_minmax(const T *samples, int64_t start, int64_t end, int64_t spp, T *values) {
    __shared__ Ext<T> s_mn[4], s_mx[4];
    const int64_t i0 = start + (int64_t)blockIdx.x * spp;
    const int64_t i1 = (i0 + spp < end) ? i0 + spp : end;
    Ext<T> mn{T(0), -1}, mx{T(0), -1};
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
        const T v = samples[i];
        if (!(v == v)) continue;                           // NaN
        if (mn.i < 0) { mn = Ext<T>{v, i}; mx = mn; }
        else if (v < mn.v) mn = Ext<T>{v, i};
        else if (v > mx.v) mx = Ext<T>{v, i};
    }
    for (int o = 32; o > 0; o >>= 1) {
        Ext<T> a{(T)__shfl_down(mn.v, o), __shfl_down(mn.i, o)}, b{(T)__shfl_down(mx.v, o), __shfl_down(mx.i, o)};
        ext_min(mn, a); ext_max(mx, b);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_mn[wave] = mn; s_mx[wave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { ext_min(mn, s_mn[w]); ext_max(mx, s_mx[w]); }
        const T first = samples[i0];
        if (!(first == first) || mn.i < 0) { values[2 * (int64_t)blockIdx.x] = first; values[2 * (int64_t)blockIdx.x + 1] = first; }
        else { values[2 * (int64_t)blockIdx.x] = mn.v; values[2 * (int64_t)blockIdx.x + 1] = mx.v; }
    }
}

template <typename T>
static void launch_path_t(const void *samples, int64_t start, int64_t end, int64_t spp, void *values, int64_t pixels, hipStream_t s) {
    hipLaunchKernelGGL(k_path_minmax<T>, dim3((unsigned)pixels), dim3(256), 0, s, (const T *)samples, start, end, spp, (T *)values);
}

// pixels = ceil((end - start) / spp) stretches; values[2 * pixels]
int launch_path_minmax(const void *samples, int dtype, int64_t start, int64_t end, int64_t spp, void *values, hipStream_t s) {
    if (spp < 1 || end <= start) return URHGPU_ERR_ARG;
    const int64_t pixels = (end - start + spp - 1) / spp;
    if (pixels > 0x7fffffff) return URHGPU_ERR_ARG;
    switch (dtype) {
        case URHGPU_DT_I8: launch_path_t<int8_t>(samples, start, end, spp, values, pixels, s); return URHGPU_OK;
        case URHGPU_DT_U8: launch_path_t<uint8_t>(samples, start, end, spp, values, pixels, s); return URHGPU_OK;
        case URHGPU_DT_I16: launch_path_t<int16_t>(samples, start, end, spp, values, pixels, s); return URHGPU_OK;
        case URHGPU_DT_U16: launch_path_t<uint16_t>(samples, start, end, spp, values, pixels, s); return URHGPU_OK;
        case URHGPU_DT_F32: launch_path_t<float>(samples, start, end, spp, values, pixels, s); return URHGPU_OK;
        default: return URHGPU_ERR_DTYPE;
    }
}

}  // namespace urh
