// copyshape.hip -- which access SHAPE bounds the hot kernel?  Pure copies with the hot kernel's structure (one workgroup of
// W wavefronts per chunk, every wavefront streams its own contiguous share, loads software-pipelined 2 rows ahead,
// non-temporal), for 8 B in / 4 B out per sample (complex64 -> qad) and 4 B in / 4 B out (complex int16 -> qad), with a
// lane owning 2 consecutive samples per row (the kernel's layout: 16 B / 8 B loads, 8 B stores) or 4 (32 B / 16 B loads,
// 16 B stores).  Developer tool: hipcc --offload-arch=gfx950 -O3 copyshape.hip -o copyshape; ./copyshape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float v4 __attribute__((ext_vector_type(4)));
typedef float v2 __attribute__((ext_vector_type(2)));

// IN: dwords read per lane and row-step, OUT: dwords written per lane and row-step; rows_per_wave row-steps per wavefront
template <int IN, int OUT>
__global__ void k_copy(const float *in, float *out, int rows_per_wave) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const float *src = in + wave * rows_per_wave * 64 * IN + lane * IN;
    float *dst = out + wave * rows_per_wave * 64 * OUT + lane * OUT;
    constexpr int NB = 2;
    float cur[NB][IN], nxt[NB][IN];
    auto ld = [&](float (&r)[IN], const float *p) {
        if (IN == 1) { r[0] = __builtin_nontemporal_load(p); }
        else if (IN == 2) { const v2 t = __builtin_nontemporal_load((const v2 *)p); r[0] = t.x; r[1] = t.y; }
        else {
#pragma unroll
            for (int q = 0; q < IN / 4; ++q) { const v4 t = __builtin_nontemporal_load((const v4 *)p + q); r[4 * q] = t.x; r[4 * q + 1] = t.y; r[4 * q + 2] = t.z; r[4 * q + 3] = t.w; }
        }
    };
#pragma unroll
    for (int j = 0; j < NB; ++j) ld(cur[j], src + (int64_t)j * 64 * IN);
#pragma unroll 1
    for (int r = 0; r < rows_per_wave; r += NB) {
        if (r + NB < rows_per_wave) {
#pragma unroll
            for (int j = 0; j < NB; ++j) ld(nxt[j], src + (int64_t)(r + NB + j) * 64 * IN);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            float o[OUT];
#pragma unroll
            for (int q = 0; q < OUT; ++q) o[q] = (IN >= OUT) ? cur[j][q * (IN / OUT)] + cur[j][q * (IN / OUT) + (IN / OUT) - 1] : cur[j][0] + (float)q;
            float *p = dst + (int64_t)(r + j) * 64 * OUT;
            if (OUT == 2) { const v2 t = {o[0], o[1]}; __builtin_nontemporal_store(t, (v2 *)p); }
            else { const v4 t = {o[0], o[1], o[2], o[3]}; __builtin_nontemporal_store(t, (v4 *)p); }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int q = 0; q < IN; ++q) cur[j][q] = nxt[j][q];
    }
}

template <int IN, int OUT>
static int run(const char *what, const float *in, float *out, int64_t n_samples, int in_dw_per_sample, int W, hipStream_t s) {
    // samples per lane-step = OUT (one qad dword per sample); a wavefront streams 16 KiB of qad-equivalent rows: 2048 samples
    const int64_t samples_per_step = 64 * OUT;
    const int rows_per_wave = (int)(2048 / samples_per_step);
    const int64_t waves = n_samples / 2048;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((k_copy<IN, OUT>), dim3((unsigned)(waves / W)), dim3(64 * W), 0, s, in, out, rows_per_wave);
    CK(hipEventRecord(e0, s));
    const int iters = 20;
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL((k_copy<IN, OUT>), dim3((unsigned)(waves / W)), dim3(64 * W), 0, s, in, out, rows_per_wave);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    const double bytes = (double)n_samples * 4 * ((double)IN / OUT + 1);
    printf("%-62s W=%d  %7.4f ms  %7.1f GB/s\n", what, W, ms, bytes / ms / 1e6);
    return 0;
}

int main() {
    const int64_t n = (int64_t)1 << 27;
    float *in, *out; CK(hipMalloc(&in, n * 8)); CK(hipMalloc(&out, n * 4));
    CK(hipMemset(in, 0, n * 8));
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int W : {4, 8}) {
        if (W == 4) {
            run<4, 2>("complex64, lane = 2 samples (16 B load, 8 B store)", in, out, n, 2, 4, s);
            run<8, 4>("complex64, lane = 4 samples (2 x 16 B load, 16 B store)", in, out, n, 2, 4, s);
            run<2, 2>("complex int16, lane = 2 samples (8 B load, 8 B store)", in, out, n, 1, 4, s);
            run<4, 4>("complex int16, lane = 4 samples (16 B load, 16 B store)", in, out, n, 1, 4, s);
            run<1, 2>("complex int8, lane = 2 samples (4 B load, 8 B store)", in, out, n, 1, 4, s);
            run<2, 4>("complex int8, lane = 4 samples (8 B load, 16 B store)", in, out, n, 1, 4, s);
        } else {
            run<4, 2>("complex64, lane = 2 samples (16 B load, 8 B store)", in, out, n, 2, 8, s);
            run<8, 4>("complex64, lane = 4 samples (2 x 16 B load, 16 B store)", in, out, n, 2, 8, s);
            run<2, 2>("complex int16, lane = 2 samples (8 B load, 8 B store)", in, out, n, 1, 8, s);
            run<4, 4>("complex int16, lane = 4 samples (16 B load, 16 B store)", in, out, n, 1, 8, s);
            run<1, 2>("complex int8, lane = 2 samples (4 B load, 8 B store)", in, out, n, 1, 8, s);
        }
    }
    return 0;
}
