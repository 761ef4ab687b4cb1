// lbench: latency of DEPENDENT VALU operations on gfx950 (one wavefront per SIMD, one chain): what bounds the Costas step
//   hipcc --offload-arch=gfx950 -O3 lbench.hip -o lbench && ./lbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int kIters = 4096, kUnroll = 16;

template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, long long *cyc, double seed) {
    double d = seed + threadIdx.x * 1e-9, c = 1.0000001, e = 1e-9;
    float f = (float)seed, g = 1.0000001f, h = 1e-9f;
    int n = 3;
    const long long t0 = clock64();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f) : "v"(g), "v"(h));
            if (MODE == 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d) : "v"(c), "v"(e));
            if (MODE == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d) : "v"(c));
            if (MODE == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d) : "v"(e));
            if (MODE == 4) { asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f) : "v"(d)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(f)); }
            if (MODE == 5) { asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(n) : "v"(d)); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d) : "v"(n)); }
            if (MODE == 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f) : "v"(g));
            if (MODE == 7) { asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %2, vcc" : "=v"(f) : "v"(f), "v"(g) : "vcc"); }
            if (MODE == 8) { asm volatile("v_cmp_gt_f64 vcc, %1, %2\n\tv_cndmask_b32 %0, %3, %4, vcc" : "=v"(f) : "v"(d), "v"(c), "v"(f), "v"(g) : "vcc"); d += (double)f * 0; }
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = d + f + n;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <class K> static void run(const char *name, K kern, double ops) {
    double *out; long long *cyc;
    CK(hipMalloc(&out, 1024 * 64 * 8)); CK(hipMalloc(&cyc, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(1024), dim3(64), 0, 0, out, cyc, 1.0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(1024), dim3(64), 0, 0, out, cyc, 1.0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-36s %7.3f ms   %6.1f ns per dependent op (%5.1f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / (kIters * kUnroll * ops),
           ms * 1e6 / (kIters * kUnroll * ops) * 2.4);
}
int main() {
    run("v_fma_f32 chain", k<0>, 1); run("v_fma_f64 chain", k<1>, 1); run("v_mul_f64 chain", k<2>, 1); run("v_add_f64 chain", k<3>, 1);
    run("cvt f64->f32->f64 (2 ops)", k<4>, 2); run("cvt f64->i32->f64 (2 ops)", k<5>, 2); run("v_mul_f32 chain", k<6>, 1);
    run("v_cmp_f32 + v_cndmask (2 ops)", k<7>, 2); run("v_cmp_f64 + cndmask + cvt,mul,add", k<8>, 5);
    return 0;
}
