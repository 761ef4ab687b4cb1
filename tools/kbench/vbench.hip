// vbench: issue cost of packed / plain fp32 VALU streams on gfx950 (which instruction mix can a non-fused complex MAC reach?)
//   hipcc --offload-arch=gfx950 -O3 vbench.hip -o vbench && ./vbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v2f __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kIters = 2048;

// 8 independent accumulator chains, one instruction each per round: MODE selects the instruction
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, long long *cyc, float seed) {
    v2f a[8], x = {seed, seed + 1.f}, h = {seed + 2.f, seed + 3.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = v2f{seed + i, seed - i};
    long long t0 = clock64();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(h));
            if (MODE == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(h));
            if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(h), "v"(x));
            if (MODE == 3) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(h.x));
            if (MODE == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(h.x), "v"(x.x));
            if (MODE == 5) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[1,1] op_sel_hi:[1,0]" : "+v"(a[i]) : "v"(h));
            if (MODE == 6) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,0]" : "+v"(a[i]) : "v"(h));
            if (MODE == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(h.x));
        }
    }
    long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// the FIR inner body: 4 complex MACs (two cmac2 blocks) per round on 4 accumulators, exactly as filters.hip issues them
template <int MODE>
__global__ __launch_bounds__(256) void kmac(float *out, long long *cyc, float seed) {
    v2f acc[8], x[8], h = {seed + 2.f, seed + 3.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc[i] = v2f{seed + i, seed - i}; x[i] = v2f{seed * i, seed + 2 * i}; }
    long long t0 = clock64();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            v2f t1a, t2a, t1b, t2b;
            if (MODE == 0) {
                asm volatile(
                    "v_pk_mul_f32 %2, %6, %8 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
                    "v_pk_mul_f32 %3, %6, %8 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
                    "v_pk_mul_f32 %4, %7, %8 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
                    "v_pk_mul_f32 %5, %7, %8 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
                    "v_pk_add_f32 %2, %2, %3 neg_lo:[0,1] neg_hi:[0,0]\n\t"
                    "v_pk_add_f32 %4, %4, %5 neg_lo:[0,1] neg_hi:[0,0]\n\t"
                    "v_pk_add_f32 %0, %0, %2\n\t"
                    "v_pk_add_f32 %1, %1, %4"
                    : "+v"(acc[i]), "+v"(acc[i + 1]), "=&v"(t1a), "=&v"(t2a), "=&v"(t1b), "=&v"(t2b)
                    : "v"(x[i]), "v"(x[i + 1]), "v"(h));
            } else {
                // plain fp32: 4 mul + 2 add/sub + 2 add per MAC, two MACs interleaved
                float p0, p1, p2, p3, q0, q1, q2, q3;
                asm volatile(
                    "v_mul_f32 %4, %12, %16\n\t"
                    "v_mul_f32 %5, %13, %17\n\t"
                    "v_mul_f32 %6, %12, %17\n\t"
                    "v_mul_f32 %7, %13, %16\n\t"
                    "v_mul_f32 %8, %14, %16\n\t"
                    "v_mul_f32 %9, %15, %17\n\t"
                    "v_mul_f32 %10, %14, %17\n\t"
                    "v_mul_f32 %11, %15, %16\n\t"
                    "v_sub_f32 %4, %4, %5\n\t"
                    "v_add_f32 %6, %6, %7\n\t"
                    "v_sub_f32 %8, %8, %9\n\t"
                    "v_add_f32 %10, %10, %11\n\t"
                    "v_add_f32 %0, %0, %4\n\t"
                    "v_add_f32 %1, %1, %6\n\t"
                    "v_add_f32 %2, %2, %8\n\t"
                    "v_add_f32 %3, %3, %10"
                    : "+v"(acc[i].x), "+v"(acc[i].y), "+v"(acc[i + 1].x), "+v"(acc[i + 1].y), "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(q0),
                      "=&v"(q1), "=&v"(q2), "=&v"(q3)
                    : "v"(x[i].x), "v"(x[i].y), "v"(x[i + 1].x), "v"(x[i + 1].y), "v"(h.x), "v"(h.y));
            }
        }
    }
    long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <class K>
static void run(const char *name, K kern, int waves_per_simd, double instr_per_iter, double flop_per_lane_iter) {
    float *out; long long *cyc;
    const int blocks = 256 * waves_per_simd;               // 256 threads = 4 waves = one per SIMD; `waves_per_simd` blocks per CU
    CK(hipMalloc(&out, (size_t)blocks * 256 * 4)); CK(hipMalloc(&cyc, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double tflops = flop_per_lane_iter * kIters * 64.0 * 4 * blocks / (ms * 1e-3) / 1e12;
    printf("%-44s waves/SIMD %d  %7.3f ms  clock64/instr %6.2f  %7.1f TFLOP/s\n", name, waves_per_simd, ms, (double)c / (kIters * instr_per_iter), tflops);
    CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
    for (int w : {1, 2, 4, 8}) {
        run("v_pk_mul_f32", k<0>, w, 8, 16);
        run("v_pk_add_f32", k<1>, w, 8, 16);
        run("v_pk_fma_f32", k<2>, w, 8, 32);
        run("v_mul_f32", k<3>, w, 8, 8);
        run("v_fma_f32", k<4>, w, 8, 16);
        run("v_pk_mul_f32 op_sel swap", k<5>, w, 8, 16);
        run("v_pk_add_f32 neg_lo", k<6>, w, 8, 16);
        run("v_add_f32", k<7>, w, 8, 8);
        run("complex MAC x4, packed (FIR body)", kmac<0>, w, 32, 32);
        run("complex MAC x4, plain fp32", kmac<1>, w, 64, 32);
    }
    return 0;
}
