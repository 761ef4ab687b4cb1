// ubench.hip -- VALU instruction issue-rate microbenchmark for gfx950 (developer tool).
// Each kernel runs ITER x 16 independent instances of one instruction per wave, 8 waves per SIMD on
// every CU; reports cycles per wave64 instruction per SIMD (wall time x clock / instructions per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int ITER = 4096;

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

#define KERNEL(name, body)                                                              \
    __global__ __launch_bounds__(256) void name(float *out, float seed) {               \
        float r[16]; float2 p[16];                                                       \
        for (int i = 0; i < 16; ++i) { r[i] = seed + i + threadIdx.x; p[i] = make_float2(r[i], r[i] * 0.5f); } \
        float a = seed * 1.0001f, b = seed * 0.9999f; float2 a2 = make_float2(a, b);    \
        (void)a2; (void)b;                                                               \
        for (int it = 0; it < ITER; ++it) { REP16(body) }                                \
        float s = 0; for (int i = 0; i < 16; ++i) s += r[i] + p[i].x + p[i].y;           \
        if (s == 12345.678f) out[0] = s;                                                 \
    }

#define B_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define B_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define B_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define B_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(a2));
#define B_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(a2));
#define B_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(a2));
#define B_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
#define B_SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(r[i]));
#define B_DIVSCALE(i) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(r[i]) : "v"(a) : "vcc");
#define B_DIVFMAS(i) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b) : "vcc");
#define B_DIVFIXUP(i) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define B_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(a) : "vcc");
#define B_CMP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(r[i]), "v"(a) : "vcc");
#define B_MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"(a));
#define B_AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define B_BFI(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define B_DPP(i) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[i]) : "v"(a));
#define B_ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define B_LSHLADD64(i) asm volatile("v_lshl_add_u64 %0, %0, 3, %0" : "+v"(p[i]));
#define B_CVT(i) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(r[i]));
#define B_CLASS(i) asm volatile("v_cmp_class_f32 vcc, %0, %1" : : "v"(r[i]), "v"(a) : "vcc");
#define B_SMOV(i) asm volatile("s_mov_b32 s20, s21" : : : "s20");

KERNEL(k_mul, B_MUL) KERNEL(k_add, B_ADD) KERNEL(k_fma, B_FMA) KERNEL(k_pkmul, B_PKMUL) KERNEL(k_pkadd, B_PKADD)
KERNEL(k_pkfma, B_PKFMA) KERNEL(k_rcp, B_RCP) KERNEL(k_sqrt, B_SQRT) KERNEL(k_divscale, B_DIVSCALE)
KERNEL(k_divfmas, B_DIVFMAS) KERNEL(k_divfixup, B_DIVFIXUP) KERNEL(k_cndmask, B_CNDMASK) KERNEL(k_cmp, B_CMP)
KERNEL(k_mov, B_MOV) KERNEL(k_and, B_AND) KERNEL(k_bfi, B_BFI) KERNEL(k_dpp, B_DPP) KERNEL(k_addu, B_ADDU)
KERNEL(k_lshladd64, B_LSHLADD64) KERNEL(k_cvt, B_CVT) KERNEL(k_class, B_CLASS) KERNEL(k_smov, B_SMOV)

int main() {
    float *out; CK(hipMalloc(&out, 64));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    printf("%s: %d CUs, clock %.3f GHz (nominal)\n", prop.name, cus, ghz);
    struct { const char *name; void (*k)(float *, float); } tests[] = {
        {"v_mul_f32", k_mul}, {"v_add_f32", k_add}, {"v_fma_f32", k_fma}, {"v_pk_mul_f32", k_pkmul}, {"v_pk_add_f32", k_pkadd},
        {"v_pk_fma_f32", k_pkfma}, {"v_rcp_f32", k_rcp}, {"v_sqrt_f32", k_sqrt}, {"v_div_scale_f32", k_divscale},
        {"v_div_fmas_f32", k_divfmas}, {"v_div_fixup_f32", k_divfixup}, {"v_cndmask_b32", k_cndmask}, {"v_cmp_lt_f32", k_cmp},
        {"v_mov_b32", k_mov}, {"v_and_b32", k_and}, {"v_bfi_b32", k_bfi}, {"v_mov_b32_dpp", k_dpp}, {"v_add_u32", k_addu},
        {"v_lshl_add_u64", k_lshladd64}, {"v_cvt_f32_i32", k_cvt}, {"v_cmp_class_f32", k_class}, {"s_mov_b32", k_smov}};
    for (int wpe : {8, 4, 1}) {     // waves per SIMD
        printf("--- %d wave(s) per SIMD\n", wpe);
        const int blocks = cus * wpe;   // 256 threads = 4 waves = one per SIMD
        for (auto &t : tests) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, 1.5f);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, 1.5f);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double inst_per_simd = (double)ITER * 16 * wpe;
            printf("%-18s %8.3f ms  %6.2f cycles/wave-instr/SIMD @%.2f GHz\n", t.name, ms, ms * 1e-3 * ghz * 1e9 / inst_per_simd, ghz);
        }
    }
    return 0;
}
