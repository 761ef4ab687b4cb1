// Does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on this runtime?  A one-workgroup kernel that spins for ~200 us,
// launched twice back to back: 200 us when they overlap, 400 when the second waits for the first.  (tools/kbench, round 4)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void k_spin(long long ticks, int *out) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (out && threadIdx.x == 0) atomicAdd(out, 1);
}
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int *d; CK(hipMalloc(&d, 4)); CK(hipMemset(d, 0, 4));
    hipEvent_t e_dt[2], e_t[2], e_t2[2];
    for (auto &e : e_dt) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : e_t) CK(hipEventCreate(&e));
    for (auto &e : e_t2) CK(hipEventCreate(&e));
    const long long ticks = 20000;      // 100 MHz wall clock: 200 us
    for (int mode = 0; mode < 5; ++mode) {
        double best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipStreamSynchronize(s));
            auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < 2; ++k) {
                if (mode == 0) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, ticks, d);
                else if (mode == 1) hipExtLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, d);
                else if (mode == 2) hipExtLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, nullptr, e_dt[k], hipExtAnyOrderLaunch, ticks, d);
                else if (mode == 3) hipExtLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, e_t[k], e_t2[k], hipExtAnyOrderLaunch, ticks, d);
                else hipExtLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, nullptr, e_dt[k], 0, ticks, d);
            }
            CK(hipStreamSynchronize(s));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (us < best) best = us;
        }
        const char *names[5] = {"hipLaunchKernelGGL x2", "ext any-order, no events x2", "ext any-order + stop event (no timing) x2", "ext any-order + start/stop timing events x2",
                                "ext in-order + stop event x2"};
        printf("%-50s %8.1f us\n", names[mode], best);
    }
    return 0;
}
