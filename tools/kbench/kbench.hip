// kbench.hip -- stand-alone A/B timing harness for the hot kernel (developer tool, not part of the product).
//   build: tools/kbench/build.sh     run (GPU box): tools/kbench/kbench_<variant> [log2_samples] [tiles_per_chunk]
// Times k_demod_runs (current source, compiled with -DURH_KBATCH / -DURH_MINWAVES variants) against
// a copy-shaped ceiling kernel on a device-generated 2-FSK capture, and checks that
// both kernels produce identical qad / chunk records (full-size parity between kernel generations).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../../urh_amd/csrc/demod_runs.hip"
#define LAUNCH(a, mod, wq, s) urh::launch_demod_runs_iq(a, URHGPU_DT_F32, mod, wq, s)

namespace urh { thread_local char g_hip_err[256] = ""; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ inline uint32_t hash32(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return (uint32_t)x;
}

// continuous-phase 2-FSK, +-0.02 cycles/sample, 100 samples/symbol, AWGN-ish noise sigma 0.05
__global__ void k_synth(float2 *iq, int64_t n, const int *sym_prefix, int sps, float sigma) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t sym = i / sps;
        const int bit = (hash32(sym * 2 + 1) >> 7) & 1;
        const int64_t steps = (int64_t)sym_prefix[sym] + (bit ? 1 : -1) * (i - sym * sps);   // signed step count
        const double ph = 2.0 * M_PI * 0.02 * (double)(steps % 50);
        // Box-Muller on two hashes
        const uint32_t h1 = hash32(i * 2 + 0x1234567), h2 = hash32(i * 2 + 0x89abcdef);
        const float u1 = ((h1 >> 8) + 1) * (1.0f / 16777217.0f), u2 = (h2 >> 8) * (1.0f / 16777216.0f);
        const float rr = sigma * sqrtf(-2.0f * logf(u1));
        iq[i] = make_float2((float)cos(ph) + rr * cosf(6.2831853f * u2), (float)sin(ph) + rr * sinf(6.2831853f * u2));
    }
}

// ceiling: same traffic shape as the hot kernel (16 B read, 8 B write per thread-row), trivial math
__global__ __launch_bounds__(256) void k_ceiling(const float4 *in, float2 *out, int64_t n_pairs) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_pairs; i += stride) {
        const float4 v = in[i];
        out[i] = make_float2(v.x + v.y, v.z + v.w);
    }
}
// same, chunked like the hot kernel (each workgroup streams its own contiguous region)
__global__ __launch_bounds__(256) void k_ceiling_chunked(const float4 *in, float2 *out, int64_t pairs_per_block) {
    const int64_t b0 = blockIdx.x * pairs_per_block;
    for (int64_t i = threadIdx.x; i < pairs_per_block; i += 256) {
        const float4 v = in[b0 + i];
        out[b0 + i] = make_float2(v.x + v.y, v.z + v.w);
    }
}
typedef float kv4 __attribute__((ext_vector_type(4)));
typedef float kv2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_ceiling_chunked_nt(const kv4 *in, kv2 *out, int64_t pairs_per_block) {
    const int64_t b0 = blockIdx.x * pairs_per_block;
    for (int64_t i = threadIdx.x; i < pairs_per_block; i += 256) {
        const kv4 v = __builtin_nontemporal_load(in + b0 + i);
        const kv2 o = {v.x + v.y, v.z + v.w};
        __builtin_nontemporal_store(o, out + b0 + i);
    }
}

// one wavefront per chunk, rows of 1 KiB, software-pipelined NB rows ahead: the hot kernel's memory structure without its math
template <int NB>
__global__ __launch_bounds__(64) void k_ceiling_wave_nt(const kv4 *in, kv2 *out, int64_t pairs_per_block) {
    const int64_t b0 = blockIdx.x * pairs_per_block;
    kv4 cur[NB], nxt[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) cur[j] = __builtin_nontemporal_load(in + b0 + j * 64 + threadIdx.x);
#pragma unroll 1
    for (int64_t i = 0; i < pairs_per_block; i += 64 * NB) {
        if (i + 64 * NB < pairs_per_block) {
#pragma unroll
            for (int j = 0; j < NB; ++j) nxt[j] = __builtin_nontemporal_load(in + b0 + i + 64 * NB + j * 64 + threadIdx.x);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const kv2 o = {cur[j].x + cur[j].y, cur[j].z + cur[j].w};
            __builtin_nontemporal_store(o, out + b0 + i + j * 64 + threadIdx.x);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) cur[j] = nxt[j];
    }
}
// same with a per-chunk rotated start row (do concurrently running wavefronts collide on memory channels when they
// all sit at the same offset of their 64 KiB chunk?)
template <int NB>
__global__ __launch_bounds__(64) void k_ceiling_wave_rot(const kv4 *in, kv2 *out, int64_t pairs_per_block, int mult) {
    const int64_t b0 = blockIdx.x * pairs_per_block;
    const int64_t steps = pairs_per_block / (64 * NB);
    int64_t st = ((int64_t)blockIdx.x * mult) % steps;
    kv4 cur[NB], nxt[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) cur[j] = __builtin_nontemporal_load(in + b0 + st * 64 * NB + j * 64 + threadIdx.x);
#pragma unroll 1
    for (int64_t k = 0; k < steps; ++k) {
        int64_t sn = st + 1; if (sn == steps) sn = 0;
        if (k + 1 < steps) {
#pragma unroll
            for (int j = 0; j < NB; ++j) nxt[j] = __builtin_nontemporal_load(in + b0 + sn * 64 * NB + j * 64 + threadIdx.x);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const kv2 o = {cur[j].x + cur[j].y, cur[j].z + cur[j].w};
            __builtin_nontemporal_store(o, out + b0 + st * 64 * NB + j * 64 + threadIdx.x);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) cur[j] = nxt[j];
        st = sn;
    }
}
static float time_ms(hipStream_t s, int iters, void (*fn)(void *), void *arg) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    fn(arg); fn(arg);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) fn(arg);
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

struct Ctx {
    urh::RunArgs a; hipStream_t s; int64_t n_chunks; bool write_qad;
    const float4 *in4; float2 *out2; int64_t n_pairs; int grid;
};

int main(int argc, char **argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 27;
    const int tpc = argc > 2 ? atoi(argv[2]) : 8;
    const int64_t n = (int64_t)1 << lg;
    const int sps = 100, tol = 5;
    hipStream_t s; CK(hipStreamCreate(&s));
    float2 *iq; float *qad;
    CK(hipMalloc(&iq, n * 8)); CK(hipMalloc(&qad, n * 4));
    // symbol prefix (signed step counts) on the host
    const int64_t nsym = n / sps + 2;
    std::vector<int> pre(nsym);
    {
        long acc = 0;
        for (int64_t k = 0; k < nsym; ++k) {
            pre[k] = (int)(acc % 50);   // phase only matters mod 50 steps (0.02 cycles each)
            uint64_t x = (uint64_t)k * 2 + 1; x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
            const int bit = ((uint32_t)x >> 7) & 1;
            acc += (bit ? 1 : -1) * sps;
            acc = ((acc % 50) + 50) % 50;
        }
    }
    int *d_pre; CK(hipMalloc(&d_pre, nsym * 4)); CK(hipMemcpy(d_pre, pre.data(), nsym * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_synth, dim3(4096), dim3(256), 0, s, iq, n, d_pre, sps, 0.05f);
    CK(hipStreamSynchronize(s));

    urh::RunArgs a; memset(&a, 0, sizeof(a));
    a.in = iq; a.qad = qad; a.left_halo = nullptr; a.n = n; a.pos_base = 0;
    a.chunk_len = (int64_t)tpc * urh::kTile;
    const int64_t n_chunks = (n + a.chunk_len - 1) / a.chunk_len;
    a.slab_stride = a.chunk_len / (tol + 1) + 2;
    a.noise_sqrd = 0.f; a.noise_val = -4.f; a.max_magnitude = sqrtf(2.f); a.order = 2; a.tol = tol; a.thr[0] = 0.f;
    CK(hipMalloc(&a.chunks, n_chunks * sizeof(urh::ChunkInfo)));
    CK(hipMalloc(&a.slab, n_chunks * a.slab_stride * 8));

    Ctx c_{a, s, n_chunks, true, (const float4 *)iq, (float2 *)qad, n / 2, 256 * 8};
    Ctx &c = c_;
    const int iters = 20;
    printf("n=2^%d samples, %lld chunks of %d tiles; variant KBATCH=%d MINWAVES=%d\n", lg, (long long)n_chunks, tpc, URH_KBATCH, URH_MINWAVES);
    float ms;
    ms = time_ms(s, iters, [](void *p) { Ctx *c = (Ctx *)p; hipLaunchKernelGGL(k_ceiling, dim3(c->grid), dim3(256), 0, c->s, c->in4, c->out2, c->n_pairs); }, &c);
    printf("ceiling grid-stride (2048 wg)   %8.4f ms  %7.1f GB/s\n", ms, n * 12.0 / ms / 1e6);
    c.grid = (int)n_chunks;
    ms = time_ms(s, iters, [](void *p) { Ctx *c = (Ctx *)p; hipLaunchKernelGGL(k_ceiling_chunked, dim3(c->grid), dim3(256), 0, c->s, c->in4, c->out2, c->n_pairs / c->grid); }, &c);
    printf("ceiling chunked (%lld wg)        %8.4f ms  %7.1f GB/s\n", (long long)n_chunks, ms, n * 12.0 / ms / 1e6);
    ms = time_ms(s, iters, [](void *p) { Ctx *c = (Ctx *)p; hipLaunchKernelGGL(k_ceiling_chunked_nt, dim3(c->grid), dim3(256), 0, c->s, (const kv4 *)c->in4, (kv2 *)c->out2, c->n_pairs / c->grid); }, &c);
    printf("ceiling chunked nt (%lld wg)     %8.4f ms  %7.1f GB/s\n", (long long)n_chunks, ms, n * 12.0 / ms / 1e6);
    ms = time_ms(s, iters, [](void *p) { Ctx *c = (Ctx *)p; hipLaunchKernelGGL(k_ceiling_wave_nt<1>, dim3(c->grid), dim3(64), 0, c->s, (const kv4 *)c->in4, (kv2 *)c->out2, c->n_pairs / c->grid); }, &c);
    printf("ceiling 1 wave/chunk nt, 1 row ahead    %8.4f ms  %7.1f GB/s\n", ms, n * 12.0 / ms / 1e6);
    ms = time_ms(s, iters, [](void *p) { Ctx *c = (Ctx *)p; hipLaunchKernelGGL(k_ceiling_wave_nt<2>, dim3(c->grid), dim3(64), 0, c->s, (const kv4 *)c->in4, (kv2 *)c->out2, c->n_pairs / c->grid); }, &c);
    printf("ceiling 1 wave/chunk nt, 2 rows ahead   %8.4f ms  %7.1f GB/s\n", ms, n * 12.0 / ms / 1e6);
    ms = time_ms(s, iters, [](void *p) { Ctx *c = (Ctx *)p; hipLaunchKernelGGL(k_ceiling_wave_nt<4>, dim3(c->grid), dim3(64), 0, c->s, (const kv4 *)c->in4, (kv2 *)c->out2, c->n_pairs / c->grid); }, &c);
    printf("ceiling 1 wave/chunk nt, 4 rows ahead   %8.4f ms  %7.1f GB/s\n", ms, n * 12.0 / ms / 1e6);
    for (int mult : {1, 3, 5, 7, 11}) {
        static int g_mult; g_mult = mult;
        ms = time_ms(s, iters, [](void *p) { Ctx *c = (Ctx *)p; hipLaunchKernelGGL(k_ceiling_wave_rot<2>, dim3(c->grid), dim3(64), 0, c->s, (const kv4 *)c->in4, (kv2 *)c->out2, c->n_pairs / c->grid, g_mult); }, &c);
        printf("ceiling 1 wave/chunk nt, 2 rows ahead, start rotated x%-2d  %8.4f ms  %7.1f GB/s\n", mult, ms, n * 12.0 / ms / 1e6);
    }
    ms = time_ms(s, iters, [](void *p) { Ctx *c = (Ctx *)p; LAUNCH(c->a, URHGPU_MOD_FSK, true, c->s); }, &c);
    printf("k_demod_runs FSK (qad written)  %8.4f ms  %7.1f GB/s (12 B/sample)\n", ms, n * 12.0 / ms / 1e6);
    ms = time_ms(s, iters, [](void *p) { Ctx *c = (Ctx *)p; LAUNCH(c->a, URHGPU_MOD_FSK, false, c->s); }, &c);
    printf("k_demod_runs FSK (bits only)    %8.4f ms  %7.1f GB/s (8 B/sample)\n", ms, n * 8.0 / ms / 1e6);
    ms = time_ms(s, iters, [](void *p) { Ctx *c = (Ctx *)p; LAUNCH(c->a, URHGPU_MOD_ASK, true, c->s); }, &c);
    printf("k_demod_runs ASK (qad written)  %8.4f ms  %7.1f GB/s (12 B/sample)\n", ms, n * 12.0 / ms / 1e6);
    // fingerprints of the FSK result (compare between kbench variants: full-size parity between kernel generations)
    CK(hipMemsetAsync(qad, 0, n * 4, s));
    { Ctx *c = &c_; LAUNCH(c->a, URHGPU_MOD_FSK, true, s); }
    CK(hipStreamSynchronize(s));
    std::vector<float> h1(n);
    CK(hipMemcpy(h1.data(), qad, n * 4, hipMemcpyDeviceToHost));
    uint64_t hq = 1469598103934665603ull;
    for (int64_t i = 0; i < n; ++i) { uint32_t u; memcpy(&u, &h1[i], 4); hq = (hq ^ u) * 1099511628211ull; }
    std::vector<urh::ChunkInfo> ci1(n_chunks);
    CK(hipMemcpy(ci1.data(), a.chunks, n_chunks * sizeof(urh::ChunkInfo), hipMemcpyDeviceToHost));
    std::vector<uint64_t> s1(n_chunks * a.slab_stride);
    CK(hipMemcpy(s1.data(), a.slab, s1.size() * 8, hipMemcpyDeviceToHost));
    uint64_t hc = 1469598103934665603ull, hs = hc; int64_t recs = 0;
    for (int64_t k = 0; k < n_chunks; ++k) {
        const urh::ChunkInfo &ci = ci1[k];
        const int64_t f[] = {ci.pend_pos, ci.lead, ci.start, ci.len, ci.cnt > 0 ? ci.last_pos : 0, ci.cnt, ci.first_state, ci.last_state,
                             ci.pend_pos >= 0 ? ci.pend_state : 0, ci.init_state};
        for (int64_t v : f) hc = (hc ^ (uint64_t)v) * 1099511628211ull;
        for (int j = 0; j < ci.cnt; ++j) { hs = (hs ^ s1[k * a.slab_stride + j]) * 1099511628211ull; ++recs; }
    }
    double dsum = 0; int64_t nn = 0; for (int64_t i = 0; i < n; i += 997) { dsum += fabs(h1[i]); ++nn; }
    printf("fingerprints: qad %016llx chunks %016llx runs %016llx (%lld run records), mean |qad| = %.5f (expect ~0.1257)\n",
           (unsigned long long)hq, (unsigned long long)hc, (unsigned long long)hs, (long long)recs, dsum / nn);
    return 0;
}
