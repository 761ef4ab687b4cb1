// filters.hip -- FIR / IIR / magnitude kernels (rows 2, 3, 6, 7 of SURVEY.md §8a) for gfx950.
//
//   k_fir              signal_functions.fir_filter   /root/reference/src/urh/cythonext/signal_functions.pyx:513-525
//   k_iir_*            signal_functions.iir_filter   /root/reference/src/urh/cythonext/signal_functions.pyx:527-542
//   k_magnitudes       util.get_magnitudes           /root/reference/src/urh/cythonext/util.pyx:128-136
//   k_mag_chunk_*      the O(N) part of AutoInterpretation.detect_noise_level
//                                                    /root/reference/src/urh/ainterpretation/AutoInterpretation.py:60-91
//
// Bit-exactness: -ffp-contract=off; every complex product is the 4-multiply / 2-add form the reference's
// generated C++ evaluates (std::complex<float> operator*), with the C99 Annex G recovery (__mulsc3) when both
// parts come out NaN; sums are accumulated in the reference's order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <math.h>
#include <utility>

#include "cmul.hpp"
#include "common.hpp"
#include "launchers.hpp"

namespace urh {

// ---------------------------------------------------------------------------------------------------------------
// FIR: out[k] = sum over i = max(0, k-M+1) .. k (ascending) of x[i] * h[k-i], each term a rounded complex64
// product, accumulated into a complex64 that starts at +0 (the reference's scatter loop on np.zeros).
// Roofline: fp32 VALU, not HBM -- 8 non-fused flops per tap per sample against 16 B of traffic per sample
// (M = 64: 512 flop / 16 B = 32 flop/B, above the fp32 ridge), so the layout serves the VALU: each lane
// owns R consecutive outputs and slides a register window over the LDS-staged input, taps broadcast from LDS;
// per tap and lane: one 8-byte LDS read of x, one broadcast read of h, R products.
// A workgroup of 256 lanes produces kFirTile = 256 * R outputs from kFirTile + H staged inputs.
// HEAD: this workgroup contains outputs k < M-1 of a capture without left halo (terms with i < 0 do not exist).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kFirBlock = 256;
constexpr int kFirR = 8;
constexpr int kFirTile = kFirBlock * kFirR;

struct FirArgs {
    const float2 *x;         // n samples
    const float2 *halo;      // m - 1 samples preceding x[0] (sharded captures) or nullptr (zero history)
    const float2 *taps;      // m taps
    float2 *out;
    int64_t n;
    int m;
    int hist;                // staged history: multiple-of-R tap blocks, >= m - 1
    int64_t tile0;           // first tile of this launch
    // fused magnitude chunk statistics of the OUTPUT (detect_noise_level's O(N) part, AutoInterpretation.py:60-91): chunk k counted
    // from the end = outputs [n - (k+1) chunk, n - k chunk); a tile of 2048 outputs meets at most two chunks (chunk >= kFirTile)
    int64_t chunk, n_chunks; // 0: no statistics
    double *tile_stats;      // [tiles][4] = {sum, max} of the tile's outputs in its first chunk and in the next one
    // k_fir_fast: taps padded with zeros to a multiple of 8 (+ 8), the list of tiles it left to k_fir (non-finite / huge operands)
    const float2 *taps_pad;
    int *redo;               // redo[0] = count, redo[1 + i] = tile
    const int *tile_list;    // k_fir: nullptr = tiles tile0 + blockIdx.x; else the redo list
};

// The products of a tile can only come out as (NaN, NaN) -- the case the reference's complex multiply repairs with
// __mulsc3 -- when an operand is non-finite or two products overflow.  While staging a tile the workgroup records the
// largest |component| of its inputs and taps; if both stay below 2^60 every product and partial sum is finite (m < 2^20
// terms of at most 2^121) and the per-product NaN test is skipped (wave-uniform branch): a third of the instructions.
constexpr uint32_t kFirSafeBits = 0x5d800000u;   // 2^60

typedef float fir_v2f __attribute__((ext_vector_type(2)));

// LDS layout of the staged input: one pad element after every 8 samples.  Lane t reads at 8 t + c (c uniform): without the
// pad that is a 64-byte lane stride -- every 8-byte read of a wavefront lands in 2 of the 32 banks; with it the stride is
// 72 bytes and the 32 lanes of a half-wavefront cover all banks once.
__device__ __forceinline__ int fir_pad(int i) { return i + (i >> 3); }

// Two complex multiply-accumulates accA += xA * h, accB += xB * h with the reference's operation sequence
// (re = fl(fl(x.re h.re) - fl(x.im h.im)), im = fl(fl(x.re h.im) + fl(x.im h.re)), then the two rounded adds into acc) as
// FOUR packed instructions per product instead of the six the compiler emits for the C++ form (it computes t1 - t2 and
// t1 + t2 separately and keeps half of each, because it does not mix neg_lo / neg_hi):
//   t1 = (x.re h.re, x.re h.im)          v_pk_mul  x low half broadcast
//   t2 = (x.im h.im, x.im h.re)          v_pk_mul  x high half broadcast, h halves swapped by op_sel
//   p  = (t1.lo - t2.lo, t1.hi + t2.hi)  v_pk_add  neg_lo on t2 only
//   acc += p                             v_pk_add
// The two products are interleaved so that no instruction reads the result of the one right before it (a dependent
// packed operation needs one wait state on gfx950; the assembler does not add it inside an asm block).
__device__ __forceinline__ void fir_cmac2(fir_v2f &accA, fir_v2f &accB, fir_v2f xA, fir_v2f xB, fir_v2f h) {
    fir_v2f t1a, t2a, t1b, t2b;
    asm volatile(
        "v_pk_mul_f32 %2, %6, %8 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %3, %6, %8 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %4, %7, %8 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %5, %7, %8 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_add_f32 %2, %2, %3 neg_lo:[0,1] neg_hi:[0,0]\n\t"
        "v_pk_add_f32 %4, %4, %5 neg_lo:[0,1] neg_hi:[0,0]\n\t"
        "v_pk_add_f32 %0, %0, %2\n\t"
        "v_pk_add_f32 %1, %1, %4"
        : "+v"(accA), "+v"(accB), "=&v"(t1a), "=&v"(t2a), "=&v"(t1b), "=&v"(t2b)
        : "v"(xA), "v"(xB), "v"(h));
}

template <bool HEAD, bool CHECKED>
__device__ __forceinline__ void fir_accumulate(const FirArgs &a, const float2 *s_taps, const float2 *s_x, int t, int64_t k0,
                                               float2 (&acc)[kFirR]) {
    constexpr int R = kFirR;
    const int jb_top = ((a.m - 1) / R) * R;
    // W[u] = x[k0 - jb - (R-1) + u], u in [0, 2R-1); in LDS: index (R*t + hist) - jb - (R-1) + u
    float2 W[2 * R - 1];
    const int lds0 = R * t + a.hist - (R - 1);
#pragma unroll
    for (int u = 0; u < R - 1; ++u) W[u + R] = s_x[fir_pad(lds0 - jb_top - R + u + R)];   // becomes W[u] after the first shift
    for (int jb = jb_top; jb >= 0; jb -= R) {
#pragma unroll
        for (int u = 0; u < R - 1; ++u) W[u] = W[u + R];
#pragma unroll
        for (int u = R - 1; u < 2 * R - 1; ++u) W[u] = s_x[fir_pad(lds0 - jb + u)];
        if (!HEAD && !CHECKED && jb + R <= a.m) {            // a full block of taps, finite operands: the packed fast path
            float2 hb[R];
#pragma unroll
            for (int tj = 0; tj < R; ++tj) hb[tj] = s_taps[jb + tj];             // LDS broadcast reads (the taps are staged once per tile)
#pragma unroll
            for (int tj = R - 1; tj >= 0; --tj) {
                const fir_v2f h = {hb[tj].x, hb[tj].y};
#pragma unroll
                for (int r = 0; r < R; r += 2) {
                    fir_v2f aA = {acc[r].x, acc[r].y}, aB = {acc[r + 1].x, acc[r + 1].y};
                    const float2 xa = W[r + (R - 1 - tj)], xb = W[r + 1 + (R - 1 - tj)];
                    fir_cmac2(aA, aB, fir_v2f{xa.x, xa.y}, fir_v2f{xb.x, xb.y}, h);
                    acc[r] = make_float2(aA.x, aA.y); acc[r + 1] = make_float2(aB.x, aB.y);
                }
            }
            continue;
        }
#pragma unroll
        for (int tj = R - 1; tj >= 0; --tj) {
            const int j = jb + tj;
            if (j < a.m) {                                   // wave-uniform
                const float2 h = a.taps[j];                  // wave-uniform address: a scalar load, no LDS traffic
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (HEAD && (k0 + r - j < 0)) continue;  // term with i < 0 does not exist
                    const float2 x = W[r + (R - 1 - tj)];
                    float2 p;
                    if (CHECKED) p = cmul(x, h);
                    else { p.x = x.x * h.x - x.y * h.y; p.y = x.x * h.y + x.y * h.x; }
                    acc[r].x += p.x;
                    acc[r].y += p.y;
                }
            }
        }
    }
}

// NaN-propagating maximum (np.max of a chunk that holds a NaN is NaN)
__device__ __forceinline__ double fir_nanmax(double a, double b) { return (a != a || b != b) ? __builtin_nan("") : ((b > a) ? b : a); }

template <bool HEAD, bool STATS>
__device__ __forceinline__ void fir_tile(const FirArgs &a, const int64_t tile, unsigned char *s_raw, uint32_t &s_maxbits, double (*s_st)[4]) {
    float2 *s_taps = (float2 *)s_raw;                       // [m]
    float2 *s_x = s_taps + ((a.m + 1) & ~1);                // [hist + kFirTile], s_x[u] = x[base - hist + u]
    const int t = threadIdx.x;
    const int64_t base = tile * (int64_t)kFirTile;
    if (t == 0) s_maxbits = 0;
    __syncthreads();
    uint32_t mb = 0;                                        // largest |component| seen, as float bits (NaN / inf compare high)
    for (int j = t; j < a.m; j += kFirBlock) {
        const float2 h = a.taps[j];
        s_taps[j] = h;
        mb = max(mb, max(__float_as_uint(h.x) & 0x7fffffffu, __float_as_uint(h.y) & 0x7fffffffu));
    }
    const int total = a.hist + kFirTile;
    for (int u = t; u < total; u += kFirBlock) {
        const int64_t i = base - a.hist + u;
        float2 v = make_float2(0.f, 0.f);
        if (i >= 0) { if (i < a.n) v = a.x[i]; }
        else if (a.halo != nullptr && i + (a.m - 1) >= 0) v = a.halo[i + (a.m - 1)];
        s_x[fir_pad(u)] = v;
        mb = max(mb, max(__float_as_uint(v.x) & 0x7fffffffu, __float_as_uint(v.y) & 0x7fffffffu));
    }
    atomicMax(&s_maxbits, mb);
    __syncthreads();
    const bool safe = s_maxbits < kFirSafeBits;
    constexpr int R = kFirR;
    const int64_t k0 = base + (int64_t)R * t;               // my first output
    if (!STATS && k0 >= a.n) return;
    float2 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_float2(0.f, 0.f);
    if (k0 < a.n) {
        if (safe) fir_accumulate<HEAD, false>(a, s_taps, s_x, t, k0, acc);
        else fir_accumulate<HEAD, true>(a, s_taps, s_x, t, k0, acc);
        if (k0 + R <= a.n) {
#pragma unroll
            for (int r = 0; r < R; r += 2) *(float4 *)(a.out + k0 + r) = make_float4(acc[r].x, acc[r].y, acc[r + 1].x, acc[r + 1].y);
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) if (k0 + r < a.n) a.out[k0 + r] = acc[r];
        }
    }
    if (STATS) {
        // magnitudes of the outputs as util.get_magnitudes computes them (float sqrtf, stored as double), summed per chunk
        const int64_t rem = a.n - a.n_chunks * a.chunk;                  // outputs before `rem` belong to no chunk
        const int64_t i0 = (base > rem) ? base : rem;                    // first output of the tile that counts
        const int64_t kfirst = (i0 < a.n) ? (a.n - 1 - i0) / a.chunk : 0;
        const int64_t split = a.n - kfirst * a.chunk;                    // outputs at or beyond it are in chunk kfirst - 1
        double sum0 = 0.0, sum1 = 0.0, mx0 = 0.0, mx1 = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t i = k0 + r;
            if (i >= rem && i < a.n) {
                const double mg = (double)__builtin_sqrtf(acc[r].x * acc[r].x + acc[r].y * acc[r].y);
                if (i < split) { sum0 += mg; mx0 = fir_nanmax(mx0, mg); } else { sum1 += mg; mx1 = fir_nanmax(mx1, mg); }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            sum0 += __shfl_down(sum0, o); sum1 += __shfl_down(sum1, o);
            mx0 = fir_nanmax(mx0, __shfl_down(mx0, o)); mx1 = fir_nanmax(mx1, __shfl_down(mx1, o));
        }
        if ((t & 63) == 0) { s_st[t >> 6][0] = sum0; s_st[t >> 6][1] = mx0; s_st[t >> 6][2] = sum1; s_st[t >> 6][3] = mx1; }
        __syncthreads();
        if (t == 0) {
            double o0 = 0.0, o1 = 0.0, o2 = 0.0, o3 = 0.0;
            for (int w = 0; w < kFirBlock / 64; ++w) { o0 += s_st[w][0]; o1 = fir_nanmax(o1, s_st[w][1]); o2 += s_st[w][2]; o3 = fir_nanmax(o3, s_st[w][3]); }
            double *dst = a.tile_stats + 4 * tile;
            dst[0] = o0; dst[1] = o1; dst[2] = o2; dst[3] = o3;
        }
    }
}

template <bool HEAD, bool STATS>
__global__ __launch_bounds__(kFirBlock) void k_fir(const FirArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    __shared__ uint32_t s_maxbits;
    __shared__ double s_st[kFirBlock / 64][4];
    if (a.tile_list == nullptr) { fir_tile<HEAD, STATS>(a, a.tile0 + blockIdx.x, s_raw, s_maxbits, s_st); return; }
    const int count = a.tile_list[0];                       // the tiles k_fir_fast handed back
    for (int i = blockIdx.x; i < count; i += gridDim.x) {
        fir_tile<HEAD, STATS>(a, a.tile_list[1 + i], s_raw, s_maxbits, s_st);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_fir_fast: the interior tiles (every output has its full history, the tile lies inside the capture) whose operands are finite
// and below 2^60 -- all of a real capture.  Same arithmetic, arranged for the VALU:
//   * LDS rows of 8 samples + 1 pad; lane t's outputs k0 .. k0 + 7 start a row, so the 8 samples a tap block needs (x[k0 - jb ..
//     k0 - jb + 7]) are ONE row: a base register that moves one row per block and constant offsets (no address arithmetic);
//   * taps from scalar registers (s_load from a zero-padded copy; v_pk_mul_f32 takes the SGPR pair directly): no LDS traffic, no
//     VGPRs for them; a zero tap adds +-0 to an accumulator that is never -0: the padding above tap m - 1 changes nothing;
//   * two row buffers in ping-pong: a block first issues the MACs that use the older row, then reloads that buffer with the NEXT
//     block's row and issues the MACs that use the newer row while the load is in flight; no register moves;
//   * tiles with a non-finite or huge operand are not computed here: their index goes to a list that k_fir works off afterwards
//     (keeping the NaN-recovery path out of this kernel halves its register count);
//   * the first outputs of a capture without halo (k < m - 1: terms with i < 0 do not exist) are computed against a zero history:
//     a zero sample times a finite tap is +-0, and adding +-0 to an accumulator that started at +0 and is never -0 (a sum that
//     cancels exactly rounds to +0) leaves it as it is -- the same identity that lets the taps be padded.
// Order of the terms of one output: tap index descending (= sample index ascending), exactly as in fir_accumulate.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void fir_cmac2s(fir_v2f &accA, fir_v2f &accB, fir_v2f xA, fir_v2f xB, fir_v2f hA, fir_v2f hB) {
    fir_v2f t1a, t2a, t1b, t2b;
    asm volatile(
        "v_pk_mul_f32 %2, %6, %8 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %3, %6, %8 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %4, %7, %9 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %5, %7, %9 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_add_f32 %2, %2, %3 neg_lo:[0,1] neg_hi:[0,0]\n\t"
        "v_pk_add_f32 %4, %4, %5 neg_lo:[0,1] neg_hi:[0,0]\n\t"
        "v_pk_add_f32 %0, %0, %2\n\t"
        "v_pk_add_f32 %1, %1, %4"
        : "+v"(accA), "+v"(accB), "=&v"(t1a), "=&v"(t2a), "=&v"(t1b), "=&v"(t2b)
        : "v"(xA), "v"(xB), "s"(hA), "s"(hB));
}
// the MACs of one tap block in issue order: phase 1 (older row): tj = 7..1, r = 0..tj-1 (28); phase 2 (newer row): tj = 7..0, r = tj..7 (36)
struct FirMac { int r, tj, u; };
__host__ __device__ constexpr FirMac fir_mac1(int idx) {       // idx in [0, 28): sample = old[u]
    int tj = 7;
    while (idx >= tj) { idx -= tj; --tj; }
    return FirMac{idx, tj, idx + 8 - tj};
}
__host__ __device__ constexpr FirMac fir_mac2(int idx) {       // idx in [0, 36): sample = nw[u]
    int tj = 7;
    while (idx >= 8 - tj) { idx -= 8 - tj; --tj; }
    return FirMac{tj + idx, tj, idx};
}
__device__ __forceinline__ void fir_load_row(fir_v2f (&w)[8], const float2 *row) {
#pragma unroll
    for (int c = 0; c < 8; ++c) { const float2 v = row[c]; w[c] = fir_v2f{v.x, v.y}; }
}
// (the indices must be compile-time constants: a register array indexed at run time goes through s_set_gpr_idx / scratch)
template <int I>
__device__ __forceinline__ void fir_pair1(fir_v2f (&acc)[8], const fir_v2f (&old)[8], const fir_v2f (&h)[8]) {
    constexpr FirMac a = fir_mac1(I), b = fir_mac1(I + 1);
    fir_cmac2s(acc[a.r], acc[b.r], old[a.u], old[b.u], h[a.tj], h[b.tj]);
}
template <int I>
__device__ __forceinline__ void fir_pair2(fir_v2f (&acc)[8], const fir_v2f (&nw)[8], const fir_v2f (&h)[8]) {
    constexpr FirMac a = fir_mac2(I), b = fir_mac2(I + 1);
    fir_cmac2s(acc[a.r], acc[b.r], nw[a.u], nw[b.u], h[a.tj], h[b.tj]);
}
template <int... P>
__device__ __forceinline__ void fir_phase1(fir_v2f (&acc)[8], const fir_v2f (&old)[8], const fir_v2f (&h)[8], std::integer_sequence<int, P...>) {
    (fir_pair1<2 * P>(acc, old, h), ...);
}
template <int... P>
__device__ __forceinline__ void fir_phase2(fir_v2f (&acc)[8], const fir_v2f (&nw)[8], const fir_v2f (&h)[8], std::integer_sequence<int, P...>) {
    (fir_pair2<2 * P>(acc, nw, h), ...);
}
// one block of 8 taps: acc[r] += sum over tj = 7..0 of W(r, tj) * h[tj]; `old` is reloaded with *next_row between the phases
__device__ __forceinline__ void fir_block(fir_v2f (&acc)[8], fir_v2f (&old)[8], const fir_v2f (&nw)[8], const float2 *taps8, const float2 *next_row) {
    // uniform address in the constant address space: scalar loads (a plain global load after a barrier is never scalarised)
    typedef const __attribute__((address_space(4))) float *fir_cptr;
    const fir_cptr tc = (fir_cptr)(uintptr_t)taps8;
    fir_v2f h[8];
#pragma unroll
    for (int tj = 0; tj < 8; ++tj) h[tj] = fir_v2f{tc[2 * tj], tc[2 * tj + 1]};
    fir_phase1(acc, old, h, std::make_integer_sequence<int, 14>{});
    if (next_row) fir_load_row(old, next_row);
    fir_phase2(acc, nw, h, std::make_integer_sequence<int, 18>{});
}

template <bool STATS>
__global__ __launch_bounds__(kFirBlock) void k_fir_fast(const FirArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    __shared__ uint32_t s_maxbits;
    __shared__ double s_st[kFirBlock / 64][4];
    constexpr int R = kFirR;
    float2 *s_x = (float2 *)s_raw;                          // rows of 9: s_x[fir_pad(u)] = x[base - hist + u], hist a multiple of 8
    const int t = threadIdx.x;
    const int64_t tile = a.tile0 + blockIdx.x;
    const int64_t base = tile * (int64_t)kFirTile;
    const int jb_top = ((a.m - 1) / R) * R, hist = jb_top + R;
    if (t == 0) s_maxbits = 0;
    __syncthreads();
    uint32_t mb = 0;
    for (int j = t; j < a.m; j += kFirBlock) {
        const float2 h = a.taps[j];
        mb = max(mb, max(__float_as_uint(h.x) & 0x7fffffffu, __float_as_uint(h.y) & 0x7fffffffu));
    }
    const int total = hist + kFirTile;
    for (int u = t; u < total; u += kFirBlock) {
        const int64_t i = base - hist + u;
        float2 v = make_float2(0.f, 0.f);
        if (i >= 0) { if (i < a.n) v = a.x[i]; }
        else if (a.halo != nullptr && i + (a.m - 1) >= 0) v = a.halo[i + (a.m - 1)];     // no halo: zero history (see above)
        s_x[fir_pad(u)] = v;
        mb = max(mb, max(__float_as_uint(v.x) & 0x7fffffffu, __float_as_uint(v.y) & 0x7fffffffu));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mb = max(mb, (uint32_t)__shfl_xor((int)mb, o));
    if ((t & 63) == 0) atomicMax(&s_maxbits, mb);
    __syncthreads();
    if (s_maxbits >= kFirSafeBits) {                        // workgroup-uniform: k_fir's checked arithmetic does this tile
        if (t == 0) a.redo[1 + atomicAdd(&a.redo[0], 1)] = (int)tile;
        return;
    }
    const int64_t k0 = base + (int64_t)R * t;               // my first output
    fir_v2f acc[R], wa[R], wb[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = fir_v2f{0.f, 0.f};
    // row of x[k0 - j .. k0 - j + 7] (j a multiple of 8): t + (hist - j) / 8
    const float2 *row0 = s_x + (t + hist / R) * (R + 1);
    fir_load_row(wa, row0 - (jb_top / R + 1) * (R + 1));    // the row before the top block's (its samples 1..7 are the window's head)
    fir_load_row(wb, row0 - (jb_top / R) * (R + 1));
    int jb = jb_top;
    for (; jb >= R; jb -= 2 * R) {                          // two blocks per trip: (old, new) = (wa, wb), then (wb, wa)
        fir_block(acc, wa, wb, a.taps_pad + jb, row0 - (jb / R - 1) * (R + 1));
        fir_block(acc, wb, wa, a.taps_pad + jb - R, (jb >= 2 * R) ? row0 - (jb / R - 2) * (R + 1) : nullptr);
    }
    if (jb == 0) fir_block(acc, wa, wb, a.taps_pad, nullptr);
    if (k0 + R <= a.n) {
#pragma unroll
        for (int r = 0; r < R; r += 2) *(float4 *)(a.out + k0 + r) = make_float4(acc[r].x, acc[r].y, acc[r + 1].x, acc[r + 1].y);
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) if (k0 + r < a.n) a.out[k0 + r] = make_float2(acc[r].x, acc[r].y);
    }
    if (STATS) {
        const int64_t rem = a.n - a.n_chunks * a.chunk;                  // outputs before `rem` belong to no chunk
        const int64_t i0 = (base > rem) ? base : rem;                    // first output of the tile that counts
        const int64_t kfirst = (i0 < a.n) ? (a.n - 1 - i0) / a.chunk : 0;
        const int64_t split = a.n - kfirst * a.chunk;                    // outputs at or beyond it are in chunk kfirst - 1
        double sum0 = 0.0, sum1 = 0.0, mx0 = 0.0, mx1 = 0.0;
        // Nearly every tile lies inside the capture and on one side of the chunk boundary (a chunk is 1 % of the capture): ONE pair of
        // accumulators and no per-output range tests there -- the same additions in the same order as the general form below, whose
        // other pair would stay at 0.  (The general epilogue -- two pairs, three 64-bit comparisons per output, both pairs through the
        // wavefront reduction -- cost 13 % of the kernel.)
        const bool one_side = base >= rem && base + kFirTile <= a.n && (base + kFirTile <= split || base >= split);     // workgroup-uniform
        if (one_side) {
            double sm = 0.0, mx = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double mg = (double)__builtin_sqrtf(acc[r].x * acc[r].x + acc[r].y * acc[r].y);
                sm += mg; mx = fir_nanmax(mx, mg);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { sm += __shfl_down(sm, o); mx = fir_nanmax(mx, __shfl_down(mx, o)); }
            if (base < split) { sum0 = sm; mx0 = mx; } else { sum1 = sm; mx1 = mx; }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int64_t i = k0 + r;
                if (i >= rem && i < a.n) {
                    const double mg = (double)__builtin_sqrtf(acc[r].x * acc[r].x + acc[r].y * acc[r].y);
                    if (i < split) { sum0 += mg; mx0 = fir_nanmax(mx0, mg); } else { sum1 += mg; mx1 = fir_nanmax(mx1, mg); }
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                sum0 += __shfl_down(sum0, o); sum1 += __shfl_down(sum1, o);
                mx0 = fir_nanmax(mx0, __shfl_down(mx0, o)); mx1 = fir_nanmax(mx1, __shfl_down(mx1, o));
            }
        }
        if ((t & 63) == 0) { s_st[t >> 6][0] = sum0; s_st[t >> 6][1] = mx0; s_st[t >> 6][2] = sum1; s_st[t >> 6][3] = mx1; }
        __syncthreads();
        if (t == 0) {
            double o0 = 0.0, o1 = 0.0, o2 = 0.0, o3 = 0.0;
            for (int w = 0; w < kFirBlock / 64; ++w) { o0 += s_st[w][0]; o1 = fir_nanmax(o1, s_st[w][1]); o2 += s_st[w][2]; o3 = fir_nanmax(o3, s_st[w][3]); }
            double *dst = a.tile_stats + 4 * tile;
            dst[0] = o0; dst[1] = o1; dst[2] = o2; dst[3] = o3;
        }
    }
}
__global__ void k_fir_prepare(const float2 *taps, int m, int m_pad, float2 *taps_pad, int *redo) {
    for (int j = threadIdx.x; j < m_pad; j += blockDim.x) taps_pad[j] = (j < m) ? taps[j] : make_float2(0.f, 0.f);
    if (threadIdx.x == 0) redo[0] = 0;
}

// chunk k = outputs [lo, hi): the tiles that meet it, added in tile order by one wavefront (lane-strided, then a fixed tree)
__global__ __launch_bounds__(64) void k_fir_stats_finish(const double *tile_stats, int64_t n, int64_t chunk, int64_t n_chunks, double *d_sum,
                                                          double *d_max) {
    const int64_t k = blockIdx.x;
    const int64_t lo = n - (k + 1) * chunk, hi = lo + chunk, rem = n - n_chunks * chunk;
    const int64_t t0 = lo / kFirTile, t1 = (hi - 1) / kFirTile;
    double sum = 0.0, mx = 0.0;
    for (int64_t tl = t0 + threadIdx.x; tl <= t1; tl += 64) {
        const int64_t base = tl * kFirTile;
        const int64_t i0 = (base > rem) ? base : rem;
        const int64_t kfirst = (n - 1 - i0) / chunk;                     // the tile's slot 0 chunk; slot 1 is kfirst - 1
        const int slot = (int)(kfirst - k);
        if (slot == 0 || slot == 1) { sum += tile_stats[4 * tl + 2 * slot]; mx = fir_nanmax(mx, tile_stats[4 * tl + 2 * slot + 1]); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sum += __shfl_down(sum, o); mx = fir_nanmax(mx, __shfl_down(mx, o)); }
    if (threadIdx.x == 0) { d_sum[k] = sum; d_max[k] = mx; }
}

size_t fir_stats_scratch_bytes(int64_t n) { return (size_t)((n + kFirTile - 1) / kFirTile + 1) * 4 * sizeof(double); }
// work area of launch_fir: the zero-padded taps and the list of tiles handed from k_fir_fast to k_fir
size_t fir_work_bytes(int64_t n, int m) { return (size_t)(m + 2 * kFirR) * sizeof(float2) + (size_t)((n + kFirTile - 1) / kFirTile + 2) * sizeof(int) + 256; }

template <class K>
static int fir_lds_attr(K kern, size_t lds) {
    if (lds <= 64 * 1024) return URHGPU_OK;
    return hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess ? URHGPU_OK : URHGPU_ERR_HIP;
}

// stats: chunk > 0 asks for the magnitude chunk statistics of the output (chunk >= kFirTile, n_chunks * chunk <= n), written to
// d_sum / d_max[n_chunks]; tile_scratch: fir_stats_scratch_bytes(n); work: fir_work_bytes(n, m) (nullptr: every tile through k_fir)
int launch_fir(const float2 *x, int64_t n, const float2 *taps, int m, const float2 *halo, float2 *out, hipStream_t s, void *work, int64_t chunk,
               int64_t n_chunks, double *d_sum, double *d_max, void *tile_scratch) {
    if (n <= 0) return URHGPU_OK;
    const bool stats = chunk > 0 && n_chunks > 0;
    if (stats && (chunk < kFirTile || n_chunks * chunk > n || !d_sum || !d_max || !tile_scratch)) return URHGPU_ERR_ARG;
    if (m <= 0 && !stats) { return hipMemsetAsync(out, 0, (size_t)n * 8, s) == hipSuccess ? URHGPU_OK : URHGPU_ERR_HIP; }
    if (m <= 0) return URHGPU_ERR_UNSUPPORTED;
    FirArgs a;
    a.x = x; a.halo = halo; a.taps = taps; a.out = out; a.n = n; a.m = m;
    a.chunk = stats ? chunk : 0; a.n_chunks = stats ? n_chunks : 0; a.tile_stats = (double *)tile_scratch;
    a.taps_pad = nullptr; a.redo = nullptr; a.tile_list = nullptr;
    a.hist = ((m - 1) / kFirR) * kFirR + kFirR - 1;
    const size_t lds = (size_t)(((m + 1) & ~1) + (a.hist + kFirTile) + ((a.hist + kFirTile) >> 3) + 1) * 8;
    if (lds > 150 * 1024) return URHGPU_ERR_UNSUPPORTED;       // m <= ~8900 taps
    URH_TRY(fir_lds_attr(k_fir<true, false>, lds)); URH_TRY(fir_lds_attr(k_fir<false, false>, lds));
    URH_TRY(fir_lds_attr(k_fir<true, true>, lds)); URH_TRY(fir_lds_attr(k_fir<false, true>, lds));
    const int64_t tiles = (n + kFirTile - 1) / kFirTile;
    if (tiles >= INT32_MAX) return URHGPU_ERR_UNSUPPORTED;
    if (work != nullptr) {
        // k_fir_fast everywhere, then k_fir for the tiles it handed back (with the i >= 0 test when there is no halo)
        const int m_pad = ((m - 1) / kFirR) * kFirR + kFirR, hist8 = m_pad;
        const size_t lds_fast = (size_t)((hist8 + kFirTile) / kFirR) * (kFirR + 1) * 8;
        URH_TRY(fir_lds_attr(k_fir_fast<false>, lds_fast)); URH_TRY(fir_lds_attr(k_fir_fast<true>, lds_fast));
        float2 *taps_pad = (float2 *)work;
        int *redo = (int *)((char *)work + (((size_t)(m + 2 * kFirR) * sizeof(float2) + 255) & ~size_t(255)));
        hipLaunchKernelGGL(k_fir_prepare, dim3(1), dim3(256), 0, s, taps, m, m_pad, taps_pad, redo);
        a.tile0 = 0; a.taps_pad = taps_pad; a.redo = redo;
        if (stats) hipLaunchKernelGGL((k_fir_fast<true>), dim3((unsigned)tiles), dim3(kFirBlock), lds_fast, s, a);
        else hipLaunchKernelGGL((k_fir_fast<false>), dim3((unsigned)tiles), dim3(kFirBlock), lds_fast, s, a);
        a.tile_list = redo;
        const unsigned gr = (unsigned)std::min<int64_t>(tiles, 1024);       // grid-stride over the list (normally empty)
        const bool head = halo == nullptr;
        if (head && stats) hipLaunchKernelGGL((k_fir<true, true>), dim3(gr), dim3(kFirBlock), lds, s, a);
        else if (head) hipLaunchKernelGGL((k_fir<true, false>), dim3(gr), dim3(kFirBlock), lds, s, a);
        else if (stats) hipLaunchKernelGGL((k_fir<false, true>), dim3(gr), dim3(kFirBlock), lds, s, a);
        else hipLaunchKernelGGL((k_fir<false, false>), dim3(gr), dim3(kFirBlock), lds, s, a);
    } else {
        // tiles that contain outputs k < m - 1 need the i >= 0 test unless a halo supplies the history
        const int64_t head_tiles = (halo == nullptr) ? std::min<int64_t>(tiles, ((int64_t)m - 1 + kFirTile - 1) / kFirTile) : 0;
        if (head_tiles > 0) {
            a.tile0 = 0;
            if (stats) hipLaunchKernelGGL((k_fir<true, true>), dim3((unsigned)head_tiles), dim3(kFirBlock), lds, s, a);
            else hipLaunchKernelGGL((k_fir<true, false>), dim3((unsigned)head_tiles), dim3(kFirBlock), lds, s, a);
        }
        if (tiles > head_tiles) {
            a.tile0 = head_tiles;
            if (stats) hipLaunchKernelGGL((k_fir<false, true>), dim3((unsigned)(tiles - head_tiles)), dim3(kFirBlock), lds, s, a);
            else hipLaunchKernelGGL((k_fir<false, false>), dim3((unsigned)(tiles - head_tiles)), dim3(kFirBlock), lds, s, a);
        }
    }
    if (stats) hipLaunchKernelGGL(k_fir_stats_finish, dim3((unsigned)n_chunks), dim3(64), 0, s, (const double *)tile_scratch, n, chunk, n_chunks, d_sum, d_max);
    return URHGPU_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// IIR (signal_functions.pyx:527-542):  for n >= max(M, N+1):
//   y[n] = (((0 + a0 x[n]) + a1 x[n-1]) + ...) + b0 y[n-1] + b1 y[n-2] ...   every product a complex128 multiply
//   (a[j] + 0j) * complex128(x) rounded to complex64, every += a complex64 add.
// The feed-forward part is a map (k_iir_ff); the feedback part is a serial recurrence with per-step rounding and
// runs on ONE lane (k_iir_fb).  Lowest priority row: no production caller in the reference and no asserting test
// (parity unpinned there; checked here against the oracle / the reference build).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float2 iir_term(double coef, float2 v) {
    // (coef + 0j) * (v.x + v.y j) in complex128, rounded to complex64
    const double re = coef * (double)v.x - 0.0 * (double)v.y;
    const double im = coef * (double)v.y + 0.0 * (double)v.x;
    return make_float2((float)re, (float)im);
}

__global__ void k_iir_ff(const double *a, int64_t M, const float2 *x, int64_t n, int64_t start, float2 *y) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float2 acc = make_float2(0.f, 0.f);
        if (i >= start) {
            for (int64_t j = 0; j < M; ++j) {
                const float2 p = iir_term(a[j], x[i - j]);
                acc.x += p.x; acc.y += p.y;
            }
        }
        y[i] = acc;
    }
}

__global__ void k_iir_fb(const double *b, int64_t N, int64_t n, int64_t start, float2 *y) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int64_t i = start; i < n; ++i) {
        float2 acc = y[i];
        for (int64_t k = 0; k < N; ++k) {
            const float2 p = iir_term(b[k], y[i - 1 - k]);
            acc.x += p.x; acc.y += p.y;
        }
        y[i] = acc;
    }
}

int launch_iir(const double *a, int64_t M, const double *b, int64_t N, const float2 *x, int64_t n, float2 *y, hipStream_t s) {
    if (n <= 0) return URHGPU_OK;
    const int64_t start = std::max<int64_t>(M, N + 1);
    const int grid = (int)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_iir_ff, dim3(grid), dim3(256), 0, s, a, M, x, n, start, y);
    if (N > 0 && start < n) hipLaunchKernelGGL(k_iir_fb, dim3(1), dim3(64), 0, s, b, N, n, start, y);
    return URHGPU_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Magnitudes (util.pyx:128-136): float input -> (double) sqrtf(I*I + Q*Q) in fp32; integer input -> products and sum
// in C `int` (wrapping, as the reference's generated code), sqrt in double.
// ---------------------------------------------------------------------------------------------------------------
template <int DT> struct MagLoad;
template <> struct MagLoad<URHGPU_DT_F32> {
    static __device__ __forceinline__ double mag(const void *p, int64_t i) {
        const float2 v = ((const float2 *)p)[i];
        return (double)__builtin_sqrtf(v.x * v.x + v.y * v.y);
    }
};
template <class T2> __device__ __forceinline__ double int_mag(int re, int im) {
    const int s = (int)((unsigned)(re * re) + (unsigned)(im * im));
    return __builtin_sqrt((double)s);
}
template <> struct MagLoad<URHGPU_DT_I8> {
    static __device__ __forceinline__ double mag(const void *p, int64_t i) { const char2 v = ((const char2 *)p)[i]; return int_mag<void>(v.x, v.y); }
};
template <> struct MagLoad<URHGPU_DT_U8> {
    static __device__ __forceinline__ double mag(const void *p, int64_t i) { const uchar2 v = ((const uchar2 *)p)[i]; return int_mag<void>(v.x, v.y); }
};
template <> struct MagLoad<URHGPU_DT_I16> {
    static __device__ __forceinline__ double mag(const void *p, int64_t i) { const short2 v = ((const short2 *)p)[i]; return int_mag<void>(v.x, v.y); }
};
template <> struct MagLoad<URHGPU_DT_U16> {
    static __device__ __forceinline__ double mag(const void *p, int64_t i) {
        const ushort2 v = ((const ushort2 *)p)[i];
        // 65535^2 overflows C int: wrap like the reference (unsigned arithmetic, same two's-complement bits)
        const int s = (int)((unsigned)v.x * (unsigned)v.x + (unsigned)v.y * (unsigned)v.y);
        return __builtin_sqrt((double)s);
    }
};

template <int DT>
__global__ __launch_bounds__(256) void k_magnitudes(const void *iq, int64_t n, double *out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = MagLoad<DT>::mag(iq, i);
}

// Chunk statistics: chunk k (counted from the END of the capture) = samples [n - (k+1)*chunk, n - k*chunk).
// Two deterministic stages: workgroup (slice, k) reduces one slice of chunk k to (sum, max); then one lane per
// chunk adds the slices in order.  Sums are fp64 (the reference's np.mean is an fp64 pairwise sum: the fp32-cast
// means agree unless an fp64 sum lands within an ulp of an fp32 rounding boundary).
constexpr int kMagSlices = 32;

template <int DT>
__global__ __launch_bounds__(256) void k_mag_chunk_partials(const void *iq, int64_t n, int64_t chunk, double *part_sum, double *part_max) {
    __shared__ double s_sum[4], s_max[4];
    const int64_t k = blockIdx.y;
    const int64_t lo = n - (k + 1) * chunk, hi = lo + chunk;
    const int64_t L = (chunk + kMagSlices - 1) / kMagSlices;
    const int64_t a0 = lo + (int64_t)blockIdx.x * L;
    const int64_t a1 = (a0 + L < hi) ? a0 + L : hi;
    double sum = 0.0, mx = 0.0;
    bool any_nan = false;
    for (int64_t i = a0 + threadIdx.x; i < a1; i += 256) {
        const double v = MagLoad<DT>::mag(iq, i);
        sum += v;
        if (v != v) any_nan = true; else mx = (v > mx) ? v : mx;
    }
    if (any_nan) mx = __builtin_nan("");
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sum += __shfl_down(sum, o);
        const double om = __shfl_down(mx, o);
        mx = (om != om || mx != mx) ? __builtin_nan("") : ((om > mx) ? om : mx);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_sum[wave] = sum; s_max[wave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ts = 0.0, tm = 0.0; bool nan = false;
        for (int w = 0; w < 4; ++w) { ts += s_sum[w]; if (s_max[w] != s_max[w]) nan = true; else tm = (s_max[w] > tm) ? s_max[w] : tm; }
        part_sum[k * kMagSlices + blockIdx.x] = ts;
        part_max[k * kMagSlices + blockIdx.x] = nan ? __builtin_nan("") : tm;
    }
}

__global__ void k_mag_chunk_finish(const double *part_sum, const double *part_max, int64_t n_chunks, double *d_sum, double *d_max) {
    const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (k >= n_chunks) return;
    double ts = 0.0, tm = 0.0; bool nan = false;
    for (int s = 0; s < kMagSlices; ++s) {
        ts += part_sum[k * kMagSlices + s];
        const double m = part_max[k * kMagSlices + s];
        if (m != m) nan = true; else tm = (m > tm) ? m : tm;
    }
    d_sum[k] = ts;
    d_max[k] = nan ? __builtin_nan("") : tm;
}

template <int DT>
static void launch_mag_dt(const void *iq, int64_t n, double *out, int64_t chunk, int64_t n_chunks, double *ps, double *pm,
                          double *d_sum, double *d_max, hipStream_t s) {
    if (out) {
        const int grid = (int)std::min<int64_t>((n + 255) / 256, 8192);
        hipLaunchKernelGGL(k_magnitudes<DT>, dim3(grid), dim3(256), 0, s, iq, n, out);
    } else {
        hipLaunchKernelGGL(k_mag_chunk_partials<DT>, dim3(kMagSlices, (unsigned)n_chunks), dim3(256), 0, s, iq, n, chunk, ps, pm);
        hipLaunchKernelGGL(k_mag_chunk_finish, dim3((unsigned)((n_chunks + 63) / 64)), dim3(64), 0, s, ps, pm, n_chunks, d_sum, d_max);
    }
}

static int launch_mag_any(int dtype, const void *iq, int64_t n, double *out, int64_t chunk, int64_t n_chunks, double *ps,
                          double *pm, double *d_sum, double *d_max, hipStream_t s) {
    switch (dtype) {
        case URHGPU_DT_F32: launch_mag_dt<URHGPU_DT_F32>(iq, n, out, chunk, n_chunks, ps, pm, d_sum, d_max, s); return URHGPU_OK;
        case URHGPU_DT_I8: launch_mag_dt<URHGPU_DT_I8>(iq, n, out, chunk, n_chunks, ps, pm, d_sum, d_max, s); return URHGPU_OK;
        case URHGPU_DT_U8: launch_mag_dt<URHGPU_DT_U8>(iq, n, out, chunk, n_chunks, ps, pm, d_sum, d_max, s); return URHGPU_OK;
        case URHGPU_DT_I16: launch_mag_dt<URHGPU_DT_I16>(iq, n, out, chunk, n_chunks, ps, pm, d_sum, d_max, s); return URHGPU_OK;
        case URHGPU_DT_U16: launch_mag_dt<URHGPU_DT_U16>(iq, n, out, chunk, n_chunks, ps, pm, d_sum, d_max, s); return URHGPU_OK;
        default: return URHGPU_ERR_DTYPE;
    }
}

int launch_magnitudes(const void *iq, int dtype, int64_t n, double *out, hipStream_t s) {
    if (n <= 0) return URHGPU_OK;
    return launch_mag_any(dtype, iq, n, out, 0, 0, nullptr, nullptr, nullptr, nullptr, s);
}

size_t mag_chunk_scratch_bytes(int64_t n_chunks) { return (size_t)n_chunks * kMagSlices * 16 + 512; }

int launch_mag_chunk_stats(const void *iq, int dtype, int64_t n, int64_t chunk, int64_t n_chunks, double *d_sum, double *d_max,
                           void *scratch, hipStream_t s) {
    if (n_chunks <= 0) return URHGPU_OK;
    if (chunk <= 0 || n_chunks * chunk > n || n_chunks > 65535) return URHGPU_ERR_ARG;
    double *ps = (double *)scratch;
    double *pm = ps + n_chunks * kMagSlices;
    return launch_mag_any(dtype, iq, n, nullptr, chunk, n_chunks, ps, pm, d_sum, d_max, s);
}

}  // namespace urh
