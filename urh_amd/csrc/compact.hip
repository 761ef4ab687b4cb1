// compact.hip -- the compact form of one pass's results, for the trip over PCIe (the timing window ends with the
// compact outputs on the host).  The wide device outputs mirror the reference's Python objects (int64 [state, length] rows as
// grab_pulse_lens returns them, one byte per bit, int64 positions: 22.8 MB per GiB of 2-FSK capture -- 0.5 ms of PCIe, more than
// the whole device pass); the blob holds the same information in 8.9 MB (3.5 MB without bit_sample_pos):
//   header   int64[16]  {magic, n_rows, n_msg, n_bits, n_pos, rows_needed, total_bytes, has_pos, section offsets ...}
//   pauses   int64[n_msg]        msg_off  int64[n_msg + 1]  (bit offsets)        pos_off  int64[n_msg + 1]
//   row_state int8[n_rows]       (state -1 = pause; modulation orders up to 128)
//   bits     uint8[(n_bits + 7) / 8]   eight bits per byte, most significant first (numpy.packbits order), messages back to back
//   row_len  int32[n_rows]       (a row is at most the capture: < 2^31 samples)
//   pos32    uint32[n_pos]       bit_sample_pos (absolute sample positions; omitted when the pass did not write positions)
// One kernel after the pass's last kernel: every section's offset follows from the counts, so ONE contiguous copy of
// header.total_bytes moves everything (the host learns the counts from a 40-byte copy first).
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "compact.hpp"
#include "launchers.hpp"

namespace urh {

struct PackArgs {
    const int64_t *rows; const uint8_t *bits; const int64_t *msg_off, *pauses, *pos_off, *pos; const int64_t *counts;
    int64_t cap_rows, cap_bits, cap_msg, cap_pos;
    int has_pos;
    char *blob; int64_t cap_blob;
};

__global__ __launch_bounds__(256) void k_pack_blob(const PackArgs a) {
    URH_TAIL_PRIO();
    const BlobLayout L = blob_layout(a.counts, a.cap_rows, a.cap_bits, a.cap_msg, a.cap_pos, a.has_pos);
    const int64_t gtid = blockIdx.x * 256ll + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    int64_t *hdr = (int64_t *)a.blob;
    const bool fits = L.total <= a.cap_blob;
    if (gtid == 0) {
        hdr[0] = URHGPU_BLOB_MAGIC; hdr[1] = L.n_rows; hdr[2] = L.n_msg; hdr[3] = L.n_bits; hdr[4] = L.n_pos; hdr[5] = a.counts[4];
        hdr[6] = fits ? L.total : -L.total; hdr[7] = a.has_pos;
        hdr[8] = L.off_pauses; hdr[9] = L.off_msg_off; hdr[10] = L.off_pos_off; hdr[11] = L.off_row_state; hdr[12] = L.off_bits;
        hdr[13] = L.off_row_len; hdr[14] = L.off_pos32;
        // truncated (bit 0); bit 2 (a value that does not fit its narrow type) is OR-ed in below: the launcher has zeroed the word
        if (a.counts[1] > a.cap_msg || a.counts[2] > a.cap_bits || (a.has_pos && a.counts[3] > a.cap_pos) || a.counts[4] > a.cap_rows)
            atomicOr((unsigned long long *)&hdr[15], 1ull);
    }
    bool narrow_fail = false;
    if (!fits) return;                                     // (cannot happen with a blob of blob_capacity bytes)
    // bits: eight bytes in, one byte out; the bits beyond n_bits of the last byte are zero
    {
        const int64_t nb = (L.n_bits + 7) / 8;
        uint8_t *out = (uint8_t *)(a.blob + L.off_bits);
        const unsigned long long *in = (const unsigned long long *)a.bits;       // 8-byte aligned (checked by the launcher)
        for (int64_t j = gtid; j < nb; j += stride) {
            unsigned long long w;
            if (8 * j + 8 <= L.n_bits) w = in[j];
            else { w = 0; for (int k = 0; k < 8 && 8 * j + k < L.n_bits; ++k) w |= (unsigned long long)a.bits[8 * j + k] << (8 * k); }
            out[j] = (uint8_t)(((w & 0x0101010101010101ull) * 0x8040201008040201ull) >> 56);
        }
    }
    {
        int32_t *len = (int32_t *)(a.blob + L.off_row_len);
        int8_t *st = (int8_t *)(a.blob + L.off_row_state);
        for (int64_t i = gtid; i < L.n_rows; i += stride) {
            const longlong2 r = *(const longlong2 *)(a.rows + 2 * i);
            // (a rank's piece of a sharded ASK capture may start with a row that was merged into the previous rank's last one:
            // URHGPU_ROW_ABSORBED, shipped as state -128)
            const bool absorbed = (r.x == kRowAbsorbed);
            st[i] = absorbed ? (int8_t)-128 : (int8_t)r.x; len[i] = (int32_t)r.y;
            // (a length may be NEGATIVE: the reference's last row is pulse_length - tolerance, signal_functions.pyx:485-493, below zero for a
            // capture shorter than the tolerance; int32 holds that)
            narrow_fail |= ((!absorbed && (r.x < -127 || r.x > 127)) || r.y < -0x80000000ll || r.y > 0x7fffffffll);
        }
    }
    if (a.has_pos) {
        uint32_t *p32 = (uint32_t *)(a.blob + L.off_pos32);
        for (int64_t i = gtid; i < L.n_pos; i += stride) {
            const int64_t v = a.pos[i];
            p32[i] = (uint32_t)v;
            narrow_fail |= (v < 0 || v > 0xffffffffll);
        }
    }
    if (narrow_fail) atomicOr((unsigned long long *)&hdr[15], 4ull);
    {
        int64_t *pa = (int64_t *)(a.blob + L.off_pauses), *mo = (int64_t *)(a.blob + L.off_msg_off), *po = (int64_t *)(a.blob + L.off_pos_off);
        for (int64_t i = gtid; i <= L.n_msg; i += stride) {
            if (i < L.n_msg) pa[i] = a.pauses[i];
            mo[i] = a.msg_off[i]; po[i] = a.pos_off[i];
        }
    }
}

int launch_pack_blob(const urhgpu_outputs *o, int write_pos, hipStream_t s) {
    if (!o->blob) return URHGPU_OK;
    if (!o->rows || !o->bits || !o->msg_off || !o->pauses || !o->pos_off || !o->counts) return URHGPU_ERR_ARG;
    if (((uintptr_t)o->bits & 7) || ((uintptr_t)o->blob & 15) || ((uintptr_t)o->rows & 15)) return URHGPU_ERR_ARG;
    const int has_pos = (write_pos && o->pos) ? 1 : 0;
    if (o->cap_blob < blob_capacity(o->cap_rows, o->cap_bits, o->cap_msg, o->cap_pos, has_pos)) return URHGPU_ERR_CAPACITY;
    PackArgs a{o->rows, o->bits, o->msg_off, o->pauses, o->pos_off, o->pos, o->counts, o->cap_rows, o->cap_bits, o->cap_msg, o->cap_pos, has_pos,
               (char *)o->blob, o->cap_blob};
    // sized for a typical result (a few million elements), stride loops for the rest: the counts are only known on the device
#ifndef URH_PACK_BLOCKS
#define URH_PACK_BLOCKS 512
#endif
    if (hipMemsetAsync((char *)o->blob + 15 * 8, 0, 8, s) != hipSuccess) return URHGPU_ERR_HIP;       // header[15]: the kernel ORs its flags in
    hipLaunchKernelGGL(k_pack_blob, dim3(URH_PACK_BLOCKS), dim3(256), 0, s, a);
    return URHGPU_OK;
}

// ---- measurement hook: what a PURE COPY with the hot kernel's access shape gets out of the HBM on this box (bench.py reports it
// next to the 8 TB/s spec figure, SURVEY 8(d): "report both denominators") -------------------------------------------------------
// shape 0: one workgroup of four wavefronts per 8192 samples, every wavefront streams its own 2048 samples (16 KiB in, 8 KiB out),
//          lane = 2 consecutive samples per row (16-byte non-temporal load, 8-byte non-temporal store), loads two rows ahead -- the
//          structure of k_demod_runs_bp without its arithmetic;  shape 1: the plain grid-stride float4 copy of the guide.
typedef float cp_v4 __attribute__((ext_vector_type(4)));
typedef float cp_v2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_copy_shape(const float *in, float *out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const float *src = in + wave * 16 * 256 + lane * 4;
    float *dst = out + wave * 16 * 128 + lane * 2;
    cp_v4 cur[2], nxt[2];
    cur[0] = __builtin_nontemporal_load((const cp_v4 *)src);
    cur[1] = __builtin_nontemporal_load((const cp_v4 *)(src + 256));
#pragma unroll 1
    for (int r = 0; r < 16; r += 2) {
        if (r + 2 < 16) {
            nxt[0] = __builtin_nontemporal_load((const cp_v4 *)(src + (int64_t)(r + 2) * 256));
            nxt[1] = __builtin_nontemporal_load((const cp_v4 *)(src + (int64_t)(r + 3) * 256));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const cp_v2 t = {cur[j].x + cur[j].y, cur[j].z + cur[j].w};
            __builtin_nontemporal_store(t, (cp_v2 *)(dst + (int64_t)(r + j) * 128));
        }
        cur[0] = nxt[0]; cur[1] = nxt[1];
    }
}
__global__ __launch_bounds__(256) void k_copy_plain(const cp_v4 *in, cp_v4 *out, int64_t n4) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) out[i] = in[i];
}
void launch_copy_shape(const float *in, float *out, int64_t n_samples, int shape, hipStream_t s) {
    if (shape == 0) hipLaunchKernelGGL(k_copy_shape, dim3((unsigned)(n_samples / 8192)), dim3(256), 0, s, in, out);
    else hipLaunchKernelGGL(k_copy_plain, dim3(256 * 20), dim3(256), 0, s, (const cp_v4 *)in, (cp_v4 *)out, n_samples / 4);
}

}  // namespace urh
