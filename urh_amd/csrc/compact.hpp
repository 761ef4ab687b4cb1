// compact.hpp -- layout of the compact result blob (include/urhgpu.h: urhgpu_outputs::blob), shared by the kernel that writes it
// (compact.hip) and the host code that sizes the D2H copy and hands out the sections (stream.hip).
#pragma once
#include <stdint.h>

#include "../../include/urhgpu.h"

namespace urh {

struct BlobLayout {          // byte offsets of the sections inside the blob; every section starts 16-byte aligned
    int64_t n_rows, n_msg, n_bits, n_pos;     // element counts actually stored (clamped to the capacities)
    int64_t off_pauses, off_msg_off, off_pos_off, off_row_state, off_bits, off_row_len, off_pos32, total;
};

// counts5: {n_rows, n_msg, n_bits, n_pos, rows_needed} as urhgpu_outputs::counts holds them (totals, possibly beyond the capacities)
__host__ __device__ inline BlobLayout blob_layout(const int64_t *counts5, int64_t cap_rows, int64_t cap_bits, int64_t cap_msg, int64_t cap_pos,
                                                  int has_pos) {
    BlobLayout L;
    L.n_rows = counts5[0] < cap_rows ? counts5[0] : cap_rows;
    L.n_msg = counts5[1] < cap_msg ? counts5[1] : cap_msg;
    L.n_bits = counts5[2] < cap_bits ? counts5[2] : cap_bits;
    L.n_pos = has_pos ? (counts5[3] < cap_pos ? counts5[3] : cap_pos) : 0;
    if (L.n_rows < 0) L.n_rows = 0;
    if (L.n_msg < 0) L.n_msg = 0;
    if (L.n_bits < 0) L.n_bits = 0;
    if (L.n_pos < 0) L.n_pos = 0;
    auto up = [](int64_t x) { return (x + 15) & ~int64_t(15); };
    int64_t o = URHGPU_BLOB_HEADER_BYTES;
    L.off_pauses = o; o = up(o + L.n_msg * 8);
    L.off_msg_off = o; o = up(o + (L.n_msg + 1) * 8);
    L.off_pos_off = o; o = up(o + (L.n_msg + 1) * 8);
    L.off_row_state = o; o = up(o + L.n_rows);
    L.off_bits = o; o = up(o + (L.n_bits + 7) / 8);
    L.off_row_len = o; o = up(o + L.n_rows * 4);
    L.off_pos32 = o; o = up(o + L.n_pos * 4);
    L.total = o;
    return L;
}

// Staged passes (urhgpu_stream_*, stream_policy 6): the blob in a SPLIT layout -- a head region (header + pauses / msg_off / pos_off / bits,
// tight by their counts: what the pass's last kernel writes) and, behind the head's capacity, the three big sections at offsets the
// CAPACITIES give (row_state, row_len: stored by the row kernel as it emits rows; pos32), so that the rows can be copied to the host
// while the bits are still being expanded.  Same total as blob_capacity (the same sections in another order).
struct StagedLayout { int64_t head_cap, off_row_state, off_row_len, off_pos32, total; };
__host__ __device__ inline StagedLayout staged_layout(int64_t cap_rows, int64_t cap_bits, int64_t cap_msg, int64_t cap_pos, int has_pos) {
    const int64_t c[5] = {0, cap_msg, cap_bits, 0, 0};
    auto up = [](int64_t x) { return (x + 15) & ~int64_t(15); };
    StagedLayout S;
    S.head_cap = blob_layout(c, cap_rows, cap_bits, cap_msg, cap_pos, 0).total;
    S.off_row_state = S.head_cap;
    S.off_row_len = up(S.off_row_state + cap_rows);
    S.off_pos32 = up(S.off_row_len + 4 * cap_rows);
    S.total = up(S.off_pos32 + (has_pos ? 4 * cap_pos : 0));
    return S;
}

// capacity of a blob for the given output capacities (what the caller allocates)
inline int64_t blob_capacity(int64_t cap_rows, int64_t cap_bits, int64_t cap_msg, int64_t cap_pos, int has_pos) {
    const int64_t c[5] = {cap_rows, cap_msg, cap_bits, cap_pos, cap_rows};
    return blob_layout(c, cap_rows, cap_bits, cap_msg, cap_pos, has_pos).total;
}

}  // namespace urh
