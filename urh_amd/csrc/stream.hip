// stream.hip -- a stream of captures, results on the host (include/urhgpu.h: urhgpu_stream_*).
//
// SURVEY §8(d)'s window for the IQ -> bits path is "IQ resident in HBM ... compact outputs on the host".  One capture at a time that is
// hot kernel + tail + copy, strictly one after the other (0.44 + 0.4 ms per GiB in round 2).  A consumer that processes capture
// after capture -- the reference's own live mode is one (ProtocolSniffer.py:161-202) -- lets three things overlap:
//     hot kernel of pass i + 1  | tail of pass i (second stream, urhgpu_ctx_set_pipelined) | pack + D2H of pass i - 1 (third stream)
// Three output slots rotate; the compact blob (compact.hip) makes the copy ONE hipMemcpyAsync of 9 MB per GiB (3.5 MB without
// bit_sample_pos) into pinned memory, which hides under the 0.29 ms hot kernel.  The host never waits for the GPU inside push() beyond
// the bounded run-ahead of pipelined passes: the copy is queued behind the tail with a predicted size (queue_copy).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <new>

#include "common.hpp"
#include "compact.hpp"
#include "launchers.hpp"

using namespace urh;

struct urhgpu_stream {
    urhgpu_ctx *ctx = nullptr;
    urhgpu_params p;
    int want_qad = 0, want_pos = 0;
    int64_t n_max = 0, cap_rows = 0, cap_bits = 0, cap_msg = 0, cap_pos = 0, cap_blob = 0;
    struct Slot {
        void *dev = nullptr;               // one allocation: rows | bits | msg_off | pauses | pos_off | pos | counts | blob | staging blob
        char *stage = nullptr;             // staged passes: the compact sections in the capacity layout, in HBM (tightened into `blob`)
        float *qad = nullptr;              // the qad buffer (of the ring below) this slot's current pass wrote
        urhgpu_outputs out;
        char *h_blob2[2] = {nullptr, nullptr};   // pinned, used alternately by the slot's passes: the result handed out at push i (pass i - 3)
                                                 // stays untouched while pass i's copy lands in the other one
        char *h_blob = nullptr;            // the one the slot's current pass copies into
        int64_t *h_counts = nullptr;       // pinned int64[8]
        hipEvent_t ev_tail = nullptr, ev_copy = nullptr, ev_rows = nullptr, ev_shipped = nullptr;   // ev_shipped: a staged pass's copies are through
        int64_t seq = -1, n = 0, copied = 0;
        bool staged = false;               // the pass's blob arrived in the split layout: copied = head bytes, copied_rows / copied_pos = elements
        int64_t copied_rows = 0, copied_pos = 0;
        int state = 0;                     // 0 free, 1 pass launched (tail pending), 2 copy issued, 3 result handed out
    } slot[3];
    // The demodulated signal of pass i lives in qad_ring[i % 4]: four buffers for three result slots, so that the result handed out by
    // push i (pass i - 3) still owns its qad while pass i's hot kernel writes another one -- it is overwritten by push i + 1.
    float *qad_ring[4] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t copy_stream = nullptr;
    int64_t seq = 0;
    int64_t streamed_passes = 0;           // passes whose tail ran in segments (diagnostics)
    bool len16 = false;                    // staged passes ship 16-bit row lengths + an escape list (the staging blob's head region holds the list)
    bool row16_ok = false;                 // ... and may pack state and length into ONE uint16 per row (URHGPU_BLOB_ROW16) when the pulse table is dense: the blobs have room for that list
    int64_t esc_extra = 0;                 // bytes behind the split layout's sections for URHGPU_BLOB_ROW16's escape list (staging and host blobs)
    int64_t last_rows = 0;                 // rows of the last result handed out (dense: more than one row per 64 samples)
    // signed integer FSK captures: a one-workgroup probe behind every pass (k_wide_probe) reports the share of wide phase steps into pinned
    // memory; the pushes that follow read it and take the hot kernel's instantiation with the wide loop from 1 % on (back below 0.3 %) --
    // a quarter faster on wide captures, 5 % slower on narrow ones (profiles/r06s_deviation_pmc.txt), so it has to be chosen
    int32_t *h_probe = nullptr;            // pinned: {per mille of wide pairs, pairs counted}
    int wide_int = 0;
    int64_t wide_passes = 0;               // passes launched with it (urhgpu_stream_wide_passes)
    hipEvent_t ev_probe = nullptr;         // behind the last probe (it reads the capture: flush waits for it before the caller may let go of d_iq)
    bool probe_pending = false;
    int64_t staged_passes = 0;             // passes whose tail stored into the staging blob (tightened + copied by the copy engine)
    int64_t uploaded_passes = 0;           // ... of which the capture was uploaded piece by piece (urhgpu_stream_push_upload)
    int64_t predicted_bytes = 0;           // blob bytes the next pass's copy is sized for (0: header only, the rest fetched on demand)
    int64_t predicted_rows = 0, predicted_pos = 0;   // staged passes: rows, positions the next pass's copies are sized for
    int64_t short_copies = 0;              // passes whose prediction fell short (diagnostics)
    bool was_pipelined = false;
};

namespace {

size_t a256(size_t x) { return (x + 255) & ~size_t(255); }

void fill_result(urhgpu_stream *st, urhgpu_stream::Slot &s, urhgpu_host_result *r) {
    memset(r, 0, sizeof(*r));
    r->seq = s.seq;
    const int64_t *hdr = (const int64_t *)s.h_blob;
    r->n_samples = s.n;
    r->n_rows = hdr[1]; r->n_msg = hdr[2]; r->n_bits = hdr[3]; r->n_pos = hdr[4]; r->rows_needed = hdr[5];
    r->blob_bytes = hdr[6] < 0 ? -hdr[6] : hdr[6];
    r->truncated = (int)hdr[15];
    r->pauses = (const int64_t *)(s.h_blob + hdr[8]);
    r->msg_off = (const int64_t *)(s.h_blob + hdr[9]);
    r->pos_off = (const int64_t *)(s.h_blob + hdr[10]);
    r->row_state = (const int8_t *)(s.h_blob + hdr[11]);
    r->bits_packed = (const uint8_t *)(s.h_blob + hdr[12]);
    r->row_len = (const int32_t *)(s.h_blob + hdr[13]);
    r->row_len16 = nullptr; r->esc = nullptr; r->n_esc = 0; r->row16 = nullptr;
    if (hdr[7] & URHGPU_BLOB_ROW16) {                        // state | length words in the row_len section, the escape list at header[11]
        r->row_len = nullptr; r->row_state = nullptr;
        r->row16 = (const uint16_t *)(s.h_blob + hdr[13]);
        const int64_t *e = (const int64_t *)(s.h_blob + hdr[11]);
        r->n_esc = e[0] < 0 ? -e[0] : e[0];
        r->esc = e + 1;
    } else if (hdr[7] & URHGPU_BLOB_LEN16) {                        // 16-bit lengths + the escape list behind the packed bits
        r->row_len = nullptr;
        r->row_len16 = (const uint16_t *)(s.h_blob + hdr[13]);
        const int64_t *e = (const int64_t *)(s.h_blob + ((hdr[12] + (hdr[3] + 7) / 8 + 15) & ~int64_t(15)));
        r->n_esc = e[0] < 0 ? -e[0] : e[0];
        r->esc = e + 1;
    }
    r->pos32 = (hdr[7] & 1) ? (const uint32_t *)(s.h_blob + hdr[14]) : nullptr;
    r->blob = s.h_blob;
    r->d_qad = s.qad;
    (void)st;
}

// Queue the pack kernel and the copy of the slot's blob behind the pass's tail, WITHOUT involving the host: the copy stream waits for
// the tail's event, and the number of bytes to copy is PREDICTED from the pass before (captures of a stream resemble each other:
// the last blob's size plus an eighth).  The host therefore never waits for the GPU inside push() -- it runs ahead as in the
// device-only loop (waiting for the counts first made the host the pacemaker: it could queue the next hot kernel only after the tail of
// the pass before last had finished, 0.31 ms per step instead of 0.296).  finish_copy() fetches what a prediction missed.
int queue_copy(urhgpu_stream *st, urhgpu_stream::Slot &s) {
    if (s.state != 1) return URHGPU_OK;
    URH_HIP(hipStreamWaitEvent(st->copy_stream, s.ev_tail, 0));
    URH_TRY(launch_pack_blob(&s.out, st->want_pos, st->copy_stream));
    URH_HIP(hipGetLastError());
    int64_t guess = st->predicted_bytes > 0 ? st->predicted_bytes : URHGPU_BLOB_HEADER_BYTES;
    if (guess > st->cap_blob) guess = st->cap_blob;
    URH_HIP(hipMemcpyAsync(s.h_blob, s.out.blob, (size_t)guess, hipMemcpyDeviceToHost, st->copy_stream));
    URH_HIP(hipEventRecord(s.ev_copy, st->copy_stream));
    s.copied = guess;
    s.state = 2;
    return URHGPU_OK;
}

// (a bounded hipEventQuery spin in front of this was measured slower in round 4: 0.420 against 0.408 ms for one capture; again in round 6 on the
// K = 20 loop, tools/k20_probe.py: 0.2831-0.2840 with it, 0.2832-0.2837 without)
int wait_event(hipEvent_t e) {
    URH_HIP(hipEventSynchronize(e));
    return URHGPU_OK;
}

int finish_copy(urhgpu_stream *st, urhgpu_stream::Slot &s, urhgpu_host_result *r) {
    if (s.state == 1) URH_TRY(queue_copy(st, s));
    if (s.state != 2) return URHGPU_ERR_ARG;
    URH_TRY(wait_event(s.ev_copy));
    if (s.staged) URH_TRY(wait_event(s.ev_shipped));
    const int64_t *hdr = (const int64_t *)s.h_blob;
    if (hdr[0] != URHGPU_BLOB_MAGIC || hdr[6] < 0 || hdr[6] > st->cap_blob) return URHGPU_ERR_ARG;
    if (hdr[15] & 2) {                                      // a segment's gate gave up waiting for the hot kernel (k_seg_gate): nothing of this pass is valid
        snprintf(urh::g_hip_err, sizeof(urh::g_hip_err), "streamed pass %lld: a segment waited 2 s for the hot kernel", (long long)s.seq);
        return URHGPU_ERR_HIP;
    }
    if (s.staged) {
        // split layout: what the predictions missed, now -- the rest of the head (it ends with the packed bits), of the two row sections, of the positions
        const StagedLayout SL = staged_layout(st->cap_rows, st->cap_bits, st->cap_msg, st->cap_pos, st->want_pos);
        const int64_t n_rows = hdr[1], n_pos = (hdr[7] & 1) ? hdr[4] : 0;
        const int64_t len_bytes = (hdr[7] & (URHGPU_BLOB_LEN16 | URHGPU_BLOB_ROW16)) ? 2 : 4;
        const bool row16 = (hdr[7] & URHGPU_BLOB_ROW16) != 0;
        const int64_t head = (hdr[12] + (hdr[3] + 7) / 8 + 15) & ~int64_t(15);
        if (n_rows < 0 || n_rows > st->cap_rows || n_pos < 0 || n_pos > st->cap_pos || head < URHGPU_BLOB_HEADER_BYTES || head > SL.head_cap) return URHGPU_ERR_ARG;
        bool more = false;
        auto fetch = [&](int64_t off, int64_t bytes) -> int {
            URH_HIP(hipMemcpyAsync(s.h_blob + off, s.stage + off, (size_t)bytes, hipMemcpyDeviceToHost, st->copy_stream));
            more = true;
            return URHGPU_OK;
        };
        if (head > s.copied) URH_TRY(fetch(s.copied, head - s.copied));
        if (n_rows > s.copied_rows) {
            if (!row16) URH_TRY(fetch(SL.off_row_state + s.copied_rows, n_rows - s.copied_rows));
            URH_TRY(fetch(SL.off_row_len + len_bytes * s.copied_rows, len_bytes * (n_rows - s.copied_rows)));
        }
        if (n_pos > s.copied_pos) URH_TRY(fetch(SL.off_pos32 + 4 * s.copied_pos, 4 * (n_pos - s.copied_pos)));
        if (more) { URH_HIP(hipStreamSynchronize(st->copy_stream)); st->short_copies += 1; }
        st->predicted_rows = n_rows + n_rows / 8 + 4096;
        st->last_rows = n_rows;
        st->predicted_pos = n_pos + n_pos / 8 + 4096;
        fill_result(st, s, r);
        s.state = 3;
        return URHGPU_OK;
    }
    const int64_t total = hdr[6];
    if (total > s.copied) {                                 // the prediction was short (the first pass of a stream, a denser capture): the rest, now
        URH_HIP(hipMemcpyAsync(s.h_blob + s.copied, (const char *)s.out.blob + s.copied, (size_t)(total - s.copied), hipMemcpyDeviceToHost,
                               st->copy_stream));
        URH_HIP(hipStreamSynchronize(st->copy_stream));
        st->short_copies += 1;
    }
    st->predicted_bytes = total + total / 8 + 65536;
    fill_result(st, s, r);
    s.state = 3;
    return URHGPU_OK;
}

}  // namespace

extern "C" {

int urhgpu_stream_capacities(int64_t n_max, const urhgpu_params *p, int64_t *cap_rows, int64_t *cap_bits, int64_t *cap_msg, int64_t *cap_pos) {
    if (!p || n_max <= 0 || p->samples_per_symbol < 1 || p->bits_per_symbol < 1 || p->tolerance < 0) return URHGPU_ERR_ARG;
    const int64_t sps = (int64_t)p->samples_per_symbol;
    // accepted runs cannot be denser than one per (tolerance + 1) samples; the default assumes at most ~4 per symbol (a capture
    // that is mostly noise needs more: urhgpu_host_result::truncated says so and rows_needed by how much)
    const int64_t rows = std::min<int64_t>(n_max / ((int64_t)p->tolerance + 1) + 2, std::max<int64_t>(4096, 4 * (n_max / sps) + 4096));
    const int64_t bits = (n_max / sps + 2 * rows / 8 + 64) * (int64_t)p->bits_per_symbol + rows;
    const int64_t msg = std::max<int64_t>(64, rows / 4);
    if (cap_rows) *cap_rows = rows;
    if (cap_bits) *cap_bits = bits;
    if (cap_msg) *cap_msg = msg;
    if (cap_pos) *cap_pos = bits + 2 * msg + 2;
    return URHGPU_OK;
}

int urhgpu_stream_create(urhgpu_ctx *ctx, int64_t n_max, const urhgpu_params *p, int want_qad, int want_pos, int64_t cap_rows, urhgpu_stream **out) {
    if (!ctx || !p || !out || n_max <= 0 || n_max > INT32_MAX) return URHGPU_ERR_ARG;      // int32 row lengths, uint32 positions
    if (p->mod == URHGPU_MOD_PSK) return URHGPU_ERR_UNSUPPORTED;                       // the Costas path synchronises with the host
    URH_HIP(hipSetDevice(ctx->device));
    urhgpu_stream *st = new (std::nothrow) urhgpu_stream();
    if (!st) return URHGPU_ERR_ARG;
    st->ctx = ctx; st->p = *p; st->want_qad = want_qad ? 1 : 0; st->want_pos = want_pos ? 1 : 0; st->n_max = n_max;
    st->p.write_bit_sample_pos = st->want_pos;
    int status = urhgpu_stream_capacities(n_max, p, &st->cap_rows, &st->cap_bits, &st->cap_msg, &st->cap_pos);
    if (status != URHGPU_OK) { delete st; return status; }
    if (cap_rows > 0) {                                    // the caller knows better (a retry after urhgpu_host_result::truncated)
        st->cap_rows = cap_rows;
        st->cap_bits = (n_max / (int64_t)p->samples_per_symbol + 2 * cap_rows / 8 + 64) * (int64_t)p->bits_per_symbol + cap_rows;
        st->cap_msg = std::max<int64_t>(64, cap_rows / 4);
        st->cap_pos = st->cap_bits + 2 * st->cap_msg + 2;
    }
    st->cap_blob = blob_capacity(st->cap_rows, st->cap_bits, st->cap_msg, st->cap_pos, st->want_pos);
    {
        // 16-bit row lengths for staged passes: their escape list lives in the staging blob's (otherwise unused) head region
        const StagedLayout SL = staged_layout(st->cap_rows, st->cap_bits, st->cap_msg, st->cap_pos, st->want_pos);
        st->len16 = SL.head_cap - URHGPU_BLOB_HEADER_BYTES >= 8 + (n_max / 65535 + 2) * 8 + 16;
        // URHGPU_BLOB_ROW16 (dense pulse tables): its escape list -- a row of 8191 samples and more -- gets a place of its own behind the split
        // layout's sections, in the staging blobs and in the host blobs
        st->esc_extra = ((8 + (n_max / 8191 + 2) * 8 + 16 + 255) & ~int64_t(255)) + std::max<int64_t>(0, SL.total - st->cap_blob);
        st->row16_ok = st->len16 && p->bits_per_symbol <= 2;                    // (state + 1 in three bits: orders 2 and 4)
    }
    if (p->mod == URHGPU_MOD_FSK && (p->dtype == URHGPU_DT_I8 || p->dtype == URHGPU_DT_I16)) {
        if (hipHostMalloc((void **)&st->h_probe, 64) != hipSuccess || hipEventCreateWithFlags(&st->ev_probe, hipEventDisableTiming) != hipSuccess) { delete st; return URHGPU_ERR_HIP; }
        memset(st->h_probe, 0, 64);
    }
    st->was_pipelined = ctx->pipelined;
    if (!ctx->pipelined) { status = urhgpu_ctx_set_pipelined(ctx, 1, nullptr); if (status != URHGPU_OK) { delete st; return status; } }
    status = urhgpu_ctx_reserve(ctx, n_max, p->tolerance);
    if (status != URHGPU_OK) { urhgpu_stream_destroy(st); return status; }
    if (hipStreamCreateWithFlags(&st->copy_stream, hipStreamNonBlocking) != hipSuccess) { urhgpu_stream_destroy(st); return URHGPU_ERR_HIP; }
    const size_t b_qad = st->want_qad ? a256((size_t)n_max * 4) : 0, b_rows = a256((size_t)st->cap_rows * 16), b_bits = a256((size_t)st->cap_bits),
                 b_off = a256((size_t)(st->cap_msg + 1) * 8), b_pos = st->want_pos ? a256((size_t)st->cap_pos * 8) : 0, b_blob = a256((size_t)(st->cap_blob + st->esc_extra));
    if (st->want_qad)
        for (auto &q : st->qad_ring)
            if (hipMalloc((void **)&q, b_qad) != hipSuccess) { urhgpu_stream_destroy(st); return URHGPU_ERR_HIP; }
    for (auto &s : st->slot) {
        memset(&s.out, 0, sizeof(s.out));
        if (hipMalloc(&s.dev, b_rows + b_bits + 3 * b_off + b_pos + 256 + 2 * b_blob) != hipSuccess ||
            hipHostMalloc((void **)&s.h_blob2[0], (size_t)(st->cap_blob + st->esc_extra)) != hipSuccess ||
            hipHostMalloc((void **)&s.h_blob2[1], (size_t)(st->cap_blob + st->esc_extra)) != hipSuccess || hipHostMalloc((void **)&s.h_counts, 64) != hipSuccess ||
            hipEventCreateWithFlags(&s.ev_tail, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s.ev_copy, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s.ev_rows, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s.ev_shipped, hipEventDisableTiming) != hipSuccess) {
            urhgpu_stream_destroy(st);
            return URHGPU_ERR_HIP;
        }
        char *d = (char *)s.dev;
        s.out.rows = (int64_t *)d; d += b_rows; s.out.cap_rows = st->cap_rows;
        s.out.bits = (uint8_t *)d; d += b_bits; s.out.cap_bits = st->cap_bits;
        s.out.msg_off = (int64_t *)d; d += b_off;
        s.out.pauses = (int64_t *)d; d += b_off; s.out.cap_msg = st->cap_msg;
        s.out.pos_off = (int64_t *)d; d += b_off;
        if (st->want_pos) { s.out.pos = (int64_t *)d; d += b_pos; s.out.cap_pos = st->cap_pos; }
        s.out.counts = (int64_t *)d; d += 256;
        s.out.blob = d; s.out.cap_blob = st->cap_blob; d += b_blob;
        s.stage = d;
        memset(s.h_counts, 0, 64);
    }
    *out = st;
    return URHGPU_OK;
}

int urhgpu_stream_destroy(urhgpu_stream *st) {
    if (!st) return URHGPU_OK;
    (void)hipSetDevice(st->ctx->device);
    (void)urhgpu_ctx_sync(st->ctx);
    if (st->copy_stream) { (void)hipStreamSynchronize(st->copy_stream); (void)hipStreamDestroy(st->copy_stream); }
    for (auto &s : st->slot) {
        if (s.dev) (void)hipFree(s.dev);
        s.dev = nullptr;
        if (s.h_blob2[0]) (void)hipHostFree(s.h_blob2[0]);
        if (s.h_blob2[1]) (void)hipHostFree(s.h_blob2[1]);
        if (s.h_counts) (void)hipHostFree(s.h_counts);
        if (s.ev_tail) (void)hipEventDestroy(s.ev_tail);
        if (s.ev_copy) (void)hipEventDestroy(s.ev_copy);
        if (s.ev_rows) (void)hipEventDestroy(s.ev_rows);
        if (s.ev_shipped) (void)hipEventDestroy(s.ev_shipped);
    }
    for (auto &q : st->qad_ring) if (q) (void)hipFree(q);
    if (st->h_probe) (void)hipHostFree(st->h_probe);
    if (st->ev_probe) (void)hipEventDestroy(st->ev_probe);
    if (!st->was_pipelined) (void)urhgpu_ctx_set_pipelined(st->ctx, 0, nullptr);
    delete st;
    return URHGPU_OK;
}

static int stream_push(urhgpu_stream *st, const void *h_iq, const void *d_iq, int64_t n, urhgpu_host_result *ready);

int urhgpu_stream_push(urhgpu_stream *st, const void *d_iq, int64_t n, urhgpu_host_result *ready) {
    return stream_push(st, nullptr, d_iq, n, ready);
}

int urhgpu_stream_push_upload(urhgpu_stream *st, const void *h_iq, void *d_iq, int64_t n, urhgpu_host_result *ready) {
    if (!h_iq) return URHGPU_ERR_ARG;
    return stream_push(st, h_iq, d_iq, n, ready);
}

static int stream_push(urhgpu_stream *st, const void *h_iq, const void *d_iq, int64_t n, urhgpu_host_result *ready) {
    if (!st || !d_iq || n <= 2 || n > st->n_max) return URHGPU_ERR_ARG;
    urhgpu_ctx *ctx = st->ctx;
    URH_HIP(hipSetDevice(ctx->device));
    const int64_t i = st->seq;
    urhgpu_stream::Slot &s = st->slot[i % 3];
    if (ready) { memset(ready, 0, sizeof(*ready)); ready->seq = -1; }
    // the slot's previous pass (i - 3): its copy was queued with it; hand the result out now (valid until push i + 3 queues the copy
    // that reuses its host buffer)
    if (s.state == 1 || s.state == 2) {
        urhgpu_host_result r;
        URH_TRY(finish_copy(st, s, &r));
        if (ready) *ready = r;
    }
    // ...and nothing of this pass may be written into the slot's device buffers before that copy has read them
    if (s.state == 3) {
        URH_HIP(hipStreamWaitEvent(ctx->tail_stream, s.ev_copy, 0));
        if (s.staged) URH_HIP(hipStreamWaitEvent(ctx->tail_stream, s.ev_shipped, 0));     // (the copies out of the staging blob)
    }
    s.state = 0;
    s.h_blob = s.h_blob2[(i / 3) & 1];
    s.qad = st->want_qad ? st->qad_ring[i & 3] : nullptr;
    urhgpu_outputs pass_out = s.out;
    pass_out.qad = s.qad;
    pass_out.blob = nullptr; pass_out.cap_blob = 0;        // packed later, on the copy stream (queue_copy) -- or, streamed, by the segments
    pass_out.h_counts = s.h_counts;                        // the counts also land in pinned host memory (a store of the kernel that finalises them)
    // Streamed pass (pulse_table.hip "Segments"): the tail runs in segments beside the hot kernel and every segment stores its share of
    // the compact blob straight into the pinned host blob -- no pack launch at the end, no copy engine, no predicted size.  Captures the
    // bit-plane kernel does not take, or too short to cut, go the ordinary way: tail behind the hot kernel, pack + copy behind the tail.
    // Staged pass (the default for back-to-back passes since round 6): the same one-segment tail, but its kernels store the compact sections
    // into the slot's staging blob in HBM (split layout) and the COPY ENGINE ships them, sized by prediction like queue_copy: the row
    // sections as soon as the row kernel is through (while the bits are expanded), the head (+ positions) behind the pass's last kernel.
    bool streamed = false, staged = false;
    s.staged = false;
    // 16-bit row lengths (3 bytes per row); a DENSE pulse table -- more than one row per 64 samples in the last result -- ships state and length
    // in one uint16 (2 bytes per row: what a step ships over PCIe bounds such captures)
#ifdef URH_NO_ROW16
    const int len_mode = st->len16 ? 1 : 0;
#else
    const int len_mode = !st->len16 ? 0 : ((st->row16_ok && st->last_rows > n / 64) ? 2 : 1);
#endif
    if (st->h_probe) {
        const int permille = ((volatile int32_t *)st->h_probe)[0], pairs = ((volatile int32_t *)st->h_probe)[1];
#ifndef URH_NO_INT_PROBE          // (A/B builds: the default integer instantiation whatever the probe says)
        if (pairs >= 256) st->wide_int = st->wide_int ? (permille > 3) : (permille >= 10);
#endif
        ctx->wide_int_next = st->wide_int;
        if (st->wide_int) st->wide_passes += 1;
    }
    const int pass_status = urh::iq_to_bits_streamed(ctx, d_iq, n, &st->p, &pass_out, s.h_blob, st->cap_blob, s.ev_copy, &streamed, h_iq, s.stage, &staged, s.ev_rows, len_mode);
    ctx->wide_int_next = 0;
    URH_TRY(pass_status);
    if (st->h_probe && streamed) {     // behind the pass's tail: the capture is complete there whichever way it arrived
        const float nt = st->p.noise_threshold;
        URH_TRY(launch_wide_probe(d_iq, st->p.dtype, n, nt * nt, st->h_probe, ctx->tail_stream));
        URH_HIP(hipEventRecord(st->ev_probe, ctx->tail_stream));
        st->probe_pending = true;
    }
    if (streamed && staged) {
        const StagedLayout SL = staged_layout(st->cap_rows, st->cap_bits, st->cap_msg, st->cap_pos, st->want_pos);
        const int64_t rows = std::min<int64_t>(st->predicted_rows, st->cap_rows), npos = st->want_pos ? std::min<int64_t>(st->predicted_pos, st->cap_pos) : 0;
        const bool skip_copy = (urh::g_tail_skip & 512) != 0;                  // (measurement hook: urhgpu_test_tail_skip)
        if (rows > 0 && !skip_copy) {
            // (behind the row kernel, not behind the pass's last kernel: measured the same to slightly better, 0.2839-0.2847 against 0.2848-0.2880 ms per step at K = 20)
            URH_HIP(hipStreamWaitEvent(st->copy_stream, s.ev_rows, 0));
            if (len_mode != 2) URH_HIP(hipMemcpyAsync(s.h_blob + SL.off_row_state, s.stage + SL.off_row_state, (size_t)rows, hipMemcpyDeviceToHost, st->copy_stream));
            URH_HIP(hipMemcpyAsync(s.h_blob + SL.off_row_len, s.stage + SL.off_row_len, (size_t)rows * (len_mode ? 2 : 4), hipMemcpyDeviceToHost, st->copy_stream));
        }
        // (recorded behind the pass's last kernel, which has stored the head -- header, pauses, offsets, packed bits: small -- into the host
        // blob itself: no copy of it, no hop to another stream at the end of the chain)
        // The host waits for BOTH ends (finish_copy): s.ev_copy on the tail stream, s.ev_shipped on the copy stream -- no hop from one stream
        // to the other at the end of the chain, unless the pass ships positions (they are complete with the pass's last kernel).
        if (npos > 0 && !skip_copy) {
            URH_HIP(hipStreamWaitEvent(st->copy_stream, s.ev_copy, 0));
            URH_HIP(hipMemcpyAsync(s.h_blob + SL.off_pos32, s.stage + SL.off_pos32, (size_t)npos * 4, hipMemcpyDeviceToHost, st->copy_stream));
        }
        URH_HIP(hipEventRecord(s.ev_shipped, st->copy_stream));
        s.state = 2; s.seq = i; s.n = n; s.copied = SL.head_cap; s.copied_rows = rows; s.copied_pos = npos; s.staged = true;
        st->seq = i + 1;
        st->staged_passes += 1;
        return URHGPU_OK;
    }
    if (streamed) {
        s.state = 2; s.seq = i; s.n = n; s.copied = st->cap_blob;
        st->seq = i + 1;
        st->streamed_passes += 1;
        if (h_iq) st->uploaded_passes += 1;
        return URHGPU_OK;
    }
    if (h_iq) {
        // a capture the segmented path does not take (ASK, a partial tile at the end, too short to cut): one copy on the caller's stream, in
        // front of the ordinary pass
        const size_t bps = st->p.dtype == URHGPU_DT_F32 ? 8 : (st->p.dtype == URHGPU_DT_I16 || st->p.dtype == URHGPU_DT_U16) ? 4 : 2;
        URH_HIP(hipMemcpyAsync(const_cast<void *>(d_iq), h_iq, (size_t)n * bps, hipMemcpyHostToDevice, ctx->stream));
    }
    URH_TRY(urhgpu_iq_to_bits_dev(ctx, d_iq, n, &st->p, &pass_out));
    URH_HIP(hipEventRecord(s.ev_tail, ctx->tail_stream));
    s.state = 1; s.seq = i; s.n = n;
    st->seq = i + 1;
    URH_TRY(queue_copy(st, s));                            // pack + copy behind this pass's tail, on the copy stream; the host does not wait
    return URHGPU_OK;
}

int urhgpu_stream_wide_passes(urhgpu_stream *st, int64_t *n_passes) {
    if (!st || !n_passes) return URHGPU_ERR_ARG;
    *n_passes = st->wide_passes;
    return URHGPU_OK;
}

int urhgpu_stream_stats(urhgpu_stream *st, int64_t *out4) {
    if (!st || !out4) return URHGPU_ERR_ARG;
    out4[0] = st->seq; out4[1] = st->short_copies; out4[2] = st->predicted_bytes; out4[3] = st->cap_blob;
    // passes whose compact sections were written by the tail's own kernels (segments, direct, staged): their count, negated
    if (st->streamed_passes + st->staged_passes > 0) out4[2] = -(st->streamed_passes + st->staged_passes);
    return URHGPU_OK;
}

int urhgpu_stream_flush(urhgpu_stream *st, urhgpu_host_result *out3, int *n_out) {
    if (!st || !out3 || !n_out) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(st->ctx->device));
    *n_out = 0;
    // oldest first
    int order[3] = {0, 1, 2};
    std::sort(order, order + 3, [&](int a, int b) { return st->slot[a].seq < st->slot[b].seq; });
    for (int k = 0; k < 3; ++k) {
        urhgpu_stream::Slot &s = st->slot[order[k]];
        if (s.state == 1 || s.state == 2) {
            URH_TRY(finish_copy(st, s, &out3[*n_out]));
            *n_out += 1;
        }
    }
    // a streamed pass's blob is complete a moment before its hot kernel has retired (the last qad stores): d_qad of the results handed
    // out here is read by the caller next
    if (st->streamed_passes + st->staged_passes > 0 && st->ctx->tail_pending) URH_TRY(wait_event(st->ctx->ev_tail[(st->ctx->flip + 2) % 3]));
    if (st->probe_pending) { URH_TRY(wait_event(st->ev_probe)); st->probe_pending = false; }
    return URHGPU_OK;
}

}  // extern "C"
