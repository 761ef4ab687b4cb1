// Shared host-side plumbing for liburhgpu.so: context, scratch arena, error handling.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "../../include/urhgpu.h"

// Wavefronts of the tail kernels (pulse table, bits, compact blob) run beside the NEXT pass's hot kernel, seven of whose wavefronts
// share every SIMD with them: the tail's wavefronts ask for the highest issue priority (s_setprio 3), so that their few hundred
// instructions are not queued behind seven streaming wavefronts' -- the tail chain, not the hot kernel, set the period of
// pipelined passes (rocprofv3 timeline, round 3).  -DURH_TAIL_PRIO_LEVEL=0: A/B.
#ifndef URH_TAIL_PRIO_LEVEL
#define URH_TAIL_PRIO_LEVEL 3
#endif
// -DURH_TAIL_PAD=1 (A/B): every tail kernel also ALLOCATES 96 VGPRs, so that none of them fits beside seven hot wavefronts on a SIMD (80 of
// 512 registers free at most when one has retired): the tail then lives on the CUs the hot stream's mask leaves out and nowhere else.
// Measured (round 6): no difference -- the row and expansion kernels need more than 80 anyway, the small ones are too short to matter.
#ifndef URH_TAIL_PAD
#define URH_TAIL_PAD 0
#endif
#if URH_TAIL_PAD
#define URH_TAIL_PRIO() do { __builtin_amdgcn_s_setprio(URH_TAIL_PRIO_LEVEL); __asm__ volatile("v_mov_b32 v95, 0" ::: "v95"); } while (0)
#else
#define URH_TAIL_PRIO() __builtin_amdgcn_s_setprio(URH_TAIL_PRIO_LEVEL)
#endif

namespace urh {

extern thread_local char g_hip_err[256];

inline int hip_fail(hipError_t e, const char *what, const char *file, int line) {
    snprintf(g_hip_err, sizeof(g_hip_err), "%s: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    return URHGPU_ERR_HIP;
}

#define URH_HIP(call)                                                        \
    do {                                                                     \
        hipError_t _e = (call);                                              \
        if (_e != hipSuccess) return urh::hip_fail(_e, #call, __FILE__, __LINE__); \
    } while (0)

#define URH_TRY(call)                 \
    do {                              \
        int _s = (call);              \
        if (_s != URHGPU_OK) return _s; \
    } while (0)

// Device scratch that only ever grows; re-used by every call on the context.
struct Arena {
    void *base = nullptr;
    size_t cap = 0;
    size_t used = 0;
    int reserve(size_t bytes);   // ensure capacity (may hipMalloc; invalidates previous pointers)
    void reset() { used = 0; }
    // bump allocation, 256-byte aligned; returns nullptr when out of space (callers size first)
    void *take(size_t bytes) {
        size_t off = (used + 255) & ~size_t(255);
        if (off + bytes > cap) return nullptr;
        used = off + bytes;
        return (char *)base + off;
    }
    void release();
};

}  // namespace urh

constexpr size_t kSmallPinned = size_t(1) << 20;
struct urhgpu_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipDeviceProp_t prop;
    urh::Arena arena;        // per-call scratch (tables, slabs, scan partials)
    urh::Arena staging;      // device mirrors of host buffers for the host-pointer entry points
    urh::Arena aux;          // Costas candidate / checkpoint states (lives across the arena / staging users)
    urh::Arena fir_work;     // urhgpu_fir_filter_dev's padded taps + tile list (its own: the filter does not wait for a pipelined pass's tail)
    hipEvent_t ev_fir = nullptr;       // behind the last filter that used fir_work, on fir_stream
    hipStream_t fir_stream = nullptr;
    int64_t *d_counts = nullptr;   // small device result block (8 x int64)
    int64_t *h_counts = nullptr;   // pinned host mirror
    bool tune_spin_wait = true;    // wait_stream polls (see there)
    bool tune_wide_int = false;    // one-shot / sharded passes over signed integer FSK captures take the wide-loop instantiation (RunArgs::wide_int)
    bool tune_shard_summary_generic = false;   // urhgpu_shard_runs_dev: the local pass as launch_resolve (A/B, tests) instead of launch_shard_summary
    int wide_int_next = 0;         // the next streamed pass over a signed integer FSK capture takes the wide-loop instantiation (set by urhgpu_stream_*, RunArgs::wide_int)
    char *h_small = nullptr;       // pinned landing zone of the estimators' small results (kSmallPinned bytes): copies into it are truly asynchronous
    int32_t *d_tickets = nullptr;  // 8 zeroed ints: elections of the fused scan kernels (scan.hpp)
    void *d_desc = nullptr;        // descriptors of the single-pass scans: dedicated, zeroed when (re)allocated
    size_t desc_cap = 0;
    void *d_rdesc = nullptr;       // look-back descriptors of the tile tail's resolve scan (same regime)
    size_t rdesc_cap = 0;
    // pass counter carried by the descriptor flags; starts far above anything a count or a position can be, so that memory
    // that held other values (another descriptor layout) can never look like a flag of the current pass
    unsigned long long scan_epoch = 0x0ACE0FBA5E000000ull;
    // optional timing of the dominant kernel (demod + run segmentation) with HIP events on `stream`
    std::vector<hipEvent_t> prof_events;   // four per record: [4k], [4k+1] around the hot launch; [4k+2], [4k+3] attached to the dispatch
    std::vector<bool> prof_dispatch;       // record k: the dispatch-attached pair was used
    int prof_used = 0;                     // pairs recorded since urhgpu_ctx_profile_begin
    bool prof_on = false;
    bool prof_bracket = false;             // also bracket the hot launch with stream-level events (URH_PROFILE_BRACKET)
    void *shard = nullptr;                 // state of a sharded pass between its phases (capi.hip: ShardSession)
    // Pipelined mode (urhgpu_ctx_set_pipelined): the hot kernel of a pass runs on `stream`, everything after it on
    // `tail_stream`, with two scratch arenas used alternately, so that the hot kernel of the NEXT pass overlaps the
    // (latency-bound, nearly empty) tail of this one -- and, with three arenas, of the one before.  Outputs are complete after urhgpu_ctx_join / urhgpu_ctx_sync.
    bool pipelined = false;
    int hot_lds_pad_sharded = 33 * 1024;   // the same for the urhgpu_shard_* passes (their tail is longer: see urhgpu_ctx_set_tuning)
    int tune_hot_cus_removed = 4;          // CUs per XCD the hot kernel of a pipelined pass leaves alone (0: no mask); see urhgpu_ctx_set_pipelined
    hipStream_t hot_masked = nullptr;      // private CU-masked stream of the hot kernel (pipelined mode)
    hipEvent_t ev_in = nullptr;
    int hot_lds_pad = 0;           // pipelined mode: dynamic LDS bytes added to every hot-kernel workgroup (see RunArgs::lds_pad)
    hipStream_t tail_stream = nullptr;
    bool own_tail_stream = false;
    urh::Arena arena_alt, arena_alt2;   // three scratch arenas in rotation: the hot kernel of pass i + 2 does not wait for the tail of pass i
    hipEvent_t ev_hot = nullptr;
    hipEvent_t ev_tail[3] = {nullptr, nullptr, nullptr};
    int flip = 0;
    bool tail_pending = false;
    int tile_parity = 0;           // which of the two huge-row counters (d_tickets[8..9]) the current pass appends to
    // streamed passes (urhgpu_stream_*; capi.hip: iq_to_bits_streamed): per scratch arena 16 progress counters + one SegState
    void *d_seg = nullptr;         // 3 x kSegBlockBytes, zero between passes
    bool seg_dirty[3] = {false, false, false};   // a pass failed between its hot launch and its last segment: counters not trusted
    int tune_stream_segments = 7;  // rows segments of a streamed pass's tail (1: no streaming), urhgpu_ctx_set_tuning("stream_segments")
    int tune_stream_policy = 5;    // 5 (default): 3 for passes that ship no positions, 0 for those that do; 3: every pass DIRECT (one segment behind the hot
                                   // kernel, rows and packed results stored into the pinned host blob by the tail's kernels); 4: segments when idle, direct
                                   // otherwise; 0: stream a pass only when the pipeline is idle (nothing of an earlier pass still running: a single
                                   // capture, where the latency of the tail counts); 1: every qualifying pass; 2: never.  Beside the hot kernel
                                   // of a FOLLOWING pass the segments' short kernels are slower than one tail over the whole capture
                                   // (memory latency under a saturated HBM), so back-to-back passes keep the one-piece tail
    long long passes_begun = 0;    // pipelined passes started on this context
    bool tune_stream_latency = false;      // policy 5: a pass that finds the pipeline idle runs its tail in segments (lowest latency for ONE capture)
    bool tune_stream_pos_direct = true;    // direct passes ship positions themselves (group scan + expansion store uint32 into the host blob)
    int tune_upload_pieces = 4;             // pieces of urhgpu_stream_push_upload: pieces - 1 equal ones and a short last one (shape 2)
    hipEvent_t ev_piece[16] = {};           // the hot kernel of piece k has finished (the rows segment k waits for it: no polling gate in upload mode)
    hipEvent_t ev_hot_done[3] = {nullptr, nullptr, nullptr};   // behind the hot kernel of the pass in arena slot k (streamed passes)
    hipStream_t bits_stream = nullptr;   // second tail stream of streamed passes: the bits segments, behind the rows they expand
    hipEvent_t ev_rows[3][16] = {};      // [arena slot][bits segment]: the rows below the bits segment's end have been written
    hipEvent_t ev_bits[3] = {nullptr, nullptr, nullptr};   // the pass's last bits segment has been packed
};
constexpr size_t kSegBlockBytes = 4096;      // 16 progress counters on their own 128-byte lines, then the SegState

namespace urh {
// A pass whose tail runs in segments beside the hot kernel, every segment storing its share of the compact blob into pinned host memory
// (pulse_table.hip "Segments").  *streamed = false: the arguments do not qualify (nothing was launched; take the ordinary path).
// ev_ready (may be nullptr) is recorded on the tail stream behind the last segment's pack kernel: the host blob is complete.
// h_iq != nullptr: upload mode -- the capture is copied from h_iq (host, pinned for PCIe speed) into d_iq piece by piece and every piece
// demodulated as it lands (urhgpu_stream_push_upload).
int iq_to_bits_streamed(urhgpu_ctx *ctx, const void *d_iq, int64_t n, const urhgpu_params *p, const urhgpu_outputs *out, void *host_blob,
                        int64_t cap_host, hipEvent_t ev_ready, bool *streamed, const void *h_iq, void *stage_blob = nullptr, bool *staged = nullptr,
                        hipEvent_t ev_rows = nullptr, int len16 = 0);
}

namespace urh {
// hipStreamSynchronize for the short waits of the estimator calls: the runtime's blocking wait wakes the host tens of microseconds after
// the stream has drained; polling the stream does not (tuning key "spin_wait" 0 takes the blocking wait; after 5 ms of polling it is
// taken anyway)
hipError_t wait_stream(const urhgpu_ctx *ctx, hipStream_t s);
int join_tail(urhgpu_ctx *ctx);    // capi.hip: the caller's stream waits for the tail of the last pipelined pass
}
