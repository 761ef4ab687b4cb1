// capi.hip -- the extern "C" surface declared in include/urhgpu.h.
#include <hip/hip_runtime.h>
#include <chrono>

#include <algorithm>
#include <math.h>
#include <new>
#include <vector>

#include "common.hpp"
#include "compact.hpp"
#include "fdlibm_atan2f.h"
#include "glibc_sincosf.h"
#include "runs.hpp"

namespace urh {

thread_local char g_hip_err[256] = "";

int Arena::reserve(size_t bytes) {
    if (bytes <= cap) return URHGPU_OK;
    if (base) { URH_HIP(hipFree(base)); base = nullptr; cap = 0; }
    const size_t want = (bytes + (size_t(1) << 20)) & ~((size_t(1) << 20) - 1);
    URH_HIP(hipMalloc(&base, want));
    cap = want;
    used = 0;
    return URHGPU_OK;
}
void Arena::release() {
    if (base) { (void)hipFree(base); base = nullptr; cap = 0; used = 0; }
}

hipError_t wait_stream(const urhgpu_ctx *ctx, hipStream_t s) {
    if (ctx->tune_spin_wait) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int it = 0;; ++it) {
            const hipError_t q = hipStreamQuery(s);
            if (q == hipSuccess) return hipSuccess;
            if (q != hipErrorNotReady) return q;
            if ((it & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) break;
        }
    }
    return hipStreamSynchronize(s);
}
// pipelined mode: make the caller's stream wait for the tail of the last pass (no host blocking).  Every entry point that takes
// scratch from ctx->arena or launches on ctx->stream calls this first: on a pipelined context the arena is the one the last pass's
// tail may still be working in.
int join_tail(urhgpu_ctx *ctx) {
    if (ctx->tail_pending) {
        URH_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_tail[(ctx->flip + 2) % 3], 0));       // the pass recorded last
        ctx->tail_pending = false;
    }
    return URHGPU_OK;
}

}  // namespace urh

#include <stdlib.h>

#include "launchers.hpp"

using namespace urh;

#define URH_PROBE_WPB 4     // wavefronts per chunk of the complex64 FSK bit-plane kernel (demod_runs.hip: URH_WPB)

namespace {

struct Plan {                 // how a capture of n samples is cut into chunks
    int64_t n_chunks;
    int64_t chunk_len;
    int64_t slab_stride;
};

// test hook (urhgpu_test_force_generic_tail): 0 routes single-GPU non-ASK captures through the generic 8-launch tail as well
bool g_tile_tail = true;

// test hook (urhgpu_test_force_tiles_per_chunk): 0 = size-dependent choice below, 1..4 = that many tiles per chunk
int g_force_tiles_per_chunk = 0;

// Argument checks shared by every entry point that cuts a capture into chunks or turns rows into bits.  The reference
// takes `tolerance` as uint16 (OverflowError outside 0..65535, signal_functions.pyx:392) and divides by
// samples_per_symbol (ZeroDivisionError, ProtocolAnalyzer.py:353); here both are URHGPU_ERR_ARG before anything is launched.
int check_params(const urhgpu_params *p, bool need_sps) {
    if (p->tolerance < 0 || p->tolerance > 65535) return URHGPU_ERR_ARG;
    if (need_sps && (p->samples_per_symbol < 1 || p->bits_per_symbol < 1)) return URHGPU_ERR_ARG;
    return URHGPU_OK;
}

Plan make_plan(const urhgpu_ctx *ctx, int64_t n, int tol) {
    Plan pl;
    // whole tiles are grouped into chunks of tiles_per_chunk tiles (one workgroup each); a partial
    // tile at the end of the capture is one more chunk (see launch_runs_4 in demod_runs.hip)
    const int64_t full_tiles = n / kTile;
    const int64_t target = (int64_t)ctx->prop.multiProcessorCount * 16;      // chunks (four wavefronts each in the bit-plane kernel): ~2 rounds of the resident set
    // at most 4 tiles = 64 rows per chunk: the bit-plane kernel parks one row per lane (kBpMaxRows)
    int64_t tiles_per_chunk = std::min<int64_t>(4, std::max<int64_t>(1, (full_tiles + target - 1) / target));
    if (g_force_tiles_per_chunk >= 1 && g_force_tiles_per_chunk <= 4) tiles_per_chunk = g_force_tiles_per_chunk;
    pl.chunk_len = tiles_per_chunk * kTile;
    pl.n_chunks = (full_tiles * kTile + pl.chunk_len - 1) / pl.chunk_len + ((n % kTile) ? 1 : 0);
    pl.slab_stride = pl.chunk_len / ((int64_t)tol + 1) + 2;
    return pl;
}

size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

constexpr int kMaxWorld = 1024;   // ranks of a sharded capture (table entries reserved for their summaries)

size_t digitize_scratch_bytes(const Plan &pl, int64_t cap_rows, bool ask, bool bits) {
    size_t b = 0;
    b += align256((size_t)(pl.n_chunks + kMaxWorld) * sizeof(ChunkInfo));
    b += align256((size_t)pl.n_chunks * pl.slab_stride * 8);
    b += align256(resolve_scratch_bytes(pl.n_chunks + kMaxWorld)) + 2 * 256;
    b += align256(tile_tail_bytes(pl.n_chunks + kMaxWorld)) + 256;
    if (ask) b += align256((size_t)cap_rows * 16) + align256(merge_scratch_bytes(cap_rows));
    if (bits) b += align256(bits_scratch_bytes(cap_rows));
    b += 4096;
    return b;
}

int fill_thresholds(RunArgs &a, const urhgpu_params *p) {
    if (p->bits_per_symbol < 1 || p->bits_per_symbol > 7) return URHGPU_ERR_UNSUPPORTED;
    a.order = 1 << p->bits_per_symbol;
    urhgpu_get_center_thresholds(p->center, p->center_spacing, a.order, a.thr);
    return URHGPU_OK;
}

float noise_for(const urhgpu_params *p) {
    switch (p->mod) {
        case URHGPU_MOD_ASK: return 0.0f;
        case URHGPU_MOD_FSK:
        case URHGPU_MOD_PSK: return -4.0f;
        default: return p->noise_other;
    }
}

// reference: signal_functions.pyx:343-354 (double sqrt of the integer constant, stored to float)
int max_magnitude_for(int dtype, float *out) {
    switch (dtype) {
        case URHGPU_DT_I8: *out = (float)sqrt(32513.0); return URHGPU_OK;
        case URHGPU_DT_U8: *out = (float)sqrt(65025.0); return URHGPU_OK;
        case URHGPU_DT_I16: *out = (float)sqrt(2147418113.0); return URHGPU_OK;
        case URHGPU_DT_U16: *out = (float)sqrt(4294836225.0); return URHGPU_OK;
        case URHGPU_DT_F32: *out = (float)sqrt(2.0); return URHGPU_OK;
        default: return URHGPU_ERR_DTYPE;
    }
}

int dtype_bytes(int dtype) {
    switch (dtype) {
        case URHGPU_DT_I8: case URHGPU_DT_U8: return 2;
        case URHGPU_DT_I16: case URHGPU_DT_U16: return 4;
        case URHGPU_DT_F32: return 8;
        default: return 0;
    }
}

// Core of grab_pulse_lens / the fused path: run-segmentation kernel (IQ or qad source), resolve,
// emit rows, optional ASK merge.  On return d_rows / d_n_rows hold the final pulse table.
// scratch must come from ctx->arena (already reserved by the caller).
// One profile record = four events: [4k], [4k+1] bracket the hot launch on the stream (what is reported for launches made of
// several kernels); [4k+2], [4k+3] are attached to the bit-plane kernel's dispatch (its own begin / end timestamps)
bool prof_begin_record(urhgpu_ctx *ctx, hipStream_t s) {
    const bool prof = ctx->prof_on && (size_t)(4 * ctx->prof_used + 3) < ctx->prof_events.size();
    if (!prof) return false;
    // the stream-level bracket costs two more barrier packets around the hot launch (about 10 us of bubbles per pass): only
    // on request (URH_PROFILE_BRACKET, comparison of the two timings); the dispatch-attached pair costs nothing extra
    if (ctx->prof_bracket && hipEventRecord(ctx->prof_events[4 * ctx->prof_used], s) != hipSuccess) return false;
    g_hot_events.start = ctx->prof_events[4 * ctx->prof_used + 2];
    g_hot_events.stop = ctx->prof_events[4 * ctx->prof_used + 3];
    g_hot_events.used = false;
    return true;
}
int prof_end_record(urhgpu_ctx *ctx, hipStream_t s) {
    const bool used = g_hot_events.used;
    g_hot_events = HotEvents();
    if (ctx->prof_bracket) URH_HIP(hipEventRecord(ctx->prof_events[4 * ctx->prof_used + 1], s));
    else if (!used) return URHGPU_OK;             // a launch made of several kernels (state-byte path): no record without the bracket
    if ((size_t)ctx->prof_used >= ctx->prof_dispatch.size()) ctx->prof_dispatch.resize((size_t)ctx->prof_used + 1);
    ctx->prof_dispatch[(size_t)ctx->prof_used] = used;
    ctx->prof_used += 1;
    return URHGPU_OK;
}

// Which captures take the CU-masked hot stream in pipelined mode: all of them.  Until round 5 integer captures kept the caller's stream --
// their kernel ALONE loses 2-5 % on 224 CUs (it is VALU-bound: profiles/r03a_mask_policy_probe.txt) --, but beside a hot kernel that
// fills all 256 CUs the previous pass's tail finds no wave slots and the passes serialise: pipelined steps through the capture stream
// 0.348 -> 0.280 ms (int16) and 0.361 -> 0.276 ms (int8) with the mask, complex64 unchanged (profiles/r05_dtype_stream_ab.txt).
static inline bool masked_hot_stream(const urhgpu_params *p) { (void)p; return true; }

// pipelined passes: the stream the hot kernel is launched on -- the CU-masked private one (see urhgpu_ctx_set_pipelined), ordered
// behind what the caller has queued on the context's stream so far
int hot_stream_begin(urhgpu_ctx *ctx, hipStream_t *out) {
    *out = ctx->stream;
    if (!ctx->pipelined || !ctx->hot_masked) return URHGPU_OK;
    hipStream_t hot = ctx->hot_masked;
    // The masked stream has default flags: what the caller has queued on the NULL stream is ordered before its work by the runtime
    // itself (and costs nothing when the NULL stream is idle).  An explicit event on the NULL stream would make THAT stream wait for the
    // previous hot kernel first and hand over afterwards: two cross-queue hand-overs between consecutive hot kernels (measured: a
    // 50 us gap instead of 5).  Any other stream of the caller's hands over through an event.
    if (ctx->stream != nullptr) {
        URH_HIP(hipEventRecord(ctx->ev_in, ctx->stream));
        URH_HIP(hipStreamWaitEvent(hot, ctx->ev_in, 0));
    }
    *out = hot;
    return URHGPU_OK;
}
// scratch (from the arena) and persistent descriptors of the tile tail over a table of n_entries chunks
int tile_tail_mem(urhgpu_ctx *ctx, int64_t n_entries, bool expands_bits, TileTailMem *tm) {
    tm->mem = ctx->arena.take(tile_tail_bytes(n_entries));
    tm->n_chunks = n_entries; tm->huge_count = ctx->d_tickets + 8;
    tm->parity = expands_bits ? (ctx->tile_parity ^= 1) : ctx->tile_parity;   // only passes that expand bits consume a counter
    tm->d_row_base = nullptr;
    if (!tm->mem) return URHGPU_ERR_ARG;
    const size_t rd = tile_rdesc_bytes(n_entries);
    if (rd > ctx->rdesc_cap) {
        if (ctx->d_rdesc) { URH_HIP(hipFree(ctx->d_rdesc)); ctx->d_rdesc = nullptr; ctx->rdesc_cap = 0; }
        const size_t want = (rd + 65535) & ~size_t(65535);
        URH_HIP(hipMalloc(&ctx->d_rdesc, want));
        URH_HIP(hipMemset(ctx->d_rdesc, 0, want));
        // (the memset is work of the NULL stream: it runs behind everything queued on the blocking streams -- a hot kernel that waits for
        // an upload --, and the tail's non-blocking streams do not wait for it: descriptors published by the pass's first kernels were
        // wiped by it.  Allocation time only: wait until it has happened.)
        URH_HIP(hipDeviceSynchronize());
        ctx->rdesc_cap = want;
    }
    tm->rdesc = ctx->d_rdesc; tm->epoch = ++ctx->scan_epoch;
    return URHGPU_OK;
}

int digitize(urhgpu_ctx *ctx, bool from_iq, const void *d_in, int64_t n, const urhgpu_params *p, float *d_qad,
             int64_t *d_rows, int64_t cap_rows, int64_t *d_n_rows, int64_t *d_n_rows_needed, int64_t *d_n_acc,
             const Plan &pl, int seg_mode = 0, hipStream_t s_tail = nullptr, const BitsParams *tile_bp = nullptr,
             TileTailMem *tile_out = nullptr) {
    hipStream_t s = ctx->stream;
    // the CU-masked hot stream pays for float32 / complex64 captures, whose kernel is bound by the HBM (1-3 % per pipelined step, 5 % for
    // the kernel on its own); integer captures and wide FSK deviations are VALU-bound and LOSE 2-5 % on 224 CUs (tools/mask_policy_probe.py)
    if (s_tail && from_iq && masked_hot_stream(p)) URH_TRY(hot_stream_begin(ctx, &s));
    if (tile_out) tile_out->mem = nullptr;
    RunArgs a;
    memset(&a, 0, sizeof(a));
    URH_TRY(fill_thresholds(a, p));
    a.in = d_in; a.qad = d_qad; a.left_halo = nullptr; a.n = n; a.pos_base = 0;
    a.chunk_len = pl.chunk_len; a.slab_stride = pl.slab_stride;
    a.noise_sqrd = p->noise_threshold * p->noise_threshold;
    a.noise_val = noise_for(p);
    a.tol = p->tolerance;
    a.lds_pad = ctx->pipelined ? ctx->hot_lds_pad : 0;
    a.wide_int = ctx->tune_wide_int ? 1 : 0;                 // (a one-shot pass has no probe of its capture to go by: the caller's word)
    if (from_iq) URH_TRY(max_magnitude_for(p->dtype, &a.max_magnitude));
    if (seg_mode) {
        // message segmentation: state = (|sample| > noise threshold) with the 10-sample outlier tolerance.  Reuses the
        // ASK arithmetic with max_magnitude 1 (q = sqrtf(I*I + Q*Q) exactly), no noise gating, threshold = noise level.
        a.seg_mode = 1; a.max_magnitude = 1.0f; a.noise_sqrd = -1.0f; a.noise_val = __builtin_nanf("");
        if (d_qad) {
            // the pass also leaves afp_demod(iq, noise_threshold, "ASK") in d_qad (float32 captures; p->center is the noise threshold here)
            if (!from_iq || p->dtype != URHGPU_DT_F32) return URHGPU_ERR_UNSUPPORTED;
            a.dm_noise_sqrd = p->center * p->center; a.dm_noise_val = 0.0f;
            URH_TRY(max_magnitude_for(p->dtype, &a.dm_max_magnitude));
        }
    }
    ChunkInfo *chunks = (ChunkInfo *)ctx->arena.take((size_t)pl.n_chunks * sizeof(ChunkInfo));
    uint64_t *slab = (uint64_t *)ctx->arena.take((size_t)pl.n_chunks * pl.slab_stride * 8);
    if (!chunks || !slab) return URHGPU_ERR_ARG;
    a.chunks = chunks; a.slab = slab;
    const bool prof = prof_begin_record(ctx, s);
    // pipelined: the tail stream waits for the completion signal of the hot dispatch itself where one launch covers the capture (no
    // partial tile at the end) -- an event recorded behind it is one more barrier packet between two hot kernels
    hipEvent_t hot_done = nullptr;
    if (s_tail && from_iq && n % kTile == 0) {
        if (!prof) { g_hot_events = HotEvents(); g_hot_events.stop = ctx->ev_hot; }
        hot_done = g_hot_events.stop;
    }
    {
        const int st = from_iq ? launch_demod_runs_iq(a, p->dtype, p->mod, d_qad != nullptr, s) : launch_runs_qad(a, s);
        if (st != URHGPU_OK) { g_hot_events = HotEvents(); return st; }
    }
    if (hot_done && !g_hot_events.used) hot_done = nullptr;          // the launch did not take the events (state-byte kernel)
    if (prof) URH_TRY(prof_end_record(ctx, s));
    else g_hot_events = HotEvents();
    if (s_tail) {                                   // pipelined: everything after the hot kernel goes to the tail stream
        if (!hot_done) { URH_HIP(hipEventRecord(ctx->ev_hot, s)); hot_done = ctx->ev_hot; }
        URH_HIP(hipStreamWaitEvent(s_tail, hot_done, 0));
        // the hot kernel ran on the private masked stream: what the caller queues on ITS stream afterwards (overwriting the capture, the
        // allocator handing its memory out again) must come behind it.  (The NULL stream synchronises with the masked stream by itself.)
        if (s != ctx->stream && ctx->stream != nullptr) URH_HIP(hipStreamWaitEvent(ctx->stream, hot_done, 0));
        s = s_tail;
    }

    const bool ask = (p->mod == URHGPU_MOD_ASK) && !seg_mode;
    int64_t *rows_stage = d_rows;
    int64_t *d_n_stage = d_n_rows;
    void *merge_scratch = nullptr;
    if (ask) {
        rows_stage = (int64_t *)ctx->arena.take((size_t)cap_rows * 16);
        merge_scratch = ctx->arena.take(merge_scratch_bytes(cap_rows));
        d_n_stage = (int64_t *)ctx->arena.take(64);
        if (!rows_stage || !merge_scratch || !d_n_stage) return URHGPU_ERR_ARG;
    }
    void *rs_mem = ctx->arena.take(resolve_scratch_bytes(pl.n_chunks));
    ResolveAux *aux = (ResolveAux *)(ctx->d_tickets + 4);
    if (!rs_mem) return URHGPU_ERR_ARG;
    const ResolveScratch rsc = resolve_scratch_carve(rs_mem, pl.n_chunks);
    ResolveArgs r;
    memset(&r, 0, sizeof(r));
    r.sc = rsc;
    r.chunks = chunks; r.n_chunks = pl.n_chunks; r.n_total = n; r.tol = p->tolerance;
    r.rows = rows_stage; r.cap_rows = cap_rows; r.d_n_acc = d_n_acc; r.d_n_rows = d_n_stage;
    r.d_n_rows_needed = d_n_rows_needed; r.write_last_row = 1;
    r.local_pass = 0; r.aux = aux; r.summary_out = nullptr; r.chunk_first = 0; r.n_local = pl.n_chunks; r.d_ts_carry = nullptr;
    EmitArgs e;
    e.sc = rsc;
    e.chunks = chunks; e.chunk_first = 0; e.slab = slab; e.slab_stride = pl.slab_stride;
    e.rows = rows_stage; e.cap_rows = cap_rows; e.d_ts_carry = nullptr; e.is_ask = ask ? 1 : 0; e.sps = p->samples_per_symbol;
    if (!ask && g_tile_tail) {
        // tile tail: one composed scan instead of three, rows + their bit aggregates in one pass (pulse_table.hip)
        TileTailMem tm;
        URH_TRY(tile_tail_mem(ctx, pl.n_chunks, tile_out != nullptr, &tm));
        URH_TRY(launch_tile_rows(r, e, tm, tile_out ? tile_bp : nullptr, s));
        if (tile_out) *tile_out = tm;
        URH_HIP(hipGetLastError());
        return URHGPU_OK;
    }
    URH_TRY(launch_resolve_emit_single(r, e, s));
    if (ask) URH_TRY(launch_merge_rows_ask(rows_stage, d_n_stage, cap_rows, d_rows, cap_rows, d_n_rows, merge_scratch, ctx->d_tickets, s));
    URH_HIP(hipGetLastError());
    return URHGPU_OK;
}

BitsParams bits_params(const urhgpu_params *p) {
    BitsParams bp;
    bp.sps = (int64_t)p->samples_per_symbol;
    bp.bps = p->bits_per_symbol;
    bp.pause_threshold = p->pause_threshold;
    bp.samples_per_bit = (int64_t)((double)p->samples_per_symbol / (double)p->bits_per_symbol);   // int(sps / bps) :344
    bp.write_pos = p->write_bit_sample_pos ? 1 : 0;
    bp.d_row_base = nullptr; bp.d_ts_carry = nullptr; bp.d_absorbed = nullptr; bp.d_extra = nullptr; bp.is_last_rank = 1;
    bp.d_rows_needed = nullptr;
    return bp;
}

// State of one sharded pass (urhgpu_shard_*): lives in the context between the phases; every pointer is
// carved from ctx->arena, which is not reset until the next pass begins.
struct ShardSession {
    int phase = 0;                 // -1: hot launch without the first chunk, -2: whole hot launch; 1: runs done, 2: rows done, 3: bits prepared
    bool piped = false;            // pipelined mode: phases after the hot kernel run on ctx->tail_stream
    RunArgs run;                   // kernel arguments of the hot launch (kept for the deferred first chunk)
    int rank = 0, world = 1;
    int64_t n_local = 0, pos_base = 0, n_total = 0;
    urhgpu_params p;
    urhgpu_outputs out;
    Plan pl;
    ChunkInfo *table = nullptr;    // [world - 1 summaries interleaved | local chunks], see shard_rows
    uint64_t *slab = nullptr;
    ResolveAux *aux = nullptr;
    void *rs_mem = nullptr;
    int64_t *rows_stage = nullptr; void *merge_scratch = nullptr; int64_t *d_n_stage = nullptr;
    void *bits_scratch = nullptr;
    int64_t *d_small = nullptr;    // [0] ts_carry, [1] absorbed, [2] extra (2 x int32), [3] n_rows (final), [4] row_base (tile tail)
    const int64_t *d_row_base = nullptr;
    bool use_tile = false;         // everything but ASK: the tile tail over the table (pulse_table.hip), as on a single GPU
    TileTailMem tile;
};

// descriptor memory of the single-pass scans for pulse tables of up to cap_rows rows
int scan_state(urhgpu_ctx *ctx, int64_t cap_rows, ScanState *out) {
    const size_t need = bits_desc_bytes(cap_rows);
    if (need > ctx->desc_cap) {
        if (ctx->d_desc) { URH_HIP(hipFree(ctx->d_desc)); ctx->d_desc = nullptr; ctx->desc_cap = 0; }
        const size_t want = (need + (size_t(1) << 20)) & ~((size_t(1) << 20) - 1);
        URH_HIP(hipMalloc(&ctx->d_desc, want));
        URH_HIP(hipMemset(ctx->d_desc, 0, want));
        URH_HIP(hipDeviceSynchronize());                   // (see tile_tail_mem: the NULL stream's memset must not land behind the pass's kernels)
        ctx->desc_cap = want;
    }
    out->tickets = ctx->d_tickets; out->desc = ctx->d_desc; out->desc_bytes = ctx->desc_cap; out->epoch = &ctx->scan_epoch;
    return URHGPU_OK;
}

using urh::join_tail;

// pipelined mode: rotate to the scratch arena used three passes ago; the caller's stream first waits for the tail that used it.
// (Two arenas made the hot kernel of pass i + 2 wait for the tail of pass i -- whose row kernel, starved of wave slots by the hot kernel
// of pass i + 1, only finishes right after it: the passes ran back to back again.  With three the hot kernels follow each other and
// the tails trail one pass behind.)
int begin_pipelined_pass(urhgpu_ctx *ctx) {
    std::swap(ctx->arena, ctx->arena_alt);
    std::swap(ctx->arena_alt, ctx->arena_alt2);
    ctx->passes_begun += 1;
    // Has the tail that last used this arena finished?  A caller that runs more than two passes ahead of the GPU (a tight loop of
    // passes) is held back HERE, on the host, until it has (bounded run-ahead; the GPU still has the previous hot kernel queued
    // behind the running one): a stream-level wait would put one more barrier packet between two hot kernels (about 4 us of the
    // gap; measured in round 2, tools/ab.sh history in profiles/HISTORY.md).
    const hipError_t q = hipEventQuery(ctx->ev_tail[ctx->flip]);
    if (q == hipErrorNotReady) {
        (void)hipGetLastError();                   // "not ready" is an answer, not an error: keep it out of the sticky last-error slot
        URH_HIP(hipEventSynchronize(ctx->ev_tail[ctx->flip]));
    } else if (q != hipSuccess) {
        URH_HIP(q);
    }
    return URHGPU_OK;
}
int end_pipelined_pass(urhgpu_ctx *ctx) {
    URH_HIP(hipEventRecord(ctx->ev_tail[ctx->flip], ctx->tail_stream));
    ctx->flip = (ctx->flip + 1) % 3;
    ctx->tail_pending = true;
    return URHGPU_OK;
}

// CU mask of the hot stream on a 256-CU part: `removed` CUs of every XCD left out (see urhgpu_ctx_set_pipelined)
void hot_cu_mask(int removed, uint32_t mask[8]) {
    for (int w = 0; w < 8; ++w) mask[w] = 0;
    for (int i = 0; i < 256; ++i) {
        const int c = ((i % 8) - (i / 32) + 8) % 8, k = (i / 8) % 4;
        if (c * 4 + k >= removed) mask[i / 32] |= 1u << (i % 32);
    }
}

ShardSession *session(urhgpu_ctx *ctx) {
    if (!ctx->shard) ctx->shard = new (std::nothrow) ShardSession();
    return (ShardSession *)ctx->shard;
}

}  // namespace

namespace urh {

// Segment boundaries (in chunks) of a streamed pass: S segments on kSegAlign chunks, the last one takes the remainder.
static int segment_bounds(int64_t n_chunks, int wanted, int shape, int last_units, int64_t *bound /*[kMaxSegments + 1]*/) {
    const int64_t blocks = n_chunks / kSegAlign;              // whole alignment units; what is left over belongs to the last segment
    int S = wanted;
    if (S > kMaxSegments) S = kMaxSegments;
    if (S > blocks) S = (int)blocks;
    if (S < 1) S = 1;
    bound[0] = 0;
    if (shape == 1 && S > 2) {
        // halving: 1/2, 1/4, ... of the capture; the last two segments are equal.  The tail of a long first segment runs beside the
        // hot kernel anyway; what is exposed at the end of the capture is the tail of the LAST segment only.
        int64_t left = blocks, at = 0;
        for (int k = 0; k < S - 1; ++k) {
            int64_t take = (k < S - 2) ? left / 2 : left / 2;
            const int64_t must_leave = S - 1 - k;             // at least one unit per remaining segment
            if (take < 1) take = 1;
            if (left - take < must_leave) take = left - must_leave;
            at += take; left -= take;
            bound[k + 1] = at * kSegAlign;
        }
    } else if (shape == 2 && S > 1) {
        // S - 1 equal segments and a SHORT last one (last_units alignment units + the remainder): the segments before it run beside the
        // hot kernel; what is exposed behind the hot kernel's end is the last segment's chain of six small kernels, whose length hardly
        // depends on the segment's size (5 - 8 us each) -- so it should be short in the hot kernel's terms too
        int64_t last = last_units < 1 ? 1 : last_units;
        if (last > blocks - (S - 1)) last = blocks - (S - 1);
        const int64_t rest = blocks - last;
        for (int k = 1; k < S; ++k) bound[k] = (rest * k / (S - 1)) * kSegAlign;
    } else {
        for (int k = 1; k < S; ++k) bound[k] = (blocks * k / S) * kSegAlign;
    }
    bound[S] = n_chunks;
    for (int k = 0; k < S; ++k)
        if (bound[k + 1] <= bound[k]) return 0;
    return S;
}

int iq_to_bits_streamed(urhgpu_ctx *ctx, const void *d_iq, int64_t n, const urhgpu_params *p, const urhgpu_outputs *out, void *host_blob,
                        int64_t cap_host, hipEvent_t ev_ready, bool *streamed, const void *h_iq, void *stage_blob, bool *staged, hipEvent_t ev_rows, int len16) {
    *streamed = false;
    if (staged) *staged = false;
    if (!ctx || !p || !out || n <= 0 || !d_iq || !out->rows || !out->counts) return URHGPU_ERR_ARG;
    if (dtype_bytes(p->dtype) == 0) return URHGPU_ERR_DTYPE;
    const bool want_bits = out->bits && out->msg_off && out->pauses && out->pos_off;
    if (!ctx->pipelined || !ctx->tail_stream || n <= 2 || p->mod == URHGPU_MOD_PSK || p->mod == URHGPU_MOD_ASK || !g_tile_tail || !want_bits ||
        (ctx->tune_stream_segments < 2 && !h_iq) || out->cap_rows < 1)
        return URHGPU_OK;
    URH_TRY(check_params(p, true));
    if (((uintptr_t)d_iq & 15) || (out->qad && ((uintptr_t)out->qad & 7))) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    const Plan pl = make_plan(ctx, n, p->tolerance);
    RunArgs a;
    memset(&a, 0, sizeof(a));
    URH_TRY(fill_thresholds(a, p));
    a.in = d_iq; a.qad = out->qad; a.left_halo = nullptr; a.n = n; a.pos_base = 0;
    a.chunk_len = pl.chunk_len; a.slab_stride = pl.slab_stride;
    a.noise_sqrd = p->noise_threshold * p->noise_threshold;
    a.noise_val = noise_for(p);
    a.tol = p->tolerance;
    a.lds_pad = ctx->hot_lds_pad;
    a.wide_int = (ctx->wide_int_next || ctx->tune_wide_int) ? 1 : 0;
    URH_TRY(max_magnitude_for(p->dtype, &a.max_magnitude));
    int64_t bound[kMaxSegments + 1];
    int S = runs_streamable(a) ? segment_bounds(pl.n_chunks, h_iq ? ctx->tune_upload_pieces : ctx->tune_stream_segments, h_iq ? 2 : 0, 1, bound) : 0;
    // DIRECT passes (stream_policy 3, or 4 for the passes that policy 0 would not stream): ONE segment -- the ordinary tail behind the hot
    // kernel (an event, no gate), but rows and packed results are STORED into the pinned host blob by the tail's own kernels: no pack of
    // the whole table at the end, no copy engine, no predicted copy size.
    // Policy 5 (the default): direct when the pass ships no positions (measured, profiles/r04c_ab_direct.txt: 0.294-0.300 ms per
    // pipelined step against 0.306 through pack + copy engine, one capture alone the same as with segments), policy 0 when it does --
    // 5.4 MB of uint32 positions stored over PCIe by a pack kernel take longer than the copy engine needs for the whole blob (0.41 ms).
    // STAGED passes (stream_policy 6; round 6): a direct pass whose "host blob" is a staging blob in HBM (stage_blob, the split layout of
    // compact.hpp: staged_layout); the caller ships it with the copy engine -- the row sections behind ev_rows, i.e. while the bits are
    // still being expanded, the head behind ev_ready.  Why: the row kernel's 1- and 4-byte
    // stores into pinned host memory -- 3.3 MB per GiB as some 10^5 partial-line PCIe writes issued over the 100 us the kernel runs beside
    // the next hot kernel -- cost that hot kernel 10 us per step (profiles/r06b_skips.txt: 0.2855 -> 0.2751 ms without them, the same as
    // without the row kernel altogether); the copy engine's writes do not pass through the shader's memory path.
    int policy = ctx->tune_stream_policy;
    if (policy == 5) policy = (p->write_bit_sample_pos && out->pos && !ctx->tune_stream_pos_direct) ? 0 : (ctx->tune_stream_latency ? 4 : (stage_blob ? 6 : 3));
    if (policy == 6 && !stage_blob) policy = 3;
    bool direct = false, to_stage = false;
    if (!h_iq && runs_streamable(a) && host_blob && (policy == 3 || policy == 6)) { direct = true; to_stage = (policy == 6); }
    if (S < 2 && !direct) return URHGPU_OK;                    // too short to cut, or not the bit-plane kernel's work: the ordinary path
    if (policy == 2 && !h_iq) return URHGPU_OK;
    if ((policy == 0 || policy == 4) && ctx->passes_begun > 0 && !h_iq) {
        // is anything of the pass before still running?  Then this pass's tail will run beside ITS successor's hot kernel as well: one piece
        const hipError_t q = hipEventQuery(ctx->ev_tail[(ctx->flip + 2) % 3]);
        if (q == hipErrorNotReady) {
            (void)hipGetLastError();
            if (policy == 0 || !host_blob) return URHGPU_OK;
            direct = true; to_stage = (stage_blob != nullptr);
        } else if (q != hipSuccess) URH_HIP(q);
    }
    if (direct) { S = 1; bound[0] = 0; bound[1] = pl.n_chunks; }
    void *const real_host_blob = host_blob;                   // (staged: the head still goes there, stored by the pass's last kernel)
    if (to_stage) { host_blob = stage_blob; if (staged) *staged = true; }
    const bool event_start = (h_iq != nullptr) || direct;      // the rows segments start behind events, not behind polling gates
    if (!ctx->d_seg) {
        URH_HIP(hipMalloc(&ctx->d_seg, 3 * kSegBlockBytes));
        URH_HIP(hipMemset(ctx->d_seg, 0, 3 * kSegBlockBytes));
        URH_HIP(hipDeviceSynchronize());
        URH_HIP(hipStreamCreateWithFlags(&ctx->bits_stream, hipStreamNonBlocking));
        for (int k = 0; k < kMaxSegments; ++k) URH_HIP(hipEventCreateWithFlags(&ctx->ev_piece[k], hipEventDisableTiming));
        for (int k = 0; k < 3; ++k) {
            URH_HIP(hipEventCreateWithFlags(&ctx->ev_hot_done[k], hipEventDisableTiming));
            URH_HIP(hipEventCreateWithFlags(&ctx->ev_bits[k], hipEventDisableTiming));
            for (int j = 0; j < kMaxSegments; ++j) URH_HIP(hipEventCreateWithFlags(&ctx->ev_rows[k][j], hipEventDisableTiming));
        }
    }
    // where the rows go on the host: the capacity layout of the compact blob (k_pack_seg) -- checked BEFORE anything of the pass is
    // queued or the arenas rotate (an error return below this point would leave a pass half-begun)
    const BitsParams bp = bits_params(p);
    const int has_pos = (bp.write_pos && out->pos) ? 1 : 0;
    int8_t *h_state = nullptr; int32_t *h_len = nullptr;
    BlobLayout host_layout;
    memset(&host_layout, 0, sizeof(host_layout));
    if (host_blob) {
        const int64_t caps[5] = {out->cap_rows, out->cap_msg, out->cap_bits, out->cap_pos, out->cap_rows};
        host_layout = blob_layout(caps, out->cap_rows, out->cap_bits, out->cap_msg, out->cap_pos, has_pos);
        if (cap_host < host_layout.total) return URHGPU_ERR_CAPACITY;
    }
    URH_TRY(begin_pipelined_pass(ctx));
    const int slot = ctx->flip;
    uint32_t *progress = (uint32_t *)((char *)ctx->d_seg + (size_t)slot * kSegBlockBytes);
    SegState *st = (SegState *)((char *)progress + kMaxSegments * kProgressStride * 4);
    static_assert(kMaxSegments * kProgressStride * 4 + sizeof(SegState) <= kSegBlockBytes && kMaxSegments <= 16, "segment block");
    if (ctx->seg_dirty[slot]) {                                // an earlier pass on this arena died half-way: its counters may not be zero
        URH_HIP(hipDeviceSynchronize());
        URH_HIP(hipMemset(progress, 0, kSegBlockBytes));
        URH_HIP(hipDeviceSynchronize());
        ctx->seg_dirty[slot] = false;
    }
    URH_TRY(ctx->arena.reserve(digitize_scratch_bytes(pl, out->cap_rows, false, true)));
    ctx->arena.reset();
    ctx->seg_dirty[slot] = true;                               // until the last segment has been queued
    hipStream_t s = ctx->stream;
    if (masked_hot_stream(p)) URH_TRY(hot_stream_begin(ctx, &s));
    ChunkInfo *chunks = (ChunkInfo *)ctx->arena.take((size_t)pl.n_chunks * sizeof(ChunkInfo));
    uint64_t *slab = (uint64_t *)ctx->arena.take((size_t)pl.n_chunks * pl.slab_stride * 8);
    void *rs_mem = ctx->arena.take(resolve_scratch_bytes(pl.n_chunks));
    if (!chunks || !slab || !rs_mem) return URHGPU_ERR_ARG;
    a.chunks = chunks; a.slab = slab;
    // everything that may allocate (and zero) descriptor memory BEFORE anything of the pass is queued
    TileTailMem tm;
    URH_TRY(tile_tail_mem(ctx, pl.n_chunks, true, &tm));
    const int64_t cap = std::max<int64_t>(out->cap_rows, 1);
    void *scratch = ctx->arena.take(bits_scratch_bytes(cap));
    if (!scratch) return URHGPU_ERR_ARG;
    ScanState ss;
    URH_TRY(scan_state(ctx, tile_desc_cap(cap, pl.n_chunks), &ss));
    // a segment's counter covers its chunks and the first chunk of the next segment (the resolve kernel's look-ahead), less the first
    // chunk of its own, which the segment before already waited for
    a.progress = progress; a.n_seg = S;
    uint32_t target[kMaxSegments];
    for (int k = 0; k < S; ++k) {
        const int64_t hi = (k < S - 1) ? bound[k + 1] + 1 : pl.n_chunks, lo = (k == 0) ? 0 : bound[k] + 1;
        a.seg_end[k] = (int32_t)hi;
        target[k] = (uint32_t)(hi - lo);
    }
    hipEvent_t hot_done = nullptr;
    if (h_iq) {
        // Upload mode (urhgpu_stream_push_upload): the capture arrives from the host in PIECES, and the hot kernel runs piece by piece
        // behind them (RunArgs::launch_lo / launch_hi; a chunk reads the two samples before it: the pieces arrive in order).  Piece k =
        // the chunks of segment k plus the first chunk of segment k + 1 -- the chunk segment k's resolve kernel looks ahead into, so that
        // the segment's tail does not wait for the next piece.  1 GiB over PCIe takes 70 times as long as its hot kernel: what is left
        // behind the last byte's arrival is the last (short) piece's kernel and the last segment's tail.
        // The copies go onto the HOT stream itself: copy 0, kernel 0, copy 1, kernel 1, ...  A stream of their own (copies fully beside
        // the kernels) was measured first and is not robust: HIP maps streams onto a handful of hardware queues, and depending on which
        // streams happened to share one the same pass took 19.4 or 35 ms (tools/upload_probe.py, round 4: the first pipeline of a
        // process was fine, later ones were not).  In one in-order stream a piece's kernel sits between two copies: 283 us of kernels
        // per GiB whatever the number of pieces, plus some 25 us of hand-over per piece -- 1.04 x the bare copy at four pieces.
        // No polling gates here: a gate kernel would spin for the milliseconds a piece takes to arrive; the rows segment of piece k
        // waits for an event behind the piece's hot kernel instead (plain stores in that kernel, no progress counters).
        a.progress = nullptr; a.n_seg = 0;
        const size_t bps = (size_t)dtype_bytes(p->dtype);
        int64_t piece_lo[kMaxSegments], piece_hi[kMaxSegments];
        for (int k = 0; k < S; ++k) {
            piece_lo[k] = k == 0 ? 0 : piece_hi[k - 1];
            piece_hi[k] = (k < S - 1) ? std::min<int64_t>(bound[k + 1] + 1, pl.n_chunks) : pl.n_chunks;
        }
        hipStream_t up = s;
        for (int k = 0; k < S; ++k) {
            {
                const int64_t s0 = piece_lo[k] * pl.chunk_len, s1 = std::min<int64_t>(piece_hi[k] * pl.chunk_len, n);
                URH_HIP(hipMemcpyAsync((char *)const_cast<void *>(d_iq) + (size_t)s0 * bps, (const char *)h_iq + (size_t)s0 * bps, (size_t)(s1 - s0) * bps,
                                       hipMemcpyHostToDevice, up));
            }
            a.launch_lo = piece_lo[k]; a.launch_hi = piece_hi[k];
            const int stl = launch_demod_runs_iq(a, p->dtype, p->mod, out->qad != nullptr, s);
            if (stl != URHGPU_OK) return stl;
            URH_HIP(hipEventRecord(ctx->ev_piece[k], s));
        }
        a.launch_lo = 0; a.launch_hi = 0;
        URH_HIP(hipEventRecord(ctx->ev_hot_done[slot], s));
        hot_done = ctx->ev_hot_done[slot];
    } else {
    if (direct) { a.progress = nullptr; a.n_seg = 0; }         // (plain stores in the hot kernel, no counters: the tail starts behind its end)
    const bool prof = prof_begin_record(ctx, s);
    // the hot kernel's completion: the dispatch's own completion signal where the launcher takes events (an event recorded behind the
    // kernel is one more barrier packet between two hot kernels); nobody waits for it before the last segment has been queued
    if (!prof) { g_hot_events = HotEvents(); g_hot_events.stop = ctx->ev_hot_done[slot]; }
    hot_done = g_hot_events.stop;
    {
        const int stl = launch_demod_runs_iq(a, p->dtype, p->mod, out->qad != nullptr, s);
        if (stl != URHGPU_OK) { g_hot_events = HotEvents(); return stl; }
    }
    if (hot_done && !g_hot_events.used) hot_done = nullptr;
    if (prof) URH_TRY(prof_end_record(ctx, s));
    else g_hot_events = HotEvents();
    if (!hot_done) { URH_HIP(hipEventRecord(ctx->ev_hot_done[slot], s)); hot_done = ctx->ev_hot_done[slot]; }
    }
    // ---- the tail in segments: rows segments on the tail stream, bits segments on the bits stream behind the rows they expand; neither
    // ever waits for the hot kernel as a whole ----
    hipStream_t ts = ctx->tail_stream, tb = ctx->bits_stream;
    const ResolveScratch rsc = resolve_scratch_carve(rs_mem, pl.n_chunks);
    ResolveArgs r;
    memset(&r, 0, sizeof(r));
    r.sc = rsc;
    r.chunks = chunks; r.n_chunks = pl.n_chunks; r.n_total = n; r.tol = p->tolerance;
    r.rows = out->rows; r.cap_rows = out->cap_rows; r.d_n_acc = &st->n_acc; r.d_n_rows = &st->rows_at[S - 1];
    r.d_n_rows_needed = &st->rows_needed; r.write_last_row = 1;
    r.local_pass = 0; r.aux = (ResolveAux *)(ctx->d_tickets + 4); r.summary_out = nullptr; r.chunk_first = 0; r.n_local = pl.n_chunks; r.d_ts_carry = nullptr;
    EmitArgs e;
    e.sc = rsc;
    e.chunks = chunks; e.chunk_first = 0; e.slab = slab; e.slab_stride = pl.slab_stride;
    e.rows = out->rows; e.cap_rows = out->cap_rows; e.d_ts_carry = nullptr; e.is_ask = 0; e.sps = p->samples_per_symbol;
    BitsOut bo{out->bits, out->cap_bits, out->msg_off, out->pauses, out->cap_msg, out->pos, out->cap_pos, out->pos_off, out->counts, out->h_counts};
    if (host_blob) { h_state = (int8_t *)((char *)host_blob + host_layout.off_row_state); h_len = (int32_t *)((char *)host_blob + host_layout.off_row_len); }
    if (to_stage) {
        const StagedLayout SL = staged_layout(out->cap_rows, out->cap_bits, out->cap_msg, out->cap_pos, has_pos);
        h_state = (int8_t *)((char *)host_blob + SL.off_row_state); h_len = (int32_t *)((char *)host_blob + SL.off_row_len);
    }
    // staged passes with 16-bit row lengths: the escape list (count, then pairs) sits in the staging blob's head region, behind the header's place;
    // len16 == 2 (URHGPU_BLOB_ROW16: state and length in one uint16, escapes from 8191 samples on): the list is longer and has a place of
    // its own behind every section of the split layout, in the staging blob while it is built and in the host blob (the caller sized both)
    const bool l16 = to_stage && len16 != 0;
    const bool row16 = l16 && len16 == 2;
    int64_t row16_esc_off = 0;
    if (row16) row16_esc_off = staged_layout(out->cap_rows, out->cap_bits, out->cap_msg, out->cap_pos, has_pos).total;
    int64_t *esc = l16 ? (int64_t *)((char *)host_blob + (row16 ? row16_esc_off : (int64_t)URHGPU_BLOB_HEADER_BYTES)) : nullptr;
    const int64_t esc_cap = row16 ? n / 8191 + 2 : n / 65535 + 2;
    SegPackDst dst{host_blob, cap_host, progress, 0, (direct && ctx->tune_stream_pos_direct) ? 1 : 0, to_stage ? 1 : 0, to_stage ? real_host_blob : nullptr,
                   esc, esc_cap, row16_esc_off};
    // bits segments: the last one is the last rows segment alone (what is exposed behind the hot kernel), the others share the rest
    int Sb = h_iq ? S : 1;       // (an upload: every piece's bits behind its rows -- the pieces are milliseconds apart)
    if (Sb > S) Sb = S;
    if (Sb < 1) Sb = 1;
    int bits_end_at[kMaxSegments];                             // bits segment j ends with rows segment bits_end_at[j]
    for (int j = 0; j < Sb - 1; ++j) bits_end_at[j] = (int)((int64_t)(S - 1) * (j + 1) / (Sb - 1)) - 1;
    bits_end_at[Sb - 1] = S - 1;
    int jb = 0;
    int64_t bits_from = 0;
    // the LAST bits segment goes onto the rows stream, right behind the last rows: no event hop between two
    // streams on the chain that is exposed behind the hot kernel's end; the rows stream then waits for the bits segments before it
    const bool final_on_rows = true;
    hipStream_t last_stream = tb;
    for (int k = 0; k < S; ++k) {
        if (h_iq) URH_HIP(hipStreamWaitEvent(ts, ctx->ev_piece[k], 0));
        else if (direct) URH_HIP(hipStreamWaitEvent(ts, hot_done, 0));
        RowsSegment sg{k, k == S - 1 ? 1 : 0, bound[k], bound[k + 1], SegGate{event_start ? nullptr : progress, k, target[k], k == 0 ? 1 : 0, st, (long long)200000000, 0},
                       h_state, h_len, 1, l16 ? (row16 ? 2 : 1) : 0, esc, esc_cap};
        URH_TRY(launch_rows_segment(r, e, tm, bp, st, sg, ts));
        if (to_stage && ev_rows) URH_HIP(hipEventRecord(ev_rows, ts));      // (one segment: every row section is in the staging blob)
        while (jb < Sb && bits_end_at[jb] < k) ++jb;           // (a bits segment that would end before the first rows segment: none)
        if (jb < Sb && bits_end_at[jb] == k) {
            const bool last = (jb == Sb - 1);
            BitsSegment bs{jb, last ? 1 : 0, bits_from, bound[k + 1], k, st};
            if (last && final_on_rows) {
                if (jb > 0) {                                  // behind the bits segments before it (their carries, their packed bytes)
                    URH_HIP(hipEventRecord(ctx->ev_bits[slot], tb));
                    URH_HIP(hipStreamWaitEvent(ts, ctx->ev_bits[slot], 0));
                }
                URH_TRY(launch_bits_segment(tm, bp, bo, scratch, ss, out->rows, out->cap_rows, bs, &dst, ts));
                last_stream = ts;
            } else {
                URH_HIP(hipEventRecord(ctx->ev_rows[slot][jb], ts));
                URH_HIP(hipStreamWaitEvent(tb, ctx->ev_rows[slot][jb], 0));
                URH_TRY(launch_bits_segment(tm, bp, bo, scratch, ss, out->rows, out->cap_rows, bs, &dst, tb));
            }
            bits_from = bound[k + 1];
            ++jb;
        }
    }
    URH_HIP(hipGetLastError());
    if (!host_blob) URH_HIP(hipMemsetAsync(progress, 0, kMaxSegments * kProgressStride * 4, last_stream));       // (nobody else zeroes the counters then)
    ctx->seg_dirty[slot] = false;
    if (ev_ready) URH_HIP(hipEventRecord(ev_ready, last_stream));         // the host blob is complete
    // the pass is over when the bits stream has finished and the hot kernel has retired (its last qad stores): cheap here, the last
    // gate has just seen its last chunk
    if (last_stream != ts) {
        URH_HIP(hipEventRecord(ctx->ev_bits[slot], tb));
        URH_HIP(hipStreamWaitEvent(ts, ctx->ev_bits[slot], 0));
    }
    URH_HIP(hipStreamWaitEvent(ts, hot_done, 0));
    if (s != ctx->stream && ctx->stream != nullptr) URH_HIP(hipStreamWaitEvent(ctx->stream, hot_done, 0));   // input reuse in stream order
    URH_TRY(end_pipelined_pass(ctx));
    *streamed = true;
    return URHGPU_OK;
}

}  // namespace urh

extern "C" {

int urhgpu_version(void) { return URHGPU_VERSION; }

const char *urhgpu_strerror(int status) {
    switch (status) {
        case URHGPU_OK: return "ok";
        case URHGPU_ERR_HIP: return "HIP runtime error";
        case URHGPU_ERR_DTYPE: return "Unsupported dtype";
        case URHGPU_ERR_ARG: return "bad argument";
        case URHGPU_ERR_CAPACITY: return "output capacity too small";
        case URHGPU_ERR_UNSUPPORTED: return "parameter outside the supported range";
        case URHGPU_ERR_NO_DEVICE: return "no usable GPU";
        default: return "unknown status";
    }
}

const char *urhgpu_last_hip_error(void) { return g_hip_err; }

int urhgpu_device_count(int *count) {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *count = 0; (void)hipGetLastError(); return URHGPU_ERR_NO_DEVICE; }
    *count = c;
    return URHGPU_OK;
}

int urhgpu_ctx_create(int device, urhgpu_ctx **out) {
    if (!out) return URHGPU_ERR_ARG;
    int c = 0;
    if (urhgpu_device_count(&c) != URHGPU_OK || c <= 0 || device < 0 || device >= c) return URHGPU_ERR_NO_DEVICE;
    urhgpu_ctx *ctx = new (std::nothrow) urhgpu_ctx();
    if (!ctx) return URHGPU_ERR_ARG;
    ctx->device = device;
    URH_HIP(hipSetDevice(device));
    URH_HIP(hipGetDeviceProperties(&ctx->prop, device));
    URH_HIP(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;
    URH_HIP(hipMalloc((void **)&ctx->d_counts, 16 * sizeof(int64_t)));
    URH_HIP(hipMalloc((void **)&ctx->d_tickets, 16 * sizeof(int32_t)));     // [0..3] elections, [4..7] ResolveAux, [8..9] tile tail's huge-row counters
    URH_HIP(hipMemset(ctx->d_tickets, 0, 16 * sizeof(int32_t)));
    {   // d_tickets[4..7] is the ResolveAux block of the resolve kernels: kAuxNone x3, -1
        const int32_t aux0[4] = {kAuxNone, kAuxNone, kAuxNone, -1};
        URH_HIP(hipMemcpy(ctx->d_tickets + 4, aux0, sizeof(aux0), hipMemcpyHostToDevice));
    }
    URH_HIP(hipHostMalloc((void **)&ctx->h_counts, 32 * sizeof(int64_t)));
    memset(ctx->h_counts, 0, 32 * sizeof(int64_t));
    if (hipHostMalloc((void **)&ctx->h_small, kSmallPinned) != hipSuccess) { (void)hipGetLastError(); ctx->h_small = nullptr; }   // (optional: pageable copies work too)
    *out = ctx;
    return URHGPU_OK;
}

int urhgpu_ctx_destroy(urhgpu_ctx *ctx) {
    if (!ctx) return URHGPU_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->arena.release();
    ctx->staging.release();
    ctx->aux.release();
    ctx->fir_work.release();
    if (ctx->ev_fir) (void)hipEventDestroy(ctx->ev_fir);
    ctx->arena_alt.release();
    ctx->arena_alt2.release();
    if (ctx->hot_masked) { (void)hipStreamSynchronize(ctx->hot_masked); (void)hipStreamDestroy(ctx->hot_masked); }
    if (ctx->ev_in) (void)hipEventDestroy(ctx->ev_in);
    if (ctx->tail_stream) (void)hipStreamSynchronize(ctx->tail_stream);
    if (ctx->own_tail_stream && ctx->tail_stream) (void)hipStreamDestroy(ctx->tail_stream);
    if (ctx->d_seg) {
        (void)hipFree(ctx->d_seg);
        if (ctx->bits_stream) { (void)hipStreamSynchronize(ctx->bits_stream); (void)hipStreamDestroy(ctx->bits_stream); }
        for (hipEvent_t e : ctx->ev_piece) if (e) (void)hipEventDestroy(e);
        for (int k = 0; k < 3; ++k) {
            if (ctx->ev_hot_done[k]) (void)hipEventDestroy(ctx->ev_hot_done[k]);
            if (ctx->ev_bits[k]) (void)hipEventDestroy(ctx->ev_bits[k]);
            for (hipEvent_t e : ctx->ev_rows[k]) if (e) (void)hipEventDestroy(e);
        }
    }
    if (ctx->ev_hot) { (void)hipEventDestroy(ctx->ev_hot); (void)hipEventDestroy(ctx->ev_tail[0]); (void)hipEventDestroy(ctx->ev_tail[1]); (void)hipEventDestroy(ctx->ev_tail[2]); }
    delete (ShardSession *)ctx->shard;
    for (hipEvent_t e : ctx->prof_events) (void)hipEventDestroy(e);
    if (ctx->d_counts) (void)hipFree(ctx->d_counts);
    if (ctx->d_tickets) (void)hipFree(ctx->d_tickets);
    if (ctx->d_desc) (void)hipFree(ctx->d_desc);
    if (ctx->d_rdesc) (void)hipFree(ctx->d_rdesc);
    if (ctx->h_counts) (void)hipHostFree(ctx->h_counts);
    if (ctx->h_small) (void)hipHostFree(ctx->h_small);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return URHGPU_OK;
}

int urhgpu_ctx_set_stream(urhgpu_ctx *ctx, void *hip_stream) {
    if (!ctx) return URHGPU_ERR_ARG;
    ctx->stream = (hipStream_t)hip_stream;
    return URHGPU_OK;
}

int urhgpu_ctx_use_private_stream(urhgpu_ctx *ctx) {
    if (!ctx) return URHGPU_ERR_ARG;
    ctx->stream = ctx->own_stream;
    return URHGPU_OK;
}

int urhgpu_ctx_sync(urhgpu_ctx *ctx) {
    if (!ctx) return URHGPU_ERR_ARG;
    if (ctx->hot_masked) URH_HIP(hipStreamSynchronize(ctx->hot_masked));
    if (ctx->bits_stream) URH_HIP(hipStreamSynchronize(ctx->bits_stream));
    if (ctx->tail_stream) URH_HIP(hipStreamSynchronize(ctx->tail_stream));
    URH_HIP(hipStreamSynchronize(ctx->stream));
    ctx->tail_pending = false;
    return URHGPU_OK;
}

int urhgpu_ctx_set_pipelined(urhgpu_ctx *ctx, int enable, void *tail_stream) {
    if (!ctx) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(urhgpu_ctx_sync(ctx));
    if (ctx->own_tail_stream && ctx->tail_stream) { (void)hipStreamDestroy(ctx->tail_stream); }
    ctx->tail_stream = nullptr; ctx->own_tail_stream = false; ctx->pipelined = false;
    if (ctx->hot_masked) { (void)hipStreamSynchronize(ctx->hot_masked); (void)hipStreamDestroy(ctx->hot_masked); ctx->hot_masked = nullptr; }
    if (!enable) return URHGPU_OK;
    if (tail_stream) ctx->tail_stream = (hipStream_t)tail_stream;
    else {
        URH_HIP(hipStreamCreateWithFlags(&ctx->tail_stream, hipStreamNonBlocking));
        ctx->own_tail_stream = true;
    }
    if (!ctx->ev_hot) {
        URH_HIP(hipEventCreateWithFlags(&ctx->ev_hot, hipEventDisableTiming));
        URH_HIP(hipEventCreateWithFlags(&ctx->ev_tail[0], hipEventDisableTiming));
        URH_HIP(hipEventCreateWithFlags(&ctx->ev_tail[1], hipEventDisableTiming));
        URH_HIP(hipEventCreateWithFlags(&ctx->ev_tail[2], hipEventDisableTiming));
    }
    // The hot kernel of a pipelined pass runs on a private stream whose CU mask leaves hot_cus_removed CUs per XCD out (default 4: 224 of
    // the 256 CUs).  Measured (round 3, profiles/HISTORY.md): the kernel -- and a pure copy of its shape -- is FASTEST there: 0.2666 ms
    // = 6.04 TB/s on 224 CUs against 0.2799 ms = 5.75 TB/s on all 256 (248 / 240 / 232 CUs: 0.2746 / 0.2718 / 0.2708; 208 / 192: 0.2746 /
    // 0.2752; 160: 0.311): 256 CUs of streaming wavefronts ask more of the HBM than it serves well.  And the 32 CUs it leaves alone are
    // where the previous pass's tail, the blob packing and the collectives of sharded passes find their wave slots at once.  The mask
    // bits of the removed CUs are chosen so that every XCD loses the same number whichever way bits map to XCDs (bit i -> XCD i / 32 or
    // i % 8): class (i % 8 - i / 32) mod 8 and slot (i / 8) % 4 enumerate 32 sets of 8 CUs, one per XCD each.
    if (ctx->tune_hot_cus_removed > 0 && ctx->prop.multiProcessorCount == 256) {
        const int words = 8;
        uint32_t mask[8];
        hot_cu_mask(ctx->tune_hot_cus_removed, mask);
        // (a runtime that cannot make the masked stream is no reason to fail: the hot kernel then runs on the caller's stream as before)
        if (hipExtStreamCreateWithCUMask(&ctx->hot_masked, (uint32_t)words, mask) != hipSuccess) { (void)hipGetLastError(); ctx->hot_masked = nullptr; }
        if (ctx->hot_masked && !ctx->ev_in && hipEventCreateWithFlags(&ctx->ev_in, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipStreamDestroy(ctx->hot_masked);
            ctx->hot_masked = nullptr;
        }
    }
    ctx->pipelined = true;
    return URHGPU_OK;
}

// Tuning values of the pipelined mode (defaults = what is measured and shipped; the A/B tool tools/ab.sh sets others through bench.py's
// URH_TUNE_* environment, read THERE -- the library itself reads no environment variable).  The knobs earlier rounds measured as useless
// are gone; their records are in profiles/HISTORY.md.
//   hot_lds_kb               dynamic LDS per hot workgroup in KiB (fewer of them per CU: room for the previous pass's tail); default 0
//   hot_lds_kb_sharded       the same for the urhgpu_shard_* passes that keep the generic tail (ASK); default 33
//   hot_cus_removed_per_xcd  CUs per XCD the hot kernel of a pipelined pass leaves alone (before urhgpu_ctx_set_pipelined); default 4, 0: no mask
//   profile_bracket          1: urhgpu_ctx_profile_* report the stream-level bracket around the hot launch instead of the dispatch's own timing
//   stream_policy            which tail a pass of urhgpu_stream_* takes (common.hpp: tune_stream_policy); default 5
//   stream_segments          rows segments of a segmented pass; default 7
//   stream_latency           1: a pass that finds the pipeline idle runs its tail in segments (lowest latency for ONE capture); default 0
//   stream_pos_direct        1 (default): direct passes ship bit_sample_pos themselves
//   upload_pieces            pieces of urhgpu_stream_push_upload; default 4
//   spin_wait                1 (default): the estimator calls poll their stream for the few hundred microseconds they wait (wait_stream)
//   wide_int                 1: passes over SIGNED INTEGER FSK captures take the hot kernel's instantiation with the wide loop (captures whose phase
//                            steps leave the fast loop's window, DESIGN 4: a quarter faster there, 5 % slower on narrow ones).  Capture streams
//                            decide by themselves (k_wide_probe); one-shot and sharded passes have no probe to go by: this key is the caller's word.  default 0
//   shard_summary_generic    1: the local pass of urhgpu_shard_runs_dev as the three generic resolve launches instead of k_shard_summary; default 0
int urhgpu_ctx_set_tuning(urhgpu_ctx *ctx, const char *key, int value) {
    if (!ctx || !key) return URHGPU_ERR_ARG;
    if (!strcmp(key, "hot_lds_kb")) { if (value < 0 || value > 150) return URHGPU_ERR_ARG; ctx->hot_lds_pad = value * 1024; }
    else if (!strcmp(key, "hot_lds_kb_sharded")) { if (value < 0 || value > 150) return URHGPU_ERR_ARG; ctx->hot_lds_pad_sharded = value * 1024; }
    else if (!strcmp(key, "profile_bracket")) ctx->prof_bracket = value != 0;
    else if (!strcmp(key, "hot_cus_removed_per_xcd")) { if (value < 0 || value > 16) return URHGPU_ERR_ARG; ctx->tune_hot_cus_removed = value; }
    else if (!strcmp(key, "stream_segments")) { if (value < 1 || value > kMaxSegments) return URHGPU_ERR_ARG; ctx->tune_stream_segments = value; }
    else if (!strcmp(key, "stream_policy")) { if (value < 0 || value > 6) return URHGPU_ERR_ARG; ctx->tune_stream_policy = value; }
    else if (!strcmp(key, "stream_latency")) { ctx->tune_stream_latency = value != 0; }
    else if (!strcmp(key, "stream_pos_direct")) { ctx->tune_stream_pos_direct = value != 0; }
    else if (!strcmp(key, "spin_wait")) { ctx->tune_spin_wait = value != 0; }
    else if (!strcmp(key, "shard_summary_generic")) { ctx->tune_shard_summary_generic = value != 0; }
    else if (!strcmp(key, "wide_int")) { ctx->tune_wide_int = value != 0; }
    else if (!strcmp(key, "upload_pieces")) { if (value < 2 || value > kMaxSegments) return URHGPU_ERR_ARG; ctx->tune_upload_pieces = value; }
    else return URHGPU_ERR_ARG;
    return URHGPU_OK;
}

int urhgpu_ctx_join(urhgpu_ctx *ctx) {
    if (!ctx) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    return join_tail(ctx);
}

// Does this host's libm evaluate sinf / cosf / atan2f the way the device code restates them?  The reference's Costas loop and FSK
// demodulation call the HOST's libm (signal_functions.pyx:252-330, :375, compiled as C++: sinf / cosf / atan2f), which is not correctly
// rounded: x86-64 glibc picks an FMA or a non-FMA build of sinf / cosf at run time, and the two differ on about one float in 10^9.  The
// device code restates the FMA build (glibc_sincosf.h, URH_SINCOSF_FMA = 1).  Checked: the 17 arguments below |x| = 120 on which the two
// builds differ (tools/libm_probe/scan.c finds them: an exhaustive scan), both signs, plus pseudo-random arguments; atan2f (one build
// in glibc) on pseudo-random operand pairs of every quadrant.  out4 = {sinf / cosf results compared, mismatches, atan2f results
// compared, mismatches}.  A mismatch means: on THIS host the reference itself would produce other bits than on the hosts the parity
// tests ran on, and the GPU's PSK / FSK output follows those, not this host's reference.  Host arithmetic only; no GPU needed.
int urhgpu_host_libm_check(int64_t *out4) {
    if (!out4) return URHGPU_ERR_ARG;
    static const uint32_t kDiscriminating[17] = {0x418a3adbu, 0x418a3adcu, 0x418a3addu, 0x418a3adeu, 0x41bc76d9u, 0x4202eb4bu, 0x4255b0a9u, 0x4280ce28u,
                                                 0x42687a55u, 0x42a35c07u, 0x42a35d44u, 0x42870e40u, 0x42a97360u, 0x42c55faau, 0x42d8d23eu, 0x42e87a55u,
                                                 0x42cf5854u};
    int64_t n_sc = 0, bad_sc = 0, n_at = 0, bad_at = 0;
    auto same = [](float a, float b) { uint32_t x, y; memcpy(&x, &a, 4); memcpy(&y, &b, 4); return x == y || (a != a && b != b); };
    auto check_sc = [&](float x) {
        volatile float vx = x;                                   // (keep the compiler from folding the libm calls)
        n_sc += 2;
        if (!same(sinf(vx), urh_sinf(x))) ++bad_sc;
        if (!same(cosf(vx), urh_cosf(x))) ++bad_sc;
    };
    for (uint32_t u : kDiscriminating) {
        float x; memcpy(&x, &u, 4);
        check_sc(x); check_sc(-x);
    }
    uint64_t s = 0x243f6a8885a308d3ull;
    auto next = [&]() { s += 0x9e3779b97f4a7c15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); };
    for (int i = 0; i < 4096; ++i) {
        const uint64_t z = next();
        check_sc((float)((double)(int64_t)(z >> 11) * (1.0 / 9007199254740992.0) * 240.0 - 120.0));      // uniform in (-120, 120)
        const uint32_t a = (uint32_t)z, b = (uint32_t)(z >> 32);
        // operands of every sign and of magnitudes 2^-20 .. 2^20
        float y, x;
        const uint32_t uy = (a & 0x807fffffu) | (((a >> 23) % 41u + 107u) << 23), ux = (b & 0x807fffffu) | (((b >> 23) % 41u + 107u) << 23);
        memcpy(&y, &uy, 4); memcpy(&x, &ux, 4);
        volatile float vy = y, vx = x;
        ++n_at;
        if (!same(atan2f(vy, vx), urh_atan2f(y, x))) ++bad_at;
    }
    out4[0] = n_sc; out4[1] = bad_sc; out4[2] = n_at; out4[3] = bad_at;
    return URHGPU_OK;
}

int urhgpu_ctx_info(urhgpu_ctx *ctx, int *compute_units, int *wavefront, int64_t *hbm_bytes, char *name, int name_cap) {
    if (!ctx) return URHGPU_ERR_ARG;
    if (compute_units) *compute_units = ctx->prop.multiProcessorCount;
    if (wavefront) *wavefront = ctx->prop.warpSize;
    if (hbm_bytes) *hbm_bytes = (int64_t)ctx->prop.totalGlobalMem;
    if (name && name_cap > 0) { strncpy(name, ctx->prop.name, (size_t)name_cap - 1); name[name_cap - 1] = 0; }
    return URHGPU_OK;
}

int urhgpu_ctx_reserve(urhgpu_ctx *ctx, int64_t n_samples, int tolerance) {
    if (!ctx || n_samples < 0 || tolerance < 0) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    const Plan pl = make_plan(ctx, n_samples, tolerance);
    const int64_t cap_rows = n_samples / ((int64_t)tolerance + 1) + 2;
    URH_TRY(ctx->arena.reserve(digitize_scratch_bytes(pl, cap_rows, true, true)));
    if (ctx->pipelined) {
        URH_TRY(ctx->arena_alt.reserve(digitize_scratch_bytes(pl, cap_rows, true, true)));
        URH_TRY(ctx->arena_alt2.reserve(digitize_scratch_bytes(pl, cap_rows, true, true)));
    }
    return URHGPU_OK;
}

int urhgpu_ctx_profile_begin(urhgpu_ctx *ctx, int max_records) {
    if (!ctx || max_records < 0) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    while ((int)ctx->prof_events.size() < 4 * max_records) {
        hipEvent_t e;
        URH_HIP(hipEventCreate(&e));
        ctx->prof_events.push_back(e);
    }
    ctx->prof_used = 0;
    ctx->prof_on = max_records > 0;
    return URHGPU_OK;
}

int urhgpu_ctx_profile_end(urhgpu_ctx *ctx, float *ms_out, int cap, int *n_records) {
    if (!ctx || !n_records) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    if (ctx->hot_masked) URH_HIP(hipStreamSynchronize(ctx->hot_masked));
    URH_HIP(hipStreamSynchronize(ctx->stream));
    ctx->prof_on = false;
    const int n = ctx->prof_used;
    *n_records = n;
    const bool bracket = ctx->prof_bracket;                          // report the stream-level bracket instead (comparison)
    for (int k = 0; k < n && k < cap; ++k) {
        const int base = 4 * k + ((ctx->prof_dispatch[(size_t)k] && !bracket) ? 2 : 0);
        URH_HIP(hipEventElapsedTime(&ms_out[k], ctx->prof_events[base], ctx->prof_events[base + 1]));
    }
    return URHGPU_OK;
}

int urhgpu_get_center_thresholds(float center, float spacing, int modulation_order, float *out) {
    // signal_functions.pyx:380-390; int -> float conversion, fp32 multiply and add/sub, no contraction
    const int n = modulation_order / 2;
    for (int i = 0; i < n; ++i) out[i] = center - (float)(n - (i + 1)) * spacing;
    for (int i = n; i < modulation_order - 1; ++i) out[i] = center + (float)(i + 1 - n) * spacing;
    return URHGPU_OK;
}

// ---- device-pointer entry points -------------------------------------------------------------------
int urhgpu_afp_demod_dev(urhgpu_ctx *ctx, const void *d_iq, int64_t n, const urhgpu_params *p, float *d_qad) {
    if (!ctx || !p || n < 0 || (n > 0 && (!d_iq || !d_qad))) return URHGPU_ERR_ARG;
    if (dtype_bytes(p->dtype) == 0) return URHGPU_ERR_DTYPE;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(join_tail(ctx));
    if (n <= 2) {                                   // signal_functions.pyx:335-336
        if (n > 0) URH_HIP(hipMemsetAsync(d_qad, 0, (size_t)n * 4, ctx->stream));
        return URHGPU_OK;
    }
    if (((uintptr_t)d_iq & 15) || ((uintptr_t)d_qad & 7)) return URHGPU_ERR_ARG;
    if (p->mod == URHGPU_MOD_PSK) {
        URH_TRY(ctx->aux.reserve(costas_scratch_bytes(n) + 1024));
        ctx->aux.reset();
        void *scratch = ctx->aux.take(costas_scratch_bytes(n));
        URH_TRY(launch_costas(ctx, d_iq, n, p, d_qad, scratch));
        URH_HIP(hipGetLastError());
        return URHGPU_OK;
    }
    RunArgs a;
    memset(&a, 0, sizeof(a));
    a.in = d_iq; a.qad = d_qad; a.n = n; a.left_halo = nullptr;
    a.noise_sqrd = p->noise_threshold * p->noise_threshold;
    a.noise_val = noise_for(p);
    URH_TRY(max_magnitude_for(p->dtype, &a.max_magnitude));
    const int64_t rows = (n + 511) / 512;                        // k_afp_demod: 512 samples per workgroup-wide load
    const int grid = (int)std::min<int64_t>(rows, (int64_t)ctx->prop.multiProcessorCount * 16);
    URH_TRY(launch_afp_demod(a, p->dtype, p->mod, grid, ctx->stream));
    URH_HIP(hipGetLastError());
    return URHGPU_OK;
}

int urhgpu_grab_pulse_lens_dev(urhgpu_ctx *ctx, const float *d_qad, int64_t n, const urhgpu_params *p,
                               int64_t *d_rows, int64_t cap_rows, int64_t *d_n_rows) {
    if (!ctx || !p || n < 0 || cap_rows < 0 || !d_n_rows) return URHGPU_ERR_ARG;
    URH_TRY(check_params(p, false));
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(join_tail(ctx));
    if (n == 0) {                                   // signal_functions.pyx:416-417
        URH_HIP(hipMemsetAsync(d_n_rows, 0, 8, ctx->stream));
        URH_HIP(hipMemsetAsync(ctx->d_counts, 0, 16 * 8, ctx->stream));
        return URHGPU_OK;
    }
    if (!d_qad || !d_rows || ((uintptr_t)d_qad & 7)) return URHGPU_ERR_ARG;
    const Plan pl = make_plan(ctx, n, p->tolerance);
    URH_TRY(ctx->arena.reserve(digitize_scratch_bytes(pl, cap_rows, p->mod == URHGPU_MOD_ASK, false)));
    ctx->arena.reset();
    return digitize(ctx, false, d_qad, n, p, nullptr, d_rows, cap_rows, d_n_rows, ctx->d_counts + 8, ctx->d_counts + 9, pl);
}

static int ppseq_to_bits_inner(urhgpu_ctx *ctx, const int64_t *d_rows, const int64_t *d_n_rows, int64_t cap,
                               const urhgpu_params *p, const urhgpu_outputs *out, void *scratch,
                               const int64_t *d_rows_needed = nullptr) {
    BitsOut bo{out->bits, out->cap_bits, out->msg_off, out->pauses, out->cap_msg, out->pos, out->cap_pos, out->pos_off, out->counts, out->h_counts};
    BitsParams bp = bits_params(p);
    bp.d_rows_needed = d_rows_needed;
    ScanState ss;
    URH_TRY(scan_state(ctx, cap, &ss));
    URH_TRY(launch_ppseq_to_bits(d_rows, d_n_rows, cap, bp, bo, scratch, ss, ctx->stream));
    URH_HIP(hipGetLastError());
    return URHGPU_OK;
}

int urhgpu_ppseq_to_bits_dev(urhgpu_ctx *ctx, const int64_t *d_rows, const int64_t *d_n_rows, int64_t cap_rows_hint,
                             const urhgpu_params *p, const urhgpu_outputs *out) {
    if (!ctx || !p || !out || !d_n_rows || cap_rows_hint < 0) return URHGPU_ERR_ARG;
    if (p->bits_per_symbol < 1 || p->samples_per_symbol < 1) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(join_tail(ctx));
    const int64_t cap = std::max<int64_t>(cap_rows_hint, 1);
    URH_TRY(ctx->arena.reserve(bits_scratch_bytes(cap) + 4096));
    ctx->arena.reset();
    void *scratch = ctx->arena.take(bits_scratch_bytes(cap));
    if (!scratch) return URHGPU_ERR_ARG;
    URH_TRY(ppseq_to_bits_inner(ctx, d_rows, d_n_rows, cap, p, out, scratch));
    if (out->blob) {                                       // compact mirror: out->rows must then be the table d_rows (and cap_rows its capacity)
        if (out->rows != d_rows) return URHGPU_ERR_ARG;
        URH_TRY(launch_pack_blob(out, p->write_bit_sample_pos, ctx->stream));
    }
    return URHGPU_OK;
}

int urhgpu_iq_to_bits_dev(urhgpu_ctx *ctx, const void *d_iq, int64_t n, const urhgpu_params *p,
                          const urhgpu_outputs *out) {
    if (!ctx || !p || !out || n <= 0 || !d_iq || !out->rows || !out->counts) return URHGPU_ERR_ARG;
    if (dtype_bytes(p->dtype) == 0) return URHGPU_ERR_DTYPE;
    URH_TRY(check_params(p, out->bits != nullptr));
    if (((uintptr_t)d_iq & 15) || (out->qad && ((uintptr_t)out->qad & 7))) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    const Plan pl = make_plan(ctx, n, p->tolerance);
    const bool ask = (p->mod == URHGPU_MOD_ASK);
    const bool fused = !(n <= 2 || p->mod == URHGPU_MOD_PSK);
    const bool piped = ctx->pipelined && fused;
    if (piped) URH_TRY(begin_pipelined_pass(ctx)); else URH_TRY(join_tail(ctx));
    URH_TRY(ctx->arena.reserve(digitize_scratch_bytes(pl, out->cap_rows, ask, true) + (out->qad ? 0 : align256((size_t)n * 4))));
    ctx->arena.reset();
    int64_t *d_n_rows = ctx->d_counts + 10;
    const bool want_bits = out->bits && out->msg_off && out->pauses && out->pos_off;
    BitsParams tile_bp = bits_params(p);
    tile_bp.d_rows_needed = ctx->d_counts + 8;
    TileTailMem tile;
    tile.mem = nullptr;
    if (!fused) {
        // no fused kernel: demodulate (zeros for n <= 2, Costas loop for PSK), then segment the qad
        float *qad = out->qad;
        if (!qad) { qad = (float *)ctx->arena.take((size_t)n * 4); if (!qad) return URHGPU_ERR_ARG; }
        URH_TRY(urhgpu_afp_demod_dev(ctx, d_iq, n, p, qad));
        URH_TRY(digitize(ctx, false, qad, n, p, nullptr, out->rows, out->cap_rows, d_n_rows, ctx->d_counts + 8,
                         ctx->d_counts + 9, pl, 0, nullptr, want_bits ? &tile_bp : nullptr, want_bits ? &tile : nullptr));
    } else {
        URH_TRY(digitize(ctx, true, d_iq, n, p, out->qad, out->rows, out->cap_rows, d_n_rows, ctx->d_counts + 8,
                         ctx->d_counts + 9, pl, 0, piped ? ctx->tail_stream : nullptr, want_bits ? &tile_bp : nullptr,
                         want_bits ? &tile : nullptr));
    }
    int st = URHGPU_OK;
    if (want_bits) {                                                         // else: pulse table only
        const int64_t cap = std::max<int64_t>(out->cap_rows, 1);
        void *scratch = ctx->arena.take(bits_scratch_bytes(cap));
        if (!scratch) return URHGPU_ERR_ARG;
        hipStream_t caller = ctx->stream;
        if (piped) ctx->stream = ctx->tail_stream;
        if (tile.mem) {
            BitsOut bo{out->bits, out->cap_bits, out->msg_off, out->pauses, out->cap_msg, out->pos, out->cap_pos, out->pos_off, out->counts, out->h_counts};
            ScanState ss;
            st = scan_state(ctx, tile_desc_cap(cap, pl.n_chunks), &ss);
            if (st == URHGPU_OK) st = launch_tile_bits(tile, out->rows, d_n_rows, cap, tile_bp, bo, scratch, ss, ctx->stream);
            if (st == URHGPU_OK && hipGetLastError() != hipSuccess) st = URHGPU_ERR_HIP;
        } else {
            st = ppseq_to_bits_inner(ctx, out->rows, d_n_rows, cap, p, out, scratch, ctx->d_counts + 8);
        }
        if (st == URHGPU_OK && out->blob) st = launch_pack_blob(out, p->write_bit_sample_pos, ctx->stream);     // compact mirror (compact.hip)
        ctx->stream = caller;
    } else if (out->blob) {
        st = URHGPU_ERR_ARG;                               // the blob mirrors the bit outputs: all of them must be given
    }
    if (piped) URH_TRY(end_pipelined_pass(ctx));
    return st;
}

// The results of a pass that is over (out: the descriptor the pass was given, with out->blob / cap_blob naming a device blob), on the
// host: pack kernel, the header, then ONE copy of header.total_bytes -- what the boundary objects (Signal.bits(), BitsResult.ppseq() ...)
// fetch instead of the wide int64 tables.  host_dst: pinned memory (hipHostMalloc / torch pin_memory) for an asynchronous copy at PCIe
// speed; pageable memory works (the runtime stages it).  Synchronous.
int urhgpu_outputs_to_host(urhgpu_ctx *ctx, const urhgpu_outputs *out, int write_pos, void *host_dst, int64_t cap_dst, int64_t *total_bytes) {
    if (!ctx || !out || !out->blob || !host_dst || !total_bytes || cap_dst < URHGPU_BLOB_HEADER_BYTES) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(join_tail(ctx));
    URH_TRY(launch_pack_blob(out, write_pos, ctx->stream));
    URH_HIP(hipGetLastError());
    int64_t *hdr = ctx->h_small ? (int64_t *)ctx->h_small : (int64_t *)host_dst;
    URH_HIP(hipMemcpyAsync(hdr, out->blob, URHGPU_BLOB_HEADER_BYTES, hipMemcpyDeviceToHost, ctx->stream));
    URH_HIP(wait_stream(ctx, ctx->stream));
    if (hdr[0] != URHGPU_BLOB_MAGIC) return URHGPU_ERR_ARG;
    const int64_t total = hdr[6] < 0 ? -hdr[6] : hdr[6];
    *total_bytes = total;
    if (hdr[6] < 0 || total > cap_dst) return URHGPU_ERR_CAPACITY;
    URH_HIP(hipMemcpyAsync(host_dst, out->blob, (size_t)total, hipMemcpyDeviceToHost, ctx->stream));
    URH_HIP(wait_stream(ctx, ctx->stream));
    return URHGPU_OK;
}

int64_t urhgpu_blob_capacity(int64_t cap_rows, int64_t cap_bits, int64_t cap_msg, int64_t cap_pos, int has_pos) {
    if (cap_rows < 0 || cap_bits < 0 || cap_msg < 0 || cap_pos < 0) return 0;
    return urh::blob_capacity(cap_rows, cap_bits, cap_msg, cap_pos, has_pos ? 1 : 0);
}

// ---- sharded captures (one rank's phases; the all-gathers in between belong to the caller) --------------
// validation + scratch + kernel arguments of a shard pass; launches the chunks selected by `part` on the hot stream
static int shard_launch(urhgpu_ctx *ctx, const void *d_iq, int64_t n_local, int64_t pos_base, int64_t n_total, int rank, int world,
                        const void *d_left_halo, const urhgpu_params *p, const urhgpu_outputs *out, int part) {
    if (!ctx || !p || !out || !d_iq || !out->rows || !out->counts) return URHGPU_ERR_ARG;
    if (world < 1 || world > kMaxWorld || rank < 0 || rank >= world || n_local < 2 || pos_base < 0 || pos_base + n_local > n_total)
        return URHGPU_ERR_ARG;
    if (rank == 0 && pos_base != 0) return URHGPU_ERR_ARG;
    if (dtype_bytes(p->dtype) == 0) return URHGPU_ERR_DTYPE;
    URH_TRY(check_params(p, true));
    if (p->mod == URHGPU_MOD_PSK) return URHGPU_ERR_UNSUPPORTED;      // the Costas loop does not shard
    if (((uintptr_t)d_iq & 15) || (out->qad && ((uintptr_t)out->qad & 7))) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    ShardSession *ss = session(ctx);
    if (!ss) return URHGPU_ERR_ARG;
    ss->piped = ctx->pipelined;
    if (ss->piped) URH_TRY(begin_pipelined_pass(ctx)); else URH_TRY(join_tail(ctx));
    // ASK passes (generic tail: a dozen under-occupied launches and one more exchange) keep the caller's stream for the hot kernel and
    // 33 KiB of LDS padding per hot workgroup; everything else runs as on a single GPU: tile tail, CU-masked hot stream for float32
    // captures (measured on a 1-rank RCCL group, round 3: the generic tail ran 0.35-0.36 ms per pass with either, 0.40 with both)
    const bool ask = (p->mod == URHGPU_MOD_ASK);
    ss->use_tile = !ask && g_tile_tail;
    hipStream_t s = ctx->stream;
    if (ss->piped && ss->use_tile && masked_hot_stream(p)) URH_TRY(hot_stream_begin(ctx, &s));
    ss->phase = 0; ss->rank = rank; ss->world = world; ss->n_local = n_local; ss->pos_base = pos_base; ss->n_total = n_total;
    ss->p = *p; ss->out = *out;
    const Plan pl = make_plan(ctx, n_local, p->tolerance);
    ss->pl = pl;
    URH_TRY(ctx->arena.reserve(digitize_scratch_bytes(pl, out->cap_rows, ask, true)));
    ctx->arena.reset();
    const int64_t n_table = pl.n_chunks + world - 1;
    ss->table = (ChunkInfo *)ctx->arena.take((size_t)n_table * sizeof(ChunkInfo));
    ss->slab = (uint64_t *)ctx->arena.take((size_t)pl.n_chunks * pl.slab_stride * 8);
    ss->rs_mem = ctx->arena.take(resolve_scratch_bytes(n_table));
    ss->aux = (ResolveAux *)(ctx->d_tickets + 4);
    ss->d_small = (int64_t *)ctx->arena.take(8 * 8);
    ss->bits_scratch = ctx->arena.take(bits_scratch_bytes(std::max<int64_t>(out->cap_rows, 1)));
    ss->rows_stage = out->rows; ss->merge_scratch = nullptr; ss->d_n_stage = ss->d_small + 3;
    if (ask) {
        ss->rows_stage = (int64_t *)ctx->arena.take((size_t)out->cap_rows * 16);
        ss->merge_scratch = ctx->arena.take(merge_scratch_bytes(out->cap_rows));
        ss->d_n_stage = (int64_t *)ctx->arena.take(64);
    }
    if (!ss->table || !ss->slab || !ss->rs_mem || !ss->d_small || !ss->bits_scratch || !ss->rows_stage ||
        !ss->d_n_stage || (ask && !ss->merge_scratch))
        return URHGPU_ERR_ARG;
    if (ss->use_tile) {
        URH_TRY(tile_tail_mem(ctx, n_table, true, &ss->tile));
        if (world > 1) ss->tile.d_row_base = ss->d_small + 4;
    }
    RunArgs &a = ss->run;
    memset(&a, 0, sizeof(a));
    URH_TRY(fill_thresholds(a, p));
    a.in = d_iq; a.qad = out->qad; a.left_halo = d_left_halo; a.n = n_local; a.pos_base = pos_base;
    a.chunk_len = pl.chunk_len; a.slab_stride = pl.slab_stride;
    a.noise_sqrd = p->noise_threshold * p->noise_threshold;
    a.noise_val = noise_for(p);
    a.tol = p->tolerance;
    a.lds_pad = !ctx->pipelined ? 0 : (ss->use_tile ? ctx->hot_lds_pad : ctx->hot_lds_pad_sharded);
    a.wide_int = ctx->tune_wide_int ? 1 : 0;
    URH_TRY(max_magnitude_for(p->dtype, &a.max_magnitude));
    a.chunks = ss->table + rank;               // this rank's chunks sit at table[rank .. rank + n_chunks)
    a.slab = ss->slab;
    a.launch_part = part;
    if (part == 1 && rank > 0 && !a.left_halo) a.left_halo = d_iq;    // any non-null value: only chunk 0 reads the halo
    const bool prof = prof_begin_record(ctx, s);
    hipEvent_t hot_done = nullptr;                 // pipelined: what the tail stream waits for (see digitize)
    if (ss->piped && n_local % kTile == 0) {
        if (!prof) { g_hot_events = HotEvents(); g_hot_events.stop = ctx->ev_hot; }
        hot_done = g_hot_events.stop;
    }
    {
        const int st = launch_demod_runs_iq(a, p->dtype, p->mod, out->qad != nullptr, s);
        if (st != URHGPU_OK) { g_hot_events = HotEvents(); return st; }
    }
    if (hot_done && !g_hot_events.used) hot_done = nullptr;
    if (prof) URH_TRY(prof_end_record(ctx, s));
    else g_hot_events = HotEvents();
    if (ss->piped) {
        // Everything after this launch goes to the tail stream -- also the first chunk of a prelaunched pass, which waits for the
        // halo exchange: the caller's stream never waits for a collective (the exchanges of one communicator run in issue order,
        // so the halo of pass i + 1 queues behind the last exchange of pass i's tail).
        if (!hot_done) { URH_HIP(hipEventRecord(ctx->ev_hot, s)); hot_done = ctx->ev_hot; }
        URH_HIP(hipStreamWaitEvent(ctx->tail_stream, hot_done, 0));
        if (s != ctx->stream && ctx->stream != nullptr) URH_HIP(hipStreamWaitEvent(ctx->stream, hot_done, 0));   // (see digitize)
    }
    URH_HIP(hipGetLastError());
    return URHGPU_OK;
}

int urhgpu_shard_prelaunch_dev(urhgpu_ctx *ctx, const void *d_iq, int64_t n_local, int64_t pos_base, int64_t n_total,
                               int rank, int world, const urhgpu_params *p, const urhgpu_outputs *out) {
    URH_TRY(shard_launch(ctx, d_iq, n_local, pos_base, n_total, rank, world, nullptr, p, out, rank > 0 ? 1 : 0));
    ((ShardSession *)ctx->shard)->phase = -1;
    return URHGPU_OK;
}

int urhgpu_shard_launch_dev(urhgpu_ctx *ctx, const void *d_iq, int64_t n_local, int64_t pos_base, int64_t n_total,
                            int rank, int world, const void *d_left_halo, const urhgpu_params *p, const urhgpu_outputs *out) {
    if ((rank == 0) != (d_left_halo == nullptr)) return URHGPU_ERR_ARG;
    URH_TRY(shard_launch(ctx, d_iq, n_local, pos_base, n_total, rank, world, d_left_halo, p, out, 0));
    ((ShardSession *)ctx->shard)->phase = -2;
    return URHGPU_OK;
}

int urhgpu_shard_runs_dev(urhgpu_ctx *ctx, const void *d_iq, int64_t n_local, int64_t pos_base, int64_t n_total,
                          int rank, int world, const void *d_left_halo, const urhgpu_params *p,
                          const urhgpu_outputs *out, void *d_summary) {
    if (!ctx || !d_summary) return URHGPU_ERR_ARG;
    if ((rank == 0) != (d_left_halo == nullptr)) return URHGPU_ERR_ARG;
    ShardSession *ss = (ShardSession *)ctx->shard;
    hipStream_t s;
    if (ss && (ss->phase == -1 || ss->phase == -2)) {
        // prelaunched: only the first chunk (it needs the halo) is still missing, or (-2) nothing
        if (ss->rank != rank || ss->world != world || ss->n_local != n_local || ss->run.in != d_iq) return URHGPU_ERR_ARG;
        URH_HIP(hipSetDevice(ctx->device));
        s = ss->piped ? ctx->tail_stream : ctx->stream;
        if (rank > 0 && ss->phase == -1) {
            RunArgs a = ss->run;
            a.left_halo = d_left_halo; a.launch_part = 2;
            URH_TRY(launch_demod_runs_iq(a, p->dtype, p->mod, out->qad != nullptr, s));
        }
    } else {
        URH_TRY(shard_launch(ctx, d_iq, n_local, pos_base, n_total, rank, world, d_left_halo, p, out, 0));
        ss = (ShardSession *)ctx->shard;
        s = ss->piped ? ctx->tail_stream : ctx->stream;      // pipelined: shard_launch made the tail stream wait for the hot kernel
    }
    // local resolve pass: the shard on its own -> its summary
    const Plan &pl = ss->pl;
    const int64_t n_table = pl.n_chunks + world - 1;
    ResolveArgs r;
    memset(&r, 0, sizeof(r));
    r.sc = resolve_scratch_carve(ss->rs_mem, n_table);
    r.chunks = ss->table + rank; r.n_chunks = pl.n_chunks; r.n_total = n_local; r.tol = ss->p.tolerance;
    r.rows = nullptr; r.cap_rows = 0; r.d_n_acc = ctx->d_counts + 9; r.d_n_rows = ctx->d_counts + 10;
    r.d_n_rows_needed = ctx->d_counts + 8; r.write_last_row = 0;
    r.local_pass = 1; r.aux = ss->aux; r.summary_out = (ChunkInfo *)d_summary; r.chunk_first = 0; r.n_local = pl.n_chunks;
    // one launch (k_shard_summary) since round 6; the three generic resolve launches stay as tuning key shard_summary_generic (tests compare the two)
    if (ctx->tune_shard_summary_generic) URH_TRY(launch_resolve(r, ctx->d_tickets, s));
    else URH_TRY(launch_shard_summary(r, ctx->d_tickets + 1, s));
    URH_HIP(hipGetLastError());
    ss->phase = 1;
    return URHGPU_OK;
}

int urhgpu_shard_rows_dev(urhgpu_ctx *ctx, const void *d_summaries, int64_t *d_merge) {
    if (!ctx || !d_summaries) return URHGPU_ERR_ARG;
    ShardSession *ss = (ShardSession *)ctx->shard;
    if (!ss || ss->phase != 1) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ss->piped ? ctx->tail_stream : ctx->stream;
    const int rank = ss->rank, world = ss->world;
    const Plan &pl = ss->pl;
    const bool ask = (ss->p.mod == URHGPU_MOD_ASK);
    if (ask && !d_merge) return URHGPU_ERR_ARG;
    const int64_t n_table = pl.n_chunks + world - 1;
    const ChunkInfo *S = (const ChunkInfo *)d_summaries;
    // table = [S_0 .. S_{rank-1} | local chunks | S_{rank+1} .. S_{world-1}]
    if (rank > 0) URH_HIP(hipMemcpyAsync(ss->table, S, (size_t)rank * sizeof(ChunkInfo), hipMemcpyDeviceToDevice, s));
    if (rank + 1 < world)
        URH_HIP(hipMemcpyAsync(ss->table + rank + pl.n_chunks, S + rank + 1, (size_t)(world - 1 - rank) * sizeof(ChunkInfo),
                               hipMemcpyDeviceToDevice, s));
    ResolveArgs r;
    memset(&r, 0, sizeof(r));
    r.sc = resolve_scratch_carve(ss->rs_mem, n_table);
    r.chunks = ss->table; r.n_chunks = n_table; r.n_total = ss->n_total; r.tol = ss->p.tolerance;
    r.rows = ss->rows_stage; r.cap_rows = ss->out.cap_rows; r.d_n_acc = ctx->d_counts + 9; r.d_n_rows = ss->d_n_stage;
    r.d_n_rows_needed = ctx->d_counts + 8; r.write_last_row = (rank == world - 1) ? 1 : 0;
    r.local_pass = 0; r.aux = ss->aux; r.summary_out = nullptr; r.chunk_first = rank; r.n_local = pl.n_chunks;
    r.d_ts_carry = ss->d_small;
    URH_HIP(hipMemsetAsync(ss->d_small, 0, 8 * 8, s));
    EmitArgs e;
    e.sc = r.sc;
    e.chunks = ss->table; e.chunk_first = rank; e.slab = ss->slab; e.slab_stride = pl.slab_stride;
    e.rows = ss->rows_stage; e.cap_rows = ss->out.cap_rows; e.d_ts_carry = ss->d_small; e.is_ask = ask ? 1 : 0;
    e.sps = ss->p.samples_per_symbol;
    if (ss->use_tile) {
        // resolve + rows + per-tile bit aggregates over the table: two launches (total_samples before my first row comes out of the
        // tile scan, the summaries being tiles of their own: no ts_carry)
        BitsParams bp = bits_params(&ss->p);
        bp.d_row_base = ss->tile.d_row_base;
        URH_TRY(launch_tile_rows(r, e, ss->tile, &bp, s));
        ss->d_row_base = ss->tile.d_row_base;
    } else {
        URH_TRY(launch_resolve(r, ctx->d_tickets, s));
        URH_TRY(launch_emit_rows(e, pl.n_chunks, s));
        ss->d_row_base = r.sc.out_off + rank;
    }
    if (ask) {
        URH_TRY(launch_merge_rows_ask(ss->rows_stage, ss->d_n_stage, ss->out.cap_rows, ss->out.rows, ss->out.cap_rows,
                                      ss->d_small + 3, ss->merge_scratch, ctx->d_tickets, s));
        launch_merge_summary(ss->out.rows, ss->d_small + 3, d_merge, s);
    }
    URH_HIP(hipGetLastError());
    ss->phase = 2;
    return URHGPU_OK;
}

int urhgpu_shard_bits_prepare_dev(urhgpu_ctx *ctx, const int64_t *d_merge_all, int64_t *d_flags) {
    if (!ctx || !d_flags) return URHGPU_ERR_ARG;
    ShardSession *ss = (ShardSession *)ctx->shard;
    if (!ss || ss->phase != 2) return URHGPU_ERR_ARG;
    const bool ask = (ss->p.mod == URHGPU_MOD_ASK);
    if (ask && !d_merge_all) return URHGPU_ERR_ARG;
    if (!ss->out.bits || !ss->out.msg_off || !ss->out.pauses || !ss->out.pos_off) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ss->piped ? ctx->tail_stream : ctx->stream;
    int64_t *d_n_rows = ss->d_small + 3;
    if (ask) launch_merge_fix(ss->out.rows, d_n_rows, d_merge_all, ss->rank, ss->world, ss->d_small + 1, s);
    BitsParams bp = bits_params(&ss->p);
    bp.d_row_base = ss->d_row_base; bp.d_ts_carry = ss->d_small; bp.d_absorbed = ask ? ss->d_small + 1 : nullptr;
    bp.d_extra = (const int32_t *)(ss->d_small + 2); bp.is_last_rank = (ss->rank == ss->world - 1) ? 1 : 0;
    ScanState sst;
    const int64_t cap = std::max<int64_t>(ss->out.cap_rows, 1);
    if (ss->use_tile) {
        bp.d_ts_carry = nullptr;
        URH_TRY(scan_state(ctx, tile_desc_cap(cap, ss->tile.n_chunks), &sst));
        URH_TRY(launch_tile_bits_prepare(ss->tile, ss->out.rows, d_n_rows, cap, bp, ss->bits_scratch, d_flags, sst, s));
    } else {
        URH_TRY(scan_state(ctx, cap, &sst));
        URH_TRY(launch_bits_prepare(ss->out.rows, d_n_rows, cap, bp, ss->bits_scratch, d_flags, sst, s));
    }
    URH_HIP(hipGetLastError());
    ss->phase = 3;
    return URHGPU_OK;
}

int urhgpu_shard_bits_finish_dev(urhgpu_ctx *ctx, const int64_t *d_flags_all) {
    if (!ctx || !d_flags_all) return URHGPU_ERR_ARG;
    ShardSession *ss = (ShardSession *)ctx->shard;
    if (!ss || ss->phase != 3) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ss->piped ? ctx->tail_stream : ctx->stream;
    const bool ask = (ss->p.mod == URHGPU_MOD_ASK);
    launch_bits_extra(d_flags_all, ss->rank, ss->world, (int32_t *)(ss->d_small + 2), s);
    BitsParams bp = bits_params(&ss->p);
    bp.d_row_base = ss->d_row_base; bp.d_ts_carry = ss->d_small; bp.d_absorbed = ask ? ss->d_small + 1 : nullptr;
    bp.d_extra = (const int32_t *)(ss->d_small + 2); bp.is_last_rank = (ss->rank == ss->world - 1) ? 1 : 0;
    bp.d_rows_needed = ctx->d_counts + 8;
    const urhgpu_outputs &o = ss->out;
    BitsOut bo{o.bits, o.cap_bits, o.msg_off, o.pauses, o.cap_msg, o.pos, o.cap_pos, o.pos_off, o.counts};
    ScanState sst;
    const int64_t cap = std::max<int64_t>(o.cap_rows, 1);
    if (ss->use_tile) {
        bp.d_ts_carry = nullptr;
        URH_TRY(scan_state(ctx, tile_desc_cap(cap, ss->tile.n_chunks), &sst));
        URH_TRY(launch_tile_bits_finish(ss->tile, o.rows, ss->d_small + 3, cap, bp, bo, ss->bits_scratch, sst, s));
    } else {
        URH_TRY(scan_state(ctx, cap, &sst));
        URH_TRY(launch_bits_finish(o.rows, ss->d_small + 3, cap, bp, bo, ss->bits_scratch, sst, s));
    }
    if (o.blob) {
        // the compact mirror of this rank's piece (compact.hip); an absorbed first row (ASK) is shipped as state -128
        URH_TRY(launch_pack_blob(&o, ss->p.write_bit_sample_pos, s));
    }
    URH_HIP(hipGetLastError());
    ss->phase = 0;
    if (ss->piped) URH_TRY(end_pipelined_pass(ctx));
    return URHGPU_OK;
}

// ---- estimator passes (device pointers; see include/urhgpu.h) ---------------------------------------------------
namespace {
__global__ void k_set_i64(int64_t *p, int64_t v) { *p = v; }
}

static int segment_runs_impl(urhgpu_ctx *ctx, const void *d_iq, int dtype, int64_t n, float noise_threshold,
                             int64_t *d_rows, int64_t cap_rows, int64_t *d_n_rows, float *d_qad_ask) {
    if (!ctx || n < 0 || !d_n_rows || cap_rows < 0) return URHGPU_ERR_ARG;
    if (dtype_bytes(dtype) == 0) return URHGPU_ERR_DTYPE;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(join_tail(ctx));
    if (n == 0) { URH_HIP(hipMemsetAsync(d_n_rows, 0, 8, ctx->stream)); return URHGPU_OK; }
    if (!d_iq || !d_rows || ((uintptr_t)d_iq & 15)) return URHGPU_ERR_ARG;
    urhgpu_params p;
    memset(&p, 0, sizeof(p));
    p.dtype = dtype; p.mod = URHGPU_MOD_ASK; p.bits_per_symbol = 1; p.center = noise_threshold; p.center_spacing = 0.f;
    p.tolerance = 9;                                   // outlier_tolerance = 10 consecutive samples (auto_interpretation.pyx:72)
    p.samples_per_symbol = 1;
    const Plan pl = make_plan(ctx, n, p.tolerance);
    URH_TRY(ctx->arena.reserve(digitize_scratch_bytes(pl, cap_rows, false, false)));
    ctx->arena.reset();
    return digitize(ctx, true, d_iq, n, &p, d_qad_ask, d_rows, cap_rows, d_n_rows, ctx->d_counts + 8, ctx->d_counts + 9, pl, 1);
}

int urhgpu_segment_runs_dev(urhgpu_ctx *ctx, const void *d_iq, int dtype, int64_t n, float noise_threshold,
                            int64_t *d_rows, int64_t cap_rows, int64_t *d_n_rows) {
    return segment_runs_impl(ctx, d_iq, dtype, n, noise_threshold, d_rows, cap_rows, d_n_rows, nullptr);
}

static int message_ranges_impl(urhgpu_ctx *ctx, const void *d_iq, int dtype, int64_t n, float noise_threshold, int64_t *seg_out, int64_t cap_seg_out,
                               int64_t *n_seg_out, int64_t *merged_out, int64_t cap_merged_out, int64_t *n_merged_out, int *merge_ambiguous,
                               float *d_qad_ask);

int urhgpu_message_ranges_dev(urhgpu_ctx *ctx, const void *d_iq, int dtype, int64_t n, float noise_threshold, int64_t *seg_out, int64_t cap_seg_out,
                              int64_t *n_seg_out, int64_t *merged_out, int64_t cap_merged_out, int64_t *n_merged_out, int *merge_ambiguous) {
    return message_ranges_impl(ctx, d_iq, dtype, n, noise_threshold, seg_out, cap_seg_out, n_seg_out, merged_out, cap_merged_out, n_merged_out,
                               merge_ambiguous, nullptr);
}

int urhgpu_message_ranges_demod_dev(urhgpu_ctx *ctx, const void *d_iq, int dtype, int64_t n, float noise_threshold, int64_t *seg_out,
                                    int64_t cap_seg_out, int64_t *n_seg_out, int64_t *merged_out, int64_t cap_merged_out, int64_t *n_merged_out,
                                    int *merge_ambiguous, float *d_qad_ask) {
    if (!d_qad_ask || ((uintptr_t)d_qad_ask & 7)) return URHGPU_ERR_ARG;
    if (dtype != URHGPU_DT_F32) return URHGPU_ERR_UNSUPPORTED;
    if (n > 0 && noise_threshold != noise_threshold) return URHGPU_ERR_UNSUPPORTED;     // (no segmentation pass runs for a NaN threshold)
    return message_ranges_impl(ctx, d_iq, dtype, n, noise_threshold, seg_out, cap_seg_out, n_seg_out, merged_out, cap_merged_out, n_merged_out,
                               merge_ambiguous, d_qad_ask);
}

static int message_ranges_impl(urhgpu_ctx *ctx, const void *d_iq, int dtype, int64_t n, float noise_threshold, int64_t *seg_out, int64_t cap_seg_out,
                               int64_t *n_seg_out, int64_t *merged_out, int64_t cap_merged_out, int64_t *n_merged_out, int *merge_ambiguous,
                               float *d_qad_ask) {
    if (!ctx || n < 0 || !n_seg_out || cap_seg_out < 0 || cap_merged_out < 0 || (cap_seg_out > 0 && !seg_out) || (cap_merged_out > 0 && !merged_out))
        return URHGPU_ERR_ARG;
    if (dtype_bytes(dtype) == 0) return URHGPU_ERR_DTYPE;
    const bool merge = n_merged_out != nullptr;
    *n_seg_out = 0;
    if (merge) *n_merged_out = 0;
    if (merge_ambiguous) *merge_ambiguous = 0;
    if (n == 0 || noise_threshold != noise_threshold) return URHGPU_OK;      // nothing compares greater than NaN: never above the noise
    URH_HIP(hipSetDevice(ctx->device));
    // the state table, the segment table and the message table live in the staging arena for the duration of the call
    const int64_t cap_rows = n / 10 + 2;                   // a state change needs 10 samples in the new state
    const int64_t cap_seg = cap_rows / 2 + 2;
    const size_t need = (size_t)cap_rows * 16 + 2 * (size_t)cap_seg * 16 + seg_scratch_bytes(cap_rows, cap_seg) + seg_ctl_bytes() + 16 * 256;
    URH_TRY(ctx->staging.reserve(need));
    ctx->staging.reset();
    int64_t *d_rows = (int64_t *)ctx->staging.take((size_t)cap_rows * 16);
    // (the control block directly in front of the segment table: the block and the first segments leave in ONE copy)
    const size_t ctl_pad = (seg_ctl_bytes() + 255) & ~size_t(255);
    char *d_ctl_seg = (char *)ctx->staging.take(ctl_pad + (size_t)cap_seg * 16);
    SegCtl *d_ctl = (SegCtl *)d_ctl_seg;
    int64_t *d_seg = d_ctl_seg ? (int64_t *)(d_ctl_seg + ctl_pad) : nullptr;
    int64_t *d_msgs = (int64_t *)ctx->staging.take((size_t)cap_seg * 16);
    void *scratch = ctx->staging.take(seg_scratch_bytes(cap_rows, cap_seg));
    int64_t *d_n_rows = (int64_t *)ctx->staging.take(64);
    if (!d_rows || !d_seg || !d_msgs || !scratch || !d_ctl || !d_n_rows) return URHGPU_ERR_ARG;
    URH_TRY(segment_runs_impl(ctx, d_iq, dtype, n, noise_threshold, d_rows, cap_rows, d_n_rows, d_qad_ask));
    if (d_qad_ask && n <= 2) URH_HIP(hipMemsetAsync(d_qad_ask, 0, (size_t)n * 4, ctx->stream));     // afp_demod of up to two samples: zeros (signal_functions.pyx:335-336)
    URH_TRY(launch_message_ranges(d_rows, d_n_rows, cap_rows, d_iq, dtype, n, noise_threshold, merge ? 1 : 0, d_seg, d_msgs, cap_seg, d_ctl, scratch,
                                  ctx->stream));
    URH_HIP(hipGetLastError());
    // one round trip for the usual case: the control block together with the first segments / merged messages the caller has room for
    // (the counts are not known yet: a prefix of each table is copied speculatively, the rest -- rarely -- afterwards)
    std::vector<char> ctl(seg_ctl_bytes());
    const int64_t spec_seg = std::min<int64_t>(std::min<int64_t>(cap_seg_out, cap_seg), 4096);
    const int64_t spec_mrg = merge ? std::min<int64_t>(std::min<int64_t>(cap_merged_out, cap_seg), 4096) : 0;     // (config 3's capture has 1500 messages: beyond the prefix costs a second round trip)
    std::vector<int64_t> spec_m((size_t)spec_mrg * 2);
    // (through the context's pinned landing zone when it fits: three copies to pageable memory are three synchronous round trips)
    const bool pinned = ctx->h_small && ctl_pad + (size_t)(spec_seg + spec_mrg) * 16 <= kSmallPinned;
    char *l_ctl = pinned ? ctx->h_small : ctl.data();
    int64_t *l_seg = pinned ? (int64_t *)(ctx->h_small + ctl_pad) : seg_out;
    int64_t *l_mrg = pinned ? l_seg + 2 * spec_seg : spec_m.data();
    if (pinned) URH_HIP(hipMemcpyAsync(l_ctl, d_ctl, ctl_pad + (size_t)spec_seg * 16, hipMemcpyDeviceToHost, ctx->stream));
    else {
        URH_HIP(hipMemcpyAsync(l_ctl, d_ctl, ctl.size(), hipMemcpyDeviceToHost, ctx->stream));
        if (spec_seg > 0) URH_HIP(hipMemcpyAsync(l_seg, d_seg, (size_t)spec_seg * 16, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (spec_mrg > 0) URH_HIP(hipMemcpyAsync(l_mrg, d_msgs, (size_t)spec_mrg * 16, hipMemcpyDeviceToHost, ctx->stream));
    URH_HIP(wait_stream(ctx, ctx->stream));
    if (pinned) {
        memcpy(ctl.data(), l_ctl, ctl.size());
        if (spec_mrg > 0) memcpy(spec_m.data(), l_mrg, (size_t)spec_mrg * 16);
    }
    int64_t n_seg = 0, n_msgs = 0;
    int ambiguous = 0;
    seg_ctl_read(ctl.data(), &n_seg, &n_msgs, &ambiguous);
    if (n_seg > cap_seg) return URHGPU_ERR_CAPACITY;       // cannot happen (a segment needs two state changes)
    *n_seg_out = n_seg;
    const int64_t take = std::min(n_seg, cap_seg_out);
    if (pinned && std::min(take, spec_seg) > 0) memcpy(seg_out, l_seg, (size_t)std::min(take, spec_seg) * 16);
    bool more = false;
    if (take > spec_seg) { URH_HIP(hipMemcpyAsync(seg_out + 2 * spec_seg, d_seg + 2 * spec_seg, (size_t)(take - spec_seg) * 16, hipMemcpyDeviceToHost, ctx->stream)); more = true; }
    if (merge) {
        const bool merged = n_seg > 1;                     // AutoInterpretation.py:108: one segment is returned as it is
        *n_merged_out = merged ? n_msgs : n_seg;
        if (merge_ambiguous) *merge_ambiguous = merged ? ambiguous : 0;
        const int64_t take_m = std::min(*n_merged_out, cap_merged_out);
        if (!merged) {
            // the single segment is the message: it is in seg_out already when the caller has room for a segment, else fetch it
            if (take_m > 0) {
                if (take >= 1) { merged_out[0] = seg_out[0]; merged_out[1] = seg_out[1]; }
                else { URH_HIP(hipMemcpyAsync(merged_out, d_seg, 16, hipMemcpyDeviceToHost, ctx->stream)); more = true; }
            }
        } else {
            const int64_t have = std::min(take_m, spec_mrg);
            if (have > 0) memcpy(merged_out, spec_m.data(), (size_t)have * 16);
            if (take_m > have) { URH_HIP(hipMemcpyAsync(merged_out + 2 * have, d_msgs + 2 * have, (size_t)(take_m - have) * 16, hipMemcpyDeviceToHost, ctx->stream)); more = true; }
        }
    }
    if (more) URH_HIP(wait_stream(ctx, ctx->stream));
    return URHGPU_OK;
}

int urhgpu_compact_gt_dev(urhgpu_ctx *ctx, const float *d_x, int64_t n, float thr, float *d_out, int64_t *d_count) {
    if (!ctx || n < 0 || !d_count || (n > 0 && (!d_x || !d_out))) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(join_tail(ctx));
    URH_TRY(ctx->arena.reserve(compact_scratch_bytes(n) + 1024));
    ctx->arena.reset();
    void *scratch = ctx->arena.take(compact_scratch_bytes(n));
    if (!scratch) return URHGPU_ERR_ARG;
    hipLaunchKernelGGL(k_set_i64, dim3(1), dim3(1), 0, ctx->stream, ctx->d_counts + 11, n);
    URH_TRY(launch_compact_gt(d_x, n, ctx->d_counts + 11, thr, d_out, d_count, scratch, ctx->d_tickets, ctx->stream));
    URH_HIP(hipGetLastError());
    return URHGPU_OK;
}

int urhgpu_edges_le_dev(urhgpu_ctx *ctx, const float *d_x, int64_t n, float center, int64_t *d_idx, int64_t cap, int64_t *d_count) {
    if (!ctx || n < 0 || cap < 0 || !d_count || (n > 0 && !d_x) || (cap > 0 && !d_idx)) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(join_tail(ctx));
    URH_TRY(ctx->arena.reserve(compact_scratch_bytes(n) + 1024));
    ctx->arena.reset();
    void *scratch = ctx->arena.take(compact_scratch_bytes(n));
    if (!scratch) return URHGPU_ERR_ARG;
    hipLaunchKernelGGL(k_set_i64, dim3(1), dim3(1), 0, ctx->stream, ctx->d_counts + 11, n);
    URH_TRY(launch_compact_edges(d_x, n, ctx->d_counts + 11, center, d_idx, cap, d_count, scratch, ctx->d_tickets, ctx->stream));
    URH_HIP(hipGetLastError());
    return URHGPU_OK;
}

int urhgpu_minmax_f32_dev(urhgpu_ctx *ctx, const float *d_x, int64_t n, float *d_out2) {
    if (!ctx || n <= 0 || !d_x || !d_out2) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(join_tail(ctx));
    URH_TRY(ctx->arena.reserve(minmax_scratch_bytes() + 1024));
    ctx->arena.reset();
    void *scratch = ctx->arena.take(minmax_scratch_bytes());
    if (!scratch) return URHGPU_ERR_ARG;
    URH_TRY(launch_minmax(d_x, n, d_out2, scratch, ctx->stream));
    URH_HIP(hipGetLastError());
    return URHGPU_OK;
}

int urhgpu_pairwise_sum_f32_dev(urhgpu_ctx *ctx, const float *d_x, int64_t n, int mode, float mean, float *sum_out) {
    if (!ctx || n < 0 || !sum_out || (n > 0 && !d_x) || (mode != 0 && mode != 1)) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(join_tail(ctx));
    return pairwise_sum_f32(ctx, d_x, n, mode, mean, sum_out);
}

int urhgpu_histogram_f32_dev(urhgpu_ctx *ctx, const float *d_x, int64_t n, const double *d_edges, int64_t n_edges, int64_t *d_counts) {
    if (!ctx || n < 0 || n_edges < 2 || n_edges > (1 << 30) || !d_edges || !d_counts || (n > 0 && !d_x)) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(join_tail(ctx));
    URH_TRY(launch_hist_edges(d_x, n, d_edges, (int)n_edges, d_counts, ctx->stream));
    URH_HIP(hipGetLastError());
    return URHGPU_OK;
}

int urhgpu_ctx_costas_stats(urhgpu_ctx *ctx, int32_t *out4) {
    if (!ctx || !out4) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    URH_HIP(hipStreamSynchronize(ctx->stream));
    const int32_t *h = (const int32_t *)(ctx->h_counts + 12);
    out4[0] = h[0]; out4[1] = h[1]; out4[2] = h[2]; out4[3] = h[4];
    return URHGPU_OK;
}

int urhgpu_test_fast_division_dev(urhgpu_ctx *ctx, uint64_t seed, int reps, uint64_t *n_mismatch) {
    if (!ctx || !n_mismatch || reps < 0) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    URH_HIP(hipMemsetAsync(ctx->d_counts, 0, 8, ctx->stream));
    launch_test_div(seed, reps, (unsigned long long *)ctx->d_counts, ctx->stream);
    URH_HIP(hipGetLastError());
    URH_HIP(hipMemcpyAsync(ctx->h_counts, ctx->d_counts, 8, hipMemcpyDeviceToHost, ctx->stream));
    URH_HIP(hipStreamSynchronize(ctx->stream));
    *n_mismatch = (uint64_t)ctx->h_counts[0];
    return URHGPU_OK;
}

int urhgpu_test_force_merge_ambiguous(int on) {
    const int was = urh::g_force_merge_ambiguous ? 1 : 0;
    urh::g_force_merge_ambiguous = on != 0;
    return was;
}

int urhgpu_test_sincosf_fast_dev(urhgpu_ctx *ctx, uint64_t *n_mismatch) {
    if (!ctx || !n_mismatch) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    URH_HIP(hipMemsetAsync(ctx->d_counts, 0, 8, ctx->stream));
    launch_test_sincosf_fast((unsigned long long *)ctx->d_counts, ctx->stream);
    URH_HIP(hipGetLastError());
    URH_HIP(hipMemcpyAsync(ctx->h_counts, ctx->d_counts, 8, hipMemcpyDeviceToHost, ctx->stream));
    URH_HIP(hipStreamSynchronize(ctx->stream));
    *n_mismatch = (uint64_t)ctx->h_counts[0];
    return URHGPU_OK;
}

static int stage_in(urhgpu_ctx *ctx, const void *host, size_t bytes, void **dev);

// mod: URHGPU_MOD_ASK / _FSK / _PSK / _OQPSK, or urh::kModGfsk with the Gaussian taps (and optionally the caller's filtered
// frequencies, one float per data sample of every message, back to back)
static int modulate_common(urhgpu_ctx *ctx, const uint8_t *bits, const int64_t *bit_off, const uint32_t *pause, const uint32_t *start,
                           int n_msgs, uint32_t samples_per_symbol, int mod, const float *parameters, int bits_per_symbol,
                           float carrier_amplitude, float carrier_frequency, float carrier_phase, float sample_rate, int dtype,
                           const float *gauss_fir, int n_taps, const float *frequencies,
                           void *d_out, int64_t cap_samples, int64_t *total_samples) {
    const bool gfsk = (mod == urh::kModGfsk);
    if (gfsk && (!gauss_fir || n_taps < 1)) return URHGPU_ERR_ARG;
    if (!ctx || n_msgs < 0 || !total_samples || (n_msgs > 0 && (!bit_off || !pause || !start || !parameters))) return URHGPU_ERR_ARG;
    const bool oqpsk = (mod == URHGPU_MOD_OQPSK);
    if (mod != URHGPU_MOD_ASK && mod != URHGPU_MOD_FSK && mod != URHGPU_MOD_PSK && !oqpsk && !gfsk) return URHGPU_ERR_UNSUPPORTED;
    if (oqpsk && bits_per_symbol != 2) return URHGPU_ERR_ARG;                    // assert bits_per_symbol == 2 (:120)
    if (dtype != URHGPU_DT_F32 && dtype != URHGPU_DT_I8 && dtype != URHGPU_DT_I16) return URHGPU_ERR_DTYPE;
    if (bits_per_symbol < 1 || bits_per_symbol > 16 || samples_per_symbol == 0 || n_msgs > 65535) return URHGPU_ERR_UNSUPPORTED;
    std::vector<ModMsg> msgs((size_t)n_msgs);
    int64_t total = 0, total_sym = 0, max_samples = 0;
    for (int m = 0; m < n_msgs; ++m) {
        const int64_t nb = bit_off[m + 1] - bit_off[m];
        if (nb < 0) return URHGPU_ERR_ARG;
        // GFSK re-derives bits_per_symbol as len(bits) // num_symbols (:201): no whole symbol -> ZeroDivisionError there;
        // a message shorter than bits_per_symbol symbols can derive a larger value: not supported
        if (gfsk && nb > 0 && (nb / bits_per_symbol == 0 || nb / (nb / bits_per_symbol) != bits_per_symbol))
            return nb / bits_per_symbol == 0 ? URHGPU_ERR_ARG : URHGPU_ERR_UNSUPPORTED;
        ModMsg &g = msgs[(size_t)m];
        g.bit_off = bit_off[m]; g.n_sym = nb / bits_per_symbol; g.sym_off = total_sym; g.out_off = total;
        g.pause = pause[m]; g.start = start[m];
        const int64_t ns = g.n_sym * (int64_t)samples_per_symbol + pause[m];
        total += ns; total_sym += g.n_sym;
        max_samples = std::max(max_samples, ns);
    }
    *total_samples = total;
    if (total > cap_samples) return URHGPU_ERR_CAPACITY;
    if (total == 0) return URHGPU_OK;
    if (!d_out || !bits) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(join_tail(ctx));
    const int64_t n_bits = bit_off[n_msgs];
    const size_t n_par = (size_t)1 << bits_per_symbol;
    URH_TRY(ctx->staging.reserve(align256((size_t)std::max<int64_t>(n_bits, 1)) + align256(msgs.size() * sizeof(ModMsg)) +
                                 align256(n_par * 4) + align256((size_t)std::max<int64_t>(total_sym, 1) * 4) + 2048 +
                                 (gfsk ? align256((size_t)n_taps * 4) + 2 * align256((size_t)std::max<int64_t>(total_sym, 1) * samples_per_symbol * 4) : 0)));
    ctx->staging.reset();
    void *d_bits = nullptr, *d_msgs = nullptr, *d_par = nullptr;
    std::vector<uint8_t> oq;
    if (oqpsk) {
        // get_oqpsk_bits (:179-194) per message: even bits stay, odd bits are delayed by one symbol; of the num_bits + 2
        // bits it returns the symbol loop reads the first 2 * n_sym (total_symbols is taken from the original length)
        oq.assign((size_t)n_bits, 0);
        for (int m = 0; m < n_msgs; ++m) {
            const uint8_t *b = bits + bit_off[m];
            uint8_t *r = oq.data() + bit_off[m];
            const int64_t nb = bit_off[m + 1] - bit_off[m];
            if (nb == 0) continue;
            r[0] = b[0];
            for (int64_t i = 2; i < nb - 2; i += 2) { r[i] = b[i]; r[i + 1] = b[i - 1]; }
        }
        bits = oq.data();
    }
    URH_TRY(stage_in(ctx, bits, (size_t)n_bits, &d_bits));
    URH_TRY(stage_in(ctx, msgs.data(), msgs.size() * sizeof(ModMsg), &d_msgs));
    URH_TRY(stage_in(ctx, parameters, n_par * 4, &d_par));
    float *d_phase = (float *)ctx->staging.take((size_t)std::max<int64_t>(total_sym, 1) * 4);
    if (!d_phase) return URHGPU_ERR_ARG;
    ModArgs a;
    a.bits = (const uint8_t *)d_bits; a.msgs = (const ModMsg *)d_msgs; a.params = (const float *)d_par; a.phase = d_phase;
    a.out = d_out; a.n_msgs = n_msgs; a.mod = oqpsk ? URHGPU_MOD_PSK : mod; a.oqpsk = oqpsk ? 1 : 0; a.dtype = dtype; a.bps = bits_per_symbol; a.sps = samples_per_symbol;
    a.carrier_amplitude = carrier_amplitude; a.carrier_frequency = carrier_frequency; a.carrier_phase = carrier_phase;
    a.sample_rate = sample_rate;
    a.taps = nullptr; a.n_taps = 0; a.freq_given = 0; a.gf_freq = a.gf_phase = nullptr;
    if (gfsk) {
        const size_t n_data = (size_t)total_sym * samples_per_symbol;
        void *d_taps = nullptr;
        URH_TRY(stage_in(ctx, gauss_fir, (size_t)n_taps * 4, &d_taps));
        a.taps = (const float *)d_taps; a.n_taps = n_taps;
        if (frequencies && n_data) {
            void *d_f = nullptr;
            URH_TRY(stage_in(ctx, frequencies, n_data * 4, &d_f));
            a.gf_freq = (float *)d_f; a.freq_given = 1;
        } else {
            a.gf_freq = (float *)ctx->staging.take(std::max<size_t>(n_data, 1) * 4);
        }
        a.gf_phase = (float *)ctx->staging.take(std::max<size_t>(n_data, 1) * 4);
        if (!a.gf_freq || !a.gf_phase) return URHGPU_ERR_ARG;
    }
    URH_TRY(launch_modulate(a, max_samples, ctx->stream));
    URH_HIP(hipGetLastError());
    return URHGPU_OK;
}

int urhgpu_modulate_dev(urhgpu_ctx *ctx, const uint8_t *bits, const int64_t *bit_off, const uint32_t *pause, const uint32_t *start,
                        int n_msgs, uint32_t samples_per_symbol, int mod, const float *parameters, int bits_per_symbol,
                        float carrier_amplitude, float carrier_frequency, float carrier_phase, float sample_rate, int dtype,
                        void *d_out, int64_t cap_samples, int64_t *total_samples) {
    if (mod == urh::kModGfsk) return URHGPU_ERR_UNSUPPORTED;
    return modulate_common(ctx, bits, bit_off, pause, start, n_msgs, samples_per_symbol, mod, parameters, bits_per_symbol, carrier_amplitude,
                           carrier_frequency, carrier_phase, sample_rate, dtype, nullptr, 0, nullptr, d_out, cap_samples, total_samples);
}

int urhgpu_modulate_gfsk_dev(urhgpu_ctx *ctx, const uint8_t *bits, const int64_t *bit_off, const uint32_t *pause, const uint32_t *start,
                             int n_msgs, uint32_t samples_per_symbol, const float *parameters, int bits_per_symbol,
                             float carrier_amplitude, float carrier_phase, float sample_rate, int dtype, const float *gauss_fir,
                             int n_taps, const float *frequencies, void *d_out, int64_t cap_samples, int64_t *total_samples) {
    return modulate_common(ctx, bits, bit_off, pause, start, n_msgs, samples_per_symbol, urh::kModGfsk, parameters, bits_per_symbol,
                           carrier_amplitude, 0.0f, carrier_phase, sample_rate, dtype, gauss_fir, n_taps, frequencies, d_out, cap_samples,
                           total_samples);
}

static int modulate_one(urhgpu_ctx *ctx, const uint8_t *bits, int64_t num_bits, uint32_t samples_per_symbol, int mod,
                        const float *parameters, int bits_per_symbol, float carrier_amplitude, float carrier_frequency,
                        float carrier_phase, float sample_rate, uint32_t pause, uint32_t start, int dtype, const float *gauss_fir,
                        int n_taps, const float *frequencies, void *out) {
    if (!ctx || num_bits < 0 || bits_per_symbol < 1) return URHGPU_ERR_ARG;
    if (dtype != URHGPU_DT_F32 && dtype != URHGPU_DT_I8 && dtype != URHGPU_DT_I16) return URHGPU_ERR_DTYPE;
    const int64_t total = (num_bits / bits_per_symbol) * (int64_t)samples_per_symbol + pause;
    if (total == 0) return URHGPU_OK;
    if (!out) return URHGPU_ERR_ARG;
    const size_t bytes = (size_t)total * 2 * (dtype == URHGPU_DT_F32 ? 4 : (dtype == URHGPU_DT_I8 ? 1 : 2));
    if (num_bits == 0) { memset(out, 0, bytes); return URHGPU_OK; }             // np.zeros, :104-106
    URH_HIP(hipSetDevice(ctx->device));
    void *d_out = nullptr;                                 // not from the staging arena: urhgpu_modulate_dev resets it
    URH_HIP(hipMalloc(&d_out, bytes));
    const int64_t off[2] = {0, num_bits};
    int64_t got = 0;
    int st = modulate_common(ctx, bits, off, &pause, &start, 1, samples_per_symbol, mod, parameters, bits_per_symbol,
                             carrier_amplitude, carrier_frequency, carrier_phase, sample_rate, dtype, gauss_fir, n_taps, frequencies,
                             d_out, total, &got);
    if (st == URHGPU_OK) {
        hipError_t e = hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) st = urh::hip_fail(e, "modulate D2H", __FILE__, __LINE__);
    }
    (void)hipFree(d_out);
    return st;
}

int urhgpu_modulate(urhgpu_ctx *ctx, const uint8_t *bits, int64_t num_bits, uint32_t samples_per_symbol, int mod,
                    const float *parameters, int bits_per_symbol, float carrier_amplitude, float carrier_frequency,
                    float carrier_phase, float sample_rate, uint32_t pause, uint32_t start, int dtype, void *out) {
    if (mod == urh::kModGfsk) return URHGPU_ERR_UNSUPPORTED;
    return modulate_one(ctx, bits, num_bits, samples_per_symbol, mod, parameters, bits_per_symbol, carrier_amplitude, carrier_frequency,
                        carrier_phase, sample_rate, pause, start, dtype, nullptr, 0, nullptr, out);
}

int urhgpu_modulate_gfsk(urhgpu_ctx *ctx, const uint8_t *bits, int64_t num_bits, uint32_t samples_per_symbol, const float *parameters,
                         int bits_per_symbol, float carrier_amplitude, float carrier_phase, float sample_rate, uint32_t pause,
                         uint32_t start, int dtype, const float *gauss_fir, int n_taps, const float *frequencies, void *out) {
    return modulate_one(ctx, bits, num_bits, samples_per_symbol, urh::kModGfsk, parameters, bits_per_symbol, carrier_amplitude, 0.0f,
                        carrier_phase, sample_rate, pause, start, dtype, gauss_fir, n_taps, frequencies, out);
}

int urhgpu_spectrogram_dev(urhgpu_ctx *ctx, const float *d_x, int64_t n, int window_size, int64_t hop, int64_t frames,
                           const double *d_window, const double *d_twiddles, double *d_stft, float *d_db) {
    if (!ctx || !d_x || n < 0 || !d_window || !d_twiddles || ((d_stft != nullptr) == (d_db != nullptr))) return URHGPU_ERR_ARG;
    if (((uintptr_t)d_x & 7) || ((uintptr_t)d_twiddles & 15) || ((uintptr_t)d_stft & 15)) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(join_tail(ctx));
    URH_TRY(launch_stft((const float2 *)d_x, n, window_size, hop, frames, d_window, (const double2 *)d_twiddles, (double2 *)d_stft, d_db,
                        ctx->stream));
    URH_HIP(hipGetLastError());
    return URHGPU_OK;
}

int urhgpu_bgra_lookup_dev(urhgpu_ctx *ctx, const float *d_db, int64_t frames, int window_size, const uint32_t *d_colormap,
                           int n_colors, float data_min, float data_max, uint32_t *d_image) {
    if (!ctx || !d_db || !d_colormap || !d_image || frames < 0 || window_size < 1) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(join_tail(ctx));
    URH_TRY(launch_bgra_lookup(d_db, frames, window_size, d_colormap, n_colors, data_min, data_max, d_image, ctx->stream));
    URH_HIP(hipGetLastError());
    return URHGPU_OK;
}

int urhgpu_convert_dev(urhgpu_ctx *ctx, const void *d_src, int src_dtype, void *d_dst, int dst_dtype, int64_t n) {
    if (!ctx || n < 0 || (n > 0 && (!d_src || !d_dst))) return URHGPU_ERR_ARG;
    if (src_dtype == dst_dtype) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(join_tail(ctx));
    URH_TRY(launch_convert(d_src, src_dtype, d_dst, dst_dtype, n, ctx->stream));
    URH_HIP(hipGetLastError());
    return URHGPU_OK;
}

int urhgpu_astype_dev(urhgpu_ctx *ctx, const void *d_src, int src_dtype, void *d_dst, int dst_dtype, int64_t n) {
    if (!ctx || n < 0 || (n > 0 && (!d_src || !d_dst))) return URHGPU_ERR_ARG;
    if (src_dtype == dst_dtype) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(join_tail(ctx));
    URH_TRY(launch_astype(d_src, src_dtype, d_dst, dst_dtype, n, ctx->stream));
    URH_HIP(hipGetLastError());
    return URHGPU_OK;
}

int urhgpu_pcm_to_iq_dev(urhgpu_ctx *ctx, const void *d_raw, int64_t n_frames, int channels, int sample_width, float *d_out) {
    if (!ctx || n_frames < 0 || (n_frames > 0 && (!d_raw || !d_out))) return URHGPU_ERR_ARG;
    if (channels < 1 || channels > 2 || sample_width < 1 || sample_width > 4) return URHGPU_ERR_ARG;      // (ValueError in the reference, :133, :164)
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(join_tail(ctx));
    URH_TRY(launch_pcm_to_iq(d_raw, n_frames, channels, sample_width, d_out, ctx->stream));
    URH_HIP(hipGetLastError());
    return URHGPU_OK;
}

int urhgpu_fft_peak_dev(urhgpu_ctx *ctx, const float *d_x, int64_t n, int64_t *peak_index) {
    if (!ctx || !peak_index || n < 1 || (n & (n - 1)) != 0 || !d_x) return URHGPU_ERR_ARG;
    int log2n = 0;
    while (((int64_t)1 << log2n) < n) ++log2n;
    if (log2n > 26) return URHGPU_ERR_UNSUPPORTED;           // (two LDS-sized factors of at most 8192 each)
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(join_tail(ctx));
    URH_TRY(ctx->arena.reserve(fft_peak_scratch_bytes(n) + 4096));
    ctx->arena.reset();
    void *scratch = ctx->arena.take(fft_peak_scratch_bytes(n));
    int64_t *d_peak = (int64_t *)ctx->arena.take(256);
    if (!scratch || !d_peak) return URHGPU_ERR_ARG;
    URH_TRY(launch_fft_peak((const float2 *)d_x, log2n, scratch, d_peak, ctx->stream));
    URH_HIP(hipGetLastError());
    URH_HIP(hipMemcpyAsync(peak_index, d_peak, 8, hipMemcpyDeviceToHost, ctx->stream));
    URH_HIP(hipStreamSynchronize(ctx->stream));
    return URHGPU_OK;
}

// IQArray.export_to_sub's run lengths (IQArray.py:275-304), host arithmetic on the bytes convert_to(uint8) produced on the device: the
// reference walks the FIRST component of every sample with (lastvalue, counter) -- equal to lastvalue: counter += 1; different: when
// counter > 1 the run is appended (positive above 127, negative otherwise) and the new value starts a run of 1, when counter is 1
// NOTHING happens (the value is dropped and lastvalue stays: a single sample never ends a run); the last run is always appended.
int urhgpu_sub_encode_runs(const uint8_t *values, int64_t n, int64_t stride, int64_t *runs_out, int64_t cap, int64_t *n_runs) {
    if (!values || n <= 0 || stride < 1 || !n_runs || cap < 0 || (cap > 0 && !runs_out)) return URHGPU_ERR_ARG;     // (an empty array: NameError in the reference)
    int64_t k = 0, counter = 0;
    uint8_t last = values[0];
    for (int64_t i = 0; i < n; ++i) {
        const uint8_t v = values[i * stride];
        if (v == last) { ++counter; continue; }
        if (counter > 1) {
            if (k < cap) runs_out[k] = last > 127 ? counter : -counter;
            ++k;
            counter = 1;
            last = v;
        }
    }
    if (k < cap) runs_out[k] = last > 127 ? counter : -counter;
    ++k;
    *n_runs = k;
    return k > cap ? URHGPU_ERR_CAPACITY : URHGPU_OK;
}

static int plot_elem_bytes(int dtype) {
    switch (dtype) {
        case URHGPU_DT_I8: case URHGPU_DT_U8: return 1;
        case URHGPU_DT_I16: case URHGPU_DT_U16: return 2;
        case URHGPU_DT_F32: return 4;
        default: return 0;
    }
}

int urhgpu_path_minmax_dev(urhgpu_ctx *ctx, const void *d_samples, int dtype, int64_t start, int64_t end,
                           int64_t samples_per_pixel, void *d_values) {
    if (!ctx || !d_samples || !d_values || start < 0 || end <= start || samples_per_pixel < 1) return URHGPU_ERR_ARG;
    if (!plot_elem_bytes(dtype)) return URHGPU_ERR_DTYPE;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(join_tail(ctx));
    URH_TRY(launch_path_minmax(d_samples, dtype, start, end, samples_per_pixel, d_values, ctx->stream));
    URH_HIP(hipGetLastError());
    return URHGPU_OK;
}

int urhgpu_path_minmax(urhgpu_ctx *ctx, const void *samples, int dtype, int64_t n, int64_t start, int64_t end,
                       int64_t samples_per_pixel, void *values) {
    if (!ctx || !samples || !values || start < 0 || end <= start || end > n || samples_per_pixel < 1) return URHGPU_ERR_ARG;
    const int eb = plot_elem_bytes(dtype);
    if (!eb) return URHGPU_ERR_DTYPE;
    const int64_t pixels = (end - start + samples_per_pixel - 1) / samples_per_pixel;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(ctx->staging.reserve(align256((size_t)(end - start) * eb) + align256((size_t)pixels * 2 * eb) + 1024));
    ctx->staging.reset();
    void *d_in = nullptr;
    URH_TRY(stage_in(ctx, (const char *)samples + (size_t)start * eb, (size_t)(end - start) * eb, &d_in));
    void *d_val = ctx->staging.take((size_t)pixels * 2 * eb);
    if (!d_val) return URHGPU_ERR_ARG;
    URH_TRY(urhgpu_path_minmax_dev(ctx, d_in, dtype, 0, end - start, samples_per_pixel, d_val));
    URH_HIP(hipMemcpyAsync(values, d_val, (size_t)pixels * 2 * eb, hipMemcpyDeviceToHost, ctx->stream));
    URH_HIP(hipStreamSynchronize(ctx->stream));
    return URHGPU_OK;
}

int urhgpu_bench_copy_ceiling_dev(urhgpu_ctx *ctx, const float *d_in, float *d_out, int64_t n_samples, int shape, int reps, float *ms_per_copy) {
    if (!ctx || !d_in || !d_out || !ms_per_copy || n_samples < 8192 || n_samples % 8192 || reps < 1 || shape < 0 || shape > 2) return URHGPU_ERR_ARG;
    if (((uintptr_t)d_in & 15) || ((uintptr_t)d_out & 15)) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(urhgpu_ctx_sync(ctx));
    hipStream_t cs = ctx->stream;
    if (shape == 2) {                                      // shape 0 on the CUs the hot kernel of pipelined passes runs on
        if (!ctx->hot_masked) return URHGPU_ERR_UNSUPPORTED;
        cs = ctx->hot_masked; shape = 0;
    }
    hipEvent_t e0, e1;
    URH_HIP(hipEventCreate(&e0));
    URH_HIP(hipEventCreate(&e1));
    for (int k = 0; k < 3; ++k) launch_copy_shape(d_in, d_out, n_samples, shape, cs);
    URH_HIP(hipEventRecord(e0, cs));
    for (int k = 0; k < reps; ++k) launch_copy_shape(d_in, d_out, n_samples, shape, cs);
    URH_HIP(hipEventRecord(e1, cs));
    URH_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    URH_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *ms_per_copy = ms / (float)reps;
    return URHGPU_OK;
}

// Probe (tools/boundary_probe.py; profiles/r05_boundary_anatomy.txt): `launches` back-to-back launches of the hot kernel ALONE (complex64
// 2-FSK, qad written, no tail) on one stream, every wavefront 0 leaving its s_memrealtime stamps in its ChunkInfo.
// synthetic company for the hot kernel (urhgpu_test_hot_probe: load_kind): dependent integer / float arithmetic for `ticks` x 10 ns ...
__global__ void k_probe_valu(long long ticks, int prio, float *sink) {
    if (prio) __builtin_amdgcn_s_setprio(3);
    const long long t0 = (long long)wall_clock64();
    float x = (float)threadIdx.x, y = 1.0f;
    unsigned long long z = threadIdx.x;
    do {
#pragma unroll
        for (int k = 0; k < 64; ++k) { x = x * 1.0001f + y; y = y * 0.9999f + x; z = z * 6364136223846793005ull + 1442695040888963407ull; }
    } while ((long long)wall_clock64() - t0 < ticks);
    if (x == 12345.678f && z == 42) *sink = y;
}
// ... or dependent random 64-byte-line loads over `lines` lines of `mem`
__global__ void k_probe_latency(long long ticks, const unsigned long long *mem, unsigned long long lines, unsigned long long *sink) {
    const long long t0 = (long long)wall_clock64();
    unsigned long long at = (blockIdx.x * 256ull + threadIdx.x) * 0x9e3779b97f4a7c15ull, acc = 0;
    do {
#pragma unroll 1
        for (int k = 0; k < 8; ++k) { const unsigned long long v = mem[(at % lines) * 8]; acc += v; at = at * 6364136223846793005ull + v + 1442695040888963407ull; }
    } while ((long long)wall_clock64() - t0 < ticks);
    if (acc == 0x1234567ull) *sink = acc;
}

__global__ void k_probe_spin(long long ticks) {              // one wavefront that does nothing for `ticks` x 10 ns (a bubble between two hot kernels)
    const long long t0 = (long long)wall_clock64();
    while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

int urhgpu_test_hot_stamps(int on) {
    urh::g_stamp_probe = (on != 0);
    return URHGPU_OK;
}

int urhgpu_test_tail_skip(int mask) {
    urh::g_tail_skip = mask;
    return URHGPU_OK;
}

// the chunk tables of the three most recent pipelined passes (current, previous, the one before), n_chunks entries each -- the table is
// the first thing a pass takes from its scratch arena
int urhgpu_test_fetch_chunk_tables(urhgpu_ctx *ctx, void *host_dst, int64_t n_chunks) {
    if (!ctx || !host_dst || n_chunks < 1) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(urhgpu_ctx_sync(ctx));
    const size_t bytes = (size_t)n_chunks * sizeof(ChunkInfo);
    const urh::Arena *order[3] = {&ctx->arena, &ctx->arena_alt2, &ctx->arena_alt};
    for (int k = 0; k < 3; ++k) {
        if (!order[k]->base || order[k]->cap < bytes) { memset((char *)host_dst + k * bytes, 0, bytes); continue; }
        URH_HIP(hipMemcpy((char *)host_dst + k * bytes, order[k]->base, bytes, hipMemcpyDeviceToHost));
    }
    return URHGPU_OK;
}

int urhgpu_test_hot_probe(urhgpu_ctx *ctx, const void *d_iq, int64_t n, const urhgpu_params *p, float *d_qad, int stream_kind, int event_mode,
                          int graded, int launches, int keep, void *d_chunks_out, int64_t *n_chunks_out, float *dur_ms, float *gap_ms, int bubble_us,
                          int load_kind) {
    if (!ctx || !d_iq || !p || !d_qad || !d_chunks_out || !n_chunks_out || launches < 1 || keep < 1 || keep > launches || n < kTile || n % kTile) return URHGPU_ERR_ARG;
    if (p->dtype != URHGPU_DT_F32 || p->mod != URHGPU_MOD_FSK || p->bits_per_symbol != 1) return URHGPU_ERR_UNSUPPORTED;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(join_tail(ctx));
    URH_TRY(urhgpu_ctx_sync(ctx));
    const Plan pl = make_plan(ctx, n, p->tolerance);
    const int64_t g = std::min<int64_t>(std::max(graded, 0), pl.n_chunks);
    const int64_t short_len = pl.chunk_len / 4;
    if (g > 0 && ((short_len % (URH_PROBE_WPB * kRowSamples)) != 0 || n % pl.chunk_len != 0)) return URHGPU_ERR_UNSUPPORTED;
    const int64_t n_launch = pl.n_chunks + 3 * g;              // the last g chunks cut into four short ones each
    URH_TRY(ctx->arena.reserve((size_t)n_launch * sizeof(ChunkInfo) + (size_t)n_launch * pl.slab_stride * 8 + 4096));
    ctx->arena.reset();
    ChunkInfo *chunks = (ChunkInfo *)ctx->arena.take((size_t)n_launch * sizeof(ChunkInfo));
    uint64_t *slab = (uint64_t *)ctx->arena.take((size_t)n_launch * pl.slab_stride * 8);
    if (!chunks || !slab) return URHGPU_ERR_ARG;
    RunArgs a;
    memset(&a, 0, sizeof(a));
    URH_TRY(fill_thresholds(a, p));
    a.in = d_iq; a.qad = d_qad; a.n = n; a.chunk_len = pl.chunk_len; a.slab_stride = pl.slab_stride;
    a.noise_sqrd = p->noise_threshold * p->noise_threshold; a.noise_val = noise_for(p); a.tol = p->tolerance;
    URH_TRY(max_magnitude_for(p->dtype, &a.max_magnitude));
    a.chunks = chunks; a.slab = slab; a.stamp_probe = 1;
    if (g > 0) { a.graded_from = pl.n_chunks - g; a.graded_len = short_len; }
    hipStream_t s = ctx->stream;
    if (stream_kind == 1) { if (!ctx->hot_masked) return URHGPU_ERR_UNSUPPORTED; s = ctx->hot_masked; }
    // event_mode 0: plain launches; 1: a completion event (timing disabled) attached to every dispatch, as the product's pipelined passes
    // do; 2: the same, created with hipEventDisableSystemFence | hipEventReleaseToDevice; 3: timing events (start + stop) on every dispatch
    std::vector<hipEvent_t> ev((size_t)launches * 2, nullptr);
    if (event_mode != 0) {
        const unsigned fl = event_mode == 1 ? hipEventDisableTiming : event_mode == 2 ? (hipEventDisableTiming | hipEventDisableSystemFence | hipEventReleaseToDevice) : hipEventDefault;
        for (auto &e : ev) URH_HIP(hipEventCreateWithFlags(&e, fl));
    }
    // load_kind: synthetic company on a second stream, started behind hot kernel j's completion event (so it runs beside hot kernel j + 1, as
    // the product's tail does).  1: 256 wavefronts of arithmetic for 200 us on the CUs the hot mask leaves out; 2: the same at s_setprio 3;
    // 3: 4096 workgroups of 4 wavefronts, 4 us of arithmetic each at s_setprio 3, anywhere on the chip; 4: 256 wavefronts of dependent random
    // loads for 200 us on the CUs left out; 5: six empty one-wavefront kernels in a row (kernel boundaries: cache write-back / invalidate)
    hipStream_t s2 = nullptr;
    void *load_mem = nullptr;
    if (load_kind != 0) {
        if (event_mode == 0 || event_mode == 3) return URHGPU_ERR_ARG;
        if (load_kind == 3 || load_kind == 5) URH_HIP(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        else {
            uint32_t m[8], inv[8];
            hot_cu_mask(ctx->tune_hot_cus_removed, m);
            for (int w = 0; w < 8; ++w) inv[w] = ~m[w];
            URH_HIP(hipExtStreamCreateWithCUMask(&s2, 8, inv));
        }
        URH_HIP(hipMalloc(&load_mem, size_t(256) << 20));
        URH_HIP(hipMemset(load_mem, 1, size_t(256) << 20));
        URH_HIP(hipDeviceSynchronize());
    }
    for (int j = 0; j < launches; ++j) {
        // the last `keep` launches write their chunk tables straight into the caller's buffer (nothing between two hot kernels)
        a.chunks = (j >= launches - keep) ? (ChunkInfo *)d_chunks_out + (size_t)(j - (launches - keep)) * n_launch : chunks;
        if (event_mode != 0) { g_hot_events = HotEvents(); g_hot_events.start = event_mode == 3 ? ev[2 * j] : nullptr; g_hot_events.stop = ev[2 * j + 1]; }
        const int st = launch_demod_runs_iq(a, p->dtype, p->mod, true, s);
        g_hot_events = HotEvents();
        if (st != URHGPU_OK) return st;
        if (bubble_us > 0) hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(64), 0, s, (long long)bubble_us * 100);
        if (load_kind != 0) {
            URH_HIP(hipStreamWaitEvent(s2, ev[2 * j + 1], 0));
            if (load_kind == 1 || load_kind == 2) hipLaunchKernelGGL(k_probe_valu, dim3(64), dim3(256), 0, s2, 20000ll, load_kind == 2 ? 1 : 0, (float *)load_mem);
            else if (load_kind == 3) hipLaunchKernelGGL(k_probe_valu, dim3(4096), dim3(256), 0, s2, 400ll, 1, (float *)load_mem);
            else if (load_kind == 4) hipLaunchKernelGGL(k_probe_latency, dim3(64), dim3(256), 0, s2, 20000ll, (const unsigned long long *)load_mem, (unsigned long long)((size_t(256) << 20) / 64), (unsigned long long *)load_mem);
            else for (int k = 0; k < 6; ++k) hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(64), 0, s2, 100ll);
        }
    }
    URH_HIP(hipStreamSynchronize(s));
    if (s2) { URH_HIP(hipStreamSynchronize(s2)); (void)hipStreamDestroy(s2); (void)hipFree(load_mem); }
    if (event_mode == 3 && dur_ms && gap_ms) {
        for (int j = 0; j < launches; ++j) {
            URH_HIP(hipEventElapsedTime(&dur_ms[j], ev[2 * j], ev[2 * j + 1]));
            if (j + 1 < launches) URH_HIP(hipEventElapsedTime(&gap_ms[j], ev[2 * j + 1], ev[2 * j + 2]));
        }
    }
    for (auto e : ev) if (e) (void)hipEventDestroy(e);
    *n_chunks_out = n_launch;
    return URHGPU_OK;
}

int urhgpu_memcpy_to_host(urhgpu_ctx *ctx, const void *d_src, void *host_dst, int64_t bytes) {
    if (!ctx || bytes < 0 || (bytes > 0 && (!d_src || !host_dst))) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(urhgpu_ctx_sync(ctx));
    if (bytes) URH_HIP(hipMemcpy(host_dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost));
    return URHGPU_OK;
}

int urhgpu_memcpy_dtod(urhgpu_ctx *ctx, void *d_dst, const void *d_src, int64_t bytes) {
    if (!ctx || bytes < 0 || (bytes > 0 && (!d_src || !d_dst))) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(urhgpu_ctx_sync(ctx));
    if (bytes) URH_HIP(hipMemcpy(d_dst, d_src, (size_t)bytes, hipMemcpyDeviceToDevice));
    return URHGPU_OK;
}

int urhgpu_test_force_state_bytes(int on) {
    urh::g_force_state_bytes = (on != 0);
    return URHGPU_OK;
}

int urhgpu_test_force_generic_tail(int on) {
    g_tile_tail = (on == 0);
    return URHGPU_OK;
}

int64_t urhgpu_test_wide_int_launches(void) { return (int64_t)urh::g_wide_int_launches.load(); }

int urhgpu_test_force_tiles_per_chunk(int tiles) {
    if (tiles < 0 || tiles > 4) return URHGPU_ERR_ARG;
    g_force_tiles_per_chunk = tiles;
    return URHGPU_OK;
}

int urhgpu_test_atan2f_dev(urhgpu_ctx *ctx, const float *d_y, const float *d_x, int64_t n, float *d_out) {
    if (!ctx) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    launch_test_atan2f(d_y, d_x, n, d_out, ctx->stream);
    URH_HIP(hipGetLastError());
    return URHGPU_OK;
}

// ---- host-buffer entry points ---------------------------------------------------------------------
static int stage_in(urhgpu_ctx *ctx, const void *host, size_t bytes, void **dev) {
    void *d = ctx->staging.take(bytes ? bytes : 16);
    if (!d) return URHGPU_ERR_ARG;
    if (bytes) URH_HIP(hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    *dev = d;
    return URHGPU_OK;
}

int urhgpu_afp_demod(urhgpu_ctx *ctx, const void *iq, int dtype, int64_t n, float noise_mag, int mod,
                     int mod_order, float costas_loop_bandwidth, float noise_other, float *qad_out) {
    if (!ctx || n < 0 || (n > 0 && (!iq || !qad_out))) return URHGPU_ERR_ARG;
    const int sb = dtype_bytes(dtype);
    if (sb == 0) return URHGPU_ERR_DTYPE;
    if (n == 0) return URHGPU_OK;
    URH_HIP(hipSetDevice(ctx->device));
    urhgpu_params p;
    memset(&p, 0, sizeof(p));
    p.dtype = dtype; p.mod = mod; p.noise_threshold = noise_mag; p.costas_loop_bandwidth = costas_loop_bandwidth;
    p.noise_other = noise_other;
    int bps = 0; while ((1 << (bps + 1)) <= mod_order) ++bps;
    p.bits_per_symbol = bps > 0 ? bps : 1;
    p.samples_per_symbol = 1;
    const size_t in_bytes = (size_t)n * sb, out_bytes = (size_t)n * 4;
    URH_TRY(ctx->staging.reserve(align256(in_bytes) + align256(out_bytes) + 1024));
    ctx->staging.reset();
    void *d_in = nullptr;
    URH_TRY(stage_in(ctx, iq, in_bytes, &d_in));
    float *d_out = (float *)ctx->staging.take(out_bytes);
    p.mod_order = mod_order;   // drives the Costas loop order directly (signal_functions.pyx:358)
    URH_TRY(urhgpu_afp_demod_dev(ctx, d_in, n, &p, d_out));
    URH_HIP(hipMemcpyAsync(qad_out, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    URH_HIP(hipStreamSynchronize(ctx->stream));
    return URHGPU_OK;
}

int urhgpu_grab_pulse_lens(urhgpu_ctx *ctx, const float *qad, int64_t n, float center, uint16_t tolerance,
                           int mod, uint32_t samples_per_symbol, uint8_t bits_per_symbol, float center_spacing,
                           float noise_other, int64_t *rows_out, int64_t cap_rows, int64_t *n_rows) {
    if (!ctx || n < 0 || !n_rows || cap_rows < 0) return URHGPU_ERR_ARG;
    *n_rows = 0;
    if (n == 0) return URHGPU_OK;
    if (!qad || (cap_rows > 0 && !rows_out)) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    urhgpu_params p;
    memset(&p, 0, sizeof(p));
    p.dtype = URHGPU_DT_F32; p.mod = mod; p.bits_per_symbol = bits_per_symbol; p.center = center;
    p.center_spacing = center_spacing; p.tolerance = tolerance; p.samples_per_symbol = samples_per_symbol;
    p.noise_other = noise_other;
    if (bits_per_symbol < 1 || bits_per_symbol > 7) return URHGPU_ERR_UNSUPPORTED;
    // worst case one row per (tolerance+1) samples; stage at that size on the device, copy what fits
    const int64_t dev_cap = n / ((int64_t)tolerance + 1) + 2;
    URH_TRY(ctx->staging.reserve(align256((size_t)n * 4) + align256((size_t)dev_cap * 16) + 1024));
    ctx->staging.reset();
    void *d_in = nullptr;
    URH_TRY(stage_in(ctx, qad, (size_t)n * 4, &d_in));
    int64_t *d_rows = (int64_t *)ctx->staging.take((size_t)dev_cap * 16);
    int64_t *d_n = ctx->d_counts + 10;
    URH_TRY(urhgpu_grab_pulse_lens_dev(ctx, (const float *)d_in, n, &p, d_rows, dev_cap, d_n));
    URH_HIP(hipMemcpyAsync(ctx->h_counts, d_n, 8, hipMemcpyDeviceToHost, ctx->stream));
    URH_HIP(hipStreamSynchronize(ctx->stream));
    const int64_t rows = ctx->h_counts[0];
    *n_rows = rows;
    if (rows > cap_rows) return URHGPU_ERR_CAPACITY;
    if (rows > 0) {
        URH_HIP(hipMemcpyAsync(rows_out, d_rows, (size_t)rows * 16, hipMemcpyDeviceToHost, ctx->stream));
        URH_HIP(hipStreamSynchronize(ctx->stream));
    }
    return URHGPU_OK;
}

int urhgpu_ppseq_to_bits(urhgpu_ctx *ctx, const int64_t *rows, int64_t n_rows, int64_t samples_per_symbol,
                         int bits_per_symbol, int write_pos, int64_t pause_threshold,
                         uint8_t *bits, int64_t cap_bits, int64_t *msg_off, int64_t *pauses, int64_t cap_msg,
                         int64_t *pos, int64_t cap_pos, int64_t *pos_off, int64_t *counts) {
    if (!ctx || n_rows < 0 || !counts || !msg_off || !pos_off || samples_per_symbol < 1 || bits_per_symbol < 1)
        return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    urhgpu_params p;
    memset(&p, 0, sizeof(p));
    p.samples_per_symbol = (uint32_t)samples_per_symbol; p.bits_per_symbol = bits_per_symbol;
    p.pause_threshold = pause_threshold; p.write_bit_sample_pos = write_pos;
    const int64_t cb = std::max<int64_t>(cap_bits, 1), cm = std::max<int64_t>(cap_msg, 1), cp = std::max<int64_t>(cap_pos, 1);
    const size_t need = align256((size_t)std::max<int64_t>(n_rows, 1) * 16) + align256((size_t)cb) + 3 * align256((size_t)(cm + 1) * 8) +
                        align256((size_t)cp * 8) + 4096;
    URH_TRY(ctx->staging.reserve(need));
    ctx->staging.reset();
    void *d_rows = nullptr;
    URH_TRY(stage_in(ctx, rows, (size_t)n_rows * 16, &d_rows));
    urhgpu_outputs o;
    memset(&o, 0, sizeof(o));
    o.bits = (uint8_t *)ctx->staging.take((size_t)cb); o.cap_bits = cap_bits;
    o.msg_off = (int64_t *)ctx->staging.take((size_t)(cm + 1) * 8);
    o.pauses = (int64_t *)ctx->staging.take((size_t)(cm + 1) * 8); o.cap_msg = cap_msg;
    o.pos_off = (int64_t *)ctx->staging.take((size_t)(cm + 1) * 8);
    o.pos = (int64_t *)ctx->staging.take((size_t)cp * 8); o.cap_pos = cap_pos;
    o.counts = ctx->d_counts;
    int64_t *d_n = ctx->d_counts + 10;
    ctx->h_counts[8] = n_rows;
    URH_HIP(hipMemcpyAsync(d_n, ctx->h_counts + 8, 8, hipMemcpyHostToDevice, ctx->stream));
    URH_TRY(urhgpu_ppseq_to_bits_dev(ctx, (const int64_t *)d_rows, d_n, n_rows, &p, &o));
    URH_HIP(hipMemcpyAsync(ctx->h_counts, ctx->d_counts, 4 * 8, hipMemcpyDeviceToHost, ctx->stream));
    URH_HIP(hipStreamSynchronize(ctx->stream));
    const int64_t n_msg = ctx->h_counts[1], n_bits = ctx->h_counts[2], n_pos = ctx->h_counts[3];
    counts[0] = n_msg; counts[1] = n_bits; counts[2] = n_pos;
    if (n_msg > cap_msg || n_bits > cap_bits || (write_pos && n_pos > cap_pos)) return URHGPU_ERR_CAPACITY;
    if (n_bits) URH_HIP(hipMemcpyAsync(bits, o.bits, (size_t)n_bits, hipMemcpyDeviceToHost, ctx->stream));
    URH_HIP(hipMemcpyAsync(msg_off, o.msg_off, (size_t)(n_msg + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    URH_HIP(hipMemcpyAsync(pos_off, o.pos_off, (size_t)(n_msg + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (n_msg) URH_HIP(hipMemcpyAsync(pauses, o.pauses, (size_t)n_msg * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (write_pos && n_pos) URH_HIP(hipMemcpyAsync(pos, o.pos, (size_t)n_pos * 8, hipMemcpyDeviceToHost, ctx->stream));
    URH_HIP(hipStreamSynchronize(ctx->stream));
    return URHGPU_OK;
}

// the FIR's own work area: one per context; a filter on another stream than the last one's waits for that one first
static int fir_work_area(urhgpu_ctx *ctx, size_t bytes, void **work) {
    if (!ctx->ev_fir) URH_HIP(hipEventCreateWithFlags(&ctx->ev_fir, hipEventDisableTiming));
    else if (ctx->fir_stream != ctx->stream || bytes + 1024 > ctx->fir_work.cap) URH_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_fir, 0));
    if (bytes + 1024 > ctx->fir_work.cap && ctx->fir_work.cap > 0) URH_HIP(hipEventSynchronize(ctx->ev_fir));      // (growing frees the old area)
    URH_TRY(ctx->fir_work.reserve(bytes + 1024));
    ctx->fir_work.reset();
    *work = ctx->fir_work.take(bytes);
    ctx->fir_stream = ctx->stream;
    return *work ? URHGPU_OK : URHGPU_ERR_ARG;
}

int urhgpu_fir_filter_dev(urhgpu_ctx *ctx, const float *d_x, int64_t n, const float *d_taps, int64_t m,
                          const float *d_left_halo, float *d_out) {
    if (!ctx || n < 0 || m < 0 || (n > 0 && (!d_x || !d_out)) || (m > 0 && !d_taps)) return URHGPU_ERR_ARG;
    if (((uintptr_t)d_x & 7) || ((uintptr_t)d_out & 15) || ((uintptr_t)d_taps & 7) || m > (int64_t)1 << 20) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    // The filter's work area (padded taps + the list of tiles handed back: kilobytes) is the context's own, not the rotating arena: on a
    // pipelined context the filter of capture i + 1 then runs BESIDE the tail of pass i instead of behind it (the FIR-halo variant of
    // configs[3] paid filter + hot kernel + tail per step).  The caller's stream is already ordered behind the last pass's HOT kernel
    // (digitize / shard_launch make it wait for that kernel's event), which is what d_out may alias: the capture that kernel read.
    void *work = nullptr;
    URH_TRY(fir_work_area(ctx, fir_work_bytes(n, (int)m), &work));
    URH_TRY(launch_fir((const float2 *)d_x, n, (const float2 *)d_taps, (int)m, (const float2 *)d_left_halo, (float2 *)d_out, ctx->stream, work));
    URH_HIP(hipEventRecord(ctx->ev_fir, ctx->stream));
    URH_HIP(hipGetLastError());
    return URHGPU_OK;
}

int urhgpu_fir_filter_stats_dev(urhgpu_ctx *ctx, const float *d_x, int64_t n, const float *d_taps, int64_t m, const float *d_left_halo,
                                float *d_out, int64_t chunk, int64_t n_chunks, double *d_sum, double *d_max) {
    if (!ctx || n <= 0 || m <= 0 || !d_x || !d_out || !d_taps || chunk <= 0 || n_chunks <= 0 || !d_sum || !d_max) return URHGPU_ERR_ARG;
    if (((uintptr_t)d_x & 7) || ((uintptr_t)d_out & 15) || ((uintptr_t)d_taps & 7) || m > (int64_t)1 << 20) return URHGPU_ERR_ARG;
    if (n_chunks * chunk > n || n_chunks > 65535) return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(join_tail(ctx));
    if (chunk < 2048) {          // a tile of outputs would meet more than two chunks: the filter, then the separate statistics pass
        URH_TRY(urhgpu_fir_filter_dev(ctx, d_x, n, d_taps, m, d_left_halo, d_out));
        return urhgpu_magnitude_chunk_stats_dev(ctx, d_out, URHGPU_DT_F32, n, chunk, n_chunks, d_sum, d_max);
    }
    URH_TRY(ctx->arena.reserve(fir_stats_scratch_bytes(n) + fir_work_bytes(n, (int)m) + 2048));
    ctx->arena.reset();
    void *scratch = ctx->arena.take(fir_stats_scratch_bytes(n));
    void *work = ctx->arena.take(fir_work_bytes(n, (int)m));
    if (!scratch || !work) return URHGPU_ERR_ARG;
    URH_TRY(launch_fir((const float2 *)d_x, n, (const float2 *)d_taps, (int)m, (const float2 *)d_left_halo, (float2 *)d_out, ctx->stream, work,
                       chunk, n_chunks, d_sum, d_max, scratch));
    URH_HIP(hipGetLastError());
    return URHGPU_OK;
}

int urhgpu_fir_filter(urhgpu_ctx *ctx, const float *x, int64_t n, const float *taps, int64_t m, float *out) {
    if (!ctx || n < 0 || m < 0 || (n > 0 && (!x || !out)) || (m > 0 && !taps)) return URHGPU_ERR_ARG;
    if (n == 0) return URHGPU_OK;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(ctx->staging.reserve(2 * align256((size_t)n * 8) + align256((size_t)std::max<int64_t>(m, 1) * 8) + 1024));
    ctx->staging.reset();
    void *d_x = nullptr, *d_t = nullptr;
    URH_TRY(stage_in(ctx, x, (size_t)n * 8, &d_x));
    URH_TRY(stage_in(ctx, taps, (size_t)m * 8, &d_t));
    float *d_out = (float *)ctx->staging.take((size_t)n * 8);
    if (!d_out) return URHGPU_ERR_ARG;
    URH_TRY(urhgpu_fir_filter_dev(ctx, (const float *)d_x, n, (const float *)d_t, m, nullptr, d_out));
    URH_HIP(hipMemcpyAsync(out, d_out, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    URH_HIP(hipStreamSynchronize(ctx->stream));
    return URHGPU_OK;
}

int urhgpu_bandpass_dev(urhgpu_ctx *ctx, const float *d_x, int64_t n, const double *d_taps, int64_t m, int64_t shift,
                        int64_t n_out, const float *d_left, int64_t n_left, const float *d_right, int64_t n_right, void *d_out,
                        int out_c64) {
    if (!ctx || n < 0 || m < 0 || n_out < 0 || n_left < 0 || n_right < 0 || (n > 0 && !d_x) || (m > 0 && !d_taps) ||
        (n_out > 0 && !d_out))
        return URHGPU_ERR_ARG;
    if (((uintptr_t)d_x & 7) || ((uintptr_t)d_taps & 15) || ((uintptr_t)d_out & (out_c64 ? 7 : 15)) || ((uintptr_t)d_left & 7) ||
        ((uintptr_t)d_right & 7) || m > (int64_t)1 << 20 || shift < -((int64_t)1 << 40) || shift > (int64_t)1 << 40)
        return URHGPU_ERR_ARG;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(join_tail(ctx));
    URH_TRY(ctx->arena.reserve(bandpass_fft_work_bytes() + 1024));
    ctx->arena.reset();
    void *work = ctx->arena.take(bandpass_fft_work_bytes());
    URH_TRY(launch_bandpass((const float2 *)d_x, n, (const float2 *)d_left, n_left, (const float2 *)d_right, n_right,
                            (const double2 *)d_taps, (int)m, shift, n_out, out_c64 ? nullptr : (double2 *)d_out,
                            out_c64 ? (float2 *)d_out : nullptr, ctx->stream, work));
    URH_HIP(hipGetLastError());
    return URHGPU_OK;
}

int urhgpu_bandpass(urhgpu_ctx *ctx, const float *x, int64_t n, const double *taps, int64_t m, int64_t shift, int64_t n_out,
                    double *out) {
    if (!ctx || n < 0 || m < 0 || n_out < 0 || (n > 0 && !x) || (m > 0 && !taps) || (n_out > 0 && !out)) return URHGPU_ERR_ARG;
    if (n_out == 0) return URHGPU_OK;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(ctx->staging.reserve(align256((size_t)std::max<int64_t>(n, 1) * 8) + align256((size_t)std::max<int64_t>(m, 1) * 16) +
                                 align256((size_t)n_out * 16) + 1024));
    ctx->staging.reset();
    void *d_x = nullptr, *d_t = nullptr;
    URH_TRY(stage_in(ctx, x, (size_t)n * 8, &d_x));
    URH_TRY(stage_in(ctx, taps, (size_t)m * 16, &d_t));
    double *d_out = (double *)ctx->staging.take((size_t)n_out * 16);
    if (!d_out) return URHGPU_ERR_ARG;
    URH_TRY(urhgpu_bandpass_dev(ctx, (const float *)d_x, n, (const double *)d_t, m, shift, n_out, nullptr, 0, nullptr, 0, d_out, 0));
    URH_HIP(hipMemcpyAsync(out, d_out, (size_t)n_out * 16, hipMemcpyDeviceToHost, ctx->stream));
    URH_HIP(hipStreamSynchronize(ctx->stream));
    return URHGPU_OK;
}

int urhgpu_iir_filter(urhgpu_ctx *ctx, const double *a, int64_t na, const double *b, int64_t nb, const float *x, int64_t n,
                      float *out) {
    if (!ctx || n < 0 || na < 0 || nb < 0 || (n > 0 && (!x || !out)) || (na > 0 && !a) || (nb > 0 && !b)) return URHGPU_ERR_ARG;
    if (n == 0) return URHGPU_OK;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(ctx->staging.reserve(2 * align256((size_t)n * 8) + align256((size_t)(na + 1) * 8) + align256((size_t)(nb + 1) * 8) + 1024));
    ctx->staging.reset();
    void *d_x = nullptr, *d_a = nullptr, *d_b = nullptr;
    URH_TRY(stage_in(ctx, x, (size_t)n * 8, &d_x));
    URH_TRY(stage_in(ctx, a, (size_t)na * 8, &d_a));
    URH_TRY(stage_in(ctx, b, (size_t)nb * 8, &d_b));
    float *d_out = (float *)ctx->staging.take((size_t)n * 8);
    if (!d_out) return URHGPU_ERR_ARG;
    URH_TRY(launch_iir((const double *)d_a, na, (const double *)d_b, nb, (const float2 *)d_x, n, (float2 *)d_out, ctx->stream));
    URH_HIP(hipGetLastError());
    URH_HIP(hipMemcpyAsync(out, d_out, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    URH_HIP(hipStreamSynchronize(ctx->stream));
    return URHGPU_OK;
}

int urhgpu_get_magnitudes(urhgpu_ctx *ctx, const void *iq, int dtype, int64_t n, double *out) {
    if (!ctx || n < 0 || (n > 0 && (!iq || !out))) return URHGPU_ERR_ARG;
    const int sb = dtype_bytes(dtype);
    if (sb == 0) return URHGPU_ERR_DTYPE;
    if (n == 0) return URHGPU_OK;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(ctx->staging.reserve(align256((size_t)n * sb) + align256((size_t)n * 8) + 1024));
    ctx->staging.reset();
    void *d_in = nullptr;
    URH_TRY(stage_in(ctx, iq, (size_t)n * sb, &d_in));
    double *d_out = (double *)ctx->staging.take((size_t)n * 8);
    if (!d_out) return URHGPU_ERR_ARG;
    URH_TRY(launch_magnitudes(d_in, dtype, n, d_out, ctx->stream));
    URH_HIP(hipGetLastError());
    URH_HIP(hipMemcpyAsync(out, d_out, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    URH_HIP(hipStreamSynchronize(ctx->stream));
    return URHGPU_OK;
}

// ---- host-array forms of the reference's util / auto_interpretation functions on the path (urh_amd/util.py, auto_interpretation.py) ----
static int value_bytes(int dtype) {
    switch (dtype) {
        case URHGPU_DT_I8: case URHGPU_DT_U8: return 1;
        case URHGPU_DT_I16: case URHGPU_DT_U16: return 2;
        case URHGPU_DT_F32: return 4;
        default: return 0;
    }
}

int urhgpu_minmax(urhgpu_ctx *ctx, const void *arr, int dtype, int64_t n, void *out2) {
    if (!ctx || n < 0 || !out2 || (n > 0 && !arr)) return URHGPU_ERR_ARG;
    const int vb = value_bytes(dtype);
    if (vb == 0) return URHGPU_ERR_DTYPE;
    if (n == 0) { memset(out2, 0, 2 * (size_t)vb); return URHGPU_OK; }              // util.pyx:22-23: (0, 0)
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(join_tail(ctx));
    URH_TRY(ctx->staging.reserve(align256((size_t)n * vb) + align256(minmax_scratch_bytes()) + 1024));
    ctx->staging.reset();
    void *d_in = nullptr;
    URH_TRY(stage_in(ctx, arr, (size_t)n * vb, &d_in));
    void *scratch = ctx->staging.take(minmax_scratch_bytes());
    void *d_out = ctx->staging.take(64);
    if (!scratch || !d_out) return URHGPU_ERR_ARG;
    URH_TRY(launch_minmax_any(d_in, dtype, n, d_out, scratch, ctx->stream));
    URH_HIP(hipGetLastError());
    URH_HIP(hipMemcpyAsync(out2, d_out, 2 * (size_t)vb, hipMemcpyDeviceToHost, ctx->stream));
    URH_HIP(hipStreamSynchronize(ctx->stream));
    return URHGPU_OK;
}

int urhgpu_segment_messages(urhgpu_ctx *ctx, const void *magnitudes, int is_f64, int64_t n, float noise_threshold, int64_t *seg_out,
                            int64_t cap_seg, int64_t *n_seg) {
    if (!ctx || n < 0 || !n_seg || cap_seg < 0 || (n > 0 && !magnitudes) || (cap_seg > 0 && !seg_out)) return URHGPU_ERR_ARG;
    *n_seg = 0;
    if (n == 0 || noise_threshold != noise_threshold) return URHGPU_OK;           // nothing compares greater than NaN
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(join_tail(ctx));
    const size_t eb = is_f64 ? 8 : 4;
    const int64_t cap_rows = n / 10 + 2, cap = cap_rows / 2 + 2;
    // magnitudes and flags live in the aux arena (urhgpu_segment_runs' digitize uses the main one), tables in the staging arena
    URH_TRY(ctx->aux.reserve(align256((size_t)n * eb) + align256((size_t)n * 4) + 1024));
    ctx->aux.reset();
    void *d_mag = ctx->aux.take((size_t)n * eb);
    float *d_flags = (float *)ctx->aux.take((size_t)n * 4);
    URH_TRY(ctx->staging.reserve((size_t)cap_rows * 16 + 2 * (size_t)cap * 16 + seg_scratch_bytes(cap_rows, cap) + seg_ctl_bytes() + 16 * 256));
    ctx->staging.reset();
    int64_t *d_rows = (int64_t *)ctx->staging.take((size_t)cap_rows * 16);
    int64_t *d_seg = (int64_t *)ctx->staging.take((size_t)cap * 16);
    int64_t *d_msgs = (int64_t *)ctx->staging.take((size_t)cap * 16);
    void *scratch = ctx->staging.take(seg_scratch_bytes(cap_rows, cap));
    SegCtl *d_ctl = (SegCtl *)ctx->staging.take(seg_ctl_bytes());
    int64_t *d_n_rows = (int64_t *)ctx->staging.take(64);
    if (!d_mag || !d_flags || !d_rows || !d_seg || !d_msgs || !scratch || !d_ctl || !d_n_rows) return URHGPU_ERR_ARG;
    URH_HIP(hipMemcpyAsync(d_mag, magnitudes, (size_t)n * eb, hipMemcpyHostToDevice, ctx->stream));
    URH_TRY(launch_above_flags(d_mag, is_f64, n, noise_threshold, d_flags, ctx->stream));
    urhgpu_params p;
    memset(&p, 0, sizeof(p));
    p.dtype = URHGPU_DT_F32; p.mod = URHGPU_MOD_ASK; p.bits_per_symbol = 1; p.center = 0.5f; p.center_spacing = 0.f;
    p.tolerance = 9;                                   // outlier_tolerance = 10 consecutive samples (auto_interpretation.pyx:72)
    p.samples_per_symbol = 1;
    const Plan pl = make_plan(ctx, n, p.tolerance);
    URH_TRY(ctx->arena.reserve(digitize_scratch_bytes(pl, cap_rows, false, false)));
    ctx->arena.reset();
    URH_TRY(digitize(ctx, false, d_flags, n, &p, nullptr, d_rows, cap_rows, d_n_rows, ctx->d_counts + 8, ctx->d_counts + 9, pl, 1));
    URH_TRY(launch_message_ranges(d_rows, d_n_rows, cap_rows, d_flags, kDtAboveFlags, n, 0.5f, 0, d_seg, d_msgs, cap, d_ctl, scratch, ctx->stream));
    URH_HIP(hipGetLastError());
    std::vector<char> ctl(seg_ctl_bytes());
    URH_HIP(hipMemcpyAsync(ctl.data(), d_ctl, ctl.size(), hipMemcpyDeviceToHost, ctx->stream));
    URH_HIP(hipStreamSynchronize(ctx->stream));
    int64_t ns = 0, nm = 0;
    int amb = 0;
    seg_ctl_read(ctl.data(), &ns, &nm, &amb);
    *n_seg = ns;
    if (ns > cap_seg) return URHGPU_ERR_CAPACITY;
    if (ns > 0) {
        URH_HIP(hipMemcpyAsync(seg_out, d_seg, (size_t)ns * 16, hipMemcpyDeviceToHost, ctx->stream));
        URH_HIP(hipStreamSynchronize(ctx->stream));
    }
    return URHGPU_OK;
}

int urhgpu_get_plateau_lengths(urhgpu_ctx *ctx, const float *rect_data, int64_t n, float center, int percentage, uint64_t *out, int64_t cap,
                               int64_t *n_out) {
    if (!ctx || n < 0 || !n_out || cap < 0 || percentage < 0 || (n > 0 && !rect_data) || (cap > 0 && !out)) return URHGPU_ERR_ARG;
    *n_out = 0;
    if (n == 0) return URHGPU_OK;                              // auto_interpretation.pyx:180-181
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(join_tail(ctx));
    URH_TRY(ctx->aux.reserve(align256((size_t)n * 4) + 1024));
    ctx->aux.reset();
    float *d_x = (float *)ctx->aux.take((size_t)n * 4);
    if (!d_x) return URHGPU_ERR_ARG;
    URH_HIP(hipMemcpyAsync(d_x, rect_data, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    const int64_t range[2] = {0, n};
    const double c = (double)center;
    int64_t off[2] = {0, 0};
    // boundaries are searched in the first percentage % + a window that doubles until it holds one beyond the mark (or the signal ends)
    for (int64_t extra = int64_t(1) << 16;; extra *= 2) {
        const int st = urhgpu_msg_plateaus(ctx, d_x, n, range, &c, 1, percentage, extra, off, out, cap);
        if (st == URHGPU_ERR_CAPACITY) { *n_out = off[1]; return st; }
        URH_TRY(st);
        if (off[1] >= 0) break;
        if (extra >= n) { off[1] = -off[1] - 1; break; }
    }
    *n_out = off[1];
    return URHGPU_OK;
}

int urhgpu_median_filter(urhgpu_ctx *ctx, const double *data, int64_t n, unsigned int k, float *out) {
    if (!ctx || n < 0 || (n > 0 && (!data || !out))) return URHGPU_ERR_ARG;
    if (k < 1 || k > 64) return URHGPU_ERR_UNSUPPORTED;
    if (n == 0) return URHGPU_OK;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(join_tail(ctx));
    URH_TRY(ctx->staging.reserve(align256((size_t)n * 8) + align256((size_t)n * 4) + 1024));
    ctx->staging.reset();
    void *d_in = nullptr;
    URH_TRY(stage_in(ctx, data, (size_t)n * 8, &d_in));
    float *d_out = (float *)ctx->staging.take((size_t)n * 4);
    if (!d_out) return URHGPU_ERR_ARG;
    URH_TRY(launch_median_filter((const double *)d_in, n, (int)k, d_out, ctx->stream));
    URH_HIP(hipGetLastError());
    URH_HIP(hipMemcpyAsync(out, d_out, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    URH_HIP(hipStreamSynchronize(ctx->stream));
    return URHGPU_OK;
}

int urhgpu_magnitude_chunk_stats_dev(urhgpu_ctx *ctx, const void *d_iq, int dtype, int64_t n, int64_t chunk, int64_t n_chunks,
                                     double *d_sum, double *d_max) {
    if (!ctx || n < 0 || n_chunks < 0 || (n_chunks > 0 && (!d_iq || !d_sum || !d_max))) return URHGPU_ERR_ARG;
    if (dtype_bytes(dtype) == 0) return URHGPU_ERR_DTYPE;
    if (n_chunks == 0) return URHGPU_OK;
    URH_HIP(hipSetDevice(ctx->device));
    URH_TRY(join_tail(ctx));
    URH_TRY(ctx->arena.reserve(mag_chunk_scratch_bytes(n_chunks) + 1024));
    ctx->arena.reset();
    void *scratch = ctx->arena.take(mag_chunk_scratch_bytes(n_chunks));
    if (!scratch) return URHGPU_ERR_ARG;
    URH_TRY(launch_mag_chunk_stats(d_iq, dtype, n, chunk, n_chunks, d_sum, d_max, scratch, ctx->stream));
    URH_HIP(hipGetLastError());
    return URHGPU_OK;
}

}  // extern "C"
