// Argument structs and host-side launcher prototypes shared by the kernel files and capi.hip.
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.hpp"
#include "runs.hpp"

namespace urh {

// ---- demod_runs.hip ---------------------------------------------------------------------------------
struct RunArgs {
    const void *in;          // IQ (dtype) or qad (float) -- device pointer
    float *qad;              // demodulated output or nullptr
    const void *left_halo;   // 2 IQ samples (or 1 qad sample) preceding in[0]; nullptr = global start
    int64_t n;               // samples in the buffer `in`
    int64_t range_begin;     // per launch (set by the launchers): samples [range_begin, range_end) of `in`,
    int64_t range_end;       //   workgroup b = chunk chunk_base + b
    int64_t chunk_base;
    int64_t pos_base;        // absolute position of in[0] (sharded captures); positions in records are absolute
    int64_t chunk_len;       // multiple of kTile
    uint64_t *slab;          // accepted-run records, slab_stride per chunk
    int64_t slab_stride;
    ChunkInfo *chunks;
    float noise_sqrd;
    float noise_val;         // NOISE sentinel (0.0 ASK, -4.0 FSK/PSK)
    float max_magnitude;     // ASK normalisation
    int order;               // modulation order = 2^bits_per_symbol
    int tol;                 // tolerance
    int launch_part;         // 0 all chunks, 1 all but the first, 2 the first only (launchers; see launch_runs_4)
    int seg_mode;            // 1: message segmentation on magnitudes (auto_interpretation.pyx:55-111): sample 0 is an
                             // ordinary sample (no result[0] = NOISE) and the state machine starts in ITS state
    // seg_mode with a qad buffer (float32 captures): the segmentation pass also leaves afp_demod(iq, noise, "ASK") there, from its own
    // constants (noise_sqrd / max_magnitude / noise_val above are the segmentation's: no gating, magnitude 1)
    float dm_noise_sqrd, dm_max_magnitude, dm_noise_val;
    int lds_pad;             // extra dynamic LDS bytes per workgroup of k_demod_runs_bp: caps its workgroups per CU so that
                             // wave slots stay free for the tail of the previous pass (pipelined mode)
    // Streamed passes (segmented tail, pulse_table.hip "segments"): the tail of the first chunks runs while the hot kernel is still
    // working on the later ones.  progress != nullptr: the bit-plane kernel writes every chunk's records THROUGH to memory (agent-scope
    // stores), waits for their acknowledgement and then counts the chunk into progress[k], k = the first segment with chunk <
    // seg_end[k]; a gate kernel on the tail stream (k_seg_gate) lets segment k's tail start when its counter is complete.
    uint32_t *progress;
    int32_t n_seg;
    int32_t seg_end[kMaxSegments];
    // chunks [launch_lo, launch_hi) only (launch_hi == 0: governed by launch_part); the partial tile at the end of the capture counts as
    // chunk n_main
    int64_t launch_lo, launch_hi;
    // Graded tail (bit-plane kernel; urhgpu_test_hot_probe only -- measured and NOT adopted, profiles/r05_boundary_anatomy.txt: 273-281 us against
    // 269): chunks [0, graded_from) are chunk_len samples long, the chunks from graded_from on graded_len (a multiple of W rows), so that the
    // launch's last residency wave consists of short-lived workgroups.  graded_from == 0: uniform chunks.
    int64_t graded_from, graded_len;
    int wide_int;            // integer FSK captures: take the bit-plane kernel's instantiation with the wide loop (capture streams set it from k_wide_probe's count)
    int stamp_probe;         // tools/boundary_probe.py: the STAMPS instantiation of the bit-plane kernel (complex64 2-FSK only); 0 in the product
    float thr[kMaxOrder - 1];
};
extern bool g_force_state_bytes;
extern std::atomic<long long> g_wide_int_launches;
extern bool g_stamp_probe;
extern int g_tail_skip;            // pulse_table.hip: measurement hook (urhgpu_test_tail_skip)
// urhgpu_ctx_profile_*: start / stop events attached to the next bit-plane hot-kernel dispatch itself (hipExtLaunchKernelGGL:
// the kernel's own begin / end timestamps, what rocprofv3 reports); `used` says the launcher took them
struct HotEvents { hipEvent_t start = nullptr, stop = nullptr; bool used = false; };
extern thread_local HotEvents g_hot_events;   // test hook: order 2 through the state-byte kernel too
int launch_wide_probe(const void *d_iq, int dtype, int64_t n, float noise_sqrd, int32_t *h_out, hipStream_t s);      // demod_runs.hip: k_wide_probe
int launch_demod_runs_iq(const RunArgs &a, int dtype, int mod, bool write_qad, hipStream_t s);
int launch_runs_qad(const RunArgs &a, hipStream_t s);
bool runs_streamable(const RunArgs &a);       // RunArgs::progress is honoured for these arguments (bit-plane kernel, whole tiles)
int launch_afp_demod(const RunArgs &a, int dtype, int mod, int grid, hipStream_t s);
void launch_test_div(uint64_t seed, int reps, unsigned long long *d_mismatches, hipStream_t s);
void launch_test_sincosf_fast(unsigned long long *d_mismatches, hipStream_t s);
void launch_test_atan2f(const float *y, const float *x, int64_t n, float *out, hipStream_t s);

// ---- modulate.hip ------------------------------------------------------------------------------------
struct ModMsg {              // one message of a modulate batch
    int64_t bit_off;         // first bit in bits[]
    int64_t n_sym;           // symbols (= bits // bits_per_symbol)
    int64_t sym_off;         // first entry in phase[] (FSK)
    int64_t out_off;         // first output sample
    uint32_t pause;          // silent samples after the symbols
    uint32_t start;          // sample index of the message's first sample (time origin of the carrier)
};
constexpr int kModGfsk = 5;  // internal to launch_modulate (public entry: urhgpu_modulate_gfsk*)
struct ModArgs {
    const uint8_t *bits;     // device: all messages' bits back to back
    const ModMsg *msgs;      // device
    const float *params;     // device: 2^bits_per_symbol amplitudes / frequencies / phases
    float *phase;            // device scratch: one float per symbol (FSK)
    void *out;               // device: (total samples, 2) of dtype
    int n_msgs, mod, dtype, bps;
    int oqpsk;               // mod == PSK on re-ordered bits, first symbol's Q and last symbol's I blanked
    // GFSK (mod == kModGfsk): Gaussian-filtered frequency and phase per data sample (entry sym_off * sps + i of message m)
    const float *taps;       // device: gauss_fir, n_taps floats
    int n_taps;
    int freq_given;          // gf_freq already holds the filtered frequencies (caller's convolution)
    float *gf_freq, *gf_phase;
    uint32_t sps;
    float carrier_amplitude, carrier_frequency, carrier_phase, sample_rate;
};
int launch_modulate(const ModArgs &a, int64_t max_samples, hipStream_t s);

// ---- spectrogram.hip ---------------------------------------------------------------------------------
int launch_stft(const float2 *x, int64_t n, int ws, int64_t hop, int64_t frames, const double *window, const double2 *tw,
                double2 *out_c128, float *out_db, hipStream_t s);
int launch_bgra_lookup(const float *data, int64_t frames, int ws, const uint32_t *colormap, int n_colors, float dmin, float dmax,
                       uint32_t *image, hipStream_t s);

// ---- convert.hip -------------------------------------------------------------------------------------
int launch_convert(const void *src, int src_dtype, void *dst, int dst_dtype, int64_t n, hipStream_t s);
int launch_astype(const void *src, int src_dtype, void *dst, int dst_dtype, int64_t n, hipStream_t s);
int launch_pcm_to_iq(const void *raw, int64_t n_frames, int channels, int width, float *out, hipStream_t s);
size_t fft_peak_scratch_bytes(int64_t n);
int launch_fft_peak(const float2 *x, int log2n, void *scratch, int64_t *d_peak, hipStream_t s);

// ---- plot.hip ----------------------------------------------------------------------------------------
int launch_path_minmax(const void *samples, int dtype, int64_t start, int64_t end, int64_t spp, void *values, hipStream_t s);

// ---- bandpass.hip ------------------------------------------------------------------------------------
int launch_bandpass(const float2 *x, int64_t n, const float2 *left, int64_t n_left, const float2 *right, int64_t n_right,
                    const double2 *taps, int m, int64_t shift, int64_t n_out, double2 *out128, float2 *out64, hipStream_t s,
                    void *work = nullptr);
size_t bandpass_fft_work_bytes();

// ---- pulse_table.hip ---------------------------------------------------------------------------------
// Scratch of the resolve stage: one entry per chunk (ints are chunk indices or -1).
struct ResolveScratch {
    int32_t *has_stable;     // c if chunk c contains a stable run (incl. a pending run that turns stable) else -1
    int32_t *prev_stable;    // last chunk before c that contains a stable run, or -1
    int32_t *has_acc;        // c if chunk c contributes at least one accepted run else -1
    int32_t *prev_acc;       // last chunk before c that contributes an accepted run, or -1
    int64_t *out_cnt;        // accepted runs contributed by chunk c
    int64_t *out_off;        // index of chunk c's first accepted run in the global accepted sequence
    int32_t *blk_stable;     // per resolve workgroup: last chunk with a stable run (or -1)
    int32_t *blk_acc;        // per resolve workgroup: last chunk that contributes an accepted run (or -1)
    int64_t *blk_cnt;        // per resolve workgroup: accepted runs
};
size_t resolve_scratch_bytes(int64_t n_chunks);
ResolveScratch resolve_scratch_carve(void *mem, int64_t n_chunks);

// Small device block shared by the resolve kernels of one pass.
struct ResolveAux {
    int32_t first_nonlead;   // first chunk that contains a run boundary (atomicMin; >= n_chunks: none)
    int32_t open_chunk;      // local pass: chunk whose trailing short run reaches the shard end undecided
    int32_t first_stable;    // first chunk with a stable run (atomicMin)
    int32_t last_stable;     // last chunk with a stable run, -1: none
};
constexpr int32_t kAuxNone = 0x7F7F7F7F;   // hipMemsetAsync(aux, 0x7F, ...)

struct ResolveArgs {
    ChunkInfo *chunks;
    int64_t n_chunks;        // entries in the table: the local chunks, for a sharded capture preceded /
                             // followed by one summary entry per other shard
    int64_t n_total;         // samples in the whole capture (local pass: in the shard)
    int tol;
    // local_pass: the table is ONE shard of a sharded capture on its own.  The state before it is unknown
    // (its first stable run counts as accepted, tentatively), a trailing short run that reaches the shard end
    // stays open, no last row is written, and *summary_out receives the ChunkInfo that stands for the whole
    // shard in the other ranks' tables.
    int local_pass;
    ResolveAux *aux;         // persistent context memory, kAuxNone / -1 between passes
    ChunkInfo *summary_out;
    int64_t chunk_first;     // table index of this GPU's first chunk
    int64_t n_local;         // this GPU's chunks
    int64_t *d_ts_carry;     // out: sum of the row lengths before this GPU's first row (sharded captures)
    int64_t *rows;           // pulse table (may be nullptr when only counting)
    int64_t cap_rows;
    int64_t *d_n_acc;        // out: number of accepted runs P
    int64_t *d_n_rows;       // out: rows of the un-merged table (P+1, or P when P == n_total), clamped to cap_rows
    int64_t *d_n_rows_needed; // out: the same, unclamped (capacity check on the host)
    int write_last_row;      // 1 on the shard/GPU that owns the table's last row
    ResolveScratch sc;
};
struct EmitArgs {
    const ChunkInfo *chunks;
    int64_t chunk_first;      // index (in the table) of this GPU's first chunk
    const uint64_t *slab;
    int64_t slab_stride;
    int64_t *rows;            // this GPU's rows: global row g goes to rows[g - out_off[chunk_first]]
    int64_t cap_rows;
    int64_t *d_ts_carry;
    int is_ask;
    int64_t sps;
    ResolveScratch sc;
};
struct BitsParams {
    int64_t sps;
    int64_t bps;
    int64_t pause_threshold;
    int64_t samples_per_bit;
    int write_pos;
    // sharded captures (all nullptr / 1 on a single GPU):
    const int64_t *d_row_base;    // global index of this GPU's first row
    const int64_t *d_ts_carry;    // sum of the row lengths before it
    const int64_t *d_absorbed;    // full length of the pause this GPU's only (absorbed) row belongs to, or -1
    const int32_t *d_extra;       // [0]: the group my first rows continue has data on earlier ranks, [1]: later ranks
    int is_last_rank;             // the capture ends on this GPU: the trailing group closes here
    const int64_t *d_rows_needed; // un-clamped row count of the pulse table (-> counts[4]) or nullptr
};
constexpr int64_t kRowAbsorbed = -(int64_t(1) << 62);   // == URHGPU_ROW_ABSORBED
struct BitsOut {
    uint8_t *bits; int64_t cap_bits;
    int64_t *msg_off; int64_t *pauses; int64_t cap_msg;
    int64_t *pos; int64_t cap_pos; int64_t *pos_off;
    int64_t *counts;
    int64_t *h_counts = nullptr;   // optional second destination of the counts: pinned HOST memory (zero-copy store), see urhgpu_outputs::h_counts
};
int launch_resolve(const ResolveArgs &a, int32_t *tickets, hipStream_t s);
int launch_shard_summary(const ResolveArgs &a, int32_t *ticket, hipStream_t s);   // local pass of a sharded capture in one launch (*ticket zero between launches)
int launch_resolve_emit_single(const ResolveArgs &r, const EmitArgs &e, hipStream_t s);
int launch_emit_rows(const EmitArgs &a, int64_t n_local_chunks, hipStream_t s);
size_t merge_scratch_bytes(int64_t cap);
int launch_merge_rows_ask(const int64_t *rows_in, const int64_t *d_n_in, int64_t cap, int64_t *rows_out,
                          int64_t cap_out, int64_t *d_n_out, void *scratch, int32_t *tickets, hipStream_t s);
size_t bits_scratch_bytes(int64_t cap_rows);
// Persistent per-context state of the scan kernels (scan.hpp): 4 device ints that are zero between launches ("last
// workgroup done" elections), and the descriptor array of the single-pass scans -- dedicated memory that is zeroed when it
// is (re)allocated and only ever holds descriptors, whose flags carry the pass counter `*epoch`.
struct ScanState {
    int32_t *tickets;
    void *desc;
    size_t desc_bytes;
    unsigned long long *epoch;
};
size_t bits_desc_bytes(int64_t cap_rows);
int launch_ppseq_to_bits(const int64_t *rows, const int64_t *d_n_rows, int64_t cap_rows, const BitsParams &bp,
                         const BitsOut &o, void *scratch, const ScanState &ss, hipStream_t s);
// The same in two halves (sharded captures: the boundary flags are exchanged in between).
//   prepare: per-row scan; d_flags (3 x int64) = {long pause present, data before the first one, data after the last one}
//   finish : groups -> messages, expansion (bp.d_extra resolved from every rank's flags)
int launch_bits_prepare(const int64_t *rows, const int64_t *d_n_rows, int64_t cap_rows, const BitsParams &bp,
                        void *scratch, int64_t *d_flags, const ScanState &ss, hipStream_t s);
int launch_bits_finish(const int64_t *rows, const int64_t *d_n_rows, int64_t cap_rows, const BitsParams &bp,
                       const BitsOut &o, void *scratch, const ScanState &ss, hipStream_t s);
// ---- streamed passes: the tile tail in SEGMENTS (pulse_table.hip "Segments") --------------------------------------------------
// Device-resident state that carries a pass from one segment of its tail to the next (one block per scratch arena).  The ROWS of the
// pulse table are made by rows segments, the bits / pauses / positions by (fewer, coarser) bits segments on a second stream;
// `in[j & 1]` is what bits segment j starts from, written by bits segment j - 1's group scan (its Final functor).
struct SegState {
    int64_t rows_at[kMaxSegments];   // [k]: pulse-table rows that are final after rows segment k (clamped to cap_rows); the last: n_rows
    int64_t rows_needed;     // unclamped row count (the table's size before clamping once the last rows segment has run)
    int64_t n_acc;           // accepted runs (the last rows segment)
    int64_t n_groups;        // groups so far, the open (trailing) one included: d_n_groups of the current bits segment
    int64_t n_groups_local;  // groups the current bits segment's group scan covers: [in.g0, n_groups)
    int64_t end_bits, end_pos, end_msgs;   // bits / positions written and messages closed after the current bits segment
    int64_t err;             // != 0: a gate gave up waiting for the hot kernel (the pass's results are void)
    struct In {
        int64_t g0;          // the group that was open at the end of the bits segment before: this one's group scan starts with it
        int64_t carry[3];    // exclusive prefix at g0: messages closed, kept bits, kept positions before it
        int64_t ship_bits, ship_pos, ship_msgs, pad;   // what the host already holds: the segment's pack kernel starts there
    } in[2];
};
static_assert(sizeof(SegState) == 128 + 64 + 128, "SegState");
// segment boundaries are multiples of kSegAlign chunks (or the capture's end): resolve workgroups, tile-scan workgroups and the
// wavefronts' chunk quadruples all start on it
constexpr int64_t kSegAlign = 256;
// the gate of a rows segment, inside its resolve kernel: every workgroup waits until `target` chunks have counted themselves into
// progress[k] (RunArgs::progress); init: the pass's first segment also clears the state the bits segments carry along
struct SegGate {
    const uint32_t *progress;   // nullptr: no gate (the hot kernel is known to have finished)
    int k;
    uint32_t target;
    int init;
    SegState *seg;
    long long max_ticks;        // of the 100 MHz wall clock: give up (seg->err) instead of hanging when the hot kernel never comes
    int fused;                  // launch_rows_segment: the segment's resolve kernel is ONE workgroup, whose first thread does the polling itself
                                // (one kernel less on the chain behind the hot kernel's end: the capture's last, short segment)
};
struct RowsSegment {         // chunks [c0, c1) of the capture's n_chunks (c0 a multiple of kSegAlign; c1 too, or n_chunks)
    int index;               // k
    int final;               // the capture's last segment: totals, the table's last row
    int64_t c0, c1;
    SegGate gate;
    int8_t *h_state;         // pinned HOST memory (device-accessible) that receives the rows as they are written: the compact blob's
    int32_t *h_len;          // row_state / row_len sections (nullptr: rows are not shipped)
    int fuse_gate;           // allow the gate inside a one-workgroup resolve kernel (SegGate::fused)
    // staged passes (round 6): lengths as uint16 INSTEAD of int32 (h_len then names the same section: 2 bytes per row), a length that does not
    // fit -- 65535 and more, or negative -- is stored as 0xFFFF and appended to esc: esc[0] = count (zeroed by the resolve kernel), then
    // {uint32 row, int32 length} pairs, esc_cap of them at most (a capture of n samples has at most n / 65535 + 2 such rows)
    // len16 == 2 (URHGPU_BLOB_ROW16, dense pulse tables): ONE uint16 per row -- (state + 1) << 13 | length, 0x1FFF = escaped (8191 samples and
    // more, or negative: at most n / 8191 + 2 rows) -- in the row_len section; nothing is stored into h_state
    int len16;
    int64_t *esc;
    int64_t esc_cap;
};
struct BitsSegment {         // tiles [c0, c1) (the last one: + the tile of the table's last row); needs the rows of chunks < c1
    int index;               // j
    int final;
    int64_t c0, c1;
    int rows_index;          // rows segment whose end is c1: d_n_rows = &state->rows_at[rows_index]
    SegState *state;
};
struct SegPackDst {          // where a bits segment's share of the compact blob goes: pinned HOST memory (device-accessible), see k_pack_seg
    void *host;              // nullptr: nothing is shipped
    int64_t cap_host;        // >= blob_capacity of the pass's capacities
    uint32_t *progress_reset;   // the counters the last segment zeroes (nullptr: none)
    int blocks;              // workgroups of the pack kernel (0: default)
    int pos_direct;          // a pass of ONE segment: positions are stored into the host blob by the kernels that write them (pulse_table.hip)
    int split;               // staged passes (ONE segment): `host` is the staging blob in HBM, in the split layout (compact.hpp: staged_layout)
    void *host_head;         // ... and the head (header + pauses / offsets / packed bits: small) goes straight into this pinned HOST blob
    const int64_t *esc;      // staged passes with 16-bit row lengths: the escape list (RowsSegment::esc) the last kernel appends to the head; nullptr: int32 lengths
    int64_t esc_cap;
    int64_t row16_esc_off;   // > 0: URHGPU_BLOB_ROW16 -- the list goes to this offset of host_head (behind every section: its own, sized place)
};

// Tile tail (single GPU, not ASK): resolve + rows in two launches, bits in three more; see pulse_table.hip.
struct TileTailMem {
    void *mem;               // tile_tail_bytes(n_chunks) of scratch that lives from launch_tile_rows to launch_tile_bits
    int64_t n_chunks;
    int32_t *huge_count;     // 2 persistent ints of the context, zero between passes
    int parity;              // which of the two this pass uses (alternates)
    void *rdesc;             // tile_rdesc_bytes(n_chunks) of persistent, initially zeroed memory: look-back descriptors of the resolve scan
    unsigned long long epoch; // pass counter carried by their flags (never repeats on a context)
    int64_t *d_row_base = nullptr;   // sharded captures: device word that receives the global index of this GPU's first row
};
size_t tile_rdesc_bytes(int64_t n_chunks);
size_t tile_tail_bytes(int64_t n_chunks);
int64_t tile_desc_cap(int64_t cap_rows, int64_t n_chunks);
int launch_tile_rows(const ResolveArgs &r, const EmitArgs &e, const TileTailMem &m, const BitsParams *bp, hipStream_t s);
int launch_tile_bits(const TileTailMem &m, const int64_t *rows, const int64_t *d_n_rows, int64_t cap_rows, const BitsParams &bp,
                     const BitsOut &o, void *scratch, const ScanState &ss, hipStream_t s);
// the same in two halves around the flags exchange of a sharded capture (d_flags[3]: long pause present, data before the first / after
// the last one; nullptr: none wanted)
int launch_tile_bits_prepare(const TileTailMem &m, const int64_t *rows, const int64_t *d_n_rows, int64_t cap_rows, const BitsParams &bp,
                             void *scratch, int64_t *d_flags, const ScanState &ss, hipStream_t s);
int launch_tile_bits_finish(const TileTailMem &m, const int64_t *rows, const int64_t *d_n_rows, int64_t cap_rows, const BitsParams &bp,
                            const BitsOut &o, void *scratch, const ScanState &ss, hipStream_t s);
// the segments of a streamed pass (see "Segments" in pulse_table.hip): r / e / m / bp / o / scratch / ss as launch_tile_rows and
// launch_tile_bits take them for the WHOLE capture, with r.d_n_rows = &state->rows_at[last rows segment], r.d_n_rows_needed =
// &state->rows_needed, r.d_n_acc = &state->n_acc; m.epoch and m.parity the same for every segment of the pass
int launch_rows_segment(const ResolveArgs &r, const EmitArgs &e, const TileTailMem &m, const BitsParams &bp, SegState *state, const RowsSegment &sg,
                        hipStream_t s);
int launch_bits_segment(const TileTailMem &m, const BitsParams &bp, const BitsOut &o, void *scratch, const ScanState &ss, int64_t *rows,
                        int64_t cap_rows, const BitsSegment &sg, const SegPackDst *dst, hipStream_t s);
// ASK, sharded: summary of the locally merged table {n_rows, first state, first length, last state, last length}
void launch_merge_summary(const int64_t *rows, const int64_t *d_n_rows, int64_t *d_out5, hipStream_t s);
// ASK, sharded: merge equal-state rows across shard boundaries (d_all: world x 5 int64)
void launch_merge_fix(int64_t *rows, const int64_t *d_n_rows, const int64_t *d_all, int rank, int world,
                      int64_t *d_absorbed, hipStream_t s);
// sharded: d_extra[0..1] from every rank's flags (d_all: world x 3 int64)
void launch_bits_extra(const int64_t *d_all, int rank, int world, int32_t *d_extra, hipStream_t s);

// ---- filters.hip ---------------------------------------------------------------------------------------
size_t fir_stats_scratch_bytes(int64_t n);
size_t fir_work_bytes(int64_t n, int m);
int launch_fir(const float2 *x, int64_t n, const float2 *taps, int m, const float2 *halo, float2 *out, hipStream_t s, void *work = nullptr,
               int64_t chunk = 0, int64_t n_chunks = 0, double *d_sum = nullptr, double *d_max = nullptr, void *tile_scratch = nullptr);
int launch_iir(const double *a, int64_t M, const double *b, int64_t N, const float2 *x, int64_t n, float2 *y, hipStream_t s);
int launch_magnitudes(const void *iq, int dtype, int64_t n, double *out, hipStream_t s);
size_t mag_chunk_scratch_bytes(int64_t n_chunks);
int launch_mag_chunk_stats(const void *iq, int dtype, int64_t n, int64_t chunk, int64_t n_chunks, double *d_sum, double *d_max,
                           void *scratch, hipStream_t s);
// ---- estimators.hip ----------------------------------------------------------------------------------------
size_t compact_scratch_bytes(int64_t n);
int launch_compact_gt(const float *x, int64_t n, const int64_t *d_n, float thr, float *out, int64_t *d_count, void *scratch,
                      int32_t *tickets, hipStream_t s);
int launch_compact_edges(const float *x, int64_t n, const int64_t *d_n, float center, int64_t *out, int64_t cap, int64_t *d_count,
                         void *scratch, int32_t *tickets, hipStream_t s);
size_t minmax_scratch_bytes();
int launch_minmax(const float *x, int64_t n, float *d_out2, void *scratch, hipStream_t s);
int launch_minmax_any(const void *x, int dtype, int64_t n, void *d_out2, void *scratch, hipStream_t s);
int launch_above_flags(const void *mag, int is_f64, int64_t n, float thr, float *flags, hipStream_t s);
int launch_median_filter(const double *data, int64_t n, int k, float *out, hipStream_t s);
constexpr int kDtAboveFlags = 100;       // launch_message_ranges: `d_iq` is a float32 array of 0 / 1 above-noise flags (launch_above_flags)
int pairwise_sum_f32(urhgpu_ctx *ctx, const float *d_x, int64_t n, int mode, float mean, float *out);
int launch_hist_edges(const float *x, int64_t n, const double *d_edges, int n_edges, int64_t *d_counts, hipStream_t s);
// ---- msg_ranges.hip ------------------------------------------------------------------------------------------
struct SegCtl;
extern bool g_force_merge_ambiguous;    // test hook
int launch_message_ranges(const int64_t *d_rows, const int64_t *d_n_rows, int64_t cap_rows, const void *d_iq, int dtype, int64_t n, float thr,
                          int ook_merge, int64_t *d_seg, int64_t *d_msgs, int64_t cap, SegCtl *d_ctl, void *scratch, hipStream_t s);
size_t seg_scratch_bytes(int64_t cap_rows, int64_t cap);
size_t seg_ctl_bytes();
void seg_ctl_read(const void *host_copy, int64_t *n_seg, int64_t *n_msgs, int *ambiguous);
// ---- compact.hip --------------------------------------------------------------------------------------------------
int launch_pack_blob(const urhgpu_outputs *o, int write_pos, hipStream_t s);
void launch_copy_shape(const float *in, float *out, int64_t n_samples, int shape, hipStream_t s);
// ---- costas.hip -----------------------------------------------------------------------------------------------
size_t costas_scratch_bytes(int64_t n);
int launch_costas(urhgpu_ctx *ctx, const void *d_iq, int64_t n, const urhgpu_params *p, float *d_qad, void *scratch);

}  // namespace urh
