/*
 * urhgpu.h -- C ABI of the MI355X-native IQ->bits path (liburhgpu.so).
 *
 * This is the drop-in boundary for the reference's native layer: every entry point replaces one
 * Python-visible function of the reference's Cython modules (urh.cythonext.signal_functions /
 * util / auto_interpretation) or the one pure-Python tail function that sits on the hot path
 * (ProtocolAnalyzer._ppseq_to_bits).  Reference citations are relative to /root/reference.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; every function returns an int status
 *     (URHGPU_OK == 0, negative == error, see urhgpu_strerror); nothing throws across the ABI.
 *   - "host" entry points take host pointers (numpy buffers): they upload, run the kernels and
 *     download, synchronously.  They mirror the reference functions 1:1 (same argument meaning,
 *     same edge cases) and are what the ctypes shim urh_amd/signal_functions.py calls.
 *   - "_dev" entry points take DEVICE pointers (hipMalloc / torch tensors' data_ptr()) and are
 *     asynchronous on the context's stream; nothing is copied to the host unless stated.  They
 *     are what a device-resident pipeline (urh_amd/pipeline.py, bench.py) calls.
 *   - Inputs are borrowed and never written; outputs are caller-allocated; device scratch is
 *     owned by the context and grows on demand (never inside a steady-state call).
 *   - dtype codes follow the reference's fused `iq` type (src/urh/cythonext/util.pxd:1-8).
 *   - Not thread-safe per context; use one context per host thread (fork/spawn safe: HIP is
 *     initialised lazily by urhgpu_ctx_create in the calling process).
 */
#ifndef URHGPU_H
#define URHGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define URHGPU_VERSION 100 /* 0.1.0 */

/* status codes */
#define URHGPU_OK 0
#define URHGPU_ERR_HIP (-1)          /* a HIP runtime call failed (urhgpu_last_hip_error) */
#define URHGPU_ERR_DTYPE (-2)        /* reference: ValueError("Unsupported dtype") signal_functions.pyx:283,354 */
#define URHGPU_ERR_ARG (-3)          /* bad argument (null pointer, negative size, misaligned device pointer) */
#define URHGPU_ERR_CAPACITY (-4)     /* caller-provided output capacity too small; *n_* holds the needed size */
#define URHGPU_ERR_UNSUPPORTED (-5)  /* parameter outside the supported range (e.g. bits_per_symbol > 7) */
#define URHGPU_ERR_NO_DEVICE (-6)    /* no usable GPU: the product path never falls back to the CPU */

/* dtype codes: element type of the interleaved (N,2) IQ array (util.pxd:1-8) */
#define URHGPU_DT_I8 0
#define URHGPU_DT_U8 1
#define URHGPU_DT_I16 2
#define URHGPU_DT_U16 3
#define URHGPU_DT_F32 4

/* modulation codes (Signal.modulation_type strings; signal_functions.pyx:31-44) */
#define URHGPU_MOD_ASK 0
#define URHGPU_MOD_FSK 1
#define URHGPU_MOD_PSK 2
#define URHGPU_MOD_OTHER 3 /* "OQPSK"/"QAM"/...: no demod branch in afp_demod (result stays 0) */
#define URHGPU_MOD_OQPSK 4 /* urhgpu_modulate* only: get_oqpsk_bits + PSK + half-symbol blanking (signal_functions.pyx:118-121, 165-169) */

typedef struct urhgpu_ctx urhgpu_ctx;

/* ---- context ---------------------------------------------------------------------------------- */
int urhgpu_version(void);
const char *urhgpu_strerror(int status);
const char *urhgpu_last_hip_error(void);
int urhgpu_device_count(int *count);
/* device = HIP ordinal.  Creates a private stream and an (initially empty) scratch arena. */
int urhgpu_ctx_create(int device, urhgpu_ctx **out);
int urhgpu_ctx_destroy(urhgpu_ctx *ctx);
/* Run on the caller's stream (hipStream_t passed as void*).  NULL is the HIP null stream -- what
 * torch calls the default stream -- NOT the private stream: work is then ordered with everything
 * else the caller enqueues there. */
int urhgpu_ctx_set_stream(urhgpu_ctx *ctx, void *hip_stream);
/* Go back to the context's private (non-blocking) stream. */
int urhgpu_ctx_use_private_stream(urhgpu_ctx *ctx);
int urhgpu_ctx_sync(urhgpu_ctx *ctx);
/* Pipelined mode for back-to-back passes (streaming one capture after the other): urhgpu_iq_to_bits_dev then runs its
 * hot kernel on the context's stream and everything after it (pulse table, bits: latency-bound kernels that leave the GPU
 * nearly empty) on a second stream with three scratch arenas in rotation, so that the NEXT pass's hot kernel overlaps this pass's tail.
 * tail_stream: a hipStream_t of the caller (e.g. a torch stream) or NULL for a private one.  In this mode the outputs of a
 * pass are complete only after urhgpu_ctx_join (the context's stream waits for the tail; the host does not block) or
 * urhgpu_ctx_sync; every other entry point joins first.  enable = 0 switches back (synchronises).
 * Measured on MI355X: 0.30 ms per 1 GiB pass against 0.34 ms one after the other; bench.py times this mode.
 * The hot kernel itself is launched on a private stream of the context (CU-masked, see "hot_cus_removed_per_xcd" below), ordered
 * behind what the caller has queued on the context's stream; it is a stream with default flags, i.e. it synchronises with the NULL
 * stream as every such stream does: callers that queue unrelated work on the NULL stream meanwhile serialise with the hot kernels.
 * A caller that runs more than two passes ahead of the GPU is held back on the HOST at the start of the next pass until the tail
 * that last used the pass's scratch arena has finished (bounded run-ahead).
 * The library reads no environment variable; tuning values of this mode are set with urhgpu_ctx_set_tuning. */
int urhgpu_ctx_set_pipelined(urhgpu_ctx *ctx, int enable, void *tail_stream);
/* Tuning values of the pipelined mode (the defaults are what is measured and shipped; tools/ab.sh sets others).  Nine keys; the knobs
 * earlier rounds measured as useless are gone (their records: profiles/HISTORY.md).
 *   "hot_lds_kb"               dynamic LDS per hot workgroup in KiB (fewer of them per CU); default 0
 *   "hot_lds_kb_sharded"       the same for the urhgpu_shard_* passes that keep the generic tail (ASK); default 33
 *   "hot_cus_removed_per_xcd"  0 .. 16, default 4; set before urhgpu_ctx_set_pipelined: the hot kernel of a pipelined pass runs on a private
 *                              stream whose CU mask leaves that many CUs of every XCD out -- on 224 of the MI355X's 256 CUs the kernel is 5 %
 *                              faster than on all of them, and the CUs left alone serve the previous pass's tail; 0: no mask
 *   "profile_bracket"          1: urhgpu_ctx_profile_* report the stream-level bracket, which reads 3-5 % longer than the kernel runs
 *   "stream_policy"            which tail a pass of urhgpu_stream_* takes.  5 (default): 6 (4 with "stream_latency") for passes that ship no
 *                              positions or ship them directly, 0 otherwise; 0: segments beside the hot kernel when nothing of an earlier pass is still running (one
 *                              capture, where the tail's latency counts), else the blob is packed at the end and copied by the copy engine;
 *                              1: every qualifying pass in segments; 2: never; 3: every qualifying pass DIRECT -- the ordinary tail behind the
 *                              hot kernel, whose kernels store rows and packed results into the pinned host blob themselves; 4: segments when
 *                              idle, direct otherwise; 6 (what 5 resolves to for back-to-back passes since round 6): STAGED -- the direct pass's
 *                              kernels store into a staging blob in HBM, one small kernel tightens it and the copy engine ships it with one
 *                              copy of the predicted size (kernel stores into pinned host memory beside the next hot kernel cost that kernel
 *                              10 us per GiB pass: profiles/r06b_skips.txt)
 *   "stream_latency"           1: under the default policy a pass that finds the pipeline idle -- ONE capture -- runs its tail in segments
 *                              (lowest latency); 0, default: direct passes throughout (highest throughput for back-to-back passes)
 *   "stream_segments"          rows segments of a segmented pass, 1 .. 16, default 7
 *   "stream_pos_direct"        1, default: direct passes ship bit_sample_pos themselves (uint32 stores of the kernels that compute them)
 *   "spin_wait"                1, default: the estimator calls poll their stream instead of blocking in the runtime (a blocking wait wakes
 *                              the host tens of microseconds late: 0.09 ms of config 3's estimate)
 *   "upload_pieces"            2 .. 16, default 4: pieces of urhgpu_stream_push_upload, the last one short
 *   "wide_int"                 1: one-shot and sharded passes over SIGNED INTEGER FSK captures take the hot kernel's instantiation with the
 *                              wide loop (phase steps beyond the fast loop's window, e.g. +-100 kHz at 1 MS/s: a quarter faster there, 5 %
 *                              slower on narrow captures); capture streams decide by themselves from a probe of their captures.  default 0
 *   "shard_summary_generic"    1: the local pass of urhgpu_shard_runs_dev as the three generic resolve launches instead of the one-launch
 *                              summary kernel (A/B and test use: the summaries are byte-equal).  default 0
 * Unknown key: URHGPU_ERR_ARG. */
int urhgpu_ctx_set_tuning(urhgpu_ctx *ctx, const char *key, int value);
int urhgpu_ctx_join(urhgpu_ctx *ctx);
/* Pre-size the scratch arena for captures of up to n samples with the given tolerance so that no
 * allocation happens inside later calls (bench / steady state). */
int urhgpu_ctx_reserve(urhgpu_ctx *ctx, int64_t n_samples, int tolerance);
/* Device properties the host side needs to size launches / report rooflines. */
int urhgpu_ctx_info(urhgpu_ctx *ctx, int *compute_units, int *wavefront, int64_t *hbm_bytes, char *name, int name_cap);

/* Run-time guard for the libm the reference would use on THIS host: the reference's Costas loop and FSK demodulation call the host's
 * sinf / cosf / atan2f (signal_functions.pyx:252-330, :375), which are not correctly rounded -- x86-64 glibc picks an FMA or a non-FMA
 * build of sinf / cosf by CPU, differing on about one float in 10^9 -- and the device code restates the FMA build.  out4 = {sinf / cosf
 * results compared with the host's (including every argument below 120 on which the two builds differ), mismatches, atan2f results
 * compared, mismatches}.  Mismatches != 0: the reference on this host is not the function the GPU path is bit-exact with.  Host
 * arithmetic; works without a GPU. */
int urhgpu_host_libm_check(int64_t *out4);

/* After a PSK demodulation (Costas loop) of more than 8192 samples: how the chunk chain of the exact parallel evaluation was
 * resolved -- out4 = {chunks whose true start state matched a speculative candidate, chunks evaluated serially until they
 * met a candidate's checkpoint, chunks evaluated serially to the end (or fully gated), re-speculation rounds}.
 * Synchronises the stream.  Diagnostics only: the output is exact in every case.
 * (The PSK path of urhgpu_afp_demod_dev / urhgpu_iq_to_bits_dev synchronises the stream itself: the host has to learn
 * whether the chunk chain closed before it launches the final pass.) */
int urhgpu_ctx_costas_stats(urhgpu_ctx *ctx, int32_t *out4);

/* Time the dominant kernel (demod + run segmentation) of subsequent fused / grab_pulse_lens calls with
 * HIP events on the context's stream: begin(max_records) arms up to max_records records (one per call); end()
 * synchronises the stream and returns the per-call durations in ms.  The bit-plane kernel's dispatch carries the start /
 * stop events itself (hipExtLaunchKernelGGL: the kernel's own begin / end timestamps, the duration rocprofv3 reports);
 * hot launches made of several kernels are bracketed by events recorded before and after them (urhgpu_ctx_set_tuning("profile_bracket",
 * 1): always report the bracket, which reads 3-5 % longer than the kernel runs). */
int urhgpu_ctx_profile_begin(urhgpu_ctx *ctx, int max_records);
int urhgpu_ctx_profile_end(urhgpu_ctx *ctx, float *ms_out, int cap, int *n_records);

/* ---- host-buffer entry points: 1:1 replacements of the reference's Cython functions ------------- */

/* util.get_magnitudes (src/urh/cythonext/util.pyx:128-136): out[i] = sqrt(I*I+Q*Q), float64[n]. */
int urhgpu_get_magnitudes(urhgpu_ctx *ctx, const void *iq, int dtype, int64_t n, double *out);

/* signal_functions.afp_demod (src/urh/cythonext/signal_functions.pyx:333-378).
 * n <= 2 -> zeros.  mod PSK runs the Costas loop (costa_demod, :252-330; out[0] is written as
 * NOISE(-4) here whereas the reference leaves it uninitialised).  noise_other is the NOISE sentinel
 * used when mod == URHGPU_MOD_OTHER (get_noise_for_mod_type, :31-44). */
int urhgpu_afp_demod(urhgpu_ctx *ctx, const void *iq, int dtype, int64_t n, float noise_mag, int mod,
                     int mod_order, float costas_loop_bandwidth, float noise_other, float *qad_out);

/* signal_functions.get_center_thresholds (:380-390): out has modulation_order-1 floats.  Host only. */
int urhgpu_get_center_thresholds(float center, float spacing, int modulation_order, float *out);

/* signal_functions.grab_pulse_lens (:392-495): qad float32[n] -> rows int64[n_rows][2] = [state, length]
 * (state -1 = pause).  rows_out has room for cap_rows rows; *n_rows receives the row count (also on
 * URHGPU_ERR_CAPACITY, so the caller can retry). */
int urhgpu_grab_pulse_lens(urhgpu_ctx *ctx, const float *qad, int64_t n, float center, uint16_t tolerance,
                           int mod, uint32_t samples_per_symbol, uint8_t bits_per_symbol, float center_spacing,
                           float noise_other, int64_t *rows_out, int64_t cap_rows, int64_t *n_rows);

/* ProtocolAnalyzer._ppseq_to_bits (src/urh/signalprocessing/ProtocolAnalyzer.py:323-414), flat outputs:
 *   bits[0..n_bits)      one byte per bit, all messages back to back
 *   msg_off[0..n_msg]    offsets of each message in bits[]
 *   pauses[0..n_msg)     pause after each message, in samples
 *   pos[0..n_pos)        bit_sample_pos arrays back to back (only when write_pos != 0)
 *   pos_off[0..n_msg]    offsets of each message in pos[]
 * counts[3] receives {n_msg, n_bits, n_pos}. */
int urhgpu_ppseq_to_bits(urhgpu_ctx *ctx, const int64_t *rows, int64_t n_rows, int64_t samples_per_symbol,
                         int bits_per_symbol, int write_pos, int64_t pause_threshold,
                         uint8_t *bits, int64_t cap_bits, int64_t *msg_off, int64_t *pauses, int64_t cap_msg,
                         int64_t *pos, int64_t cap_pos, int64_t *pos_off, int64_t *counts);

/* signal_functions.fir_filter (:513-525): complex64 x[n] (*) complex64 taps[m], causal, zero history,
 * accumulation order of the reference's scatter loop. */
int urhgpu_fir_filter(urhgpu_ctx *ctx, const float *x, int64_t n, const float *taps, int64_t m, float *out);

/* The same on device memory (asynchronous).  d_left_halo: NULL = zero history (the reference), or DEVICE pointer
 * to the m - 1 samples that precede d_x[0] (sharded captures: the left neighbour's tail).
 * Limits (URHGPU_ERR_UNSUPPORTED beyond them, nothing is computed): the taps and one tile of input are staged in LDS, which
 * holds about 8 900 taps; bits_per_symbol <= 7 (modulation order <= 128) in every entry point that slices. */
int urhgpu_fir_filter_dev(urhgpu_ctx *ctx, const float *d_x, int64_t n, const float *d_taps, int64_t m,
                          const float *d_left_halo, float *d_out);
/* The same with the magnitude chunk statistics of the OUTPUT fused into the filter's epilogue (Signal.filter_range followed by
 * detect_noise_level, AutoInterpretation.py:60-91, in one pass over the samples): d_sum[k] / d_max[k] (device, float64) for chunk k
 * counted from the end, as urhgpu_magnitude_chunk_stats_dev computes them on a float32 capture. */
int urhgpu_fir_filter_stats_dev(urhgpu_ctx *ctx, const float *d_x, int64_t n, const float *d_taps, int64_t m, const float *d_left_halo,
                                float *d_out, int64_t chunk, int64_t n_chunks, double *d_sum, double *d_max);

/* Filter.apply_bandpass_filter (src/urh/signalprocessing/Filter.py:84-101; np.convolve(data, h, "same") at :98 and
 * Filter.fft_convolve_1d at :70-82 are the same centred linear convolution): complex64 x[n] (*) complex128 taps[m],
 *   out[i] = sum_{k=0}^{m-1} taps[k] * X(i + shift - k),  i in [0, n_out),  X = x extended by zeros,
 * accumulated in complex128 (fp64 FMAs, taps ascending) -- the reference's summation order is numpy's (BLAS / FFT) and
 * undefined, so this entry point is floating point with a tolerance, not bit-exact.  The caller designs the taps
 * (Filter.py:103-131, O(m) on the host) and passes shift = (min(n, m) - 1) / 2, n_out = max(n, m) for "same".
 * out: n_out complex128 values. */
int urhgpu_bandpass(urhgpu_ctx *ctx, const float *x, int64_t n, const double *taps, int64_t m, int64_t shift,
                    int64_t n_out, double *out);

/* The same on device memory (asynchronous).  d_left / d_right: NULL or DEVICE pointers to n_left samples preceding d_x[0] /
 * n_right samples following d_x[n-1] (sharded captures: the neighbours' edges) instead of zeros.  out_c64 != 0 writes
 * complex64 (the cast of SignalFrame.py:1578-1580 fused into the store), else complex128. */
int urhgpu_bandpass_dev(urhgpu_ctx *ctx, const float *d_x, int64_t n, const double *d_taps, int64_t m, int64_t shift,
                        int64_t n_out, const float *d_left, int64_t n_left, const float *d_right, int64_t n_right,
                        void *d_out, int out_c64);

/* signal_functions.iir_filter (:527-542). */
int urhgpu_iir_filter(urhgpu_ctx *ctx, const double *a, int64_t na, const double *b, int64_t nb,
                      const float *x, int64_t n, float *out);

/* ---- device-resident entry points ------------------------------------------------------------------ */

/* Demodulation + digitization parameters (Signal.py:42-109 defaults in comments). */
typedef struct urhgpu_params {
    int dtype;                  /* URHGPU_DT_* of the IQ buffer */
    int mod;                    /* URHGPU_MOD_* */
    int bits_per_symbol;        /* 1 */
    float noise_threshold;      /* Signal.noise_threshold (magnitude, not squared) */
    float center;               /* 0.02 */
    float center_spacing;       /* 1.0 */
    int tolerance;              /* 5 */
    uint32_t samples_per_symbol; /* 100 */
    float costas_loop_bandwidth; /* 0.1 */
    int64_t pause_threshold;    /* 8 */
    int write_bit_sample_pos;   /* 1 in the reference (get_protocol_from_signal) */
    float noise_other;          /* NOISE sentinel for URHGPU_MOD_OTHER */
    int mod_order;              /* 0 = 2^bits_per_symbol; afp_demod's mod_order argument (Costas loop order) */
} urhgpu_params;

/* Output descriptor of the fused path.  All pointers are DEVICE pointers owned by the caller; any
 * of qad / pos may be NULL (not materialised).  counts (device int64[5]) receives
 * {n_rows, n_msg, n_bits, n_pos, n_rows_needed}; the caller reads it back (40 bytes) when it needs the sizes.
 * n_rows is clamped to cap_rows; a table that did not fit shows as n_rows == cap_rows. */
typedef struct urhgpu_outputs {
    float *qad;            /* float32[n]  demodulated signal (Signal.qad) or NULL */
    int64_t *rows;         /* int64[cap_rows][2] pulse table (grab_pulse_lens result) */
    int64_t cap_rows;
    uint8_t *bits;         /* uint8[cap_bits] */
    int64_t cap_bits;
    int64_t *msg_off;      /* int64[cap_msg+1] */
    int64_t *pauses;       /* int64[cap_msg] */
    int64_t cap_msg;
    int64_t *pos;          /* int64[cap_pos] or NULL */
    int64_t cap_pos;
    int64_t *pos_off;      /* int64[cap_msg+1] */
    int64_t *counts;       /* int64[5]: {n_rows, n_msg, n_bits, n_pos, rows the table needed (> cap_rows: it was truncated)} */
    /* Optional compact mirror of the results for the trip over PCIe (NULL: not produced): one contiguous blob written by one more
     * kernel at the end of the pass -- see "compact result blob" below.  cap_blob >= urhgpu_blob_capacity(...). */
    void *blob;            /* device, 16-byte aligned */
    int64_t cap_blob;
    /* Optional: pinned HOST memory (hipHostMalloc / torch pin_memory; device-accessible) that receives the five counts as well, stored
     * by the kernel that finalises them -- valid once the pass has completed (an event recorded behind it).  Saves the 40-byte D2H copy
     * a streaming consumer would otherwise queue behind every pass (a copy packet costs the tail chain about 13 us). */
    int64_t *h_counts;
} urhgpu_outputs;

/* ---- compact result blob ---------------------------------------------------------------------------------------------------
 * The wide outputs above mirror the reference's Python objects (grab_pulse_lens' int64 [state, length] rows, one byte per bit,
 * int64 bit_sample_pos): 22.8 MB per GiB of 2-FSK capture at 100 samples per symbol -- 0.5 ms of PCIe, more than the device pass.
 * The blob holds the same information in 8.9 MB (3.5 MB without positions); every section starts 16-byte aligned:
 *   int64 header[16] = {URHGPU_BLOB_MAGIC, n_rows, n_msg, n_bits, n_pos, rows_needed, total_bytes (negative: the blob was too small),
 *                       has_pos, off_pauses, off_msg_off, off_pos_off, off_row_state, off_bits, off_row_len, off_pos32, truncated}
 *   int64 pauses[n_msg]; int64 msg_off[n_msg + 1] (bit offsets); int64 pos_off[n_msg + 1];
 *   int8  row_state[n_rows]   (-1 = pause; grab_pulse_lens' column 0)
 *   uint8 bits[(n_bits + 7) / 8]   eight bits per byte, most significant first (numpy.packbits / unpackbits order)
 *   int32 row_len[n_rows]     (grab_pulse_lens' column 1; captures of up to 2^31 - 1 samples)
 *   (row_state -128: URHGPU_ROW_ABSORBED -- the first row of a rank's piece of a sharded ASK capture that was merged into the previous rank's last row)
 *   uint32 pos32[n_pos]       (bit_sample_pos; absent when the pass wrote no positions)
 * header[7] holds flags: bit 0 has_pos; bit 1 (URHGPU_BLOB_LEN16; staged passes of urhgpu_stream_*, round 6): the row_len section holds
 *   uint16 row_len16[n_rows] instead -- 3 bytes per pulse-table row over PCIe instead of 5 --, a length that does not fit (65535 and more,
 *   or negative) is stored as 0xFFFF and listed behind the packed bits, at the next 16-byte boundary: int64 n_esc, then n_esc pairs
 *   {uint32 row, int32 length} (a capture of n samples has at most n / 65535 + 2 of them)
 *   bit 2 (URHGPU_BLOB_ROW16; the same passes when the pulse table is dense -- more than one row per 64 samples in the pass before): there is
 *   no row_state section; the row_len section holds uint16 row16[n_rows] = (row_state + 1) << 13 | length -- 2 bytes per row --, a length
 *   of 8191 samples and more (or negative) is stored as 0x1FFF and listed at offset header[11] of the blob (the slot names the row_state
 *   section otherwise): int64 n_esc, then n_esc pairs {uint32 row, int32 length} (at most n / 8191 + 2)
 * truncated != 0: bit 0: a capacity was exceeded (rows_needed > cap_rows, or more messages / bits / positions than fit): the sections
 * hold what fitted, the caller repeats the pass with larger capacities; bit 2: a row length did not fit int32, a position uint32 or a
 * state int8 (captures of 2^31 samples and more; sharded captures whose positions are absolute): use the wide outputs; bit 1: a
 * streamed pass's segment gave up waiting for the hot kernel (urhgpu_stream_* reports that as an error).  The counts come first (40 bytes from `counts`), then ONE copy of
 * total_bytes moves everything; urhgpu_stream_* below does that overlapped with the following passes. */
#define URHGPU_BLOB_MAGIC INT64_C(0x55524842424C4F42) /* "URHBBLOB" */
#define URHGPU_BLOB_HEADER_BYTES 128
#define URHGPU_BLOB_LEN16 2      /* header[7] bit 1: 16-bit row lengths + escape list (see above) */
#define URHGPU_BLOB_ROW16 4      /* header[7] bit 2: state and length of a row in one uint16 + escape list (see above) */
int64_t urhgpu_blob_capacity(int64_t cap_rows, int64_t cap_bits, int64_t cap_msg, int64_t cap_pos, int has_pos);

/* The results of a pass that is over, on the host through the compact blob: out = the descriptor the pass was given with out->blob /
 * cap_blob naming a device blob (the pass itself need not have packed it); pack kernel, header, ONE copy of header.total_bytes into
 * host_dst (pinned memory: at PCIe speed).  *total_bytes = the blob's size (also on URHGPU_ERR_CAPACITY: cap_dst too small).
 * Synchronous.  What the boundary objects fetch (urh_amd.pipeline.BitsResult.host(), urh_amd.signal.Signal.bits()) instead of the
 * wide int64 tables; reference shape: ProtocolAnalyzer.py:227-287, :323-414. */
int urhgpu_outputs_to_host(urhgpu_ctx *ctx, const urhgpu_outputs *out, int write_pos, void *host_dst, int64_t cap_dst, int64_t *total_bytes);

/* afp_demod on device memory (ASK/FSK/OTHER: one streaming kernel; PSK: Costas loop). */
int urhgpu_afp_demod_dev(urhgpu_ctx *ctx, const void *d_iq, int64_t n, const urhgpu_params *p, float *d_qad);

/* grab_pulse_lens on device memory: d_qad float32[n] -> d_rows; d_n_rows is a device int64. */
int urhgpu_grab_pulse_lens_dev(urhgpu_ctx *ctx, const float *d_qad, int64_t n, const urhgpu_params *p,
                               int64_t *d_rows, int64_t cap_rows, int64_t *d_n_rows);

/* _ppseq_to_bits on device memory (rows -> bits/pauses/positions); uses out->{bits..counts}. */
int urhgpu_ppseq_to_bits_dev(urhgpu_ctx *ctx, const int64_t *d_rows, const int64_t *d_n_rows, int64_t cap_rows_hint,
                             const urhgpu_params *p, const urhgpu_outputs *out);

/* THE fused hot path: IQ (device) -> [qad] -> pulse table -> bits, everything device resident.
 * ASK/FSK run demodulation and run segmentation in ONE pass over the IQ stream (8 B read +
 * 4 B qad write per sample); PSK runs the Costas kernel and then the qad-input segmentation.
 * msg_off[m] .. msg_off[m + 1] delimit message m in bits[] (msg_off[0] = 0); pos_off likewise.
 * Sharded captures use the urhgpu_shard_* phases below instead. */
int urhgpu_iq_to_bits_dev(urhgpu_ctx *ctx, const void *d_iq, int64_t n, const urhgpu_params *p,
                          const urhgpu_outputs *out);

/* ---- a stream of captures, results on the host -------------------------------------------------------------------------------
 * SURVEY.md §8(d)'s window for this path ends with the compact outputs ON THE HOST.  For capture after capture (the reference's live
 * mode is such a consumer: ProtocolSniffer.py:161-202) three things overlap: the hot kernel of pass i, the tail of pass i - 1 (second
 * stream, urhgpu_ctx_set_pipelined: the stream switches the context to that mode) and the pack kernel + D2H copy of pass i - 2's blob
 * (third stream / copy engine, pinned memory owned by the stream).  Three output slots rotate.
 *   urhgpu_stream_create   n_max: largest capture (samples, < 2^31); p: demodulation + slicing parameters of every pass (ASK / FSK /
 *                          OTHER; PSK synchronises with the host inside the Costas loop: URHGPU_ERR_UNSUPPORTED); want_qad: the
 *                          demodulated signal is materialised (stays in HBM: urhgpu_host_result::d_qad); want_pos: bit_sample_pos
 *                          is produced and shipped; cap_rows: 0 = the default (urhgpu_stream_capacities), else the pulse-table capacity.
 *   urhgpu_stream_push     queue pass i on d_iq (device; must stay valid until the pass has run), its pack kernel and its D2H copy and
 *                          return WITHOUT waiting for any of it (the copy's size is predicted from the pass before; what a prediction
 *                          misses is fetched when the result is handed out); *ready receives the result of pass i - 3 (seq = -1: none
 *                          yet), valid until three pushes later.
 *   urhgpu_stream_flush    wait for every outstanding pass; out3 receives up to three results, oldest first.
 * The pointers of a result are pinned host memory owned by the stream. */
typedef struct urhgpu_stream urhgpu_stream;
typedef struct urhgpu_host_result {
    int64_t seq;                 /* index of the push this result belongs to */
    int64_t n_samples;
    int64_t n_rows, n_msg, n_bits, n_pos, rows_needed;
    int64_t blob_bytes;          /* bytes that crossed PCIe for this pass (+ 40 for the counts) */
    int truncated;               /* see "compact result blob" */
    const int32_t *row_len;      /* pulse table: grab_pulse_lens' [state, length] columns; NULL when the blob carries 16-bit lengths: */
    const uint16_t *row_len16;   /* ... then these (URHGPU_BLOB_LEN16), 0xFFFF = look the row up in esc */
    const void *esc;             /* n_esc pairs {uint32 row, int32 length} */
    int64_t n_esc;
    const int8_t *row_state;
    const uint8_t *bits_packed;  /* numpy.unpackbits(bits_packed)[msg_off[m] : msg_off[m + 1]] = message m */
    const int64_t *msg_off, *pauses, *pos_off;
    const uint32_t *pos32;       /* bit_sample_pos or NULL */
    const void *blob;            /* the whole blob (header first) */
    const float *d_qad;          /* DEVICE: the pass's demodulated signal (overwritten three pushes later) or NULL */
    const uint16_t *row16;       /* URHGPU_BLOB_ROW16: (row_state + 1) << 13 | length per row, 0x1FFF = look the row up in esc; row_len, row_len16 and
                                    row_state are NULL then */
} urhgpu_host_result;
int urhgpu_stream_capacities(int64_t n_max, const urhgpu_params *p, int64_t *cap_rows, int64_t *cap_bits, int64_t *cap_msg, int64_t *cap_pos);
int urhgpu_stream_create(urhgpu_ctx *ctx, int64_t n_max, const urhgpu_params *p, int want_qad, int want_pos, int64_t cap_rows, urhgpu_stream **out);
int urhgpu_stream_destroy(urhgpu_stream *st);
int urhgpu_stream_push(urhgpu_stream *st, const void *d_iq, int64_t n, urhgpu_host_result *ready);
/* The same for a capture that is still ON THE HOST (the reference: IQArray.from_file, IQArray.py:206-227 -> Signal.py:111-112, then
 * ProtocolAnalyzer.get_protocol_from_signal): h_iq (host; pinned memory for PCIe speed) is copied into d_iq (device, n samples of the
 * stream's dtype, the caller's: the capture stays resident there) in PIECES, and every piece is demodulated as it lands -- hot kernel
 * on the piece, the piece's share of the tail, its share of the compact blob stored into pinned host memory.  1 GiB takes some 20 ms
 * over PCIe and 0.3 ms to demodulate: "file in host memory -> bits on the host" costs the upload plus the last piece's kernels.
 * Captures the segmented path does not take (ASK, a partial tile at the end, too short) are uploaded in one copy in front of an
 * ordinary pass.  The copies are ordered behind what the caller has queued on the context's stream; h_iq must stay valid and unchanged, and d_iq
 * untouched, until the pass's result has been handed out (a later push's `ready`, or urhgpu_stream_flush). */
int urhgpu_stream_push_upload(urhgpu_stream *st, const void *h_iq, void *d_iq, int64_t n, urhgpu_host_result *ready);
int urhgpu_stream_flush(urhgpu_stream *st, urhgpu_host_result *out3, int *n_out);
/* Diagnostics: out4 = {passes pushed, passes whose predicted copy size fell short (their rest was fetched when the result was handed
 * out), bytes the next copy is sized for, blob capacity}. */
int urhgpu_stream_stats(urhgpu_stream *st, int64_t *out4);
/* passes of the stream whose hot kernel was the instantiation with the wide loop for integer captures (signed integer FSK streams probe their
 * captures -- the share of phase steps beyond atan(7/16) per sample -- and pick it from 1 % on; 0 for every other stream) */
int urhgpu_stream_wide_passes(urhgpu_stream *st, int64_t *n_passes);

/* ---- sharded captures: one long capture split sample-contiguously over the GPUs of a node ------------------
 * (there is no reference counterpart: the reference processes a capture in one process.)
 * Each rank calls the four phases below in order on its own context; between the phases the CALLER
 * all-gathers a few bytes per rank (urh_amd/sharding.py does it with torch.distributed = RCCL over xGMI):
 *
 *   halo    : the last 2 IQ samples of every shard                          -> d_left_halo of the next rank
 *             (not exchanged at all when whoever distributed the capture gave every rank the two samples before its shard:
 *             urhgpu_shard_launch_dev; a pass then needs two all-gathers, ASK three)
 *   runs    : hot kernel on the shard + its 72-byte summary (d_summary out) -> all-gather -> d_summaries
 *   rows    : this rank's pulse-table rows; ASK: d_merge (5 x int64) out    -> all-gather -> d_merge_all
 *   prepare : cross-shard ASK merge, per-row scan; d_flags (3 x int64) out  -> all-gather -> d_flags_all
 *   finish  : bits / pauses / bit_sample_pos of this rank's rows
 *
 * The result stays sharded: `out` (given to urhgpu_shard_runs_dev, filled by the later phases) holds the rows
 * that END in this shard and what they expand to; concatenating the ranks' pieces in rank order gives exactly
 * the single-GPU result.  Differences to urhgpu_iq_to_bits_dev's outputs:
 *   - rows[0] may carry state URHGPU_ROW_ABSORBED (ASK: the row merged into the previous rank's last row;
 *     skip it when concatenating);
 *   - msg_off[m + 1] / pos_off[m + 1] are the LOCAL end offsets of the m-th message that closes on this rank;
 *     bits / pos may continue past the last one (the message closes on a later rank);
 *   - counts = {local rows, messages closed here, local bits, local positions}.
 * out->blob (FSK / other, not ASK: URHGPU_ERR_UNSUPPORTED): the finish phase also packs this rank's piece into the compact blob.
 * PSK (Costas loop) does not shard: URHGPU_ERR_UNSUPPORTED. */
#define URHGPU_ROW_ABSORBED (-(INT64_C(1) << 62))
#define URHGPU_SHARD_SUMMARY_BYTES 72 /* one shard summary (d_summary; d_summaries = world of them, back to back) */
int urhgpu_shard_runs_dev(urhgpu_ctx *ctx, const void *d_iq, int64_t n_local, int64_t pos_base, int64_t n_total,
                          int rank, int world, const void *d_left_halo, const urhgpu_params *p,
                          const urhgpu_outputs *out, void *d_summary);
/* Optional: start the hot kernel BEFORE the halo has arrived -- every chunk but the first, which alone reads it -- so that
 * the halo all-gather overlaps the kernel; urhgpu_shard_runs_dev (same arguments + the halo) then only adds the first chunk.
 * On a pipelined context (urhgpu_ctx_set_pipelined) that first chunk and every later phase run on the TAIL stream: d_left_halo
 * must be ready in tail-stream order (make the tail stream, not the context's stream, wait for the halo exchange), and the
 * context's stream never waits for a collective. */
int urhgpu_shard_prelaunch_dev(urhgpu_ctx *ctx, const void *d_iq, int64_t n_local, int64_t pos_base, int64_t n_total,
                               int rank, int world, const urhgpu_params *p, const urhgpu_outputs *out);
/* The halo is known up front (d_left_halo: the two samples before the shard, NULL on rank 0): the whole hot launch, on the context's
 * stream (pipelined contexts: their hot stream); urhgpu_shard_runs_dev (same arguments) then only adds the summary. */
int urhgpu_shard_launch_dev(urhgpu_ctx *ctx, const void *d_iq, int64_t n_local, int64_t pos_base, int64_t n_total,
                            int rank, int world, const void *d_left_halo, const urhgpu_params *p, const urhgpu_outputs *out);
int urhgpu_shard_rows_dev(urhgpu_ctx *ctx, const void *d_summaries, int64_t *d_merge);
int urhgpu_shard_bits_prepare_dev(urhgpu_ctx *ctx, const int64_t *d_merge_all, int64_t *d_flags);
int urhgpu_shard_bits_finish_dev(urhgpu_ctx *ctx, const int64_t *d_flags_all);

/* Magnitude chunk statistics for AutoInterpretation.detect_noise_level
 * (src/urh/ainterpretation/AutoInterpretation.py:60-91): chunks of `chunk` samples taken from the END
 * of the capture backwards; d_sum[k] (float64) / d_max[k] (float64) for chunk k counted from the end. */
int urhgpu_magnitude_chunk_stats_dev(urhgpu_ctx *ctx, const void *d_iq, int dtype, int64_t n, int64_t chunk,
                                     int64_t n_chunks, double *d_sum, double *d_max);

/* ---- estimator passes: the O(N) parts of AutoInterpretation.estimate (src/urh/ainterpretation/AutoInterpretation.py:373-470);
 * the decisions on the few hundred resulting values stay on the host (urh_amd/estimators.py). ------------------------- */

/* auto_interpretation.segment_messages_from_magnitudes (src/urh/cythonext/auto_interpretation.pyx:55-111) as a pulse table:
 * state 1 = |sample| > noise_threshold, 0 = below, switched after 10 consecutive samples (tolerance 9), row layout and
 * length conventions of urhgpu_grab_pulse_lens_dev; the state machine starts in the state of sample 0.  The host turns
 * the rows into (start, end) tuples.  Integer captures: magnitudes as util.get_magnitudes computes them (C int sum, double sqrt). */
int urhgpu_segment_runs_dev(urhgpu_ctx *ctx, const void *d_iq, int dtype, int64_t n, float noise_threshold,
                            int64_t *d_rows, int64_t cap_rows, int64_t *d_n_rows);
/* segment_messages_from_magnitudes (auto_interpretation.pyx:55-111) and -- when n_merged_out is not NULL --
 * merge_message_segments_for_ook (AutoInterpretation.py:107-148) entirely on the device: state table (urhgpu_segment_runs_dev),
 * prefix sum -> (start, end) of every segment, outlier-free minimum pulse length, cuts at pauses >= 8 x that.  Only ranges cross
 * PCIe: seg_out = HOST int64[cap_seg_out][2] receives the first min(*n_seg_out, cap_seg_out) segments (AutoInterpretation.estimate
 * looks at the first 100 to tell the modulation), merged_out likewise the merged messages.  *merge_ambiguous = 1: a pulse length
 * lies within 1e-9 of the outlier bound mean +- std, where the order of the floating-point sum of squares decides -- the caller
 * then fetches every segment and takes that decision in numpy's order.  Synchronous. */
int urhgpu_message_ranges_dev(urhgpu_ctx *ctx, const void *d_iq, int dtype, int64_t n, float noise_threshold, int64_t *seg_out, int64_t cap_seg_out,
                              int64_t *n_seg_out, int64_t *merged_out, int64_t cap_merged_out, int64_t *n_merged_out, int *merge_ambiguous);
/* The same pass that also leaves afp_demod(iq, noise_threshold, "ASK") (signal_functions.pyx:343-378) in d_qad_ask (DEVICE float[n]):
 * AutoInterpretation.estimate segments by the noise threshold and then demodulates with it (AutoInterpretation.py:386-404) -- for
 * an OOK / ASK capture both read the same samples, so one pass over them serves both (8 B read + 4 B written per sample instead of
 * 8 + 8 + 4).  float32 / complex64 captures only (URHGPU_ERR_UNSUPPORTED otherwise, and for a NaN threshold). */
int urhgpu_message_ranges_demod_dev(urhgpu_ctx *ctx, const void *d_iq, int dtype, int64_t n, float noise_threshold, int64_t *seg_out,
                                    int64_t cap_seg_out, int64_t *n_seg_out, int64_t *merged_out, int64_t cap_merged_out,
                                    int64_t *n_merged_out, int *merge_ambiguous, float *d_qad_ask);
/* rect[rect > thr] (AutoInterpretation.py:227), order preserved; *d_count (device) = number kept. */
int urhgpu_compact_gt_dev(urhgpu_ctx *ctx, const float *d_x, int64_t n, float thr, float *d_out, int64_t *d_count);
/* positions i >= 1 where (x[i] <= center) != (x[i-1] <= center), ascending (get_plateau_lengths,
 * auto_interpretation.pyx:179-208); at most cap are stored, *d_count (device) = number found. */
int urhgpu_edges_le_dev(urhgpu_ctx *ctx, const float *d_x, int64_t n, float center, int64_t *d_idx, int64_t cap, int64_t *d_count);
/* util.minmax (util.pyx:20-36) of a float32 array: d_out2 = {min, max}. */
int urhgpu_minmax_f32_dev(urhgpu_ctx *ctx, const float *d_x, int64_t n, float *d_out2);
/* numpy's pairwise float32 sum (what np.mean / np.var reduce with): mode 0 sum x[i], mode 1 sum (x[i] - mean)^2.
 * Synchronous; *sum_out is a HOST float. */
int urhgpu_pairwise_sum_f32_dev(urhgpu_ctx *ctx, const float *d_x, int64_t n, int mode, float mean, float *sum_out);
/* np.histogram(x, bins=edges) for ascending float64 edges: d_counts[n_edges - 1]. */
int urhgpu_histogram_f32_dev(urhgpu_ctx *ctx, const float *d_x, int64_t n, const double *d_edges, int64_t n_edges, int64_t *d_counts);

/* AutoInterpretation.estimate's per-message statistics for ALL messages of a capture in one go (AutoInterpretation.py:392-446 walks
 * them one by one: detect_center :226-277 and get_plateau_lengths, auto_interpretation.pyx:179-208).  d_x: the demodulated signal
 * (device, float32[n]); ranges: HOST int64[n_msgs][2] = (start, end) of every message.  Synchronous; host outputs.
 *
 * urhgpu_msg_center_stats: per message rect = x[start:end][x > -4] trimmed to [int(0.05 k), int(0.95 k)), then
 * out_stats[m] = {kept k, trimmed length, min, max, float32 mean, float32 np.var (the bin width), n_edges, first edge} and
 * out_hist[m][0 .. n_edges - 2] = np.histogram(rect, bins = np.arange(min, max + var, var)); n_edges == 0: no histogram (empty
 * message, zero or NaN variance, fewer than two edges -> detect_center returns None); n_edges - 1 > max_bins: the histogram does not
 * fit the pool (the caller takes that message through urhgpu_histogram_f32_dev).  out_stats: double[n_msgs][8], out_hist:
 * int64[n_msgs][max_bins] or NULL (the histograms stay on the device).
 * The peak picking (AutoInterpretation.py:250-277) happens on the device too: out_flag[m] = 1: out_center[m] is detect_center's
 * result; 0: None; 2: too many bins (see above); 3: the second and third most populated peaks hold the same count, so the result
 * depends on how np.argsort orders equal keys -- the caller asks again with out_hist, for THOSE messages only, and lets numpy decide.
 * Both may be NULL.  ranges must be ascending and disjoint (start[m] >= end[m - 1]; URHGPU_ERR_ARG otherwise): per-message scratch
 * lives at [start, end) of capture-sized arrays.  The histogram pool is bounded: messages are processed in batches of at most
 * 64 MiB / (4 max_bins) (16 384 at max_bins = 4096), whatever n_msgs is. */
int urhgpu_msg_center_stats(urhgpu_ctx *ctx, const float *d_x, int64_t n, const int64_t *ranges, int n_msgs, int64_t max_bins,
                            double *out_stats, int64_t *out_hist, double *out_center, int32_t *out_flag);
/* urhgpu_msg_plateaus: get_plateau_lengths(x[start:end], centers[m], percentage) for every message with a center (NaN: none,
 * no plateaus): out_len[out_off[m] .. |out_off[m + 1]|) (uint64, back to back; out_off[0] = 0).  Boundaries are searched in the
 * first percentage % + extra_window samples of a message; when that window holds no boundary at or beyond the percentage mark the
 * message's end offset comes back as -(end + 1) and the caller repeats it with a larger window.  More than cap_total lengths:
 * URHGPU_ERR_CAPACITY with the needed total in out_off[n_msgs].  Messages longer than 2^31 - 1 samples: URHGPU_ERR_UNSUPPORTED. */
int urhgpu_msg_plateaus(urhgpu_ctx *ctx, const float *d_x, int64_t n, const int64_t *ranges, const double *centers, int n_msgs,
                        int percentage, int64_t extra_window, int64_t *out_off, uint64_t *out_len, int64_t cap_total);

/* auto_interpretation.merge_plateaus (auto_interpretation.pyx:145-176): host arrays, host arithmetic (sequential, a few thousand
 * values); out needs room for n values, *n_out = number of merged plateaus. */
int urhgpu_merge_plateaus(const uint64_t *plateaus, int64_t n, uint64_t tolerance, uint64_t max_count, uint64_t *out, int64_t *n_out);

/* AutoInterpretation.detect_modulation (AutoInterpretation.py:150-205; Wavelet.cwt_haar, Wavelet.py:15-43; median_filter,
 * auto_interpretation.pyx:213-240) for n_msgs messages of a complex64 capture on the device (d_iq: float32 pairs; ranges: HOST
 * int64[n_msgs][2]): labels_out[m] = 0 none, 1 OOK, 2 ASK, 3 FSK, 4 PSK; vars_out (or NULL): double[n_msgs][4] = the variances of
 * |CWT|, |CWT of the unit-magnitude samples| and of both after the median filter (NaN when the decision fell before them).
 * Floating-point classifier (double-precision radix-2 FFTs here, single-precision pocketfft forward transforms in the
 * reference): the label is the parity criterion.  wavelet_scale 4 and median_filter_order 11 are the reference's defaults.
 * Synchronous. */
int urhgpu_detect_modulation_dev(urhgpu_ctx *ctx, const float *d_iq, int64_t n, const int64_t *ranges, int n_msgs, int wavelet_scale,
                                 int median_filter_order, int *labels_out, double *vars_out);

/* The per-message decisions of AutoInterpretation.estimate after get_plateau_lengths (AutoInterpretation.py:416-433; tolerance :280-298,
 * merge_plateaus, round_plateau_lengths :313-326, divisor histogram, bit length :344-370) for every message in one call, host
 * arithmetic on the lengths urhgpu_msg_plateaus returned (same `off` convention).  tol_out[m]: tolerance or -1 (None);
 * bitlen_out[m]: bit length, -1 (fewer than two merged plateaus: no vote) or -2 (the reference's result depends on numpy's order of
 * equal counts: decide that message with numpy). */
int urhgpu_msg_bit_lengths(const uint64_t *lens, const int64_t *off, int n_msgs, int64_t *tol_out, int64_t *bitlen_out);
/* urhgpu_msg_plateaus and urhgpu_msg_bit_lengths in one call: the plateau lengths stay on the GPU, which counts every message's
 * distinct lengths (a few dozen per message against thousands of plateaus); the decisions that depend on the multiset only -- all of
 * them for a message without glitches (tolerance 0) -- are taken from those (value, count) pairs, and only messages with a positive
 * tolerance (merge_plateaus walks the sequence, auto_interpretation.pyx:145-176) have their sequences fetched.  Arguments as
 * urhgpu_msg_plateaus, results as urhgpu_msg_bit_lengths; tol_out[m] = bitlen_out[m] = -3: the search window of message m did not reach
 * the percentage mark (repeat that message through urhgpu_msg_plateaus with a larger window).  Synchronous. */
int urhgpu_msg_plateau_decisions(urhgpu_ctx *ctx, const float *d_x, int64_t n, const int64_t *ranges, const double *centers, int n_msgs,
                                 int percentage, int64_t extra_window, int64_t *tol_out, int64_t *bitlen_out);
/* urhgpu_msg_center_stats followed by urhgpu_msg_plateau_decisions with the centers it picked, in ONE call: the centers never leave the
 * device, the scratch of both stages is one reservation, one synchronisation ends the call (AutoInterpretation.py:397-433 per message).
 * out_stats[8 * n_msgs], out_center, out_flag as urhgpu_msg_center_stats gives them; tol_out / bitlen_out as urhgpu_msg_plateau_decisions,
 * -4 for a message whose center needs the host first (out_flag 2 or 3: settle it, then urhgpu_msg_plateau_decisions for that message).
 * URHGPU_ERR_UNSUPPORTED: n_msgs * max_bins counters do not fit one batch of the histogram pool (take the two calls). */
int urhgpu_msg_estimate(urhgpu_ctx *ctx, const float *d_x, int64_t n, const int64_t *ranges, int n_msgs, int64_t max_bins, int percentage,
                        int64_t extra_window, double *out_stats, double *out_center, int32_t *out_flag, int64_t *tol_out, int64_t *bitlen_out);
/* Test hook (host arithmetic): the multiset form of one message's decision; -3 in both where the sequence would be asked for. */
int urhgpu_test_bit_length_from_counts(const uint64_t *lens, int64_t n, int64_t *tol_out, int64_t *bitlen_out);
/* The two halves around np.argsort for a message urhgpu_msg_bit_lengths reported as -2 (equal counts in the divisor histogram: the
 * reference's result is whatever order np.argsort gives equal keys).  urhgpu_msg_divisor_histogram: tolerance, merged and rounded
 * plateaus (AutoInterpretation.py:280-326), then the dense uint64 histogram the reference sorts (auto_interpretation.pyx:113-143):
 * *hist_len = max value + 1 (cap = 0 only asks for it), -1 = fewer than two merged plateaus.  urhgpu_bit_length_from_order: the
 * selection loop of get_bit_length_from_plateau_lengths (AutoInterpretation.py:358-370) over order_desc = np.argsort(hist)[::-1]. */
int urhgpu_msg_divisor_histogram(const uint64_t *lens, int64_t n, uint64_t *hist_out, int64_t cap, int64_t *hist_len, int64_t *tol_out);
int urhgpu_bit_length_from_order(const uint64_t *hist, const int64_t *order_desc, int64_t len, int64_t *bitlen_out);

/* ---- host-array forms of the reference's remaining `util` / `auto_interpretation` functions on the path (the signatures
 * AutoInterpretation.py:8-10 imports by name; urh_amd/util.py and urh_amd/auto_interpretation.py bind them) --------------------- */
/* util.minmax (src/urh/cythonext/util.pyx:20-36): arr = n values of dtype (the fused `iq` element types); out2 = {min, max} in the
 * same type; n == 0: {0, 0}.  The reference's comparisons (a NaN never replaces min / max unless it is element 0). */
int urhgpu_minmax(urhgpu_ctx *ctx, const void *arr, int dtype, int64_t n, void *out2);
/* auto_interpretation.segment_messages_from_magnitudes (src/urh/cythonext/auto_interpretation.pyx:55-111) on caller-supplied
 * magnitudes (float32, or float64 as util.get_magnitudes returns them: is_f64): seg_out = int64[cap_seg][2] (start, end) tuples,
 * *n_seg their number (also on URHGPU_ERR_CAPACITY).  At most n / 20 + 3 segments exist. */
int urhgpu_segment_messages(urhgpu_ctx *ctx, const void *magnitudes, int is_f64, int64_t n, float noise_threshold, int64_t *seg_out,
                            int64_t cap_seg, int64_t *n_seg);
/* auto_interpretation.get_plateau_lengths (auto_interpretation.pyx:179-208): rect_data float32[n] (host) -> out uint64[cap], *n_out
 * lengths (also on URHGPU_ERR_CAPACITY). */
int urhgpu_get_plateau_lengths(urhgpu_ctx *ctx, const float *rect_data, int64_t n, float center, int percentage, uint64_t *out, int64_t cap,
                               int64_t *n_out);
/* auto_interpretation.get_threshold_divisor_histogram (auto_interpretation.pyx:113-143): dense uint64[max(lens) + 1]; cap = 0 only
 * asks for *hist_len.  Host arithmetic (sorting the multiset of values instead of the reference's P^2 / 2 pair loop). */
int urhgpu_threshold_divisor_histogram(const uint64_t *lens, int64_t n, float threshold, uint64_t *hist_out, int64_t cap, int64_t *hist_len);
/* auto_interpretation.median_filter (auto_interpretation.pyx:227-240): out[i] = sorted(float(data[i : min(i + k, n)]))[k' / 2];
 * k <= 64 (URHGPU_ERR_UNSUPPORTED beyond; the reference's callers use 3 .. 11). */
int urhgpu_median_filter(urhgpu_ctx *ctx, const double *data, int64_t n, unsigned int k, float *out);

/* Test hook: the hot kernel's fast-path division (Newton + residual chain without scaling) against the IEEE
 * division on 2^20 * reps pseudo-random operand pairs from the range the fast path accepts; *n_mismatch must be 0. */
int urhgpu_test_fast_division_dev(urhgpu_ctx *ctx, uint64_t seed, int reps, uint64_t *n_mismatch);
/* Test hook: the branch-free sinf / cosf pair of the Costas loop (glibc_sincosf.h: urh_sincosf_fast) against the branchy restatement
 * of glibc's sinf and cosf for EVERY float with |y| < 120; *n_mismatch = arguments where either result differs in a bit. */
int urhgpu_test_sincosf_fast_dev(urhgpu_ctx *ctx, uint64_t *n_mismatch);
/* Test hook: urhgpu_message_ranges_dev reports every OOK merge as borderline (*merge_ambiguous = 1), so that the caller's numpy
 * route for that case can be exercised; returns the previous setting. */
int urhgpu_test_force_merge_ambiguous(int on);

/* signal_functions.modulate_c (signal_functions.pyx:56-177) for ASK / FSK / PSK / OQPSK (URHGPU_MOD_OQPSK: bits_per_symbol must be 2; GFSK: urhgpu_modulate_gfsk*),
 * n_msgs messages rendered back to back by one launch (URH modulates message by message, Modulator.py:215-255):
 * message m has bits[bit_off[m] .. bit_off[m+1]), is followed by pause[m] zero samples and starts at sample index start[m]
 * (the time origin of its carrier).  parameters: 2^bits_per_symbol amplitudes / frequencies / phases as the reference takes
 * them.  dtype: URHGPU_DT_F32 / _I8 / _I16 (get_numpy_dtype, :46-54; anything else URHGPU_ERR_DTYPE).  All pointers are HOST
 * pointers except d_out (device, (cap_samples, 2) of dtype); *total_samples = sum over m of
 * (n_bits_m / bits_per_symbol) * samples_per_symbol + pause[m]; more than cap_samples: URHGPU_ERR_CAPACITY, nothing written.
 * Asynchronous on the context's stream. */
int urhgpu_modulate_dev(urhgpu_ctx *ctx, const uint8_t *bits, const int64_t *bit_off, const uint32_t *pause, const uint32_t *start,
                        int n_msgs, uint32_t samples_per_symbol, int mod, const float *parameters, int bits_per_symbol,
                        float carrier_amplitude, float carrier_frequency, float carrier_phase, float sample_rate, int dtype,
                        void *d_out, int64_t cap_samples, int64_t *total_samples);
/* One message, host output: exactly modulate_c's signature; out holds ((num_bits / bits_per_symbol) * samples_per_symbol + pause, 2). */
int urhgpu_modulate(urhgpu_ctx *ctx, const uint8_t *bits, int64_t num_bits, uint32_t samples_per_symbol, int mod,
                    const float *parameters, int bits_per_symbol, float carrier_amplitude, float carrier_frequency,
                    float carrier_phase, float sample_rate, uint32_t pause, uint32_t start, int dtype, void *out);

/* modulate_c(..., "GFSK", ...) (signal_functions.pyx:118-125, 156-163, 196-228).  gauss_fir: the n_taps Gaussian taps of
 * gauss_fir (:230-243), which the reference computes with numpy (host; urh_amd.signal_functions.gauss_fir restates it).
 * The per-sample symbol frequencies are convolved with the taps ("same" mode), every output an exactly accumulated dot product
 * rounded once to float32; the reference takes this value from numpy's float32 BLAS dot product, whose summation order (and last
 * bit) depends on the host CPU.  frequencies (HOST, optional): the filtered frequencies, one float per data sample of every
 * message back to back (e.g. numpy's convolution, for the reference's bits on this host) -- used instead of the device
 * convolution.  The phase recurrence (:219-226, numpy's float32 arange restated) and the carrier are evaluated exactly as
 * the reference does.  A message with fewer bits than one symbol: URHGPU_ERR_ARG (ZeroDivisionError in the reference, :201). */
int urhgpu_modulate_gfsk_dev(urhgpu_ctx *ctx, const uint8_t *bits, const int64_t *bit_off, const uint32_t *pause, const uint32_t *start,
                             int n_msgs, uint32_t samples_per_symbol, const float *parameters, int bits_per_symbol,
                             float carrier_amplitude, float carrier_phase, float sample_rate, int dtype, const float *gauss_fir,
                             int n_taps, const float *frequencies, void *d_out, int64_t cap_samples, int64_t *total_samples);
int urhgpu_modulate_gfsk(urhgpu_ctx *ctx, const uint8_t *bits, int64_t num_bits, uint32_t samples_per_symbol, const float *parameters,
                         int bits_per_symbol, float carrier_amplitude, float carrier_phase, float sample_rate, uint32_t pause,
                         uint32_t start, int dtype, const float *gauss_fir, int n_taps, const float *frequencies, void *out);

/* IQArray.convert_to (IQArray.py:127-203) on device memory: n VALUES (two per IQ sample) of src_dtype -> dst_dtype, any
 * pair of different URHGPU_DT_* codes, with the reference's wrapping / truncating numpy semantics.  Asynchronous. */
int urhgpu_convert_dev(urhgpu_ctx *ctx, const void *d_src, int src_dtype, void *d_dst, int dst_dtype, int64_t n);
/* Signal.__load_wav_file (Signal.py:114-173) on device memory: the PCM frames of a WAV file (d_raw: n_frames * channels * sample_width
 * bytes as wave.readframes returns them; sample_width 1: unsigned bytes, 2 / 3 / 4: signed little endian) -> float32 IQ d_out[n_frames][2]:
 * (sample - center) * (1 / max) in float64, rounded once to float32, as numpy evaluates the reference's expression; one channel: the
 * imaginary part is 0 (an already demodulated capture, :155-156), two: left -> real, right -> imag.  Also what a Flipper .sub capture goes
 * through once its run lengths are expanded to bytes 255 / 0 (:175-205).  Other channel counts / widths: URHGPU_ERR_ARG (the reference
 * raises ValueError, :133, :164).  Asynchronous. */
int urhgpu_pcm_to_iq_dev(urhgpu_ctx *ctx, const void *d_raw, int64_t n_frames, int channels, int sample_width, float *d_out);
/* The run lengths IQArray.export_to_sub writes (IQArray.py:275-304) from the uint8 conversion of a capture: values[i * stride] is sample
 * i's first component (HOST memory; stride 2 for an (N, 2) array).  runs_out[k] > 0: that many samples above 127, < 0: at or below; the
 * reference's walk is restated exactly (a run of ONE sample never ends: the samples that differ from it are dropped until its value
 * comes back).  *n_runs is set also on URHGPU_ERR_CAPACITY.  Host arithmetic; works without a GPU. */
/* Signal.estimate_frequency's core (Signal.py:578-601): the bin k of the largest |FFT| of n = 2^m complex64 samples (d_x: float32[n][2] on the
 * device), the smallest such k -- what np.argmax(np.abs(np.fft.fft(data))) returns whenever the peak stands out by more than float32 rounding
 * (single-precision butterflies like numpy's complex64 transform, twiddles rounded from float64).  n up to 2^26 (URHGPU_ERR_UNSUPPORTED
 * beyond; n not a power of two: URHGPU_ERR_ARG).  Synchronous: *peak_index is a host value. */
int urhgpu_fft_peak_dev(urhgpu_ctx *ctx, const float *d_x, int64_t n, int64_t *peak_index);
int urhgpu_sub_encode_runs(const uint8_t *values, int64_t n, int64_t stride, int64_t *runs_out, int64_t cap, int64_t *n_runs);
/* The plain numpy cast between float32 and one of the four integer sample types (no IQArray scaling): what Filter.apply_fir_filter
 * does to an integer capture before filtering (`tmp.real = input_signal[0::2]`, Filter.py:37-41: the raw values as float32) and what
 * IQArray.__setitem__ does with the filtered complex64 range (`self.real[key] = value.real`, IQArray.py:31-33: truncation toward zero
 * through int32, low bits kept).  Exactly one of the two dtypes is URHGPU_DT_F32.  Asynchronous. */
int urhgpu_astype_dev(urhgpu_ctx *ctx, const void *d_src, int src_dtype, void *d_dst, int dst_dtype, int64_t n);

/* path_creator.create_path's pass over the samples (path_creator.pyx:46-66): 1-D samples of dtype (the five IQ sample
 * types), stretches of samples_per_pixel samples from `start` (the last one ends at `end`); values[2k] / values[2k+1] =
 * minimum / maximum of stretch k exactly as the reference's sequential scan finds them (NaN and signed-zero behaviour
 * included).  values holds 2 * ceil((end - start) / samples_per_pixel) elements of dtype.  _dev: device pointers,
 * asynchronous; the host form takes the whole array (n samples) and returns when values is filled. */
int urhgpu_path_minmax_dev(urhgpu_ctx *ctx, const void *d_samples, int dtype, int64_t start, int64_t end,
                           int64_t samples_per_pixel, void *d_values);
int urhgpu_path_minmax(urhgpu_ctx *ctx, const void *samples, int dtype, int64_t n, int64_t start, int64_t end,
                       int64_t samples_per_pixel, void *values);

/* Spectrogram.stft + __calculate_spectrogram (Spectrogram.py:94-116, :158-164; util.arr2decibel, util.pyx:38-48) on device
 * memory: d_x complex64[n]; frames = max(1, (n - window_size) / hop + 1) frames of window_size samples (a power of two,
 * 8 .. 4096; samples at or beyond n read as zero: the reference's zero padding of captures shorter than one window),
 * d_window float64[window_size] (np.hanning by default), d_twiddles complex128[window_size / 2] = exp(-2 pi i m / window_size).
 * Exactly one output: d_stft complex128[frames * window_size] (the stft, divided by window_size) or d_db float32[frames *
 * window_size] (fftshift along frequency, complex64, 10 log10 |.|^2, fliplr).  Double-precision FFT like numpy's; floating
 * point, not bit-exact.  Asynchronous. */
int urhgpu_spectrogram_dev(urhgpu_ctx *ctx, const float *d_x, int64_t n, int window_size, int64_t hop, int64_t frames,
                           const double *d_window, const double *d_twiddles, double *d_stft, float *d_db);
/* Spectrogram.apply_bgra_lookup (Spectrogram.py:196-210, normalize=True): d_db float32 (frames, window_size) ->
 * d_image uint32 BGRA (window_size, frames) through d_colormap[n_colors]. */
int urhgpu_bgra_lookup_dev(urhgpu_ctx *ctx, const float *d_db, int64_t frames, int window_size, const uint32_t *d_colormap,
                           int n_colors, float data_min, float data_max, uint32_t *d_image);

/* Measurement hook (bench.py, SURVEY.md 8(d) "also measure an on-box copy-kernel ceiling and report both denominators"): a PURE COPY
 * on this GPU, timed with events over `reps` launches.  shape 0: the hot kernel's access structure without its arithmetic (one
 * workgroup of four wavefronts per 8192 samples, 16-byte non-temporal loads two rows ahead, 8-byte non-temporal stores: 8 B in + 4 B
 * out per sample, d_in float32[2 n], d_out float32[n]); shape 1: a plain grid-stride float4 copy of n float32 values (4 B in + 4 B
 * out per value); shape 2: shape 0 on the CU-masked stream the hot kernel of pipelined passes runs on (URHGPU_ERR_UNSUPPORTED on a context
 * without one).  n_samples a multiple of 8192.  Synchronous. */
int urhgpu_bench_copy_ceiling_dev(urhgpu_ctx *ctx, const float *d_in, float *d_out, int64_t n_samples, int shape, int reps, float *ms_per_copy);
/* Measurement hook (tools/boundary_probe.py -> profiles/r05_boundary_anatomy.txt): `launches` back-to-back launches of the hot kernel alone
 * (complex64 2-FSK, n a multiple of 2048, qad written, no tail) on the context's stream (stream_kind 0) or its CU-masked hot stream (1; a
 * pipelined context).  event_mode 0: plain launches; 1: a completion event attached to every dispatch as pipelined passes do; 2: the same
 * with hipEventDisableSystemFence | hipEventReleaseToDevice; 3: timing events on every dispatch (dur_ms[launches], gap_ms[launches - 1]:
 * the dispatches' own durations and the time from one's end to the next one's begin).  graded: that many of the launch's last chunks are
 * cut into four short ones each.  Wavefront 0 of every workgroup leaves three s_memrealtime stamps (10 ns units, low 32 bits: entry,
 * streaming phase over, ChunkInfo written) and its hardware ids in the ChunkInfo's first_acc / pend_stable / pad / pend_acc fields; the
 * tables of the last `keep` launches are copied to d_chunks_out (keep * *n_chunks_out * URHGPU_SHARD_SUMMARY_BYTES bytes).  Synchronous. */
int urhgpu_test_hot_probe(urhgpu_ctx *ctx, const void *d_iq, int64_t n, const urhgpu_params *p, float *d_qad, int stream_kind, int event_mode,
                          int graded, int launches, int keep, void *d_chunks_out, int64_t *n_chunks_out, float *dur_ms, float *gap_ms, int bubble_us,
                          int load_kind);
/* ... bubble_us > 0: a one-wavefront kernel that idles for that long between two hot kernels; load_kind != 0 (event_mode 1 or 2): synthetic
 * company on a second stream beside every hot kernel (capi.hip: arithmetic / random loads on the CUs the hot mask leaves out, thousands of
 * short high-priority workgroups, six empty kernels).  The same stamps inside the PRODUCT's passes
 * (tools/inrun_anatomy.py): urhgpu_test_hot_stamps(1) makes every complex64 2-FSK pass run the stamped instantiation (process-wide; the
 * stamps sit in ChunkInfo fields the tile tail does not read), urhgpu_test_fetch_chunk_tables copies the chunk tables of the context's three
 * most recent pipelined passes (newest first, n_chunks x URHGPU_SHARD_SUMMARY_BYTES bytes each) to host_dst.  Synchronous. */
int urhgpu_test_hot_stamps(int on);
/* Measurement hook: leave kernels of the tile tail out (bit 0 k_resolve_one, 1 k_emit_rows_tiles, 2 k_tile_scan, 3 group scan, 4
 * k_expand_tiles, 5 k_pack_seg) to see what each costs the hot kernel it runs beside; outputs are only meaningful while every pass
 * processes the same capture (the buffers then hold the previous pass's identical results).  Round 6: bit 6 the row kernel without its stores
 * into the (host or staging) blob, 7 without the int64 table, 9 a staged pass without its copies, 11 the expansion without its byte stores.
 * Process-wide; 0 restores the product. */
int urhgpu_test_tail_skip(int mask);
int urhgpu_test_fetch_chunk_tables(urhgpu_ctx *ctx, void *host_dst, int64_t n_chunks);
/* Synchronous device -> host copy after urhgpu_ctx_sync (for callers that hold raw device pointers, e.g. urhgpu_host_result::d_qad). */
int urhgpu_memcpy_to_host(urhgpu_ctx *ctx, const void *d_src, void *host_dst, int64_t bytes);
/* ... and device -> device (a result's d_qad into memory the caller owns, before the stream's qad ring moves on). */
int urhgpu_memcpy_dtod(urhgpu_ctx *ctx, void *d_dst, const void *d_src, int64_t bytes);

/* Test hook: modulation order 2 (2-FSK, OOK, message segmentation) normally runs the bit-plane kernel
 * (k_demod_runs_bp) and every other order the state-byte kernel (k_demod_runs); on != 0 routes order 2 through the
 * state-byte kernel as well, so that tests can compare the two on the same input.  Process-wide. */
int urhgpu_test_force_state_bytes(int on);

/* Test hook: captures of every modulation but ASK (on one GPU and sharded) go from chunk records to bits through the five-launch "tile"
 * tail; on != 0 routes them through the generic tail (the one ASK captures use) so that tests can compare the two
 * on the same input.  Process-wide. */
int urhgpu_test_force_generic_tail(int on);

/* Test hook: the chunk plan normally depends on the capture size (1 tile = 16 rows of 128 samples per chunk below
 * about 8 M samples, 4 tiles = 64 rows -- every lane of the run phase populated, four wavefronts exchanging bit planes
 * through LDS -- above about 25 M).  tiles = 1..4 forces that many tiles per chunk for every size, so that small
 * captures the oracle finishes in seconds exercise the plan the 1 GiB benchmark runs; 0 restores the default.  Process-wide. */
int urhgpu_test_force_tiles_per_chunk(int tiles);

/* Test hook: hot launches of this process that took the signed-integer instantiation WITH the wide loop (capture streams by their probe,
 * one-shot and sharded passes by the tuning key "wide_int"): what a test of that instantiation checks it has exercised. */
int64_t urhgpu_test_wide_int_launches(void);

/* Test hook: elementwise bit-faithful atan2f (the device port of glibc 2.35 atan2f), device pointers. */
int urhgpu_test_atan2f_dev(urhgpu_ctx *ctx, const float *d_y, const float *d_x, int64_t n, float *d_out);

#ifdef __cplusplus
}
#endif
#endif /* URHGPU_H */
