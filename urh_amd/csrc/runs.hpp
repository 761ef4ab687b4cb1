// Device-side data structures shared by the demod/run-segmentation kernels (demod_runs.hip),
// the pulse-table kernels and the host glue.
//
// Vocabulary (follows the reference, signal_functions.pyx:392-495):
//   state      per-sample symbol class: PAUSE (-1) when the demodulated sample equals the NOISE
//              sentinel, else the index of the first center threshold it does not exceed
//   run        maximal stretch of equal per-sample states
//   stable run run longer than `tolerance` samples -- the only runs that can switch the
//              reference's hysteresis state machine
//   accepted   stable run whose state differs from the previous stable run's state: each one is a
//              row boundary of the pulse table ("ppseq")
//
// Work split: the capture is cut into sample-contiguous CHUNKS, one WAVEFRONT (= one 64-thread
// workgroup) each, so that no wavefront ever waits for another one; a chunk is walked in TILES of
// 2048 samples.  Everything a chunk cannot decide alone (does its first stable
// run differ from the previous chunk's last one?  does its last, still-short run continue into
// the next chunk?) goes into one ChunkInfo record, resolved by k_resolve_chunks over all chunks
// (and, for sharded captures, over the chunks of all GPUs).
#pragma once
#include <stdint.h>

namespace urh {

constexpr int kBlock = 64;                     // threads per workgroup: ONE wavefront
constexpr int kRows = 16;                      // 16-byte IQ loads per thread per tile
constexpr int kRowSamples = kBlock * 2;        // 128 samples (1 KiB of complex64) per wavefront-wide load
constexpr int kTile = kRowSamples * kRows;     // 2048 samples per tile
constexpr int kSpan = kTile / kBlock;          // 32 samples per thread in the run phase

// State byte stored in LDS: reference state + 1 (PAUSE=-1 -> 0); 0xFF = "no sample".
constexpr uint32_t kStPause = 0;
constexpr uint32_t kStNone = 0xFF;
constexpr int kMaxOrder = 128;                 // bits_per_symbol <= 7
constexpr int kMaxSegments = 16;               // segments of a streamed pass (segmented tail, pulse_table.hip)
constexpr int kProgressStride = 32;            // uint32 words between two segments' progress counters: a 128-byte line each

// One accepted/stable run start: sample position | state byte << 56.
__host__ __device__ inline uint64_t rec_make(int64_t pos, uint32_t st) { return (uint64_t)pos | ((uint64_t)st << 56); }
__host__ __device__ inline int64_t rec_pos(uint64_t r) { return (int64_t)(r & 0x00FFFFFFFFFFFFFFull); }
__host__ __device__ inline uint32_t rec_state(uint64_t r) { return (uint32_t)(r >> 56); }

struct ChunkInfo {
    // ---- written by k_demod_runs (one per chunk) ----
    int64_t pend_pos;      // start of the chunk's last run if it is still <= tolerance long at chunk end, else -1
    int64_t lead;          // samples at the chunk start that continue the previous chunk's last run
    int64_t start;         // absolute position of the chunk's first sample
    int64_t len;           // valid samples in the chunk
    int64_t last_pos;      // position of the last record written (valid when cnt > 0)
    int32_t cnt;           // records in this chunk's slab region; record 0 is tentative (see resolve)
    uint16_t first_state;  // state byte of record 0
    uint16_t last_state;   // state byte of the chunk's last stable run
    uint16_t pend_state;
    uint16_t init_state;   // chunk 0 only: initial cur_state of the reference state machine
    // ---- written by the resolve kernels (pulse_table.hip) ----
    int32_t first_acc;     // record 0 accepted?
    int32_t pend_acc;      // pending run resolved stable AND accepted?
    int32_t pend_stable;   // pending run grows past `tolerance` in the following chunks?
    int32_t pad;
};
static_assert(sizeof(ChunkInfo) == 72, "URHGPU_SHARD_SUMMARY_BYTES (include/urhgpu.h) is sizeof(ChunkInfo)");

}  // namespace urh
