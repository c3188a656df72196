// prf_host.h -- launch wrappers shared between the kernel translation units and api.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "prf_device.h"

// device counter block (u64 each)
enum {
    PRF_CNT_CAND = 0,      // generic path: phase-1 candidates
    PRF_CNT_HITS = 1,      // generic path: rows
    PRF_CNT_BADPOS = 2,    // packer: first byte that is not a letter
    PRF_CNT_EXOTIC = 7,    // packer: 1 = the input holds letters other than A, C, G, T, N
    PRF_CNT_UNSORTED = 3,  // fused path: a tile wrote rows past its LDS sort list -> the row array is not fully sorted
    PRF_CNT_HIT_OVF = 4,   // fused path: largest per-tile row demand above the slab capacity
    PRF_CNT_ROWS = 5,      // fused path: rows in the compact array (written by the gather kernel)
    PRF_CNT_TICKET = 6,    // fused path: gather workgroups finished (the last one hands the counters to the host)
    PRF_CNT_SHARD0 = 8,    // fused path: candidate-record counts (statistics), sharded over 16 cache lines by tile
    PRF_CNT_NSHARD = 16,
    PRF_CNT_SHARD_STRIDE = 8,
    PRF_SH_CAND = 1,
    PRF_SH_EARLY = 4,        // fused path: candidates verified on the spot because a tile's list was full (diagnostic)
    PRF_SH_TILE_TICKET = 2,  // fused path, shards 0..7: launch slots handed out to the persistent workgroups of XCD 0..7
    PRF_CNT_LONG_OVF = 8 + 3,  // fused path (shard 0, word 3): a tile held more than PRF_LONG_PER_TILE rows with a clipped span (cannot happen)
    PRF_CNT_N = 8 + 16 * 8
};

hipError_t prf_launch_pack_linear(hipStream_t s, const uint8_t *asc, u64 nwords, u64 *H, u64 *L, u64 *X,
                                  u64 *bad_pos, u64 *exotic);
hipError_t prf_launch_pack_exotic(hipStream_t s, const uint8_t *asc, u64 nwords, u64 *const *E);
hipError_t prf_launch_fill_u64(hipStream_t s, u64 *p, u64 n, u64 v);
hipError_t prf_launch_synth(hipStream_t s, uint8_t *asc, u64 n, u64 seed);
// stand-in recipe 2 (synth.py::standin2): background + N blocks + one planted repeat per 588-position slot
hipError_t prf_launch_standin2(hipStream_t s, uint8_t *asc, u64 n, u64 seed);

hipError_t prf_launch_scan_generic(hipStream_t s, const prf_planes &pl, u64 w_begin, u64 w_end, u32 kmin, u32 kmax,
                                   u32 min_repeats, u32 min_span, u64 *cand, u64 cand_cap, u64 *counters);

hipError_t prf_launch_verify(hipStream_t s, const prf_planes &pl, const u64 *cand, u64 cand_cap, u32 min_repeats,
                             u32 min_span, const u64 *contig_base, u32 n_contigs, prf_hit_dev *hits, u64 hit_cap,
                             u64 *counters);

hipError_t prf_launch_hbm_read(hipStream_t s, const void *p, u64 bytes, u32 *sink);
hipError_t prf_launch_pack_rows(hipStream_t s, const prf_hit_dev *rows, u64 n, const u64 *contig_base, u64 *dst, u64 cap, u64 side_cap,
                                u64 *side_cnt, u64 *host_word);

// the same with the row count read on the device (behind a scan that is still in flight); side_cnt must be zero and is zero again afterwards
hipError_t prf_launch_pack_rows_dev(hipStream_t s, const prf_hit_dev *rows, const u64 *n_ptr, u64 cap, const u64 *contig_base, u64 *dst,
                                    u64 side_cap, u64 *side_cnt);

// literal lane (scan_literal.hip): upper-case in place + first non-letter; one thread per (position, motif size) event
hipError_t prf_launch_lit_upper(hipStream_t s, uint8_t *seq, u64 n, u64 *bad_pos);
hipError_t prf_launch_lit_events(hipStream_t s, const uint8_t *seq, u64 L, u32 kmin, u32 kmax, u32 min_repeats, u32 min_span,
                                 u64 stop, u32 contig, prf_hit_dev *rows, u64 cap, u64 *counters);
// the same lane on a resident genome: first / one-past-last position of a contig that is not N (atomicMin / atomicMax into
// first_last[0..1]); the upper-cased bytes of n positions from global position g0 rebuilt from the linear planes
hipError_t prf_launch_lit_trim(hipStream_t s, const u64 *X, const u64 *const *E, u64 word0, u64 len, u64 *first_last);
hipError_t prf_launch_lit_unpack(hipStream_t s, const u64 *H, const u64 *L, const u64 *X, const u64 *const *E, u64 g0, u64 n,
                                 uint8_t *out);
// the rows of the event kernel sorted by (start, end) and reduced to the shortest motif per (start, end), on the device
hipError_t prf_lit_sort_unique(hipStream_t s, const prf_hit_dev *rows, u64 n, prf_hit_dev *out, u64 *n_out, void **scratch,
                               size_t *scratch_bytes);
