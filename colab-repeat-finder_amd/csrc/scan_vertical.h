// scan_vertical.h -- host interface of the fused bit-sliced ("vertical") scan kernel and its row gather.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "prf_device.h"

#include "prf_plan.h"

// Bit-sliced planes: see scan_vertical.hip for the layout.
struct prf_vplanes {
    u32 *VH = nullptr, *VL = nullptr;     // (the not-ACGT plane exists in the linear layout only)
    unsigned char *tile_class = nullptr;  // per tile: 0 clean, 1 has not-ACGT positions in reach, 2 nothing but not-ACGT,
                                          // 3 a symbol outside ACGTN in reach (generic kernels)
    u32 *launch_list = nullptr;           // device: the tiles to scan in position order, PRF_LAUNCH_MIXED set on class-1 tiles
    u32 n_launch = 0;
    u32 flat_base = ~0u;                  // first tile if the list is one contiguous range of clean tiles
    u64 ntiles_alloc = 0;
    std::vector<unsigned char> h_class;   // host copies (planners, selections)
    std::vector<u32> h_list;
};

// One row as the scan kernel leaves it in its tile's slab: 8 bytes.
//   [15:0]  span = end - start, clipped to 65535     [31:16] start - first position of the tile
//   [47:32] motif size                               [50:48] 0, or 1 + index of the row's true end among the tile's
//                                                            long ends (span >= 65535: at most two rows of a tile, Fine and Wilf)
// Sorted by the low word = by (start, end).  The gather kernel turns them into 24-byte prf_hit_dev rows.
#define PRF_LONG_PER_TILE 4u

// everything the fused kernel needs (passed by value)
struct prf_vscan_args {
    const u32 *VH, *VL;            // bit-sliced planes
    const u64 *H, *L, *X;          // linear planes (readable padding in front and behind)
    const u64 *const *E;           // device array of the five planes of the symbols outside ACGTN, or nullptr
    const u32 *launch_list;        // tiles in position order (PRF_LAUNCH_MIXED flags); one launch slot per entry
    u32 n_launch;
    u32 flat_base;                 // != ~0u: entry i is the clean tile flat_base + i (no dependent load)
    u64 *slabs;                    // [launch slot][slab_cap]: the tile's rows (8 bytes each), sorted by (start, end)
    u64 *long_ends;                // [launch slot][PRF_LONG_PER_TILE]: true ends (global positions) of the rows whose span is clipped
    u32 *slab_count;               // [launch slot]: rows the tile produced (> slab_cap: the slab overflowed)
    u32 *block_sum;                // [launch slot >> gather_shift]: rows stored by those slots (zero when the kernel starts);
    u32 super_off;                 // from block_sum[super_off] on: the same per PRF_GATHER_SUPER gather workgroups, then their tickets
    u32 gather_shift;              // log2 of the launch slots per gather workgroup
    u32 slab_cap;
    u32 min_repeats, min_span;
    const uint4 *tile_info;        // [tile]: {contig, 0, contig base lo, hi}
    u64 *counters;                 // this scan's counter block (zero when the kernel starts)
    u64 *dbg;                      // diagnostic (PRF_STAMPS) builds only; nullptr otherwise
    u32 skip;                      // diagnostic (PRF_SKIP): phases left out to time the others; 0 in every real scan
    prf_vplan plan;
};

// the row gather: slabs in launch order -> one compact array of 24-byte rows, counters -> host
struct prf_vgather_args {
    const u64 *slabs;
    const u64 *long_ends;
    const u32 *slab_count;
    const u32 *launch_list;        // as in prf_vscan_args: the tile of a launch slot
    u32 flat_base;
    const uint4 *tile_info;
    u32 *block_sum;                // read, then cleared by the last workgroup
    u32 super_off;
    u32 gather_shift;
    u32 slab_cap;
    u32 n_launch;
    prf_hit_dev *rows;             // the compact row array, sorted by (contig, start, end) because the slabs are
    u64 rows_cap;
    u32 count_row;                 // also write {rows, 0, 0} as record rows[rows_cap]
    u64 *counters;
    u64 *host_counters;            // mapped host memory: the last workgroup copies the counter block there ...
    u64 seq;                       // ... followed by this serial number at host_counters[PRF_CNT_N]
    u64 *next_counters;            // ... and clears the block the next scan will use
};


// ASCII (global coordinate space, G bytes) -> bit-sliced planes + tile classes + launch list. Synchronises the stream.
int prf_vertical_pack(hipStream_t s, const uint8_t *asc, u64 G, prf_vplanes *vp);

// first tile if the list is one contiguous range of clean tiles, else ~0u
u32 prf_flat_base(const u32 *list, size_t n);

hipError_t prf_vertical_launch(hipStream_t s, const prf_vscan_args &args);
hipError_t prf_vertical_gather(hipStream_t s, const prf_vgather_args &args);
