// scan_vertical.h -- host interface of the fused bit-sliced ("vertical") scan kernel.
#pragma once
#include <hip/hip_runtime.h>
#include "prf_device.h"

#define PRF_VMAX_K 480       // largest motif size the fused kernel takes (9-bit k field, LDS image width)
#define PRF_VMAX_TASKS 80
#define PRF_VMAX_WAVES 16

// Bit-sliced planes: see scan_vertical.hip for the layout.
struct prf_vplanes {
    u32 *VH = nullptr, *VL = nullptr, *VX = nullptr;
    unsigned char *tile_class = nullptr;  // per tile: 0 clean, 1 has not-ACGT positions in reach, 2 nothing but not-ACGT
    u32 *tile_list = nullptr;             // device: clean tiles, then mixed tiles (the sentinel tile excluded)
    u32 n_clean = 0, n_mixed = 0;
    u32 clean_base = ~0u;                 // first clean tile if they form one contiguous range
    u64 ntiles_alloc = 0;
};

// One unit of scan work for one wave on one tile.
//  kind 0     : "group" task -- the 8 motif sizes k0 .. k0+7 (k0 % 4 == 0) selected by `valid`, all with M(k) >= 15
//  kind 1..14 : "exact" task -- the single motif size k0, whose minimum run length M(k0) equals `kind`
struct prf_vtask {
    unsigned short k0;
    unsigned char kind;
    unsigned char valid;
    unsigned char stride;   // group tasks: examine every `stride`-th aligned group of 8 rows (1, 2 or 4)
    unsigned char pad[3];
};

// Work plan of one scan (host-built from kmin,kmax,min_repeats,min_span): tasks grouped per wave.
struct prf_vplan {
    u32 n_waves;                                // workgroup = 64 * n_waves threads
    u32 n_tasks;
    u32 wave_begin[PRF_VMAX_WAVES + 1];         // wave w runs tasks[wave_begin[w] .. wave_begin[w+1])
    prf_vtask tasks[PRF_VMAX_TASKS];
    u32 nc;                                     // virtual lanes of the LDS image (64 + extra)
    u32 lds_bytes;
    u32 cof_words;                              // entries of the cofactor table staged in LDS (covers 0 .. kmax)
};

// everything the fused kernel needs (passed by value)
struct prf_vscan_args {
    const u32 *VH, *VL, *VX;       // bit-sliced planes
    const u64 *H, *L, *X;          // linear planes (readable padding in front and behind)
    const u32 *tile_list;          // clean tiles first, then mixed
    u32 n_clean, n_mixed;
    u32 clean_base;                // first clean tile if the clean tiles are one contiguous range, else ~0u
    prf_hit_dev *hit_slabs;        // [tile*4 + part][hit_cap]  (clean tiles use part 0 only)
    u32 hit_cap;
    prf_hit_dev *rows;             // the compact row array: every workgroup reserves its range with one atomic
    u64 rows_cap;
    u32 count_row;                 // the last workgroup also writes {rows, 0, 0} as record rows[rows_cap]
    u32 min_repeats, min_span;
    const u64 *contig_base;
    u32 n_contigs;
    u64 *counters;                 // this scan's counter block (zero when the kernel starts)
    u64 *host_counters;            // mapped host memory: the last workgroup copies the counter block there ...
    u64 seq;                       // ... followed by this serial number at host_counters[PRF_CNT_N]
    u64 *next_counters;            // ... and clears the block the next scan will use
    u64 *dbg;                      // diagnostic (PRF_STAMPS) builds only; nullptr otherwise
    prf_vplan plan;
};

// false if the parameters are outside what the fused kernel takes (-> generic kernel)
bool prf_vertical_plan(u32 kmin, u32 kmax, u32 min_repeats, u32 min_span, prf_vplan *plan);

// ASCII (global coordinate space, G bytes) -> bit-sliced planes + tile classes + tile lists. Synchronises the stream.
int prf_vertical_pack(hipStream_t s, const uint8_t *asc, u64 G, prf_vplanes *vp);

// n_cus: compute units of the device (picks the register budget: one round of workgroups at 3 per CU, or 4 per CU)
hipError_t prf_vertical_launch(hipStream_t s, const prf_vscan_args &args, int n_cus);


