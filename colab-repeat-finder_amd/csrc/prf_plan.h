// prf_plan.h -- the work plan of the fused kernel and the LDS layout constants it is sized by: plain C++, no HIP headers, so
// that the planner (plan.cpp) also builds with g++ under AddressSanitizer / UBSan (make asan; tests/test_asan_host.py).
#pragma once
#include <stdint.h>

typedef unsigned long long u64;
typedef unsigned int u32;

#define PRF_VMAX_K 480       // largest motif size the fused kernel takes (9-bit k field, LDS image width)
#define PRF_VMAX_TASKS 80
#define PRF_VMAX_WAVES 4
#define PRF_GATHER_SLOTS_MAX 64u      // launch slots per workgroup of the row gather: 8 or 64 (gather_shift 3 / 6)
#define PRF_GATHER_SUPER 64u          // gather workgroups per second-level sum
#define PRF_LAUNCH_MIXED 0x80000000u  // launch-list entry: the tile has not-ACGT positions in reach

// One unit of scan work for one wave on one tile.
//  kind 0     : "group" task -- the 8 motif sizes k0 .. k0+7 (k0 % 4 == 0) selected by `valid`, all with M(k) >= 15
//  kind 1..14 : "exact" task -- the single motif size k0 (<= 14), whose minimum run length M(k0) equals `kind`
struct alignas(4) prf_vtask {   // (two dwords: the kernel reads a task with one scalar load)
    unsigned short k0;
    unsigned char kind;
    unsigned char valid;
    unsigned char stride;   // group tasks: examine every `stride`-th aligned group of 8 rows (1, 2 or 4)
    unsigned char pad;
    unsigned short item0;   // group tasks: index of the task's first motif size among all motif sizes of group tasks;
                            // exact tasks: index among the exact tasks
};

// Work plan of one scan (host-built from kmin,kmax,min_repeats,min_span): tasks grouped per wave.
struct prf_vplan {
    u32 n_waves;                                // waves that have tasks (the workgroup always has PRF_VMAX_WAVES)
    u32 n_tasks;
    u32 wave_begin[PRF_VMAX_WAVES + 1];         // wave w runs tasks[wave_begin[w] .. wave_begin[w+1])
    prf_vtask tasks[PRF_VMAX_TASKS];
    u32 nc;                                     // virtual lanes of the LDS image (64 + extra)
    u32 lds_bytes;
    u32 cof_words;                              // entries of the cofactor table staged in LDS (covers 0 .. kmax)
    u32 n_group_k;                              // motif sizes scanned by group tasks (boundary items of a tile)
    u32 n_exact;                                // exact tasks: motif sizes k_exact0 .. k_exact0 + n_exact - 1 (item0 = k - k_exact0)
    u32 k_exact0;
    u32 ticket_wave;                            // the wave whose first lane draws the workgroup's next launch slot (it waits for the atomic)
    u32 slack_waves;                            // bit w: wave w's scan work is < 90 % of the busiest wave's
    u32 per_cu;                                 // workgroups per CU the plan's LDS and the kernel's registers allow
    u32 prio;                                   // issue priorities (s_setprio), 2 bits each: stage | scan, busy wave << 2 | scan, slack
                                                // wave << 4 | verify, flag waves << 6 | verify, record waves << 8 | rows << 10
};


// LDS layout of the fused kernel (scan_vertical.hip), shared with the planner, which sizes the launch by it
namespace prf_layout {
constexpr int T = 32;                                         // rows (= positions) per stream
constexpr int RG = T / 4;                                     // row groups of 4 rows = one 16-byte slot per lane
constexpr int LIN_PRE = 2;                                    // linear window: words before the tile (even: 16-byte DMA pieces)
constexpr int LIN_POST = 24;                                  // ... and after it
constexpr int TILE_WORDS = 1024;                              // 64-bit linear words per tile (PRF_TILE_WORDS)
constexpr int LW = TILE_WORDS + LIN_PRE + LIN_POST;           // words per plane in the LDS window
constexpr int REC_CAP = 192;                                  // group-task candidate records of a tile (one LDS list)
constexpr int FLAG_CAP = 960;                                 // (stream, exact task) flags of a tile (one LDS list, 2 bytes each)
constexpr int SLOW_CAP = 16;                                  // candidates of a tile that leave the LDS window: finished after the others
constexpr int SMALL_M = 15;                                   // M(k) below this -> exact task
constexpr int ROW_CAP_LDS = 448;                              // rows of a tile in the LDS list (more: the slab holds the rest, sorted in R1)
constexpr int SMEM_HDR = 240;                                 // tile context, counters, next-slot words, long ends
}  // namespace prf_layout

// minimum number of consecutive matching positions of a reportable run: the reference filters
// run+k-1 >= min_span and >= min_repeats*k (perfect_repeat_tracker.py:86,:91)
static inline long long prf_plan_min_matches(u32 k, u32 min_repeats, u32 min_span) {
    const long long a = (long long)(min_repeats - 1) * (long long)k, b = (long long)min_span - (long long)k;
    return a > b ? a : b;
}

// false if the parameters are outside what the fused kernel takes (-> generic kernel)
bool prf_vertical_plan(u32 kmin, u32 kmax, u32 min_repeats, u32 min_span, prf_vplan *plan);
// JSON description of the plan (include/prf.h: prf_plan_describe); returns the length or -1 if buf is too small
int prf_plan_json(u32 kmin, u32 kmax, u32 min_repeats, u32 min_span, char *buf, u64 buf_len);
