// scan_vertical.h -- host interface of the bit-sliced ("vertical") phase-1 kernel family.
#pragma once
#include <hip/hip_runtime.h>
#include "prf_device.h"

// Bit-sliced planes: see scan_vertical.hip for the layout.
struct prf_vplanes {
    u32 *VH = nullptr, *VL = nullptr, *VX = nullptr;
    unsigned char *tile_class = nullptr;  // per tile: 0 clean, 1 has not-ACGT positions, 2 nothing but not-ACGT
    u64 ntiles_alloc = 0;
};

struct prf_vspec {
    u32 kmin, kmax, min_repeats, min_span;
    u32 waves;     // waves (k-chunks) per tile workgroup
    u32 launches;  // kernel launches per scan
    int id;
};

// returns the compiled specialisation for exactly these parameters, or nullptr (-> generic kernel)
const prf_vspec *prf_vertical_find(u32 kmin, u32 kmax, u32 min_repeats, u32 min_span);

int prf_vertical_pack(hipStream_t s, const uint8_t *asc, u64 G, prf_vplanes *vp);

hipError_t prf_vertical_launch(hipStream_t s, const prf_vspec *vs, const prf_vplanes &vp, u64 ntiles, u64 *slabs,
                               u32 *slab_counts, u32 slab_cap, u64 *cand, u64 cand_cap, u64 *counters);
