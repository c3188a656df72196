// scan_vertical.h -- host interface of the bit-sliced ("vertical") phase-1 kernel family.
#pragma once
#include <hip/hip_runtime.h>
#include "prf_device.h"

// Bit-sliced planes: see scan_vertical_impl.h for the layout.
struct prf_vplanes {
    u32 *VH = nullptr, *VL = nullptr, *VX = nullptr;
    unsigned char *tile_class = nullptr;  // per tile: 0 clean, 1 has not-ACGT positions in reach, 2 nothing but not-ACGT
    u32 *tile_list = nullptr;             // device: clean tiles, then mixed tiles (the sentinel tile excluded)
    u32 n_clean = 0, n_mixed = 0;
    u64 ntiles_alloc = 0;
};

struct prf_vspec {
    u32 kmin, kmax, min_repeats, min_span;
    u32 waves;     // waves (k-chunks) per tile workgroup
    u32 launches;  // kernel launches per scan (upper bound)
    int id;
};

typedef hipError_t (*prf_vlaunch_fn)(hipStream_t, const prf_vplanes &, u64 *slabs, u32 *slab_counts, u32 slab_cap,
                                     u64 *counters);
struct prf_ventry {
    prf_vspec spec;
    prf_vlaunch_fn fn;
};

// returns the compiled specialisation for exactly these parameters, or nullptr (-> generic kernel)
const prf_vspec *prf_vertical_find(u32 kmin, u32 kmax, u32 min_repeats, u32 min_span);

// ASCII (global coordinate space, G bytes) -> bit-sliced planes + tile classes + tile lists. Synchronises the stream.
int prf_vertical_pack(hipStream_t s, const uint8_t *asc, u64 G, prf_vplanes *vp);

hipError_t prf_vertical_launch(hipStream_t s, const prf_vspec *vs, const prf_vplanes &vp, u64 *slabs, u32 *slab_counts,
                               u32 slab_cap, u64 *counters);
