// scan_vertical.hip -- placeholder until the bit-sliced kernel lands: no specialisations yet.
#include "scan_vertical.h"
const prf_vspec *prf_vertical_find(u32, u32, u32, u32) { return nullptr; }
int prf_vertical_pack(hipStream_t, const uint8_t *, u64, prf_vplanes *) { return 0; }
hipError_t prf_vertical_launch(hipStream_t, const prf_vspec *, const prf_vplanes &, u64, u64 *, u32 *, u32, u64 *, u64, u64 *) {
    return hipErrorNotSupported;
}
