// scan_generic.hip -- phase 1, generic kernel: any k range, any thresholds.
//
// What it replaces: the L x n_k calls of PerfectRepeatTracker.advance()
// (reference utils/perfect_repeat_tracker.py:43-61, driven by perfect_repeat_finder.py:66-70).
// Instead of one character compare per call, one lane evaluates 64 positions of one k per step
// on the linear bit planes:
//     mismatch(j,k) = (H[j]^H[j+k]) | (L[j]^L[j+k]) | X[j] | X[j+k]
// and finds, exactly, every position where a maximal run of matches begins that is at least
// min(M(k), 64) long (shift-AND doubling).  Those run starts are the phase-1 candidates; phase 2
// (verify.hip) measures the run, applies the reference's filters and emits the row.
//
// This is the always-available path.  The bit-sliced kernel in scan_vertical.hip is the fast
// path for the parameter sets it is instantiated for.
#include "prf_device.h"
#include "prf_host.h"

struct u128 {
    u64 lo, hi;
};
__device__ __forceinline__ u128 shr128(u128 v, unsigned s) {  // s in [0,63]
    u128 r;
    r.lo = prf_fsr(v.lo, v.hi, s);
    r.hi = s ? (v.hi >> s) : v.hi;
    return r;
}

// grid-stride over linear words [w_begin, w_end); each lane owns one word and loops over k.
__global__ __launch_bounds__(256) void prf_scan_generic_kernel(prf_planes pl, u64 w_begin, u64 w_end, u32 kmin,
                                                               u32 kmax, u32 min_repeats, u32 min_span,
                                                               u64 *__restrict__ cand, u64 cand_cap,
                                                               u64 *__restrict__ counters) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 w = w_begin + (u64)blockIdx.x * blockDim.x + threadIdx.x; w < w_end; w += stride) {
        const u64 xa = pl.X[w];
        // a word of nothing but N / guard gap cannot start a run
        if (xa == ~0ull && !pl.E[0]) continue;
        const u64 ha = pl.H[w], hb = pl.H[w + 1];
        const u64 la = pl.L[w], lb = pl.L[w + 1];
        const u64 xb = pl.X[w + 1];
        // position 64w-1 (the bit before this word); for w == 0 there is none -> "mismatch"
        const u64 hp = w ? pl.H[w - 1] : 0, lp = w ? pl.L[w - 1] : 0, xp = w ? pl.X[w - 1] : ~0ull;
        for (u32 k = kmin; k <= kmax; k++) {
            const u64 q = w + (k >> 6);
            const unsigned s = k & 63;
            const u64 h0 = pl.H[q], h1 = pl.H[q + 1], h2 = pl.H[q + 2];
            const u64 l0 = pl.L[q], l1 = pl.L[q + 1], l2 = pl.L[q + 2];
            const u64 x0 = pl.X[q], x1 = pl.X[q + 1], x2 = pl.X[q + 2];
            u128 m;  // mismatch bits of positions 64w .. 64w+127
            m.lo = (ha ^ prf_fsr(h0, h1, s)) | (la ^ prf_fsr(l0, l1, s)) | xa | prf_fsr(x0, x1, s);
            m.hi = (hb ^ prf_fsr(h1, h2, s)) | (lb ^ prf_fsr(l1, l2, s)) | xb | prf_fsr(x1, x2, s);
            // mismatch bit of position 64w-1
            u64 mprev = 1;
            if (w) {
                const u64 hq = pl.H[q - 1], lq = pl.L[q - 1], xq = pl.X[q - 1];
                const u64 mp = (hp ^ prf_fsr(hq, h0, s)) | (lp ^ prf_fsr(lq, l0, s)) | xp | prf_fsr(xq, x0, s);
                mprev = mp >> 63;
            }
            if (pl.E[0]) {  // the genome holds symbols outside ACGTN: equal ones match (prf_planes::E)
                m.lo &= ~prf_exotic_equal64(pl.E, w * 64, k);
                m.hi &= ~prf_exotic_equal64(pl.E, w * 64 + 64, k);
                if (w && mprev) mprev = (prf_exotic_equal64(pl.E, w * 64 - 64, k) >> 63) ^ 1ull;
            }
            const long long M = prf_min_matches(k, min_repeats, min_span);
            const unsigned mcap = (unsigned)(M < 1 ? 1 : (M > 64 ? 64 : M));
            u128 acc;  // bit i: positions i .. i+len-1 all match
            acc.lo = ~m.lo;
            acc.hi = ~m.hi;
            unsigned len = 1;
            while (2 * len <= mcap) {
                const u128 t = shr128(acc, len);
                acc.lo &= t.lo;
                acc.hi &= t.hi;
                len *= 2;
            }
            if (len < mcap) {
                const u128 t = shr128(acc, mcap - len);
                acc.lo &= t.lo;
            }
            // run start: match here, mismatch just before
            const u64 before = (m.lo << 1) | mprev;
            u64 c = acc.lo & before;
            while (c) {
                const unsigned i = (unsigned)__builtin_ctzll(c);
                c &= c - 1;
                const u64 slot = atomicAdd(&counters[PRF_CNT_CAND], 1ull);
                if (slot < cand_cap)
                    cand[slot] = (w * 64 + i) | ((u64)k << PRF_CAND_K_SHIFT) | (PRF_KIND_START << PRF_CAND_KIND_SHIFT);
            }
        }
    }
}

hipError_t prf_launch_scan_generic(hipStream_t s, const prf_planes &pl, u64 w_begin, u64 w_end, u32 kmin, u32 kmax,
                                   u32 min_repeats, u32 min_span, u64 *cand, u64 cand_cap, u64 *counters) {
    if (w_end <= w_begin) return hipSuccess;
    const u32 bs = 256;
    u64 nb = (w_end - w_begin + bs - 1) / bs;
    if (nb > 256u * 32u) nb = 256u * 32u;
    hipLaunchKernelGGL(prf_scan_generic_kernel, dim3((u32)nb), dim3(bs), 0, s, pl, w_begin, w_end, kmin, kmax,
                       min_repeats, min_span, cand, cand_cap, counters);
    return hipGetLastError();
}
