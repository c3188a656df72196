// verify_impl.h -- candidate -> row logic shared by the flat phase-2 kernel (verify.hip) and the fused
// tail of the bit-sliced kernel (scan_vertical.hip).
//
// What it replaces: PerfectRepeatTracker.output_interval_if_it_passes_filters()
// (reference utils/perfect_repeat_tracker.py:71-101) and consists_of_perfect_repeats() (:108-142),
// in the closed form of SURVEY 3.4 (valid for min_repeats >= 2):
//     a maximal run [a,b) of matches at period k is a row (a, b+k, k)  iff
//     b-a >= M(k) = max((min_repeats-1)*k, min_span-k)  and  seq[a:a+k] is a primitive word.
// The "N in motif" drop (:83) is implied: with M(k) >= k every position of [a, b+k) is a non-N base.
// The keep-shorter de-duplication (:94-96) is dead in this regime (Fine-Wilf).
//
// `View` supplies 64 consecutive bits of a linear plane at an arbitrary position:
//     u64 bits(int plane /*0=H 1=L 2=X*/, u64 q) const
#pragma once
#include "prf_device.h"

struct prf_global_view {
    const u64 *P[3];
    const u64 *E[5];  // planes of the symbols outside ACGTN (prf_planes::E); E[0] == nullptr: none
    __device__ __forceinline__ u64 bits(int plane, u64 q) const { return prf_bits_at(P[plane], q); }
    // mismatch bits (1 = differs, or either side is N) of positions q .. q+63 against q+k ..
    __device__ __forceinline__ u64 mismatch64(u64 q, u32 k) const {
        const u64 h = bits(0, q) ^ bits(0, q + k);
        const u64 l = bits(1, q) ^ bits(1, q + k);
        const u64 x = bits(2, q) | bits(2, q + k);
        u64 m = h | l | x;
        if (E[0] && x) m &= ~prf_exotic_equal64(E, q, k);
        return m;
    }
};

// a window of the linear planes staged in LDS ([plane][nwords], first word = global word w0); positions
// outside the window fall through to global memory.  For clean tiles only H and L are staged: the
// not-ACGT plane X is known to be zero on [xz_lo, xz_hi) (the tile and its successor) and read from global
// memory elsewhere (the 64 positions in front of the tile).  Tiles with N in reach stage X as well.
typedef __attribute__((address_space(3))) const u64 prf_lds_cu64;  // explicitly LDS: ds_read, not flat_load
struct prf_window_view {
    prf_lds_cu64 *lds;
    u64 w0;
    u32 nwords;
    u64 xz_lo, xz_hi;
    u32 x_in_lds;  // 1: the window also holds the X plane (third), used by tiles with N in reach
    const u64 *P[3];
    const u64 *const *E;  // device array of the five planes of the symbols outside ACGTN, or nullptr (prf_planes::E)
    __device__ __forceinline__ u64 bits(int plane, u64 q) const {
        if (plane == 2 && !x_in_lds) {
            if (q >= xz_lo && q + 64 <= xz_hi) return 0;
            return prf_bits_at(P[2], q);
        }
        const u64 rel = (q >> 6) - w0;  // wraps to a huge value below the window
        const unsigned s = (unsigned)(q & 63);
        if (rel + 1 < (u64)nwords) {
            prf_lds_cu64 *p = lds + (u32)plane * nwords + (u32)rel;
            return prf_fsr(p[0], p[1], s);
        }
        return prf_bits_at(P[plane], q);
    }
    __device__ __forceinline__ u64 mismatch64(u64 q, u32 k) const {
        // common case: both 64-position looks lie inside the LDS window and inside the N-free range
        const u64 ra = (q >> 6) - w0, rb = ((q + k) >> 6) - w0;
        if (ra + 1 < (u64)nwords && rb + 1 < (u64)nwords && q >= xz_lo && q + k + 64 <= xz_hi) {
            const unsigned sa = (unsigned)(q & 63), sb = (unsigned)((q + k) & 63);
            prf_lds_cu64 *ha = lds + (u32)ra, *hb = lds + (u32)rb;
            prf_lds_cu64 *la = ha + nwords, *lb = hb + nwords;
            return (prf_fsr(ha[0], ha[1], sa) ^ prf_fsr(hb[0], hb[1], sb)) | (prf_fsr(la[0], la[1], sa) ^ prf_fsr(lb[0], lb[1], sb));
        }
        const u64 h = bits(0, q) ^ bits(0, q + k);
        const u64 l = bits(1, q) ^ bits(1, q + k);
        const u64 x = bits(2, q) | bits(2, q + k);
        u64 m = h | l | x;
        if (E && x) m &= ~prf_exotic_equal64(E, q, k);  // (a window or a tile with such a symbol in reach is never scanned here)
        return m;
    }
};

template <class View>
__device__ __forceinline__ u64 prf_vmismatch64(const View &v, u64 q, u32 k) {
    return v.mismatch64(q, k);
}

// does seq[a : a+k] have period d (d < k)?
template <class View>
__device__ __forceinline__ bool prf_vhas_period(const View &v, u64 a, u32 k, u32 d) {
    const u32 need = k - d;  // positions a .. a+need-1 must equal the ones d later
    for (u32 off = 0; off < need; off += 64) {
        u64 mm = prf_vmismatch64(v, a + off, d);
        const u32 left = need - off;
        if (left < 64) mm &= (1ull << left) - 1ull;
        if (mm) return false;
    }
    return true;
}

// is seq[a : a+k] a whole number (>= 2) of copies of a shorter word?  A word of length k has a proper
// divisor period iff it has period k/p for some prime p | k.  Primes up to 23 are tried with constant
// divisors (cheap); what is left of k after that is 1 or, for k < 29*29, a single larger prime.
template <class View>
__device__ __forceinline__ bool prf_vmotif_is_repeat(const View &v, u64 a, u32 k) {
    if (k < 2) return false;
    u32 rest = k;
#define PRF_TRY_PRIME(P)                                  \
    if (rest % P == 0) {                                  \
        if (prf_vhas_period(v, a, k, k / P)) return true; \
        do rest /= P; while (rest % P == 0);              \
    }
    PRF_TRY_PRIME(2u)
    PRF_TRY_PRIME(3u)
    if (rest == 1) return false;
    PRF_TRY_PRIME(5u)
    PRF_TRY_PRIME(7u)
    if (rest == 1) return false;
    PRF_TRY_PRIME(11u)
    PRF_TRY_PRIME(13u)
    PRF_TRY_PRIME(17u)
    PRF_TRY_PRIME(19u)
    PRF_TRY_PRIME(23u)
#undef PRF_TRY_PRIME
    for (u32 p = 29; rest > 1; p += 2) {  // only reached for k with a prime factor > 23
        if (p * p > rest) p = rest;       // what is left is prime
        if (rest % p) continue;
        if (prf_vhas_period(v, a, k, k / p)) return true;
        do rest /= p; while (rest % p == 0);
    }
    return false;
}

// One candidate (p, k, kind) -> true and the run [a, b) if it is a row.
//  kind GROUP/GROUP2/GROUP4: [p, p+8) all match; the run is reported only by the first of its examined
//              aligned all-match groups (every 1st / 2nd / 4th aligned group of 8 is examined).
//  kind START: p is expected to be the first matching position of the run (a conservatively reported p
//              that sits inside a run is dropped: the real start reports it).
template <class View>
__device__ __forceinline__ bool prf_candidate_to_run(const View &v, u64 p, u32 k, u32 kind, u32 min_repeats, u32 min_span,
                                                     u64 &a_out, u64 &b_out) {
    // One 64-position look from just before the candidate usually shows the whole run.
    // `back`: how far before p the previous examined position/group lies (1 for START; 8, 16, 32 for groups
    // examined at every 1st, 2nd, 4th aligned group of 8).
    const u32 back = kind == (u32)PRF_KIND_START ? 1u : (8u << (kind - 1u));
    const u32 look = p >= back ? back : (u32)p;  // the contig / array starts less than `back` before p
    u64 a = p, b;
    u64 mm = ~0ull;
    if (look) {
        mm = prf_vmismatch64(v, p - look, k);
        const u64 lead = mm & ((1ull << look) - 1ull);
        if (kind != (u32)PRF_KIND_START) {
            if (lead == 0) {
                if (look == back) return false;  // the previous examined group lies in the same run: it reports
                a = p - look;                    // run starts at position 0
            } else {
                a = p - (u64)__builtin_clzll(lead << (64 - look));  // matches directly before p
            }
        } else if (lead == 0) {
            return false;  // p sits inside a run: its real start reports it
        }
        mm >>= look;  // bit i = mismatch at p+i, for i < 64-look
    }
    {
        const u32 known = 64 - look;  // valid low bits of mm (0 when look == 0 ... then mm is all ones: unknown)
        const u64 seen = look ? (mm & ((known < 64) ? ((1ull << known) - 1ull) : ~0ull)) : 0ull;
        if (look && seen) {
            b = p + (u64)__builtin_ctzll(seen);
        } else {
            b = look ? p + known : p;
            for (;;) {  // long run (or nothing seen yet): keep walking; the guard gap guarantees an end
                const u64 m2 = prf_vmismatch64(v, b, k);
                if (m2) {
                    b += (u64)__builtin_ctzll(m2);
                    break;
                }
                b += 64;
            }
        }
    }
    if ((long long)(b - a) < prf_min_matches(k, min_repeats, min_span)) return false;
    if (prf_vmotif_is_repeat(v, a, k)) return false;
    a_out = a;
    b_out = b;
    return true;
}

// contig of a global position: last base <= a
__device__ __forceinline__ u32 prf_contig_of(const u64 *__restrict__ contig_base, u32 n_contigs, u64 a) {
    u32 lo = 0, hi = n_contigs;
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (contig_base[mid] <= a) lo = mid; else hi = mid;
    }
    return lo;
}
