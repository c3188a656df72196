// prf_device.h -- shared layout constants and device helpers for the libprf kernels (gfx950 only).
//
// Global coordinate space: every contig is placed at a base position that is a multiple of
// PRF_TILE and is followed by a guard gap of not-ACGT positions at least kmax_hint+64 long, so
// that (a) no comparison seq[j] vs seq[j+k] can pair bases of two contigs and (b) positions
// j >= L-k of a contig are mismatches by construction -- the end test of the reference's
// PerfectRepeatTracker.advance() (reference utils/perfect_repeat_tracker.py:50).
//
// Linear bit planes (one bit per position, bit i of word w = position 64*w+i):
//   H, L : the two bits of the base code   ((ascii >> 1) & 3 after case folding: A=0 C=1 T=2 G=3)
//   X    : 1 = not one of ACGT (N, guard gap, padding).  The reference's "seq[i] != 'N'" test
//          (perfect_repeat_tracker.py:53) becomes  mismatch |= X[j] | X[j+k].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef unsigned int u32;

#define PRF_TILE 65536u          // positions per tile (2048 bit-sliced streams x 32 words)
#define PRF_TILE_WORDS 1024u     // 64-bit linear words per tile

// candidate record: [43:0] global position, [59:44] k, [63:60] kind
#define PRF_CAND_POS_BITS 44
#define PRF_CAND_K_SHIFT 44
#define PRF_CAND_KIND_SHIFT 60
#define PRF_KIND_START 0ull   // pos is the exact first matching position of a maximal run
#define PRF_KIND_GROUP 1ull   // pos is the first position of an aligned all-match group of 8; the run
                              // may have begun up to 7 (leader) or more (not a leader) positions earlier
#define PRF_KIND_GROUP2 2ull  // the same, but only every 2nd aligned group is examined (M(k) >= 23) ...
#define PRF_KIND_GROUP4 3ull  // ... every 4th (M(k) >= 39): "leader" = first EXAMINED all-match group of the run

struct prf_hit_dev {
    u64 start, end;
    u32 k, contig;
};

struct prf_planes {
    const u64 *H, *L, *X;
    // Symbols other than A, C, G, T, N (IUPAC codes ...) are ORDINARY symbols to the reference: R == R matches, only the
    // literal N never does (utils/perfect_repeat_tracker.py:53).  They are rare, so they do not get planes of their own on the
    // fast path: X is set for them (like N), and five extra linear planes E[0..4] hold the low five bits of the upper-cased
    // letter (A = 1 ... Z = 26; 0 for A, C, G, T and N).  E[0] == nullptr: the genome holds no such symbol.
    const u64 *E[5];
};

// funnel shift right of the 128-bit value hi:lo by s in [0,63]
__device__ __forceinline__ u64 prf_fsr(u64 lo, u64 hi, unsigned s) {
    return s ? (lo >> s) | (hi << (64u - s)) : lo;
}

// 64 consecutive bits of plane P starting at absolute bit position q
__device__ __forceinline__ u64 prf_bits_at(const u64 *__restrict__ P, u64 q) {
    const u64 w = q >> 6;
    return prf_fsr(P[w], P[w + 1], (unsigned)(q & 63));
}

// bits j = q .. q+63 at which seq[j] and seq[j+k] are both symbols outside ACGTN and equal
__device__ __forceinline__ u64 prf_exotic_equal64(const u64 *const *E, u64 q, u32 k) {
    u64 nz_a = 0, nz_b = 0, diff = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const u64 a = prf_bits_at(E[i], q), b = prf_bits_at(E[i], q + k);
        nz_a |= a;
        nz_b |= b;
        diff |= a ^ b;
    }
    return nz_a & nz_b & ~diff;
}

// mismatch bits (1 = "seq[j] != seq[j+k] or seq[j] is N or seq[j+k] is N") for j = q .. q+63
__device__ __forceinline__ u64 prf_mismatch64(const prf_planes &p, u64 q, u32 k) {
    const u64 h = prf_bits_at(p.H, q) ^ prf_bits_at(p.H, q + k);
    const u64 l = prf_bits_at(p.L, q) ^ prf_bits_at(p.L, q + k);
    const u64 x = prf_bits_at(p.X, q) | prf_bits_at(p.X, q + k);
    u64 m = h | l | x;
    if (p.E[0] && x) m &= ~prf_exotic_equal64(p.E, q, k);  // both sides the same symbol outside ACGTN: a match
    return m;
}

// minimum number of consecutive matching positions of a reportable run:
// reference filters run+k-1 >= min_span and >= min_repeats*k (perfect_repeat_tracker.py:86,:91)
__host__ __device__ __forceinline__ long long prf_min_matches(u32 k, u32 min_repeats, u32 min_span) {
    long long a = (long long)(min_repeats - 1) * (long long)k;
    long long b = (long long)min_span - (long long)k;
    return a > b ? a : b;
}
