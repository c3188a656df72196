// scan_vertical.hip -- the fast path: fused bit-sliced ("vertical") scan + verification kernel for gfx950,
// its work planner, the ASCII -> bit-sliced packer and the row compaction.
//
// What the kernel replaces: the L x n_k calls of PerfectRepeatTracker.advance()
// (reference utils/perfect_repeat_tracker.py:43-61) AND the per-run filter/emit step (:71-101, :108-142),
// for any kmin..kmax <= 480 and any thresholds with min_repeats >= 2, in ONE launch per scan.
//
// Layout.  A tile is 65536 consecutive positions, cut into 2048 streams of T = 32 positions.  Stream
// s = bit*64 + lane lives in bit `bit` of lane `lane`: the 32-bit word W[t][lane] holds, in bit b,
// position  tile*65536 + (b*64 + lane)*32 + t.  One wave-wide word row therefore advances 2048
// independent streams by one position, and "position j+k" is simply row t+k of the same lane (or,
// past the end of the stream, row (t+k)%32 of lane + (t+k)/32, because the next stream of a lane is the
// same bit of the next lane).  Lanes 64.. of that virtual lane axis are the first lanes again, moved up
// one bit, with bit 31 taken from the next tile; they are materialised once per tile in LDS.
// In HBM a plane of a tile is stored [t/4][lane][t%4] so that one lane reads 4 rows with one 16-byte
// access and a wave reads 1 KiB contiguously; the LDS image has the same shape with 64+J lanes.
// The shift by k therefore costs no instruction: it is an LDS address.
//
// One workgroup (up to 4 waves) per tile, three steps:
//  1. stage: the tile's bit-sliced planes and a window of the LINEAR H/L planes (tile - 64 .. tile + 65536 +
//     1536 positions) go to LDS.
//  2. scan: every wave runs its share of the plan's tasks (host-built, balanced by cost):
//      * group task, 8 motif sizes k0..k0+7 with M(k) >= 15: a run of >= 15 matches contains an aligned
//        group of 8 rows that all match.  Per (group, k) the 8 rows of (H^H')|(L^L') are OR-ed with 16
//        v_bitop3_b32; a zero bit whose previous group was not all-match (or that is the first group of
//        its stream) is a candidate.
//      * exact task, one motif size with M(k) = M < 15 (templated on M): sliding OR over exactly M rows; a zero
//        bit whose previous row is a mismatch (or that is row 0 of its stream) is a candidate.
//     The loops over the four 8-row blocks of a stream are rolled and k0 is a run-time LDS offset, so the
//     whole scan is a few KB of code that stays in the instruction cache (a fully unrolled per-parameter
//     version measured 8x slower: 140 KB of straight-line code).
//     Candidates leave the scan as 8-byte records (group position, k or k0, 8-bit mask of rows or k's)
//     in a per-wave LDS list -- no atomics.  A wave whose list fills up verifies it on the spot.
//  3. verify: all lanes expand the remaining records and turn each candidate into a row or nothing
//     (verify_impl.h), reading the LDS window; rows go to the tile's slab in HBM.
// Exactness argument: DESIGN.md.
#include <algorithm>
#include <utility>
#include <vector>

#include "prf_host.h"
#include "scan_vertical.h"
#include "verify_impl.h"

// Diagnostic build only (make STAMPS=1 -> libprf_stamps.so): per-wave s_memtime stamps at the phase
// boundaries, written to a debug buffer that nothing else reads.  The product build has no stamp.
#ifdef PRF_STAMPS
#define PRF_STAMP(i)                                                                          \
    do {                                                                                      \
        if (g.dbg && lane == 0) g.dbg[((u64)blockIdx.x * MAX_WAVES + wave) * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define PRF_STAMP(i) do { } while (0)
#endif

namespace {

constexpr int T = 32;      // rows (= positions) per stream
constexpr int RG = T / 4;  // row groups of 4 rows = one 16-byte slot per lane
constexpr int LIN_PRE = 1;                                    // linear window: words before the tile
constexpr int LIN_POST = 24;                                  // ... and after it
constexpr int LW = (int)PRF_TILE_WORDS + LIN_PRE + LIN_POST;  // words per plane in the LDS window
constexpr int REC_PER_WAVE = 128;                             // candidate records per wave (LDS list)
constexpr int MAX_WAVES = 4;
constexpr int SMALL_M = 15;                                   // M(k) below this -> exact task

template <int A, class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, A + I>{}), ...);
}
// f(integral_constant<int,i>) for i in [A, B)
template <int A, int B, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B > A) static_for_impl<A>(static_cast<F &&>(f), std::make_integer_sequence<int, B - A>{});
}

// ---- candidate records: [36:0] position/8, [38:37] kind, [47:39] k (START) or k0 (GROUP*), [55:48] mask ----
__device__ __forceinline__ u64 make_rec(u64 pos, u64 kind, u32 k, u32 mask) {
    return (pos >> 3) | (kind << 37) | ((u64)k << 39) | ((u64)mask << 48);
}

// dynamic LDS: [TileCtx][rec_cnt: MAX_WAVES u32][hit_cnt] (192 B) [vimg: P*RG*nc uint4][lin: 2*LW u64][recs: MAX_WAVES*REC_PER_WAVE u64]
// [cof: plan.cof_words u32].  P = 3: 48.3 KB, 3 workgroups per CU.  P = 2: the third plane of a tile with N in reach overlays
// lin, 39.9 KB, 4 workgroups per CU.
extern __shared__ __attribute__((aligned(16))) unsigned char prf_smem[];
constexpr int SMEM_HDR = 192;

// what the verification step needs about the tile; lives at the start of LDS (filled by thread 0 while staging)
struct TileCtx {
    u64 w0;                   // first word of the linear window
    u64 xz_lo, xz_hi;         // positions known to hold no not-ACGT symbol
    const u64 *H, *L, *X;     // linear planes in HBM
    prf_hit_dev *slab;        // this tile's row slab in HBM
    u64 contig_base;          // a tile lies inside one contig
    u32 contig;
    u32 hit_cap;
    u32 min_repeats, min_span;
    u32 lin_off;              // byte offset of the linear window in LDS
    u32 x_in_lds;             // the window holds X too
    u32 has_lin;              // the linear window is staged (clean tiles)
#ifdef PRF_STAMPS
    u64 *dbg;
#endif
};
static_assert(sizeof(TileCtx) <= 128, "TileCtx must fit its LDS header slot");

__device__ __forceinline__ u32 *smem_rec_cnt() { return reinterpret_cast<u32 *>(prf_smem + 128); }
__device__ __forceinline__ u32 *smem_hit_cnt() { return reinterpret_cast<u32 *>(prf_smem + 128 + 4 * MAX_WAVES); }

// cof[k]: the cofactors k/p of the distinct primes p | k, one per byte, largest first (k <= 480 has at most 4
// distinct primes and k/p <= 240).  The motif seq[a:a+k] is primitive iff it has none of these periods
// (reference consists_of_perfect_repeats, utils/perfect_repeat_tracker.py:108-142, tries every divisor).
struct CofTable {
    u32 v[PRF_VMAX_K + 4];
    constexpr CofTable() : v{} {
        for (u32 k = 2; k <= PRF_VMAX_K; k++) {
            u32 rest = k, packed = 0, n = 0;
            for (u32 p = 2; p <= rest; p++) {
                if (rest % p) continue;
                packed |= (k / p) << (8 * n++);
                while (rest % p == 0) rest /= p;
            }
            v[k] = packed;
        }
    }
};
__constant__ const CofTable prf_cof_table{};
// LDS copy behind the candidate lists: entries 0 .. kmax of the scan, rounded up to 4 words (plan.cof_words)

// ---- lean verification for the common case: a candidate well inside a clean tile ----
// All looks are 32 positions wide and read the LDS window only (H and L; the not-ACGT plane is known to be
// zero there).  Positions are window-relative bit offsets (bit 0 = 64 positions before the tile).
typedef __attribute__((address_space(3))) const u32 prf_lds_cu32;

__device__ __forceinline__ u32 look32(prf_lds_cu32 *plane, u32 q) {
    const u32 w = q >> 5;
    return __builtin_amdgcn_alignbit(plane[w + 1], plane[w], q & 31u);
}
// mismatch bits of window positions q .. q+31 against q+k ..
__device__ __forceinline__ u32 mismatch32(prf_lds_cu32 *h, prf_lds_cu32 *l, u32 q, u32 k) {
    return (look32(h, q) ^ look32(h, q + k)) | (look32(l, q) ^ look32(l, q + k));
}
// does the word at window positions [a, a+k) have period d?  (k - d positions to compare)
__device__ __forceinline__ bool has_period32(prf_lds_cu32 *h, prf_lds_cu32 *l, u32 a, u32 k, u32 d) {
    const u32 need = k - d;
    for (u32 off = 0; off < need; off += 32) {
        u32 mm = mismatch32(h, l, a + off, d);
        const u32 left = need - off;
        if (left < 32) mm &= (1u << left) - 1u;
        if (mm) return false;
    }
    return true;
}
// 0: not a row; 1: row, run [a, b) in window positions; 2: outside the fast path's reach -> generic routine
__device__ __forceinline__ int fast_candidate(prf_lds_cu32 *h, prf_lds_cu32 *l, prf_lds_cu32 *cof, u32 p, u32 k, u32 kind, u32 min_repeats,
                                              u32 min_span, u32 lo_ok, u32 hi_ok, u32 &a_out, u32 &b_out) {
    if (p < lo_ok + 32u || p + k + 64u > hi_ok) return 2;
    u32 a = p;  // START (exact tasks): p is the first position of its run -- the tasks know the row before every
                // stream but a tile's first, and candidates there take the generic routine (p < lo_ok + 32)
    if (kind != (u32)PRF_KIND_START) {
        const u32 back = 8u << (kind - 1u);                   // 8, 16, 32
        const u32 before = mismatch32(h, l, p - 32u, k);      // bit 31 = position p-1
        const u32 nmatch = (u32)__builtin_clz(before | 1u);   // matches directly before p (31 if none seen: >= back then)
        if (before == 0 || nmatch >= back) return 0;          // an earlier examined group of the run reports
        a = p - nmatch;
    }
    u32 b = p;
    for (;;) {
        if (b + k + 64u > hi_ok) return 2;
        const u32 mm = mismatch32(h, l, b, k);
        if (mm) {
            b += (u32)__builtin_ctz(mm);
            break;
        }
        b += 32u;
    }
    if ((long long)(b - a) < prf_min_matches(k, min_repeats, min_span)) return 0;
    // primitive motif: no period k/p for a prime p | k
    // (START: the periods 1 and 2 were decided when the record was pushed, see Emit::push_start)
    const u32 dmin = kind == (u32)PRF_KIND_START ? 3u : 1u;
    for (u32 cf = cof[k]; cf; cf >>= 8) {
        const u32 d = cf & 255u;
        if (d >= dmin && has_period32(h, l, a, k, d)) return 0;
    }
    a_out = a;
    b_out = b;
    return 1;
}

// Candidate records -> rows.  only_list >= 0: the records [0, n) of that wave's list, taken by lanes
// first, first+stride, ... (a wave emptying its own full list in the middle of the scan).  only_list < 0:
// the records of all lists, as one index space [0, n) (the cooperative pass at the end of the tile).
__device__ __forceinline__ void verify_records_impl(prf_lds_cu64 *recs, int only_list, u32 n, u32 first, u32 stride) {
    const TileCtx &tc = *reinterpret_cast<const TileCtx *>(prf_smem);
    prf_window_view view;
    view.lds = (prf_lds_cu64 *)(prf_smem + tc.lin_off);
    view.w0 = tc.w0;
    view.nwords = tc.has_lin ? LW : 0;  // 0: every look goes to the global planes
    view.xz_lo = tc.xz_lo;
    view.xz_hi = tc.xz_hi;
    view.x_in_lds = tc.x_in_lds;
    view.P[0] = tc.H; view.P[1] = tc.L; view.P[2] = tc.X;
    const u32 min_repeats = tc.min_repeats, min_span = tc.min_span, hit_cap = tc.hit_cap, contig = tc.contig;
    const u64 contig_base = tc.contig_base;
    prf_hit_dev *slab = tc.slab;
    u32 *hit_cnt = smem_hit_cnt();
    // fast path: window-relative positions; valid where the window is staged and holds no not-ACGT position
    const u64 win_pos0 = tc.w0 * 64;
    prf_lds_cu32 *fh = (prf_lds_cu32 *)(prf_smem + tc.lin_off);
    prf_lds_cu32 *fl = fh + 2 * LW;
    prf_lds_cu32 *cof = fl + 2 * LW + 2 * MAX_WAVES * REC_PER_WAVE;  // behind the candidate lists
    const u32 fast_lo = 64u;                                             // the tile starts 64 positions into the window
    const u32 fast_hi = tc.xz_hi > tc.xz_lo ? (u32)LW * 64u - 64u : 0u;  // 0: tile with N in reach, no fast path
    const u32 *rec_cnt = smem_rec_cnt();
    const u32 c0 = rec_cnt[0], c1 = c0 + rec_cnt[1], c2 = c1 + rec_cnt[2];
    for (u32 idx = first; idx < n; idx += stride) {
        u32 slot_idx;
        if (only_list >= 0) slot_idx = (u32)only_list * REC_PER_WAVE + idx;
        else if (idx < c0) slot_idx = idx;
        else if (idx < c1) slot_idx = REC_PER_WAVE + (idx - c0);
        else if (idx < c2) slot_idx = 2 * REC_PER_WAVE + (idx - c1);
        else slot_idx = 3 * REC_PER_WAVE + (idx - c2);
        const u64 rec = recs[slot_idx];
#ifdef PRF_STAMPS
        const u64 vt0 = __builtin_amdgcn_s_memtime();
        u32 vslow = 0, vwalk = 0;
#endif
        const u64 p8 = (rec & ((1ull << 37) - 1ull)) << 3;
        const u32 kind = (u32)(rec >> 37) & 3u;
        const u32 kk = (u32)(rec >> 39) & 511u;
        u32 mask = (u32)(rec >> 48) & 255u;
        while (mask) {
            const u32 bit = (u32)__builtin_ctz(mask);
            mask &= mask - 1;
            const u64 p = kind == (u32)PRF_KIND_START ? p8 + bit : p8;  // START: the mask selects rows, else motif sizes
            const u32 k = kind == (u32)PRF_KIND_START ? kk : kk + bit;
            u64 a, b;
            u32 fa, fb;
            int st = 2;
            if (fast_hi) st = fast_candidate(fh, fl, cof, (u32)(p - win_pos0), k, kind, min_repeats, min_span, fast_lo, fast_hi, fa, fb);
            if (st == 1) {
                a = win_pos0 + fa;
                b = win_pos0 + fb;
            } else if (st == 2) {
#ifdef PRF_STAMPS
                vslow++;
#endif
                st = prf_candidate_to_run(view, p, k, kind, min_repeats, min_span, a, b) ? 1 : 0;
            }
#ifdef PRF_STAMPS
            if (st == 1) vwalk += (u32)(b - a);
#endif
            if (st == 1) {
                const u32 slot = atomicAdd(hit_cnt, 1u);
                if (slot < hit_cap) {
                    prf_hit_dev h;
                    h.start = a - contig_base;
                    h.end = b + k - contig_base;
                    h.k = k;
                    h.contig = contig;
                    slab[slot] = h;
                }
            }
        }
#ifdef PRF_STAMPS
        if (only_list < 0 && tc.dbg) {
            u64 *d = tc.dbg + (1ull << 24) + ((u64)blockIdx.x * 1024 + idx) * 2;
            d[0] = __builtin_amdgcn_s_memtime() - vt0;
            d[1] = (u64)kind | ((u64)kk << 8) | ((u64)vslow << 24) | ((u64)vwalk << 32) | ((rec >> 48 & 255ull) << 56);
        }
#endif
    }
}

// A wave emptying its own full list in the middle of the scan: rare, and called from inside every task, so not inlined
// (a function call: the callee saves the registers it uses to scratch memory).
__device__ __noinline__ void verify_records(prf_lds_cu64 *recs, int only_list, u32 n, u32 first, u32 stride) {
    verify_records_impl(recs, only_list, n, first, stride);
}

// bit t of the result: bits t .. t+M-1 of z are all ones (M <= 16)
template <int M>
__device__ __forceinline__ u32 ones_run(u32 z) {
    if constexpr (M == 1) return z;
    else {
        constexpr int L = M >= 8 ? 8 : (M >= 4 ? 4 : 2);
        u32 r = z & (z >> 1);
        if constexpr (L >= 4) r &= r >> 2;
        if constexpr (L >= 8) r &= r >> 4;
        return r & (r >> (M - L));
    }
}

struct Emit {
    u64 *recs;           // this wave's list in LDS, REC_PER_WAVE records
    u64 *all_recs;       // all lists
    int wave;
    u32 cnt;             // records in it (wave-uniform)
    u32 flushed;         // records verified in early flushes (wave-uniform)
    u64 lane_pos;        // tile base + lane*32
    int lane;
    prf_lds_cu32 *lin_h, *lin_l;  // the linear window in LDS
    u32 win_q;           // window position of the lane's first stream: 64 + lane*32
    bool filter;         // the tile has a linear window: echo filter on

    // Exact tasks.  Like push(), plus the echo filter: a candidate at position t says that [t, t+M+k) has period
    // k and holds no N.  If the M positions from t on ALSO all match at a shift d < k, then [t, t+M+d) has both
    // periods, M+d >= k+d-gcd(k,d) because M >= k, so by Fine and Wilf it has period gcd(k,d) -- a proper divisor of
    // k: the motif seq[t:t+k] is a power of a shorter word and the reference's consists_of_perfect_repeats test
    // (utils/perfect_repeat_tracker.py:108-142) would drop the row.  Such echoes of homopolymer and dinucleotide
    // repeats are about half of all exact candidates on genomic sequence.  The test reads 32 positions of the
    // linear window (LDS) per record: shifts 1 and 2, all 8 rows of the record at once.
    template <int M>
    __device__ __forceinline__ void push_start(u32 hot, const u32 (&c)[8], int row, u32 k) {
        while (__builtin_amdgcn_ballot_w64(hot != 0) != 0) {
            u32 mask = 0, b = 0;
            if (hot) {
                b = (u32)__builtin_ctz(hot);
                hot &= hot - 1;
                static_for<0, 8>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    mask |= __builtin_amdgcn_ubfe(c[i], b, 1u) << i;  // v_bfe_u32 + v_lshl_or_b32
                });
                if (k > 1 && filter) {  // wave-uniform
                    const u32 q0 = win_q + b * (64u * T) + (u32)row;
                    const u32 wh = look32(lin_h, q0), wl = look32(lin_l, q0);
                    mask &= ~ones_run<M>(~((wh ^ (wh >> 1)) | (wl ^ (wl >> 1))));
                    // (shift 2 only for even k: with k odd, periods k and 2 give period 1, which the line above caught)
                    if (k > 2 && !(k & 1u)) mask &= ~ones_run<M>(~((wh ^ (wh >> 2)) | (wl ^ (wl >> 2))));
                }
            }
            const u64 bal = __builtin_amdgcn_ballot_w64(mask != 0);
            if (bal == 0) continue;
            const u32 n = (u32)__builtin_popcountll(bal);
            if (cnt + n > (u32)REC_PER_WAVE) {  // wave-uniform: list full -> this wave verifies it now
                verify_records((prf_lds_cu64 *)all_recs, wave, cnt, (u32)lane, 64u);
                flushed += cnt;
                cnt = 0;
            }
            if (mask) {
                const u32 idx = cnt + __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0));
                recs[idx] = make_rec(lane_pos + (u64)b * (64u * T) + (u64)row, PRF_KIND_START, k, mask);
            }
            cnt += n;
        }
    }

    // Group tasks.  Every lane of the wave calls this together.  `hot`: bit b set = stream (lane, b) reports for the
    // 8-row group starting at `row`; bit b of c[i] set = it reports for motif size k0+i.  One record per (stream,
    // motif size): a record that named several sizes would be verified by one lane, size after size, while the
    // other lanes of its wave wait.
    __device__ __forceinline__ void push(u32 hot, const u32 (&c)[8], int row, u64 kind, u32 k0, u32 valid) {
        u32 mask = 0, b = 0;
        while (__builtin_amdgcn_ballot_w64((hot | mask) != 0) != 0) {
            if (mask == 0 && hot) {
                b = (u32)__builtin_ctz(hot);
                hot &= hot - 1;
                static_for<0, 8>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    mask |= __builtin_amdgcn_ubfe(c[i], b, 1u) << i;  // v_bfe_u32 + v_lshl_or_b32
                });
                mask &= valid;  // motif sizes of the chunk that are outside the scan's range
            }
            const u64 bal = __builtin_amdgcn_ballot_w64(mask != 0);
            const u32 n = (u32)__builtin_popcountll(bal);
            if (cnt + n > (u32)REC_PER_WAVE) {  // wave-uniform: list full -> this wave verifies it now
                verify_records((prf_lds_cu64 *)all_recs, wave, cnt, (u32)lane, 64u);
                flushed += cnt;
                cnt = 0;
            }
            if (mask) {
                const u32 one = mask & (0u - mask);
                mask ^= one;
                const u32 idx = cnt + __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0));
                recs[idx] = make_rec(lane_pos + (u64)b * (64u * T) + (u64)row, kind, k0, one);
            }
            cnt += n;
        }
    }
};

// v_bitop3_b32: any boolean function of three words in one VALU operation.  Truth-table operands:
constexpr u32 TA = 0xF0, TB = 0xCC, TC = 0xAA;
template <u32 TT>
__device__ __forceinline__ u32 bitop3(u32 a, u32 b, u32 c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, TT);
}
// acc | (b ^ c)
__device__ __forceinline__ u32 or_xor(u32 acc, u32 b, u32 c) { return bitop3<(TA | (TB ^ TC)) & 0xFF>(acc, b, c); }
// ~(a | b) & c
__device__ __forceinline__ u32 nor_and(u32 a, u32 b, u32 c) { return bitop3<(~(TA | TB) & TC) & 0xFF>(a, b, c); }

__device__ __forceinline__ void unpack4(u32 *dst, const uint4 v) {
    dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
}

// LDS image addressing.  The image is [plane][row group][virtual lane] of 16-byte slots, NC virtual lanes wide
// (compile-time, so plane and row-group strides are instruction immediates).  Row group gg of a lane's
// *extended* stream (gg >= 8: the stream continues in the next virtual lane) is slot (gg & 7) * NC + (gg >> 3)
// from the lane's own slot.
template <int NC>
__device__ __forceinline__ const uint4 *slot_of(const uint4 *lane_base, int gg) {
    return lane_base + ((gg & 7) * NC + (gg >> 3));
}

// Slot g (compile-time) after a run-time first slot gg0 whose address `first` = slot_of(lane_base, gg0) and
// a = gg0 & 7 are computed once per block: the stream wraps into the next virtual lane at most once within a block.
template <int NC, int G>
__device__ __forceinline__ const uint4 *slot_after(const uint4 *first, int a) {
    return first + G * NC + (a + G >= 8 ? 1 - 8 * NC : 0);
}

// ---- group task: motif sizes k0 .. k0+7 (those in `valid`), the 8-row blocks tb0 .. tb1-1 of the stream ----
// S1: every block is examined (stride 1) and a group reports only if the group before it was not all-match; otherwise
// (stride 2 / 4) every examined all-match group reports.
template <bool HASX, int NC, bool S1>
__device__ __forceinline__ void group_task(const uint4 *vimg, int lane, u32 k0, u32 valid, u32 stride, int tb0, int tb1, Emit &em) {
    constexpr int NP = HASX ? 3 : 2;
    constexpr int PS = RG * NC;  // slots per plane
    const uint4 *lane_base = vimg + lane;
    u32 prev[8];
    static_for<0, 8>([&](auto ic) { prev[decltype(ic)::value] = ~0u; });  // first group of a stream: report, verify decides
    // stride 2 / 4: all motif sizes of the chunk have M(k) >= 23 / 39, and a run that long contains an aligned
    // all-match group whose index is a multiple of 2 / 4, so only those blocks are examined.  The group before
    // an examined one is then unknown: every examined all-match group reports, verify keeps the first of a run.
    const u64 kind = stride == 1 ? PRF_KIND_GROUP : (stride == 2 ? PRF_KIND_GROUP2 : PRF_KIND_GROUP4);
#pragma unroll 1
    for (int tb = tb0; tb < tb1; tb++) {
        if (tb & (int)(stride - 1)) continue;
        u32 a[3][8];   // rows 8tb .. 8tb+7
        u32 w[3][16];  // rows 8tb+k0 .. 8tb+k0+15 (k0 % 4 == 0: whole 16-byte slots)
        const uint4 *pa = lane_base + 2 * tb * NC;
        static_for<0, NP>([&](auto pc) {
            constexpr int p = decltype(pc)::value;
            unpack4(&a[p][0], pa[p * PS]);
            unpack4(&a[p][4], pa[p * PS + NC]);
        });
        const int g0 = 2 * tb + (int)(k0 >> 2);
        const uint4 *pw0 = slot_of<NC>(lane_base, g0);
        const int wa = g0 & 7;
        static_for<0, 4>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            const uint4 *pw = slot_after<NC, g>(pw0, wa);
            static_for<0, NP>([&](auto pc) {
                constexpr int p = decltype(pc)::value;
                unpack4(&w[p][4 * g], pw[p * PS]);
            });
        });
        // All 8 motif sizes are computed in one straight-line block so that their 8 independent OR chains
        // interleave (a chain alone is 16 dependent operations); sizes outside `valid` are dropped when records are made.
        u32 cand[8];
        u32 hot = 0;
        static_for<0, 8>([&](auto kc) {
            constexpr int kk = decltype(kc)::value;
            // OR over the 8 rows of (H^H')|(L^L'): 16 operations, no per-row mismatch word
            u32 o = a[0][0] ^ w[0][kk];
            o = or_xor(o, a[1][0], w[1][kk]);
            static_for<1, 8>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                o = or_xor(o, a[0][i], w[0][kk + i]);
                o = or_xor(o, a[1][i], w[1][kk + i]);
            });
            if constexpr (HASX) {
                static_for<0, 8>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    o = o | a[2][i] | w[2][kk + i];
                });
            }
            u32 c = ~o;  // all 8 rows match ...
            if constexpr (S1) {
                c &= prev[kk];  // ... and the previous group did not (or is unknown: first block)
                prev[kk] = o;
            }
            cand[kk] = c;
            hot |= c;
        });

        if (__builtin_amdgcn_ballot_w64(hot != 0) != 0) em.push(hot, cand, 8 * tb, kind, k0, valid);
    }
}

// candidate word of row t (relative to the block): rows t .. t+M-1 all match and row t-1 does not.
// m is indexed by row+1 (m[0] = the row before the block), o3[t] = OR of rows t..t+2.
template <int M, int t, int LM, int LO>
__device__ __forceinline__ u32 start_word(const u32 (&m)[LM], const u32 (&o3)[LO]) {
    const u32 before = m[t];
    if constexpr (M == 1) return ~m[t + 1] & before;
    else if constexpr (M == 2) return nor_and(m[t + 1], m[t + 2], before);
    else if constexpr (M == 3) return ~o3[t] & before;
    else if constexpr (M <= 6) return nor_and(o3[t], o3[t + M - 3], before);
    else if constexpr (M <= 9) return ~(o3[t] | o3[t + 3] | o3[t + M - 3]) & before;
    else if constexpr (M <= 12) return nor_and(o3[t] | o3[t + 3] | o3[t + 6], o3[t + M - 3], before);
    else return ~((o3[t] | o3[t + 3] | o3[t + 6]) | o3[t + 9] | o3[t + M - 3]) & before;
}

// ---- exact task: one motif size k whose minimum run length is M < 15; O = k % 4 ----
template <int M, int O, bool HASX, int NC>
__device__ __forceinline__ void exact_task(const uint4 *vimg, int lane, u32 k, int tb0, int tb1, Emit &em) {
    constexpr int NP = HASX ? 3 : 2;
    constexpr int PS = RG * NC;
    constexpr int NR = 8 + M - 1;          // mismatch words needed per block: rows t0 .. t0+NR-1
    constexpr int NGB = (NR + 3) / 4;      // 16-byte slots of base rows
    constexpr int NGS = (O + NR + 3) / 4;  // 16-byte slots of the rows shifted by k (first one starts O rows early)
    const uint4 *lane_base = vimg + lane;
    u32 mprev = ~0u;  // mismatch word of the row before the block; unknown -> report, verify decides
    if (tb0 == 0) {
        // Row -1 of stream (lane, b) is row T-1 of stream (lane-1, b); for lane 0 it is row T-1 of stream (63, b-1):
        // lane 63's word one bit up, with bit 0 (the previous tile's last stream) unknown.  Its partner, row k-1 of
        // the own stream, lies in the first slots (k <= 14).  With this, a reported start IS the start of its run
        // everywhere but at the first position of a tile.
        const int pl = (lane + 63) & 63;
        const uint4 *pp = vimg + pl + (RG - 1) * NC;
        const uint4 *ps = lane_base + (int)((k - 1) >> 2) * NC;
        u32 v = 0;
        static_for<0, NP>([&](auto pc) {
            constexpr int p = decltype(pc)::value;
            u32 prev_row = pp[p * PS].w;
            if (lane == 0) prev_row <<= 1;
            const uint4 sv = ps[p * PS];
            constexpr int c = (O + 3) & 3;  // (k - 1) % 4
            const u32 own_row = c == 0 ? sv.x : (c == 1 ? sv.y : (c == 2 ? sv.z : sv.w));
            if constexpr (p < 2) v |= prev_row ^ own_row;
            else v |= prev_row | own_row;
        });
        mprev = lane == 0 ? (v | 1u) : v;
    }
#pragma unroll 1
    for (int tb = tb0; tb < tb1; tb++) {
        u32 a[3][4 * NGB];
        u32 s[3][4 * NGS];
        const int gs0 = 2 * tb + (int)(k >> 2);
        const uint4 *pb0 = lane_base + 2 * tb * NC;  // slot 2*tb < 8: no wrap yet
        const uint4 *ps0 = slot_of<NC>(lane_base, gs0);
        const int ba = 2 * tb, sa = gs0 & 7;
        static_for<0, NGB>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            const uint4 *pb = slot_after<NC, g>(pb0, ba);
            static_for<0, NP>([&](auto pc) {
                constexpr int p = decltype(pc)::value;
                unpack4(&a[p][4 * g], pb[p * PS]);
            });
        });
        static_for<0, NGS>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            const uint4 *pw = slot_after<NC, g>(ps0, sa);
            static_for<0, NP>([&](auto pc) {
                constexpr int p = decltype(pc)::value;
                unpack4(&s[p][4 * g], pw[p * PS]);
            });
        });
        u32 m[NR + 1];
        u32 o3[NR];
        m[0] = mprev;
        static_for<0, NR>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            u32 v = or_xor(a[0][i] ^ s[0][O + i], a[1][i], s[1][O + i]);
            if constexpr (HASX) v = v | a[2][i] | s[2][O + i];
            m[i + 1] = v;
        });
        if constexpr (M >= 3) {
            static_for<0, NR - 2>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                o3[i] = m[i + 1] | m[i + 2] | m[i + 3];
            });
        }
        u32 cand[8];
        u32 hot = 0;
        static_for<0, 8>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            cand[t] = start_word<M, t>(m, o3);
            hot |= cand[t];
        });
        mprev = m[8];
        if (__builtin_amdgcn_ballot_w64(hot != 0) != 0) em.template push_start<M>(hot, cand, 8 * tb, k);
    }
}

template <int M, bool HASX, int NC>
__device__ __forceinline__ void exact_task_any(const uint4 *vimg, int lane, u32 k, int tb0, int tb1, Emit &em) {
    switch (k & 3u) {  // wave-uniform
        case 0: exact_task<M, 0, HASX, NC>(vimg, lane, k, tb0, tb1, em); break;
        case 1: exact_task<M, 1, HASX, NC>(vimg, lane, k, tb0, tb1, em); break;
        case 2: exact_task<M, 2, HASX, NC>(vimg, lane, k, tb0, tb1, em); break;
        default: exact_task<M, 3, HASX, NC>(vimg, lane, k, tb0, tb1, em); break;
    }
}

template <bool HASX, int NC>
__device__ __forceinline__ void run_tasks(const uint4 *vimg, const prf_vplan &plan, int wave, int lane, int tb0, int tb1, Emit &em,
                                          u64 *dbg = nullptr) {
    const u32 t_end = plan.wave_begin[wave + 1];
    for (u32 ti = plan.wave_begin[wave]; ti < t_end; ti++) {
        const prf_vtask task = plan.tasks[ti];
#ifdef PRF_STAMPS
        if (dbg && lane == 0) dbg[8 + (ti - plan.wave_begin[wave])] = __builtin_amdgcn_s_memtime();
#endif
        switch (task.kind) {
            case 0:
                if (task.stride == 1) group_task<HASX, NC, true>(vimg, lane, task.k0, task.valid, 1u, tb0, tb1, em);
                else group_task<HASX, NC, false>(vimg, lane, task.k0, task.valid, task.stride, tb0, tb1, em);
                break;
            case 1: exact_task_any<1, HASX, NC>(vimg, lane, task.k0, tb0, tb1, em); break;
            case 2: exact_task_any<2, HASX, NC>(vimg, lane, task.k0, tb0, tb1, em); break;
            case 3: exact_task_any<3, HASX, NC>(vimg, lane, task.k0, tb0, tb1, em); break;
            case 4: exact_task_any<4, HASX, NC>(vimg, lane, task.k0, tb0, tb1, em); break;
            case 5: exact_task_any<5, HASX, NC>(vimg, lane, task.k0, tb0, tb1, em); break;
            case 6: exact_task_any<6, HASX, NC>(vimg, lane, task.k0, tb0, tb1, em); break;
            case 7: exact_task_any<7, HASX, NC>(vimg, lane, task.k0, tb0, tb1, em); break;
            case 8: exact_task_any<8, HASX, NC>(vimg, lane, task.k0, tb0, tb1, em); break;
            case 9: exact_task_any<9, HASX, NC>(vimg, lane, task.k0, tb0, tb1, em); break;
            case 10: exact_task_any<10, HASX, NC>(vimg, lane, task.k0, tb0, tb1, em); break;
            case 11: exact_task_any<11, HASX, NC>(vimg, lane, task.k0, tb0, tb1, em); break;
            case 12: exact_task_any<12, HASX, NC>(vimg, lane, task.k0, tb0, tb1, em); break;
            case 13: exact_task_any<13, HASX, NC>(vimg, lane, task.k0, tb0, tb1, em); break;
            default: exact_task_any<14, HASX, NC>(vimg, lane, task.k0, tb0, tb1, em); break;
        }
    }
}

// Grid: first 4 workgroups per tile with N in reach (the 3-plane variant is slower, so each takes one of the
// four 8-row blocks and they start first), then one workgroup per clean tile.
// OCC = workgroups per CU the registers are budgeted for.  39.9 KB of LDS allow 4; at 4 the compiler has 128 VGPRs
// instead of 168 and spills a little (measured: 9-11 % faster on multi-round launches, 3 % slower per workgroup), so
// launches that fit one round at 3 per CU take the OCC = 3 build.
template <int NC, int OCC>
__global__ __launch_bounds__(64 * MAX_WAVES, OCC) void prf_vscan_kernel(prf_vscan_args g) {
    constexpr int nc = NC;
    uint4 *vimg = reinterpret_cast<uint4 *>(prf_smem + SMEM_HDR);
    // OCC == 4 (39.9 KB): the third (not-ACGT) plane of a tile with N in reach lies where a clean tile keeps its
    // linear window; such a tile then verifies on the global planes and takes no echo filter.  OCC == 3 (48.3 KB,
    // one-round launches, where the slowest workgroup sets the kernel time): room for both.
    constexpr int LDS_PLANES = OCC >= 4 ? 2 : 3;
    const u32 lin_off = (u32)SMEM_HDR + (u32)((size_t)LDS_PLANES * RG * nc * sizeof(uint4));
    u64 *lin = reinterpret_cast<u64 *>(prf_smem + lin_off);
    u64 *recs = lin + 2 * LW;
    u32 *rec_cnt = smem_rec_cnt();
    u32 *hit_cnt = smem_hit_cnt();

    const int nt = (int)blockDim.x;
    const int tid = (int)threadIdx.x;
    const bool hasx = blockIdx.x < 4u * g.n_mixed;
    const u32 part = hasx ? (blockIdx.x & 3u) : 0u;
    const int tb0 = hasx ? (int)part : 0, tb1 = hasx ? (int)part + 1 : 4;
    // the list holds the clean tiles first, then the mixed ones
    // (when the clean tiles are one contiguous range -- a contig without interior N blocks -- the tile index is
    // arithmetic: no dependent load in front of the staging loads)
    const u64 tile = hasx ? g.tile_list[g.n_clean + (blockIdx.x >> 2)]
                          : (g.clean_base != ~0u ? g.clean_base + (blockIdx.x - 4u * g.n_mixed)
                                                 : g.tile_list[blockIdx.x - 4u * g.n_mixed]);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int extra = nc - 64;

    PRF_STAMP(0);
    // ---- 1. stage ----
    // All global loads of a thread are issued back to back before the first LDS store, so the workgroup pays
    // one memory round trip (a load -> store loop pays one per iteration: measured 7 k cycles per tile).
    {
        const uint4 *ph = reinterpret_cast<const uint4 *>(g.VH), *pL = reinterpret_cast<const uint4 *>(g.VL),
                    *px = reinterpret_cast<const uint4 *>(g.VX);
        const int np = hasx ? 3 : 2;
        const long long w0 = (long long)(tile * PRF_TILE_WORDS) - LIN_PRE;  // the planes have readable padding in front
        if (nt == 64 * MAX_WAVES) {
            // 256 threads: H and L bit-sliced planes = 4 slots per thread, X plane (tiles with N) 2 more;
            // linear window = 8 words per thread + a tail; virtual lanes: one slot pair for the first threads
            constexpr int NTH = 64 * MAX_WAVES;
            constexpr int NLF = (2 * LW) / NTH;          // full rounds of linear words
            constexpr int NLT = (2 * LW) - NLF * NTH;    // tail
            const int rg = tid >> 6, l = tid & 63;       // slot (rg + 4*j, l) of plane p
            const uint4 *th = ph + (tile * RG + rg) * 64 + l, *tl = pL + (tile * RG + rg) * 64 + l;
            const uint4 vh0 = th[0], vh1 = th[4 * 64], vl0 = tl[0], vl1 = tl[4 * 64];
            uint4 vx0 = make_uint4(0, 0, 0, 0), vx1 = vx0;
            if (hasx) {
                const uint4 *tx = px + (tile * RG + rg) * 64 + l;
                vx0 = tx[0];
                vx1 = tx[4 * 64];
            }
            u64 lw[NLF];
            static_for<0, NLF>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int p = (i * NTH) / LW;  // which plane the round starts in (a round may straddle H -> L)
                const int idx = tid + i * NTH;
                lw[i] = idx < LW ? g.H[w0 + idx] : g.L[w0 + idx - LW];
                (void)p;
            });
            u64 lt = 0;
            if (tid < NLT) lt = g.L[w0 + (NLF * NTH + tid) - LW];
            // virtual-lane slots: (plane, row group, first lanes again).  3 planes x 8 row groups x 16 extra lanes (NC = 80,
            // tile with N in reach) are 384 slots: two rounds of the 256 threads cover every instantiated width.
            constexpr int NXR = (3 * RG * (NC - 64) + NTH - 1) / NTH;  // 1 for NC = 66, 72; 2 for NC = 80
            static_assert(3 * RG * (NC - 64) <= NXR * NTH, "virtual-lane staging rounds do not cover the image width");
            uint4 ev[NXR], en[NXR];
            const int n_extra = np * RG * extra;
            static_for<0, NXR>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const int s = tid + r * NTH;
                ev[r] = make_uint4(0, 0, 0, 0);
                en[r] = ev[r];
                if (s < n_extra) {
                    const int p = s / (RG * extra), erg = (s / extra) % RG, el = s % extra;
                    const uint4 *src = p == 0 ? ph : (p == 1 ? pL : px);
                    ev[r] = src[(tile * RG + erg) * 64 + el];
                    en[r] = src[((tile + 1) * RG + erg) * 64 + el];
                }
            });
            vimg[(0 * RG + rg) * nc + l] = vh0;
            vimg[(0 * RG + rg + 4) * nc + l] = vh1;
            vimg[(1 * RG + rg) * nc + l] = vl0;
            vimg[(1 * RG + rg + 4) * nc + l] = vl1;
            if (hasx) {
                vimg[(2 * RG + rg) * nc + l] = vx0;
                vimg[(2 * RG + rg + 4) * nc + l] = vx1;
            }
            if (LDS_PLANES == 3 || !hasx) {
                static_for<0, NLF>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    lin[tid + i * NTH] = lw[i];
                });
                if (tid < NLT) lin[NLF * NTH + tid] = lt;
            }
            static_for<0, NXR>([&](auto rc) {  // virtual lanes 64..: the first lanes again, one bit up, bit 31 from the next tile
                constexpr int r = decltype(rc)::value;
                const int s = tid + r * NTH;
                if (s < n_extra) {
                    const int p = s / (RG * extra), erg = (s / extra) % RG, el = s % extra;
                    uint4 v;
                    v.x = (ev[r].x >> 1) | (en[r].x << 31);
                    v.y = (ev[r].y >> 1) | (en[r].y << 31);
                    v.z = (ev[r].z >> 1) | (en[r].z << 31);
                    v.w = (ev[r].w >> 1) | (en[r].w << 31);
                    vimg[(p * RG + erg) * nc + 64 + el] = v;
                }
            });
        } else {
            for (int idx = tid; idx < np * RG * 64; idx += nt) {
                const int p = idx / (RG * 64), rg = (idx / 64) % RG, l = idx % 64;
                const uint4 *src = p == 0 ? ph : (p == 1 ? pL : px);
                const uint4 v = src[(tile * RG + rg) * 64 + l];
                vimg[(p * RG + rg) * nc + l] = v;
                if (l < extra) {
                    const uint4 nx = src[((tile + 1) * RG + rg) * 64 + l];
                    uint4 r;
                    r.x = (v.x >> 1) | (nx.x << 31);
                    r.y = (v.y >> 1) | (nx.y << 31);
                    r.z = (v.z >> 1) | (nx.z << 31);
                    r.w = (v.w >> 1) | (nx.w << 31);
                    vimg[(p * RG + rg) * nc + 64 + l] = r;
                }
            }
            for (int idx = tid; idx < ((LDS_PLANES == 2 && hasx) ? 0 : 2 * LW); idx += nt) {
                const int p = idx / LW, j = idx % LW;
                lin[idx] = (p == 0 ? g.H : g.L)[w0 + j];
            }
        }
        {
            u32 *cof_lds = reinterpret_cast<u32 *>(recs + MAX_WAVES * REC_PER_WAVE);
            for (int i = tid; i < (int)g.plan.cof_words; i += nt) cof_lds[i] = prf_cof_table.v[i];
        }
        if (tid < 2 * MAX_WAVES + 1 && tid != MAX_WAVES) rec_cnt[tid] = 0;  // list lengths, [MAX_WAVES] = row count, flushed counts
        if (tid == 0) {
            *hit_cnt = 0;
            TileCtx tc;
            tc.w0 = tile * PRF_TILE_WORDS - LIN_PRE;
            tc.xz_lo = hasx ? 0 : tile * PRF_TILE;  // a clean tile and its successor hold no not-ACGT position
            tc.xz_hi = hasx ? 0 : (tile + 2) * PRF_TILE;
            tc.H = g.H; tc.L = g.L; tc.X = g.X;
            tc.slab = g.hit_slabs + (tile * 4 + part) * (u64)g.hit_cap;
            // contigs start on tile boundaries, so every run that this tile reports lies in the tile's contig
            tc.contig = prf_contig_of(g.contig_base, g.n_contigs, tile * PRF_TILE);
            tc.contig_base = g.contig_base[tc.contig];
            tc.hit_cap = g.hit_cap;
            tc.min_repeats = g.min_repeats;
            tc.min_span = g.min_span;
            tc.lin_off = lin_off;
            tc.x_in_lds = 0u;
            tc.has_lin = (LDS_PLANES == 2 && hasx) ? 0u : 1u;
#ifdef PRF_STAMPS
            tc.dbg = g.dbg;
#endif
            *reinterpret_cast<TileCtx *>(prf_smem) = tc;
        }
    }
    PRF_STAMP(1);
    __syncthreads();
    PRF_STAMP(2);

    // ---- 2. scan ----
    Emit em;
    em.recs = recs + wave * REC_PER_WAVE;
    em.all_recs = recs;
    em.wave = wave;
    em.cnt = 0;
    em.flushed = 0;
    em.lane_pos = tile * PRF_TILE + (u64)lane * T;
    em.lane = lane;
    em.lin_h = (prf_lds_cu32 *)(prf_smem + lin_off);
    em.lin_l = em.lin_h + 2 * LW;
    em.win_q = 64u + (u32)lane * T;
    em.filter = LDS_PLANES == 3 || !hasx;
#ifdef PRF_STAMPS
    u64 *task_dbg = g.dbg ? g.dbg + ((u64)blockIdx.x * MAX_WAVES + wave) * 16 : nullptr;
#else
    u64 *task_dbg = nullptr;
#endif
    if (hasx) run_tasks<true, NC>(vimg, g.plan, wave, lane, tb0, tb1, em, task_dbg);
    else run_tasks<false, NC>(vimg, g.plan, wave, lane, tb0, tb1, em, task_dbg);
    if (lane == 0) {  // waves the plan does not use keep the 0 from staging
        rec_cnt[wave] = em.cnt;
        rec_cnt[MAX_WAVES + 1 + wave] = em.flushed;
    }
    PRF_STAMP(3);
    __syncthreads();
    PRF_STAMP(4);

    // ---- 3. verify what is left in the lists, all waves together: every candidate -> a row in the tile's slab, or nothing ----
    u32 total = 0;
    for (int w = 0; w < MAX_WAVES; w++) total += rec_cnt[w];
    // (statistics: the candidate-record count goes out now, so that the atomic is long acknowledged when the barrier
    // after the verification waits for outstanding memory operations)
    if (tid == 0) {
        u32 n_records = total;
        for (int w = 0; w < MAX_WAVES; w++) n_records += rec_cnt[MAX_WAVES + 1 + w];
        if (n_records)
            atomicAdd(&g.counters[PRF_CNT_SHARD0 + (tile % PRF_CNT_NSHARD) * PRF_CNT_SHARD_STRIDE + PRF_SH_CAND], (u64)n_records);
    }
    verify_records_impl((prf_lds_cu64 *)recs, -1, total, (u32)tid, (u32)nt);  // inlined: no call, no register saves
    PRF_STAMP(5);
    __syncthreads();
    PRF_STAMP(6);
    // ---- 4. the tile's rows -> the compact row array: ONE atomic per workgroup reserves its range (low 40 bits:
    // row cursor) and draws its finishing ticket (high 24 bits), then the slab (written by this workgroup, still in
    // this CU's L1/L2) is copied there.  The order of the ranges is whatever order the workgroups get here in; the
    // host sorts rows after the fetch anyway (reference perfect_repeat_finder.py:81).
    u64 *xch = reinterpret_cast<u64 *>(recs);  // the candidate lists are dead now
    const u32 n_rows = *hit_cnt;  // final since the barrier above
    const u32 stored = n_rows < g.hit_cap ? n_rows : g.hit_cap;
    if (tid == 0) {
        if (n_rows > g.hit_cap) {  // rare: wait for the result, so that the maximum is in place before the ticket is drawn
            const u64 prev = atomicMax(&g.counters[PRF_CNT_HIT_OVF], (u64)n_rows);
            asm volatile("" ::"v"(prev));
        }
        xch[0] = atomicAdd(&g.counters[PRF_CNT_ROWS], (u64)stored | (1ull << PRF_ROWS_TICKET_SHIFT));
    }
    // while thread 0 waits for its atomic, everybody reads the first words of the slab (up to 170 rows: all of them
    // on ordinary sequence); the two memory round trips overlap
    const u64 *src = reinterpret_cast<const u64 *>(g.hit_slabs + (tile * 4 + part) * (u64)g.hit_cap);
    const u32 n_words = 3u * stored;
    const u64 w0 = (u32)tid < n_words ? src[tid] : 0ull;
    const u64 w1 = (u32)tid + (u32)nt < n_words ? src[tid + nt] : 0ull;
    __syncthreads();
    const u64 base = xch[0] & ((1ull << PRF_ROWS_TICKET_SHIFT) - 1ull);
    const u64 ticket = xch[0] >> PRF_ROWS_TICKET_SHIFT;
    if (base + stored <= g.rows_cap) {  // else: the host sees the cursor beyond the capacity, grows the array, rescans
        u64 *dst = reinterpret_cast<u64 *>(g.rows + base);
        if ((u32)tid < n_words) dst[tid] = w0;
        if ((u32)tid + (u32)nt < n_words) dst[tid + nt] = w1;
        for (u32 i = (u32)tid + 2u * (u32)nt; i < n_words; i += (u32)nt) dst[i] = src[i];
    }
    // ---- 5. the last workgroup to draw a ticket hands the counter block to the host (mapped memory, no copy call)
    // and clears the block of the next scan (no memset call).  The counters are only ever touched by device-scope
    // atomics, performed at the coherence point; every wave's counter atomics were issued before the barrier in
    // front of step 4, which waits for outstanding memory operations, so they precede the ticket.  (No
    // __threadfence(): at agent scope it writes back the XCD's whole L2, once per workgroup -- measured 1.4x on
    // the chr22 scan and 3.5x on 400 Mbp.)
    PRF_STAMP(7);
    if (ticket == (u64)gridDim.x - 1ull) {
        for (u32 i = (u32)tid; i < (u32)PRF_CNT_N; i += (u32)nt) {
            const u64 v = atomicAdd(&g.counters[i], 0ull);
            g.host_counters[i] = v;
            g.next_counters[i] = 0;
            if (i == (u32)PRF_CNT_ROWS && g.count_row) {  // a caller-owned row array carries its own length
                prf_hit_dev h;
                h.start = v & ((1ull << PRF_ROWS_TICKET_SHIFT) - 1ull);
                h.end = 0;
                h.k = 0;
                h.contig = 0;
                g.rows[g.rows_cap] = h;
            }
        }
        __threadfence_system();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&g.host_counters[PRF_CNT_N], g.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---------------------------------------------------------------------------------------------------
// ASCII -> bit-sliced planes.  One wave per tile; lane l, for bit b = 0..31, reads the 32 consecutive
// bytes of stream b*64+l (a wave reads 2 KiB contiguous per b) and spreads them over its 32 row words.
__global__ __launch_bounds__(64) void prf_pack_vertical_kernel(const uint8_t *__restrict__ asc, u32 *__restrict__ VH,
                                                               u32 *__restrict__ VL, u32 *__restrict__ VX,
                                                               unsigned char *__restrict__ any_all) {
    const u64 tile = blockIdx.x;
    const int lane = (int)threadIdx.x;
    u32 h[T], l[T], x[T];
#pragma unroll
    for (int t = 0; t < T; t++) h[t] = l[t] = x[t] = 0;
    const uint8_t *base = asc + tile * PRF_TILE + (u64)lane * T;
    for (int b = 0; b < 32; b++) {
        const uint4 *src = reinterpret_cast<const uint4 *>(base + (u64)b * (64 * T));
        const uint4 v0 = src[0], v1 = src[1];
        const u32 d[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int t = 0; t < T; t++) {
            const u32 f = (d[t >> 2] >> (8 * (t & 3))) & 0xDFu;
            const u32 ok = (f == 'A') | (f == 'C') | (f == 'G') | (f == 'T');
            h[t] |= ((f >> 2) & 1u & ok) << b;
            l[t] |= ((f >> 1) & 1u & ok) << b;
            x[t] |= (ok ^ 1u) << b;
        }
    }
    u32 any = 0, all = ~0u;
#pragma unroll
    for (int t = 0; t < T; t++) {
        any |= x[t];
        all &= x[t];
    }
    uint4 *oh = reinterpret_cast<uint4 *>(VH) + tile * RG * 64 + lane;
    uint4 *ol = reinterpret_cast<uint4 *>(VL) + tile * RG * 64 + lane;
    uint4 *ox = reinterpret_cast<uint4 *>(VX) + tile * RG * 64 + lane;
#pragma unroll
    for (int rg = 0; rg < RG; rg++) {
        oh[rg * 64] = make_uint4(h[4 * rg], h[4 * rg + 1], h[4 * rg + 2], h[4 * rg + 3]);
        ol[rg * 64] = make_uint4(l[4 * rg], l[4 * rg + 1], l[4 * rg + 2], l[4 * rg + 3]);
        ox[rg * 64] = make_uint4(x[4 * rg], x[4 * rg + 1], x[4 * rg + 2], x[4 * rg + 3]);
    }
    const bool w_any = __builtin_amdgcn_ballot_w64(any != 0) != 0;
    const bool w_all = __builtin_amdgcn_ballot_w64(all != ~0u) == 0;
    if (lane == 0) any_all[tile] = (unsigned char)((w_any ? 1 : 0) | (w_all ? 2 : 0));
}

// class: 2 = only not-ACGT, 1 = some not-ACGT in this tile or the next (whose first lanes are this tile's
// virtual lanes 64..), 0 = clean
__global__ void prf_tile_class_kernel(const unsigned char *__restrict__ any_all, unsigned char *__restrict__ cls, u64 ntiles) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ntiles) return;
    const unsigned char a = any_all[i];
    const unsigned char b = (i + 1 < ntiles) ? any_all[i + 1] : (unsigned char)3;
    cls[i] = (a & 2) ? 2 : (((a | b) & 1) ? 1 : 0);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// Work plan: which wave runs which motif sizes.  Pure host code.
bool prf_vertical_plan(u32 kmin, u32 kmax, u32 min_repeats, u32 min_span, prf_vplan *plan) {
    if (kmin < 1 || kmax < kmin || kmax > PRF_VMAX_K || min_repeats < 2) return false;
    struct Item {
        prf_vtask t;
        u32 cost;
    };
    std::vector<Item> items;
    u32 reach = 0;  // furthest row offset any task reads, relative to the first row of a block
    u32 covered_to = 0;  // group chunks cover motif sizes below this
    for (u32 k = kmin; k <= kmax; k++) {
        const long long M = prf_min_matches(k, min_repeats, min_span);
        if (M < SMALL_M) {
            Item it;
            it.t.k0 = (unsigned short)k;
            it.t.kind = (unsigned char)M;  // M >= 1 because min_repeats >= 2
            it.t.valid = 1;
            it.t.stride = 1;
            it.cost = 88;  // measured (stamps build, chr22 stand-in): 7.5-9.9 k cycles per exact task, whatever M
            items.push_back(it);
            const u32 nr = 8 + (u32)M - 1;  // rows per block; whole 16-byte slots are read
            reach = std::max<u32>(reach, (k & ~3u) + 4 * (((k & 3u) + nr + 3) / 4) - 1);
        } else if (k >= covered_to) {
            const u32 k0 = k & ~3u;
            u32 valid = 0;
            for (u32 kk = 0; kk < 8; kk++) {
                const u32 kx = k0 + kk;
                if (kx >= kmin && kx <= kmax && prf_min_matches(kx, min_repeats, min_span) >= SMALL_M) valid |= 1u << kk;
            }
            // examine every group, every 2nd or every 4th: a run of >= 8*S + 7 positions contains an aligned group
            // of 8 whose index is a multiple of S
            long long mmin = 1ll << 40;
            for (u32 kk = 0; kk < 8; kk++)
                if ((valid >> kk) & 1u) mmin = std::min(mmin, prf_min_matches(k0 + kk, min_repeats, min_span));
            const u32 stride = mmin >= 39 ? 4u : (mmin >= 23 ? 2u : 1u);
            Item it;
            it.t.k0 = (unsigned short)k0;
            it.t.kind = 0;
            it.t.valid = (unsigned char)valid;
            it.t.stride = (unsigned char)stride;
            it.cost = 8 + 27 * (4 / stride);  // measured: ~2.7 k cycles per examined block + ~0.8 k per task
            items.push_back(it);
            reach = std::max<u32>(reach, k0 + 15);
            covered_to = k0 + 8;
        }
    }
    if (items.size() > PRF_VMAX_TASKS) return false;
    // longest-processing-time-first assignment to at most 4 waves
    u32 total = 0;
    for (const Item &it : items) total += it.cost;
    const u32 nw = std::max<u32>(1, std::min<u32>({4u, (u32)items.size(), (total + 199) / 200}));
    std::vector<std::vector<Item>> bins(nw);
    std::vector<u32> load(nw, 0);
    std::vector<Item> sorted = items;
    std::stable_sort(sorted.begin(), sorted.end(), [](const Item &a, const Item &b) { return a.cost > b.cost; });
    for (const Item &it : sorted) {
        const u32 w = (u32)(std::min_element(load.begin(), load.end()) - load.begin());
        bins[w].push_back(it);
        load[w] += it.cost;
    }
    plan->n_waves = nw;
    plan->n_tasks = 0;
    for (u32 w = 0; w < nw; w++) {
        plan->wave_begin[w] = plan->n_tasks;
        for (const Item &it : bins[w]) plan->tasks[plan->n_tasks++] = it.t;
    }
    for (u32 w = nw; w <= PRF_VMAX_WAVES; w++) plan->wave_begin[w] = plan->n_tasks;
    const u32 need_nc = 64 + (24 + reach) / T;  // the last block starts at row 24; virtual lanes 64 .. 63+offset
    plan->nc = need_nc <= 66 ? 66 : (need_nc <= 72 ? 72 : 80);  // the widths the kernel is instantiated for
    plan->cof_words = (kmax + 1 + 3) & ~3u;  // <= PRF_VMAX_K + 4: the table is declared with that many entries
    plan->lds_bytes = (u32)(SMEM_HDR + (size_t)3 * RG * plan->nc * sizeof(uint4) + (size_t)2 * LW * sizeof(u64) +
                            (size_t)MAX_WAVES * REC_PER_WAVE * sizeof(u64) + (size_t)plan->cof_words * sizeof(u32));
    return true;
}

hipError_t prf_vertical_launch(hipStream_t s, const prf_vscan_args &args, int n_cus) {
    const u32 n = args.n_clean + 4u * args.n_mixed;
    if (n == 0) return hipSuccess;
    const dim3 grid(n), block(64 * args.plan.n_waves);
    const u32 lds4 = args.plan.lds_bytes - (u32)((size_t)RG * args.plan.nc * sizeof(uint4));  // the third plane overlays the window
    // the 4-per-CU build only where 4 workgroups fit a CU's 160 KB of LDS (motif sizes up to ~130) and the launch
    // has more workgroups than 3 per CU can hold at once
    const bool one_round = n <= 3u * (u32)n_cus || 4u * ((lds4 + 1023u) & ~1023u) > 160u * 1024u;
#define PRF_LAUNCH(NC)                                                                                              \
    if (one_round) hipLaunchKernelGGL((prf_vscan_kernel<NC, 3>), grid, block, args.plan.lds_bytes, s, args);        \
    else hipLaunchKernelGGL((prf_vscan_kernel<NC, 4>), grid, block, lds4, s, args)
    switch (args.plan.nc) {
        case 66: PRF_LAUNCH(66); break;
        case 72: PRF_LAUNCH(72); break;
        case 80: PRF_LAUNCH(80); break;
        default: return hipErrorInvalidValue;
    }
#undef PRF_LAUNCH
    return hipGetLastError();
}

int prf_vertical_pack(hipStream_t s, const uint8_t *asc, u64 G, prf_vplanes *vp) {
    const u64 ntiles = G / PRF_TILE;  // includes the sentinel tile
    hipError_t e;
    const size_t plane_bytes = (size_t)ntiles * RG * 64 * sizeof(uint4);
    if ((e = hipMalloc((void **)&vp->VH, plane_bytes)) != hipSuccess) return (int)e;
    if ((e = hipMalloc((void **)&vp->VL, plane_bytes)) != hipSuccess) return (int)e;
    if ((e = hipMalloc((void **)&vp->VX, plane_bytes)) != hipSuccess) return (int)e;
    if ((e = hipMalloc((void **)&vp->tile_class, 2 * ntiles)) != hipSuccess) return (int)e;
    if ((e = hipMalloc((void **)&vp->tile_list, sizeof(u32) * ntiles)) != hipSuccess) return (int)e;
    vp->ntiles_alloc = ntiles;
    unsigned char *any_all = vp->tile_class + ntiles;
    hipLaunchKernelGGL(prf_pack_vertical_kernel, dim3((u32)ntiles), dim3(64), 0, s, asc, vp->VH, vp->VL, vp->VX, any_all);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    hipLaunchKernelGGL(prf_tile_class_kernel, dim3((u32)((ntiles + 255) / 256)), dim3(256), 0, s, any_all, vp->tile_class, ntiles);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    // tile lists (host side: one byte per 65536 positions)
    std::vector<unsigned char> cls(ntiles);
    if ((e = hipMemcpyAsync(cls.data(), vp->tile_class, ntiles, hipMemcpyDeviceToHost, s)) != hipSuccess) return (int)e;
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return (int)e;
    std::vector<u32> list;
    list.reserve(ntiles);
    for (u64 t = 0; t + 1 < ntiles; t++)  // the sentinel tile is never scanned
        if (cls[t] == 0) list.push_back((u32)t);
    vp->n_clean = (u32)list.size();
    vp->clean_base = (!list.empty() && list.back() - list.front() + 1 == list.size()) ? list.front() : ~0u;
    for (u64 t = 0; t + 1 < ntiles; t++)
        if (cls[t] == 1) list.push_back((u32)t);
    vp->n_mixed = (u32)list.size() - vp->n_clean;
    if (!list.empty()) {
        if ((e = hipMemcpyAsync(vp->tile_list, list.data(), sizeof(u32) * list.size(), hipMemcpyHostToDevice, s)) != hipSuccess)
            return (int)e;
        if ((e = hipStreamSynchronize(s)) != hipSuccess) return (int)e;
    }
    return 0;
}
