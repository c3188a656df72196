// scan_vertical.hip -- the fast path: fused bit-sliced ("vertical") scan + verification kernel for gfx950,
// the row gather that follows it, the work planner and the ASCII -> bit-sliced packer.
//
// What the kernels replace: the L x n_k calls of PerfectRepeatTracker.advance()
// (reference utils/perfect_repeat_tracker.py:43-61), the per-run filter/emit step (:71-101, :108-142) and the
// final sorted() of the rows (reference perfect_repeat_finder.py:81), for any kmin..kmax <= 480 and any thresholds
// with min_repeats >= 2.
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
// One 256-thread workgroup per tile:
//  1. stage: the tile's bit-sliced planes and a window of the LINEAR H/L planes (tile - 64 .. tile + 65536 +
//     1536 positions) go to LDS.
//  2. scan: every wave runs its share of the plan's tasks (host-built, balanced by cost).  A task answers one
//     question per (stream, motif size): "may a reportable run be found from this stream?" --
//      * exact task, one motif size k <= 14 with M(k) = M < 15 (compiled per (k, M)): the whole stream in one
//        straight-line block: mismatch word per row (2 operations), sliding OR over exactly M rows; a row whose
//        M successors all match and whose predecessor does not is the start of a run of >= M.
//      * group task, 8 motif sizes k0..k0+7 with M(k) >= 15: a run of >= 15 matches contains an aligned
//        group of 8 rows that all match.  Per (group, k) the 8 rows of (H^H')|(L^L') are OR-ed with 16
//        v_bitop3_b32; motif sizes with M >= 23 / 39 examine only every 2nd / 4th group.
//     The answer is ONE 32-bit word per lane, task and motif size (bit b = stream b*64+lane); lanes with a
//     non-zero word append an 8-byte record (lane, k, word) to their wave's LDS list.  No bit loop, no atomics.
//  3. verify: all lanes expand the records; for every flagged (stream, k) the candidates are re-derived EXACTLY
//     from the linear planes (64-position looks, verify_impl.h) and each becomes a row or nothing.  A row belongs
//     to the tile that holds its first position: a run whose first examined group lies in the next tile is
//     reported by a look at the tile's end (boundary pass), and dropped by the next tile.
//  4. rows: sorted by (start, end) in LDS, written to the tile's slab.
// A second, small kernel (prf_vgather_kernel) concatenates the slabs in launch (= position) order: the row array
// leaves the device sorted by (contig, start, end), which is what the reference's sorted() returns (:81).
// Exactness argument: DESIGN.md.
#include <algorithm>
#include <cstdlib>
#include <utility>
#include <vector>

#include "prf_host.h"
#include "scan_vertical.h"
#include "verify_impl.h"

namespace {

constexpr int T = 32;      // rows (= positions) per stream
constexpr int RG = T / 4;  // row groups of 4 rows = one 16-byte slot per lane
constexpr int LIN_PRE = 1;                                    // linear window: words before the tile
constexpr int LIN_POST = 24;                                  // ... and after it
constexpr int LW = (int)PRF_TILE_WORDS + LIN_PRE + LIN_POST;  // words per plane in the LDS window
constexpr int REC_PER_WAVE = 96;                              // group-task candidate records per wave (LDS list)
constexpr int MAX_WAVES = PRF_VMAX_WAVES;
constexpr int NTH = 64 * MAX_WAVES;                           // threads per workgroup, always
constexpr int SMALL_M = 15;                                   // M(k) below this -> exact task
constexpr u32 FLAGS_PER_TASK = 128;                           // flags (lane, stream) an exact task's list holds: 256 B per task
constexpr int ROW_CAP_LDS = 256;                              // rows of a tile sorted in LDS (more: unsorted, host sorts)

template <int A, class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, A + I>{}), ...);
}
// f(integral_constant<int,i>) for i in [A, B)
template <int A, int B, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B > A) static_for_impl<A>(static_cast<F &&>(f), std::make_integer_sequence<int, B - A>{});
}

// ---- candidate records: [5:0] lane, [14:6] k, [16:15] 0 = exact task, 1/2/3 = group task examining every 1st/2nd/4th
// group, [48:17] stream word (bit b = stream b*64 + lane may hold a candidate) ----
__device__ __forceinline__ u64 make_rec(u32 lane, u32 k, u32 sc, u32 word) {
    return (u64)(lane | (k << 6) | (sc << 15)) | ((u64)word << 17);
}

// dynamic LDS: [header 192 B][vimg: 2*RG*NC uint4][lin: 2*LW u64][recs: MAX_WAVES*REC_PER_WAVE u64]
// A tile with N in reach keeps its third (not-ACGT) plane where clean tiles keep the linear window and verifies on the
// global planes.  After the scan the image is dead: the row keys (ROW_CAP_LDS u64) and motif sizes (u32) lie there.
extern __shared__ __attribute__((aligned(16))) unsigned char prf_smem[];
constexpr int SMEM_HDR = 192;

// LDS is addressed through explicit address-space pointers everywhere: a generic pointer that the compiler cannot trace back
// to prf_smem becomes a flat_load, which is slower and waits on both memory counters.
typedef u32 prf_u32x4 __attribute__((ext_vector_type(4)));  // (HIP's uint4 class cannot be copied out of an explicit address space)
typedef __attribute__((address_space(3))) const prf_u32x4 prf_lds_cu4;
typedef __attribute__((address_space(3))) prf_u32x4 prf_lds_u4;
typedef __attribute__((address_space(3))) u64 prf_lds_u64;
typedef __attribute__((address_space(3))) u32 prf_lds_u32;
typedef __attribute__((address_space(3))) const u32 prf_lds_cu32;

// Diagnostic build only (make STAMPS=1 -> libprf_stamps.so): per-wave s_memtime stamps at the phase boundaries, written
// to a debug buffer that nothing else reads.  The product build has no stamp.
#ifdef PRF_STAMPS
#define PRF_STAMP(i)                                                                                               \
    do {                                                                                                           \
        if (g.dbg && lane == 0) g.dbg[((u64)slot * MAX_WAVES + wave) * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define PRF_STAMP(i) do { } while (0)
#endif

// what the verification step needs about the tile; lives at the start of LDS (filled by thread 0 while staging)
struct TileCtx {
    u64 w0;                   // first word of the linear window
    u64 xz_lo, xz_hi;         // positions known to hold no not-ACGT symbol
    const u64 *H, *L, *X;     // linear planes in HBM
    const u64 *const *E;      // device array of the five planes of the symbols outside ACGTN, or nullptr (prf_planes::E)
    prf_hit_dev *slab;        // this tile's row slab in HBM
    u64 contig_base;          // a tile lies inside one contig
    u64 tile_base;            // first position of the tile
    u32 contig;
    u32 slab_cap;
    u32 min_repeats, min_span;
    u32 lin_off;              // byte offset of the linear window in LDS
    u32 has_lin;              // the linear window is staged (clean tiles: from the start; tiles with N in reach: after the scan)
    u32 xwin_off;             // byte offset of the not-ACGT plane's window (tiles with N in reach, after the scan), else 0
    u32 cof_off;              // byte offset of the cofactor table in LDS
    u32 hotw_off;             // byte offset of the exact tasks' stream words in LDS: [exact task][lane], then (k | M << 16) per task
    u32 n_exact;              // exact tasks of the plan: motif sizes k_exact0 .. k_exact0 + n_exact - 1, task index = k - k_exact0
    u32 k_exact0;
};
static_assert(sizeof(TileCtx) <= 128, "TileCtx must fit its LDS header slot");

__device__ __forceinline__ u32 *smem_row_cnt() { return reinterpret_cast<u32 *>(prf_smem + 160); }          // rows sent to the LDS list
__device__ __forceinline__ u32 *smem_direct_cnt() { return reinterpret_cast<u32 *>(prf_smem + 164); }       // rows written straight to the slab
// the row list (lies where the image was): 32-bit sort keys, motif sizes, ends, ROW_CAP_LDS of each
__device__ __forceinline__ prf_lds_u32 *smem_row_keys() { return (prf_lds_u32 *)(prf_smem + SMEM_HDR); }
__device__ __forceinline__ prf_lds_u32 *smem_row_ks() { return (prf_lds_u32 *)(prf_smem + SMEM_HDR + ROW_CAP_LDS * 4); }
__device__ __forceinline__ prf_lds_u64 *smem_row_ends() { return (prf_lds_u64 *)(prf_smem + SMEM_HDR + ROW_CAP_LDS * 8); }

// cof[k]: the cofactors k/p of the distinct primes p | k, one per byte, largest first (k <= 480 has at most 4
// distinct primes and k/p <= 240).  The motif seq[a:a+k] is primitive iff it has none of these periods
// (reference consists_of_perfect_repeats, utils/perfect_repeat_tracker.py:108-142, tries every divisor).
// Entries 0 .. kmax of the scan are copied to LDS per tile: a table look, not a run-time division, per candidate.
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

// One row.  to_lds: into the LDS list that is sorted at the end of the tile.  Sort key: start in the tile (16 bits), then
// length clipped to 16 bits -- exact, because of the rows that share a start at most one is longer than two motif sizes
// (two periods on a long common stretch force their gcd, Fine and Wilf).  A full list, or a list that cannot be
// used yet (a wave emptying its record list in the middle of the scan, while the image still lies there), sends the row
// straight to the slab, unsorted.
__device__ __forceinline__ void emit_row(const TileCtx &tc, bool to_lds, u64 a, u64 b, u32 k) {
    const u64 end = b + k;
    if (to_lds) {
        const u32 i = atomicAdd(smem_row_cnt(), 1u);
        if (i < (u32)ROW_CAP_LDS) {
            const u64 span = end - a;
            smem_row_keys()[i] = ((u32)(a - tc.tile_base) << 16) | (span < 65535ull ? (u32)span : 65535u);
            smem_row_ks()[i] = k;
            smem_row_ends()[i] = end;
            return;
        }
    }
    const u32 j = atomicAdd(smem_direct_cnt(), 1u);
    if (j < tc.slab_cap) {
        prf_hit_dev h;
        h.start = a - tc.contig_base;
        h.end = end - tc.contig_base;
        h.k = k;
        h.contig = tc.contig;
        tc.slab[j] = h;
    }
}

// One 64-position look: mismatch bits (1 = differs, or either side is not ACGT) of positions q .. q+63 against q+k ..,
// served from the LDS window where it covers both sides, from the global planes elsewhere.  NOT inlined: verification is
// a few looks per candidate in divergent code, and forty inlined copies of the look were 90 KB of kernel (the
// instruction cache is shared by two CUs).
__device__ __noinline__ u64 tile_mismatch64(u64 q, u32 k) {
    const TileCtx &tc = *reinterpret_cast<const TileCtx *>(prf_smem);
    prf_window_view view;
    view.lds = (prf_lds_cu64 *)(prf_smem + tc.lin_off);
    view.w0 = tc.w0;
    view.nwords = tc.has_lin ? LW : 0;  // 0: every look goes to the global planes
    view.xz_lo = tc.xz_lo;
    view.xz_hi = tc.xz_hi;
    view.x_in_lds = 0;
    view.P[0] = tc.H; view.P[1] = tc.L; view.P[2] = tc.X;
    view.E = tc.E;
    return view.mismatch64(q, k);
}

// run at motif size k, known to match up to `from`: where does it end?  (the guard gap guarantees an end)
__device__ __forceinline__ u64 run_end(u64 from, u32 k) {
    u64 b = from;
    for (;;) {
        const u64 m2 = tile_mismatch64(b, k);
        if (m2) return b + (u64)__builtin_ctzll(m2);
        b += 64;
    }
}

// Is seq[a : a+k] a whole number (>= 2) of copies of a shorter word?  (reference consists_of_perfect_repeats,
// utils/perfect_repeat_tracker.py:108-142, tries every divisor.)  A word of length k has a proper divisor period iff it has
// period k/p for some prime p | k: one period test per entry of cof[k].
__device__ __forceinline__ bool motif_is_repeat(u64 a, u32 k) {
    const TileCtx &tc = *reinterpret_cast<const TileCtx *>(prf_smem);
    for (u32 cf = ((prf_lds_cu32 *)(prf_smem + tc.cof_off))[k]; cf; cf >>= 8) {
        const u32 d = cf & 255u, need = k - d;  // period d: positions a .. a+need-1 equal the ones d later
        bool has = true;
        for (u32 off = 0; off < need; off += 64) {
            u64 mm = tile_mismatch64(a + off, d);
            const u32 left = need - off;
            if (left < 64) mm &= (1ull << left) - 1ull;
            if (mm) {
                has = false;
                break;
            }
        }
        if (has) return true;
    }
    return false;
}

// Every candidate of motif size k that the flagged stream [sp, sp+32) owns, re-derived from the linear planes.
//  sc == 0 (exact task, M = M(k) < 15): every position a in the stream that starts a maximal run of >= M matches.
//  sc >= 1 (group task, every S = 1 << (sc-1) th aligned group of 8 examined): every examined all-match group of the
//          stream that is the FIRST examined all-match group of its run; the run is dropped if it starts before the
//          tile (the previous tile reports it, see boundary_pass).
__device__ __noinline__ void verify_stream(u64 sp, u32 k, u32 sc, bool to_lds) {
    const TileCtx &tc = *reinterpret_cast<const TileCtx *>(prf_smem);
    const long long M = prf_min_matches(k, tc.min_repeats, tc.min_span);
    if (sc == 0) {
        // bit i of m = mismatch at position sp - 1 + i
        const u64 m = sp ? tile_mismatch64(sp - 1, k) : ((tile_mismatch64(0, k) << 1) | 1ull);
        u64 r = ~m;  // bit i: positions i .. i+len-1 all match
        u32 len = 1;
        while (2 * len <= (u32)M) {
            r &= r >> len;
            len *= 2;
        }
        if (len < (u32)M) r &= r >> ((u32)M - len);
        u64 st = r & (m << 1) & 0x1FFFFFFFEull;  // starts at bits 1 .. 32 = the stream's own positions
        while (st) {
            const u32 i = (u32)__builtin_ctzll(st);
            st &= st - 1;
            const u64 a = sp - 1 + i;
            const u64 after = m >> i;  // bit j = mismatch at a + j, known for j < 64 - i
            const u64 b = after ? a + (u64)__builtin_ctzll(after) : run_end(a + (64 - i), k);
            if (!motif_is_repeat(a, k)) emit_row(tc, to_lds, a, b, k);
        }
        return;
    }
    const u32 S = 1u << (sc - 1u);
    const u32 back = 8u * S;  // distance between examined groups
    for (u32 j = 0; j < 4u; j += S) {
        const u64 p = sp + 8u * j;
        const u32 look = p >= back ? back : (u32)p;   // the arrays start less than `back` before p (first tile only)
        const u64 mm = tile_mismatch64(p - look, k);  // bit i = mismatch at p - look + i
        if ((mm >> look) & 0xFFull) continue;         // the group [p, p+8) does not match throughout
        const u64 lead = mm & ((1ull << look) - 1ull);
        u64 a;
        if (lead == 0) {
            if (look == back) continue;  // the previous examined group lies in the same run: it reports
            a = p - look;                // the run starts at position 0
        } else {
            a = p - (u64)__builtin_clzll(lead << (64 - look));  // matches directly before p
        }
        if (a < tc.tile_base) continue;  // owned by the tile that holds the start
        const u64 seen = (mm >> look) >> 8;  // bit i = mismatch at p + 8 + i, known for i < 56 - look
        const u64 b = seen ? p + 8 + (u64)__builtin_ctzll(seen) : run_end(p + (64 - look), k);
        if ((long long)(b - a) < M) continue;
        if (!motif_is_repeat(a, k)) emit_row(tc, to_lds, a, b, k);
    }
}

// ---- lean verification for the common case: a candidate of a clean tile whose looks stay inside the LDS window ----
// Window positions: bit 0 of the window = 64 positions before the tile; the window holds H and L (the not-ACGT plane is known
// to be zero there).  32-bit words, v_alignbit funnel shifts.
constexpr u32 WIN_POS = (u32)LW * 64u;  // positions in the window

__device__ __forceinline__ u32 look32(prf_lds_cu32 *plane, u32 q) {
    const u32 w = q >> 5;
    return __builtin_amdgcn_alignbit(plane[w + 1], plane[w], q & 31u);
}
__device__ __forceinline__ u64 look64(prf_lds_cu32 *plane, u32 q) {
    const u32 w = q >> 5, sft = q & 31u;
    const u32 w0 = plane[w], w1 = plane[w + 1], w2 = plane[w + 2];
    return (u64)__builtin_amdgcn_alignbit(w1, w0, sft) | ((u64)__builtin_amdgcn_alignbit(w2, w1, sft) << 32);
}
struct WinCtx {
    prf_lds_cu32 *h, *l, *x, *cof;  // x: the not-ACGT plane's window (tiles with N in reach), nullptr for clean tiles
    u64 win0;  // global position of window bit 0
    u32 min_repeats, min_span;
};

// mismatch bits of window positions q .. q+31 / q+63 against q+k ..; the caller guarantees q + k + 96 <= WIN_POS
__device__ __forceinline__ u32 win_mismatch32(const WinCtx &wc, u32 q, u32 k) {
    u32 r = (look32(wc.h, q) ^ look32(wc.h, q + k)) | (look32(wc.l, q) ^ look32(wc.l, q + k));
    if (wc.x) r |= look32(wc.x, q) | look32(wc.x, q + k);
    return r;
}
__device__ __forceinline__ u64 win_mismatch64(const WinCtx &wc, u32 q, u32 k) {
    u64 r = (look64(wc.h, q) ^ look64(wc.h, q + k)) | (look64(wc.l, q) ^ look64(wc.l, q + k));
    if (wc.x) r |= look64(wc.x, q) | look64(wc.x, q + k);
    return r;
}

__device__ __forceinline__ u32 min_matches32(u32 k, u32 min_repeats, u32 min_span) {
    const u32 a = (min_repeats - 1u) * k, b = min_span > k ? min_span - k : 0u;
    return a > b ? a : b;
}

// end of the run at motif size k that matches up to window position `from` (global position returned); leaves the
// window -> the general routine
__device__ __forceinline__ u64 win_run_end(const WinCtx &wc, u32 from, u32 k) {
    for (;;) {
        if (from + k + 96u > WIN_POS) return run_end(wc.win0 + from, k);
        const u64 m2 = win_mismatch64(wc, from, k);
        if (m2) return wc.win0 + from + (u64)__builtin_ctzll(m2);
        from += 64u;
    }
}

// motif [a, a+k) at window position a: a power of a shorter word?  (see motif_is_repeat)
__device__ __forceinline__ bool win_motif_is_repeat(const WinCtx &wc, u32 a, u32 k) {
    if (a + 2u * k + 96u > WIN_POS) return motif_is_repeat(wc.win0 + a, k);
    for (u32 cf = wc.cof[k]; cf; cf >>= 8) {
        const u32 d = cf & 255u, need = k - d;
        bool has = true;
        for (u32 off = 0; off < need; off += 32) {
            u32 mm = win_mismatch32(wc, a + off, d);
            const u32 left = need - off;
            if (left < 32) mm &= (1u << left) - 1u;
            if (mm) {
                has = false;
                break;
            }
        }
        if (has) return true;
    }
    return false;
}

// funnel shift right of the 128-bit value hi:lo by s in [1, 63]
__device__ __forceinline__ u64 shr128(u64 lo, u64 hi, u32 sft) { return (lo >> sft) | (hi << (64u - sft)); }

// cofactors k/p of the distinct primes p | k for k <= 15, two 4-bit fields per byte (see CofTable): no table look for the exact tasks
__device__ __forceinline__ u32 small_cof(u32 k) {
    const u64 t = k < 8u ? 0x0123010201010000ull : 0x0027014601250304ull;
    return (u32)(t >> (8u * (k & 7u))) & 255u;
}

// One (stream, exact task) flag of a clean tile: stream (lane rl, bit `bit`), motif size k.  128 positions of both planes
// from the position in front of the stream are read; the mismatch word, the run starts, the run ends and the periods of the
// primitive-motif test are funnel shifts of those registers.
__device__ __forceinline__ void win_verify_flag(const TileCtx &tc, const WinCtx &wc, u32 rl, u32 bit, u32 k) {
    const u32 q = 64u + (bit * 64u + rl) * T;  // window position of the stream's first position
    const u32 w = (q - 1u) >> 5, sft = (q - 1u) & 31u;
    const u32 a0 = wc.h[w], a1 = wc.h[w + 1], a2 = wc.h[w + 2], a3 = wc.h[w + 3], a4 = wc.h[w + 4];
    const u32 b0 = wc.l[w], b1 = wc.l[w + 1], b2 = wc.l[w + 2], b3 = wc.l[w + 3], b4 = wc.l[w + 4];
    // bit i = window position q - 1 + i
    const u64 hlo = (u64)__builtin_amdgcn_alignbit(a1, a0, sft) | ((u64)__builtin_amdgcn_alignbit(a2, a1, sft) << 32);
    const u64 hhi = (u64)__builtin_amdgcn_alignbit(a3, a2, sft) | ((u64)__builtin_amdgcn_alignbit(a4, a3, sft) << 32);
    const u64 llo = (u64)__builtin_amdgcn_alignbit(b1, b0, sft) | ((u64)__builtin_amdgcn_alignbit(b2, b1, sft) << 32);
    const u64 lhi = (u64)__builtin_amdgcn_alignbit(b3, b2, sft) | ((u64)__builtin_amdgcn_alignbit(b4, b3, sft) << 32);
    u64 xlo = 0, xhi = 0;
    if (wc.x) {
        const u32 c0 = wc.x[w], c1 = wc.x[w + 1], c2 = wc.x[w + 2], c3 = wc.x[w + 3], c4 = wc.x[w + 4];
        xlo = (u64)__builtin_amdgcn_alignbit(c1, c0, sft) | ((u64)__builtin_amdgcn_alignbit(c2, c1, sft) << 32);
        xhi = (u64)__builtin_amdgcn_alignbit(c3, c2, sft) | ((u64)__builtin_amdgcn_alignbit(c4, c3, sft) << 32);
    }
    const u32 M = min_matches32(k, wc.min_repeats, wc.min_span);
    // bit i = mismatch at window position q - 1 + i
    const u64 m = (hlo ^ shr128(hlo, hhi, k)) | (llo ^ shr128(llo, lhi, k)) | xlo | shr128(xlo, xhi, k);
    u64 r = ~m;  // -> bit i: positions i .. i+M-1 all match (M <= 14: three doublings and a rest)
    if (M >= 2) r &= r >> 1;
    if (M >= 4) r &= r >> 2;
    if (M >= 8) r &= r >> 4;
    {
        const u32 len = M >= 8 ? 8u : (M >= 4 ? 4u : (M >= 2 ? 2u : 1u));
        r &= r >> (M - len);
    }
    u64 st = r & (m << 1) & 0x1FFFFFFFEull;  // starts at bits 1 .. 32 = the stream's own positions
    const u32 cof_k = small_cof(k);
    while (st) {
        const u32 i = (u32)__builtin_ctzll(st);
        st &= st - 1;
        // primitive motif: no period k/p for a prime p | k (k - d <= 13 positions from the start on)
        bool rep = false;
        for (u32 cf = cof_k; cf && !rep; cf >>= 4) {
            const u32 d = cf & 15u;
            const u64 md = (hlo ^ shr128(hlo, hhi, d)) | (llo ^ shr128(llo, lhi, d));  // (no N inside a run of >= M >= k matches)
            rep = ((md >> i) & ((1ull << (k - d)) - 1ull)) == 0;
        }
        if (rep) continue;
        const u32 a = q - 1u + i;
        const u64 after = m >> i;  // bit j = mismatch at a + j, known for j < 64 - i
        const u64 b = after ? wc.win0 + a + (u64)__builtin_ctzll(after) : win_run_end(wc, a + (64u - i), k);
        emit_row(tc, true, wc.win0 + a, b, k);
    }
}

// group-task record, one flagged stream at window position q, every S-th aligned group of 8 examined: the examined
// all-match groups that are the first of their run, if the run starts inside the tile.
// The cheap part (which of the stream's groups qualify) is a loop of its own; the expensive part (run end, length,
// primitive motif, row) then runs once per qualifying group -- almost always once per stream -- instead of once per group
// index at which ANY lane of the wave has something.
__device__ __forceinline__ void win_verify_group(const TileCtx &tc, const WinCtx &wc, u32 q, u32 k, u32 S) {
    const u32 M = min_matches32(k, wc.min_repeats, wc.min_span);
    const u32 back = 8u * S;
    const u64 m = win_mismatch64(wc, q - 32u, k);  // bit i = mismatch at window position q - 32 + i
    const u32 cof_k = wc.cof[k];
    u32 leaders = 0;  // bit j: group j of the stream is all-match, the first examined one of its run, and the run starts in the tile
    u32 nbs = 0;      // 5 bits per group: matches directly before it
    for (u32 j = 0; j < 4u; j += S) {
        const u32 gb = 32u + 8u * j;  // bit of the group's first position
        const u64 lead = m << (64u - gb);  // bit 63 = the position directly before the group
        const u32 nb = lead ? (u32)__builtin_clzll(lead) : 64u;  // matches directly before it (>= 32 seen)
        const bool ok = ((m >> gb) & 0xFFull) == 0 && nb < back && q - 32u + gb - nb >= 64u;
        leaders |= (ok ? 1u : 0u) << j;
        nbs |= (nb & 31u) << (5u * j);
    }
    while (leaders) {
        const u32 j = (u32)__builtin_ctz(leaders);
        leaders &= leaders - 1;
        const u32 gb = 32u + 8u * j, nb = (nbs >> (5u * j)) & 31u;
        const u32 a = q - 32u + gb - nb;
        // One batch of looks, issued together (one LDS round trip): the first 32 positions of the period test of up to three
        // cofactors, and the 64 positions behind the first look for the run's end.  Primitive motif first: most group
        // candidates are echoes of a short motif.
        if (a + 2u * k + 96u > WIN_POS) {
            if (motif_is_repeat(wc.win0 + a, k)) continue;
        } else {
            const u32 d1 = cof_k & 255u, d2 = (cof_k >> 8) & 255u, d3 = (cof_k >> 16) & 255u;
            const u32 mm1 = win_mismatch32(wc, a, d1 ? d1 : 1u);
            const u32 mm2 = win_mismatch32(wc, a, d2 ? d2 : 1u);
            const u32 mm3 = win_mismatch32(wc, a, d3 ? d3 : 1u);
            bool rep = false;
            for (u32 ci = 0; ci < 4u && !rep; ci++) {
                const u32 d = (cof_k >> (8u * ci)) & 255u;
                if (d == 0) break;
                const u32 need = k - d;
                u32 mm = ci == 0 ? mm1 : (ci == 1 ? mm2 : (ci == 2 ? mm3 : win_mismatch32(wc, a, d)));
                if (need < 32) mm &= (1u << need) - 1u;
                rep = mm == 0;
                for (u32 off = 32; off < need && rep; off += 32) {
                    u32 m2 = win_mismatch32(wc, a + off, d);
                    const u32 left = need - off;
                    if (left < 32) m2 &= (1u << left) - 1u;
                    rep = m2 == 0;
                }
            }
            if (rep) continue;
        }
        const u64 seen = gb + 8u < 64u ? m >> (gb + 8u) : 0ull;  // bit i = mismatch at group end + i
        u64 b;
        if (seen) {
            b = wc.win0 + (q - 32u + gb + 8u) + (u64)__builtin_ctzll(seen);
        } else {
            const u64 m2 = win_mismatch64(wc, q + 32u, k);  // (q + 32 + k + 96 <= WIN_POS for every stream of the tile)
            b = m2 ? wc.win0 + (q + 32u) + (u64)__builtin_ctzll(m2) : win_run_end(wc, q + 96u, k);
        }
        if (b - (wc.win0 + a) < (u64)M) continue;
        emit_row(tc, true, wc.win0 + a, b, k);
    }
}

// Boundary pass.  A group task's run is found at the FIRST examined all-match group it contains.  For a run that starts
// in the last 8S-1 positions of this tile that group lies in the next tile, whose workgroup drops the run because it does
// not start there; this tile reports it: per motif size one look at the 32 positions in front of the
// tile's end.  c = matches directly in front of the end: 1 <= c < 8S <=> such a run exists and starts at end - c.
__device__ __forceinline__ void boundary_item(const TileCtx &tc, const WinCtx &wc, bool fast, u32 k, u32 S) {
    const u64 tile_end = tc.tile_base + PRF_TILE;
    const u32 back = 8u * S;
    const u64 mm = fast ? win_mismatch64(wc, 64u + PRF_TILE - 32u, k) : tile_mismatch64(tile_end - 32, k);
    const u32 lo = (u32)mm;  // bit i = mismatch at tile_end - 32 + i
    const u32 c = lo ? (u32)__builtin_clz(lo) : 32u;
    if (c == 0 || c >= back) return;
    const u64 a = tile_end - c;
    const u64 hi = mm >> 32;  // bit i = mismatch at tile_end + i
    const u64 b = hi ? tile_end + (u64)__builtin_ctzll(hi) : (fast ? win_run_end(wc, 64u + PRF_TILE + 32u, k) : run_end(tile_end + 32, k));
    if (b - a < (u64)min_matches32(k, tc.min_repeats, tc.min_span)) return;
    if (!(fast ? win_motif_is_repeat(wc, 64u + PRF_TILE - c, k) : motif_is_repeat(a, k))) emit_row(tc, true, a, b, k);
}

__device__ __forceinline__ prf_lds_u32 *smem_rec_cnt() { return (prf_lds_u32 *)(prf_smem + 128); }   // [MAX_WAVES]: group-task records

// A wave emptying its own full list in the middle of the scan: rare, and called from inside the tasks, so not inlined.
// General routine only; its rows go straight to the slab (the LDS row list lies where the image still is).
__device__ __noinline__ void flush_records(prf_lds_cu64 *recs, int wave, u32 n, u32 lane) {
    const u64 tile_base = reinterpret_cast<const TileCtx *>(prf_smem)->tile_base;
    for (u32 idx = lane; idx < n; idx += 64u) {
        const u64 rec = recs[(u32)wave * REC_PER_WAVE + idx];
        const u32 rl = (u32)rec & 63u, k = ((u32)rec >> 6) & 511u, sc = ((u32)rec >> 15) & 3u;
        u32 word = (u32)(rec >> 17);
        while (word) {
            const u32 bit = (u32)__builtin_ctz(word);
            word &= word - 1;
            verify_stream(tile_base + (u64)(bit * 64u + rl) * T, k, sc, false);
        }
    }
}

// Candidates -> rows, all waves together at the end of the tile.
//  * exact tasks left ballot-compacted lists of (stream, task) flags in LDS: one index space, dealt to the threads from
//    thread 0 up;
//  * group-task records are taken by the upper two waves, alternately; the boundary items by the lower half, from its last
//    thread down.
// Returns the number of (stream, exact task) flags this thread looked at (statistics).
__device__ __forceinline__ u32 verify_all(prf_lds_cu64 *recs, prf_lds_cu32 *bitems, u32 n_bitems, u32 tid, u64 *dbg) {
#ifdef PRF_STAMPS
#define PRF_VSTAMP(i) do { if (dbg && (tid & 63u) == 0) dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PRF_VSTAMP(i) do { } while (0)
#endif
    const TileCtx &tc = *reinterpret_cast<const TileCtx *>(prf_smem);
    const bool fast = tc.has_lin != 0;
    WinCtx wc;
    wc.h = (prf_lds_cu32 *)(prf_smem + tc.lin_off);
    wc.l = wc.h + 2 * LW;
    wc.x = tc.xwin_off ? (prf_lds_cu32 *)(prf_smem + tc.xwin_off) : nullptr;
    wc.cof = (prf_lds_cu32 *)(prf_smem + tc.cof_off);
    wc.win0 = tc.tile_base - 64;
    wc.min_repeats = tc.min_repeats;
    wc.min_span = tc.min_span;
    // ---- exact tasks: every task left a ballot-compacted list of its flags (lane, stream bit) in LDS (Emit::push_flags).
    // The flags of all tasks are one index space, dealt to the threads one by one: a wave runs the body about once, not as
    // often as its unluckiest lane has flags.
    u32 n_flags = 0;
    if (tc.n_exact) {
        constexpr u32 MAX_EXACT = SMALL_M - 1;  // motif sizes 1 .. 14 at most
        typedef __attribute__((address_space(3))) const unsigned short prf_lds_cu16;
        prf_lds_cu16 *lists = (prf_lds_cu16 *)(prf_smem + tc.hotw_off);
        prf_lds_cu32 *counts = (prf_lds_cu32 *)(prf_smem + tc.hotw_off) + tc.n_exact * 64u;
        u32 pre[MAX_EXACT + 1];
        pre[0] = 0;
        static_for<0, (int)MAX_EXACT>([&](auto ec) {  // (words past the last task are other data: masked)
            constexpr u32 e = (u32)decltype(ec)::value;
            const u32 c = counts[e];
            pre[e + 1] = pre[e] + (e < tc.n_exact ? c : 0u);
        });
        const u32 total = pre[MAX_EXACT];
        if (tid == 0) n_flags = total;
        for (u32 idx = tid; idx < total; idx += (u32)NTH) {
            u32 e = 0;
            static_for<1, (int)MAX_EXACT>([&](auto ec) { e += idx >= pre[decltype(ec)::value] ? 1u : 0u; });
            u32 first = 0;
            static_for<1, (int)MAX_EXACT>([&](auto ec) {
                constexpr u32 i = (u32)decltype(ec)::value;
                first = e >= i ? pre[i] : first;
            });
            const u32 f = lists[e * FLAGS_PER_TASK + (idx - first)], frl = f & 63u, fbit = (f >> 6) & 31u, k = tc.k_exact0 + e;
            // (a clean tile's first stream looks at positions in front of the tile, where N is possible and the window
            // has no not-ACGT plane: general routine)
            if (fast && ((frl | fbit) || wc.x)) win_verify_flag(tc, wc, frl, fbit, k);
            else verify_stream(tc.tile_base + (u64)(fbit * 64u + frl) * T, k, 0u, true);
        }
    }
    PRF_VSTAMP(14);
    // ---- group-task records: the upper half of the workgroup, alternating between its two waves (a wave's pass costs the
    // same with 1 or 64 records; the flags keep the lower waves busy meanwhile)
    if (tid >= (u32)NTH / 2u) {
        prf_lds_u32 *cg = smem_rec_cnt();
        const u32 c0 = cg[0], c1 = c0 + cg[1], c2 = c1 + cg[2], n = c2 + cg[3];
        const u32 up = (u32)NTH - 1u - tid;  // 0 .. 127: thread 255, 254, ...
        for (u32 idx = 2u * (up & 63u) + (up >> 6); idx < n; idx += (u32)NTH / 2u) {
            const u32 slot_idx = idx < c0 ? idx : (idx < c1 ? REC_PER_WAVE + (idx - c0) : (idx < c2 ? 2 * REC_PER_WAVE + (idx - c1) : 3 * REC_PER_WAVE + (idx - c2)));
            const u64 rec = recs[slot_idx];
            const u32 rl = (u32)rec & 63u, k = ((u32)rec >> 6) & 511u, sc = ((u32)rec >> 15) & 3u;
            u32 word = (u32)(rec >> 17);
            while (word) {
                const u32 bit = (u32)__builtin_ctz(word);
                word &= word - 1;
                const u32 sq = (bit * 64u + rl) * T;
                if (fast && (sq >= 32u || wc.x)) win_verify_group(tc, wc, 64u + sq, k, 1u << (sc - 1u));
                else verify_stream(tc.tile_base + sq, k, sc, true);
            }
        }
    } else {
        // ---- boundary items: the lower half, from its last thread down (the flags fill it from the first thread up)
        for (u32 idx = (u32)NTH / 2u - 1u - tid; idx < n_bitems; idx += (u32)NTH / 2u) {
            const u32 it = bitems[idx];
            boundary_item(tc, wc, fast, it & 0xFFFFu, it >> 16);
        }
    }
    return n_flags;
}

// Group tasks: one 32-bit word per lane (bit b = stream b*64 + lane is flagged for motif size k) -> records of the lanes
// with a non-zero word.  Every lane of the wave calls this together.
struct Emit {
    prf_lds_u64 *recs;       // this wave's list in LDS, REC_PER_WAVE records
    prf_lds_cu64 *all_recs;  // all lists
    int wave;
    int lane;
    u32 cnt;                 // records in it (wave-uniform)
    u32 flushed;             // records verified in early flushes (wave-uniform)

    // Exact tasks: the lanes' words of ONE task -> the task's list of flags (lane | stream bit << 6), ballot-compacted; the
    // count goes behind the lists.  A task with more flags than its list holds (a tile of long runs) verifies the surplus on
    // the spot with the general routine (rows straight to the slab, like a flushed record list).
    __device__ __forceinline__ void push_flags(u32 word, u32 e, u32 k, prf_lds_u32 *hotw, u32 n_exact) {
        typedef __attribute__((address_space(3))) unsigned short prf_lds_u16;
        prf_lds_u16 *list = (prf_lds_u16 *)hotw + e * FLAGS_PER_TASK;
        u32 n = 0;  // wave-uniform
        for (;;) {
            const u64 bal = __builtin_amdgcn_ballot_w64(word != 0);
            if (bal == 0) break;
            if (word) {
                const u32 bit = (u32)__builtin_ctz(word);
                word &= word - 1;
                const u32 at = n + __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0));
                if (at < FLAGS_PER_TASK) list[at] = (unsigned short)((u32)lane | (bit << 6));
                else verify_stream(reinterpret_cast<const TileCtx *>(prf_smem)->tile_base + (u64)(bit * 64u + (u32)lane) * T, k, 0u, false);
            }
            n += (u32)__builtin_popcountll(bal);
        }
        if (lane == 0) hotw[n_exact * 64u + e] = n < FLAGS_PER_TASK ? n : FLAGS_PER_TASK;
    }

    __device__ __forceinline__ void push_word(u32 word, u32 k, u32 sc) {
        const u64 bal = __builtin_amdgcn_ballot_w64(word != 0);
        if (bal == 0) return;
        const u32 n = (u32)__builtin_popcountll(bal);
        if (cnt + n > (u32)REC_PER_WAVE) {  // wave-uniform: list full -> this wave verifies it now
            flush_records(all_recs, wave, cnt, (u32)lane);
            flushed += cnt;
            cnt = 0;
        }
        if (word) {
            const u32 idx = cnt + __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0));
            recs[idx] = make_rec((u32)lane, k, sc, word);
        }
        cnt += n;
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
// a | b | c
__device__ __forceinline__ u32 or3(u32 a, u32 b, u32 c) { return bitop3<(TA | TB | TC) & 0xFF>(a, b, c); }

__device__ __forceinline__ void unpack4(u32 *dst, const prf_u32x4 v) {
    dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
}

// LDS image addressing.  The image is [plane][row group][virtual lane] of 16-byte slots, NC virtual lanes wide
// (compile-time, so plane and row-group strides are instruction immediates).  Row group gg of a lane's
// *extended* stream (gg >= 8: the stream continues in the next virtual lane) is slot (gg & 7) * NC + (gg >> 3)
// from the lane's own slot.
template <int NC, class P>
__device__ __forceinline__ P *slot_of(P *lane_base, int gg) {
    return lane_base + ((gg & 7) * NC + (gg >> 3));
}

// Slot g (compile-time) after a run-time first slot gg0 whose address `first` = slot_of(lane_base, gg0) and
// a = gg0 & 7 are computed once per block: the stream wraps into the next virtual lane at most once within a block.
template <int NC, int G>
__device__ __forceinline__ prf_lds_cu4 *slot_after(prf_lds_cu4 *first, int a) {
    return first + G * NC + (a + G >= 8 ? 1 - 8 * NC : 0);
}

// ---- group task: motif sizes k0 .. k0+7 (those in `valid`), the 8-row blocks 0 .. 3 of the stream ----
// S1: every block is examined (stride 1) and a group counts only if the group before it was not all-match; otherwise
// (stride 2 / 4) every examined all-match group counts.  The per-size words are OR-ed over the blocks and leave as records
// at the end of the task.
template <bool HASX, int NC, bool S1>
__device__ __forceinline__ void group_task(prf_lds_cu4 *vimg, prf_lds_cu4 *ximg, int lane, u32 k0, u32 valid, u32 stride, Emit &em) {
    constexpr int PS = RG * NC;  // slots per plane
    prf_lds_cu4 *lane_base = vimg + lane;
    prf_lds_cu4 *xlane_base = ximg + lane;
    u32 prev[8], acc[8];
    static_for<0, 8>([&](auto ic) {
        prev[decltype(ic)::value] = ~0u;  // first group of a stream: counts, verification decides
        acc[decltype(ic)::value] = 0u;
    });
#pragma unroll 1
    for (int tb = 0; tb < 4; tb += (int)stride) {
        u32 a[3][8];   // rows 8tb .. 8tb+7
        u32 w[3][16];  // rows 8tb+k0 .. 8tb+k0+15 (k0 % 4 == 0: whole 16-byte slots)
        const int g0 = 2 * tb + (int)(k0 >> 2);
        const int wa = g0 & 7;
        {
            prf_lds_cu4 *pa = lane_base + 2 * tb * NC;
            prf_lds_cu4 *pw0 = slot_of<NC>(lane_base, g0);
            static_for<0, 2>([&](auto pc) {
                constexpr int p = decltype(pc)::value;
                unpack4(&a[p][0], pa[p * PS]);
                unpack4(&a[p][4], pa[p * PS + NC]);
            });
            static_for<0, 4>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                prf_lds_cu4 *pw = slot_after<NC, g>(pw0, wa);
                static_for<0, 2>([&](auto pc) {
                    constexpr int p = decltype(pc)::value;
                    unpack4(&w[p][4 * g], pw[p * PS]);
                });
            });
        }
        if constexpr (HASX) {
            prf_lds_cu4 *pa = xlane_base + 2 * tb * NC;
            prf_lds_cu4 *pw0 = slot_of<NC>(xlane_base, g0);
            unpack4(&a[2][0], pa[0]);
            unpack4(&a[2][4], pa[NC]);
            static_for<0, 4>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                unpack4(&w[2][4 * g], slot_after<NC, g>(pw0, wa)[0]);
            });
        }
        // All 8 motif sizes are computed in one straight-line block so that their 8 independent OR chains
        // interleave (a chain alone is 16 dependent operations); sizes outside `valid` are dropped when records are made.
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
                    o = or3(o, a[2][i], w[2][kk + i]);
                });
            }
            if constexpr (S1) {
                acc[kk] = bitop3<(TA | (~TB & TC)) & 0xFF>(acc[kk], o, prev[kk]);  // acc | (~o & prev)
                prev[kk] = o;
            } else {
                acc[kk] |= ~o;
            }
        });
    }
    const u32 sc = stride == 1 ? 1u : (stride == 2 ? 2u : 3u);
    static_for<0, 8>([&](auto kc) {
        constexpr int kk = decltype(kc)::value;
        if ((valid >> kk) & 1u) em.push_word(acc[kk], k0 + (u32)kk, sc);  // wave-uniform condition
    });
}

// candidate word of row t: rows t .. t+M-1 all match and row t-1 does not.
// m is indexed by row+1 (m[0] = the row before the stream), o3[t] = OR of rows t..t+2.
template <int M, int t, int LM, int LO>
__device__ __forceinline__ u32 start_word(const u32 (&m)[LM], const u32 (&o3)[LO]) {
    const u32 before = m[t];
    if constexpr (M == 1) return ~m[t + 1] & before;
    else if constexpr (M == 2) return nor_and(m[t + 1], m[t + 2], before);
    else if constexpr (M == 3) return ~o3[t] & before;
    else if constexpr (M <= 6) return nor_and(o3[t], o3[t + M - 3], before);
    else if constexpr (M <= 9) return ~or3(o3[t], o3[t + 3], o3[t + M - 3]) & before;
    else if constexpr (M <= 12) return nor_and(or3(o3[t], o3[t + 3], o3[t + 6]), o3[t + M - 3], before);
    else return ~or3(or3(o3[t], o3[t + 3], o3[t + 6]), o3[t + 9], o3[t + M - 3]) & before;
}

// ---- exact task: motif size K whose minimum run length is M < 15; the whole stream in one straight-line block ----
// Returns the lane's word: bit b set = stream (lane, b) holds a row t in 0..31 that starts a run of >= M matches.
// Rows 0 .. 31+M-1+K of the extended stream are read ONCE (row i+K is the partner of row i, both in registers).
// Not inlined: one compact function per (K, M), called by the one wave that runs the task.
template <int K, int M, int NC>
__device__ __attribute__((noinline)) u32 exact_stream(prf_lds_cu4 *vimg, prf_lds_cu4 *ximg, int lane, bool hasx) {
    constexpr int PS = RG * NC;
    constexpr int NM = T + M - 1;            // mismatch words of rows 0 .. NM-1
    constexpr int NG = (NM + K + 3) / 4;     // 16-byte slots of rows read
    static_assert(4 * NG <= 2 * T, "an exact task reads its own lane and the next one");
    prf_lds_cu4 *lane_base = vimg + lane;
    u32 r0[4 * NG], r1[4 * NG];
    static_for<0, NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        prf_lds_cu4 *ps = slot_of<NC>(lane_base, g);
        const prf_u32x4 v0 = ps[0], v1 = ps[PS];
        r0[4 * g] = v0.x; r0[4 * g + 1] = v0.y; r0[4 * g + 2] = v0.z; r0[4 * g + 3] = v0.w;
        r1[4 * g] = v1.x; r1[4 * g + 1] = v1.y; r1[4 * g + 2] = v1.z; r1[4 * g + 3] = v1.w;
    });
    // Row -1 of stream (lane, b) is row T-1 of stream (lane-1, b); for lane 0 it is row T-1 of stream (63, b-1):
    // lane 63's word one bit up, with bit 0 (the previous tile's last stream) unknown -> "mismatch", verification decides.
    const int pl = (lane + 63) & 63;
    prf_lds_cu4 *pp = vimg + pl + (RG - 1) * NC;
    u32 p0 = pp[0].w, p1 = pp[PS].w;
    if (lane == 0) {
        p0 <<= 1;
        p1 <<= 1;
    }
    u32 m[NM + 1];
    m[0] = or_xor(p0 ^ r0[K - 1], p1, r1[K - 1]);
    static_for<0, NM>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        m[i + 1] = or_xor(r0[i] ^ r0[i + K], r1[i], r1[i + K]);
    });
    if (hasx) {  // wave-uniform: tile with not-ACGT positions in reach
        prf_lds_cu4 *xlane_base = ximg + lane;
        u32 rx[4 * NG];
        static_for<0, NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            const prf_u32x4 v = slot_of<NC>(xlane_base, g)[0];
            rx[4 * g] = v.x; rx[4 * g + 1] = v.y; rx[4 * g + 2] = v.z; rx[4 * g + 3] = v.w;
        });
        u32 px = (ximg + pl + (RG - 1) * NC)[0].w;
        if (lane == 0) px <<= 1;
        m[0] = or3(m[0], px, rx[K - 1]);
        static_for<0, NM>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            m[i + 1] = or3(m[i + 1], rx[i], rx[i + K]);
        });
    }
    if (lane == 0) m[0] |= 1u;
    u32 o3[NM];
    if constexpr (M >= 3) {
        static_for<0, NM - 2>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            o3[i] = or3(m[i + 1], m[i + 2], m[i + 3]);
        });
    }
    u32 hot = 0;
    static_for<0, T / 2>([&](auto tc) {
        constexpr int t = 2 * decltype(tc)::value;
        hot = or3(hot, start_word<M, t>(m, o3), start_word<M, t + 1>(m, o3));
    });
    return hot;
}

// The (K, M) variants, K <= M < SMALL_M, numbered densely in (K, M) order; the dispatch is a binary search over that number
// (7 wave-uniform branches; a chain of `if (k == K)` tests cost a task about thirty taken branches).
constexpr int exact_variants() { return (SMALL_M - 1) * SMALL_M / 2; }
constexpr int exact_variant_of(int K, int M) { return (K - 1) * (2 * SMALL_M - K) / 2 + (M - K); }
constexpr int exact_variant_k(int v) {
    int K = 1;
    while (exact_variant_of(K + 1, K + 1) <= v) K++;
    return K;
}
template <int LO, int HI, int NC>
__device__ __forceinline__ u32 exact_dispatch(prf_lds_cu4 *vimg, prf_lds_cu4 *ximg, int lane, bool hasx, u32 v) {
    if constexpr (LO == HI) {
        constexpr int K = exact_variant_k(LO), M = K + (LO - exact_variant_of(K, K));
        static_assert(M >= K && M < SMALL_M && exact_variant_of(K, M) == LO, "variant numbering");
        return exact_stream<K, M, NC>(vimg, ximg, lane, hasx);
    } else {
        constexpr int MID = (LO + HI) / 2;
        if (v <= (u32)MID) return exact_dispatch<LO, MID, NC>(vimg, ximg, lane, hasx, v);  // wave-uniform
        return exact_dispatch<MID + 1, HI, NC>(vimg, ximg, lane, hasx, v);
    }
}

template <int NC>
__device__ __forceinline__ u32 exact_any(prf_lds_cu4 *vimg, prf_lds_cu4 *ximg, int lane, bool hasx, u32 k, u32 M) {
    // (min_repeats - 1) * k <= M < SMALL_M and min_repeats >= 2: k <= M
    const u32 v = (k - 1u) * (2u * (u32)SMALL_M - k) / 2u + (M - k);
    return exact_dispatch<0, exact_variants() - 1, NC>(vimg, ximg, lane, hasx, v);
}

template <bool HASX, int NC>
__device__ __forceinline__ void run_tasks(prf_lds_cu4 *vimg, prf_lds_cu4 *ximg, prf_lds_u32 *hotw, const prf_vplan &plan, int wave, int lane,
                                          Emit &em, u64 *dbg) {
    const u32 t_end = plan.wave_begin[wave + 1];
#ifdef PRF_STAMPS
    u64 t_call = 0;
#endif
    for (u32 ti = plan.wave_begin[wave]; ti < t_end; ti++) {
        const prf_vtask task = plan.tasks[ti];
#ifdef PRF_STAMPS
        if (dbg && lane == 0 && ti - plan.wave_begin[wave] < 8u) dbg[8 + (ti - plan.wave_begin[wave])] = __builtin_amdgcn_s_memtime();
#endif
        if (task.kind == 0) {
            if (task.stride == 1) group_task<HASX, NC, true>(vimg, ximg, lane, task.k0, task.valid, 1u, em);
            else group_task<HASX, NC, false>(vimg, ximg, lane, task.k0, task.valid, task.stride, em);
        } else {
#ifdef PRF_STAMPS
            const u64 tc0 = __builtin_amdgcn_s_memtime();
            const u32 word = exact_any<NC>(vimg, ximg, lane, HASX, task.k0, task.kind);
            asm volatile("" ::"v"(word));
            t_call += __builtin_amdgcn_s_memtime() - tc0;
            em.push_flags(word, task.item0, task.k0, hotw, plan.n_exact);
#else
            em.push_flags(exact_any<NC>(vimg, ximg, lane, HASX, task.k0, task.kind), task.item0, task.k0, hotw, plan.n_exact);
#endif
        }
    }
#ifdef PRF_STAMPS
    if (dbg && lane == 0) dbg[15] = t_call;
#endif
}

__device__ __forceinline__ void set_prio(u32 p) {  // (s_setprio takes an immediate; p is wave-uniform)
    if (p == 0u) __builtin_amdgcn_s_setprio(0);
    else if (p == 1u) __builtin_amdgcn_s_setprio(1);
    else if (p == 2u) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
}

// Grid: persistent workgroups, 4 per CU (or one per entry of the launch list if that is fewer): workgroup b takes the launch
// slots b, b + gridDim, ...  (tiles in position order; slabs and counts are indexed by slot, so the order of execution does
// not show in the output).  The global loads of the NEXT tile's staging data are issued at the start of the rows phase of
// the current one and land in registers while the rows are sorted: the memory round trip in front of every tile (5-8 k
// cycles of a 45 k-cycle tile, every wave waiting) is off the critical path, and the per-workgroup set-up (boundary items,
// cofactor table, the constant part of the tile context) is paid once per workgroup instead of once per tile.
// __launch_bounds__(256, 4): 128 VGPRs; with NC = 72 the 39.6 KB of LDS allow 4 workgroups per CU.
template <int NC>
struct StageRegs {
    static constexpr int extra = NC - 64;
    static constexpr int NLF = (2 * LW) / NTH;        // full rounds of linear words
    static constexpr int NLT = (2 * LW) - NLF * NTH;  // tail
    // virtual-lane slots: (plane, row group, first lanes again).  3 planes x 8 row groups x 16 extra lanes (NC = 80,
    // tile with N in reach) are 384 slots: two rounds of the 256 threads cover every instantiated width.
    static constexpr int NXR = (3 * RG * extra + NTH - 1) / NTH;
    static_assert(3 * RG * extra <= NXR * NTH, "virtual-lane staging rounds do not cover the image width");
    static_assert(NLF >= 4, "the not-ACGT plane of a tile with N in reach travels in the window's registers");
    static_assert(NLF % 2 == 0, "linear words travel as pairs");
    prf_u32x4 vh0, vh1, vl0, vl1;
    // clean tile: the linear window, two 64-bit words per vector; tile with N in reach: q[0], q[1] = its two X-plane slots.
    // (One set of registers for both, written by loads only -- no conversion, which would wait for the data -- and the
    // two kinds of tile are separate branches from the first load to the last: a shared prefix made the compiler wait for
    // the image loads before it issued the window loads.)
    prf_u32x4 q[NLF / 2];
    u64 lt;
    prf_u32x4 ev[NXR], en[NXR];
    uint4 info;
    u32 entry;    // the launch-list entry these registers belong to

    // All global loads of a thread are issued back to back, nothing waits for them here.  Every address is a wave-uniform
    // base (scalar registers) plus a 32-bit per-thread offset: per-thread 64-bit pointers would be hoisted out of the tile
    // loop and spilled.
    __device__ __forceinline__ static prf_u32x4 ld16(const void *ubase, u32 byte_off) {
        return *reinterpret_cast<const prf_u32x4 *>(reinterpret_cast<const char *>(ubase) + byte_off);
    }
    __device__ __forceinline__ static u64 ld8(const void *ubase, u32 byte_off) {
        return *reinterpret_cast<const u64 *>(reinterpret_cast<const char *>(ubase) + byte_off);
    }
    __device__ __forceinline__ void load(const prf_vscan_args &g, u32 e, int tid) {
        e = (u32)__builtin_amdgcn_readfirstlane((int)e);
        entry = e;
        const bool hasx = (e & PRF_LAUNCH_MIXED) != 0;
        const u64 tile = e & ~PRF_LAUNCH_MIXED;
        const prf_u32x4 *ph = reinterpret_cast<const prf_u32x4 *>(g.VH) + tile * (RG * 64), *pL = reinterpret_cast<const prf_u32x4 *>(g.VL) + tile * (RG * 64),
                        *px = reinterpret_cast<const prf_u32x4 *>(g.VX) + tile * (RG * 64);
        const int np = hasx ? 3 : 2;
        // slot (rg + 4*j, l) of a plane, rg = tid >> 6, l = tid & 63: slot index tid + 256 j
        u32 ut = (u32)tid;
        asm volatile("" : "+v"(ut));  // (opaque: keeps the address arithmetic inside the loop, see above)
        const u32 o16 = ut * 16u, o8 = ut * 8u;
        lt = 0;
        if (hasx) {
            vh0 = ld16(ph, o16); vh1 = ld16(ph, o16 + 4u * 64u * 16u); vl0 = ld16(pL, o16); vl1 = ld16(pL, o16 + 4u * 64u * 16u);
            q[0] = ld16(px, o16);
            q[1] = ld16(px, o16 + 4u * 64u * 16u);
            static_for<2, NLF / 2>([&](auto ic) { q[decltype(ic)::value] = prf_u32x4{0, 0, 0, 0}; });
        } else {
            vh0 = ld16(ph, o16); vh1 = ld16(ph, o16 + 4u * 64u * 16u); vl0 = ld16(pL, o16); vl1 = ld16(pL, o16 + 4u * 64u * 16u);
            // the planes have readable padding in front
            const u64 *wh = g.H + ((long long)(tile * PRF_TILE_WORDS) - LIN_PRE), *wl = g.L + ((long long)(tile * PRF_TILE_WORDS) - LIN_PRE - LW);
            static_for<0, NLF>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const u32 idx = ut + (u32)(i * NTH);  // one round straddles H -> L
                const u32 ob = o8 + (u32)(i * NTH * 8);
                u64 w;
                if constexpr ((i + 1) * NTH <= LW) w = ld8(wh, ob);
                else if constexpr (i * NTH >= LW) w = ld8(wl, ob);
                else w = idx < (u32)LW ? ld8(wh, ob) : ld8(wl, ob);
                if constexpr (i % 2 == 0) { q[i / 2].x = (u32)w; q[i / 2].y = (u32)(w >> 32); }
                else { q[i / 2].z = (u32)w; q[i / 2].w = (u32)(w >> 32); }
            });
            if (tid < NLT) lt = ld8(wl, o8 + (u32)(NLF * NTH * 8));
        }
        const u32 n_extra = (u32)(np * RG * extra);
        static_for<0, NXR>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            const u32 sx = ut + (u32)(r * NTH);
            ev[r] = prf_u32x4{0, 0, 0, 0};
            en[r] = ev[r];
            if (sx < n_extra) {
                // the plane is the same for a whole wave (RG * extra is a multiple of 64): a scalar select of the base
                static_assert((RG * extra) % 64 == 0, "virtual-lane staging: one plane per wave");
                const u32 pw = (u32)__builtin_amdgcn_readfirstlane((int)(sx / (u32)(RG * extra)));
                const u32 erg = (sx / (u32)extra) % (u32)RG, el = sx % (u32)extra;
                const u32 i0 = (erg * 64u + el) * 16u, i1 = i0 + (u32)(RG * 64) * 16u;  // the same slot of the next tile
                if (pw == 0) {  // (three branches, not a selected pointer: the compiler makes a table in scratch memory of that)
                    ev[r] = ld16(ph, i0);
                    en[r] = ld16(ph, i1);
                } else if (pw == 1) {
                    ev[r] = ld16(pL, i0);
                    en[r] = ld16(pL, i1);
                } else {
                    ev[r] = ld16(px, i0);
                    en[r] = ld16(px, i1);
                }
            }
        });
        // the tile's contig (contigs start on tile boundaries, so every run that starts in this tile lies in it): one load
        info = make_uint4(0, 0, 0, 0);
        if (tid == 0) info = g.tile_info[tile];
    }

    __device__ __forceinline__ void clear() {
        const prf_u32x4 z = {0, 0, 0, 0};
        vh0 = vh1 = vl0 = vl1 = z;
        static_for<0, NLF / 2>([&](auto ic) { q[decltype(ic)::value] = z; });
        lt = 0;
        static_for<0, NXR>([&](auto rc) { ev[decltype(rc)::value] = en[decltype(rc)::value] = z; });
        info = make_uint4(0, 0, 0, 0);
        entry = 0;
    }

    // registers -> the LDS image, the linear window (clean tile) or the not-ACGT plane in its place (tile with N in reach)
    __device__ __forceinline__ void store(prf_lds_u4 *vimg, prf_lds_u4 *ximg, prf_lds_u64 *lin, int tid) const {
        constexpr int nc = NC;
        const bool hasx = (entry & PRF_LAUNCH_MIXED) != 0;
        const int np = hasx ? 3 : 2;
        const int rg = tid >> 6, l = tid & 63;
        vimg[(0 * RG + rg) * nc + l] = vh0;
        vimg[(0 * RG + rg + 4) * nc + l] = vh1;
        vimg[(1 * RG + rg) * nc + l] = vl0;
        vimg[(1 * RG + rg + 4) * nc + l] = vl1;
        if (hasx) {
            ximg[rg * nc + l] = q[0];
            ximg[(rg + 4) * nc + l] = q[1];
        } else {
            static_for<0, NLF>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                lin[tid + i * NTH] = i % 2 == 0 ? ((u64)q[i / 2].x | ((u64)q[i / 2].y << 32)) : ((u64)q[i / 2].z | ((u64)q[i / 2].w << 32));
            });
            if (tid < NLT) lin[NLF * NTH + tid] = lt;
        }
        const u32 n_extra = (u32)(np * RG * extra);
        static_for<0, NXR>([&](auto rc) {  // virtual lanes 64..: the first lanes again, one bit up, bit 31 from the next tile
            constexpr int r = decltype(rc)::value;
            const u32 sx = (u32)tid + (u32)(r * NTH);
            if (sx < n_extra) {
                const u32 p = sx / (u32)(RG * extra), erg = (sx / (u32)extra) % (u32)RG, el = sx % (u32)extra;
                const prf_u32x4 v = (ev[r] >> 1) | (en[r] << 31);
                const u32 dst = erg * (u32)nc + 64u + el;
                if (p == 2) ximg[dst] = v;
                else vimg[p * RG * nc + dst] = v;
            }
        });
    }
};

template <int NC>
__global__ __launch_bounds__(NTH, 4) void prf_vscan_kernel(prf_vscan_args g) {
    constexpr int nc = NC;
    prf_lds_u4 *vimg = (prf_lds_u4 *)(prf_smem + SMEM_HDR);
    constexpr u32 lin_off = (u32)SMEM_HDR + (u32)((size_t)2 * RG * nc * sizeof(prf_u32x4));
    prf_lds_u64 *lin = (prf_lds_u64 *)(prf_smem + lin_off);
    prf_lds_u4 *ximg = (prf_lds_u4 *)(prf_smem + lin_off);  // tiles with N in reach: instead of the window
    prf_lds_u64 *recs = lin + 2 * LW;
    prf_lds_u32 *hotw = (prf_lds_u32 *)(recs + MAX_WAVES * REC_PER_WAVE);  // exact tasks: [task][lane] stream words, then 16 x (k | M << 16)
    prf_lds_u32 *bitems = hotw + g.plan.n_exact * 64u + 16u;                // boundary items, plan.n_group_k of them
    prf_lds_u32 *hdr_cnt = (prf_lds_u32 *)(prf_smem + 128);

    const int tid0 = (int)threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    // (when the launch list is one contiguous range of clean tiles -- a contig without N blocks -- the tile index is
    // arithmetic: no dependent load in front of the staging loads)
    auto entry_of = [&](u32 sl) -> u32 { return g.flat_base != ~0u ? g.flat_base + sl : g.launch_list[sl]; };

    // ---- once per workgroup ----
    set_prio(g.plan.prio & 3u);
    StageRegs<NC> sr;
    sr.load(g, entry_of(blockIdx.x), tid0);
    // boundary items: (motif size, examined-group stride) of every motif size a group task scans
    for (u32 v = (u32)tid0; v < 8u * g.plan.n_tasks; v += (u32)NTH) {
        const prf_vtask task = g.plan.tasks[v >> 3];
        const u32 kk = v & 7u;
        if (task.kind == 0 && ((task.valid >> kk) & 1u))
            bitems[(u32)task.item0 + (u32)__builtin_popcount((u32)task.valid & ((1u << kk) - 1u))] = ((u32)task.k0 + kk) | ((u32)task.stride << 16);
    }
    {
        prf_lds_u32 *cof_lds = bitems + g.plan.n_group_k;
        for (int i = tid0; i < (int)g.plan.cof_words; i += NTH) cof_lds[i] = prf_cof_table.v[i];
    }

    if (tid0 == 0) {  // the constant part of the tile context
        TileCtx *tcw = reinterpret_cast<TileCtx *>(prf_smem);
        tcw->H = g.H; tcw->L = g.L; tcw->X = g.X;
        tcw->E = g.E;
        tcw->slab_cap = g.slab_cap;
        tcw->min_repeats = g.min_repeats;
        tcw->min_span = g.min_span;
        tcw->lin_off = lin_off;
        tcw->hotw_off = lin_off + (u32)(2 * LW * sizeof(u64) + MAX_WAVES * REC_PER_WAVE * sizeof(u64));
        tcw->n_exact = g.plan.n_exact;
        tcw->k_exact0 = g.plan.k_exact0;
        tcw->cof_off = tcw->hotw_off + 4u * (g.plan.n_exact * 64u + 16u + g.plan.n_group_k);
    }

    // Launch slots are handed out dynamically (tiles differ in cost by a factor of three; a fixed stride leaves the last
    // workgroups running alone for 8 % of the scan): XCD x -- the workgroups b = x mod 8 -- takes the slots = x mod 8, the
    // first one per workgroup by index, the following ones by a ticket counter of its own (one atomic per tile on eight
    // separate words; the ticket is drawn at the top of a tile and needed at its end).
    const u32 xcd = blockIdx.x & 7u;
    const u32 first_ticket = (gridDim.x - xcd + 7u) >> 3;  // workgroups of this XCD = slots taken without a ticket
    u64 *ticket_word = g.counters + (PRF_CNT_SHARD0 + xcd * PRF_CNT_SHARD_STRIDE + PRF_SH_TILE_TICKET);
    prf_lds_u32 *next_words = (prf_lds_u32 *)(prf_smem + 176);  // {next slot, its launch-list entry}
    u32 slot_next = 0;
    for (u32 slot = blockIdx.x; slot < g.n_launch; slot = slot_next) {
    // (opaque per round: what derives from the thread index is recomputed, not carried through the scan's calls in
    // registers that would have to be spilled)
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const u32 entry = sr.entry;
    const bool hasx = (entry & PRF_LAUNCH_MIXED) != 0;
    const u64 tile = entry & ~PRF_LAUNCH_MIXED;

    PRF_STAMP(0);
#ifdef PRF_STAMPS
    if (g.dbg && lane == 0) g.dbg[((u64)slot * MAX_WAVES + wave) * 16 + 12] = __builtin_amdgcn_s_memrealtime();  // 100 MHz, chip-wide
#endif
    // ---- 1. stage: the registers loaded during the previous tile's rows phase (or above) -> LDS ----
    set_prio(g.plan.prio & 3u);
    {
        sr.store(vimg, ximg, lin, tid);
        if (tid < 2 * MAX_WAVES + 3) hdr_cnt[tid] = 0;  // list lengths, row count, direct-row count, flushed records
        if (tid == 0) {  // the tile's part of the context (the rest was written once, above)
            TileCtx *tcw = reinterpret_cast<TileCtx *>(prf_smem);
            tcw->w0 = tile * PRF_TILE_WORDS - LIN_PRE;
            tcw->xz_lo = hasx ? 0 : tile * PRF_TILE;  // a clean tile and its successor hold no not-ACGT position
            tcw->xz_hi = hasx ? 0 : (tile + 2) * PRF_TILE;
            tcw->slab = g.slabs + (u64)slot * g.slab_cap;
            tcw->contig = sr.info.x;
            tcw->contig_base = (u64)sr.info.z | ((u64)sr.info.w << 32);
            tcw->tile_base = tile * PRF_TILE;
            tcw->has_lin = hasx ? 0u : 1u;
            tcw->xwin_off = 0u;
        }
    }
    PRF_STAMP(1);
    __syncthreads();
    PRF_STAMP(2);
    // the ticket for the tile after this one: drawn here, behind the wait for the staging loads (an atomic in front of it
    // would be waited for with them: 4 k cycles), used before the last barrier of the tile.  The compiler turns the
    // returning atomic into a wave-aggregated one and waits for its value on the spot: the plan deals it to the wave
    // with the least other work.
    u64 ticket = 0;
    const bool ticket_thread = tid == (int)(g.plan.ticket_wave * 64u);  // (the plan's least loaded wave)
    if (ticket_thread) ticket = atomicAdd(ticket_word, 1ull);

    // ---- 2. scan ----
    // Issue priority (plan.prio): each SIMD hosts waves of four workgroups in different phases, about a third busy; the
    // phases that are chains of dependent LDS round trips go first, the scan takes the slots that are left.
    set_prio(((g.plan.slack_waves >> wave) & 1u) ? (g.plan.prio >> 4) & 3u : (g.plan.prio >> 2) & 3u);
    Emit em;
    em.recs = recs + wave * REC_PER_WAVE;
    em.all_recs = (prf_lds_cu64 *)recs;
    em.wave = wave;
    em.lane = lane;
    em.cnt = 0;
    em.flushed = 0;
#ifdef PRF_STAMPS
    u64 *task_dbg = g.dbg ? g.dbg + ((u64)slot * MAX_WAVES + wave) * 16 : nullptr;
#else
    u64 *task_dbg = nullptr;
#endif
    if (hasx) run_tasks<true, NC>((prf_lds_cu4 *)vimg, (prf_lds_cu4 *)ximg, hotw, g.plan, wave, lane, em, task_dbg);
    else run_tasks<false, NC>((prf_lds_cu4 *)vimg, (prf_lds_cu4 *)ximg, hotw, g.plan, wave, lane, em, task_dbg);
    if (lane == 0) {
        hdr_cnt[wave] = em.cnt;
        if (em.flushed) atomicAdd((u32 *)(prf_smem + 168), em.flushed);
    }
    PRF_STAMP(3);
    __syncthreads();  // the image is dead from here on: the row list may lie there
    if (hasx) {
        // A tile with N in reach kept its not-ACGT plane where the linear window belongs.  Now that the scan is over the
        // windows of all three linear planes are staged -- H and L in the window's place, X in the dead image behind the row
        // and flag lists -- so that this tile, too, verifies from LDS (a look at the global planes is a memory round trip).
        constexpr u32 xwin_off = (u32)SMEM_HDR + (u32)ROW_CAP_LDS * 16u + 4u * 512u * 2u;
        static_assert(xwin_off + LW * 8 <= SMEM_HDR + 2 * RG * NC * 16, "the X window must fit the dead image");
        prf_lds_u64 *xwin = (prf_lds_u64 *)(prf_smem + xwin_off);
        // (wave-uniform bases + an opaque 32-bit thread offset, as in StageRegs::load)
        const long long w0 = (long long)(tile * PRF_TILE_WORDS) - LIN_PRE;
        const u64 *wh = g.H + w0, *wl = g.L + (w0 - LW), *wx = g.X + (w0 - 2 * LW);
        u32 ut = (u32)tid;
        asm volatile("" : "+v"(ut));
        constexpr int NR = (3 * LW + NTH - 1) / NTH;
        u64 v[NR];
        static_for<0, NR>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const u32 idx = ut + (u32)(i * NTH), ob = idx * 8u;
            if constexpr ((i + 1) * NTH <= LW) v[i] = StageRegs<NC>::ld8(wh, ob);
            else if constexpr (i * NTH >= LW && (i + 1) * NTH <= 2 * LW) v[i] = StageRegs<NC>::ld8(wl, ob);
            else if constexpr (i * NTH >= 2 * LW && (i + 1) * NTH <= 3 * LW) v[i] = StageRegs<NC>::ld8(wx, ob);
            else v[i] = idx < (u32)LW ? StageRegs<NC>::ld8(wh, ob) : (idx < 2u * LW ? StageRegs<NC>::ld8(wl, ob) : (idx < 3u * LW ? StageRegs<NC>::ld8(wx, ob) : 0ull));
        });
        static_for<0, NR>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const u32 idx = ut + (u32)(i * NTH);
            if (idx < 2u * LW) lin[idx] = v[i];
            else if (idx < 3u * LW) xwin[idx - 2u * LW] = v[i];
        });
        if (tid == 0) {
            TileCtx *tcw = reinterpret_cast<TileCtx *>(prf_smem);
            tcw->has_lin = 1u;
            tcw->xwin_off = xwin_off;
        }
        __syncthreads();
    }
    PRF_STAMP(4);

    // ---- 3. verify what is left in the lists, all waves together: every candidate -> a row in the tile's LDS list, or nothing ----
    // (the record waves are the critical path of this phase, the flag waves wait for them at the barrier below)
    set_prio(wave < MAX_WAVES / 2 ? (g.plan.prio >> 6) & 3u : (g.plan.prio >> 8) & 3u);
    u32 n_flags = verify_all((prf_lds_cu64 *)recs, (prf_lds_cu32 *)bitems, g.plan.n_group_k, (u32)tid, task_dbg);
    set_prio((g.plan.prio >> 10) & 3u);
    if (tid == 0) {
        // statistics: candidates looked at = (stream, exact task) flags (thread 0 holds their number) + group-task records
        n_flags += *(prf_lds_u32 *)(prf_smem + 168) + hdr_cnt[0] + hdr_cnt[1] + hdr_cnt[2] + hdr_cnt[3];
        if (n_flags)
            atomicAdd(&g.counters[PRF_CNT_SHARD0 + (tile % PRF_CNT_NSHARD) * PRF_CNT_SHARD_STRIDE + PRF_SH_CAND], (u64)n_flags);
    }
    if (ticket_thread) {  // the next slot and its entry, for everybody behind the barrier
        const u32 sn = (first_ticket + (u32)ticket) * 8u + xcd;
        next_words[0] = sn;
        next_words[1] = sn < g.n_launch ? entry_of(sn) : entry;  // (last round: this tile again, unused)
    }
    PRF_STAMP(5);
    __syncthreads();
    PRF_STAMP(6);
    const u64 nw = *(prf_lds_cu64 *)next_words;  // (one read)
    slot_next = (u32)__builtin_amdgcn_readfirstlane((int)(u32)nw);
    const u32 entry_next = (u32)(nw >> 32);

    // the next tile's staging data: loads issued now, consumed at the top of the loop
    // (every register is written on both paths: dead from the stage to here, not carried around the loop)
    if (slot_next < g.n_launch) sr.load(g, entry_next, tid);
    else sr.clear();

    // ---- 4. the tile's rows, sorted by (start, end), into its slab: [rows written directly, unsorted][the LDS list, sorted].
    // Rank of a row = number of rows of the list with a smaller key; keys are distinct ((start, end) pairs never collide
    // between motif sizes).
    const TileCtx &tc = *reinterpret_cast<const TileCtx *>(prf_smem);
    // (the slab through the kernel argument, not through the pointer in the LDS context: that one is generic, and a FLAT
    // store counts as an LDS operation too -- the next LDS wait would sit behind the prefetch loads just issued)
    prf_hit_dev *slab = g.slabs + (u64)slot * g.slab_cap;
    u32 n_listed = *smem_row_cnt();
    u32 n_direct = *smem_direct_cnt();
    if (n_direct && n_listed + n_direct <= (u32)ROW_CAP_LDS && n_direct <= tc.slab_cap) {
        // Rows that went straight to the slab (a wave emptied its record list in the middle of the scan) and still fit the
        // list: read them back and sort them with the others -- the tile stays sorted.
        if ((u32)tid < n_direct) {
            const prf_hit_dev h = slab[tid];
            const u64 a = h.start + tc.contig_base, end = h.end + tc.contig_base, span = end - a;
            smem_row_keys()[n_listed + tid] = ((u32)(a - tc.tile_base) << 16) | (span < 65535ull ? (u32)span : 65535u);
            smem_row_ks()[n_listed + tid] = h.k;
            smem_row_ends()[n_listed + tid] = end;
        }
        __syncthreads();  // wave-uniform condition: every thread gets here
        n_listed += n_direct;
        n_direct = 0;
    }
    const u32 n_sorted = n_listed < (u32)ROW_CAP_LDS ? n_listed : (u32)ROW_CAP_LDS;
    const u32 n_rows = n_sorted + n_direct;  // rows beyond the list's capacity were counted in n_direct
    if (n_sorted) {
        // P = 256 / n threads per row (a power of two, adjacent lanes): each counts the smaller keys of its share of the
        // list, the shares are added up across the P lanes.  Dependent LDS round trips are what this phase costs (~500 cycles
        // each with the other workgroups' scans on the CU): the row's own key, end and motif size and the first 32 keys of
        // the lane's share are ONE batch of reads; the shares are added with DPP moves, not LDS shuffles.
        // (P <= 8: eight reads of a lane, 4 P keys apart, stay inside the 256 padded keys)
        const u32 lg = n_sorted > 128u ? 0u : (n_sorted > 64u ? 1u : (n_sorted > 32u ? 2u : 3u));
        const u32 P = 1u << lg, row = (u32)tid >> lg, part = (u32)tid & (P - 1u);
        prf_lds_u32 *keys = smem_row_keys();
        // the list is padded to its capacity with the largest key: no bounds test per key in the loop below
        if ((u32)tid >= n_sorted) keys[tid] = 0xFFFFFFFFu;
        __syncthreads();  // (n_sorted is the same for every thread)
        typedef __attribute__((address_space(3))) const prf_u32x4 prf_lds_ckey4;
        prf_lds_ckey4 *k4 = (prf_lds_ckey4 *)keys;
        const u32 r = row < n_sorted ? row : 0u;  // (lanes without a row read row 0: harmless)
        const u32 mine = keys[r];
        const u64 end = smem_row_ends()[r];
        const u32 kk = smem_row_ks()[r];
        u32 rank = 0;
        // part p takes the keys 4 p .. 4 p + 3, then 4 P further on, ...: one 16-byte read per four keys, eight reads in flight
        for (u32 c0 = part; 4u * c0 < n_sorted; c0 += 8u * P) {
            prf_u32x4 v[8];
#pragma unroll
            for (u32 j = 0; j < 8u; j++) v[j] = k4[c0 + j * P];
            // (all eight reads in flight before the first compare: left alone the compiler issues them two at a time, a
            // round trip per pair, to save registers it does not need here)
            asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
            for (u32 j = 0; j < 8u; j++) {
                rank += v[j].x < mine ? 1u : 0u;  // (keys past the list are 0xFFFFFFFF: never smaller)
                rank += v[j].y < mine ? 1u : 0u;
                rank += v[j].z < mine ? 1u : 0u;
                rank += v[j].w < mine ? 1u : 0u;
            }
        }
        // sum over the P adjacent lanes of a row (wave-uniform P): xor 1, xor 2 by quad permutes; after those all lanes of a
        // quad agree, so the half-row and row mirrors pair the right partners for 4 and 8
        if (P >= 2u) rank += (u32)__builtin_amdgcn_update_dpp(0, (int)rank, 0xB1, 0xF, 0xF, true);
        if (P >= 4u) rank += (u32)__builtin_amdgcn_update_dpp(0, (int)rank, 0x4E, 0xF, 0xF, true);
        if (P >= 8u) rank += (u32)__builtin_amdgcn_update_dpp(0, (int)rank, 0x141, 0xF, 0xF, true);
        const u32 dst = n_direct + rank;
        if (row < n_sorted && part == 0 && dst < tc.slab_cap) {
            prf_hit_dev h;
            h.start = tc.tile_base + (mine >> 16) - tc.contig_base;
            h.end = end - tc.contig_base;
            h.k = kk;
            h.contig = tc.contig;
            slab[dst] = h;
        }
    }
    if (tid == 0) {
        g.slab_count[slot] = n_rows;
        const u32 stored = n_rows < tc.slab_cap ? n_rows : tc.slab_cap;
        if (stored) {  // rows in front of a gather workgroup's slots: two levels of sums
            atomicAdd(&g.block_sum[slot >> g.gather_shift], stored);
            atomicAdd(&g.block_sum[g.super_off + ((slot >> g.gather_shift) / PRF_GATHER_SUPER)], stored);
        }
        if (n_rows > tc.slab_cap) atomicMax(&g.counters[PRF_CNT_HIT_OVF], (u64)n_rows);
        if (n_direct) atomicMax(&g.counters[PRF_CNT_UNSORTED], 1ull);
    }
    PRF_STAMP(7);
#ifdef PRF_STAMPS
    if (g.dbg && lane == 0) g.dbg[((u64)slot * MAX_WAVES + wave) * 16 + 13] = __builtin_amdgcn_s_memrealtime();
#endif
    __syncthreads();  // the header, the row list and the image region are rewritten for the next tile
    }
}

// ---------------------------------------------------------------------------------------------------
// Row gather: the slabs, in launch (= position) order, become ONE compact array.  Workgroup w owns the launch slots
// [8 w, 8 w + 8): the rows in front of them are sums the scan kernel has added up (two atomics per tile: per 8 slots and per
// 512 slots); it scans its own 8 counts and copies its slabs word by word, four loads in flight per thread.  The workgroup
// that finishes last hands the counter block to the host (mapped memory, no copy call), and clears the sums and the
// counter block of the next scan (no memset call).
__global__ __launch_bounds__(256) void prf_vgather_kernel(prf_vgather_args g) {
    __shared__ u64 part[4];
    __shared__ u32 offs[PRF_GATHER_SLOTS_MAX + 1];  // in words (3 per row)
    __shared__ u64 ticket_lds;
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const u32 n_slots = 1u << g.gather_shift;  // launch slots per workgroup: 8 (small launches: more workgroups) or 64
    const u32 first = blockIdx.x << g.gather_shift;
    const u32 my_super = blockIdx.x / PRF_GATHER_SUPER;
    u64 before = 0;
    for (u32 i = tid; i < my_super; i += 256u) before += g.block_sum[g.super_off + i];
    if (tid < blockIdx.x - my_super * PRF_GATHER_SUPER) before += g.block_sum[my_super * PRF_GATHER_SUPER + tid];
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
    if (lane == 0) part[wave] = before;
    if (tid < 64u) {  // exclusive scan of the counts (loaded in parallel)
        u32 c = (tid < n_slots && first + tid < g.n_launch) ? g.slab_count[first + tid] : 0u;
        c = 3u * (c < g.slab_cap ? c : g.slab_cap);
        u32 incl = c;
        for (int o = 1; o < 64; o <<= 1) {
            const u32 up = __shfl_up(incl, o, 64);
            if ((int)tid >= o) incl += up;
        }
        offs[tid + 1] = incl;
        if (tid == 0) offs[0] = 0;
    }
    __syncthreads();
    const u64 base0 = part[0] + part[1] + part[2] + part[3];  // rows in front of this workgroup's slots
    const u32 n_words = offs[n_slots];
    // rows beyond the capacity stay behind: the host sees the total beyond the capacity, grows the array, rescans
    const u64 room_rows = base0 < g.rows_cap ? g.rows_cap - base0 : 0;
    const u64 room_words = 3ull * room_rows;
    const u32 n_copy = (u64)n_words < room_words ? n_words : (u32)room_words;
    u64 *dst = reinterpret_cast<u64 *>(g.rows + base0);
    for (u32 w0 = tid; w0 < n_copy; w0 += 1024u) {
        u64 v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const u32 w = w0 + 256u * (u32)j;
            v[j] = 0;
            if (w < n_copy) {
                u32 lo = 0, hi = n_slots;  // the slot whose words hold w: offs[lo] <= w < offs[lo + 1]
                while (hi - lo > 1) {
                    const u32 mid = (lo + hi) >> 1;
                    if (offs[mid] <= w) lo = mid; else hi = mid;
                }
                v[j] = reinterpret_cast<const u64 *>(g.slabs + (u64)(first + lo) * g.slab_cap)[w - offs[lo]];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const u32 w = w0 + 256u * (u32)j;
            if (w < n_copy) dst[w] = v[j];
        }
    }
    // the workgroup of the last slots knows the total
    if (blockIdx.x == gridDim.x - 1 && tid == 0) atomicAdd(&g.counters[PRF_CNT_ROWS], base0 + n_words / 3u);
    __syncthreads();  // every wave's stores and atomics are issued; the barrier waits for outstanding memory operations
    // Finishing tickets in two levels (one word takes ~90 atomics per microsecond: thousands of workgroups on ONE ticket word
    // would cost more than the copy): a ticket per 64 workgroups, and the last of each 64 draws a global one.
    const u32 n_supers = (gridDim.x - 1u) / PRF_GATHER_SUPER + 1u;
    if (tid == 0) {
        const u32 in_super = my_super + 1u < n_supers ? PRF_GATHER_SUPER : gridDim.x - my_super * PRF_GATHER_SUPER;
        u64 t = 0;
        if (atomicAdd(&g.block_sum[g.super_off + n_supers + my_super], 1u) == in_super - 1u)
            t = atomicAdd(&g.counters[PRF_CNT_TICKET], 1ull) + 1ull;
        ticket_lds = t;  // n_supers: this workgroup is the last one of the whole grid
    }
    __syncthreads();
    // ---- the last workgroup hands the counter block to the host.  The counters are only ever touched
    // by device-scope atomics, performed at the coherence point, and every workgroup's were issued in front of the
    // barrier that precedes its ticket (s_waitcnt vmcnt(0) before s_barrier), so they precede the last ticket.  Every
    // other workgroup has read its sums by then: they are cleared for the next scan.
    if (ticket_lds == (u64)n_supers) {
        for (u32 i = tid; i < 2u * n_supers; i += 256u) g.block_sum[g.super_off + i] = 0;
        for (u32 i = tid; i < gridDim.x; i += 256u) g.block_sum[i] = 0;
        for (u32 i = tid; i < (u32)PRF_CNT_N; i += 256u) {
            const u64 v = atomicAdd(&g.counters[i], 0ull);
            g.host_counters[i] = v;
            g.next_counters[i] = 0;
            if (i == (u32)PRF_CNT_ROWS && g.count_row) {  // a caller-owned row array carries its own length
                prf_hit_dev h;
                h.start = v < g.rows_cap ? v : g.rows_cap;
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
    u32 exo = 0;  // any letter other than A, C, G, T, N
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
            exo |= (ok | (f == 'N')) ^ 1u;
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
    const bool w_exo = __builtin_amdgcn_ballot_w64(exo != 0) != 0;
    if (lane == 0) any_all[tile] = (unsigned char)((w_any ? 1 : 0) | (w_all ? 2 : 0) | (w_exo ? 4 : 0));
}

// class: 3 = a symbol outside ACGTN in this tile, the one before or the one after (such tiles are scanned by the generic
// kernels, with the symbols' own planes); 2 = only not-ACGT; 1 = some not-ACGT in this tile or the next (whose first lanes
// are this tile's virtual lanes 64..); 0 = clean
__global__ void prf_tile_class_kernel(const unsigned char *__restrict__ any_all, unsigned char *__restrict__ cls, u64 ntiles) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ntiles) return;
    const unsigned char a = any_all[i];
    const unsigned char b = (i + 1 < ntiles) ? any_all[i + 1] : (unsigned char)3;
    const unsigned char p = i ? any_all[i - 1] : (unsigned char)0;
    cls[i] = ((a | b | p) & 4) ? 3 : ((a & 2) ? 2 : (((a | b) & 1) ? 1 : 0));
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
    u32 reach = 0;       // furthest row of a lane's extended stream any task reads
    u32 covered_to = 0;  // group chunks cover motif sizes below this
    for (u32 k = kmin; k <= kmax; k++) {
        const long long M = prf_min_matches(k, min_repeats, min_span);
        if (M < SMALL_M) {
            Item it;
            it.t.k0 = (unsigned short)k;   // <= 14: M >= (min_repeats - 1) * k >= k
            it.t.kind = (unsigned char)M;  // M >= 1 because min_repeats >= 2
            it.t.valid = 1;
            it.t.stride = 1;
            // measured (stamps build, 4 workgroups per CU, units of 8.5 cycles): 2.8 k cycles for M = 6 (one operation per start
            // word), 4.0 - 4.4 k for M = 7 .. 14 (two)
            it.cost = M <= 6 ? 250u + 13u * (u32)M : 440u + 5u * (u32)M;
            items.push_back(it);
            reach = std::max<u32>(reach, 4 * (((u32)T + (u32)M - 1 + k + 3) / 4) - 1);
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
            // measured: 6.3 k / 4.2 k / 3.9 k cycles with 8 sizes, 2.4 k for stride 4 with 3
            it.cost = (stride == 1 ? 500u : (stride == 2 ? 260u : 200u)) + 30u * (u32)__builtin_popcount(valid);
            items.push_back(it);
            reach = std::max<u32>(reach, 24 + k0 + 15);
            covered_to = k0 + 8;
        }
    }
    if (items.size() > PRF_VMAX_TASKS) return false;
    // longest-processing-time-first assignment to at most 4 waves
    const u32 nw = std::max<u32>(1, std::min<u32>(PRF_VMAX_WAVES, (u32)items.size()));
    std::vector<std::vector<Item>> bins(nw);
    std::vector<u32> load(nw, 0);
    std::vector<Item> sorted = items;
    std::stable_sort(sorted.begin(), sorted.end(), [](const Item &a, const Item &b) { return a.cost > b.cost; });
    // The ticket for the workgroup's next launch slot is an atomic whose value the compiler waits for on the spot (~3 k
    // cycles with a thousand workgroups drawing): it is dealt like a task, to the wave with the least other work -- or to
    // a wave without tasks, if there is one.  (Tried and dropped: the waves of a workgroup pulling tasks from one list
    // through an LDS counter at run time -- every wave's scan got 2-3 k cycles longer; the stride-1 group task cut in two
    // halves of four sizes -- a half costs three quarters of the whole, its four blocks are LDS latency, not arithmetic.)
    constexpr u32 TICKET_COST = 350;
    bool ticket_dealt = nw < (u32)PRF_VMAX_WAVES;
    plan->ticket_wave = nw < (u32)PRF_VMAX_WAVES ? nw : 0;
    for (const Item &it : sorted) {
        if (!ticket_dealt && it.cost <= TICKET_COST) {
            plan->ticket_wave = (u32)(std::min_element(load.begin(), load.end()) - load.begin());
            load[plan->ticket_wave] += TICKET_COST;
            ticket_dealt = true;
        }
        const u32 w = (u32)(std::min_element(load.begin(), load.end()) - load.begin());
        bins[w].push_back(it);
        load[w] += it.cost;
    }
    if (!ticket_dealt) plan->ticket_wave = (u32)(std::min_element(load.begin(), load.end()) - load.begin());
    {
        // default: the short, latency-bound phases (stage, the record waves of the verify phase, rows) at priority 2, the scan
        // (long, plenty of independent arithmetic) and the flag waves (they wait at the barrier anyway) at 0.  Measured on
        // the default workload (tools/prio_sweep.sh, gpurun_out/prio_sweep*.txt): 0.775 ms without priorities, 0.748-0.757
        // with any setting that raises records and rows; random sequence (few candidates) is indifferent.
        // PRF_PRIO (diagnostic) overrides.
        static const u32 prio_cfg = getenv("PRF_PRIO") ? (u32)strtoul(getenv("PRF_PRIO"), nullptr, 0) : 0xA02u;
        plan->prio = prio_cfg;
        const u32 busiest = *std::max_element(load.begin(), load.end());
        plan->slack_waves = 0;
        for (u32 w = 0; w < (u32)PRF_VMAX_WAVES; w++)
            if (w >= nw || 10u * load[w] < 9u * busiest) plan->slack_waves |= 1u << w;
    }
    plan->n_waves = nw;
    plan->n_tasks = 0;
    plan->n_group_k = 0;
    plan->n_exact = 0;
    // the motif sizes of the exact tasks are consecutive: M(k) = max((r-1) k, span - k) is V-shaped, so {k : M(k) < 15} is an interval
    u32 k_exact0 = ~0u, k_exact1 = 0, n_exact_items = 0;
    for (const Item &it : items)
        if (it.t.kind) {
            k_exact0 = std::min<u32>(k_exact0, it.t.k0);
            k_exact1 = std::max<u32>(k_exact1, it.t.k0);
            n_exact_items++;
        }
    if (n_exact_items && k_exact1 - k_exact0 + 1 != n_exact_items) return false;
    plan->k_exact0 = n_exact_items ? k_exact0 : 0;
    for (u32 w = 0; w < nw; w++) {
        plan->wave_begin[w] = plan->n_tasks;
        for (const Item &it : bins[w]) {
            prf_vtask t = it.t;
            t.pad = 0;
            t.item0 = 0;
            if (t.kind == 0) {
                t.item0 = (unsigned short)plan->n_group_k;
                plan->n_group_k += (u32)__builtin_popcount((unsigned)t.valid);
            } else {
                plan->n_exact++;
                t.item0 = (unsigned short)(t.k0 - k_exact0);
            }
            plan->tasks[plan->n_tasks++] = t;
        }
    }
    for (u32 w = nw; w <= PRF_VMAX_WAVES; w++) plan->wave_begin[w] = plan->n_tasks;
    const u32 need_nc = 64 + reach / T;  // row r of a lane's extended stream lies in virtual lane + r / 32
    plan->nc = need_nc <= 72 ? 72 : 80;  // the widths the kernel is instantiated for
    plan->cof_words = (kmax + 1 + 3) & ~3u;  // <= PRF_VMAX_K + 4: the table is declared with that many entries
    plan->lds_bytes = (u32)(SMEM_HDR + (size_t)2 * RG * plan->nc * sizeof(uint4) + (size_t)2 * LW * sizeof(u64) +
                            (size_t)MAX_WAVES * REC_PER_WAVE * sizeof(u64) + (size_t)(plan->n_exact * 64 + 16 + plan->n_group_k) * sizeof(u32) +
                            (size_t)plan->cof_words * sizeof(u32));
    return need_nc <= 80;
}

hipError_t prf_vertical_launch(hipStream_t s, const prf_vscan_args &args) {
    if (args.n_launch == 0) return hipSuccess;
    // PRF_LDS_PAD (diagnostic): extra dynamic LDS per workgroup, to measure the scan at a lower occupancy
    static const u32 lds_pad = getenv("PRF_LDS_PAD") ? (u32)atoi(getenv("PRF_LDS_PAD")) : 0u;
    const u32 lds = args.plan.lds_bytes + lds_pad;
    // persistent workgroups: as many as are resident at once (LDS- and register-bound: 4 per CU at most)
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidDevice;
        n_cu = prop.multiProcessorCount;
    }
    const u32 per_cu = std::max(1u, std::min(4u, (160u * 1024u) / std::max(1u, lds)));
    const dim3 grid(std::min(args.n_launch, per_cu * (u32)n_cu)), block(NTH);
    switch (args.plan.nc) {
        case 72: hipLaunchKernelGGL((prf_vscan_kernel<72>), grid, block, lds, s, args); break;
        case 80: hipLaunchKernelGGL((prf_vscan_kernel<80>), grid, block, lds, s, args); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t prf_vertical_gather(hipStream_t s, const prf_vgather_args &args) {
    const u32 n_slots = 1u << args.gather_shift;
    const u32 nb = args.n_launch ? (args.n_launch + n_slots - 1u) / n_slots : 1u;
    hipLaunchKernelGGL(prf_vgather_kernel, dim3(nb), dim3(256), 0, s, args);
    return hipGetLastError();
}

// first tile if the list is one contiguous range of clean tiles, else ~0u
u32 prf_flat_base(const u32 *list, size_t n) {
    if (n == 0 || (list[0] & PRF_LAUNCH_MIXED)) return ~0u;
    for (size_t i = 1; i < n; i++)
        if (list[i] != list[0] + (u32)i) return ~0u;
    return list[0];
}

int prf_vertical_pack(hipStream_t s, const uint8_t *asc, u64 G, prf_vplanes *vp) {
    const u64 ntiles = G / PRF_TILE;  // includes the sentinel tile
    hipError_t e;
    const size_t plane_bytes = (size_t)ntiles * RG * 64 * sizeof(uint4);
    if ((e = hipMalloc((void **)&vp->VH, plane_bytes)) != hipSuccess) return (int)e;
    if ((e = hipMalloc((void **)&vp->VL, plane_bytes)) != hipSuccess) return (int)e;
    if ((e = hipMalloc((void **)&vp->VX, plane_bytes)) != hipSuccess) return (int)e;
    if ((e = hipMalloc((void **)&vp->tile_class, 2 * ntiles)) != hipSuccess) return (int)e;
    if ((e = hipMalloc((void **)&vp->launch_list, sizeof(u32) * ntiles)) != hipSuccess) return (int)e;
    vp->ntiles_alloc = ntiles;
    unsigned char *any_all = vp->tile_class + ntiles;
    hipLaunchKernelGGL(prf_pack_vertical_kernel, dim3((u32)ntiles), dim3(64), 0, s, asc, vp->VH, vp->VL, vp->VX, any_all);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    hipLaunchKernelGGL(prf_tile_class_kernel, dim3((u32)((ntiles + 255) / 256)), dim3(256), 0, s, any_all, vp->tile_class, ntiles);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    // launch list (host side: one byte per 65536 positions)
    vp->h_class.resize(ntiles);
    if ((e = hipMemcpyAsync(vp->h_class.data(), vp->tile_class, ntiles, hipMemcpyDeviceToHost, s)) != hipSuccess) return (int)e;
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return (int)e;
    vp->h_list.clear();
    vp->h_list.reserve(ntiles);
    for (u64 t = 0; t + 1 < ntiles; t++) {  // the sentinel tile is never scanned
        if (vp->h_class[t] == 0) vp->h_list.push_back((u32)t);
        else if (vp->h_class[t] == 1) vp->h_list.push_back((u32)t | PRF_LAUNCH_MIXED);
    }
    vp->n_launch = (u32)vp->h_list.size();
    vp->flat_base = prf_flat_base(vp->h_list.data(), vp->h_list.size());
    if (!vp->h_list.empty()) {
        if ((e = hipMemcpyAsync(vp->launch_list, vp->h_list.data(), sizeof(u32) * vp->h_list.size(), hipMemcpyHostToDevice, s)) != hipSuccess)
            return (int)e;
        if ((e = hipStreamSynchronize(s)) != hipSuccess) return (int)e;
    }
    return 0;
}
