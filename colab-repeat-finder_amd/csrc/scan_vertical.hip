// scan_vertical.hip -- the fast path: fused bit-sliced ("vertical") scan + verification kernel for gfx950,
// the row gather that follows it and the ASCII -> bit-sliced packer (the work planner is host code: plan.cpp).
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
// One 256-thread workgroup per tile, SIX resident per CU (round 3; four before): <= 80 VGPRs and <= 27.3 KB of LDS.
// What made room: ONE 18 KB region R1 is the bit-sliced image while the tile is scanned and the window of the linear
// planes while its candidates are verified (before: two regions, 35 KB); the image arrives by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write pass) while the previous tile's rows are sorted, the
// window through the registers of the two waves that finish their tasks first; the exact tasks slide over the stream
// with a window of K + 4 rows in registers instead of all 60.
//  1. stage: the image arrived by DMA; 128 threads add the virtual lanes 64.. (one shift/or of two prefetched slots).
//  2. scan: every wave runs its share of the plan's tasks (host-built, balanced by cost).  A task answers one
//     question per (stream, motif size): "may a reportable run be found from this stream?" --
//      * exact task, one motif size k with M(k) = M <= 8 (compiled per (k, M)): mismatch word per row
//        (2 operations), sliding OR over exactly M rows; a row whose M successors all match and whose predecessor
//        does not is the start of a run of >= M.
//      * coarse task, one motif size with 9 <= M <= 14: the same on aligned groups of 2 or 4 rows (a run of >= M rows
//        holds floor((M + 1) / G) - 1 consecutive all-match groups).
//      * group task, 8 motif sizes k0..k0+7 with M(k) >= 15: a run of >= 15 matches contains an aligned
//        group of 8 rows that all match.  Per (group, k) the 8 rows of (H^H')|(L^L') are OR-ed with 16
//        v_bitop3_b32; motif sizes with M >= 23 / 39 examine only every 2nd / 4th group.
//     The answer is ONE 32-bit word per lane, task and motif size (bit b = stream b*64+lane).
//     A tile with not-ACGT positions in reach ("mixed") is scanned on the same two planes, where such a position reads
//     as A: that can only add matches, so with the flag rule relaxed to "M matching rows start in this stream" (exact)
//     and "an all-match group lies in this stream" (group) no stream that holds a row is missed; the streams that
//     consist of nothing but N are masked out (they cannot hold the start of a run).  False flags cost time only.
//  3. verify: R1 is refilled with the window of the linear planes H and L (tile - 128 .. tile + 65536 + 1536
//     positions); for every flagged (stream, k) the candidates are re-derived EXACTLY from the linear planes -- a
//     mixed tile reads the not-ACGT plane from global memory -- and each becomes a row or nothing.  A row belongs
//     to the tile that holds its first position: a run whose first examined group lies in the next tile is
//     reported by a look at the tile's end (boundary pass), and dropped by the next tile.
//  4. rows: sorted by (start, end) in LDS, written to the tile's slab as 8-byte rows.
// A second, small kernel (prf_vgather_kernel) concatenates the slabs in launch (= position) order and expands the
// rows to 24 bytes: the row array leaves the device sorted by (contig, start, end), which is what the reference's
// sorted() returns (:81).
// Exactness argument: DESIGN.md.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

#include "prf_host.h"
#include "scan_vertical.h"
#include "verify_impl.h"

namespace {

using prf_layout::T; using prf_layout::RG; using prf_layout::LIN_PRE; using prf_layout::LIN_POST; using prf_layout::LW;  // (prf_plan.h)
static_assert(prf_layout::TILE_WORDS == (int)PRF_TILE_WORDS, "tile size");
static_assert(LW % 2 == 0, "the window travels in 16-byte pieces");
constexpr u32 WIN_LEAD = 64u * (u32)LIN_PRE;                  // window positions in front of the tile
using prf_layout::REC_CAP; using prf_layout::FLAG_CAP; using prf_layout::SLOW_CAP;
constexpr int MAX_WAVES = PRF_VMAX_WAVES;
constexpr int NTH = 64 * MAX_WAVES;                           // threads per workgroup, always
using prf_layout::SMALL_M;
using prf_layout::ROW_CAP_LDS;
static_assert(ROW_CAP_LDS % 32 == 0, "the rank loop reads the padded key list 32 keys at a time");

template <int A, class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, A + I>{}), ...);
}
// f(integral_constant<int,i>) for i in [A, B)
template <int A, int B, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B > A) static_for_impl<A>(static_cast<F &&>(f), std::make_integer_sequence<int, B - A>{});
}

// ---- group-task candidate records: [5:0] lane, [14:6] k, [16:15] 1/2/3 = every 1st/2nd/4th group examined,
// [48:17] stream word (bit b = stream b*64 + lane may hold a candidate) ----
__device__ __forceinline__ u64 make_rec(u32 lane, u32 k, u32 sc, u32 word) {
    return (u64)(lane | (k << 6) | (sc << 15)) | ((u64)word << 17);
}

// dynamic LDS: [header][R1: image / window][recs][row keys][row motif sizes][all-N stream masks][flag lists, counts][boundary items][cofactors]
extern __shared__ __attribute__((aligned(16))) unsigned char prf_smem[];
using prf_layout::SMEM_HDR;
// header words: 128.. two sets of tile counters used alternately (a set is reset while the other one is still read)
constexpr int HDR_CNT = 128;       // [parity][8] u32
constexpr int HDR_NEXT = 192;      // {next launch slot, its entry}
constexpr int HDR_LONG = 200;      // [PRF_LONG_PER_TILE] u64: true ends of the rows whose span is clipped
constexpr int HDR_STATS = 232;     // {candidates looked at, of which verified on the spot} by this workgroup so far (thread 0's)
static_assert(HDR_LONG + 8 * (int)PRF_LONG_PER_TILE <= HDR_STATS && HDR_STATS + 8 <= SMEM_HDR, "LDS header layout");
constexpr u32 CNT_ROWS = 0, CNT_RECS = 1, CNT_EARLY = 2, CNT_LONG = 3, CNT_FLAGS = 4, CNT_SLOW = 5,
              CNT_ROWS0 = 6, CNT_LONG0 = 7;  // rows / long rows listed before the verification began (the scan's overflow paths)

// LDS is addressed through explicit address-space pointers everywhere: a generic pointer that the compiler cannot trace back
// to prf_smem becomes a flat_load, which is slower and waits on both memory counters.
typedef u32 prf_u32x4 __attribute__((ext_vector_type(4)));  // (HIP's uint4 class cannot be copied out of an explicit address space)
typedef __attribute__((address_space(3))) const prf_u32x4 prf_lds_cu4;
typedef __attribute__((address_space(3))) prf_u32x4 prf_lds_u4;
typedef __attribute__((address_space(3))) u64 prf_lds_u64;
typedef __attribute__((address_space(3))) u32 prf_lds_u32;
typedef __attribute__((address_space(3))) const u32 prf_lds_cu32;
typedef __attribute__((address_space(3))) void prf_lds_void;
typedef __attribute__((address_space(1))) const void prf_glb_cvoid;
typedef __attribute__((address_space(1))) const u32 prf_glb_cu32;
typedef __attribute__((address_space(1))) u64 prf_glb_u64;

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

// what the verification step needs about the tile; lives at the start of LDS
struct TileCtx {
    u64 w0;                   // first word of the linear window
    u64 xz_lo, xz_hi;         // positions known to hold no not-ACGT symbol
    const u64 *H, *L, *X;     // linear planes in HBM
    const u64 *const *E;      // device array of the five planes of the symbols outside ACGTN, or nullptr (prf_planes::E)
    prf_glb_u64 *slab;        // this tile's row slab in HBM (explicitly global: a generic pointer read back from LDS becomes FLAT stores)
    u64 tile_base;            // first position of the tile
    u32 slab_cap;
    u32 min_repeats, min_span;
    u32 lin_off;              // byte offset of R1 (the linear window, once staged) in LDS
    u32 has_lin;              // the linear window is staged (after the scan)
    u32 keys_off;             // byte offset of the row list (keys, then motif sizes) in LDS
    u32 cof_off;              // byte offset of the cofactor table in LDS
    u32 slow_off;             // byte offset of the list of deferred candidates in LDS
    u32 k_exact0;             // exact tasks of the plan: motif sizes k_exact0 ..., task index = k - k_exact0
    u32 cnt_off;              // byte offset of this tile's counter set in LDS
};
static_assert(sizeof(TileCtx) <= 128, "TileCtx must fit its LDS header slot");

// cof[k]: the cofactors k/p of the distinct primes p | k, one per byte, largest first (k <= 480 has at most 4
// distinct primes and k/p <= 240).  The motif seq[a:a+k] is primitive iff it has none of these periods
// (reference consists_of_perfect_repeats, utils/perfect_repeat_tracker.py:108-142, tries every divisor).
// Entries 0 .. kmax of the scan are copied to LDS once per workgroup: a table look, not a run-time division, per candidate.
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

// One row, as 8 bytes (scan_vertical.h): the first ROW_CAP_LDS of a tile into the LDS list, the others straight to the
// slab behind them (a tile that dense sorts them in R1 at its end).  Sort key: start in the tile (16 bits), then length
// clipped to 16 bits -- exact, because of the rows that share a start at most one is longer than two motif sizes (two
// periods on a long common stretch force their gcd, Fine and Wilf).  The true end of a clipped row goes to
// the tile's short list of long ends.
__device__ __forceinline__ void emit_row(const TileCtx &tc, u64 a, u64 b, u32 k) {
    const u64 span = b + k - a;
    prf_lds_u32 *cnt = (prf_lds_u32 *)(prf_smem + tc.cnt_off);
    u32 kv = k;
    if (span >= 65535ull) {
        const u32 j = atomicAdd((u32 *)(cnt + CNT_LONG), 1u);
        if (j < PRF_LONG_PER_TILE) {
            ((prf_lds_u64 *)(prf_smem + HDR_LONG))[j] = b + k;
            kv |= (j + 1u) << 16;
        }
    }
    const u32 key = ((u32)(a - tc.tile_base) << 16) | (span < 65535ull ? (u32)span : 65535u);
    const u32 i = atomicAdd((u32 *)(cnt + CNT_ROWS), 1u);
    if (i < (u32)ROW_CAP_LDS) {
        prf_lds_u32 *keys = (prf_lds_u32 *)(prf_smem + tc.keys_off);
        keys[i] = key;
        keys[ROW_CAP_LDS + i] = kv;
    } else if (i < tc.slab_cap) {
        tc.slab[i] = (u64)key | ((u64)kv << 32);
    }
}

// One 64-position look: mismatch bits (1 = differs, or either side is not ACGT) of positions q .. q+63 against q+k ..,
// served from the LDS window where it covers both sides, from the global planes elsewhere.  NOT inlined: the general
// routine is a few looks per candidate in divergent code, and forty inlined copies of the look were 90 KB of kernel (the
// instruction cache is shared by two CUs).
__device__ __noinline__ u64 tile_mismatch64(u64 q, u32 k) {
    const TileCtx &tc = *reinterpret_cast<const TileCtx *>(prf_smem);
    prf_window_view view;
    view.lds = (prf_lds_cu64 *)(prf_smem + tc.lin_off);
    view.w0 = tc.w0;
    view.nwords = tc.has_lin ? LW : 0;  // 0: every look goes to the global planes (while R1 holds the image)
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
__device__ __noinline__ void verify_stream(u64 sp, u32 k, u32 sc) {
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
            if (!motif_is_repeat(a, k)) emit_row(tc, a, b, k);
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
        if (!motif_is_repeat(a, k)) emit_row(tc, a, b, k);
    }
}

// ---- lean verification for the common case: a candidate whose looks stay inside the LDS window ----
// Window positions: bit 0 of the window = WIN_LEAD positions before the tile; the window holds H and L.  The not-ACGT plane is
// known to be zero there for a clean tile; a mixed tile reads it from global memory (L2), 32 bits at a time like the window.
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
// the same on the global not-ACGT plane: xw = the plane's 32-bit words from window position 0 on (wave-uniform), q per thread
__device__ __forceinline__ u32 xword(const u32 *xw, u32 w) {
    return *(prf_glb_cu32 *)(reinterpret_cast<const char *>(xw) + 4u * w);
}
__device__ __forceinline__ u32 xlook32(const u32 *xw, u32 q) {
    const u32 w = q >> 5;
    return __builtin_amdgcn_alignbit(xword(xw, w + 1), xword(xw, w), q & 31u);
}
__device__ __forceinline__ u64 xlook64(const u32 *xw, u32 q) {
    const u32 w = q >> 5, sft = q & 31u;
    const u32 w0 = xword(xw, w), w1 = xword(xw, w + 1), w2 = xword(xw, w + 2);
    return (u64)__builtin_amdgcn_alignbit(w1, w0, sft) | ((u64)__builtin_amdgcn_alignbit(w2, w1, sft) << 32);
}
struct WinCtx {
    prf_lds_cu32 *h, *l, *cof;
    const u32 *xw;  // mixed tile: the not-ACGT plane from window position 0 on (global memory); nullptr for clean tiles
    u64 win0;       // global position of window bit 0
    u32 min_repeats, min_span;
};

// mismatch bits of window positions q .. q+31 / q+63 against q+k ..; the caller guarantees q + k + 96 <= WIN_POS
__device__ __forceinline__ u32 win_mismatch32(const WinCtx &wc, u32 q, u32 k) {
    u32 r = (look32(wc.h, q) ^ look32(wc.h, q + k)) | (look32(wc.l, q) ^ look32(wc.l, q + k));
    if (wc.xw) r |= xlook32(wc.xw, q) | xlook32(wc.xw, q + k);
    return r;
}
__device__ __forceinline__ u64 win_mismatch64(const WinCtx &wc, u32 q, u32 k) {
    u64 r = (look64(wc.h, q) ^ look64(wc.h, q + k)) | (look64(wc.l, q) ^ look64(wc.l, q + k));
    if (wc.xw) r |= xlook64(wc.xw, q) | xlook64(wc.xw, q + k);
    return r;
}

__device__ __forceinline__ u32 min_matches32(u32 k, u32 min_repeats, u32 min_span) {
    const u32 a = (min_repeats - 1u) * k, b = min_span > k ? min_span - k : 0u;
    return a > b ? a : b;
}

// The verification loops below contain NO call: a call site in a loop makes the register allocator keep everything that is
// live around it in the 24 callee-saved registers a six-workgroup kernel has, or in scratch memory -- the first version of
// this kernel spilled the loops' own state that way (412 scratch operations per tile).  The few candidates that cannot be
// finished inside the LDS window (a run that reaches past it, the first stream of a clean tile, whose look-back lies in
// front of the tile) are put on a short list and finished by the general routine once the loops are over; a tile with more
// of them than the list holds is verified again from its flags and records by the general routine alone.
//   word 0: [39:0] start a (or the stream's first position), [48:40] k, [50:49] sc, [51] 1 = a whole stream (verify_stream),
//           [52] the primitive-motif test is still to be done;   word 1: position the run is known to match up to
__device__ __forceinline__ void defer(const TileCtx &tc, u64 a, u32 k, u32 sc, u32 whole_stream, u32 need_motif, u64 from) {
    const u32 i = atomicAdd((u32 *)((prf_lds_u32 *)(prf_smem + tc.cnt_off) + CNT_SLOW), 1u);
    if (i < (u32)SLOW_CAP) {
        prf_lds_u64 *slow = (prf_lds_u64 *)(prf_smem + tc.slow_off);
        slow[2u * i] = a | ((u64)k << 40) | ((u64)sc << 49) | ((u64)whole_stream << 51) | ((u64)need_motif << 52);
        slow[2u * i + 1u] = from;
    }
}

// one deferred candidate, by the general routine
__device__ __noinline__ void slow_item(u64 w0, u64 from) {
    const TileCtx &tc = *reinterpret_cast<const TileCtx *>(prf_smem);
    const u64 a = w0 & ((1ull << 40) - 1ull);
    const u32 k = (u32)(w0 >> 40) & 511u, sc = (u32)(w0 >> 49) & 3u;
    if ((w0 >> 51) & 1ull) {
        verify_stream(a, k, sc);
        return;
    }
    if (((w0 >> 52) & 1ull) && motif_is_repeat(a, k)) return;
    const u64 b = run_end(from, k);
    if ((long long)(b - a) < prf_min_matches(k, tc.min_repeats, tc.min_span)) return;
    emit_row(tc, a, b, k);
}

// end of the run at motif size k that matches up to window position `from`: true and the end (global position), or false
// and `from` = the window position at which the walk leaves the window
__device__ __forceinline__ bool win_run_end(const WinCtx &wc, u32 &from, u32 k, u64 &b) {
    for (;;) {
        if (from + k + 96u > WIN_POS) return false;
        const u64 m2 = win_mismatch64(wc, from, k);
        if (m2) {
            b = wc.win0 + from + (u64)__builtin_ctzll(m2);
            return true;
        }
        from += 64u;
    }
}

// motif [a, a+k) at window position a with a + 2 k + 96 <= WIN_POS: a power of a shorter word?  (see motif_is_repeat)
__device__ __forceinline__ bool win_motif_is_repeat(const WinCtx &wc, u32 a, u32 k) {
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

// One (stream, exact task) flag: stream (lane rl, bit `bit`), motif size k.  128 positions of both planes
// from the position in front of the stream are read; the mismatch word, the run starts, the run ends and the periods of the
// primitive-motif test are funnel shifts of those registers.
__device__ __forceinline__ void win_verify_flag(const TileCtx &tc, const WinCtx &wc, u32 rl, u32 bit, u32 k) {
    const u32 q = WIN_LEAD + (bit * 64u + rl) * T;  // window position of the stream's first position
    const u32 w = (q - 1u) >> 5, sft = (q - 1u) & 31u;
    const u32 a0 = wc.h[w], a1 = wc.h[w + 1], a2 = wc.h[w + 2], a3 = wc.h[w + 3], a4 = wc.h[w + 4];
    const u32 b0 = wc.l[w], b1 = wc.l[w + 1], b2 = wc.l[w + 2], b3 = wc.l[w + 3], b4 = wc.l[w + 4];
    // bit i = window position q - 1 + i
    const u64 hlo = (u64)__builtin_amdgcn_alignbit(a1, a0, sft) | ((u64)__builtin_amdgcn_alignbit(a2, a1, sft) << 32);
    const u64 hhi = (u64)__builtin_amdgcn_alignbit(a3, a2, sft) | ((u64)__builtin_amdgcn_alignbit(a4, a3, sft) << 32);
    const u64 llo = (u64)__builtin_amdgcn_alignbit(b1, b0, sft) | ((u64)__builtin_amdgcn_alignbit(b2, b1, sft) << 32);
    const u64 lhi = (u64)__builtin_amdgcn_alignbit(b3, b2, sft) | ((u64)__builtin_amdgcn_alignbit(b4, b3, sft) << 32);
    u64 xlo = 0, xhi = 0;
    if (wc.xw) {
        const u32 c0 = xword(wc.xw, w), c1 = xword(wc.xw, w + 1), c2 = xword(wc.xw, w + 2), c3 = xword(wc.xw, w + 3), c4 = xword(wc.xw, w + 4);
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
        if (after) {
            emit_row(tc, wc.win0 + a, wc.win0 + a + (u64)__builtin_ctzll(after), k);
        } else {
            u32 from = a + (64u - i);
            u64 b;
            if (win_run_end(wc, from, k, b)) emit_row(tc, wc.win0 + a, b, k);
            else defer(tc, wc.win0 + a, k, 0u, 0u, 0u, wc.win0 + from);
        }
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
        const bool ok = ((m >> gb) & 0xFFull) == 0 && nb < back && q - 32u + gb - nb >= WIN_LEAD;
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
        if (a + 2u * k + 96u > WIN_POS) {  // (the group itself is known to match)
            defer(tc, wc.win0 + a, k, 0u, 0u, 1u, wc.win0 + (q - 32u + gb + 8u));
            continue;
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
            if (m2) {
                b = wc.win0 + (q + 32u) + (u64)__builtin_ctzll(m2);
            } else {
                u32 from = q + 96u;
                if (!win_run_end(wc, from, k, b)) {
                    defer(tc, wc.win0 + a, k, 0u, 0u, 0u, wc.win0 + from);
                    continue;
                }
            }
        }
        if (b - (wc.win0 + a) < (u64)M) continue;
        emit_row(tc, wc.win0 + a, b, k);
    }
}

// Boundary pass.  A group task's run is found at the FIRST examined all-match group it contains.  For a run that starts
// in the last 8S-1 positions of this tile that group lies in the next tile, whose workgroup drops the run because it does
// not start there; this tile reports it: per motif size one look at the 32 positions in front of the
// tile's end.  c = matches directly in front of the end: 1 <= c < 8S <=> such a run exists and starts at end - c.
__device__ __forceinline__ void boundary_item(const TileCtx &tc, const WinCtx &wc, u32 k, u32 S) {
    const u64 tile_end = tc.tile_base + PRF_TILE;
    const u32 back = 8u * S;
    const u64 mm = win_mismatch64(wc, WIN_LEAD + PRF_TILE - 32u, k);
    const u32 lo = (u32)mm;  // bit i = mismatch at tile_end - 32 + i
    const u32 c = lo ? (u32)__builtin_clz(lo) : 32u;
    if (c == 0 || c >= back) return;
    const u64 a = tile_end - c;
    const u64 hi = mm >> 32;  // bit i = mismatch at tile_end + i
    u64 b;
    if (hi) {
        b = tile_end + (u64)__builtin_ctzll(hi);
    } else {
        u32 from = WIN_LEAD + PRF_TILE + 32u;
        if (!win_run_end(wc, from, k, b)) {
            defer(tc, a, k, 0u, 0u, 1u, wc.win0 + from);
            return;
        }
    }
    if (b - a < (u64)min_matches32(k, tc.min_repeats, tc.min_span)) return;
    static_assert(WIN_LEAD + PRF_TILE + 2u * PRF_VMAX_K + 96u <= WIN_POS, "the motif of a boundary item lies inside the window");
    if (!win_motif_is_repeat(wc, WIN_LEAD + PRF_TILE - c, k)) emit_row(tc, a, b, k);
}

// the same by the general routine (a tile that is verified again, see defer())
__device__ __forceinline__ void boundary_general(const TileCtx &tc, u32 k, u32 S) {
    const u64 tile_end = tc.tile_base + PRF_TILE;
    const u32 back = 8u * S;
    const u64 mm = tile_mismatch64(tile_end - 32, k);
    const u32 lo = (u32)mm;
    const u32 c = lo ? (u32)__builtin_clz(lo) : 32u;
    if (c == 0 || c >= back) return;
    const u64 a = tile_end - c;
    const u64 hi = mm >> 32;
    const u64 b = hi ? tile_end + (u64)__builtin_ctzll(hi) : run_end(tile_end + 32, k);
    if (b - a < (u64)min_matches32(k, tc.min_repeats, tc.min_span)) return;
    if (!motif_is_repeat(a, k)) emit_row(tc, a, b, k);
}

__device__ __forceinline__ prf_lds_u32 *smem_cnt(u32 parity) { return (prf_lds_u32 *)(prf_smem + HDR_CNT) + 8u * parity; }

// Candidates -> rows, all waves together once the window is staged.
//  * exact tasks left ONE ballot-compacted list of (stream, task) flags in LDS, dealt to the threads from thread 0 up;
//  * group-task records (one list) are taken by the upper two waves, alternately; the boundary items by the lower half, from
//    its last thread down.
__device__ __forceinline__ void verify_all(prf_lds_cu64 *recs, u32 n_recs, prf_lds_cu32 *bitems, u32 n_bitems, const unsigned short __attribute__((address_space(3))) *flags,
                                           u32 n_flags, const u32 *xw, u32 tid, u64 *dbg) {
#ifdef PRF_STAMPS
#define PRF_VSTAMP(i) do { if (dbg && (tid & 63u) == 0) dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PRF_VSTAMP(i) do { } while (0)
#endif
    const TileCtx &tc = *reinterpret_cast<const TileCtx *>(prf_smem);
    WinCtx wc;
    wc.h = (prf_lds_cu32 *)(prf_smem + tc.lin_off);
    wc.l = wc.h + 2 * LW;
    wc.xw = xw;
    wc.cof = (prf_lds_cu32 *)(prf_smem + tc.cof_off);
    wc.win0 = tc.tile_base - WIN_LEAD;
    wc.min_repeats = tc.min_repeats;
    wc.min_span = tc.min_span;
    // The two halves of the workgroup run different code side by side: a pass over the group-task records (upper half) takes
    // about as long as two passes over the flags plus one over the boundary items (lower half); a wave's pass costs the same
    // with 1 or 64 candidates.
    if (tid >= (u32)NTH / 2u) {
        // ---- group-task records, alternating between the two waves
        const u32 up = (u32)NTH - 1u - tid;  // 0 .. 127: thread 255, 254, ...
        for (u32 idx = 2u * (up & 63u) + (up >> 6); idx < n_recs; idx += (u32)NTH / 2u) {
            const u64 rec = recs[idx];
            const u32 rl = (u32)rec & 63u, k = ((u32)rec >> 6) & 511u, sc = ((u32)rec >> 15) & 3u;
            u32 word = (u32)(rec >> 17);
            while (word) {
                const u32 bit = (u32)__builtin_ctz(word);
                word &= word - 1;
                const u32 sq = (bit * 64u + rl) * T;
                if (sq >= 32u || xw) win_verify_group(tc, wc, WIN_LEAD + sq, k, 1u << (sc - 1u));
                else defer(tc, tc.tile_base, k, sc, 1u, 0u, 0ull);
            }
        }
    } else {
        // ---- exact tasks' flags: (lane, stream bit, task), dealt to the threads one by one: a wave runs the body once per 64
        // flags, not as often as its unluckiest lane has flags
        for (u32 idx = tid; idx < n_flags; idx += (u32)NTH / 2u) {
            const u32 f = flags[idx], frl = f & 63u, fbit = (f >> 6) & 31u, k = tc.k_exact0 + (f >> 11);
            // (a clean tile's first stream looks at positions in front of the tile, where N is possible and nothing says so
            // in the window: general routine, later)
            if ((frl | fbit) || xw) win_verify_flag(tc, wc, frl, fbit, k);
            else defer(tc, tc.tile_base, k, 0u, 1u, 0u, 0ull);
        }
        PRF_VSTAMP(14);
        // ---- boundary items: from the half's last thread down (the last round of flags fills it from the first thread up)
        for (u32 idx = (u32)NTH / 2u - 1u - tid; idx < n_bitems; idx += (u32)NTH / 2u) {
            const u32 it = bitems[idx];
            boundary_item(tc, wc, it & 0xFFFFu, it >> 16);
        }
    }
}

// A tile with more deferred candidates than their list holds: everything again, by the general routine alone (the rows the
// first attempt listed have been dropped by the caller).  Cold code: not inlined.
__device__ __noinline__ void verify_general(prf_lds_cu64 *recs, u32 n_recs, prf_lds_cu32 *bitems, u32 n_bitems,
                                            const unsigned short __attribute__((address_space(3))) *flags, u32 n_flags, u32 tid) {
    const TileCtx &tc = *reinterpret_cast<const TileCtx *>(prf_smem);
    for (u32 idx = tid; idx < n_flags; idx += (u32)NTH) {
        const u32 f = flags[idx], frl = f & 63u, fbit = (f >> 6) & 31u, k = tc.k_exact0 + (f >> 11);
        verify_stream(tc.tile_base + (u64)(fbit * 64u + frl) * T, k, 0u);
    }
    for (u32 idx = tid; idx < n_recs; idx += (u32)NTH) {
        const u64 rec = recs[idx];
        const u32 rl = (u32)rec & 63u, k = ((u32)rec >> 6) & 511u, sc = ((u32)rec >> 15) & 3u;
        u32 word = (u32)(rec >> 17);
        while (word) {
            const u32 bit = (u32)__builtin_ctz(word);
            word &= word - 1;
            verify_stream(tc.tile_base + (u64)(bit * 64u + rl) * T, k, sc);
        }
    }
    for (u32 idx = tid; idx < n_bitems; idx += (u32)NTH) {
        const u32 it = bitems[idx];
        boundary_general(tc, it & 0xFFFFu, it >> 16);
    }
}

// The tasks' answers -> LDS lists.  Every lane of the wave calls these together.
struct Emit {
    prf_lds_u64 *recs;       // the tile's record list in LDS, REC_CAP records
    prf_lds_u32 *cnt;        // this tile's counter set
    int lane;

    // Exact tasks: the lanes' words of ONE task -> flags (lane | stream bit << 6 | task << 11) appended to the tile's list.
    // ONE reservation per task (the stamps of the first version showed 2.7 k cycles per task in here against 2 k in the task
    // itself: an LDS atomic round trip per round of the loop): a first pass of ballots counts the flags level by level (level j =
    // the lanes with more than j flags), one atomic reserves them, a second pass places them -- level j behind the levels below
    // it, a lane's flag behind those of the lower lanes.  Flags beyond the list's capacity (a tile of long runs) are verified on
    // the spot with the general routine.
    __device__ __forceinline__ void push_flags(u32 word, u32 e, u32 k, prf_lds_u32 *flag_words) {
        typedef __attribute__((address_space(3))) unsigned short prf_lds_u16;
        prf_lds_u16 *list = (prf_lds_u16 *)flag_words;
        const u32 pc = (u32)__builtin_popcount(word);
        u32 total = 0;  // wave-uniform
        for (u32 j = 0;; j++) {
            const u64 bal = __builtin_amdgcn_ballot_w64(pc > j);
            if (bal == 0) break;
            total += (u32)__builtin_popcountll(bal);
        }
        if (total == 0) return;
        u32 base = 0;
        if (lane == 0) base = atomicAdd((u32 *)(cnt + CNT_FLAGS), total);
        base = (u32)__builtin_amdgcn_readfirstlane((int)base);
        for (;;) {
            const u64 bal = __builtin_amdgcn_ballot_w64(word != 0);
            if (bal == 0) break;
            if (word) {
                const u32 bit = (u32)__builtin_ctz(word);
                word &= word - 1;
                const u32 at = base + __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0));
                if (at < (u32)FLAG_CAP) {
                    list[at] = (unsigned short)((u32)lane | (bit << 6) | (e << 11));
                } else {
                    atomicAdd((u32 *)(cnt + CNT_EARLY), 1u);
                    verify_stream(reinterpret_cast<const TileCtx *>(prf_smem)->tile_base + (u64)(bit * 64u + (u32)lane) * T, k, 0u);
                }
            }
            base += (u32)__builtin_popcountll(bal);
        }
    }

    // Group tasks: one 32-bit word per lane (bit b = stream b*64 + lane is flagged for motif size k) -> one record per lane
    // with a non-zero word, appended to the tile's list (one LDS atomic per call); a full list -> verified on the spot.
    __device__ __forceinline__ void push_word(u32 word, u32 k, u32 sc) {
        const u64 bal = __builtin_amdgcn_ballot_w64(word != 0);
        if (bal == 0) return;
        const u32 n = (u32)__builtin_popcountll(bal);
        u32 base = 0;
        if (lane == (int)__builtin_ctzll(bal)) base = atomicAdd((u32 *)(cnt + CNT_RECS), n);
        base = (u32)__builtin_amdgcn_readlane((int)base, (int)__builtin_ctzll(bal));
        if (word) {
            const u32 idx = base + __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0));
            if (idx < (u32)REC_CAP) {
                recs[idx] = make_rec((u32)lane, k, sc, word);
            } else {
                atomicAdd((u32 *)(cnt + CNT_EARLY), 1u);
                const u64 tile_base = reinterpret_cast<const TileCtx *>(prf_smem)->tile_base;
                while (word) {
                    const u32 bit = (u32)__builtin_ctz(word);
                    word &= word - 1;
                    verify_stream(tile_base + (u64)(bit * 64u + (u32)lane) * T, k, sc);
                }
            }
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
// (stride 2 / 4, and every task of a mixed tile) every examined all-match group counts.  The per-size words are OR-ed over the
// blocks and leave as records at the end of the task.  The eight sizes are computed as two halves of four, the rows of the
// second half's last slot loaded in between: 40 row registers instead of 48, four OR chains interleaved.
// HALF: only the sizes k0 .. k0+3 (a task whose second half wants another stride, or lies beyond the largest motif size).
template <int NC, bool S1, bool HALF>
__device__ __forceinline__ void group_task(prf_lds_cu4 *vimg, int lane, u32 k0, u32 valid, u32 stride, u32 allow, Emit &em) {
    constexpr int PS = RG * NC;  // slots per plane
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));  // (the address is recomputed here: hoisted out of the task loop it was kept in scratch memory)
    prf_lds_cu4 *lane_base = vimg + lane_o;
    u32 prev[8], acc[8];
    static_for<0, 8>([&](auto ic) {
        prev[decltype(ic)::value] = ~0u;  // first group of a stream: counts, verification decides
        acc[decltype(ic)::value] = 0u;
    });
#pragma unroll 1
    for (int tb = 0; tb < 4; tb += (int)stride) {
        u32 a[2][8];   // rows 8tb .. 8tb+7
        u32 w[2][16];  // rows 8tb+k0 .. 8tb+k0+15 (k0 % 4 == 0: whole 16-byte slots)
        const int g0 = 2 * tb + (int)(k0 >> 2);
        const int wa = g0 & 7;
        prf_lds_cu4 *pa = lane_base + 2 * tb * NC;
        prf_lds_cu4 *pw0 = slot_of<NC>(lane_base, g0);
        static_for<0, 2>([&](auto pc) {
            constexpr int p = decltype(pc)::value;
            unpack4(&a[p][0], pa[p * PS]);
            unpack4(&a[p][4], pa[p * PS + NC]);
        });
        auto load_w = [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            prf_lds_cu4 *pw = slot_after<NC, g>(pw0, wa);
            static_for<0, 2>([&](auto pc) {
                constexpr int p = decltype(pc)::value;
                unpack4(&w[p][4 * g], pw[p * PS]);
            });
        };
        // the four motif sizes 4h .. 4h+3 in one straight-line block: their independent OR chains interleave
        auto sizes = [&](auto hc) {
            constexpr int h = decltype(hc)::value;
            static_for<4 * h, 4 * h + 4>([&](auto kc) {
                constexpr int kk = decltype(kc)::value;
                // OR over the 8 rows of (H^H')|(L^L'): 16 operations, no per-row mismatch word
                u32 o = a[0][0] ^ w[0][kk];
                o = or_xor(o, a[1][0], w[1][kk]);
                static_for<1, 8>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    o = or_xor(o, a[0][i], w[0][kk + i]);
                    o = or_xor(o, a[1][i], w[1][kk + i]);
                });
                if constexpr (S1) {
                    acc[kk] = bitop3<(TA | (~TB & TC)) & 0xFF>(acc[kk], o, prev[kk]);  // acc | (~o & prev)
                    prev[kk] = o;
                } else {
                    acc[kk] |= ~o;
                }
            });
        };
        load_w(std::integral_constant<int, 0>{});
        load_w(std::integral_constant<int, 1>{});
        load_w(std::integral_constant<int, 2>{});
        sizes(std::integral_constant<int, 0>{});
        if constexpr (!HALF) {
            load_w(std::integral_constant<int, 3>{});
            sizes(std::integral_constant<int, 1>{});
        }
    }
    const u32 sc = stride == 1 ? 1u : (stride == 2 ? 2u : 3u);
    static_for<0, (HALF ? 4 : 8)>([&](auto kc) {
        constexpr int kk = decltype(kc)::value;
        if ((valid >> kk) & 1u) em.push_word(acc[kk] & allow, k0 + (u32)kk, sc);  // wave-uniform condition
    });
}

// OR of the mismatch words of the M rows t .. t+M-1.  mm is indexed by row + 1 (mm[0] = the row before the stream),
// o3[i] = mm[i] | mm[i+1] | mm[i+2] (the rows i-1 .. i+1).
template <int M, int t, int LM, int LO>
__device__ __forceinline__ u32 window_or(const u32 (&mm)[LM], const u32 (&o3)[LO]) {
    constexpr int j = t + 1;  // first index
    if constexpr (M == 1) return mm[j];
    else if constexpr (M == 2) return mm[j] | mm[j + 1];
    else if constexpr (M == 3) return o3[j];
    else if constexpr (M <= 6) return o3[j] | o3[j + M - 3];
    else if constexpr (M <= 9) return or3(o3[j], o3[j + 3], o3[j + M - 3]);
    else if constexpr (M <= 12) return or3(o3[j], o3[j + 3], o3[j + 6]) | o3[j + M - 3];
    else return or3(or3(o3[j], o3[j + 3], o3[j + 6]), o3[j + 9], o3[j + M - 3]);
}

// ---- exact task: motif size K whose minimum run length is M < 15; the whole stream in one straight-line block ----
// Returns the lane's word: bit b set = stream (lane, b) holds a row t in 0..31 that starts a run of >= M matches:
// rows t .. t+M-1 all match and row t-1 does not, i.e. the window of M rows at t matches and the window at t-1 does not.
// The rows 0 .. 31+M-1+K of the extended stream are read ONCE, slot by slot (4 rows of both planes); a mismatch word is
// computed as soon as its partner row (K further on) is there, a window as soon as its last row is: the compiler sees
// straight-line code in that order and keeps only what is live -- K + 4 rows of two planes, M - 2 triple ORs, three
// mismatch words, the previous window -- under the 56 registers a function may use without saving any for its caller
// (round 2 read all 60 rows first: 128 VGPRs, four workgroups per CU).
// relax (mixed tile): a stream whose first M rows all match counts as well -- together with the starts that is "some M
// matching rows begin in this stream", which no added match (a not-ACGT position reads as A) can take away.
// Not inlined: one compact function per (K, M), called by the one wave that runs the task.
template <int K, int M, int NC>
__device__ __attribute__((noinline)) u32 exact_stream(prf_lds_cu4 *vimg, int lane, bool relax) {
    constexpr int PS = RG * NC;
    constexpr int NM = T + M - 1;            // mismatch words of rows 0 .. NM-1
    constexpr int NG = (NM + K + 3) / 4;     // 16-byte slots of rows read
    static_assert(4 * NG <= 2 * T, "an exact task reads its own lane and the next one");
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));  // (the address is recomputed here: hoisted out of the task loop it was kept in scratch memory)
    prf_lds_cu4 *lane_base = vimg + lane_o;
    u32 r0[4 * NG], r1[4 * NG];
    u32 mm[NM + 1];
    u32 o3[NM + 1];
    u32 hot = 0, prev = 0;  // prev: the window one row earlier
    auto load_slot = [&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if constexpr (g < NG) {
            prf_lds_cu4 *ps = slot_of<NC>(lane_base, g);
            const prf_u32x4 v0 = ps[0], v1 = ps[PS];
            r0[4 * g] = v0.x; r0[4 * g + 1] = v0.y; r0[4 * g + 2] = v0.z; r0[4 * g + 3] = v0.w;
            r1[4 * g] = v1.x; r1[4 * g + 1] = v1.y; r1[4 * g + 2] = v1.z; r1[4 * g + 3] = v1.w;
        }
    };
    load_slot(std::integral_constant<int, 0>{});
    static_for<0, NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        load_slot(std::integral_constant<int, g + 1>{});  // one slot ahead of the rows that are computed: its latency hides behind them
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 4>([&](auto jc) {
            constexpr int i = 4 * g + decltype(jc)::value - K;  // the mismatch row whose partner row has just arrived
            if constexpr (i == -1) {
                // Row -1 of stream (lane, b) is row T-1 of stream (lane-1, b); for lane 0 it is row T-1 of stream (63, b-1):
                // lane 63's word one bit up, with bit 0 (the previous tile's last stream) unknown -> "mismatch", verification
                // decides.
                const int pl = (lane + 63) & 63;
                prf_lds_cu4 *pp = vimg + pl + (RG - 1) * NC;
                u32 p0 = pp[0].w, p1 = pp[PS].w;
                if (lane == 0) {
                    p0 <<= 1;
                    p1 <<= 1;
                }
                mm[0] = or_xor(p0 ^ r0[K - 1], p1, r1[K - 1]);
                if (lane == 0) mm[0] |= 1u;
                if constexpr (M == 1) prev = mm[0];  // (the window of one row at t = -1)
            } else if constexpr (i >= 0 && i < NM) {
                mm[i + 1] = or_xor(r0[i] ^ r0[i + K], r1[i], r1[i + K]);
                if constexpr (M >= 3 && i >= 1) o3[i - 1] = or3(mm[i - 1], mm[i], mm[i + 1]);
                constexpr int t = i - (M - 1);  // the window whose last row this is
                if constexpr (t >= -1 && t < T) {
                    const u32 win = window_or<M, t>(mm, o3);
                    if constexpr (t >= 0) hot = bitop3<(TA | (~TB & TC)) & 0xFF>(hot, win, prev);  // hot | (~win & prev)
                    if constexpr (t == 0) {
                        if (relax) hot |= ~win;
                    }
                    prev = win;
                }
            }
        });
    });
    return hot;
}

// ---- the same question answered more coarsely for M >= 9: rows in aligned groups of G = 2 (M <= 10) or 4 ----
// A run of >= M matching rows holds C = floor((M + 1) / G) - 1 consecutive aligned groups of G rows that match throughout:
// the first of them, group j0 = ceil(a / G), follows a group that does not (it holds row a - 1).  So the stream is
// flagged if, for some j in 0 .. T/G, the groups j .. j+C-1 all match and group j-1 does not.  j = T/G -- the first group of the
// NEXT stream -- is included because the run's first row may be one of the last G - 1 rows of this stream (the next stream's
// lane flags itself for the same group: a false flag there, which costs a look and nothing else).  Two operations per row
// for the groups' ORs, two or three per group for window and flag: 106 - 135 operations per task instead of 200 - 250; the price
// is false flags where G C rows match by chance without M doing so (6 rows: 8 per tile and motif size on random sequence,
// 8 rows: 0.5) -- the verification re-derives the run starts exactly either way (win_verify_flag uses M itself).  M = 7 and 8
// would get groups of 2 with C = 3: those 8 false flags per tile and motif size (52 per tile on the default workload, a third
// pass over the flags for one wave) cost more than the 70 operations they save: they keep the exact form.
// relax (mixed tile): also "the groups 0 .. C-1 match", which with the rule above is "some C matching groups begin here".
template <int K, int M, int NC>
__device__ __attribute__((noinline)) u32 coarse_stream(prf_lds_cu4 *vimg, int lane, bool relax) {
    constexpr int PS = RG * NC;
    constexpr int G = M >= 11 ? 4 : 2, C = (M + 1) / G - 1;
    constexpr int NJ = T / G + 1;                   // windows j = 0 .. T/G
    constexpr int NGRP = NJ + C;                    // groups -1 .. T/G + C - 1, stored at index + 1
    constexpr int NR = T + G * C;                   // mismatch rows -G .. NR - 1
    constexpr int NG = (NR + K + 3) / 4;            // 16-byte slots of rows read
    static_assert(C >= 2 && 4 * NG <= 2 * T, "a coarse task reads its own lane and the next one");
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));  // (the address is recomputed here: hoisted out of the task loop it was kept in scratch memory)
    prf_lds_cu4 *lane_base = vimg + lane_o;
    // rows -4 .. -1: the last slot of the previous stream, (lane-1, b); for lane 0 that is stream (63, b-1): lane 63's words one
    // bit up, with bit 0 (the previous tile's last stream) unknown -> "mismatch", verification decides
    u32 q0[4], q1[4];
    {
        const int pl = (lane + 63) & 63;
        prf_lds_cu4 *pp = vimg + pl + (RG - 1) * NC;
        prf_u32x4 v0 = pp[0], v1 = pp[PS];
        if (lane == 0) {
            v0 <<= 1;
            v1 <<= 1;
        }
        q0[0] = v0.x; q0[1] = v0.y; q0[2] = v0.z; q0[3] = v0.w;
        q1[0] = v1.x; q1[1] = v1.y; q1[2] = v1.z; q1[3] = v1.w;
    }
    u32 r0[4 * NG], r1[4 * NG];
    u32 grp[NGRP];
    u32 hot = 0;
    auto load_slot = [&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if constexpr (g < NG) {
            prf_lds_cu4 *ps = slot_of<NC>(lane_base, g);
            const prf_u32x4 v0 = ps[0], v1 = ps[PS];
            r0[4 * g] = v0.x; r0[4 * g + 1] = v0.y; r0[4 * g + 2] = v0.z; r0[4 * g + 3] = v0.w;
            r1[4 * g] = v1.x; r1[4 * g + 1] = v1.y; r1[4 * g + 2] = v1.z; r1[4 * g + 3] = v1.w;
        }
    };
    // row r of plane p, r >= -4 (compile-time r)
    auto h = [&](auto rc) -> u32 { constexpr int r = decltype(rc)::value; if constexpr (r < 0) return q0[r + 4]; else return r0[r]; };
    auto l = [&](auto rc) -> u32 { constexpr int r = decltype(rc)::value; if constexpr (r < 0) return q1[r + 4]; else return r1[r]; };
    load_slot(std::integral_constant<int, 0>{});
    static_for<0, NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        load_slot(std::integral_constant<int, g + 1>{});  // one slot ahead of the rows that are computed
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 4>([&](auto jc) {
            constexpr int i = 4 * g + decltype(jc)::value - K;  // the mismatch row whose partner row has just arrived
            // the last row of group j = (i + 1) / G - 1 (groups -1 .. NGRP - 2): the whole group is there now
            if constexpr (i >= -1 && i < NR && (i + 1) % G == 0) {
                constexpr int j = (i + 1) / G - 1, first = G * j;
                u32 t = h(std::integral_constant<int, first>{}) ^ h(std::integral_constant<int, first + K>{});
                t = or_xor(t, l(std::integral_constant<int, first>{}), l(std::integral_constant<int, first + K>{}));
                static_for<1, G>([&](auto ic) {
                    constexpr int r = first + decltype(ic)::value;
                    t = or_xor(t, h(std::integral_constant<int, r>{}), h(std::integral_constant<int, r + K>{}));
                    t = or_xor(t, l(std::integral_constant<int, r>{}), l(std::integral_constant<int, r + K>{}));
                });
                if constexpr (j == -1) {
                    if (lane == 0) t |= 1u;
                }
                grp[j + 1] = t;
                constexpr int w = j - C + 1;  // the window whose last group this is
                if constexpr (w >= 0 && w < NJ) {
                    u32 win;
                    if constexpr (C == 2) win = grp[w + 1] | grp[w + 2];
                    else if constexpr (C == 3) win = or3(grp[w + 1], grp[w + 2], grp[w + 3]);
                    else if constexpr (C == 4) win = or3(grp[w + 1], grp[w + 2], grp[w + 3]) | grp[w + 4];
                    else win = or3(or3(grp[w + 1], grp[w + 2], grp[w + 3]), grp[w + 4], grp[w + 5]);
                    static_assert(C <= 5, "window of at most five groups");
                    hot = bitop3<(TA | (~TB & TC)) & 0xFF>(hot, win, grp[w]);  // hot | (~win & group w-1)
                    if constexpr (w == 0) {
                        if (relax) hot |= ~win;
                    }
                }
            }
        });
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
__device__ __forceinline__ u32 exact_dispatch(prf_lds_cu4 *vimg, int lane, bool relax, u32 v) {
    if constexpr (LO == HI) {
        constexpr int K = exact_variant_k(LO), M = K + (LO - exact_variant_of(K, K));
        static_assert(M >= K && M < SMALL_M && exact_variant_of(K, M) == LO, "variant numbering");
        if constexpr (M >= 9) return coarse_stream<K, M, NC>(vimg, lane, relax);  // (M = 7, 8: groups of 2 rows give 8 false flags per tile and size)
        else return exact_stream<K, M, NC>(vimg, lane, relax);
    } else {
        constexpr int MID = (LO + HI) / 2;
        if (v <= (u32)MID) return exact_dispatch<LO, MID, NC>(vimg, lane, relax, v);  // wave-uniform
        return exact_dispatch<MID + 1, HI, NC>(vimg, lane, relax, v);
    }
}

template <int NC>
__device__ __forceinline__ u32 exact_any(prf_lds_cu4 *vimg, int lane, bool relax, u32 k, u32 M) {
    // (min_repeats - 1) * k <= M < SMALL_M and min_repeats >= 2: k <= M
    const u32 v = (k - 1u) * (2u * (u32)SMALL_M - k) / 2u + (M - k);
    return exact_dispatch<0, exact_variants() - 1, NC>(vimg, lane, relax, v);
}

// relax: mixed tile (see the head of the file); allow = ~(streams of this lane that hold nothing but N), all ones on a clean tile
template <int NC>
__device__ __forceinline__ void run_tasks(prf_lds_cu4 *vimg, prf_lds_u32 *hotw, const prf_vplan &plan, int wave, int lane, bool relax, u32 allow,
                                          Emit &em, u64 *dbg) {
    const u32 t_end = plan.wave_begin[wave + 1];
    {   // (opaque: the exact tasks are functions, and a callee that knows the image's address as a constant looks the dynamic LDS
        // base up in a table in memory on every call -- handed over as an argument it is a register)
        u32 a = (u32)(__UINTPTR_TYPE__)vimg;
        asm volatile("" : "+s"(a));
        vimg = (prf_lds_cu4 *)(__UINTPTR_TYPE__)a;
    }
#ifdef PRF_STAMPS
    u64 t_call = 0;
#endif
    for (u32 ti = plan.wave_begin[wave]; ti < t_end; ti++) {
        // (the task as two dwords, decoded by hand: left to the compiler the one-byte fields came by vector loads from the kernel's
        // arguments -- a global-memory round trip, waited for on the spot, in front of every group task)
        static_assert(sizeof(prf_vtask) == 8 && alignof(prf_vtask) == 4, "a task is read as two dwords");
        const u32 *tw = reinterpret_cast<const u32 *>(&plan.tasks[ti]);
        const u32 tw0 = (u32)__builtin_amdgcn_readfirstlane((int)tw[0]), tw1 = (u32)__builtin_amdgcn_readfirstlane((int)tw[1]);
        prf_vtask task;
        task.k0 = (unsigned short)(tw0 & 0xFFFFu);
        task.kind = (unsigned char)((tw0 >> 16) & 0xFFu);
        task.valid = (unsigned char)(tw0 >> 24);
        task.stride = (unsigned char)(tw1 & 0xFFu);
        task.pad = 0;
        task.item0 = (unsigned short)(tw1 >> 16);
#ifdef PRF_STAMPS
        if (dbg && lane == 0 && ti - plan.wave_begin[wave] < 8u) dbg[8 + (ti - plan.wave_begin[wave])] = __builtin_amdgcn_s_memtime();
#endif
        if (task.kind == 0) {
            const bool half = (task.valid & 0xF0u) == 0;
            if (task.stride == 1 && !relax) {
                if (half) group_task<NC, true, true>(vimg, lane, task.k0, task.valid, 1u, allow, em);
                else group_task<NC, true, false>(vimg, lane, task.k0, task.valid, 1u, allow, em);
            } else {
                if (half) group_task<NC, false, true>(vimg, lane, task.k0, task.valid, task.stride, allow, em);
                else group_task<NC, false, false>(vimg, lane, task.k0, task.valid, task.stride, allow, em);
            }
        } else {
#ifdef PRF_STAMPS
            const u64 tc0 = __builtin_amdgcn_s_memtime();
            const u32 word = exact_any<NC>(vimg, lane, relax, task.k0, task.kind);
            asm volatile("" ::"v"(word));
            t_call += __builtin_amdgcn_s_memtime() - tc0;
            em.push_flags(word & allow, task.item0, task.k0, hotw);
#else
            em.push_flags(exact_any<NC>(vimg, lane, relax, task.k0, task.kind) & allow, task.item0, task.k0, hotw);
#endif
        }
    }
#ifdef PRF_STAMPS
    if (dbg && lane == 0) dbg[15] = t_call;
#endif
}

__device__ __forceinline__ void set_prio(u32 p) {  // (s_setprio takes an immediate; p is wave-uniform)
#ifdef PRF_NO_PRIO  // (diagnostic: what the priorities and the branches that select them cost)
    return;
#endif
    if (p == 0u) __builtin_amdgcn_s_setprio(0);
    else if (p == 1u) __builtin_amdgcn_s_setprio(1);
    else if (p == 2u) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
}

__device__ __forceinline__ u32 lds_add(prf_lds_u32 *p, u32 v) {
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// 16 bytes per lane, global memory -> LDS without a register in between (LDS-DMA): lane l's 16 bytes land at lds + 16 l
__device__ __forceinline__ void dma16(const void *src_lane, u32 lds_byte_off) {
    __builtin_amdgcn_global_load_lds((prf_glb_cvoid *)src_lane, (prf_lds_void *)(prf_smem + lds_byte_off), 16, 0, 0);
}

// Grid: persistent workgroups, as many as are resident at once (or one per entry of the launch list if that is fewer):
// workgroup b takes the launch slots b, then tickets of its XCD (tiles in position order; slabs and counts are indexed by
// slot, so the order of execution does not show in the output).
// What a tile needs from HBM arrives without waiting threads: the NEXT tile's bit-sliced image is sent to R1 by LDS-DMA
// when this tile's candidates are verified (R1 is dead then) and lands while the rows are sorted; the two slots per thread
// that become the virtual lanes 64.. are ordinary loads issued at the same point and held in 8 registers.  The window of the
// linear planes follows the scan by DMA: that wait is exposed to this workgroup and covered by the five others on the CU.
template <int NC>
struct NextRegs {
    static constexpr int extra = NC - 64;
    static constexpr u32 n_extra = (u32)(2 * RG * extra);  // virtual-lane slots of a tile: (plane, row group, first lanes again)
    static_assert(n_extra <= (u32)NTH, "one virtual-lane slot per thread");
    static_assert((RG * extra) % 64 == 0, "virtual-lane staging: one plane per wave");
    prf_u32x4 ev, en;
    u32 entry;  // the launch-list entry these registers belong to

    __device__ __forceinline__ static prf_u32x4 ld16(const void *ubase, u32 byte_off) {
        return *reinterpret_cast<const prf_u32x4 *>(reinterpret_cast<const char *>(ubase) + byte_off);
    }
    // Every address is a wave-uniform base (scalar registers) plus a 32-bit per-thread offset.
    __device__ __forceinline__ void load(const prf_vscan_args &g, u32 e, int tid, int wave, u32 r1_off) {
        e = (u32)__builtin_amdgcn_readfirstlane((int)e);
        entry = e;
        const u64 tile = e & ~PRF_LAUNCH_MIXED;
        const prf_u32x4 *ph = reinterpret_cast<const prf_u32x4 *>(g.VH) + tile * (RG * 64), *pL = reinterpret_cast<const prf_u32x4 *>(g.VL) + tile * (RG * 64);
        u32 ut = (u32)tid;
        asm volatile("" : "+v"(ut));  // (opaque: keeps the address arithmetic inside the loop)
        // the image: 2 planes x 8 row groups of 1 KiB, four pieces per wave; piece (p, rg) -> slots [(p RG + rg) NC, + 64)
        {
            const u32 p = (u32)wave >> 1, rg0 = ((u32)wave & 1u) * 4u;
            const prf_u32x4 *pp = p ? pL : ph;
            const u32 l16 = (ut & 63u) * 16u;
            static_for<0, 4>([&](auto ic) {
                constexpr u32 i = (u32)decltype(ic)::value;
                dma16(reinterpret_cast<const char *>(pp) + ((rg0 + i) * 64u * 16u + l16), r1_off + ((p * RG + rg0 + i) * (u32)NC) * 16u);
            });
        }
        const prf_u32x4 z = {0, 0, 0, 0};
        ev = z;
        en = z;
        if (ut < n_extra) {
            const u32 pw = (u32)__builtin_amdgcn_readfirstlane((int)(ut / (u32)(RG * extra)));
            const u32 erg = (ut / (u32)extra) % (u32)RG, el = ut % (u32)extra;
            const u32 i0 = (erg * 64u + el) * 16u, i1 = i0 + (u32)(RG * 64) * 16u;  // the same slot of the next tile
            if (pw == 0) {
                ev = ld16(ph, i0);
                en = ld16(ph, i1);
            } else {
                ev = ld16(pL, i0);
                en = ld16(pL, i1);
            }
        }
    }
    __device__ __forceinline__ void clear() {
        const prf_u32x4 z = {0, 0, 0, 0};
        ev = en = z;
        entry = 0;
    }
    // virtual lanes 64..: the first lanes again, one bit up, bit 31 from the next tile
    __device__ __forceinline__ void store(prf_lds_u4 *vimg, int tid) const {
        const u32 sx = (u32)tid;
        if (sx < n_extra) {
            const u32 p = sx / (u32)(RG * extra), erg = (sx / (u32)extra) % (u32)RG, el = sx % (u32)extra;
            vimg[p * RG * NC + erg * (u32)NC + 64u + el] = (ev >> 1) | (en << 31);
        }
    }
};

template <int NC>
__global__ __launch_bounds__(NTH, 6) void prf_vscan_kernel(prf_vscan_args g) {
    constexpr u32 R1_OFF = (u32)SMEM_HDR, R1_BYTES = (u32)(2 * RG * NC * 16);
    // (the deferred candidates' list is written while the candidates are verified, the mask of the all-N streams of a mixed tile
    // is read before its scan: they share 512 bytes)
    constexpr u32 RECS_OFF = R1_OFF + R1_BYTES, KEYS_OFF = RECS_OFF + (u32)REC_CAP * 8u, SLOW_OFF = KEYS_OFF + 2u * (u32)ROW_CAP_LDS * 4u,
                  HOTW_OFF = SLOW_OFF + (u32)SLOW_CAP * 16u, BITEMS_OFF = HOTW_OFF + (u32)FLAG_CAP * 2u;
    static_assert(SLOW_CAP * 16 >= 256, "the all-N stream masks lie in the deferred candidates' list");
    static_assert(2 * LW * 8 <= (int)R1_BYTES, "the window of the linear planes must fit the image's region");
    static_assert((2 * LW / 2 + 63) / 64 <= 5 * MAX_WAVES, "window pieces per wave");
    prf_lds_u4 *vimg = (prf_lds_u4 *)(prf_smem + R1_OFF);
    prf_lds_u64 *recs = (prf_lds_u64 *)(prf_smem + RECS_OFF);
    prf_lds_u32 *keys = (prf_lds_u32 *)(prf_smem + KEYS_OFF);
    prf_lds_u32 *nostart = (prf_lds_u32 *)(prf_smem + SLOW_OFF);
    prf_lds_u32 *hotw = (prf_lds_u32 *)(prf_smem + HOTW_OFF);      // exact tasks: the tile's list of (stream, task) flags, 2 bytes each
    prf_lds_u32 *bitems = (prf_lds_u32 *)(prf_smem + BITEMS_OFF);  // boundary items, plan.n_group_k of them

    const int tid0 = (int)threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    // (when the launch list is one contiguous range of clean tiles -- a contig without N blocks -- the tile index is
    // arithmetic: no dependent load in front of the staging loads)
    auto entry_of = [&](u32 sl) -> u32 { return g.flat_base != ~0u ? g.flat_base + sl : g.launch_list[sl]; };

    // ---- once per workgroup ----
    set_prio(g.plan.prio & 3u);
    NextRegs<NC> sr;
    sr.load(g, entry_of(blockIdx.x), tid0, wave, R1_OFF);
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
        tcw->lin_off = R1_OFF;
        tcw->keys_off = KEYS_OFF;
        tcw->slow_off = SLOW_OFF;
        tcw->k_exact0 = g.plan.k_exact0;
        tcw->cof_off = BITEMS_OFF + 4u * g.plan.n_group_k;
    }

    // Launch slots are handed out dynamically (tiles differ in cost by a factor of three; a fixed stride leaves the last
    // workgroups running alone for 8 % of the scan): XCD x -- the workgroups b = x mod 8 -- takes the slots = x mod 8, the
    // first one per workgroup by index, the following ones by a ticket counter of its own (one atomic per tile on eight
    // separate words; the ticket is drawn behind the tile's scan and needed at its end).
    const u32 xcd = blockIdx.x & 7u;
    const u32 first_ticket = (gridDim.x - xcd + 7u) >> 3;  // workgroups of this XCD = slots taken without a ticket
    u64 *ticket_word = g.counters + (PRF_CNT_SHARD0 + xcd * PRF_CNT_SHARD_STRIDE + PRF_SH_TILE_TICKET);
    prf_lds_u32 *next_words = (prf_lds_u32 *)(prf_smem + HDR_NEXT);  // {next slot, its launch-list entry}
    u32 slot_next = 0;
    u32 parity = 0;
    // (thread 0) candidates looked at by this workgroup, and those verified on the spot by the general routine because a list was
    // full: ONE atomic each when the workgroup ends.  Kept in LDS: as registers they lived in scratch memory across the tile
    // loop -- two scratch loads and two stores per tile in front of wave 0's verification.
    prf_lds_u32 *wg_stats = (prf_lds_u32 *)(prf_smem + HDR_STATS);
    if (tid0 == 0) wg_stats[0] = wg_stats[1] = 0;
    for (u32 slot = blockIdx.x; slot < g.n_launch; slot = slot_next, parity ^= 1u) {
    // (opaque per round: what derives from the thread index is recomputed, not carried through the scan's calls in
    // registers that would have to be spilled)
    int wave_s = wave;
    asm volatile("" : "+s"(wave_s));  // (opaque: recomputed from a scalar and the hardware's lane index, two operations)
    u32 ones = ~0u;
    asm volatile("" : "+s"(ones));  // (opaque too: the lane index is not to be computed once and kept in scratch memory either)
    int tid = wave_s * 64 + (int)__builtin_amdgcn_mbcnt_hi(ones, __builtin_amdgcn_mbcnt_lo(ones, 0u));
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const u32 entry = sr.entry;
    const bool hasx = (entry & PRF_LAUNCH_MIXED) != 0;
    const u64 tile = entry & ~PRF_LAUNCH_MIXED;
    prf_lds_u32 *cnt = smem_cnt(parity);

    PRF_STAMP(0);
#ifdef PRF_STAMPS
    if (g.dbg && lane == 0) g.dbg[((u64)slot * MAX_WAVES + wave) * 16 + 12] = __builtin_amdgcn_s_memrealtime();  // 100 MHz, chip-wide
#endif
    // ---- 1. stage: the image is in R1 (DMA issued during the previous tile's rows phase, or above); the virtual lanes from
    // the two prefetched slots; a mixed tile's mask of the streams that hold nothing but N ----
    set_prio(g.plan.prio & 3u);
    {
        sr.store(vimg, tid);
        if (hasx && tid < 64) {
            // bit b = stream (b, lane) is all N: its 32 positions are one aligned dword of the LINEAR not-ACGT plane (round 3: the
            // bit-sliced copy of that plane existed for this mask alone -- 0.125 B per position of HBM; mixed tiles are rare)
            const u32 *px = reinterpret_cast<const u32 *>(g.X + tile * PRF_TILE_WORDS);
            u32 m = 0;
#pragma unroll 1
            for (u32 b0 = 0; b0 < 32u; b0 += 8u) {  // (eight loads in flight: the registers of 32 would be spilled around this block)
                u32 v[8];
                static_for<0, 8>([&](auto bc) { v[decltype(bc)::value] = *(prf_glb_cu32 *)(px + ((b0 + (u32)decltype(bc)::value) * 64u + (u32)tid)); });
                static_for<0, 8>([&](auto bc) { m |= (v[decltype(bc)::value] == ~0u ? 1u : 0u) << (b0 + (u32)decltype(bc)::value); });
            }
            nostart[tid] = m;
        }
        if (tid < 8) cnt[tid] = 0;  // rows, records, flags, ... (the other set is still read by slow waves)
        if (tid == 0) {  // the tile's part of the context (the rest was written once, above)
            TileCtx *tcw = reinterpret_cast<TileCtx *>(prf_smem);
            tcw->w0 = tile * PRF_TILE_WORDS - LIN_PRE;
            tcw->xz_lo = hasx ? 0 : tile * PRF_TILE;  // a clean tile and its successor hold no not-ACGT position
            tcw->xz_hi = hasx ? 0 : (tile + 2) * PRF_TILE;
            tcw->slab = (prf_glb_u64 *)(g.slabs + (u64)slot * g.slab_cap);
            tcw->tile_base = tile * PRF_TILE;
            tcw->has_lin = 0u;
            tcw->cnt_off = (u32)HDR_CNT + 32u * parity;
        }
    }
    PRF_STAMP(1);
    __syncthreads();  // (waits for the image's DMA too)
    PRF_STAMP(2);

    // ---- 2. scan ----
    set_prio(((g.plan.slack_waves >> wave) & 1u) ? (g.plan.prio >> 4) & 3u : (g.plan.prio >> 2) & 3u);
    Emit em;
    em.recs = recs;
    em.cnt = cnt;
    em.lane = lane;
#ifdef PRF_STAMPS
    u64 *task_dbg = g.dbg ? g.dbg + ((u64)slot * MAX_WAVES + wave) * 16 : nullptr;
#else
    u64 *task_dbg = nullptr;
#endif
    {
        u32 allow = hasx ? ~nostart[lane] : ~0u;
        if (g.skip & 1u) allow = 0u;  // (diagnostic, PRF_SKIP: no flags, no records)
        run_tasks<NC>((prf_lds_cu4 *)vimg, hotw, g.plan, wave, lane, hasx, allow, em, task_dbg);
    }
    // ---- 3a. R1 <- the window of the linear planes H and L (tile - 128 .. tile + 65536 + 1536 positions), in 16-byte units: unit
    // u < LW/2 is H's word pair u, the others L's; piece = 64 units = 1 KiB.  R1 is the image until the LAST wave has finished its
    // tasks, so the window cannot be sent there earlier -- but the waves do not finish together (18.7 - 22.2 k cycles): the first
    // two to arrive load the window into registers (9 and 8 pieces: 36 / 32 VGPRs, free at this point) while they would
    // otherwise wait at the barrier, and store it behind the barrier.  The first version sent it by DMA behind the barrier: 2 - 4 k
    // cycles of HBM latency with every wave waiting.
    constexpr u32 WIN_UNITS = (u32)LW;  // 2 planes x LW / 2
    constexpr u32 WIN_PIECES = (WIN_UNITS + 63u) / 64u;
    static_assert(WIN_PIECES == 17, "the window's pieces are dealt 9 + 8 to the first two waves to arrive");
    prf_u32x4 wv[9];
    u32 order = 0;
    {
        if (lane == 0) order = atomicAdd((u32 *)(cnt + CNT_ROWS0), 1u);  // (the counter is free until the barrier: arrival order)
        order = (u32)__builtin_amdgcn_readfirstlane((int)order);
        const long long w0 = (long long)(tile * PRF_TILE_WORDS) - LIN_PRE;
        const u64 *wh = g.H + w0, *wl = g.L + w0;
        const prf_u32x4 z = {0, 0, 0, 0};
        static_for<0, 9>([&](auto ic) { wv[decltype(ic)::value] = z; });
        if (order < 2u) {
            const u32 first = order * 9u, n = order ? 8u : 9u;
            static_for<0, 9>([&](auto ic) {
                constexpr u32 i = (u32)decltype(ic)::value;
                if (i < n) {  // wave-uniform
                    const u32 u = (first + i) * 64u + (u32)lane;
                    const u64 *src = u < (u32)LW / 2u ? wh + 2u * u : wl + 2u * (u - (u32)LW / 2u);
                    if (u < WIN_UNITS) wv[i] = *reinterpret_cast<const prf_u32x4 *>(src);
                }
            });
        }
    }
    // The ticket for the tile after this one is drawn by the wave that leaves the scan first, behind its window loads: it waits for
    // those at the barrier below anyway, and the atomic's round trip is no longer than theirs.  (Drawn at the top of the tile -- the
    // first version -- the value was waited for at once all the same: a function waits for every outstanding memory operation on
    // entry, and the first task is a call; kept across the tasks' calls it lived in scratch memory.  Drawn behind the scan by a
    // fixed thread it was waited for on the spot as well, by a wave with work to do.)
    u64 ticket = 0;
    const bool ticket_thread = order == 0u && lane == 0;
    if (ticket_thread) ticket = atomicAdd(ticket_word, 1ull);
    PRF_STAMP(3);
    __syncthreads();  // the image is dead from here on
    {
        if (order < 2u) {
            const u32 first = order * 9u, n = order ? 8u : 9u;
            static_for<0, 9>([&](auto ic) {
                constexpr u32 i = (u32)decltype(ic)::value;
                if (i < n) {
                    const u32 u = (first + i) * 64u + (u32)lane;
                    if (u < WIN_UNITS) *(prf_lds_u4 *)(prf_smem + R1_OFF + 16u * u) = wv[i];
                }
            });
        }
        // the row list is padded to its capacity with the largest key: no bounds test per key when the rows are ranked
        // (rows that the scan's overflow paths have listed already stay; the verification appends behind them)
        const u32 n0 = cnt[CNT_ROWS];
        for (u32 i = (u32)tid; i < (u32)ROW_CAP_LDS; i += (u32)NTH)
            if (i >= n0) keys[i] = 0xFFFFFFFFu;
        if (tid == 0) {
            reinterpret_cast<TileCtx *>(prf_smem)->has_lin = 1u;
            cnt[CNT_ROWS0] = n0;
            cnt[CNT_LONG0] = cnt[CNT_LONG];
        }
    }
    // The next launch slot's entry (a load from the launch list unless the list is one run of clean tiles): issued here by the
    // thread that drew the ticket, looked at behind the verification.
    u32 slot_pre = 0, entry_pre = entry;
    if (ticket_thread) {
        slot_pre = (first_ticket + (u32)ticket) * 8u + xcd;
        if (slot_pre < g.n_launch) entry_pre = entry_of(slot_pre);  // (last round: this tile again, unused)
    }
    __syncthreads();
    PRF_STAMP(4);

    // ---- 3b. verify, all waves together: every candidate -> a row in the tile's list, or nothing ----
    // (the record waves are the critical path of this phase, the flag waves wait for them at the barrier below)
    set_prio(wave < MAX_WAVES / 2 ? (g.plan.prio >> 6) & 3u : (g.plan.prio >> 8) & 3u);
    const u32 *xw = hasx ? reinterpret_cast<const u32 *>(g.X + ((long long)(tile * PRF_TILE_WORDS) - LIN_PRE)) : nullptr;
    typedef const unsigned short __attribute__((address_space(3))) prf_lds_cu16;
    u32 n_recs = cnt[CNT_RECS], n_flags = cnt[CNT_FLAGS];
    {
        if (tid == 0) {
            // statistics: candidates looked at = (stream, exact task) flags + group-task records (+ those verified on the spot)
            const u32 early = cnt[CNT_EARLY];
            wg_stats[0] += n_flags + n_recs + early;
            wg_stats[1] += early;
        }
        n_recs = n_recs < (u32)REC_CAP ? n_recs : (u32)REC_CAP;
        n_flags = n_flags < (u32)FLAG_CAP ? n_flags : (u32)FLAG_CAP;
#ifdef PRF_STAMPS
        if (g.dbg && tid == 0) g.dbg[((u64)slot * MAX_WAVES + 0) * 16 + 11] = (u64)n_flags | ((u64)n_recs << 32);  // (wave 0 has three tasks)
#endif
        if (g.skip & 2u) n_flags = 0;   // (diagnostic) the flags are listed but not verified
        if (g.skip & 4u) n_recs = 0;    // (diagnostic) the same for the records
        verify_all((prf_lds_cu64 *)recs, n_recs, (prf_lds_cu32 *)bitems, (g.skip & 8u) ? 0u : g.plan.n_group_k, (prf_lds_cu16 *)hotw, n_flags, xw, (u32)tid, task_dbg);
    }
    set_prio((g.plan.prio >> 10) & 3u);
    if (ticket_thread) {  // the next slot and its entry, for everybody behind the barrier
        next_words[0] = slot_pre;
        next_words[1] = entry_pre;
    }
    PRF_STAMP(5);
    __syncthreads();
    {
        // the candidates that could not be finished inside the window (rare: wave-uniform, read behind the barrier)
        const u32 n_slow = (u32)__builtin_amdgcn_readfirstlane((int)cnt[CNT_SLOW]);
        if (n_slow) {
            if (n_slow <= (u32)SLOW_CAP) {
                if ((u32)tid < n_slow) {
                    prf_lds_u64 *slow = (prf_lds_u64 *)(prf_smem + SLOW_OFF);
                    slow_item(slow[2u * (u32)tid], slow[2u * (u32)tid + 1u]);
                }
            } else {
                // more than the list holds: the tile's rows so far are dropped, the general routine does everything again
                if (tid == 0) {
                    cnt[CNT_ROWS] = cnt[CNT_ROWS0];
                    cnt[CNT_LONG] = cnt[CNT_LONG0];
                }
                __syncthreads();
                verify_general((prf_lds_cu64 *)recs, n_recs, (prf_lds_cu32 *)bitems, g.plan.n_group_k, (prf_lds_cu16 *)hotw, n_flags, (u32)tid);
            }
            __syncthreads();
        }
    }
    // the window is dead from here on
    PRF_STAMP(6);
    const u64 nw = *(prf_lds_u64 *)next_words;  // (one read)
    slot_next = (u32)__builtin_amdgcn_readfirstlane((int)(u32)nw);
    const u32 entry_next = (u32)(nw >> 32);
    const u32 n_rows = (g.skip & 16u) ? 0u : (u32)__builtin_amdgcn_readfirstlane((int)cnt[CNT_ROWS]);  // (diagnostic: rows counted as none)
    const u32 n_long = (u32)__builtin_amdgcn_readfirstlane((int)cnt[CNT_LONG]);
    u64 *slab = g.slabs + (u64)slot * g.slab_cap;
    const u32 n_store = n_rows < g.slab_cap ? n_rows : g.slab_cap;
    bool unsorted = false;
    // The tile's row count, and its share of the sum the gather kernel starts from, leave NOW: a device-scope atomic stays
    // outstanding for ~3 k cycles when every CU issues them, and the tile's last barrier waits for it.  Issued behind the
    // ranking (first version) that wait was exposed: 7 % of the scan on the default workload, 27 % on random sequence
    // (PRF_SKIP=32, profiles/r03_notes.md); here it hides behind the rows phase.  ONE atomic per tile (the second level of sums
    // is gone: the gather adds the first level up itself).
    if (tid == 0) {
        g.slab_count[slot] = n_rows;
        if (n_store && !(g.skip & 32u)) atomicAdd(&g.block_sum[slot >> g.gather_shift], n_store);
    }

    // ---- 4. the tile's rows, sorted by (start, end), into its slab.  Rank of a row = number of rows with a smaller key; keys
    // are distinct ((start, end) pairs never collide between motif sizes).
    if (n_rows > (u32)ROW_CAP_LDS) {
        // A dense tile (the reference's golden chr22 BED has tiles of 750 rows): the rows behind the list went to the slab as
        // they came.  All of them are collected in R1 -- the window is dead -- and ranked there.
        constexpr u32 R1_ROWS = R1_BYTES / 8u;
        static_assert(R1_ROWS % 32u == 0, "padding of the dense-tile key list");
        if (n_store <= R1_ROWS) {
            prf_lds_u32 *k2 = (prf_lds_u32 *)(prf_smem + R1_OFF), *v2 = k2 + R1_ROWS;
            for (u32 i = (u32)tid; i < (u32)ROW_CAP_LDS; i += (u32)NTH) {
                k2[i] = keys[i];
                v2[i] = keys[ROW_CAP_LDS + i];
            }
            for (u32 i = (u32)ROW_CAP_LDS + (u32)tid; i < n_store; i += (u32)NTH) {
                const u64 r = slab[i];
                k2[i] = (u32)r;
                v2[i] = (u32)(r >> 32);
            }
            for (u32 i = n_store + (u32)tid; i < ((n_store + 31u) & ~31u); i += (u32)NTH) k2[i] = 0xFFFFFFFFu;
            __syncthreads();
            typedef __attribute__((address_space(3))) const prf_u32x4 prf_lds_ckey4;
            prf_lds_ckey4 *k4 = (prf_lds_ckey4 *)k2;
            for (u32 r = (u32)tid; r < n_store; r += (u32)NTH) {
                const u32 mine = k2[r], kv = v2[r];
                u32 rank = 0;
                for (u32 c0 = 0; 4u * c0 < n_store; c0 += 8u) {
#pragma unroll
                    for (u32 j = 0; j < 8u; j++) {
                        const prf_u32x4 v = k4[c0 + j];
                        rank += (v.x < mine ? 1u : 0u) + (v.y < mine ? 1u : 0u) + (v.z < mine ? 1u : 0u) + (v.w < mine ? 1u : 0u);
                    }
                }
                slab[rank] = (u64)mine | ((u64)kv << 32);
            }
            __syncthreads();  // R1 is refilled below
        } else {
            // more rows than R1 holds (no genome does that): the list goes to the slab as it is, the host sorts
            for (u32 i = (u32)tid; i < (u32)ROW_CAP_LDS && i < n_store; i += (u32)NTH) slab[i] = (u64)keys[i] | ((u64)keys[ROW_CAP_LDS + i] << 32);
            unsorted = true;
        }
    }

    // the next tile's staging data: the image's DMA and the virtual lanes' loads are issued now and land while the rows are ranked
    // (every register is written on both paths: dead from the stage to here, not carried around the loop)
    if (slot_next < g.n_launch) sr.load(g, entry_next, tid, wave, R1_OFF);
    else sr.clear();

    if (n_rows && n_rows <= (u32)ROW_CAP_LDS) {
        // P = 256 / n threads per row (a power of two, adjacent lanes): each counts the smaller keys of its share of the
        // list, the shares are added up across the P lanes.  Dependent LDS round trips are what this phase costs: the row's own
        // key and motif size and the first 32 keys of the lane's share are ONE batch of reads; the shares are added with DPP
        // moves, not LDS shuffles.  More than 256 rows: two passes.
        const u32 lg = n_rows > 128u ? 0u : (n_rows > 64u ? 1u : (n_rows > 32u ? 2u : 3u));
        const u32 P = 1u << lg, part = (u32)tid & (P - 1u);
        typedef __attribute__((address_space(3))) const prf_u32x4 prf_lds_ckey4;
        prf_lds_ckey4 *k4 = (prf_lds_ckey4 *)keys;
        for (u32 row0 = 0; row0 < n_rows; row0 += (u32)NTH >> lg) {
            const u32 row = row0 + ((u32)tid >> lg);
            const u32 r = row < n_rows ? row : 0u;  // (lanes without a row read row 0: harmless)
            const u32 mine = keys[r];
            const u32 kv = keys[ROW_CAP_LDS + r];
            u32 rank = 0;
            // part p takes the keys 4 p .. 4 p + 3, then 4 P further on, ...: one 16-byte read per four keys, eight reads in flight
            for (u32 c0 = part; 4u * c0 < n_rows; c0 += 8u * P) {
                prf_u32x4 v[8];
#pragma unroll
                for (u32 j = 0; j < 8u; j++) v[j] = k4[c0 + j * P];
                // (all eight reads in flight before the first compare: left alone the compiler issues them two at a time)
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
            if (row < n_rows && part == 0 && rank < g.slab_cap && !(g.skip & 64u)) slab[rank] = (u64)mine | ((u64)kv << 32);
        }
    }
    if ((u32)tid < n_long && (u32)tid < PRF_LONG_PER_TILE)
        g.long_ends[(u64)slot * PRF_LONG_PER_TILE + (u32)tid] = ((prf_lds_u64 *)(prf_smem + HDR_LONG))[tid];
    if (tid == 0) {
        if (n_rows > g.slab_cap) atomicMax(&g.counters[PRF_CNT_HIT_OVF], (u64)n_rows);
        if (unsorted) atomicMax(&g.counters[PRF_CNT_UNSORTED], 1ull);
        if (n_long > PRF_LONG_PER_TILE) atomicMax(&g.counters[PRF_CNT_LONG_OVF], (u64)n_long);
    }
    PRF_STAMP(7);
#ifdef PRF_STAMPS
    if (g.dbg && lane == 0) g.dbg[((u64)slot * MAX_WAVES + wave) * 16 + 13] = __builtin_amdgcn_s_memrealtime();
#endif
    // (no barrier here: the next round's first barrier separates this tile's reads of the row list from the next tile's writes)
    }
    if (tid0 == 0) {
        const u64 cand_total = wg_stats[0], early_total = wg_stats[1];
        if (cand_total) atomicAdd(&g.counters[PRF_CNT_SHARD0 + (blockIdx.x % PRF_CNT_NSHARD) * PRF_CNT_SHARD_STRIDE + PRF_SH_CAND], cand_total);
        if (early_total) atomicAdd(&g.counters[PRF_CNT_SHARD0 + (blockIdx.x % PRF_CNT_NSHARD) * PRF_CNT_SHARD_STRIDE + PRF_SH_EARLY], early_total);
    }
}

// ---------------------------------------------------------------------------------------------------
// Row gather: the slabs (8-byte rows, sorted per tile), in launch (= position) order, become ONE compact array of 24-byte
// rows.  Workgroup w owns the launch slots [w << shift, (w + 1) << shift): the rows in front of them are sums the scan kernel
// has added up (two atomics per tile: per gather workgroup and per 64 of them); it scans its own counts and writes its rows
// word by word -- three threads decode a row, each stores one of its words: coalesced 8-byte stores, four loads in flight.
// The workgroup that finishes last hands the counter block to the host (mapped memory, no copy call), and clears the sums
// and the counter block of the next scan (no memset call).
__global__ __launch_bounds__(256) void prf_vgather_kernel(prf_vgather_args g) {
    __shared__ u64 part[4];
    __shared__ u32 offs[PRF_GATHER_SLOTS_MAX + 1];   // in rows
    __shared__ u64 tbase[PRF_GATHER_SLOTS_MAX];      // first position of the slot's tile
    __shared__ u64 cbase[PRF_GATHER_SLOTS_MAX];      // first position of its contig
    __shared__ u32 contig[PRF_GATHER_SLOTS_MAX];
    __shared__ u64 ticket_lds;
    __shared__ u64 stage[2 * 3 * 256];
    __shared__ u32 fix_n;                                               // rows whose span is clipped: their true ends are filled in
    __shared__ u64 fix[PRF_GATHER_SLOTS_MAX * PRF_LONG_PER_TILE];       // behind the copy (row | slot << 32 | index of the end << 40)
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const u32 n_slots = 1u << g.gather_shift;  // launch slots per workgroup: 8 (small launches: more workgroups) .. 64
    const u32 first = blockIdx.x << g.gather_shift;
    const u32 my_super = blockIdx.x / PRF_GATHER_SUPER;
    // The loads of the prologue are issued together: the slots' counts and tiles (-> tile table), then the sums of the workgroups
    // in front of this one.  (One after the other they were five to six L2 round trips before the first row moved.)
    const bool live = tid < n_slots && first + tid < g.n_launch;  // (n_slots <= 64: the first wave)
    u32 c = 0;
    uint4 ti = make_uint4(0, 0, 0, 0);
    u64 tile = 0;
    if (live) {
        c = g.slab_count[first + tid];
        tile = (g.flat_base != ~0u ? g.flat_base + first + tid : g.launch_list[first + tid]) & ~PRF_LAUNCH_MIXED;
        ti = g.tile_info[tile];
    }
    u64 before = 0;  // rows in front of this workgroup's slots: the sums of the workgroups before it (one atomic per tile in the scan)
    for (u32 i = tid; i < blockIdx.x; i += 1024u) {  // (four independent loads per pass)
        u32 v[4];
#pragma unroll
        for (u32 u = 0; u < 4u; u++) v[u] = i + 256u * u < blockIdx.x ? g.block_sum[i + 256u * u] : 0u;
        before += (u64)v[0] + v[1] + v[2] + v[3];
    }
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
    if (lane == 0) part[wave] = before;
    if (tid == 0) fix_n = 0;
    if (tid < 64u) {  // exclusive scan of the counts, the slots' tiles and contigs
        c = c < g.slab_cap ? c : g.slab_cap;
        u32 incl = c;
        for (int o = 1; o < 64; o <<= 1) {
            const u32 up = __shfl_up(incl, o, 64);
            if ((int)tid >= o) incl += up;
        }
        offs[tid + 1] = incl;
        if (tid == 0) offs[0] = 0;
        if (live) {
            tbase[tid] = tile * PRF_TILE;
            cbase[tid] = (u64)ti.z | ((u64)ti.w << 32);
            contig[tid] = ti.x;
        }
    }
    __syncthreads();
    const u64 base0 = part[0] + part[1] + part[2] + part[3];  // rows in front of this workgroup's slots
    const u32 n_mine = offs[n_slots];
    // rows beyond the capacity stay behind: the host sees the total beyond the capacity, grows the array, rescans
    const u64 room_rows = base0 < g.rows_cap ? g.rows_cap - base0 : 0;
    const u32 n_copy = (u64)n_mine < room_rows ? n_mine : (u32)room_rows;  // rows
    u64 *dst = reinterpret_cast<u64 *>(g.rows + base0);
    // 256 rows per round: thread t decodes row r0 + t into three words in LDS, then the 768 words leave as coalesced stores; two
    // staging buffers used alternately, one barrier per round.  The slab rows of EIGHT rounds are fetched (slot search + load) in
    // one batch in front of them.  gfx950 counts loads and stores on ONE counter, in issue order (MI355X_MICROARCH.md), so a load's
    // data waits for every store issued before it -- and the compiler, once loads and stores are both in flight, waits for all of
    // them (vmcnt(0)): with a fetch per round every round ended with a full write round trip.  Now a workgroup waits for memory
    // once per eight rounds.  It bought 2 us of 49 on the default workload and costs 7 of 54 on ONE 10 Gbp sequence (a
    // workgroup with two rounds of rows still searches for eight): the kernel moves 183 MB in its 47 us, of which ~22 us do
    // not depend on the row count (profiles/r03_notes.md 2b).  (The rounds' barrier orders LDS only: s_waitcnt lgkmcnt(0) +
    // s_barrier -- which is also all that __syncthreads() is on this target.)
    constexpr u32 DEPTH = 8;
    u32 buf = 0;
    for (u32 R = 0; R < n_copy; R += DEPTH * 256u) {  // (n_copy is uniform: every thread takes the same barriers)
        u64 sr[DEPTH];
        u32 los[DEPTH];
        static_for<0, (int)DEPTH>([&](auto jc) {
            constexpr u32 j = (u32)decltype(jc)::value;
            const u32 row = R + j * 256u + tid;
            u32 lo = 0, hi = n_slots;  // the slot that holds the row: offs[lo] <= row < offs[lo + 1]
            while (hi - lo > 1) {
                const u32 mid = (lo + hi) >> 1;
                if (offs[mid] <= row) lo = mid; else hi = mid;
            }
            los[j] = lo;
            sr[j] = row < n_copy ? g.slabs[(u64)(first + lo) * g.slab_cap + (row - offs[lo])] : 0ull;
        });
        // (the ONE wait for memory of the eight rounds, outside their divergent blocks: a wait inside a block that a wave may skip
        // does not count behind it, and the compiler would wait again -- for every store issued since -- in each round)
        static_for<0, (int)DEPTH>([&](auto jc) {
            u64 &x = sr[decltype(jc)::value];
            asm volatile("" : "+v"(x));
        });
        static_for<0, (int)DEPTH>([&](auto jc) {
            constexpr u32 j = (u32)decltype(jc)::value;
            const u32 r0 = R + j * 256u;
            if (r0 < n_copy) {
                u64 *st = stage + buf * 768u;
                if (r0 + tid < n_copy) {
                    const u32 lo = los[j];
                    const u32 key = (u32)sr[j], kv = (u32)(sr[j] >> 32);
                    const u64 start = tbase[lo] + (key >> 16);
                    const u32 li = kv >> 16;  // 1 + index of the true end of a row whose span is clipped (at most PRF_LONG_PER_TILE per
                    // tile): listed, and filled in behind the copy -- a load in here, however rare, makes every round wait for memory
                    if (li) fix[atomicAdd(&fix_n, 1u)] = (u64)(r0 + tid) | ((u64)lo << 32) | ((u64)(li - 1u) << 40);
                    st[3u * tid] = start - cbase[lo];
                    st[3u * tid + 1u] = start + (key & 0xFFFFu) - cbase[lo];
                    st[3u * tid + 2u] = (u64)(kv & 0xFFFFu) | ((u64)contig[lo] << 32);
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                const u32 n_words = 3u * (n_copy - r0 < 256u ? n_copy - r0 : 256u);
                // 16-byte stores (8-byte ones run at 0.5 - 0.7 of their rate): the round's first word alone if it sits on an odd
                // 8-byte boundary, pairs from there on, the last word alone if one is left over
                u64 *d = dst + 3ull * r0;
                const u32 head = (u32)((reinterpret_cast<uintptr_t>(d) >> 3) & 1u);
                if (tid == 0 && head) d[0] = st[0];
                for (u32 p = tid; head + 2u * p + 1u < n_words; p += 256u) {
                    const u32 w = head + 2u * p;
                    ulonglong2 v;
                    v.x = st[w];
                    v.y = st[w + 1u];
                    *reinterpret_cast<ulonglong2 *>(d + w) = v;
                }
                if (tid == 1 && ((n_words - head) & 1u)) d[n_words - 1u] = st[n_words - 1u];
                buf ^= 1u;
            }
        });
    }
    // the workgroup of the last slots knows the total
    if (blockIdx.x == gridDim.x - 1 && tid == 0) {
        atomicAdd(&g.counters[PRF_CNT_ROWS], base0 + n_mine);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // performed before this thread draws the finishing ticket below
    }
    __syncthreads();
    if (fix_n) {  // (uniform) the true ends of the clipped rows, over the clipped ones the rounds have stored
        // (a barrier does not wait for stores on this target: every wave waits for its own, then they meet)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (u32 i = tid; i < fix_n; i += 256u) {
            const u64 e = fix[i];
            const u32 row = (u32)e, lo = (u32)(e >> 32) & 255u, li = (u32)(e >> 40);
            dst[3ull * row + 1u] = g.long_ends[(u64)(first + lo) * PRF_LONG_PER_TILE + li] - cbase[lo];
        }
        __syncthreads();
    }
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
    // by device-scope atomics, performed at the coherence point, and the one this kernel adds (the row total) has been waited
    // for by the thread that draws its workgroup's ticket, so it precedes the last ticket.  Every
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
                                                               u32 *__restrict__ VL,
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
#pragma unroll
    for (int rg = 0; rg < RG; rg++) {
        oh[rg * 64] = make_uint4(h[4 * rg], h[4 * rg + 1], h[4 * rg + 2], h[4 * rg + 3]);
        ol[rg * 64] = make_uint4(l[4 * rg], l[4 * rg + 1], l[4 * rg + 2], l[4 * rg + 3]);
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

// persistent workgroups: as many as are resident at once (LDS- and register-bound: 6 per CU at most)
static u32 resident_per_cu(u32 nc, u32 lds) {
    int n = 0;
    hipError_t e = nc == 72 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, prf_vscan_kernel<72>, NTH, lds)
                            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, prf_vscan_kernel<80>, NTH, lds);
    if (e != hipSuccess || n < 1) {
        (void)hipGetLastError();
        n = (int)std::max(1u, (160u * 1024u) / std::max(1u, lds + 1280u));
    }
    // (MI355X_MICROARCH.md: 256-thread workgroups are admitted up to floor(800 / (ceil(sgpr / 16) * 16 + 16)) per CU, which the
    // API overstates by one for 81 .. 112 SGPRs; the kernel is built for six.  A grid larger than what is resident is harmless
    // here -- no workgroup ever waits for another -- it only makes the launch slots taken "by index" start late.)
    return (u32)std::min(n, 6);
}

hipError_t prf_vertical_launch(hipStream_t s, const prf_vscan_args &args) {
    if (args.n_launch == 0) return hipSuccess;
    // PRF_LDS_PAD (diagnostic): extra dynamic LDS per workgroup, to measure the scan at a lower occupancy
    static const u32 lds_pad = getenv("PRF_LDS_PAD") ? (u32)atoi(getenv("PRF_LDS_PAD")) : 0u;
    const u32 lds = args.plan.lds_bytes + lds_pad;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidDevice;
        n_cu = prop.multiProcessorCount;
    }
    static u32 cache_lds = ~0u, cache_nc = 0, cache_per_cu = 0;
    if (cache_lds != lds || cache_nc != args.plan.nc) {
        cache_per_cu = resident_per_cu(args.plan.nc, lds);
        cache_lds = lds;
        cache_nc = args.plan.nc;
        if (getenv("PRF_DEBUG")) fprintf(stderr, "[prf] fused kernel: nc %u, %u bytes of LDS, %u workgroups per CU\n", args.plan.nc, lds, cache_per_cu);
    }
    const dim3 grid(std::min(args.n_launch, cache_per_cu * (u32)n_cu)), block(NTH);
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
    if ((e = hipMalloc((void **)&vp->tile_class, 2 * ntiles)) != hipSuccess) return (int)e;
    if ((e = hipMalloc((void **)&vp->launch_list, sizeof(u32) * ntiles)) != hipSuccess) return (int)e;
    vp->ntiles_alloc = ntiles;
    unsigned char *any_all = vp->tile_class + ntiles;
    hipLaunchKernelGGL(prf_pack_vertical_kernel, dim3((u32)ntiles), dim3(64), 0, s, asc, vp->VH, vp->VL, any_all);
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
