// scan_literal.hip -- the reference's state machine evaluated EVENT BY EVENT, for the regimes the closed form of the
// bit-sliced kernels does not cover: min_repeats == 1, and a lock-step loop that stops early (interval mode).
//
// The reference walks one PerfectRepeatTracker per motif size k over the sequence (utils/perfect_repeat_tracker.py:43-61)
// and calls output_interval_if_it_passes_filters() (:71-101) at every position i < len-k whose comparison fails, and once
// more from done() (:67-69) at the position the tracker stopped at.  Nothing such a call does depends on an earlier call of
// the same tracker (the run length restarts at 1, :60), and the shared (start, end) -> motif dictionary ends up, whatever
// the order of the calls, with the SHORTEST motif among the calls that passed every test (:93-101: a longer one never
// replaces a shorter one, a shorter or equal one always does).  So every call is an independent event:
//
//   one thread per (four positions, motif size k); a position is left at once unless it is a failed comparison or the final
//   position; it finds the run that ends at i by walking back, applies :81-101 as written (the N test moved behind the
//   filters, see there) -- the motif slice clamped at the
//   end of the sequence (:82), the "N" in motif test (:83), the first filter (:86), the extension loop over seq[i+1] ==
//   seq[i+1-k] with Python's negative-index wrap-around and its IndexError (:87-89), the second filter (:91), the
//   primitive-motif test (:98, :108-142) -- and appends (start, end, len(motif)) to the row array.
//
// prf_lit_sort_unique() then sorts the rows and keeps, per (start, end), the one with the shortest motif slice (on the device).  Work is one byte compare per
// (i, k) plus O(run) per event: HBM/L2-bound byte work, nothing to tile.  It is the slow lane on purpose -- the default
// regime (min_repeats >= 2, whole sequences) never comes here.
#include "prf_host.h"

#include <hipcub/hipcub.hpp>

#include <cstdlib>

namespace {

typedef long long i64;

__device__ __forceinline__ bool lit_match(const uint8_t *__restrict__ s, i64 j, i64 k) {
    const uint8_t a = s[j];
    return a == s[j + k] && a != 'N';  // tracker :53
}

// eight bytes from any position (two aligned dwords and a third, v_alignbyte_b32): the buffer has 16 readable bytes behind the
// sequence, callers read from positions below L
__device__ __forceinline__ u64 lit_load8(const uint8_t *__restrict__ s, i64 pos) {
    const u32 *w = reinterpret_cast<const u32 *>(s + (pos & ~3ll));
    const u32 sh = (u32)(pos & 3ll);
    const u32 w0 = w[0], w1 = w[1], w2 = w[2];
    return (u64)__builtin_amdgcn_alignbyte(w1, w0, sh) | ((u64)__builtin_amdgcn_alignbyte(w2, w1, sh) << 32);
}
__device__ __forceinline__ bool lit_has_zero_byte(u64 x) { return ((x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull) != 0; }

// "N" in s[start : start + n]   (tracker :83), eight letters per step
__device__ __forceinline__ bool lit_has_n(const uint8_t *__restrict__ s, i64 start, i64 n) {
    u64 any = 0;
    for (i64 t = 0; t < n; t += 8) {
        u64 x = lit_load8(s, start + t) ^ 0x4E4E4E4E4E4E4E4Eull;
        const i64 rem = n - t;
        if (rem < 8) x |= ~0ull << (8 * rem);  // letters behind the slice: never N
        any |= (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;
    }
    return any != 0;
}

// tracker :108-142: is the word a whole number (>= 2) of copies of a shorter unit?  The reference tries the units 1, 2, 3 and
// then every divisor u >= 4 with 2u <= n by counting non-overlapping copies of the prefix; n/u copies in n letters tile the
// word, so each branch is "the word has period u".  Eight letters per comparison (round 3: byte by byte, every step a dependent
// load, this and the N test were most of the lane's time with min_repeats == 1).
__device__ bool lit_is_repeat(const uint8_t *__restrict__ s, i64 start, i64 n) {
    for (i64 u = 1; 2 * u <= n; u++) {
        if (n % u) continue;
        bool per = true;
        for (i64 t = u; t < n; t += 8) {
            u64 x = lit_load8(s, start + t) ^ lit_load8(s, start + t - u);
            const i64 rem = n - t;
            if (rem < 8) x &= (1ull << (8 * rem)) - 1ull;
            if (x) {
                per = false;
                break;
            }
        }
        if (per) return true;
    }
    return false;
}

// bytes -> upper case in place (reference perfect_repeat_finder.py:33); the first byte that is not a letter is reported
__global__ void prf_lit_upper_kernel(uint8_t *__restrict__ s, u64 n, u64 *__restrict__ bad_pos) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint8_t c = s[i];
        if (c >= 'a' && c <= 'z') s[i] = c = (uint8_t)(c - 32);
        if (c < 'A' || c > 'Z') atomicMin(bad_pos, i);
    }
}

// the end of a flush call: the run [start, i] has passed both filters (or Python has raised IndexError in :87) -- the N test,
// the primitive-motif test, the row
// wave buffer (the 64-positions kernel): rows are collected in LDS, a wave reserves their places in the row array with ONE atomic per
// flush -- every row drawing its own place from the one counter word was 240 k atomics on one address on 50 Mbp of random sequence,
// 2.7 ms at the ~90 per microsecond such a word takes
struct lit_wave_buf {
    prf_hit_dev rows[64];
    u32 n;
};

__device__ void lit_finish(const uint8_t *__restrict__ s, i64 L, i64 k, i64 start, i64 i, bool index_error, u32 contig,
                           prf_hit_dev *__restrict__ rows, u64 cap, u64 *__restrict__ counters, lit_wave_buf *wb = nullptr) {
    const i64 mlen = (start + k <= L ? start + k : L) - start;  // :82, slice clamped at the end
    if (lit_has_n(s, start, mlen)) return;                       // :83
    if (index_error) {
        atomicOr(&counters[PRF_CNT_CAND], 1ull);
        return;
    }
    if (lit_is_repeat(s, start, mlen)) return;                   // :98
    // k of the row = length of the motif slice (< the tracker's k only where :82 clamped it): motif = seq[start : start + k]
    const prf_hit_dev row{(u64)start, (u64)(i + 1), (u32)mlen, contig};
    if (wb) {
        const u32 at = atomicAdd(&wb->n, 1u);
        if (at < 64u) {
            wb->rows[at] = row;
            return;
        }  // (a full buffer: this row goes straight to the array; the flush takes min(n, 64))
    }
    const u64 slot = atomicAdd(&counters[PRF_CNT_HITS], 1ull);
    if (slot < cap) rows[slot] = row;
}

// one flush call of tracker k at position i0 (a failed comparison, or the position done() finds the tracker at)
__device__ void lit_event(const uint8_t *__restrict__ s, i64 L, i64 k, i64 i0, u32 min_repeats, u32 min_span, u32 contig,
                          prf_hit_dev *__restrict__ rows, u64 cap, u64 *__restrict__ counters, lit_wave_buf *wb = nullptr) {
    // the run that ends here: run_length - 1 matching positions directly in front of i0
    i64 start = i0;
    while (start > 0 && lit_match(s, start - 1, k)) start--;
    i64 run = i0 - start + 1;

    // :83 ("N" in motif -> return) reads up to k bytes and nearly every event fails a filter anyway, so it is evaluated
    // LAST: nothing between :83 and :98 has a side effect except the IndexError of :87, which is therefore raised only after
    // the N test has been made at that point.
    const i64 need = (i64)min_repeats * k;
    i64 i = i0;
    bool index_error = false;
    if (run + k - 1 >= (i64)min_span && run + k - 1 >= need) {  // :86
        while (i < L - 1) {                                      // :87
            i64 j = i + 1 - k;
            if (j < 0) j += L;                                   // Python: a negative index counts from the end ...
            if (j < 0) {                                         // ... and raises IndexError past the front
                index_error = true;
                break;
            }
            if (s[i + 1] != s[j]) break;
            run++;
            i++;
        }
    }
    if (!index_error && (run < (i64)min_span || run < need)) return;  // :91
    lit_finish(s, L, k, start, i, index_error, contig, rows, cap, counters, wb);
}

// Four positions per thread: one aligned dword of the sequence against the (unaligned) dword k bytes further on, taken from
// two aligned dwords with v_alignbyte_b32.  The buffer has 16 readable bytes behind the sequence.
__global__ void __launch_bounds__(256) prf_lit_events_kernel(const uint8_t *__restrict__ s, i64 L, u32 kmin, u32 n_k, u32 min_repeats,
                                                             u32 min_span, i64 stop, u32 contig, prf_hit_dev *__restrict__ rows,
                                                             u64 cap, u64 *__restrict__ counters, u64 block0) {
    // the motif size is the fast index of the grid: the workgroups resident at one time read the same stretch of the
    // sequence for all motif sizes, which L2 then serves (with the motif size as the slow index every size streamed the
    // whole sequence from HBM again: FETCH_SIZE 1.5 GB per launch for 50 MB x 50 sizes)
    const i64 k = (i64)kmin + blockIdx.x % n_k;
    const i64 Lk = L > k ? L - k : 0;            // tracker :50: the tracker never moves past len - k
    const i64 pos_f = stop < Lk ? stop : Lk;     // where it stands when done() is called
    const i64 i_base = 4 * ((i64)(block0 + blockIdx.x / n_k) * blockDim.x + threadIdx.x);  // (block0: a long sequence takes several launches)
    if (i_base > pos_f) return;
    u32 a = 0, b = 0;
    if (i_base < pos_f) {                        // then i_base + k < L: both dwords lie inside the buffer
        a = *(const u32 *)(s + i_base);
        const u64 q = (u64)(i_base + k);
        const u32 *w = (const u32 *)(s + (q & ~3ull));
        b = __builtin_amdgcn_alignbyte(w[1], w[0], (u32)(q & 3ull));
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const i64 i0 = i_base + t;
        if (i0 > pos_f) break;
        const u32 ca = (a >> (8 * t)) & 0xffu, cb = (b >> (8 * t)) & 0xffu;
        if (i0 < pos_f && ca == cb && ca != 'N') continue;  // a matching position only lengthens the run (:53-56)
        lit_event(s, L, k, i0, min_repeats, min_span, contig, rows, cap, counters);
    }
}

// ---- the same events, 64 positions per thread in registers (round 3) ----
// The kernel above spends its time in the event routine: with min_repeats == 1 three positions in four are events, and each walks
// back and forth over the sequence byte by byte.  Here a thread owns the 64 positions [64 B, 64 B + 64) and keeps, per motif size k
// <= 63, two masks of them: mm (bit i: position i is a failed comparison of the tracker -- the letters differ, or the letter is N,
// tracker :53) and pm (bit i: the letters are equal -- what the extension loop :87-89 compares, without the N rule).  The masks
// of the 64 positions in front come from the lane below (lane 0 of a wave owns nothing: it computes the block in front of lane
// 1's).  For an event at position i, "run length in front" is the distance to the next set bit of mm below i, and "extension" the
// length of the row of ones of pm from i + 1 - k on, so the two filters (:86, :91) are bit arithmetic.  What the masks can
// decide is only that an event FAILS a filter -- which is what nearly every event does; an event that passes both, or whose run or
// extension leaves the 128 positions the thread sees, is handed to lit_event() above, which evaluates it from the bytes as
// before.  A bit-parallel test first drops, for all 64 positions at once, every event with run < a and extension < b, where
// (a - 1) + (b - 1) < T = max(min_span, min_repeats k): such an event cannot reach T.  Blocks near the ends of the sequence
// (the wrap-around of :87 at the front, the clamped slice and done() at the back) go through the byte routine position by
// position.
__device__ __forceinline__ u32 lit_nonzero_bytes(u32 x) {  // bit t = byte t of x is not zero
    const u32 nz = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
    const u32 y = nz >> 7;
    return (y | (y >> 7) | (y >> 14) | (y >> 21)) & 15u;
}
// 128-bit value hi:lo shifted left / right by s in [1, 63]; the right shift fills with ones (positions past the window: unknown,
// counted as matching so that nothing is dropped on their account)
__device__ __forceinline__ void lit_shl(u64 &lo, u64 &hi, u32 sft) {
    hi = (hi << sft) | (lo >> (64u - sft));
    lo <<= sft;
}
__device__ __forceinline__ void lit_shr1(u64 &lo, u64 &hi, u32 sft) {
    lo = (lo >> sft) | (hi << (64u - sft));
    hi = (hi >> sft) | (~0ull << (64u - sft));
}

__global__ void __launch_bounds__(256) prf_lit_events64_kernel(const uint8_t *__restrict__ s, i64 L, u32 kmin, u32 n_k, u32 min_repeats,
                                                               u32 min_span, i64 stop, u32 contig, prf_hit_dev *__restrict__ rows,
                                                               u64 cap, u64 *__restrict__ counters, u64 wave0) {
    __shared__ lit_wave_buf wbufs[4];
    const u32 lane = threadIdx.x & 63u;
    lit_wave_buf *wb = &wbufs[threadIdx.x >> 6];
    if (lane == 0) wb->n = 0;
    auto flush = [&]() {  // the wave's rows -> the row array: one reservation (every lane of the wave gets here together)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const u32 n_all = *(volatile u32 *)&wb->n;
        const u32 n = n_all < 64u ? n_all : 64u;
        u64 base = 0;
        if (lane == 0 && n) base = atomicAdd(&counters[PRF_CNT_HITS], (u64)n);
        base = __shfl(base, 0, 64);
        if (lane < n && base + lane < cap) rows[base + lane] = wb->rows[lane];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (lane == 0) wb->n = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };
    const i64 W = (i64)(wave0 + (u64)blockIdx.x * 4u + (threadIdx.x >> 6));
    const i64 B = W * 63 + (i64)lane - 1;  // the block of 64 positions of this lane (lane 0: the one in front of lane 1's)
    const i64 p0 = B * 64;
    // the block's own letters, once for all motif sizes (readable: whole dwords below L + 16)
    const bool has_a = B >= 0 && p0 + 64 <= L + 16;
    u32 a[16];
    u64 is_n = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) a[j] = 0;
    if (has_a) {
        const uint4 *pa = reinterpret_cast<const uint4 *>(s + p0);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint4 v = pa[j];
            a[4 * j] = v.x; a[4 * j + 1] = v.y; a[4 * j + 2] = v.z; a[4 * j + 3] = v.w;
        }
#pragma unroll
        for (int j = 0; j < 16; j++) is_n |= (u64)(lit_nonzero_bytes(a[j] ^ 0x4E4E4E4Eu) ^ 15u) << (4 * j);
    }
    for (u32 ki = 0; ki < n_k; ki++) {
        const i64 k = (i64)kmin + ki;  // <= 63
        const i64 Lk = L > k ? L - k : 0;
        const i64 pos_f = stop < Lk ? stop : Lk;
        const i64 T = (i64)min_span > (i64)min_repeats * k ? (i64)min_span : (i64)min_repeats * k;
        // ---- masks of the own block: valid if every byte read lies below L + 16
        u64 mm = 0, pm = 0;
        const bool has_b = has_a && p0 + 64 + k + 3 < L + 16;
        if (has_b) {
            const u64 q = (u64)(p0 + k);
            const u32 *w = reinterpret_cast<const u32 *>(s + (q & ~3ull));
            const u32 sh = (u32)(q & 3ull);
            u32 wv[17];
#pragma unroll
            for (int j = 0; j < 17; j++) wv[j] = w[j];
            u64 ne = 0;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const u32 b = __builtin_amdgcn_alignbyte(wv[j + 1], wv[j], sh);
                ne |= (u64)lit_nonzero_bytes(a[j] ^ b) << (4 * j);
            }
            mm = ne | is_n;
            pm = ~ne;
        }
        // the block in front: the lane below (lane 0's own value is never used: it owns nothing)
        const u64 mm_lo = __shfl_up(mm, 1, 64), pm_lo = __shfl_up(pm, 1, 64);
        const bool owner = !(lane == 0 || B < 0 || p0 > pos_f);
        const bool interior = owner && B >= 1 && p0 + 128 + k <= pos_f;  // (then every byte of both blocks' masks was readable, i >= k, i < pos_f)
        if (owner && !interior) {
            for (int t = 0; t < 64; t++) {
                const i64 i0 = p0 + t;
                if (i0 > pos_f) break;
                if (i0 < pos_f && lit_match(s, i0, k)) continue;
                lit_event(s, L, k, i0, min_repeats, min_span, contig, rows, cap, counters, wb);
            }
        }
        if (interior) {
        // ---- all 64 positions at once: events that cannot reach T.  With run = length in front (the event's own position
        // counted) and ext = extension: run + ext >= T implies run >= av or ext >= bv, where (av - 1) + (bv - 1) < T; and then
        //   run >= av:  ext >= 1, or the run alone reaches T;      ext >= bv:  run >= 2, or the extension alone reaches T - 1.
        // (Thresholds beyond the 60 positions a thread can look across are cut to that -- a weaker test, never a wrong one.)
        u64 cand = mm;
        if (T > 1) {
            i64 av = T / 2 + 1;  // 2 (av - 1) >= T - 1: the "run alone" window is one more step of the same doubling
            if (av > 60) av = 60;
            i64 bv = T - av + 1;  // 2 bv >= T
            if (bv > 60) bv = 60;
            if (bv < 1) bv = 1;
            const u32 tr = (u32)((T < 61 ? T : 61) - 1);  // matches below for "the run alone": <= 2 (av - 1)
            const u32 te = (u32)(T - 1 < 60 ? T - 1 : 60);  // "the extension alone": <= 2 bv
            // dn: the av - 1 positions directly below match (tracker rule); d1: the one below does; dt: tr of them do
            u64 lo = ~mm_lo, hi = ~mm;
            lit_shl(lo, hi, 1);  // bit n: position n - 1 matches
            const u64 d1 = hi;
            const u32 r = (u32)(av - 1);  // >= 1
            u32 have = 1;
            while (2 * have <= r) {
                u64 l2 = lo, h2 = hi;
                lit_shl(l2, h2, have);
                lo &= l2; hi &= h2;
                have *= 2;
            }
            if (have < r) {
                u64 l2 = lo, h2 = hi;
                lit_shl(l2, h2, r - have);
                lo &= l2; hi &= h2;
            }
            const u64 dn = hi;
            u64 dt = hi;
            if (tr > r) {
                u64 l2 = lo, h2 = hi;
                lit_shl(l2, h2, tr - r);
                dt = hi & h2;
            }
            // e1 / eb / et: 1 / bv / te equal letters from position i + 1 - k on (bit n <- bit n + 1 - k: a left shift by k - 1)
            u64 plo = pm_lo, phi = pm;
            u64 e1lo = plo, e1 = phi;
            if (k > 1) lit_shl(e1lo, e1, (u32)(k - 1));
            have = 1;
            const u32 bb = (u32)bv;
            while (2 * have <= bb) {
                u64 l2 = plo, h2 = phi;
                lit_shr1(l2, h2, have);
                plo &= l2; phi &= h2;
                have *= 2;
            }
            if (have < bb) {
                u64 l2 = plo, h2 = phi;
                lit_shr1(l2, h2, bb - have);
                plo &= l2; phi &= h2;
            }
            u64 tlo = plo, thi = phi;
            if (te > bb) {
                u64 l2 = plo, h2 = phi;
                lit_shr1(l2, h2, te - bb);
                tlo &= l2; thi &= h2;
            }
            u64 eb = phi, et = thi;
            if (k > 1) {
                lit_shl(plo, eb, (u32)(k - 1));
                lit_shl(tlo, et, (u32)(k - 1));
            }
            cand &= (dn & e1) | dt | (eb & d1) | et;
        }
        // ---- the events left, one by one, still in registers
        const i64 R1 = T - k + 1;  // :86  run + k - 1 >= T
        while (cand) {
            const u32 i = (u32)__builtin_ctzll(cand);
            cand &= cand - 1;
            bool slow = false;
            i64 run = 0;
            const u64 below = i ? mm & ((1ull << i) - 1ull) : 0ull;
            if (below) run = (i64)i - (63 - (i64)__builtin_clzll(below));
            else if (mm_lo) run = (i64)i + 1 + (i64)__builtin_clzll(mm_lo);
            else slow = true;  // the run begins in front of the 128 positions
            if (!slow) {
                if (run < R1) continue;  // :86 fails, and then :91 does (run <= run + k - 1 < T)
                // extension: ones of pm from window bit m = 64 + i + 1 - k on (m >= 2)
                const u32 m = 65u + i - (u32)k;
                i64 ext = -1;
                if (m < 64u) {
                    const u64 v = ~((pm_lo >> m) | (pm << (64u - m)));
                    if (v) ext = (i64)__builtin_ctzll(v);
                    else {
                        const u64 v2 = ~(pm >> m) & (~0ull >> m);
                        if (v2) ext = 64 + (i64)__builtin_ctzll(v2);
                    }
                } else {
                    const u32 m2 = m - 64u;
                    const u64 v = m2 ? ~(pm >> m2) & (~0ull >> m2) : ~pm;
                    if (v) ext = (i64)__builtin_ctzll(v);
                }
                if (ext < 0) slow = true;  // the extension reaches the end of the 128 positions
                else if (run + ext < T) continue;  // :91
                else {
                    // both filters passed, run and extension known: the run is [i0 - run + 1, i0 + ext]
                    const i64 i0 = p0 + (i64)i;
                    lit_finish(s, L, k, i0 - run + 1, i0 + ext, false, contig, rows, cap, counters, wb);
                    continue;
                }
            }
            lit_event(s, L, k, p0 + (i64)i, min_repeats, min_span, contig, rows, cap, counters, wb);
        }
        }
        // (all lanes again) a buffer half full is emptied: the next motif size may add as many rows again
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if ((u32)__builtin_amdgcn_readfirstlane((int)*(volatile u32 *)&wb->n) >= 32u) flush();
    }
    flush();
}

struct lit_codes {
    const u64 *e[5];  // code planes of the symbols outside ACGTN (prf_planes::E); e[0] == nullptr: the genome has none
};

// first / one-past-last position of a contig that is not N: one thread per 64-position word of the X plane
__global__ void prf_lit_trim_kernel(const u64 *__restrict__ X, lit_codes E, u64 word0, u64 len, u64 *__restrict__ first_last) {
    const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (w * 64 >= len) return;
    u64 is_n = X[word0 + w];  // not ACGT ...
    if (E.e[0]) is_n &= ~(E.e[0][word0 + w] | E.e[1][word0 + w] | E.e[2][word0 + w] | E.e[3][word0 + w] | E.e[4][word0 + w]);  // ... and no other letter
    u64 other = ~is_n;
    const u64 left = len - w * 64;
    if (left < 64) other &= (1ull << left) - 1ull;
    if (!other) return;
    const u64 first = w * 64 + (u64)__builtin_ctzll(other), last = w * 64 + 64 - (u64)__builtin_clzll(other);
    // most words improve neither bound: look before the atomic
    if (first < __atomic_load_n(&first_last[0], __ATOMIC_RELAXED)) atomicMin(&first_last[0], first);
    if (last > __atomic_load_n(&first_last[1], __ATOMIC_RELAXED)) atomicMax(&first_last[1], last);
}

// upper-cased bytes of positions g0 .. g0+n-1 (global coordinate space) from the planes: H, L = bits 2 and 1 of the letter
// for A, C, G, T (pack.hip); X = not ACGT: the letter with the five-bit code of the E planes, or N where the code is 0
__global__ void prf_lit_unpack_kernel(const u64 *__restrict__ H, const u64 *__restrict__ L, const u64 *__restrict__ X, lit_codes E,
                                      u64 g0, u64 n, uint8_t *__restrict__ out) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        const u64 q = g0 + p, w = q >> 6;
        const unsigned b = (unsigned)(q & 63);
        uint8_t ch;
        if (!((X[w] >> b) & 1ull)) {
            const unsigned h = (unsigned)((H[w] >> b) & 1ull), l = (unsigned)((L[w] >> b) & 1ull);
            ch = h ? (l ? 'G' : 'T') : (l ? 'C' : 'A');
        } else {
            unsigned code = 0;
            if (E.e[0])
                for (int i = 0; i < 5; i++) code |= (unsigned)((E.e[i][w] >> b) & 1ull) << i;
            ch = code ? (uint8_t)('@' + code) : (uint8_t)'N';
        }
        out[p] = ch;
    }
}

}  // namespace

hipError_t prf_launch_lit_trim(hipStream_t st, const u64 *X, const u64 *const *E, u64 word0, u64 len, u64 *first_last) {
    lit_codes codes{{E[0], E[1], E[2], E[3], E[4]}};
    const u64 words = (len + 63) / 64;
    if (!words) return hipSuccess;
    hipLaunchKernelGGL(prf_lit_trim_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, X, codes, word0, len, first_last);
    return hipGetLastError();
}

hipError_t prf_launch_lit_unpack(hipStream_t st, const u64 *H, const u64 *L, const u64 *X, const u64 *const *E, u64 g0, u64 n,
                                 uint8_t *out) {
    lit_codes codes{{E[0], E[1], E[2], E[3], E[4]}};
    if (!n) return hipSuccess;
    const u64 blocks = (n + 256ull * 8 - 1) / (256ull * 8);
    hipLaunchKernelGGL(prf_lit_unpack_kernel, dim3((unsigned)(blocks < 262144 ? blocks : 262144)), dim3(256), 0, st, H, L, X, codes, g0, n, out);
    return hipGetLastError();
}

// ---- rows of the event kernel -> sorted by (start, end), one row per (start, end): the shortest motif ----
// What the reference's dictionary and its sorted() leave (utils/perfect_repeat_tracker.py:93-101, perfect_repeat_finder.py:81),
// on the device: two stable radix sorts of row indices (by (end, motif length), then by start), a gather that flags the first
// row of every (start, end) group, and a flagged compaction.
namespace {
__global__ void prf_lit_key_end_kernel(const prf_hit_dev *__restrict__ rows, u64 n, u64 *__restrict__ key, u32 *__restrict__ idx) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    key[i] = (rows[i].end << 16) | (u64)(rows[i].k & 0xffffu);  // end < 2^41, motif length <= 60000
    idx[i] = (u32)i;
}
__global__ void prf_lit_key_start_kernel(const prf_hit_dev *__restrict__ rows, u64 n, const u32 *__restrict__ idx, u64 *__restrict__ key) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) key[i] = rows[idx[i]].start;
}
__global__ void prf_lit_gather_kernel(const prf_hit_dev *__restrict__ rows, u64 n, const u32 *__restrict__ idx,
                                      prf_hit_dev *__restrict__ sorted, unsigned char *__restrict__ first) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const prf_hit_dev r = rows[idx[i]];
    sorted[i] = r;
    bool f = true;
    if (i) {
        const prf_hit_dev q = rows[idx[i - 1]];
        f = q.start != r.start || q.end != r.end;
    }
    first[i] = f ? 1 : 0;
}
}  // namespace

// rows[0..n) -> out[0..*n_out) (device memory, n_out a device word).  The scratch (two key and two index arrays, the sorted rows,
// the flags, hipCUB's own) is ONE allocation kept by the caller (*scratch, *scratch_bytes: the context owns it and frees it) and
// grown when a call needs more: the first version made and freed seven allocations per call.
hipError_t prf_lit_sort_unique(hipStream_t st, const prf_hit_dev *rows, u64 n, prf_hit_dev *out, u64 *n_out, void **scratch,
                               size_t *scratch_bytes) {
    if (n == 0) return hipMemsetAsync(n_out, 0, sizeof(u64), st);
    if (n > 0x7fffffffull) return hipErrorInvalidValue;
    const int ni = (int)n;
    u64 *key_a = nullptr, *key_b = nullptr;
    u32 *idx_a = nullptr, *idx_b = nullptr;
    prf_hit_dev *sorted = nullptr;
    unsigned char *first = nullptr;
    void *tmp = nullptr;
    size_t tmp_sort = 0, tmp_sel = 0;
    hipError_t e = hipSuccess;
    const unsigned nb = (unsigned)((n + 255) / 256);
    auto step = [&](hipError_t r) { if (e == hipSuccess) e = r; return e == hipSuccess; };
    step(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_sort, key_a, key_b, idx_a, idx_b, ni, 0, 64, st));
    step(hipcub::DeviceSelect::Flagged(nullptr, tmp_sel, sorted, first, out, n_out, ni, st));
    const size_t tmp_bytes = (tmp_sort > tmp_sel ? tmp_sort : tmp_sel) + 256;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_key_b = up(n * 8), o_idx_a = o_key_b + up(n * 8), o_idx_b = o_idx_a + up(n * 4), o_sorted = o_idx_b + up(n * 4),
                 o_first = o_sorted + up(n * sizeof(prf_hit_dev)), o_tmp = o_first + up(n), total = o_tmp + tmp_bytes;
    if (e == hipSuccess && total > *scratch_bytes) {
        step(hipStreamSynchronize(st));  // nothing in flight may still use the old block
        (void)hipFree(*scratch);
        *scratch = nullptr;
        *scratch_bytes = 0;
        if (step(hipMalloc(scratch, total + total / 4))) *scratch_bytes = total + total / 4;
    }
    if (e == hipSuccess) {
        char *base = (char *)*scratch;
        key_a = (u64 *)base; key_b = (u64 *)(base + o_key_b);
        idx_a = (u32 *)(base + o_idx_a); idx_b = (u32 *)(base + o_idx_b);
        sorted = (prf_hit_dev *)(base + o_sorted);
        first = (unsigned char *)(base + o_first);
        tmp = base + o_tmp;
        size_t b = tmp_bytes;
        hipLaunchKernelGGL(prf_lit_key_end_kernel, dim3(nb), dim3(256), 0, st, rows, n, key_a, idx_a);
        step(hipGetLastError());
        step(hipcub::DeviceRadixSort::SortPairs(tmp, b, key_a, key_b, idx_a, idx_b, ni, 0, 58, st));   // by (end, motif length)
        hipLaunchKernelGGL(prf_lit_key_start_kernel, dim3(nb), dim3(256), 0, st, rows, n, idx_b, key_a);
        step(hipGetLastError());
        b = tmp_bytes;
        step(hipcub::DeviceRadixSort::SortPairs(tmp, b, key_a, key_b, idx_b, idx_a, ni, 0, 42, st));   // stable: by start, ties keep (end, length)
        hipLaunchKernelGGL(prf_lit_gather_kernel, dim3(nb), dim3(256), 0, st, rows, n, idx_a, sorted, first);
        step(hipGetLastError());
        b = tmp_bytes;
        step(hipcub::DeviceSelect::Flagged(tmp, b, sorted, first, out, n_out, ni, st));
    }
    return e;
}

hipError_t prf_launch_lit_upper(hipStream_t st, uint8_t *s, u64 n, u64 *bad_pos) {
    if (n == 0) return hipSuccess;
    const u64 blocks = (n + 256ull * 16 - 1) / (256ull * 16);
    hipLaunchKernelGGL(prf_lit_upper_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, st, s, n, bad_pos);
    return hipGetLastError();
}

hipError_t prf_launch_lit_events(hipStream_t st, const uint8_t *s, u64 L, u32 kmin, u32 kmax, u32 min_repeats, u32 min_span,
                                 u64 stop, u32 contig, prf_hit_dev *rows, u64 cap, u64 *counters) {
    // motif sizes up to 63: 64 positions per thread in registers (PRF_LIT_BYTEWISE=1: the byte routine for every size, diagnostic)
    static const bool bytewise = getenv("PRF_LIT_BYTEWISE") && atoi(getenv("PRF_LIT_BYTEWISE")) != 0;
    if (!bytewise && kmin <= 63u) {
        const u32 k_hi = kmax < 63u ? kmax : 63u;
        const u64 Lk = L > kmin ? L - kmin : 0;
        const u64 pos_f = stop < Lk ? stop : Lk;
        const u64 n_blocks = pos_f / 64 + 1;               // blocks 0 .. pos_f / 64 hold the positions 0 .. pos_f
        const u64 n_waves = (n_blocks + 62) / 63;          // 63 owned blocks per wave
        const u64 n_wg = (n_waves + 3) / 4;
        const u64 max_wg = (1ull << 24) - 1ull;
        for (u64 b0 = 0; b0 < n_wg; b0 += max_wg) {
            const u64 nb = n_wg - b0 < max_wg ? n_wg - b0 : max_wg;
            hipLaunchKernelGGL(prf_lit_events64_kernel, dim3((unsigned)nb), dim3(256), 0, st, s, (long long)L, kmin, k_hi - kmin + 1u,
                               min_repeats, min_span, (long long)stop, contig, rows, cap, counters, b0 * 4u);
            const hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
        if (kmax <= 63u) return hipSuccess;
        kmin = 64u;
    }
    // positions 0 .. pos_f of the smallest k cover every k
    const u64 Lk = L > kmin ? L - kmin : 0;
    const u64 n_threads = (stop < Lk ? stop : Lk) / 4 + 1;  // four positions per thread
    const u64 bx = (n_threads + 255) / 256;
    const u64 n_k = kmax - kmin + 1;
    // one-dimensional grid: workgroup b = (positions b / n_k, size b % n_k).  A grid holds fewer than 2^32 threads (2^24
    // workgroups of 256): an hg38 chr1-sized sequence with 100 motif sizes takes two launches (it used to fail)
    const u64 max_wg = (1ull << 24) - 1ull;
    if (n_k > max_wg) return hipErrorInvalidValue;
    const u64 bx_per_launch = max_wg / n_k;
    for (u64 b0 = 0; b0 < bx; b0 += bx_per_launch) {
        const u64 nb = bx - b0 < bx_per_launch ? bx - b0 : bx_per_launch;
        hipLaunchKernelGGL(prf_lit_events_kernel, dim3((unsigned)(nb * n_k)), dim3(256), 0, st, s, (long long)L, kmin, (u32)n_k,
                           min_repeats, min_span, (long long)stop, contig, rows, cap, counters, b0);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
