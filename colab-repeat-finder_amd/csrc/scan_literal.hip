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

namespace {

typedef long long i64;

__device__ __forceinline__ bool lit_match(const uint8_t *__restrict__ s, i64 j, i64 k) {
    const uint8_t a = s[j];
    return a == s[j + k] && a != 'N';  // tracker :53
}

// tracker :108-142: is the word a whole number (>= 2) of copies of a shorter unit?  The reference tries the units 1, 2, 3 and
// then every divisor u >= 4 with 2u <= n by counting non-overlapping copies of the prefix; n/u copies in n letters tile the
// word, so each branch is "the word has period u".
__device__ bool lit_is_repeat(const uint8_t *__restrict__ m, i64 n) {
    for (i64 u = 1; 2 * u <= n; u++) {
        if (n % u) continue;
        bool per = true;
        for (i64 t = u; t < n; t++)
            if (m[t] != m[t - u]) {
                per = false;
                break;
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

// one flush call of tracker k at position i0 (a failed comparison, or the position done() finds the tracker at)
__device__ void lit_event(const uint8_t *__restrict__ s, i64 L, i64 k, i64 i0, u32 min_repeats, u32 min_span, u32 contig,
                          prf_hit_dev *__restrict__ rows, u64 cap, u64 *__restrict__ counters) {
    // the run that ends here: run_length - 1 matching positions directly in front of i0
    i64 start = i0;
    while (start > 0 && lit_match(s, start - 1, k)) start--;
    i64 run = i0 - start + 1;

    const i64 mlen = (start + k <= L ? start + k : L) - start;  // :82, slice clamped at the end
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
    for (i64 t = 0; t < mlen; t++)
        if (s[start + t] == 'N') return;  // :83
    if (index_error) {
        atomicOr(&counters[PRF_CNT_CAND], 1ull);
        return;
    }
    if (lit_is_repeat(s + start, mlen)) return;                  // :98
    const u64 slot = atomicAdd(&counters[PRF_CNT_HITS], 1ull);
    // k of the row = length of the motif slice (< the tracker's k only where :82 clamped it): motif = seq[start : start + k]
    if (slot < cap) rows[slot] = prf_hit_dev{(u64)start, (u64)(i + 1), (u32)mlen, contig};
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

// rows[0..n) -> out[0..*n_out) (device memory, n_out a device word); scratch allocated and freed here (the slow lane)
hipError_t prf_lit_sort_unique(hipStream_t st, const prf_hit_dev *rows, u64 n, prf_hit_dev *out, u64 *n_out) {
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
    const size_t tmp_bytes = tmp_sort > tmp_sel ? tmp_sort : tmp_sel;
    step(hipMalloc((void **)&key_a, n * 8));
    step(hipMalloc((void **)&key_b, n * 8));
    step(hipMalloc((void **)&idx_a, n * 4));
    step(hipMalloc((void **)&idx_b, n * 4));
    step(hipMalloc((void **)&sorted, n * sizeof(prf_hit_dev)));
    step(hipMalloc((void **)&first, n));
    step(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
    if (e == hipSuccess) {
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
        step(hipStreamSynchronize(st));  // the scratch is freed below
    }
    (void)hipFree(key_a); (void)hipFree(key_b); (void)hipFree(idx_a); (void)hipFree(idx_b);
    (void)hipFree(sorted); (void)hipFree(first); (void)hipFree(tmp);
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
