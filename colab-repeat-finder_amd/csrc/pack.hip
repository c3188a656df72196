// pack.hip -- ASCII -> bit planes, on the device.
//
// Replaces the reference's input_sequence.upper() plus its two further whole-sequence copies
// (reference perfect_repeat_finder.py:33, :46): the ASCII bytes are read once, case-folded in
// registers and written as three 1-bit planes (H, L = base code, X = not-ACGT).
#include "prf_device.h"
#include "prf_host.h"

// One thread packs one 64-position word: 64 bytes in (4 x 16-B loads), 3 x 8 bytes out.
// Symbols: A, C, G, T -> code; N -> X; any other LETTER is an ordinary symbol to the reference (R == R matches,
// utils/perfect_repeat_tracker.py:53): X is set for it here and *exotic is raised, so that the planes of its code are built
// (prf_pack_exotic_kernel); a byte that is no letter at all is refused (bad_pos).
__global__ __launch_bounds__(256) void prf_pack_linear_kernel(const uint8_t *__restrict__ asc, u64 nwords,
                                                              u64 *__restrict__ H, u64 *__restrict__ L,
                                                              u64 *__restrict__ X, u64 *__restrict__ bad_pos, u64 *__restrict__ exotic) {
    const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwords) return;
    const uint4 *src = reinterpret_cast<const uint4 *>(asc + w * 64);
    u64 h = 0, l = 0, x = 0, bad = 0, exo = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint4 v = src[q];
        const u32 d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const u32 c = ((d[j] >> (8 * b)) & 0xFFu);
                const u32 f = c & 0xDFu;  // ASCII case fold for letters (str.upper(), reference :33)
                const u32 is_acgt = (f == 'A') | (f == 'C') | (f == 'G') | (f == 'T');
                // only letters fold onto letters, so comparing the folded byte is exact for a-z/A-Z;
                // bytes outside both letter ranges can never equal 'A','C','G','T','N' after the fold
                // except 0x41..0x5A themselves.
                const u32 is_n = (f == 'N');
                const u32 is_letter = (f >= 'A') & (f <= 'Z');
                const int bit = q * 16 + j * 4 + b;
                h |= (u64)((f >> 2) & 1u & is_acgt) << bit;
                l |= (u64)((f >> 1) & 1u & is_acgt) << bit;
                x |= (u64)(is_acgt ^ 1u) << bit;
                bad |= (u64)(is_letter ^ 1u) << bit;
                exo |= (u64)(is_letter & ((is_acgt | is_n) ^ 1u)) << bit;
            }
        }
    }
    H[w] = h;
    L[w] = l;
    X[w] = x;
    if (bad) atomicMin(bad_pos, w * 64 + (u64)__builtin_ctzll(bad));
    if (exo) atomicMax(exotic, 1ull);
}

// the five code planes of the symbols outside ACGTN (prf_planes::E): low five bits of the upper-cased letter
__global__ __launch_bounds__(256) void prf_pack_exotic_kernel(const uint8_t *__restrict__ asc, u64 nwords, u64 *__restrict__ E0,
                                                              u64 *__restrict__ E1, u64 *__restrict__ E2, u64 *__restrict__ E3,
                                                              u64 *__restrict__ E4) {
    const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwords) return;
    const uint4 *src = reinterpret_cast<const uint4 *>(asc + w * 64);
    u64 e[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint4 v = src[q];
        const u32 d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const u32 f = ((d[j] >> (8 * b)) & 0xFFu) & 0xDFu;
                const u32 plain = (f == 'A') | (f == 'C') | (f == 'G') | (f == 'T') | (f == 'N');
                const u32 code = ((f >= 'A') & (f <= 'Z') & (plain ^ 1u)) ? (f & 31u) : 0u;
                const int bit = q * 16 + j * 4 + b;
#pragma unroll
                for (int i = 0; i < 5; i++) e[i] |= (u64)((code >> i) & 1u) << bit;
            }
        }
    }
    E0[w] = e[0]; E1[w] = e[1]; E2[w] = e[2]; E3[w] = e[3]; E4[w] = e[4];
}

hipError_t prf_launch_pack_exotic(hipStream_t s, const uint8_t *asc, u64 nwords, u64 *const *E) {
    const u32 bs = 256;
    const u64 nb = (nwords + bs - 1) / bs;
    hipLaunchKernelGGL(prf_pack_exotic_kernel, dim3((u32)nb), dim3(bs), 0, s, asc, nwords, E[0], E[1], E[2], E[3], E[4]);
    return hipGetLastError();
}

// fill value for guard gaps and padding: N
__global__ void prf_fill_u64_kernel(u64 *__restrict__ p, u64 n, u64 v) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

// SURVEY 8(d) counter-based generator, 16 bases per lane (one 16-byte store)
__global__ __launch_bounds__(256) void prf_synth_kernel(uint8_t *__restrict__ asc, u64 n, u64 seed) {
    const u64 chunk = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 j0 = chunk * 16;
    if (j0 >= n) return;
    u32 w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 16; i++) {
        u64 z = seed + (j0 + (u64)i + 1ull) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        const u32 code = (u32)(z >> 62);
        const u32 ch = (0x54474341u >> (8 * code)) & 0xFFu;  // "ACGT"[code]
        w[i >> 2] |= (j0 + (u64)i < n ? ch : (u32)'N') << (8 * (i & 3));
    }
    *reinterpret_cast<uint4 *>(asc + j0) = make_uint4(w[0], w[1], w[2], w[3]);
}

// ---- stand-in recipe 2 (colab-repeat-finder_amd/synth.py::standin2, integer arithmetic only, so that numpy and this kernel
// agree bit for bit): uniform background; N at [0, n_head), [n - n_tail, n) and [gap_lo, gap_hi); the body
// [n_head, n - n_tail) is cut into slots of 588 positions, each with ONE planted perfect tandem repeat whose motif is the
// background at the repeat's own first k positions.
__device__ __forceinline__ u64 prf_splitmix(u64 z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ u32 prf_background(u64 seed, u64 j) { return (u32)(prf_splitmix(seed + (j + 1ull) * 0x9E3779B97F4A7C15ull) >> 62); }

struct prf_slot_repeat {
    u64 p;       // first position
    u32 k, span;
};
__device__ __forceinline__ prf_slot_repeat prf_slot_params(u64 seed, u64 body_lo, u64 slot) {
    const u64 d = prf_splitmix((seed ^ 0xD1B54A32D192ED03ull) + (slot + 1ull) * 0x9E3779B97F4A7C15ull);
    const u64 d2 = prf_splitmix(d + 0x9E3779B97F4A7C15ull);
    const u32 u = (u32)(d >> 48), v = (u32)(d >> 24) & 0xFFFFFFu, w = (u32)d & 0xFFFFFFu, x = (u32)(d2 >> 32);
    // motif size: the reference's golden chr22 BED histogram for 1..6 (95 %), a thin tail up to 50
    u32 k;
    if (u < 18412u) k = 1; else if (u < 29561u) k = 2; else if (u < 49738u) k = 3; else if (u < 58461u) k = 4;
    else if (u < 61413u) k = 5; else if (u < 62268u) k = 6;
    else { const u32 t = u - 62268u; k = 7u + (t * t * 44u) / (3268u * 3268u); }
    // copies: 3 + geometric(0.67); thresholds floor(2^24 * 0.67^(i+1))
    const u32 th[16] = {11240734u, 7531292u, 5045965u, 3380797u, 2265134u, 1517639u, 1016818u, 681268u,
                        456449u, 305821u, 204900u, 137283u, 91979u, 61626u, 41289u, 27664u};
    u32 copies = 3;
#pragma unroll
    for (int i = 0; i < 16; i++) copies += v < th[i] ? 1u : 0u;
    u32 span = k * copies + w % k;
    span = span > 500u ? 500u : span;
    span = span < 10u ? 10u : span;
    prf_slot_repeat r;
    r.k = k;
    r.span = span;
    r.p = body_lo + slot * 588ull + (u64)(x % (588u - span));
    return r;
}

__global__ __launch_bounds__(256) void prf_standin2_kernel(uint8_t *__restrict__ asc, u64 n, u64 seed, u64 n_head, u64 n_tail,
                                                            u64 gap_lo, u64 gap_hi) {
    const u64 chunk = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 j0 = chunk * 16;
    if (j0 >= n) return;
    const u64 body_lo = n_head, body_hi = n - n_tail;
    const u64 n_slots = (body_hi - body_lo) / 588ull;
    // 16 consecutive positions touch at most two slots
    u64 cur_slot = ~0ull;
    prf_slot_repeat rep = {0, 1, 0};
    u32 w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const u64 j = j0 + (u64)i;
        u32 ch = (u32)'N';
        if (j < n && j >= n_head && j < body_hi && !(j >= gap_lo && j < gap_hi)) {
            u64 src = j;
            const u64 slot = (j - body_lo) / 588ull;
            if (slot < n_slots) {
                if (slot != cur_slot) {
                    cur_slot = slot;
                    rep = prf_slot_params(seed, body_lo, slot);
                }
                if (j >= rep.p && j < rep.p + rep.span) src = rep.p + (j - rep.p) % rep.k;
            }
            ch = (0x54474341u >> (8 * prf_background(seed, src))) & 0xFFu;  // "ACGT"[code]
        }
        w[i >> 2] |= ch << (8 * (i & 3));
    }
    *reinterpret_cast<uint4 *>(asc + j0) = make_uint4(w[0], w[1], w[2], w[3]);
}

hipError_t prf_launch_standin2(hipStream_t s, uint8_t *asc, u64 n, u64 seed) {
    if (n == 0) return hipSuccess;
    const u64 n_head = n / 100 < 10000 ? n / 100 : 10000, n_tail = n_head;
    u64 gap_lo = 0, gap_hi = 0;
    if (n >= 1000000) {
        gap_lo = n / 8 * 3 + 12345;
        gap_hi = gap_lo + n / 25;
    }
    const u64 chunks = (n + 15) / 16;
    hipLaunchKernelGGL(prf_standin2_kernel, dim3((u32)((chunks + 255) / 256)), dim3(256), 0, s, asc, n, seed, n_head, n_tail, gap_lo,
                       gap_hi);
    return hipGetLastError();
}

hipError_t prf_launch_synth(hipStream_t s, uint8_t *asc, u64 n, u64 seed) {
    if (n == 0) return hipSuccess;
    const u64 chunks = (n + 15) / 16;
    hipLaunchKernelGGL(prf_synth_kernel, dim3((u32)((chunks + 255) / 256)), dim3(256), 0, s, asc, n, seed);
    return hipGetLastError();
}

hipError_t prf_launch_pack_linear(hipStream_t s, const uint8_t *asc, u64 nwords, u64 *H, u64 *L, u64 *X,
                                  u64 *bad_pos, u64 *exotic) {
    const u32 bs = 256;
    const u64 nb = (nwords + bs - 1) / bs;
    hipLaunchKernelGGL(prf_pack_linear_kernel, dim3((u32)nb), dim3(bs), 0, s, asc, nwords, H, L, X, bad_pos, exotic);
    return hipGetLastError();
}

hipError_t prf_launch_fill_u64(hipStream_t s, u64 *p, u64 n, u64 v) {
    if (n == 0) return hipSuccess;
    u64 nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(prf_fill_u64_kernel, dim3((u32)nb), dim3(256), 0, s, p, n, v);
    return hipGetLastError();
}
