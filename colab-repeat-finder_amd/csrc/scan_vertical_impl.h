// scan_vertical_impl.h -- phase 1, fast path: bit-sliced ("vertical") scan kernels for gfx950.
//
// What it replaces: the same L x n_k calls of PerfectRepeatTracker.advance()
// (reference utils/perfect_repeat_tracker.py:43-61) as scan_generic.hip, but organised so that the
// shift by k costs no instruction at all.
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
//
// Work split.  One workgroup per tile; its NW waves share the LDS image and split the motif sizes into
// contiguous chunks of roughly equal cost.  All indices are compile-time (the kernel is a template
// over kmin,kmax,min_repeats,min_span), so every operand is a register and the inner loop per
// (row, k) is   m = (H ^ H') | (L ^ L')   -- three VALU operations for 2048 positions x 1 wave.
//
// Candidate rules (exactness argument in DESIGN.md):
//  * k with M(k) >= 15 ("group" path): a run of >= 15 matches contains an aligned group of 8 rows that
//    all match.  Per group the 8 mismatch words are OR-ed; a zero bit whose previous group was not
//    all-match (or that is the first group of its stream) is a candidate (kind GROUP).
//  * k with M(k) < 15 ("exact" path): sliding OR over exactly M(k) rows; a zero bit whose previous row
//    is a mismatch (or that is row 0 of its stream) is a candidate (kind START).
// Phase 2 (verify.hip) turns candidates into rows; spurious first-of-stream candidates die there.
// Candidates go to a private slab per (tile, wave): no atomics on the hot path.
#pragma once
#include <utility>

#include "prf_host.h"
#include "scan_vertical.h"

namespace prf_vertical {


constexpr int T = 32;      // rows (= positions) per stream
constexpr int RG = T / 4;  // row groups of 4 rows = one 16-byte slot per lane

template <int KMIN, int KMAX, int R, int SPAN>
struct Spec {
    static constexpr int kmin = KMIN, kmax = KMAX, r = R, span = SPAN;
    static constexpr int M(int k) {
        long long a = (long long)(R - 1) * k, b = (long long)SPAN - k;
        long long m = a > b ? a : b;
        return m > (1 << 24) ? (1 << 24) : (int)m;
    }
    static constexpr bool small(int k) { return M(k) < 15; }
    static constexpr int cost(int k) { return small(k) ? 55 : 22; }  // ~VALU operations x10 per row
    static constexpr int total_cost() {
        int s = 0;
        for (int k = KMIN; k <= KMAX; k++) s += cost(k);
        return s;
    }
    static constexpr int nw_raw = (total_cost() + 349) / 350;
    static constexpr int NW = nw_raw < 1 ? 1 : (nw_raw > 16 ? 16 : nw_raw);
    static constexpr int wave_of(int k) {
        int before = 0;
        for (int j = KMIN; j < k; j++) before += cost(j);
        int w = (int)(((long long)before * NW) / total_cost());
        return w > NW - 1 ? NW - 1 : w;
    }
    static constexpr int wave_lo(int w) {
        for (int k = KMIN; k <= KMAX; k++)
            if (wave_of(k) == w) return k;
        return 1;
    }
    static constexpr int wave_hi(int w) {
        for (int k = KMAX; k >= KMIN; k--)
            if (wave_of(k) == w) return k;
        return 0;
    }
    static constexpr bool any_small(int lo, int hi) {
        for (int k = lo; k <= hi; k++)
            if (small(k)) return true;
        return false;
    }
    static constexpr bool any_group(int lo, int hi) {
        for (int k = lo; k <= hi; k++)
            if (!small(k)) return true;
        return false;
    }
    // furthest row offset an exact-path k of [lo,hi] looks at: (M-1) ahead, then k ahead of that
    static constexpr int small_reach(int lo, int hi) {
        int m = 0;
        for (int k = lo; k <= hi; k++)
            if (small(k) && M(k) - 1 + k > m) m = M(k) - 1 + k;
        return m;
    }
    static constexpr int pmax() {
        int p = T - 1 + KMAX;
        int s = T - 1 + small_reach(KMIN, KMAX);
        return p > s ? p : s;
    }
    static constexpr int J = pmax() / T;  // extra virtual lanes
    static constexpr int NC = 64 + J;
};

template <int A, class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, A + I>{}), ...);
}
// f(integral_constant<int,i>) for i in [A, B)
template <int A, int B, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B > A) static_for_impl<A>(static_cast<F &&>(f), std::make_integer_sequence<int, B - A>{});
}

struct Emit {
    u64 *slab;
    u32 cap;
    u32 cnt;       // wave-uniform
    u64 lane_pos;  // tile base + lane*32
    // every lane of the wave must call this together (the caller's branch is wave-uniform)
    __device__ __forceinline__ void push(u32 c, int row, int k, u64 kind) {
        u64 bal = __builtin_amdgcn_ballot_w64(c != 0);
        while (bal) {
            if (c) {
                const u32 b = (u32)__builtin_ctz(c);
                c &= c - 1;
                const u32 idx = cnt + __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0));
                if (idx < cap)
                    slab[idx] = (lane_pos + (u64)b * (64u * T) + (u64)row) | ((u64)k << PRF_CAND_K_SHIFT) |
                                (kind << PRF_CAND_KIND_SHIFT);
            }
            cnt += (u32)__builtin_popcountll(bal);
            bal = __builtin_amdgcn_ballot_w64(c != 0);
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

// mismatch word of one row: two operations (three with the not-ACGT plane)
template <bool HASX>
__device__ __forceinline__ u32 mis(u32 ha, u32 la, u32 xa, u32 hb, u32 lb, u32 xb) {
    const u32 m = or_xor(ha ^ hb, la, lb);
    if constexpr (HASX) return m | xa | xb;
    else return m;
}

// LDS image: plane p, row group rg, virtual lane c  ->  uint4 index
template <class S>
__device__ __forceinline__ int lds_idx(int p, int rg, int c) {
    return (p * RG + rg) * S::NC + c;
}

// ---- group path: motif sizes of [KA,KB] with M(k) >= 15, rows 8*TB .. 8*TB+7 ----
template <class S, bool HASX, int KA, int KB, int TB>
__device__ __forceinline__ void group_block(const uint4 *lds, int lane, Emit &em, u32 (&prev)[KB - KA + 1]) {
    constexpr int NP = HASX ? 3 : 2;
    constexpr int GF = (8 * TB + KA) / 4;      // first row group of the shifted window
    constexpr int GL = (8 * TB + KB + 7) / 4;  // last
    constexpr int NG = GL - GF + 1;
    u32 a[3][8];
    u32 w[3][4 * NG];
    static_for<0, NP>([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        static_for<0, 2>([&](auto hc) {
            constexpr int h = decltype(hc)::value;
            const uint4 v = lds[lds_idx<S>(p, 2 * TB + h, lane)];
            a[p][4 * h + 0] = v.x; a[p][4 * h + 1] = v.y; a[p][4 * h + 2] = v.z; a[p][4 * h + 3] = v.w;
        });
        static_for<0, NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            const uint4 v = lds[lds_idx<S>(p, (GF + g) % RG, lane + (GF + g) / RG)];
            w[p][4 * g + 0] = v.x; w[p][4 * g + 1] = v.y; w[p][4 * g + 2] = v.z; w[p][4 * g + 3] = v.w;
        });
    });
    u32 cand[KB - KA + 1];
    u32 any = 0;
    static_for<KA, KB + 1>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (!S::small(k)) {
            constexpr int o0 = 8 * TB + k - 4 * GF;  // window index of row 8*TB shifted by k
            // OR over the 8 rows of the group of (H^H')|(L^L'): 16 operations, no per-row mismatch word
            u32 o = a[0][0] ^ w[0][o0];
            o = or_xor(o, a[1][0], w[1][o0]);
            static_for<1, 8>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                o = or_xor(o, a[0][i], w[0][o0 + i]);
                o = or_xor(o, a[1][i], w[1][o0 + i]);
            });
            if constexpr (HASX) {
                static_for<0, 8>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    o = o | a[2][i] | w[2][o0 + i];
                });
            }
            const u32 c = ~o & prev[k - KA];
            prev[k - KA] = o;
            cand[k - KA] = c;
            any |= c;
        }
    });
    if (__builtin_amdgcn_ballot_w64(any != 0) != 0) {
        static_for<KA, KB + 1>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if constexpr (!S::small(k)) em.push(cand[k - KA], 8 * TB, k, PRF_KIND_GROUP);
        });
    }
}

template <class S, bool HASX, int KA, int KB>
__device__ __forceinline__ void run_group(const uint4 *lds, int lane, Emit &em) {
    u32 prev[KB - KA + 1];
#pragma unroll
    for (int i = 0; i < KB - KA + 1; i++) prev[i] = ~0u;  // first group of a stream: report, phase 2 decides
    group_block<S, HASX, KA, KB, 0>(lds, lane, em, prev);
    group_block<S, HASX, KA, KB, 1>(lds, lane, em, prev);
    group_block<S, HASX, KA, KB, 2>(lds, lane, em, prev);
    group_block<S, HASX, KA, KB, 3>(lds, lane, em, prev);
}

// candidate word of row t: rows t .. t+M-1 all match (their mismatch words OR to 0) and row t-1 does not.
// m is indexed by row+1 (m[0] = "row -1"), o3[t] = OR of rows t..t+2.
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

// ---- exact path: motif sizes of [KA,KB] with M(k) < 15, all 32 rows of the stream ----
template <class S, bool HASX, int KA, int KB>
__device__ __forceinline__ void run_small(const uint4 *lds, int lane, Emit &em) {
    constexpr int NP = HASX ? 3 : 2;
    constexpr int WT = T + S::small_reach(KA, KB);  // rows 0 .. WT-1 are read
    constexpr int NG = (WT + 3) / 4;
    u32 w[3][4 * NG];
    static_for<0, NP>([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        static_for<0, NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            const uint4 v = lds[lds_idx<S>(p, g % RG, lane + g / RG)];
            w[p][4 * g + 0] = v.x; w[p][4 * g + 1] = v.y; w[p][4 * g + 2] = v.z; w[p][4 * g + 3] = v.w;
        });
    });
    static_for<KA, KB + 1>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (S::small(k)) {
            constexpr int M = S::M(k);
            constexpr int NS = T + M - 1;  // rows whose mismatch word is needed
            u32 m[NS + 1];                 // m[t+1] = mismatch word of row t;  m[0]: "row -1", unknown -> report
            u32 o3[NS];
            u32 cand[8];
            u32 any = 0;
            m[0] = ~0u;
            static_for<0, NS>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                m[s + 1] = mis<HASX>(w[0][s], w[1][s], HASX ? w[2][s] : 0u, w[0][s + k], w[1][s + k], HASX ? w[2][s + k] : 0u);
                if constexpr (M >= 3 && s >= 2) o3[s - 2] = m[s - 1] | m[s] | m[s + 1];
                constexpr int t = s - (M - 1);
                if constexpr (t >= 0) {
                    const u32 c = start_word<M, t>(m, o3);
                    cand[t & 7] = c;
                    any |= c;
                    if constexpr ((t & 7) == 7) {
                        if (__builtin_amdgcn_ballot_w64(any != 0) != 0) {
                            static_for<0, 8>([&](auto ic) {
                                constexpr int i = decltype(ic)::value;
                                em.push(cand[i], t - 7 + i, k, PRF_KIND_START);
                            });
                        }
                        any = 0;
                    }
                }
            });
        }
    });
}

template <class S, bool HASX, int W>
__device__ __forceinline__ void run_wave(const uint4 *lds, int lane, Emit &em) {
    constexpr int lo = S::wave_lo(W), hi = S::wave_hi(W);
    if constexpr (hi >= lo) {
        if constexpr (S::any_small(lo, hi)) run_small<S, HASX, lo, hi>(lds, lane, em);
        if constexpr (S::any_group(lo, hi)) run_group<S, HASX, lo, hi>(lds, lane, em);
    }
}

template <class S, bool HASX>
__device__ __forceinline__ void tile_body(uint4 *lds, const u32 *__restrict__ VH, const u32 *__restrict__ VL,
                                          const u32 *__restrict__ VX, u64 tile, int wave, int lane, Emit &em) {
    constexpr int NP = HASX ? 3 : 2;
    constexpr int NT = 64 * S::NW;
    const uint4 *ph = reinterpret_cast<const uint4 *>(VH), *pL = reinterpret_cast<const uint4 *>(VL),
                *px = reinterpret_cast<const uint4 *>(VX);
    // stage the tile (and the first J lanes again, moved up one bit) into LDS
    for (int idx = wave * 64 + lane; idx < NP * RG * 64; idx += NT) {
        const int p = idx / (RG * 64), rg = (idx / 64) % RG, l = idx % 64;
        const uint4 *src = p == 0 ? ph : (p == 1 ? pL : px);
        const uint4 v = src[(tile * RG + rg) * 64 + l];
        lds[lds_idx<S>(p, rg, l)] = v;
        if (l < S::J) {
            const uint4 nx = src[((tile + 1) * RG + rg) * 64 + l];
            uint4 r;
            r.x = (v.x >> 1) | (nx.x << 31);
            r.y = (v.y >> 1) | (nx.y << 31);
            r.z = (v.z >> 1) | (nx.z << 31);
            r.w = (v.w >> 1) | (nx.w << 31);
            lds[lds_idx<S>(p, rg, 64 + l)] = r;
        }
    }
    __syncthreads();
    static_for<0, S::NW>([&](auto wc) {
        constexpr int W = decltype(wc)::value;
        if (wave == W) run_wave<S, HASX, W>(lds, lane, em);
    });
}

template <class S, bool HASX>
__global__ __launch_bounds__(64 * S::NW) void prf_vscan_kernel(const u32 *__restrict__ VH, const u32 *__restrict__ VL,
                                                               const u32 *__restrict__ VX,
                                                               const u32 *__restrict__ tile_list,
                                                               u64 *__restrict__ slabs, u32 *__restrict__ slab_counts,
                                                               u32 slab_cap, u64 *__restrict__ counters) {
    __shared__ uint4 lds[(HASX ? 3 : 2) * RG * S::NC];
    const u64 tile = tile_list[blockIdx.x];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    const u64 unit = tile * S::NW + (u64)wave;
    Emit em;
    em.slab = slabs + unit * slab_cap;
    em.cap = slab_cap;
    em.cnt = 0;
    em.lane_pos = tile * PRF_TILE + (u64)lane * T;
    tile_body<S, HASX>(lds, VH, VL, VX, tile, wave, lane, em);
    if (lane == 0) {
        slab_counts[unit] = em.cnt;
        if (em.cnt > slab_cap) atomicMax(&counters[PRF_CNT_SLAB_OVF], (u64)em.cnt);
    }
}

// tile_list: first n_clean tiles without any not-ACGT position in reach, then n_mixed tiles with some.
// Tiles of nothing but not-ACGT positions are in neither list (no run can start there).
template <class S>
hipError_t launch_spec(hipStream_t s, const prf_vplanes &vp, u64 *slabs, u32 *slab_counts, u32 slab_cap, u64 *counters) {
    if (vp.n_clean)
        hipLaunchKernelGGL((prf_vscan_kernel<S, false>), dim3(vp.n_clean), dim3(64 * S::NW), 0, s, vp.VH, vp.VL, vp.VX,
                           vp.tile_list, slabs, slab_counts, slab_cap, counters);
    if (vp.n_mixed)
        hipLaunchKernelGGL((prf_vscan_kernel<S, true>), dim3(vp.n_mixed), dim3(64 * S::NW), 0, s, vp.VH, vp.VL, vp.VX,
                           vp.tile_list + vp.n_clean, slabs, slab_counts, slab_cap, counters);
    return hipGetLastError();
}

}  // namespace prf_vertical

// Defines the registry entry of one compiled parameter set (one translation unit per set, see Makefile).
#define PRF_DEFINE_VSPEC(KMIN, KMAX, R, SPAN)                                                                   \
    extern "C" prf_ventry prf_ventry_##KMIN##_##KMAX##_##R##_##SPAN = {                                     \
        prf_vspec{KMIN, KMAX, R, SPAN, (u32)prf_vertical::Spec<KMIN, KMAX, R, SPAN>::NW, 2u, 0},                   \
        &prf_vertical::launch_spec<prf_vertical::Spec<KMIN, KMAX, R, SPAN>>};
