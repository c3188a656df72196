// verify.hip -- phase 2 of the generic path: one lane per phase-1 candidate (logic in verify_impl.h);
// plus the streaming-read probe used for the measured HBM roofline.
#include "prf_device.h"
#include "prf_host.h"
#include "verify_impl.h"

// candidates in one flat array (generic phase 1).  own_lo/own_hi restrict the rows to those "owned" by a
// position range, with the ownership rule of the fused kernel (first aligned all-match group of 8 for k with
// M(k) >= 15, the run start otherwise): used when single tiles are re-done by the generic path.
__global__ __launch_bounds__(256) void prf_verify_kernel(prf_planes pl, const u64 *__restrict__ cand, u64 cand_cap,
                                                         u32 min_repeats, u32 min_span,
                                                         const u64 *__restrict__ contig_base, u32 n_contigs,
                                                         prf_hit_dev *__restrict__ hits, u64 hit_cap,
                                                         u64 *__restrict__ counters) {
    u64 n = counters[PRF_CNT_CAND];
    if (n > cand_cap) n = cand_cap;  // overflow is reported to the host, which re-runs with a larger buffer
    prf_global_view view;
    view.P[0] = pl.H; view.P[1] = pl.L; view.P[2] = pl.X;
    for (int i = 0; i < 5; i++) view.E[i] = pl.E[i];
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const u64 rec = cand[i];
        const u64 p = rec & ((1ull << PRF_CAND_POS_BITS) - 1ull);
        const u32 k = (u32)((rec >> PRF_CAND_K_SHIFT) & 0xFFFFu);
        const u32 kind = (u32)(rec >> PRF_CAND_KIND_SHIFT);
        u64 a, b;
        if (!prf_candidate_to_run(view, p, k, kind, min_repeats, min_span, a, b)) continue;
        const u32 c = prf_contig_of(contig_base, n_contigs, a);
        const u64 slot = atomicAdd(&counters[PRF_CNT_HITS], 1ull);
        if (slot < hit_cap) {
            prf_hit_dev h;
            h.start = a - contig_base[c];
            h.end = b + k - contig_base[c];
            h.k = k;
            h.contig = c;
            hits[slot] = h;
        }
    }
}

hipError_t prf_launch_verify(hipStream_t s, const prf_planes &pl, const u64 *cand, u64 cand_cap, u32 min_repeats,
                             u32 min_span, const u64 *contig_base, u32 n_contigs, prf_hit_dev *hits, u64 hit_cap,
                             u64 *counters) {
    hipLaunchKernelGGL(prf_verify_kernel, dim3(1024), dim3(256), 0, s, pl, cand, cand_cap, min_repeats, min_span,
                       contig_base, n_contigs, hits, hit_cap, counters);
    return hipGetLastError();
}

// streaming-read probe for the measured HBM roofline: 16 B per lane, grid-stride
__global__ __launch_bounds__(256) void prf_hbm_read_kernel(const uint4 *__restrict__ p, u64 n16, u32 *__restrict__ sink) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    u32 acc = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x9E3779B9u) sink[0] = acc;  // keeps the loads alive, practically never stores
}

hipError_t prf_launch_hbm_read(hipStream_t s, const void *p, u64 bytes, u32 *sink) {
    hipLaunchKernelGGL(prf_hbm_read_kernel, dim3(256 * 8), dim3(256), 0, s, (const uint4 *)p, bytes / 16, sink);
    return hipGetLastError();
}

// ---- rows (24 bytes) -> 8-byte wire words for the multi-GPU gather (include/prf.h, prf_last_hits_packed_to_device):
// [63:41] tile (global position / 65536), [40:25] start in the tile, [24:9] span (end - start, clipped to 65535), [8:0] motif size.
// A row whose span does not fit 16 bits also goes, whole, to the side list behind the count word.
__global__ __launch_bounds__(256) void prf_pack_rows_kernel(const prf_hit_dev *__restrict__ rows, u64 n, const u64 *__restrict__ contig_base,
                                                            u64 *__restrict__ dst, u64 cap, u64 side_cap, u64 *__restrict__ side_cnt) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || i >= cap) return;
    const prf_hit_dev h = rows[i];
    const u64 gpos = contig_base[h.contig] + h.start, span = h.end - h.start;
    const u64 s16 = span < 65535ull ? span : 65535ull;
    dst[i] = ((gpos >> 16) << 41) | ((gpos & 0xFFFFull) << 25) | (s16 << 9) | (u64)(h.k & 511u);
    if (span >= 65535ull) {
        const u64 j = atomicAdd(side_cnt, 1ull);
        if (j < side_cap) {
            u64 *o = dst + cap + 1 + 3 * j;
            o[0] = h.start;
            o[1] = h.end;
            o[2] = (u64)h.k | ((u64)h.contig << 32);
        }
    }
}

// behind the pack kernel on the same stream: the count word (rows | long rows << 40) goes into the buffer, the number of long
// rows to mapped host memory -- no copy calls, one wait on the host
__global__ void prf_pack_finish_kernel(u64 *side_cnt, u64 n, u64 *count_word, u64 *host_word) {
    const u64 c = *side_cnt;
    *count_word = n | (c << 40);
    *host_word = c;
}

// The same behind a scan that is still running (prf_scan_genome_async_packed): the row count is not known on the host, the
// kernels read it from the scan's counter block; a count beyond the buffer's capacity becomes a poisoned count word.
__global__ __launch_bounds__(256) void prf_pack_rows_dev_kernel(const prf_hit_dev *__restrict__ rows, const u64 *__restrict__ n_ptr,
                                                                const u64 *__restrict__ contig_base, u64 *__restrict__ dst, u64 cap, u64 side_cap,
                                                                u64 *__restrict__ side_cnt) {
    const u64 n = *n_ptr;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n && i < cap; i += stride) {
        const prf_hit_dev h = rows[i];
        const u64 gpos = contig_base[h.contig] + h.start, span = h.end - h.start;
        const u64 s16 = span < 65535ull ? span : 65535ull;
        dst[i] = ((gpos >> 16) << 41) | ((gpos & 0xFFFFull) << 25) | (s16 << 9) | (u64)(h.k & 511u);
        if (span >= 65535ull) {
            const u64 j = atomicAdd(side_cnt, 1ull);
            if (j < side_cap) {
                u64 *o = dst + cap + 1 + 3 * j;
                o[0] = h.start;
                o[1] = h.end;
                o[2] = (u64)h.k | ((u64)h.contig << 32);
            }
        }
    }
}
__global__ void prf_pack_finish_dev_kernel(u64 *side_cnt, const u64 *n_ptr, u64 cap, u64 side_cap, u64 *count_word) {
    const u64 c = *side_cnt, n = *n_ptr;
    *count_word = (n > cap || c > side_cap) ? ~0ull : (n | (c << 40));   // all ones: the buffer was too small (multi_gpu.unpack_rows raises)
    *side_cnt = 0;
}

hipError_t prf_launch_pack_rows_dev(hipStream_t s, const prf_hit_dev *rows, const u64 *n_ptr, u64 cap, const u64 *contig_base, u64 *dst,
                                    u64 side_cap, u64 *side_cnt) {
    const u64 blocks = (cap + 255) / 256;
    hipLaunchKernelGGL(prf_pack_rows_dev_kernel, dim3((u32)(blocks < 4096 ? (blocks ? blocks : 1) : 4096)), dim3(256), 0, s, rows, n_ptr, contig_base,
                       dst, cap, side_cap, side_cnt);
    hipLaunchKernelGGL(prf_pack_finish_dev_kernel, dim3(1), dim3(1), 0, s, side_cnt, n_ptr, cap, side_cap, dst + cap);
    return hipGetLastError();
}

hipError_t prf_launch_pack_rows(hipStream_t s, const prf_hit_dev *rows, u64 n, const u64 *contig_base, u64 *dst, u64 cap, u64 side_cap,
                                u64 *side_cnt, u64 *host_word) {
    if (n) hipLaunchKernelGGL(prf_pack_rows_kernel, dim3((u32)((n + 255) / 256)), dim3(256), 0, s, rows, n, contig_base, dst, cap, side_cap, side_cnt);
    hipLaunchKernelGGL(prf_pack_finish_kernel, dim3(1), dim3(1), 0, s, side_cnt, n, dst + cap, host_word);
    return hipGetLastError();
}
