// api.cpp -- the C ABI of libprf (include/prf.h): contexts, genome residency, scan orchestration.
//
// Host-side shape of one scan (what replaces the body of the reference's detect_repeats(),
// perfect_repeat_finder.py:33-81):
//   fused path (kmax <= 480): two launches -- the scan kernel (persistent workgroups: scan + verify + the rows of every
//   tile sorted into its slab) and the row gather (slabs -> ONE array sorted by (contig, start, end), as the reference
//   sorts its dict, :81; its last workgroup posts the counters to mapped host memory, where the host polls the scan's
//   serial number).  Generic path (any k): memset counters -> candidate kernel -> verify kernel -> copy back two
//   counters -> host sort.  Either way: grow buffers and repeat on overflow, then copy the rows back.
// Everything runs on the context's own HIP stream; timings are HIP events on that stream.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/prf.h"
#include "prf_host.h"
#include "scan_vertical.h"

static thread_local std::string g_err;

static const u64 PRF_FRONT_PAD = 8;  // readable words in front of every linear plane (X = all ones there)

static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

// for the other host-side translation units of the library
int prf_set_error(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess)                                                                                 \
            return fail(e_ == hipErrorOutOfMemory ? PRF_ENOMEM : PRF_EHIP, "%s failed: %s (%s:%d)", #expr,    \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                           \
    } while (0)

struct prf_ctx {
    int dev = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // fused path: one event pair per scan, in a ring, so that the kernel times of the last PRF_TIMING_RING scans can
    // be read after a timing loop (prf_scan_timings) instead of waiting for the events inside every scan
    hipEvent_t ring[3 * PRF_TIMING_RING] = {};  // per scan: before the scan kernel, between the two kernels, after the gather
    u64 *d_counters = nullptr;   // generic path + packer
    u64 *h_counters = nullptr;   // pinned, device-mapped: the fused path's last kernel writes the counters here
    u64 *h_counters_dev = nullptr;  // device address of h_counters
    u64 *d_vcounters = nullptr;  // fused path: two counter blocks used alternately (the idle one is cleared on the device)
    u64 *d_side_cnt = nullptr;   // pipelined wire hand-off: long rows packed so far (zero between packs)
    void *lit_scratch = nullptr;  // the literal lane's sort scratch (scan_literal.hip::prf_lit_sort_unique): kept between calls
    size_t lit_scratch_bytes = 0;
    hipEvent_t ev_handoff = nullptr;  // prf_stream_wait_for
    u32 parity = 0;
    u64 scan_seq = 0;
    // generic path scratch
    u64 *d_cand = nullptr;
    u64 cand_cap = 0;
    prf_hit_dev *d_hits = nullptr;  // flat rows: generic path output, or the compacted rows of the fused path
    u64 hit_cap = 0;
    prf_hit_dev *sink = nullptr;    // caller-owned device array the rows go to instead (prf_set_row_sink)
    u64 sink_cap = 0;
    const prf_hit_dev *last_rows = nullptr;  // where the rows of the last scan are
    // pipelined scans (prf_scan_genome_async / prf_scan_wait): two slots used alternately, each with its own host
    // counter block and row array; slot 0 shares them with the synchronous path
    struct async_slot {
        u64 seq = 0;            // scan in this slot (0: free)
        u64 *h = nullptr;       // mapped host counter block (+ serial number word)
        u64 *h_dev = nullptr;
        prf_hit_dev *rows = nullptr;
        u64 positions = 0;
        u32 tiles = 0;
        u32 kmax = 0;
    } slot[2];
    u64 async_n = 0;
    u64 *h_async = nullptr;         // slot 1's counter block
    prf_hit_dev *d_hits_async = nullptr;
    u64 hit_cap_async = 0;
    // fused (bit-sliced) path scratch: one row slab and one row count per launch slot (= scanned tile)
    u64 *d_slabs = nullptr;         // 8-byte rows (scan_vertical.h)
    u64 *d_long_ends = nullptr;     // per launch slot: true ends of the rows whose span is clipped in the 8-byte form
    u32 *d_slab_count = nullptr;
    u32 *d_block_sum = nullptr;     // rows per PRF_GATHER_SLOTS launch slots; zero between scans (the gather clears it)
    u64 slab_slots = 0;
    u32 slab_cap = 0;
    bool last_sorted = true;        // the rows of the last scan left the device sorted by (contig, start, end)
    u64 *stamps_buf = nullptr;      // diagnostic (PRF_STAMPS) builds only
    u32 stamps_n = 0;
    // where the rows of the last scan are
    u64 last_nhits = 0;
    u32 last_kmax = 0;              // largest motif size of the last scan (the 8-byte wire rows hold 9 bits)
};

struct prf_genome {
    prf_ctx *ctx = nullptr;
    std::vector<u64> base, len;
    u64 positions = 0;   // sum of contig lengths
    u64 G = 0;           // positions in the global coordinate space incl. gaps and the sentinel tile
    u64 nwords = 0;      // G / 64
    u64 padw = 0;        // readable words past nwords in every linear plane
    u32 kmax_hint = 0;
    u64 *H = nullptr, *L = nullptr, *X = nullptr;  // point PRF_FRONT_PAD words into their allocations
    // symbols outside ACGTN (ordinary symbols to the reference): five planes of their codes, present only if the input has
    // any; the tiles with such a symbol in reach (class 3) are scanned by the generic kernels
    u64 *E[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    u64 **d_E = nullptr;                            // the five pointers, on the device (fused kernel's general routine)
    std::vector<std::pair<u64, u64>> exotic_tiles;  // merged [first, last) ranges of class-3 tiles, ascending
    u64 *d_base = nullptr;
    uint4 *d_tile_info = nullptr;  // per tile: {contig, 0, contig base lo, hi}
    prf_vplanes vp;      // bit-sliced copy for scan_vertical
    // active selection (prf_genome_select): the scans of this genome cover only these tiles.  Off: the whole genome.
    bool sel_on = false;
    u32 *d_sel_list = nullptr;
    u32 sel_n = 0, sel_flat = ~0u;
    u64 sel_positions = 0;
    std::vector<std::pair<u64, u64>> sel_tiles;  // merged [first, last) tile ranges, ascending
};

// what a scan of the genome launches
struct launch_view {
    const u32 *list;
    u32 n, flat;
    u64 positions;
};
static launch_view active_launch(const prf_genome *g) {
    if (g->sel_on) return launch_view{g->d_sel_list, g->sel_n, g->sel_flat, g->sel_positions};
    return launch_view{g->vp.launch_list, g->vp.n_launch, g->vp.flat_base, g->positions};
}

extern "C" {

int prf_abi_version(void) { return PRF_ABI_VERSION; }

const char *prf_last_error(void) { return g_err.c_str(); }

int prf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int prf_open(int device_id, prf_ctx **out) {
    if (!out) return fail(PRF_EINVAL, "prf_open: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(PRF_ENODEV, "prf_open: no HIP device visible (%s); libprf has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= n) return fail(PRF_EINVAL, "prf_open: device %d out of range [0,%d)", device_id, n);
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(PRF_ENODEV, "prf_open: device %d is %s; libprf is built for gfx950 (MI355X) only", device_id,
                    prop.gcnArchName);
    HIPCHK(hipSetDevice(device_id));
    prf_ctx *c = new (std::nothrow) prf_ctx();
    if (!c) return fail(PRF_ENOMEM, "prf_open: out of host memory");
    c->dev = device_id;
    struct guard_t {  // a failure below must not leak the half-built context
        prf_ctx *c;
        ~guard_t() { if (c) prf_close(c); }
    } guard{c};
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (auto &ev : c->ev) HIPCHK(hipEventCreate(&ev));
    for (auto &ev : c->ring) HIPCHK(hipEventCreate(&ev));
    HIPCHK(hipMalloc((void **)&c->d_counters, PRF_CNT_N * sizeof(u64)));
    HIPCHK(hipHostMalloc((void **)&c->h_counters, (PRF_CNT_N + 8) * sizeof(u64), hipHostMallocMapped | hipHostMallocCoherent));
    // recycled pinned blocks are not zeroed, and the scan serial numbers restart at 1 for every context: a stale
    // serial number left by a closed context must never look like a finished scan
    memset(c->h_counters, 0, (PRF_CNT_N + 8) * sizeof(u64));
    HIPCHK(hipHostGetDevicePointer((void **)&c->h_counters_dev, c->h_counters, 0));
    HIPCHK(hipMalloc((void **)&c->d_vcounters, 2 * PRF_CNT_N * sizeof(u64)));
    HIPCHK(hipMemset(c->d_vcounters, 0, 2 * PRF_CNT_N * sizeof(u64)));
    HIPCHK(hipMalloc((void **)&c->d_side_cnt, sizeof(u64)));
    HIPCHK(hipMemset(c->d_side_cnt, 0, sizeof(u64)));
    HIPCHK(hipEventCreateWithFlags(&c->ev_handoff, hipEventDisableTiming));
    guard.c = nullptr;
    *out = c;
    return PRF_OK;
}

void prf_close(prf_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->dev);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->d_counters);
    (void)hipHostFree(c->h_counters);
    (void)hipHostFree(c->h_async);
    (void)hipFree(c->d_hits_async);
    (void)hipFree(c->d_vcounters);
    (void)hipFree(c->d_side_cnt);
    (void)hipFree(c->lit_scratch);
    if (c->ev_handoff) (void)hipEventDestroy(c->ev_handoff);
    (void)hipFree(c->d_cand);
    (void)hipFree(c->d_hits);
    (void)hipFree(c->d_slabs);
    (void)hipFree(c->d_long_ends);
    (void)hipFree(c->d_slab_count);
    (void)hipFree(c->d_block_sum);
    for (auto &ev : c->ev)
        if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : c->ring)
        if (ev) (void)hipEventDestroy(ev);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

void prf_genome_free(prf_genome *g) {
    if (!g) return;
    if (g->ctx) (void)hipSetDevice(g->ctx->dev);
    if (g->H) (void)hipFree(g->H - PRF_FRONT_PAD);
    if (g->L) (void)hipFree(g->L - PRF_FRONT_PAD);
    if (g->X) (void)hipFree(g->X - PRF_FRONT_PAD);
    for (u64 *e : g->E)
        if (e) (void)hipFree(e - PRF_FRONT_PAD);
    (void)hipFree(g->d_E);
    (void)hipFree(g->d_base);
    (void)hipFree(g->d_tile_info);
    (void)hipFree(g->vp.VH);
    (void)hipFree(g->vp.VL);
    (void)hipFree(g->vp.tile_class);
    (void)hipFree(g->vp.launch_list);
    (void)hipFree(g->d_sel_list);
    delete g;
}

uint64_t prf_genome_positions(const prf_genome *g) { return g ? g->positions : 0; }

// contigs[i].ascii == nullptr with seeds != nullptr: contig i is generated on the device from seeds[i]
// recipe (with seeds): 1 = uniform background, 2 = stand-in recipe 2 (N blocks + planted repeats)
static int genome_load_impl(prf_ctx *c, const prf_contig *contigs, int n_contigs, uint32_t kmax_hint, prf_genome **out,
                            const uint64_t *seeds = nullptr, int recipe = 1) {
    if (!c || !out || n_contigs < 0 || (n_contigs > 0 && !contigs))
        return fail(PRF_EINVAL, "prf_genome_load: bad arguments");
    *out = nullptr;
    HIPCHK(hipSetDevice(c->dev));
    if (kmax_hint < 1) kmax_hint = 1;
    if (kmax_hint > 60000) return fail(PRF_EUNSUPPORTED, "prf_genome_load: kmax_hint %u > 60000", kmax_hint);
    prf_genome *g = new (std::nothrow) prf_genome();
    if (!g) return fail(PRF_ENOMEM, "prf_genome_load: out of host memory");
    struct guard_t {
        prf_genome *g;
        ~guard_t() { if (g) prf_genome_free(g); }
    } guard{g};
    g->ctx = c;
    g->kmax_hint = kmax_hint;
    const u64 gap = (u64)kmax_hint + 64;
    u64 cur = 0;
    for (int i = 0; i < n_contigs; i++) {
        if (contigs[i].len && !contigs[i].ascii && !seeds) return fail(PRF_EINVAL, "prf_genome_load: contig %d has NULL data", i);
        g->base.push_back(cur);
        g->len.push_back(contigs[i].len);
        g->positions += contigs[i].len;
        cur = (cur + contigs[i].len + gap + PRF_TILE - 1) / PRF_TILE * PRF_TILE;
    }
    if (cur == 0) cur = PRF_TILE;
    g->G = cur + PRF_TILE;  // one all-gap sentinel tile: every walk to the right ends inside the arrays
    if (g->G >= (1ull << 40)) return fail(PRF_EUNSUPPORTED, "prf_genome_load: input too large (2^40 positions)");  // row keys: 40-bit offsets
    g->nwords = g->G / 64;
    g->padw = kmax_hint / 64 + 8;

    uint8_t *asc = nullptr;
    HIPCHK(hipMalloc((void **)&asc, g->G));
    struct asc_guard_t {
        uint8_t *p;
        ~asc_guard_t() { (void)hipFree(p); }
    } asc_guard{asc};
    HIPCHK(prf_launch_fill_u64(c->stream, (u64 *)asc, g->G / 8, 0x4E4E4E4E4E4E4E4Eull));  // 'N' everywhere
    for (int i = 0; i < n_contigs; i++) {
        if (!contigs[i].len) continue;
        if (seeds && recipe == 2)
            HIPCHK(prf_launch_standin2(c->stream, asc + g->base[i], contigs[i].len, seeds[i]));
        else if (seeds)  // the generator writes whole 16-byte groups; the tail beyond len is 'N' again, like the gap
            HIPCHK(prf_launch_synth(c->stream, asc + g->base[i], contigs[i].len, seeds[i]));
        else
            HIPCHK(hipMemcpyAsync(asc + g->base[i], contigs[i].ascii, contigs[i].len, hipMemcpyHostToDevice, c->stream));
    }
    const u64 tot = PRF_FRONT_PAD + g->nwords + g->padw;
    {
        u64 *p = nullptr;
        HIPCHK(hipMalloc((void **)&p, tot * 8));
        g->H = p + PRF_FRONT_PAD;
        HIPCHK(hipMalloc((void **)&p, tot * 8));
        g->L = p + PRF_FRONT_PAD;
        HIPCHK(hipMalloc((void **)&p, tot * 8));
        g->X = p + PRF_FRONT_PAD;
    }
    HIPCHK(hipMemsetAsync(g->H - PRF_FRONT_PAD, 0, PRF_FRONT_PAD * 8, c->stream));
    HIPCHK(hipMemsetAsync(g->L - PRF_FRONT_PAD, 0, PRF_FRONT_PAD * 8, c->stream));
    HIPCHK(hipMemsetAsync(g->X - PRF_FRONT_PAD, 0xFF, PRF_FRONT_PAD * 8, c->stream));
    HIPCHK(hipMemsetAsync(c->d_counters, 0xFF, PRF_CNT_N * sizeof(u64), c->stream));
    HIPCHK(hipMemsetAsync(c->d_counters + PRF_CNT_EXOTIC, 0, sizeof(u64), c->stream));
    HIPCHK(prf_launch_pack_linear(c->stream, asc, g->nwords, g->H, g->L, g->X, c->d_counters + PRF_CNT_BADPOS,
                                  c->d_counters + PRF_CNT_EXOTIC));
    HIPCHK(hipMemsetAsync(g->H + g->nwords, 0, g->padw * 8, c->stream));
    HIPCHK(hipMemsetAsync(g->L + g->nwords, 0, g->padw * 8, c->stream));
    HIPCHK(hipMemsetAsync(g->X + g->nwords, 0xFF, g->padw * 8, c->stream));
    HIPCHK(hipMalloc((void **)&g->d_base, sizeof(u64) * (size_t)(n_contigs > 0 ? n_contigs : 1)));
    if (n_contigs > 0)
        HIPCHK(hipMemcpyAsync(g->d_base, g->base.data(), sizeof(u64) * (size_t)n_contigs, hipMemcpyHostToDevice, c->stream));
    {   // the contig of every tile (contigs start on tile boundaries; the gap tiles behind a contig count as its own)
        const u64 ntiles = g->G / PRF_TILE;
        std::vector<uint4> info(ntiles, make_uint4(0, 0, 0, 0));
        for (size_t ci = 0; ci < g->base.size(); ci++) {
            const u64 t0 = g->base[ci] / PRF_TILE, t1 = ci + 1 < g->base.size() ? g->base[ci + 1] / PRF_TILE : ntiles;
            for (u64 t = t0; t < t1; t++) info[t] = make_uint4((u32)ci, 0u, (u32)g->base[ci], (u32)(g->base[ci] >> 32));
        }
        HIPCHK(hipMalloc((void **)&g->d_tile_info, sizeof(uint4) * ntiles));
        HIPCHK(hipMemcpy(g->d_tile_info, info.data(), sizeof(uint4) * ntiles, hipMemcpyHostToDevice));
    }
    // bit-sliced copy for the vertical kernel (built from the ASCII while it is still resident)
    {
        int rc = prf_vertical_pack(c->stream, asc, g->G, &g->vp);
        if (rc != hipSuccess) return fail(rc == (int)hipErrorOutOfMemory ? PRF_ENOMEM : PRF_EHIP, "vertical pack failed: %s",
                                          hipGetErrorString((hipError_t)rc));
    }
    HIPCHK(hipMemcpyAsync(c->h_counters, c->d_counters, PRF_CNT_N * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    const u64 bad = c->h_counters[PRF_CNT_BADPOS];
    if (bad != ~0ull) {
        size_t ci = std::upper_bound(g->base.begin(), g->base.end(), bad) - g->base.begin() - 1;
        return fail(PRF_ESYMBOL,
                    "unsupported symbol at contig %zu position %llu: only letters can be packed (A, C, G, T, N and -- as ordinary "
                    "symbols, like the reference -- any other letter, in either case); libprf refuses other bytes instead of guessing",
                    ci, (unsigned long long)(bad - g->base[ci]));
    }
    if (c->h_counters[PRF_CNT_EXOTIC]) {
        // letters other than A, C, G, T, N: the planes of their codes, and the tile ranges the generic kernels take
        for (int i = 0; i < 5; i++) {
            u64 *p = nullptr;
            HIPCHK(hipMalloc((void **)&p, tot * 8));
            g->E[i] = p + PRF_FRONT_PAD;
            HIPCHK(hipMemsetAsync(p, 0, PRF_FRONT_PAD * 8, c->stream));
            HIPCHK(hipMemsetAsync(g->E[i] + g->nwords, 0, g->padw * 8, c->stream));
        }
        HIPCHK(prf_launch_pack_exotic(c->stream, asc, g->nwords, g->E));
        HIPCHK(hipMalloc((void **)&g->d_E, 5 * sizeof(u64 *)));
        HIPCHK(hipMemcpyAsync(g->d_E, g->E, 5 * sizeof(u64 *), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        const std::vector<unsigned char> &cls = g->vp.h_class;
        for (u64 t = 0; t + 1 < cls.size(); t++) {
            if (cls[t] != 3) continue;
            if (!g->exotic_tiles.empty() && g->exotic_tiles.back().second == t) g->exotic_tiles.back().second = t + 1;
            else g->exotic_tiles.emplace_back(t, t + 1);
        }
    }
    guard.g = nullptr;
    *out = g;
    return PRF_OK;
}

int prf_genome_load(prf_ctx *c, const prf_contig *contigs, int n_contigs, uint32_t kmax_hint, prf_genome **out) {
    try {
        return genome_load_impl(c, contigs, n_contigs, kmax_hint, out);
    } catch (const std::bad_alloc &) {
        return fail(PRF_ENOMEM, "prf_genome_load: out of host memory");
    } catch (...) {
        return fail(PRF_EHIP, "prf_genome_load: unexpected exception");
    }
}

int prf_genome_synth(prf_ctx *c, const uint64_t *lens, const uint64_t *seeds, int n_contigs, uint32_t kmax_hint, prf_genome **out) {
    if (n_contigs < 0 || (n_contigs > 0 && (!lens || !seeds))) return fail(PRF_EINVAL, "prf_genome_synth: bad arguments");
    try {
        std::vector<prf_contig> cs((size_t)(n_contigs > 0 ? n_contigs : 1));
        for (int i = 0; i < n_contigs; i++) cs[i] = prf_contig{nullptr, lens[i]};
        return genome_load_impl(c, cs.data(), n_contigs, kmax_hint, out, seeds);
    } catch (const std::bad_alloc &) {
        return fail(PRF_ENOMEM, "prf_genome_synth: out of host memory");
    } catch (...) {
        return fail(PRF_EHIP, "prf_genome_synth: unexpected exception");
    }
}

int prf_genome_standin(prf_ctx *c, const uint64_t *lens, const uint64_t *seeds, int n_contigs, uint32_t kmax_hint, prf_genome **out) {
    if (n_contigs < 0 || (n_contigs > 0 && (!lens || !seeds))) return fail(PRF_EINVAL, "prf_genome_standin: bad arguments");
    try {
        std::vector<prf_contig> cs((size_t)(n_contigs > 0 ? n_contigs : 1));
        for (int i = 0; i < n_contigs; i++) cs[i] = prf_contig{nullptr, lens[i]};
        return genome_load_impl(c, cs.data(), n_contigs, kmax_hint, out, seeds, 2);
    } catch (const std::bad_alloc &) {
        return fail(PRF_ENOMEM, "prf_genome_standin: out of host memory");
    } catch (...) {
        return fail(PRF_EHIP, "prf_genome_standin: unexpected exception");
    }
}

// After a fused launch failed or never posted its counters the device-side counter blocks are in an unknown state
// (the kernel clears the NEXT scan's block only when its last workgroup gets there): drain the stream, clear both.
static void reset_vcounters(prf_ctx *c) {
    (void)hipStreamSynchronize(c->stream);
    (void)hipMemsetAsync(c->d_vcounters, 0, 2 * PRF_CNT_N * sizeof(u64), c->stream);
    if (c->d_block_sum)
        (void)hipMemsetAsync(c->d_block_sum, 0,
                             ((size_t)(c->slab_slots / 8 + 1) + 2 * (size_t)(c->slab_slots / (8 * PRF_GATHER_SUPER) + 2)) * sizeof(u32),
                             c->stream);
    (void)hipStreamSynchronize(c->stream);
}

// Poll the serial number the fused kernel's last workgroup writes behind the counter block (mapped host memory).
// Bounded by the wall clock (PRF_SCAN_TIMEOUT_S, default 120 s): a wedged kernel becomes PRF_EHIP, not a wedged host.
static int wait_for_seq(prf_ctx *c, u64 seq, const u64 *block = nullptr) {
    const u64 *seqp = (block ? block : c->h_counters) + PRF_CNT_N;
    static const double limit_s = [] {
        const char *e = getenv("PRF_SCAN_TIMEOUT_S");
        const double v = e ? atof(e) : 0.0;
        return v > 0.0 ? v : 120.0;
    }();
    std::chrono::steady_clock::time_point t0;
    bool timing = false;
    for (u64 spins = 1;; spins++) {
        if (__atomic_load_n(seqp, __ATOMIC_ACQUIRE) == seq) return PRF_OK;
        __builtin_ia32_pause();
        if ((spins & 0xFFFFu) == 0) {  // every ~millisecond: is the stream still alive?
            const hipError_t e = hipStreamQuery(c->stream);
            if (e == hipSuccess) {
                if (__atomic_load_n(seqp, __ATOMIC_ACQUIRE) == seq) return PRF_OK;
                reset_vcounters(c);
                return fail(PRF_EHIP, "fused scan kernel finished without posting its counters (scan %llu)", (unsigned long long)seq);
            }
            if (e != hipErrorNotReady) {
                reset_vcounters(c);
                return fail(PRF_EHIP, "fused scan kernel failed: %s", hipGetErrorString(e));
            }
            if (!timing) {
                t0 = std::chrono::steady_clock::now();
                timing = true;
            } else if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s) {
                // the device-side blocks are not touched: the kernel may still be running
                return fail(PRF_EHIP, "fused scan %llu did not post its counters within %.0f s (PRF_SCAN_TIMEOUT_S)",
                            (unsigned long long)seq, limit_s);
            }
        }
    }
}

static int scan_timings_impl(prf_ctx *c, uint64_t first_seq, uint32_t n, float *total_ms, float *scan_ms, float *gather_ms) {
    if (!c || first_seq == 0 || first_seq + n - 1 > c->scan_seq || c->scan_seq - first_seq >= PRF_TIMING_RING)
        return fail(PRF_EINVAL, "prf_scan_timings: scans %llu..%llu are not among the last %d fused scans of this context",
                    (unsigned long long)first_seq, (unsigned long long)(first_seq + n - 1), PRF_TIMING_RING);
    HIPCHK(hipSetDevice(c->dev));
    for (uint32_t i = 0; i < n; i++) {
        const u64 q = first_seq + i;
        hipEvent_t ev_a = c->ring[3 * (q % PRF_TIMING_RING)], ev_m = c->ring[3 * (q % PRF_TIMING_RING) + 1],
                   ev_b = c->ring[3 * (q % PRF_TIMING_RING) + 2];
        HIPCHK(hipEventSynchronize(ev_b));
        if (total_ms) HIPCHK(hipEventElapsedTime(&total_ms[i], ev_a, ev_b));
        if (scan_ms) HIPCHK(hipEventElapsedTime(&scan_ms[i], ev_a, ev_m));
        if (gather_ms) HIPCHK(hipEventElapsedTime(&gather_ms[i], ev_m, ev_b));
    }
    return PRF_OK;
}

int prf_scan_timings(prf_ctx *c, uint64_t first_seq, uint32_t n, float *kernel_ms) {
    if (!c || (n && !kernel_ms)) return fail(PRF_EINVAL, "prf_scan_timings: bad arguments");
    return scan_timings_impl(c, first_seq, n, kernel_ms, nullptr, nullptr);
}

int prf_scan_timings_split(prf_ctx *c, uint64_t first_seq, uint32_t n, float *scan_ms, float *gather_ms) {
    if (!c || (n && (!scan_ms || !gather_ms))) return fail(PRF_EINVAL, "prf_scan_timings_split: bad arguments");
    return scan_timings_impl(c, first_seq, n, nullptr, scan_ms, gather_ms);
}

static int ensure_buffers(prf_ctx *c, u64 want_cand, u64 want_hits) {
    if (want_cand > c->cand_cap) {
        (void)hipFree(c->d_cand);
        c->d_cand = nullptr;
        c->cand_cap = 0;
        HIPCHK(hipMalloc((void **)&c->d_cand, want_cand * sizeof(u64)));
        c->cand_cap = want_cand;
    }
    if (want_hits > c->hit_cap) {
        (void)hipFree(c->d_hits);
        c->d_hits = nullptr;
        c->hit_cap = 0;
        HIPCHK(hipMalloc((void **)&c->d_hits, want_hits * sizeof(prf_hit_dev)));
        c->hit_cap = want_hits;
    }
    return PRF_OK;
}

static int ensure_slabs(prf_ctx *c, u64 nslots, u32 cap) {
    if (nslots > c->slab_slots || cap > c->slab_cap) {
        nslots = std::max<u64>(nslots, c->slab_slots);
        cap = std::max<u32>(cap, c->slab_cap);
        HIPCHK(hipStreamSynchronize(c->stream));  // no scan in flight may still use the old buffers
        (void)hipFree(c->d_slabs);
        (void)hipFree(c->d_long_ends);
        (void)hipFree(c->d_slab_count);
        (void)hipFree(c->d_block_sum);
        c->d_slabs = nullptr;
        c->d_long_ends = nullptr;
        c->d_slab_count = nullptr;
        c->d_block_sum = nullptr;
        c->slab_slots = 0;
        c->slab_cap = 0;
        const size_t n_blocks = (size_t)(nslots / 8 + 1) + 2 * (size_t)(nslots / (8 * PRF_GATHER_SUPER) + 2);
        HIPCHK(hipMalloc((void **)&c->d_slabs, nslots * (u64)cap * sizeof(u64)));
        HIPCHK(hipMalloc((void **)&c->d_long_ends, nslots * (u64)PRF_LONG_PER_TILE * sizeof(u64)));
        HIPCHK(hipMalloc((void **)&c->d_slab_count, nslots * sizeof(u32)));
        HIPCHK(hipMalloc((void **)&c->d_block_sum, n_blocks * sizeof(u32)));
        HIPCHK(hipMemset(c->d_block_sum, 0, n_blocks * sizeof(u32)));
        c->slab_slots = nslots;
        c->slab_cap = cap;
    }
    return PRF_OK;
}

// The two launches of a fused scan: tiles -> sorted slabs, slabs -> one compact sorted array + counters to the host.
// HIP events bracket the pair.
static int launch_fused(prf_ctx *c, const prf_genome *g, const prf_vplan &plan, u32 min_repeats, u32 min_span, prf_hit_dev *rows,
                        u64 rows_cap, u32 count_row, u64 *host_counters_dev, u64 *seq_out) {
    const launch_view lv = active_launch(g);
    prf_vscan_args a;
    a.VH = g->vp.VH; a.VL = g->vp.VL;
    a.H = g->H; a.L = g->L; a.X = g->X;
    a.E = g->d_E;
    a.launch_list = lv.list; a.n_launch = lv.n; a.flat_base = lv.flat;
    a.slabs = c->d_slabs; a.long_ends = c->d_long_ends; a.slab_count = c->d_slab_count; a.block_sum = c->d_block_sum; a.slab_cap = c->slab_cap;
    // the gather takes 8 launch slots per workgroup on small launches (more workgroups in flight), 64 on large ones
    a.gather_shift = lv.n <= 8192u ? 3u : 5u;
    {   // PRF_GATHER_SHIFT (diagnostic): launch slots per gather workgroup = 1 << shift, 3 .. 6
        static const int gs_env = getenv("PRF_GATHER_SHIFT") ? atoi(getenv("PRF_GATHER_SHIFT")) : -1;
        if (gs_env >= 3 && gs_env <= 6) a.gather_shift = (u32)gs_env;
    }
    a.super_off = (lv.n + (1u << a.gather_shift) - 1u) >> a.gather_shift;  // the gather's grid
    a.min_repeats = min_repeats; a.min_span = min_span;
    a.tile_info = g->d_tile_info;
    a.counters = c->d_vcounters + (size_t)c->parity * PRF_CNT_N;
    a.dbg = nullptr;
    {
        static const u32 skip_env = getenv("PRF_SKIP") ? (u32)strtoul(getenv("PRF_SKIP"), nullptr, 0) : 0u;  // diagnostic: wrong rows, timing only
        a.skip = skip_env;
    }
#ifdef PRF_STAMPS
    {
        static u64 *dbg_buf = nullptr;
        const size_t nu = (size_t)(1u << 18) * PRF_VMAX_WAVES * 16;  // up to 262144 tiles
        if (!dbg_buf) HIPCHK(hipMalloc((void **)&dbg_buf, nu * sizeof(u64)));
        if (lv.n <= (1u << 18)) {
            HIPCHK(hipMemsetAsync(dbg_buf, 0, (size_t)lv.n * PRF_VMAX_WAVES * 16 * sizeof(u64), c->stream));
            a.dbg = dbg_buf;
            c->stamps_buf = dbg_buf;
            c->stamps_n = lv.n;
        }
    }
#endif
    a.plan = plan;
    prf_vgather_args ga;
    ga.slabs = c->d_slabs; ga.long_ends = c->d_long_ends; ga.launch_list = lv.list; ga.flat_base = lv.flat; ga.tile_info = g->d_tile_info;
    ga.slab_count = c->d_slab_count; ga.block_sum = c->d_block_sum; ga.slab_cap = c->slab_cap; ga.n_launch = lv.n;
    ga.super_off = a.super_off;
    ga.gather_shift = a.gather_shift;
    ga.rows = rows; ga.rows_cap = rows_cap; ga.count_row = count_row;
    ga.counters = a.counters;
    ga.host_counters = host_counters_dev;
    ga.seq = ++c->scan_seq;
    ga.next_counters = c->d_vcounters + (size_t)(c->parity ^ 1u) * PRF_CNT_N;
    c->parity ^= 1u;
    hipEvent_t ev_a = c->ring[3 * (ga.seq % PRF_TIMING_RING)], ev_m = c->ring[3 * (ga.seq % PRF_TIMING_RING) + 1],
               ev_b = c->ring[3 * (ga.seq % PRF_TIMING_RING) + 2];
    HIPCHK(hipEventRecord(ev_a, c->stream));
    hipError_t le = prf_vertical_launch(c->stream, a);
    if (le == hipSuccess) le = hipEventRecord(ev_m, c->stream);
    if (le == hipSuccess) le = prf_vertical_gather(c->stream, ga);
    if (le != hipSuccess) {
        reset_vcounters(c);
        return fail(PRF_EHIP, "fused scan launch failed: %s", hipGetErrorString(le));
    }
    HIPCHK(hipEventRecord(ev_b, c->stream));
    *seq_out = ga.seq;
    return PRF_OK;
}

static int literal_genome(prf_ctx *c, const prf_genome *g, u32 kmin, u32 kmax, u32 min_repeats, u32 min_span, prf_hits *out,
                          prf_scan_stats *stats);

static int scan_impl(prf_ctx *c, const prf_genome *g, uint32_t kmin, uint32_t kmax, uint32_t min_repeats,
                     uint32_t min_span, uint32_t flags, prf_hits *out, prf_scan_stats *stats) {
    if (!c || !g) return fail(PRF_EINVAL, "prf_scan_genome: NULL context or genome");
    if (g->ctx != c) return fail(PRF_EINVAL, "prf_scan_genome: genome belongs to another context");
    if (out) { out->rows = nullptr; out->n = 0; }
    // same conditions, same wording as the reference's ValueErrors (perfect_repeat_finder.py:23-30)
    if (kmin < 1) return fail(PRF_EINVAL, "min_motif_size is set to %u. It must be at least 1.", kmin);
    if (kmax < kmin) return fail(PRF_EINVAL, "max_motif_size is set to %u. It must be at least min_motif_size.", kmax);
    if (min_repeats < 1) return fail(PRF_EINVAL, "min_repeats is set to %u. It must be at least 1.", min_repeats);
    if (min_span < 1) return fail(PRF_EINVAL, "min_span is set to %u. It must be at least 1.", min_span);
    if (min_repeats > 1000000u || min_span > (1u << 30)) return fail(PRF_EINVAL, "threshold out of range");
    if (min_repeats < 2) {  // outside the closed form of the packed kernels: the literal lane on the bytes rebuilt from the planes
        HIPCHK(hipSetDevice(c->dev));
        return literal_genome(c, g, kmin, kmax, min_repeats, min_span, out, stats);
    }
    if (kmax > g->kmax_hint)
        return fail(PRF_EUNSUPPORTED, "max_motif_size %u exceeds the kmax_hint %u this genome was packed with", kmax,
                    g->kmax_hint);
    if (min_repeats > 1000000u || min_span > (1u << 30)) return fail(PRF_EINVAL, "threshold out of range");
    HIPCHK(hipSetDevice(c->dev));

    prf_planes pl{g->H, g->L, g->X, {g->E[0], g->E[1], g->E[2], g->E[3], g->E[4]}};
    prf_vplan plan;
    bool vs = !(flags & PRF_SCAN_FORCE_GENERIC) && prf_vertical_plan(kmin, kmax, min_repeats, min_span, &plan);
    const launch_view lv = active_launch(g);
    u64 want_cand = std::max<u64>(c->cand_cap, lv.positions / 8 + 65536);
    u64 want_hits = std::max<u64>(c->hit_cap, lv.positions / 32 + 65536);
    // rows per tile the slabs hold (8 bytes each): the densest 65536-position tile of the reference's golden chr22 BED has 748
    // rows at motif sizes 1-6; a tile beyond the capacity grows the slabs and scans again
    u32 slab_cap = std::max<u32>(c->slab_cap, 1024u);
    float ms01 = 0, ms12 = 0;
    u64 ncand = 0, nhits = 0;
    u32 launches = 0;
    bool sorted_on_device = false;
    for (int attempt = 0;; attempt++) {
        if (attempt > 8) return fail(PRF_EHIP, "prf_scan_genome: buffers still overflowing after 8 attempts");
        bool again = false;
        if (vs) {
            // ---- fused bit-sliced kernel (scan + verify + sorted rows per tile), then the row gather ----
            // tiles with a symbol outside ACGTN in reach are not on the launch list: the generic kernels take them below
            std::vector<std::pair<u64, u64>> exotic;
            if (!g->exotic_tiles.empty()) {
                if (!g->sel_on) exotic = g->exotic_tiles;
                else
                    for (const auto &e : g->exotic_tiles)
                        for (const auto &r : g->sel_tiles) {
                            const u64 lo = std::max(e.first, r.first), hi = std::min(e.second, r.second);
                            if (lo < hi) exotic.emplace_back(lo, hi);
                        }
            }
            if (lv.n == 0 && exotic.empty()) {  // nothing but N (or no contig at all): no tile to launch, no rows
                nhits = ncand = 0;
                launches = 0;
                sorted_on_device = true;
                if (c->sink) {
                    HIPCHK(hipMemsetAsync(c->sink + c->sink_cap, 0, sizeof(prf_hit_dev), c->stream));
                    HIPCHK(hipStreamSynchronize(c->stream));
                }
                break;
            }
            if (lv.n == 0) {  // only such tiles: no fused launch
                int rc0 = ensure_buffers(c, want_cand, want_hits);
                if (rc0) return rc0;
                nhits = ncand = 0;
                launches = 0;
                sorted_on_device = true;
            } else {
            int rc = ensure_slabs(c, lv.n, slab_cap);
            if (rc) return rc;
            rc = ensure_buffers(c, c->cand_cap, want_hits);  // the compact row array
            if (rc) return rc;
            u64 seq = 0;
            rc = launch_fused(c, g, plan, min_repeats, min_span, c->sink ? c->sink : c->d_hits, c->sink ? c->sink_cap : c->hit_cap,
                              c->sink ? 1u : 0u, c->h_counters_dev, &seq);
            if (rc) return rc;
            hipEvent_t ev_a = c->ring[3 * (seq % PRF_TIMING_RING)], ev_b = c->ring[3 * (seq % PRF_TIMING_RING) + 2];
            // The scan is over for the host when the gather's last workgroup has posted the counter block and this
            // scan's serial number in mapped host memory: poll that word instead of waiting for the stream to drain
            // (the kernel's end-of-grid handshake, the event and the wake-up cost ~5 us).
            // Everything that consumes the rows is enqueued on the same stream, hence ordered after the kernels.
            rc = wait_for_seq(c, seq);
            if (rc) return rc;
#ifdef PRF_STAMPS
            if (const char *path = getenv("PRF_STAMPS_OUT")) {
                fprintf(stderr, "[prf] plan: %u waves, %u tasks\n", plan.n_waves, plan.n_tasks);
                for (u32 w = 0; w < plan.n_waves; w++)
                    for (u32 ti = plan.wave_begin[w]; ti < plan.wave_begin[w + 1]; ti++)
                        fprintf(stderr, "[prf]   wave %u slot %u: kind %u k0 %u valid %02x stride %u\n", w, ti - plan.wave_begin[w],
                                plan.tasks[ti].kind, plan.tasks[ti].k0, plan.tasks[ti].valid, plan.tasks[ti].stride);
                if (c->stamps_buf) {
                    HIPCHK(hipStreamSynchronize(c->stream));
                    const size_t nu = (size_t)c->stamps_n * PRF_VMAX_WAVES * 16;
                    std::vector<u64> host(nu);
                    HIPCHK(hipMemcpy(host.data(), c->stamps_buf, nu * sizeof(u64), hipMemcpyDeviceToHost));
                    if (FILE *f = fopen(path, "wb")) { fwrite(host.data(), 8, nu, f); fclose(f); }
                }
            }
#endif
            // a row sink is read by the caller on streams of its own: wait until every workgroup has copied its rows
            if (c->sink) HIPCHK(hipStreamSynchronize(c->stream));
            if (stats && !(flags & PRF_SCAN_DEFER_TIMING)) {
                HIPCHK(hipEventSynchronize(ev_b));
                HIPCHK(hipEventElapsedTime(&ms01, ev_a, ev_b));
            }
            ms12 = 0;
            launches = 2u * (u32)(attempt + 1);  // (a scan that had to grow its buffers and run again shows here)
            nhits = c->h_counters[PRF_CNT_ROWS];
            sorted_on_device = c->h_counters[PRF_CNT_UNSORTED] == 0;
            ncand = 0;
            for (int sh = 0; sh < PRF_CNT_NSHARD; sh++)
                ncand += c->h_counters[PRF_CNT_SHARD0 + sh * PRF_CNT_SHARD_STRIDE + PRF_SH_CAND];
            if (getenv("PRF_DEBUG")) {
                u64 nearly = 0;
                for (int sh = 0; sh < PRF_CNT_NSHARD; sh++) nearly += c->h_counters[PRF_CNT_SHARD0 + sh * PRF_CNT_SHARD_STRIDE + PRF_SH_EARLY];
                fprintf(stderr, "[prf] fused: hits %llu cand-records %llu (verified on the spot, a list being full: %llu) hit_ovf %llu unsorted %llu ms %.4f\n",
                        (unsigned long long)nhits, (unsigned long long)ncand, (unsigned long long)nearly,
                        (unsigned long long)c->h_counters[PRF_CNT_HIT_OVF], (unsigned long long)c->h_counters[PRF_CNT_UNSORTED], ms01);
            }
            if (c->h_counters[PRF_CNT_LONG_OVF])
                return fail(PRF_EHIP, "internal: a tile reported %llu rows longer than 65534 positions (at most %u can exist)",
                            (unsigned long long)c->h_counters[PRF_CNT_LONG_OVF], PRF_LONG_PER_TILE);
            const u64 hit_ovf = c->h_counters[PRF_CNT_HIT_OVF];
            if (hit_ovf > c->slab_cap) { slab_cap = (u32)std::min<u64>(hit_ovf + hit_ovf / 4 + 64, 1u << 22); again = true; }
            if (!again && c->sink && nhits > c->sink_cap)
                return fail(PRF_EINVAL, "the row sink holds %llu rows, the scan found %llu", (unsigned long long)c->sink_cap,
                            (unsigned long long)nhits);
            if (!again && !c->sink && nhits > c->hit_cap) {  // the compact row array was too small: grow it and scan again
                want_hits = nhits + nhits / 8 + 1024;
                again = true;
            }
            }  // fused launch
            if (!again && !exotic.empty()) {
                // ---- tiles with symbols outside ACGTN in reach: generic kernels with the symbols' own planes (R == R matches,
                // reference utils/perfect_repeat_tracker.py:53); a row belongs to the tile of its first position there too, so
                // the two row sets are disjoint.  The rows are appended; the array is then sorted on the host.
                int rc = ensure_buffers(c, want_cand, c->hit_cap);
                if (rc) return rc;
                prf_hit_dev *rows_base = c->sink ? c->sink : c->d_hits;
                const u64 rows_cap = c->sink ? c->sink_cap : c->hit_cap;
                const u64 room = rows_cap > nhits ? rows_cap - nhits : 0;
                HIPCHK(hipMemsetAsync(c->d_counters, 0, PRF_CNT_N * sizeof(u64), c->stream));
                for (const auto &tr : exotic)
                    HIPCHK(prf_launch_scan_generic(c->stream, pl, tr.first * PRF_TILE_WORDS, tr.second * PRF_TILE_WORDS, kmin, kmax,
                                                   min_repeats, min_span, c->d_cand, c->cand_cap, c->d_counters));
                HIPCHK(prf_launch_verify(c->stream, pl, c->d_cand, c->cand_cap, min_repeats, min_span, g->d_base, (u32)g->base.size(),
                                         rows_base + nhits, room, c->d_counters));
                u64 *stage = c->h_counters + PRF_CNT_N + 2;  // spare pinned words behind the counter block
                HIPCHK(hipMemcpyAsync(stage, c->d_counters, 2 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
                const u64 ncand_g = stage[PRF_CNT_CAND], nhits_g = stage[PRF_CNT_HITS];
                launches += 1 + (u32)exotic.size();
                if (ncand_g > c->cand_cap) { want_cand = ncand_g + ncand_g / 8 + 1024; again = true; }
                if (!again && nhits_g > room) {
                    if (c->sink)
                        return fail(PRF_EINVAL, "the row sink holds %llu rows, the scan found %llu", (unsigned long long)c->sink_cap,
                                    (unsigned long long)(nhits + nhits_g));
                    want_hits = nhits + nhits_g + (nhits + nhits_g) / 8 + 1024;
                    again = true;
                }
                if (!again) {
                    ncand += ncand_g;
                    nhits += nhits_g;
                    if (nhits_g) sorted_on_device = false;
                    if (c->sink) {  // the count record behind a caller-owned row array
                        stage[0] = nhits; stage[1] = 0; stage[2] = 0;
                        HIPCHK(hipMemcpyAsync(c->sink + c->sink_cap, stage, sizeof(prf_hit_dev), hipMemcpyHostToDevice, c->stream));
                        HIPCHK(hipStreamSynchronize(c->stream));
                    }
                }
            }
        } else {
            // ---- generic path: candidates, then rows ----
            int rc = ensure_buffers(c, want_cand, want_hits);
            if (rc) return rc;
            HIPCHK(hipMemsetAsync(c->d_counters, 0, PRF_CNT_N * sizeof(u64), c->stream));
            HIPCHK(hipEventRecord(c->ev[0], c->stream));
            // a run belongs to the word that holds its first position: disjoint word ranges give disjoint row sets
            if (g->sel_on) {
                for (const auto &tr : g->sel_tiles)
                    HIPCHK(prf_launch_scan_generic(c->stream, pl, tr.first * PRF_TILE_WORDS, tr.second * PRF_TILE_WORDS, kmin, kmax,
                                                   min_repeats, min_span, c->d_cand, c->cand_cap, c->d_counters));
            } else {
                HIPCHK(prf_launch_scan_generic(c->stream, pl, 0, g->nwords - PRF_TILE_WORDS, kmin, kmax, min_repeats, min_span,
                                               c->d_cand, c->cand_cap, c->d_counters));
            }
            HIPCHK(hipEventRecord(c->ev[1], c->stream));
            HIPCHK(prf_launch_verify(c->stream, pl, c->d_cand, c->cand_cap, min_repeats, min_span, g->d_base,
                                     (u32)g->base.size(), c->d_hits, c->hit_cap, c->d_counters));
            HIPCHK(hipEventRecord(c->ev[2], c->stream));
            HIPCHK(hipMemcpyAsync(c->h_counters, c->d_counters, PRF_CNT_N * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            HIPCHK(hipEventElapsedTime(&ms01, c->ev[0], c->ev[1]));
            HIPCHK(hipEventElapsedTime(&ms12, c->ev[1], c->ev[2]));
            launches = 2;
            ncand = c->h_counters[PRF_CNT_CAND];
            nhits = c->h_counters[PRF_CNT_HITS];
            if (ncand > c->cand_cap) { want_cand = ncand + ncand / 8 + 1024; again = true; }
            if (nhits > c->hit_cap) { want_hits = nhits + nhits / 8 + 1024; again = true; }
            if (!again && c->sink) {  // the generic kernels write the internal array: hand the rows over afterwards
                if (nhits > c->sink_cap)
                    return fail(PRF_EINVAL, "the row sink holds %llu rows, the scan found %llu", (unsigned long long)c->sink_cap,
                                (unsigned long long)nhits);
                u64 *stage = c->h_counters + PRF_CNT_N + 2;
                stage[0] = nhits; stage[1] = 0; stage[2] = 0;
                if (nhits) HIPCHK(hipMemcpyAsync(c->sink, c->d_hits, nhits * sizeof(prf_hit_dev), hipMemcpyDeviceToDevice, c->stream));
                HIPCHK(hipMemcpyAsync(c->sink + c->sink_cap, stage, sizeof(prf_hit_dev), hipMemcpyHostToDevice, c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
            }
        }
        if (!again) break;
    }
    c->last_nhits = nhits;
    c->last_kmax = kmax;
    c->last_rows = c->sink ? c->sink : c->d_hits;
    c->last_sorted = sorted_on_device;
    if (stats) {
        stats->phase1_ms = ms01;
        stats->phase2_ms = ms12;
        stats->scan_ms = (double)ms01 + (double)ms12;
        stats->positions = lv.positions;
        stats->packed_bytes = (lv.positions + 3) / 4;
        stats->n_candidates = ncand;
        stats->n_hits = nhits;
        stats->n_launches = launches;
        stats->path = vs ? 1 : 0;
        stats->seq = vs && launches ? c->scan_seq : 0;
        stats->sorted_on_device = sorted_on_device ? 1u : 0u;
        stats->tiles_launched = vs ? lv.n : 0u;
    }
    if ((flags & PRF_SCAN_NO_FETCH) || !out) return PRF_OK;
    if (nhits == 0) return PRF_OK;
    prf_hit *rows = (prf_hit *)malloc(nhits * sizeof(prf_hit));
    if (!rows) return fail(PRF_ENOMEM, "prf_scan_genome: cannot allocate %llu rows", (unsigned long long)nhits);
    static_assert(sizeof(prf_hit) == sizeof(prf_hit_dev), "row layouts must agree");
    // on the library's stream: the scan returned when the counters were posted, the last workgroups may still be
    // copying their rows, and the stream is a non-blocking one (the null stream does not wait for it)
    hipError_t e = hipMemcpyAsync(rows, c->last_rows, nhits * sizeof(prf_hit), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        free(rows);
        return fail(PRF_EHIP, "row copy failed: %s", hipGetErrorString(e));
    }
    // The fused path hands the rows over sorted by (contig, start, end) (reference perfect_repeat_finder.py:81).  The
    // generic path, and a fused scan in which some tile held more rows than its LDS sort list, are sorted here.
    if (!sorted_on_device)
        std::sort(rows, rows + nhits, [](const prf_hit &a, const prf_hit &b) {
            if (a.contig != b.contig) return a.contig < b.contig;
            if (a.start != b.start) return a.start < b.start;
            if (a.end != b.end) return a.end < b.end;
            return a.k < b.k;
        });
    out->rows = rows;
    out->n = nhits;
    return PRF_OK;
}

int prf_scan_genome(prf_ctx *c, const prf_genome *g, uint32_t kmin, uint32_t kmax, uint32_t min_repeats,
                    uint32_t min_span, uint32_t flags, prf_hits *out, prf_scan_stats *stats) {
    try {
        return scan_impl(c, g, kmin, kmax, min_repeats, min_span, flags, out, stats);
    } catch (const std::bad_alloc &) {
        return fail(PRF_ENOMEM, "prf_scan_genome: out of host memory");
    } catch (...) {
        return fail(PRF_EHIP, "prf_scan_genome: unexpected exception");
    }
}

// ---- the literal lane (scan_literal.hip): regimes outside the closed form ----
// One sequence at a time, straight from its ASCII bytes (upper-cased on the device): one thread per (position, motif size)
// evaluates the reference's flush call as written; the rows are then sorted like reference :81 and reduced to the shortest
// motif per (start, end) -- what the reference's dictionary holds at the end (utils/perfect_repeat_tracker.py:93-101) -- on the
// device as well (prf_lit_sort_unique).
namespace {
struct dev_free {
    void *p = nullptr;
    ~dev_free() { if (p) (void)hipFree(p); }
};
}  // namespace

// d_seq: L bytes on the device (16-byte aligned: the 64-positions kernel reads whole 16-byte groups, 16 readable bytes behind them); upper: not upper-cased / validated yet
static int literal_device(prf_ctx *c, uint8_t *d_seq, u64 L, bool upper, u32 contig_index, u32 kmin, u32 kmax, u32 min_repeats,
                          u32 min_span, u64 stop, std::vector<prf_hit> &rows_out, float *ms, u32 *launches, u64 pos_offset = 0) {
    if (stop > L) stop = L;
    dev_free rows;
    u64 cap = L / 16 + 4096;
    u64 *h = c->h_counters;
    for (int attempt = 0;; attempt++) {
        HIPCHK(hipMalloc(&rows.p, cap * sizeof(prf_hit_dev)));
        HIPCHK(hipMemsetAsync(c->d_counters, 0, PRF_CNT_N * sizeof(u64), c->stream));
        HIPCHK(hipMemsetAsync(c->d_counters + PRF_CNT_BADPOS, 0xFF, sizeof(u64), c->stream));
        HIPCHK(hipEventRecord(c->ev[0], c->stream));
        if (attempt == 0 && upper) HIPCHK(prf_launch_lit_upper(c->stream, d_seq, L, c->d_counters + PRF_CNT_BADPOS));
        HIPCHK(prf_launch_lit_events(c->stream, d_seq, L, kmin, kmax, min_repeats, min_span, stop, contig_index,
                                     (prf_hit_dev *)rows.p, cap, c->d_counters));
        HIPCHK(hipEventRecord(c->ev[1], c->stream));
        HIPCHK(hipMemcpyAsync(h, c->d_counters, PRF_CNT_N * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        float t = 0;
        HIPCHK(hipEventElapsedTime(&t, c->ev[0], c->ev[1]));
        *ms += t;
        *launches += attempt == 0 && upper ? 2 : 1;
        if (h[PRF_CNT_BADPOS] != ~0ull)
            return fail(PRF_ESYMBOL,
                        "unsupported symbol at contig %u position %llu: only letters are accepted (A, C, G, T, N and -- as ordinary "
                        "symbols, like the reference -- any other letter, in either case); libprf refuses other bytes instead of guessing",
                        contig_index, (unsigned long long)(h[PRF_CNT_BADPOS] + pos_offset));  // (pos_offset: the N trimmed off the front)
        if (h[PRF_CNT_CAND])
            return fail(PRF_EINDEX, "string index out of range");  // the message of Python's IndexError (tracker :87)
        const u64 n = h[PRF_CNT_HITS];
        if (n <= cap) {
            // sorted by (start, end) and reduced to the shortest motif per (start, end) on the device (scan_literal.hip)
            static_assert(sizeof(prf_hit) == sizeof(prf_hit_dev), "row layouts must agree");
            if (n) {
                dev_free uniq;
                HIPCHK(hipMalloc(&uniq.p, n * sizeof(prf_hit_dev)));
                u64 *d_n = c->d_counters + PRF_CNT_HITS;  // read above; reused for the number of rows that stay
                HIPCHK(prf_lit_sort_unique(c->stream, (const prf_hit_dev *)rows.p, n, (prf_hit_dev *)uniq.p, d_n, &c->lit_scratch,
                                           &c->lit_scratch_bytes));
                HIPCHK(hipMemcpyAsync(h, d_n, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
                const u64 kept = h[0];
                if (kept > n) return fail(PRF_EHIP, "prf_scan_literal: row compaction returned %llu of %llu rows", (unsigned long long)kept,
                                          (unsigned long long)n);
                const size_t at = rows_out.size();
                rows_out.resize(at + kept);
                if (kept) {
                    HIPCHK(hipMemcpyAsync(rows_out.data() + at, uniq.p, kept * sizeof(prf_hit), hipMemcpyDeviceToHost, c->stream));
                    HIPCHK(hipStreamSynchronize(c->stream));
                }
                *launches += 5;
            }
            return PRF_OK;
        }
        if (attempt >= 2) return fail(PRF_EHIP, "prf_scan_literal: the row count changed between two runs");
        (void)hipFree(rows.p);
        rows.p = nullptr;
        cap = n;
    }
}

static int literal_one(prf_ctx *c, const prf_contig &ct, u32 contig_index, u32 kmin, u32 kmax, u32 min_repeats, u32 min_span,
                       u64 stop, std::vector<prf_hit> &rows_out, float *ms, u32 *launches, u64 pos_offset) {
    const u64 L = ct.len;
    if (L && !ct.ascii) return fail(PRF_EINVAL, "prf_scan_literal: NULL sequence");
    if (L >= (1ull << 40)) return fail(PRF_EUNSUPPORTED, "prf_scan_literal: input too large (2^40 positions)");
    dev_free seq;
    HIPCHK(hipMalloc(&seq.p, L + 16));
    if (L) HIPCHK(hipMemcpyAsync(seq.p, ct.ascii, L, hipMemcpyHostToDevice, c->stream));
    return literal_device(c, (uint8_t *)seq.p, L, true, contig_index, kmin, kmax, min_repeats, min_span, stop, rows_out, ms, launches, pos_offset);
}

static int literal_finish(prf_ctx *c, std::vector<prf_hit> &rows, float ms, u32 launches, u64 positions, prf_hits *out,
                          prf_scan_stats *stats) {
    c->last_nhits = 0;  // the rows of this lane live on the host only
    c->last_rows = nullptr;
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->scan_ms = stats->phase1_ms = ms;
        stats->positions = positions;
        stats->packed_bytes = positions;  // this lane reads the bytes themselves
        stats->n_candidates = stats->n_hits = rows.size();
        stats->n_launches = launches;
        stats->path = 2;
    }
    if (!out || rows.empty()) return PRF_OK;
    prf_hit *r = (prf_hit *)malloc(rows.size() * sizeof(prf_hit));
    if (!r) return fail(PRF_ENOMEM, "prf_scan_literal: cannot allocate %zu rows", rows.size());
    memcpy(r, rows.data(), rows.size() * sizeof(prf_hit));
    out->rows = r;
    out->n = rows.size();
    return PRF_OK;
}

// min_repeats == 1 on a RESIDENT genome: the upper-cased bytes of every contig are rebuilt on the device from the linear
// planes (H, L, X and the code planes of the symbols outside ACGTN), trimmed of the N at both ends (reference
// perfect_repeat_finder.py:40-46) and handed to the literal lane -- nothing crosses PCIe but the rows.
static int literal_genome(prf_ctx *c, const prf_genome *g, u32 kmin, u32 kmax, u32 min_repeats, u32 min_span, prf_hits *out,
                          prf_scan_stats *stats) {
    if (g->sel_on)
        return fail(PRF_EUNSUPPORTED, "min_repeats == 1 scans whole contigs (its rows depend on where a sequence begins and ends): "
                                      "clear the selection of parts (prf_genome_select with n_parts == 0)");
    if (kmax > 60000) return fail(PRF_EUNSUPPORTED, "max_motif_size %u > 60000", kmax);
    if (c->sink) return fail(PRF_EUNSUPPORTED, "min_repeats == 1: not with a row sink (the rows of the literal lane are handed over on the host)");
    if (c->slot[0].seq || c->slot[1].seq) return fail(PRF_EINVAL, "prf_scan_genome: pipelined scans are in flight on this context");
    std::vector<prf_hit> rows;
    float ms = 0;
    u32 launches = 0;
    for (size_t ci = 0; ci < g->len.size(); ci++) {
        const u64 len = g->len[ci];
        u64 lo = len, hi = len;  // nothing but N (or empty): the reference's window is seq[len:len]
        if (len) {
            u64 fl[2] = {~0ull, 0ull};
            HIPCHK(hipMemcpyAsync(c->d_counters, fl, sizeof fl, hipMemcpyHostToDevice, c->stream));
            HIPCHK(prf_launch_lit_trim(c->stream, g->X, g->E, g->base[ci] >> 6, len, c->d_counters));
            HIPCHK(hipMemcpyAsync(c->h_counters, c->d_counters, sizeof fl, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            launches++;
            if (c->h_counters[0] != ~0ull) { lo = c->h_counters[0]; hi = c->h_counters[1]; }
        }
        const u64 n = hi - lo;
        dev_free seq;
        HIPCHK(hipMalloc(&seq.p, n + 16));
        if (n) {
            HIPCHK(prf_launch_lit_unpack(c->stream, g->H, g->L, g->X, g->E, g->base[ci] + lo, n, (uint8_t *)seq.p));
            launches++;
        }
        const size_t at = rows.size();
        const int rc = literal_device(c, (uint8_t *)seq.p, n, false, (u32)ci, kmin, kmax, min_repeats, min_span, n, rows, &ms, &launches);
        if (rc) return rc;
        for (size_t r = at; r < rows.size(); r++) {
            rows[r].start += lo;
            rows[r].end += lo;
        }
    }
    return literal_finish(c, rows, ms, launches, g->positions, out, stats);
}

static int literal_impl(prf_ctx *c, const prf_contig *contigs, int n_contigs, const u64 *stops, u32 kmin, u32 kmax, u32 min_repeats,
                        u32 min_span, prf_hits *out, prf_scan_stats *stats) {
    if (!c) return fail(PRF_EINVAL, "prf_scan_literal: NULL context");
    if (out) { out->rows = nullptr; out->n = 0; }
    if (n_contigs < 0 || (n_contigs && !contigs)) return fail(PRF_EINVAL, "prf_scan_literal: bad contig array");
    // same conditions, same wording as the reference's ValueErrors (perfect_repeat_finder.py:23-30)
    if (kmin < 1) return fail(PRF_EINVAL, "min_motif_size is set to %u. It must be at least 1.", kmin);
    if (kmax < kmin) return fail(PRF_EINVAL, "max_motif_size is set to %u. It must be at least min_motif_size.", kmax);
    if (min_repeats < 1) return fail(PRF_EINVAL, "min_repeats is set to %u. It must be at least 1.", min_repeats);
    if (min_span < 1) return fail(PRF_EINVAL, "min_span is set to %u. It must be at least 1.", min_span);
    if (kmax > 60000) return fail(PRF_EUNSUPPORTED, "max_motif_size %u > 60000", kmax);
    if (min_repeats > 1000000u || min_span > (1u << 30)) return fail(PRF_EINVAL, "threshold out of range");
    if (c->slot[0].seq || c->slot[1].seq) return fail(PRF_EINVAL, "prf_scan_literal: pipelined scans are in flight on this context");
    if (c->sink) return fail(PRF_EUNSUPPORTED, "min_repeats == 1: not with a row sink (the rows of the literal lane are handed over on the host)");
    HIPCHK(hipSetDevice(c->dev));
    std::vector<prf_hit> rows;
    float ms = 0;
    u32 launches = 0;
    u64 positions = 0;
    for (int i = 0; i < n_contigs; i++) {
        prf_contig w = contigs[i];
        u64 lo = 0;
        if (!stops && w.len) {
            // prf_scan: the whole of reference detect_repeats() without an interval, whose :40-46 drop the N at both ends
            // first -- not a no-op in this regime, the rows depend on where the sequence begins and ends
            if (!w.ascii) return fail(PRF_EINVAL, "prf_scan: NULL sequence");
            u64 hi = w.len;
            while (lo < hi && (w.ascii[lo] | 0x20) == 'n') lo++;
            while (hi > lo && (w.ascii[hi - 1] | 0x20) == 'n') hi--;
            w.ascii += lo;
            w.len = hi - lo;
        }
        const size_t at = rows.size();
        const int rc = literal_one(c, w, (u32)i, kmin, kmax, min_repeats, min_span, stops ? stops[i] : w.len, rows, &ms, &launches, lo);
        if (rc) return rc;
        for (size_t r = at; r < rows.size(); r++) {
            rows[r].start += lo;
            rows[r].end += lo;
        }
        positions += contigs[i].len;
    }
    return literal_finish(c, rows, ms, launches, positions, out, stats);
}

int prf_scan_literal(prf_ctx *c, const prf_contig *contig, uint32_t kmin, uint32_t kmax, uint32_t min_repeats, uint32_t min_span,
                     uint64_t stop, prf_hits *out, prf_scan_stats *stats) {
    try {
        if (!contig) return fail(PRF_EINVAL, "prf_scan_literal: NULL contig");
        const u64 stop_ = stop;
        return literal_impl(c, contig, 1, &stop_, kmin, kmax, min_repeats, min_span, out, stats);
    } catch (const std::bad_alloc &) {
        return fail(PRF_ENOMEM, "prf_scan_literal: out of host memory");
    } catch (...) {
        return fail(PRF_EHIP, "prf_scan_literal: unexpected exception");
    }
}

int prf_scan(prf_ctx *c, const prf_contig *contigs, int n_contigs, uint32_t kmin, uint32_t kmax, uint32_t min_repeats,
             uint32_t min_span, uint32_t flags, prf_hits *out, prf_scan_stats *stats) {
    if (min_repeats == 1) {  // outside the closed form of the packed kernels: the literal lane, contig by contig
        try {
            return literal_impl(c, contigs, n_contigs, nullptr, kmin, kmax, min_repeats, min_span, out, stats);
        } catch (const std::bad_alloc &) {
            return fail(PRF_ENOMEM, "prf_scan: out of host memory");
        } catch (...) {
            return fail(PRF_EHIP, "prf_scan: unexpected exception");
        }
    }
    prf_genome *g = nullptr;
    int rc = prf_genome_load(c, contigs, n_contigs, kmax, &g);
    if (rc) return rc;
    rc = prf_scan_genome(c, g, kmin, kmax, min_repeats, min_span, flags, out, stats);
    prf_genome_free(g);
    return rc;
}

// ---- pipelined scans: enqueue now, collect later (at most two in flight) ----
// The launch latency and the host's share of a scan (~9 of ~37 us on the chr22 scan) then overlap the previous
// scan's kernel: kernels of one stream run back to back.  Fused path only, no row sink, buffers already sized by an
// ordinary scan of the same genome and parameters; anything else is refused and the caller scans synchronously.
static int scan_async_impl(prf_ctx *c, const prf_genome *g, uint32_t kmin, uint32_t kmax, uint32_t min_repeats, uint32_t min_span,
                           uint64_t *seq_out, const u64 **counters_out = nullptr, const prf_hit_dev **rows_out = nullptr) {
    if (!c || !g || !seq_out) return fail(PRF_EINVAL, "prf_scan_genome_async: bad arguments");
    if (g->ctx != c) return fail(PRF_EINVAL, "prf_scan_genome_async: genome belongs to another context");
    if (c->sink) return fail(PRF_EUNSUPPORTED, "prf_scan_genome_async: not with a row sink");
    if (kmin < 1 || kmax < kmin || min_repeats < 2 || min_span < 1 || kmax > g->kmax_hint)
        return fail(PRF_EUNSUPPORTED, "prf_scan_genome_async: parameters the synchronous call would refuse or serve otherwise");
    prf_vplan plan;
    if (!prf_vertical_plan(kmin, kmax, min_repeats, min_span, &plan))
        return fail(PRF_EUNSUPPORTED, "prf_scan_genome_async: no fused plan for these parameters");
    const launch_view lv = active_launch(g);
    if (lv.n == 0) return fail(PRF_EUNSUPPORTED, "prf_scan_genome_async: nothing to launch");
    if (!g->exotic_tiles.empty())
        return fail(PRF_EUNSUPPORTED, "prf_scan_genome_async: the genome holds symbols outside ACGTN (a second pass follows the fused scan); scan synchronously");
    if (lv.n > c->slab_slots || c->slab_cap == 0 || c->hit_cap == 0)
        return fail(PRF_EUNSUPPORTED, "prf_scan_genome_async: scan this genome synchronously once first (buffers are sized there)");
    HIPCHK(hipSetDevice(c->dev));
    prf_ctx::async_slot &sl = c->slot[c->async_n & 1u];
    if (sl.seq) return fail(PRF_EINVAL, "prf_scan_genome_async: two scans are in flight already; prf_scan_wait() first");
    if ((c->async_n & 1u) == 0) {
        sl.h = c->h_counters; sl.h_dev = c->h_counters_dev; sl.rows = c->d_hits;
    } else {
        if (!c->h_async) {
            HIPCHK(hipHostMalloc((void **)&c->h_async, (PRF_CNT_N + 8) * sizeof(u64), hipHostMallocMapped | hipHostMallocCoherent));
            memset(c->h_async, 0, (PRF_CNT_N + 8) * sizeof(u64));
        }
        if (c->hit_cap_async < c->hit_cap) {
            (void)hipFree(c->d_hits_async);
            c->d_hits_async = nullptr;
            c->hit_cap_async = 0;
            HIPCHK(hipMalloc((void **)&c->d_hits_async, c->hit_cap * sizeof(prf_hit_dev)));
            c->hit_cap_async = c->hit_cap;
        }
        sl.h = c->h_async; sl.rows = c->d_hits_async;
        if (!sl.h_dev) HIPCHK(hipHostGetDevicePointer((void **)&sl.h_dev, c->h_async, 0));
    }
    // (the slabs are shared by the two scans in flight: kernels of one stream run in order, and a scan's gather has read
    // the slabs before the next scan's tiles write them)
    u64 seq = 0;
    {
        if (counters_out) *counters_out = c->d_vcounters + (size_t)c->parity * PRF_CNT_N;  // (the block this scan counts in)
        if (rows_out) *rows_out = sl.rows;
        const int rc = launch_fused(c, g, plan, min_repeats, min_span, sl.rows, c->hit_cap, 0u, sl.h_dev, &seq);
        if (rc) return rc;
    }
    sl.seq = seq;
    sl.positions = lv.positions;
    sl.tiles = lv.n;
    sl.kmax = kmax;
    c->async_n++;
    *seq_out = seq;
    return PRF_OK;
}

int prf_scan_genome_async(prf_ctx *c, const prf_genome *g, uint32_t kmin, uint32_t kmax, uint32_t min_repeats, uint32_t min_span,
                          uint64_t *seq_out) {
    try {
        return scan_async_impl(c, g, kmin, kmax, min_repeats, min_span, seq_out);
    } catch (...) {
        return fail(PRF_EHIP, "prf_scan_genome_async: unexpected exception");
    }
}

// A pipelined scan whose rows leave as 8-byte wire words without the host in between: the pack kernels are enqueued behind the
// scan's gather on the library's stream and read the row count from the scan's counter block (which the NEXT scan's gather
// clears: stream order keeps it alive until then).
int prf_scan_genome_async_packed(prf_ctx *c, const prf_genome *g, uint32_t kmin, uint32_t kmax, uint32_t min_repeats, uint32_t min_span,
                                 void *dst_device, uint64_t capacity_rows, uint64_t side_capacity, uint64_t *seq_out) {
    try {
        if (!dst_device) return fail(PRF_EINVAL, "prf_scan_genome_async_packed: no destination");
        if (kmax > 511u) return fail(PRF_EUNSUPPORTED, "the 8-byte wire rows hold motif sizes up to 511 (max_motif_size %u)", kmax);
        const u64 *counters = nullptr;
        const prf_hit_dev *rows = nullptr;
        const int rc = scan_async_impl(c, g, kmin, kmax, min_repeats, min_span, seq_out, &counters, &rows);
        if (rc) return rc;
        HIPCHK(prf_launch_pack_rows_dev(c->stream, rows, counters + PRF_CNT_ROWS, capacity_rows, g->d_base, (u64 *)dst_device, side_capacity,
                                        c->d_side_cnt));
        return PRF_OK;
    } catch (...) {
        return fail(PRF_EHIP, "prf_scan_genome_async_packed: unexpected exception");
    }
}

// Everything enqueued on the library's stream so far happens before whatever is enqueued on `other_stream` (a hipStream_t)
// from now on: the hand-off of a send buffer to a communication stream without a host wait.
int prf_stream_wait_for(prf_ctx *c, void *other_stream) {
    if (!c) return fail(PRF_EINVAL, "prf_stream_wait_for: NULL context");
    HIPCHK(hipSetDevice(c->dev));
    HIPCHK(hipEventRecord(c->ev_handoff, c->stream));
    HIPCHK(hipStreamWaitEvent((hipStream_t)other_stream, c->ev_handoff, 0));
    return PRF_OK;
}

int prf_scan_wait(prf_ctx *c, uint64_t seq, prf_scan_stats *stats) {
    if (!c || !seq) return fail(PRF_EINVAL, "prf_scan_wait: bad arguments");
    prf_ctx::async_slot *sl = c->slot[0].seq == seq ? &c->slot[0] : (c->slot[1].seq == seq ? &c->slot[1] : nullptr);
    if (!sl) return fail(PRF_EINVAL, "prf_scan_wait: scan %llu is not in flight", (unsigned long long)seq);
    int rc = wait_for_seq(c, seq, sl->h);
    sl->seq = 0;
    if (rc) return rc;
    const u64 nhits = sl->h[PRF_CNT_ROWS];
    u64 ncand = 0;
    for (int sh = 0; sh < PRF_CNT_NSHARD; sh++) ncand += sl->h[PRF_CNT_SHARD0 + sh * PRF_CNT_SHARD_STRIDE + PRF_SH_CAND];
    if (sl->h[PRF_CNT_HIT_OVF] > c->slab_cap || nhits > c->hit_cap || sl->h[PRF_CNT_LONG_OVF])
        return fail(PRF_EUNSUPPORTED, "prf_scan_wait: the buffers sized by the last synchronous scan overflowed; scan synchronously");
    c->last_nhits = nhits;
    c->last_kmax = sl->kmax;
    c->last_rows = sl->rows;
    c->last_sorted = sl->h[PRF_CNT_UNSORTED] == 0;
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->sorted_on_device = c->last_sorted ? 1u : 0u;
        stats->tiles_launched = sl->tiles;
        stats->positions = sl->positions;
        stats->packed_bytes = (sl->positions + 3) / 4;
        stats->n_candidates = ncand;
        stats->n_hits = nhits;
        stats->n_launches = 2;
        stats->path = 1;
        stats->seq = seq;
    }
    return PRF_OK;
}

int prf_set_row_sink(prf_ctx *c, void *dst_device, uint64_t capacity_rows) {
    if (!c) return fail(PRF_EINVAL, "prf_set_row_sink: NULL context");
    c->sink = (prf_hit_dev *)dst_device;
    c->sink_cap = dst_device ? capacity_rows : 0;
    return PRF_OK;
}

int prf_last_hits_to_device(prf_ctx *c, void *dst, uint64_t capacity_rows, int count_row, uint64_t *n_rows) {
    if (!c || !n_rows || ((capacity_rows || count_row) && !dst)) return fail(PRF_EINVAL, "prf_last_hits_to_device: bad arguments");
    HIPCHK(hipSetDevice(c->dev));
    *n_rows = c->last_nhits;
    const u64 n = std::min<u64>(c->last_nhits, capacity_rows);
    if (n && dst != (void *)c->last_rows)
        HIPCHK(hipMemcpyAsync(dst, c->last_rows, n * sizeof(prf_hit_dev), hipMemcpyDeviceToDevice, c->stream));
    if (count_row) {  // staged in the spare words behind the counter block (pinned); the stream is drained below
        u64 *stage = c->h_counters + PRF_CNT_N + 2;
        stage[0] = n; stage[1] = 0; stage[2] = 0;
        HIPCHK(hipMemcpyAsync((prf_hit_dev *)dst + capacity_rows, stage, sizeof(prf_hit_dev), hipMemcpyHostToDevice, c->stream));
    }
    if (n || count_row) HIPCHK(hipStreamSynchronize(c->stream));
    return PRF_OK;
}

uint64_t prf_tile_positions(void) { return PRF_TILE; }

int prf_genome_tile_classes(const prf_genome *g, uint32_t contig, uint8_t *dst, uint64_t capacity, uint64_t *n_tiles) {
    if (!g || !n_tiles || contig >= g->base.size() || (capacity && !dst)) return fail(PRF_EINVAL, "prf_genome_tile_classes: bad arguments");
    const u64 t0 = g->base[contig] / PRF_TILE, n = (g->len[contig] + PRF_TILE - 1) / PRF_TILE;
    *n_tiles = n;
    for (u64 i = 0; i < n && i < capacity; i++) dst[i] = g->vp.h_class[t0 + i];
    return PRF_OK;
}

static int genome_select_impl(prf_genome *g, const prf_part *parts, int n_parts) {
    if (!g || n_parts < 0 || (n_parts > 0 && !parts)) return fail(PRF_EINVAL, "prf_genome_select: bad arguments");
    prf_ctx *c = g->ctx;
    HIPCHK(hipSetDevice(c->dev));
    HIPCHK(hipStreamSynchronize(c->stream));  // no scan of this genome may be reading the old selection
    if (n_parts == 0) {
        g->sel_on = false;
        return PRF_OK;
    }
    std::vector<std::pair<u64, u64>> tr;
    u64 positions = 0;
    for (int i = 0; i < n_parts; i++) {
        const prf_part &p = parts[i];
        if (p.contig >= g->base.size()) return fail(PRF_EINVAL, "prf_genome_select: part %d names contig %u of %zu", i, p.contig, g->base.size());
        const u64 len = g->len[p.contig];
        const u64 end = std::min<u64>(p.end, len);
        if (p.begin % PRF_TILE != 0 || (end % PRF_TILE != 0 && end != len) || p.begin > end)
            return fail(PRF_EINVAL, "prf_genome_select: part %d [%llu, %llu) of contig %u must begin on a multiple of %u and end on one or at "
                        "the contig's end", i, (unsigned long long)p.begin, (unsigned long long)p.end, p.contig, PRF_TILE);
        if (end == p.begin) continue;
        const u64 t0 = g->base[p.contig] / PRF_TILE;
        tr.emplace_back(t0 + p.begin / PRF_TILE, t0 + (end + PRF_TILE - 1) / PRF_TILE);
        positions += end - p.begin;
    }
    std::sort(tr.begin(), tr.end());
    std::vector<std::pair<u64, u64>> merged;
    for (const auto &r : tr) {
        if (!merged.empty() && r.first < merged.back().second)
            return fail(PRF_EINVAL, "prf_genome_select: parts overlap");
        if (!merged.empty() && r.first == merged.back().second) merged.back().second = r.second;
        else merged.push_back(r);
    }
    std::vector<u32> list;
    {
        const std::vector<u32> &all = g->vp.h_list;  // ascending by tile
        size_t i = 0;
        for (const auto &r : merged) {
            while (i < all.size() && (all[i] & ~PRF_LAUNCH_MIXED) < r.first) i++;
            while (i < all.size() && (all[i] & ~PRF_LAUNCH_MIXED) < r.second) list.push_back(all[i++]);
        }
    }
    if (!g->d_sel_list) HIPCHK(hipMalloc((void **)&g->d_sel_list, sizeof(u32) * std::max<size_t>(1, g->vp.h_list.size())));
    if (!list.empty()) HIPCHK(hipMemcpy(g->d_sel_list, list.data(), sizeof(u32) * list.size(), hipMemcpyHostToDevice));
    g->sel_n = (u32)list.size();
    g->sel_flat = prf_flat_base(list.data(), list.size());
    g->sel_positions = positions;
    g->sel_tiles = std::move(merged);
    g->sel_on = true;
    return PRF_OK;
}

int prf_genome_select(prf_genome *g, const prf_part *parts, int n_parts) {
    try {
        return genome_select_impl(g, parts, n_parts);
    } catch (const std::bad_alloc &) {
        return fail(PRF_ENOMEM, "prf_genome_select: out of host memory");
    } catch (...) {
        return fail(PRF_EHIP, "prf_genome_select: unexpected exception");
    }
}

int prf_last_hits_packed_to_device(prf_ctx *c, const prf_genome *g, void *dst, uint64_t capacity_rows, uint64_t side_capacity,
                                   uint64_t *n_rows) {
    if (!c || !g || g->ctx != c || !n_rows || !dst) return fail(PRF_EINVAL, "prf_last_hits_packed_to_device: bad arguments");
    HIPCHK(hipSetDevice(c->dev));
    *n_rows = c->last_nhits;
    if (c->last_kmax > 511u)  // (such rows used to leave with a truncated motif size)
        return fail(PRF_EUNSUPPORTED, "the 8-byte wire rows hold motif sizes up to 511; the last scan went up to %u -- hand its rows over whole "
                    "(prf_last_hits_to_device)", c->last_kmax);
    if (c->last_nhits > capacity_rows)
        return fail(PRF_EINVAL, "the packed row buffer holds %llu rows, the scan found %llu", (unsigned long long)capacity_rows,
                    (unsigned long long)c->last_nhits);
    u64 *words = (u64 *)dst;
    // (a spare device word: the generic path's counters are idle here; spare pinned words behind the counter block,
    // written by the device)
    u64 *side_cnt = c->d_counters + PRF_CNT_CAND;
    HIPCHK(hipMemsetAsync(side_cnt, 0, sizeof(u64), c->stream));
    volatile u64 *stage = c->h_counters + PRF_CNT_N + 2;
    HIPCHK(prf_launch_pack_rows(c->stream, c->last_rows, c->last_nhits, g->d_base, words, capacity_rows, side_capacity, side_cnt,
                                c->h_counters_dev + PRF_CNT_N + 2));
    HIPCHK(hipStreamSynchronize(c->stream));
    const u64 n_side = stage[0];
    if (n_side > side_capacity)
        return fail(PRF_EINVAL, "the side list holds %llu rows, %llu rows are longer than 65534", (unsigned long long)side_capacity,
                    (unsigned long long)n_side);
    return PRF_OK;
}

int prf_genome_contig_bases(const prf_genome *g, uint64_t *bases, uint64_t capacity, uint64_t *n_contigs) {
    if (!g || !n_contigs || (capacity && !bases)) return fail(PRF_EINVAL, "prf_genome_contig_bases: bad arguments");
    *n_contigs = g->base.size();
    for (size_t i = 0; i < g->base.size() && i < capacity; i++) bases[i] = g->base[i];
    return PRF_OK;
}

int prf_genome_footprint(const prf_genome *g, uint64_t *device_bytes, uint64_t *positions) {
    if (!g || !device_bytes || !positions) return fail(PRF_EINVAL, "prf_genome_footprint: bad arguments");
    const u64 tot = PRF_FRONT_PAD + g->nwords + g->padw;      // words of one linear plane (genome_load_impl)
    const u64 ntiles = g->G / PRF_TILE;
    u64 bytes = 3 * tot * 8;                                   // H, L, X
    if (g->d_E) bytes += 5 * tot * 8 + 5 * sizeof(u64 *);      // the code planes of the letters outside ACGTN
    bytes += 2 * ntiles * (PRF_TILE / 8);                      // VH, VL
    bytes += 2 * ntiles + sizeof(u32) * ntiles;                // tile classes (+ scratch), launch list
    bytes += sizeof(uint4) * ntiles + sizeof(u64) * std::max<size_t>(1, g->base.size());  // tile table, contig bases
    if (g->d_sel_list) bytes += sizeof(u32) * std::max<size_t>(1, g->vp.h_list.size());
    *device_bytes = bytes;
    *positions = g->G;
    return PRF_OK;
}

int prf_plan_describe(uint32_t kmin, uint32_t kmax, uint32_t min_repeats, uint32_t min_span, char *buf, uint64_t buf_len) {
    if (!buf || buf_len == 0) return fail(PRF_EINVAL, "prf_plan_describe: no buffer");
    const int n = prf_plan_json(kmin, kmax, min_repeats, min_span, buf, buf_len);
    if (n < 0) return fail(PRF_EINVAL, "prf_plan_describe: buffer too small");
    return n;
}

void prf_free_hits(prf_hits *h) {
    if (!h) return;
    free(h->rows);
    h->rows = nullptr;
    h->n = 0;
}

int prf_measure_hbm_read(prf_ctx *c, uint64_t bytes, int iters, double *gbps) {
    if (!c || !gbps || bytes < (1u << 20) || iters < 1) return fail(PRF_EINVAL, "prf_measure_hbm_read: bad arguments");
    HIPCHK(hipSetDevice(c->dev));
    bytes &= ~(uint64_t)15;
    void *buf = nullptr;
    u32 *sink = nullptr;
    HIPCHK(hipMalloc(&buf, bytes));
    hipError_t e = hipMalloc((void **)&sink, 64);
    if (e != hipSuccess) { (void)hipFree(buf); return fail(PRF_ENOMEM, "hipMalloc failed"); }
    (void)hipMemsetAsync(buf, 0x5A, bytes, c->stream);
    float best = 1e30f;
    for (int i = 0; i < iters + 1; i++) {  // first pass is a warm-up
        (void)hipEventRecord(c->ev[0], c->stream);
        (void)prf_launch_hbm_read(c->stream, buf, bytes, sink);
        (void)hipEventRecord(c->ev[1], c->stream);
        (void)hipStreamSynchronize(c->stream);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, c->ev[0], c->ev[1]);
        if (i > 0 && ms < best) best = ms;
    }
    (void)hipFree(buf);
    (void)hipFree(sink);
    *gbps = (double)bytes / (best * 1e-3) / 1e9;
    return PRF_OK;
}

}  // extern "C"
