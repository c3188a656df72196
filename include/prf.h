/*
 * prf.h -- C ABI of libprf, the MI355X (gfx950) perfect-tandem-repeat scanner.
 *
 * This is the drop-in boundary for the ONE hot path of
 * broadinstitute/colab-repeat-finder: "for every motif size k in [min,max]
 * compare seq[i] with seq[i+k] and report every maximal run that passes
 * min_repeats / min_span and whose motif is primitive".
 *
 * The reference has no FFI of its own (it is pure Python); the seam this
 * library sits behind is the Python call
 *
 *     detect_repeats(input_sequence, filter_settings, ...)      reference perfect_repeat_finder.py:10-81
 *
 * which drives one PerfectRepeatTracker per motif size
 * (reference utils/perfect_repeat_tracker.py:3-142).  A maintainer of the
 * reference binds these entry points with ctypes -- see INTEGRATION.md for the
 * stub -- and replaces the body of detect_repeats() with one prf_scan() call.
 *
 * Conventions: plain C types only; every function returns PRF_OK (0) or a
 * negative prf_status; nothing throws or exits across the boundary;
 * prf_last_error() gives a thread-local message for the last failure.
 * A prf_ctx is bound to one GPU and one HIP stream; calls on one ctx must be
 * serialised by the caller (the reference is single-threaded, re-entrant and
 * has no module globals -- neither has this library beyond the error string).
 * There is NO CPU fallback: without a usable gfx950 device prf_open() fails.
 */
#ifndef PRF_H
#define PRF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRF_ABI_VERSION 4

typedef enum prf_status {
    PRF_OK = 0,
    PRF_EINVAL = -1,       /* bad argument (same conditions the reference raises ValueError for, :23-30)  */
    PRF_ENODEV = -2,       /* no HIP device / wrong architecture                                          */
    PRF_EHIP = -3,         /* a HIP runtime call failed (see prf_last_error)                              */
    PRF_ENOMEM = -4,       /* host or device allocation failed                                            */
    PRF_EUNSUPPORTED = -5, /* this entry point does not serve the regime (kmax > hint; min_repeats == 1 on parts of a genome) */
    PRF_ESYMBOL = -6,      /* a byte that is not a letter: refused loudly, never guessed                  */
    PRF_EINDEX = -7        /* the reference raises IndexError on this input (utils/perfect_repeat_tracker.py:87: the
                            * extension loop's seq[i+1-period] reaches in front of the sequence; only with min_repeats == 1
                            * and a motif size larger than the sequence + 1)                               */
} prf_status;

typedef struct prf_ctx prf_ctx;       /* device, stream, scratch                           */
typedef struct prf_genome prf_genome; /* contigs packed and resident in HBM                */

/* One input sequence; caller-owned, read-only for the duration of the call.
 * Replaces the Python str handed to detect_repeats() (reference :10, :33). */
typedef struct prf_contig {
    const uint8_t *ascii; /* raw bytes, any case; upper-casing (reference :33) is folded into the packer */
    uint64_t len;
} prf_contig;

/* One output row = one (start_0based, end, motif) tuple of the reference (:81, :101);
 * the motif text is seq.upper()[start:start+k], materialised by the caller. */
typedef struct prf_hit {
    uint64_t start;  /* 0-based, contig-local                                  */
    uint64_t end;    /* exclusive                                              */
    uint32_t k;      /* motif size                                             */
    uint32_t contig; /* index into the contigs array                           */
} prf_hit;

typedef struct prf_hits {
    prf_hit *rows; /* library-owned; sorted by (contig, start, end) like reference :81 (on the device, fused path) */
    uint64_t n;
} prf_hits;

/* Per-call counters for bench.py / --stats (none of this exists in the reference). */
typedef struct prf_scan_stats {
    double scan_ms;         /* HIP-event time on the ctx stream: first scan kernel start -> hit count ready */
    double phase1_ms;       /* candidate-finding kernel(s)                                               */
    double phase2_ms;       /* candidate verification / extension / filters / compaction                 */
    uint64_t positions;     /* sum of contig lengths scanned                                             */
    uint64_t packed_bytes;  /* ceil(positions/4): the algorithmic input bytes                            */
    uint64_t n_candidates;  /* phase-1 candidates                                                        */
    uint64_t n_hits;        /* rows                                                                      */
    uint32_t n_launches;    /* kernel launches in the timed region                                       */
    uint32_t path;          /* 0 = generic kernel, 1 = vertical bit-sliced kernel, 2 = literal lane      */
    uint64_t seq;           /* fused path: serial number of this scan on its context (prf_scan_timings)  */
    uint32_t sorted_on_device; /* 1: the rows left the device sorted by (contig, start, end), no host sort   */
    uint32_t tiles_launched;   /* fused path: 65536-position tiles scanned (tiles of nothing but N are skipped)     */
} prf_scan_stats;

/* prf_scan flags */
#define PRF_SCAN_DEFAULT 0u
#define PRF_SCAN_FORCE_GENERIC 1u /* use the generic (any k, any thresholds) kernel even if a tuned one exists */
#define PRF_SCAN_NO_FETCH 2u      /* leave rows on the device (out may be NULL); for timing loops               */
#define PRF_SCAN_DEFER_TIMING 4u  /* fused path: do not wait for this scan's HIP events; scan_ms / phase*_ms are
                                   * reported as 0 and read later, for up to the last PRF_TIMING_RING scans, with
                                   * prf_scan_timings()                                                         */
#define PRF_TIMING_RING 128

int prf_abi_version(void);
int prf_device_count(void);
const char *prf_last_error(void);

/* Bind a context to HIP device `device_id` (must be gfx950). */
int prf_open(int device_id, prf_ctx **out);
void prf_close(prf_ctx *ctx);

/* Upload + pack contigs: ASCII -> 2-bit planes (+ a not-ACGT plane) resident in HBM.
 * Replaces input_sequence.upper() and the three transient copies of reference :33-46.
 * kmax_hint: largest motif size later scans will use (sizes the inter-contig guard gap). */
int prf_genome_load(prf_ctx *ctx, const prf_contig *contigs, int n_contigs, uint32_t kmax_hint, prf_genome **out);
/* The same, with the contigs generated ON the device (BASELINE config C5: 10^10 bp never cross PCIe):
 * contig i is lens[i] bases of the counter-based generator of SURVEY 8(d) with seed seeds[i],
 * base j = "ACGT"[splitmix64(seed + (j+1)*0x9E3779B97F4A7C15) >> 62]. */
int prf_genome_synth(prf_ctx *ctx, const uint64_t *lens, const uint64_t *seeds, int n_contigs, uint32_t kmax_hint,
                     prf_genome **out);
/* The same with the stand-in recipe for a chromosome (synth.py::standin2; no genome FASTA exists offline): uniform
 * background, N blocks at both ends and a centromere-like gap, one planted perfect tandem repeat per 588 positions. */
int prf_genome_standin(prf_ctx *ctx, const uint64_t *lens, const uint64_t *seeds, int n_contigs, uint32_t kmax_hint,
                       prf_genome **out);
void prf_genome_free(prf_genome *g);
uint64_t prf_genome_positions(const prf_genome *g);

/* Sharding ONE resident genome over several GPUs (north_star: "contigs x motif-sizes shard embarrassingly"; the
 * reference's analogue is the 500 kb interval fan-out of hail_batch_pipeline/run_hail_batch_pipeline.py:76-77).
 * Every rank holds the genome and selects the parts it scans: position ranges of contigs, cut at multiples of
 * prf_tile_positions().  A row belongs to the part that holds its first position, so the row sets of disjoint parts
 * are disjoint and their union over a cover of the genome is exactly the whole scan -- no halo, no exchange, no
 * repair step.  The selection stays in force for every following scan of the genome (synchronous or pipelined);
 * n_parts == 0 selects the whole genome again.  prf_scan_stats.positions then counts the selected positions. */
typedef struct prf_part {
    uint32_t contig;
    uint64_t begin; /* contig-local, a multiple of prf_tile_positions()                         */
    uint64_t end;   /* exclusive; a multiple of prf_tile_positions(), or >= the contig's length */
} prf_part;
uint64_t prf_tile_positions(void);
int prf_genome_select(prf_genome *g, const prf_part *parts, int n_parts);
/* Cost classes of a contig's tiles, for planners: 0 = ordinary, 1 = not-ACGT symbols in reach (slower variant),
 * 2 = nothing but not-ACGT (never scanned).  *n_tiles = ceil(len / prf_tile_positions()); at most `capacity` bytes
 * are written. */
int prf_genome_tile_classes(const prf_genome *g, uint32_t contig, uint8_t *dst, uint64_t capacity, uint64_t *n_tiles);

/* The hot path on a resident genome: every k in [kmin,kmax], rows as reference :81.
 * min_repeats == 1 (outside the closed form of the packed kernels: the reference's rows then depend on the text in front of
 * a run, on where the sequence begins and ends and on Python's negative-index wrap-around) is served by the literal lane:
 * the upper-cased bytes of every contig are rebuilt on the device from the planes, trimmed of the N at both ends (reference
 * :40-46) and scanned as prf_scan_literal() does; whole contigs only (PRF_EUNSUPPORTED while a selection of parts or a row
 * sink is set), rows on the host only (prf_last_hits_* then have nothing to hand over), PRF_EINDEX as there. */
int prf_scan_genome(prf_ctx *ctx, const prf_genome *g, uint32_t kmin, uint32_t kmax, uint32_t min_repeats,
                    uint32_t min_span, uint32_t flags, prf_hits *out, prf_scan_stats *stats);

/* One-shot convenience = prf_genome_load + prf_scan_genome + prf_genome_free.
 * This is the call that replaces the body of reference detect_repeats() (:33-81).
 * min_repeats == 1 is served by the literal lane (prf_scan_literal, contig by contig, flags ignored). */
int prf_scan(prf_ctx *ctx, const prf_contig *contigs, int n_contigs, uint32_t kmin, uint32_t kmax,
             uint32_t min_repeats, uint32_t min_span, uint32_t flags, prf_hits *out, prf_scan_stats *stats);

/* The literal lane: the reference's per-tracker flush call (utils/perfect_repeat_tracker.py:71-101) evaluated as written,
 * per (position, motif size) (round 3: a thread owns 64 positions and decides from bit masks which events can pass the
 * filters at all; motif sizes above 63 one thread per four positions), on the upper-cased bytes of ONE sequence -- for the regimes outside the
 * closed form of the packed kernels: min_repeats == 1 (:86-91 then depend on the text in front of the run, on the slice
 * clamp at the end of the sequence and on Python's negative-index wrap-around), and a lock-step loop that stops early
 * (interval mode, reference perfect_repeat_finder.py:66-74).  `stop` = the number of iterations that loop performs
 * (>= contig->len: it runs to the end): every tracker stands at min(stop, len - k) when done() is called (:79).  Any
 * min_repeats >= 1 is accepted (the rows equal prf_scan's for min_repeats >= 2 and stop >= len); work is O(len x motif
 * sizes) byte compares, so this is the slow lane.  Rows sorted by (start, end), one row per (start, end): the shortest motif,
 * as the reference's dictionary keeps it (:93-101); prf_hit.k = len(motif), which is smaller than the motif size only where
 * the motif slice was clamped at the end of the sequence (:82).  PRF_EINDEX where the reference raises IndexError.  No
 * N-trimming here (reference perfect_repeat_finder.py:40-46 is the caller's; prf_scan does it for min_repeats == 1). */
int prf_scan_literal(prf_ctx *ctx, const prf_contig *contig, uint32_t kmin, uint32_t kmax, uint32_t min_repeats,
                     uint32_t min_span, uint64_t stop, prf_hits *out, prf_scan_stats *stats);

/* Pipelined scans.  prf_scan_genome_async() enqueues a scan and returns its serial number at once;
 * prf_scan_wait() collects it (row count and candidate count in *stats; kernel time through prf_scan_timings;
 * rows through prf_last_hits_to_device, valid until the next-but-one scan is enqueued).  At most two scans are in
 * flight, so the launch and the host's share of scan i+1 overlap the kernel of scan i.  Served only where it needs no
 * decisions on the way: fused path, no row sink, buffers sized by an earlier prf_scan_genome() of the same genome and
 * parameters; PRF_EUNSUPPORTED otherwise, and from prf_scan_wait() if those buffers overflowed -- scan
 * synchronously then.  No other call on the context while scans are in flight. */
int prf_scan_genome_async(prf_ctx *ctx, const prf_genome *g, uint32_t kmin, uint32_t kmax, uint32_t min_repeats,
                          uint32_t min_span, uint64_t *seq_out);
int prf_scan_wait(prf_ctx *ctx, uint64_t seq, prf_scan_stats *stats);

/* Row sink: the rows of the following scans on this context are compacted straight into caller-owned device
 * memory (e.g. the send buffer of an RCCL gather) instead of the library's own array.  dst_device must hold
 * capacity_rows + 1 prf_hit records; record number capacity_rows receives {start = number of rows, 0, 0, 0}.
 * With a sink set, prf_scan_genome returns only when all rows are in place (it drains its stream), and fails with
 * PRF_EINVAL if the scan finds more than capacity_rows rows.  dst_device == NULL: back to the internal array. */
int prf_set_row_sink(prf_ctx *ctx, void *dst_device, uint64_t capacity_rows);

/* HIP-event times (ms) of the two kernels of the fused scans first_seq .. first_seq+n-1 of this context (prf_scan_stats.seq);
 * they must be among its last PRF_TIMING_RING fused scans.  Waits for those scans' events. */
int prf_scan_timings(prf_ctx *ctx, uint64_t first_seq, uint32_t n, float *kernel_ms);
/* The same, per kernel: the fused scan kernel and the row gather that follows it (a third event lies between them). */
int prf_scan_timings_split(prf_ctx *ctx, uint64_t first_seq, uint32_t n, float *scan_ms, float *gather_ms);

void prf_free_hits(prf_hits *hits);

/* Device-side hand-off of the rows of the LAST prf_scan_genome() on this context (unsorted, 24-byte
 * prf_hit records): copied device-to-device into caller-owned device memory (e.g. a torch tensor that
 * an RCCL gather then ships to rank 0).  *n_rows receives the row count; at most capacity_rows are copied.
 * count_row != 0: the record at index capacity_rows (the buffer must hold capacity_rows+1 records) receives
 * {start = number of rows copied, end = 0, k = 0, contig = 0}, so that a padded gather carries its own length.
 * The copy is complete when the call returns. */
int prf_last_hits_to_device(prf_ctx *ctx, void *dst_device, uint64_t capacity_rows, int count_row, uint64_t *n_rows);

/* The same hand-off in the 8-byte wire format of the multi-GPU gather (a third of the bytes on the xGMI links): word i =
 * tile << 41 | start in the tile << 25 | min(end - start, 65535) << 9 | k, where tile * prf_tile_positions() + start in the
 * tile is the row's first position in the genome's coordinate space (contig c begins at prf_genome_contig_bases()[c]).
 * dst_device holds capacity_rows words, then ONE count word (rows | long rows << 40), then side_capacity full rows of three
 * words (start, end, k | contig << 32) for the rows whose span does not fit 16 bits.  multi_gpu.unpack_rows() decodes.
 * Fails with PRF_EINVAL if the scan found more rows than capacity_rows (or more long rows than side_capacity), and with
 * PRF_EUNSUPPORTED if the last scan's max motif size exceeds 511 (k has 9 bits on the wire: hand such rows over whole). */
int prf_last_hits_packed_to_device(prf_ctx *ctx, const prf_genome *g, void *dst_device, uint64_t capacity_rows,
                                   uint64_t side_capacity, uint64_t *n_rows);
/* A pipelined scan (prf_scan_genome_async) whose rows leave in that wire format with no host step in between: the pack
 * kernels run behind the scan on the library's stream and take the row count from the device.  dst_device as above; a scan
 * with more rows than capacity_rows (or more long rows than side_capacity) leaves an all-ones count word, which
 * multi_gpu.unpack_rows() refuses.  Collect the scan with prf_scan_wait() as usual. */
int prf_scan_genome_async_packed(prf_ctx *ctx, const prf_genome *g, uint32_t kmin, uint32_t kmax, uint32_t min_repeats,
                                 uint32_t min_span, void *dst_device, uint64_t capacity_rows, uint64_t side_capacity,
                                 uint64_t *seq_out);
/* Order: everything enqueued on the context's stream so far happens before what is enqueued on other_stream (a hipStream_t,
 * e.g. torch.cuda.Stream().cuda_stream) from now on.  The hand-off of a send buffer to a communication stream. */
int prf_stream_wait_for(prf_ctx *ctx, void *other_stream);

/* First position of every contig in the genome's coordinate space (multiples of prf_tile_positions()). */
int prf_genome_contig_bases(const prf_genome *g, uint64_t *bases, uint64_t capacity, uint64_t *n_contigs);

/* Device memory a resident genome holds (bytes: the packed planes in both layouts, tile tables, launch lists) and the positions
 * of its coordinate space (contigs + guard gaps, what the planes cover).  Round 3: 0.625 bytes per position -- three linear
 * planes (H, L, not-ACGT) and two bit-sliced ones; genomes with letters outside ACGTN add five linear planes. */
int prf_genome_footprint(const prf_genome *g, uint64_t *device_bytes, uint64_t *positions);

/* ---- the data formats either side of the path (host code, no GPU) ------------------------------------------
 * FASTA reader: what the reference takes from pyfastx.Fasta (perfect_repeat_finder.py:117,130,136-143): entries in
 * file order, name = header up to the first white space, sequence = the record's lines joined (case kept).
 * Plain or gzip-compressed files. */
typedef struct prf_fasta prf_fasta;
int prf_fasta_open(const char *path, prf_fasta **out);
/* One record only (reference fasta[chrom].seq, :130): read by seeking if an uncompressed file has a samtools-style
 * index path + ".fai" next to it, else parsed and filtered.  *out holds one entry, or none if the name is absent. */
int prf_fasta_open_contig(const char *path, const char *name, prf_fasta **out);
int prf_fasta_count(const prf_fasta *f);
int prf_fasta_entry(const prf_fasta *f, int i, const char **name, const uint8_t **seq, uint64_t *len);
void prf_fasta_close(prf_fasta *f);

/* Row writers: the reference's BED lines "chrom\tstart\tend\tmotif\n" (:148-149; names[i] / contigs[i] belong to
 * hit.contig == i) and its TSV file with the header "start_0based\tend\tmotif" (:166-170).  The motif is
 * seq.upper()[start:start+k], taken from the contig bytes. */
int prf_write_bed(const char *path, int append, const char *const *names, const prf_contig *contigs, int n_contigs,
                  const prf_hits *hits, uint64_t *n_written);
int prf_write_tsv(const char *path, const prf_contig *contig, const prf_hits *hits, uint64_t *n_written);

/* Host-only: describe, as one line of JSON, how a scan with these parameters is dealt to the waves of the fused
 * kernel (tasks, motif sizes, examined-group strides) or that the generic kernel is used.  Needs no GPU; used by
 * the CPU tests to check that every motif size is covered exactly once and that the sampling strides are legal.
 * Returns the length written (excluding the NUL), or PRF_EINVAL if buf is too small. */
int prf_plan_describe(uint32_t kmin, uint32_t kmax, uint32_t min_repeats, uint32_t min_span, char *buf, uint64_t buf_len);

/* Roofline probe: streaming 16-byte-per-lane read of `bytes` bytes, best of `iters`; GB/s (1e9). */
int prf_measure_hbm_read(prf_ctx *ctx, uint64_t bytes, int iters, double *gbps);

#ifdef __cplusplus
}
#endif
#endif /* PRF_H */
