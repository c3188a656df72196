#!/usr/bin/env python3
"""Drop-in for the reference's perfect_repeat_finder.py, running on an MI355X through libprf.

API kept from the reference:
  detect_repeats(input_sequence, filter_settings, verbose=False, show_progress_bar=False, debug=False)
      -> [(start_0based, end, motif), ...] sorted by (start, end)   (reference perfect_repeat_finder.py:10-81)
  find_repeats = detect_repeats   (the name BASELINE.json's north_star uses)
  main()                           same flags, defaults, stdout lines and output files (reference :83-183)

What differs underneath: the reference walks one PerfectRepeatTracker per motif size over the string,
one character per call; here the sequence is packed into bit planes in HBM and scanned by HIP
kernels.  The rows are bit-identical.  There is no CPU fallback: without libprf.so
and a gfx950 device every scan raises.
"""
import argparse
import ctypes
import os
import re
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

import prf_native  # noqa: E402


_LEADING_N = re.compile("N*")


def _check_settings(fs):
    """Same checks, same messages, same AttributeError-on-missing behaviour as reference :23-30."""
    checks = (
        ("min_motif_size", lambda v: v < 1, "It must be at least 1."),
        ("max_motif_size", lambda v: v < fs.min_motif_size, "It must be at least min_motif_size."),
        ("min_repeats", lambda v: v < 1, "It must be at least 1."),
        ("min_span", lambda v: v < 1, "It must be at least 1."),
    )
    for name, bad, tail in checks:
        value = getattr(fs, name)
        if not value or bad(value):
            raise ValueError(f"{name} is set to {value}. {tail}")


def _to_ascii(seq):
    try:
        return seq.encode("ascii")
    except UnicodeEncodeError as exc:
        raise ValueError(f"input_sequence contains a non-ASCII character at offset {exc.start}; "
                         "the packed GPU path accepts letters only (A, C, G, T, N and, as ordinary symbols, any other letter)") from None


def _gpu_rows(seq, fs, context=None, stop=None):
    """Rows of one sequence: list of (start, end, k), sorted by (start, end).  min_repeats >= 2: the packed kernels (closed
    form).  min_repeats == 1: the literal lane (csrc/scan_literal.hip), which evaluates the reference's flush call as written;
    `stop` is then the number of lock-step iterations the reference performs (default: all of them)."""
    ctx = context or prf_native.default_context()
    try:
        if fs.min_repeats < 2:
            rows, _ = ctx.scan_literal(_to_ascii(seq), fs.min_motif_size, fs.max_motif_size, fs.min_repeats, fs.min_span, stop)
        else:
            rows, _ = ctx.scan([_to_ascii(seq)], fs.min_motif_size, fs.max_motif_size, fs.min_repeats, fs.min_span)
    except prf_native.PrfError as exc:
        if exc.code in (prf_native.PRF_EINVAL, prf_native.PRF_ESYMBOL):
            raise ValueError(exc.message) from None
        if exc.code == prf_native.PRF_EINDEX:
            raise IndexError(exc.message) from None      # reference utils/perfect_repeat_tracker.py:87
        raise
    return [(int(r["start"]), int(r["end"]), int(r["k"])) for r in rows]


def _interval_cutoff(seq, fs, end_position):
    """Number of lock-step iterations the reference's position loop performs (reference :66-74) when
    it may stop early: it stops at the first position past the interval end at which no motif size
    is "in the middle of a repeat" (>= k consecutive matches so far, tracker :63-65).  Host-side
    control logic over a handful of positions; the scan itself stays on the GPU."""
    n = len(seq)
    ks = range(fs.min_motif_size, fs.max_motif_size + 1)
    end_position = max(end_position, -1)   # an interval that ends in front of its start: the loop may stop at position 0

    def matches(j, k):
        return seq[j] == seq[j + k] and seq[j] != "N"

    # consecutive matches ending at the last position each tracker has processed at time end_position
    run = {}
    for k in ks:
        last = min(end_position, n - k - 1)
        c = 0
        while last - c >= 0 and c < k and matches(last - c, k):
            c += 1
        run[k] = c
    for t in range(end_position + 1, n):
        busy = False
        for k in ks:
            if t <= n - k - 1:
                run[k] = min(run[k] + 1, k) if matches(t, k) else 0
            busy = busy or run[k] >= k
        if not busy:
            return t + 1
    return n


def detect_repeats(input_sequence, filter_settings, verbose=False, show_progress_bar=False, debug=False, context=None):
    """Detect perfect tandem repeats; see the module docstring.  `context` (a prf_native.Context) is an
    extension: by default a process-wide context on device PRF_DEVICE / LOCAL_RANK / 0 is used."""
    _check_settings(filter_settings)
    fs = filter_settings
    has_interval = hasattr(fs, "interval_start_0based") or hasattr(fs, "interval_end")
    if not has_interval and fs.min_repeats >= 2:
        # no interval: N-trimming is a no-op on the rows (N never matches), SURVEY 3.4
        rows = _gpu_rows(input_sequence, fs, context)
        return [(s, e, input_sequence[s:s + k].upper()) for s, e, k in rows]

    # interval mode, reference :35-46 and :61-81.  min_repeats == 1 comes here too: its rows depend on where the sequence
    # begins and ends (slice clamp, negative-index wrap-around), so the N-trimming of :40-46 is not a no-op there.
    seq = input_sequence.upper()
    lo = getattr(fs, "interval_start_0based", 0)
    hi = getattr(fs, "interval_end", len(seq))
    total = len(seq)
    if 0 <= lo <= hi <= total:
        # the two loops of reference :40-44 without a Python iteration per base (chr22 begins with 10.5 M of N)
        lo = _LEADING_N.match(seq, lo, hi).end()
        dropped = (hi - lo) - len(seq[lo:hi].rstrip("N"))
        hi -= dropped
        total -= dropped  # the reference shortens the whole sequence by the trimmed count (:44)
    else:                # out-of-range interval: the loops as the reference has them (negative indices wrap, IndexError)
        while lo < hi and seq[lo] == "N":
            lo += 1
        while hi > lo and seq[hi - 1] == "N":
            hi -= 1
            total -= 1
    window = seq[lo:total]
    end_position = hi - lo
    stop = _interval_cutoff(window, fs, end_position)
    if hi == len(window) and stop < len(window) - fs.min_motif_size:
        # reference :77-78: a tracker that has not reached the end of the sequence trips the assertion (before any done())
        raise AssertionError(f"{fs.min_motif_size}bp motif RepeatTracker did not reach end of the sequence")
    if fs.min_repeats < 2:
        rows = _gpu_rows(window, fs, context, stop=stop)       # the literal lane stops its trackers where the loop stopped
    else:
        # The rows the reference has written when its loop stops after `stop` iterations are the runs whose failed comparison
        # lies below `stop`; such a run ends at most max_motif_size bases later, so only that much of the sequence is scanned
        # (an interval at the front of a chromosome does not pay for the rest of it).  Runs cut off by the shortened end
        # have their failed comparison at or behind `stop` and are dropped with the others.
        scanned = window if stop >= len(window) else window[:stop + fs.max_motif_size + 1]
        rows = _gpu_rows(scanned, fs, context)
        if stop < len(window):
            rows = [(s, e, k) for s, e, k in rows if e - k <= stop - 1]
    return [(s + lo, e + lo, window[s:s + k]) for s, e, k in rows]


find_repeats = detect_repeats


# ----------------------------------------------------------------------------------------------
# command line (reference :83-183)

def _build_parser():
    p = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter,
                                description="Find perfect tandem repeats on an AMD MI355X.")
    g = p.add_argument_group("Repeat Filters")
    g.add_argument("-min", "--min-motif-size", type=int, default=1, help="Smallest motif size (bp).")
    g.add_argument("-max", "--max-motif-size", type=int, default=50, help="Largest motif size (bp).")
    g.add_argument("--min-repeats", type=int, default=3, help="Fewest copies of the motif a repeat must have.")
    g.add_argument("--min-span", type=int, default=9, help="Fewest consecutive bases a repeat must cover.")
    p.add_argument("-i", "--interval", help="Restrict the scan to chrom:start_0based-end.")
    p.add_argument("-p", "--plot", help="Accepted for compatibility; plotting is not part of this build.")
    p.add_argument("-o", "--output-prefix", help="Prefix of the output TSV (and BED, for FASTA input).")
    p.add_argument("--verbose", action="store_true", help="Print verbose output.")
    p.add_argument("--debug", action="store_true", help="Print debugging output.")
    p.add_argument("--show-progress-bar", action="store_true", help="Accepted for compatibility (scans take milliseconds).")
    p.add_argument("--stats", action="store_true",
                   help="(not in the reference) after a whole-FASTA scan print one JSON line: positions, rows, device time of the scan, "
                        "Gbp/s, algorithmic bytes (2 bits per position + 24 bytes per row) and GB/s")
    p.add_argument("input_sequence", help="A nucleotide sequence, or the path of a FASTA file")
    return p


def _scan_whole_fasta(fasta, bed_path, fs, report):
    """Every contig of the FASTA: one resident genome, one scan, BED written by libprf (reference :135-149 loops
    over the contigs one detect_repeats() call at a time).  report(entry, n_rows) is called per contig in order."""
    try:
        # min_repeats == 1: prf_scan serves it contig by contig on the literal lane (N-trimming included)
        _counts, stats = prf_native.scan_fasta_to_bed(prf_native.default_context(), fasta, bed_path, fs.min_motif_size,
                                                      fs.max_motif_size, fs.min_repeats, fs.min_span, on_contig=report)
        return stats
    except prf_native.PrfError as exc:
        if exc.code in (prf_native.PRF_EINVAL, prf_native.PRF_ESYMBOL):
            raise ValueError(exc.message) from None
        raise


def _gpu_scan_parts(entries, settings, parts):
    """This rank's share: its contigs as one resident genome, the parts it owns selected, one scan.  entries: FASTA entries
    (native memory); parts: (index into entries, begin, end).  Returns the numpy row array (contig = index into entries)."""
    ctx = prf_native.default_context()
    genome = ctx.load([(e.addr, e.length) for e in entries], settings[1])
    try:
        genome.select(parts)
        rows, _stats = genome.scan(*settings)
    finally:
        genome.free()
    return rows


def _gpu_scan_whole_contigs(entries, settings, parts):
    """The same for min_repeats == 1: this rank's contigs whole, through prf_scan (literal lane, N-trimming included)."""
    rows, _stats = prf_native.default_context().scan([(e.addr, e.length) for e in entries], *settings)
    return rows


def _tile_classes_from_the_index(path, index, whole, dist, rank, world):
    """Per-contig cost classes of the 65536-position tiles for multi_gpu.plan_parts (0 ordinary, 2 nothing but N: never
    scanned).  The shares are then balanced by what is scanned, not by length (hg38's centromeres and telomeres are tens of
    Mbp of N).  Contig c is looked at by rank c mod N only (through the index, or in the parsed file), the class arrays -- one
    byte per 65536 positions -- are exchanged with one all_gather_object: every rank plans from the same classes, and no
    rank reads the whole file for it."""
    import numpy as np
    tile = prf_native.tile_positions()
    classes = []
    for c, (name, n) in enumerate(index):
        nt = -(-n // tile)
        cls = np.zeros(nt, dtype=np.uint8)
        if nt and c % world == rank:
            if whole is not None:
                e = whole[c]
            else:
                r = prf_native.Fasta(path, only=name)
                e = r[0] if len(r) else None
            if e is not None and e.length == n:
                a = np.frombuffer((ctypes.c_char * n).from_address(e.addr), dtype=np.uint8)
                is_n = (a | 0x20) == ord("n")
                pad = np.ones(nt * tile - n, dtype=bool)
                cls[np.concatenate((is_n, pad)).reshape(nt, tile).all(axis=1)] = 2
        classes.append(cls)
    if world > 1:
        everyone = [None] * world
        dist.all_gather_object(everyone, [cls.tobytes() for cls in classes[rank::world]])
        for r in range(world):
            for i, raw in enumerate(everyone[r]):
                classes[r + i * world] = np.frombuffer(raw, dtype=np.uint8)
    return classes


def _scan_whole_fasta_sharded(path, bed_path, fs, report, scan_fn=None):
    """The same under `python -m torch.distributed.run --nproc-per-node N perfect_repeat_finder.py genome.fa`: one
    process per GPU.  The genome is cut into N shares of equal size at tile multiples (multi_gpu.plan_parts: a contig longer
    than a share is split; a row belongs to the share that holds its first position, so nothing is repaired afterwards);
    every rank reads ONLY the contigs its share touches (by seeking, if a .fai index lies next to the file), scans its share
    and writes its rows -- a contiguous piece of the final BED, because the shares are in genome order -- to a part file;
    rank 0 concatenates the parts.  Replaces the reference's interval fan-out over Hail Batch jobs and its
    `cat | sort | uniq` merge (hail_batch_pipeline/run_hail_batch_pipeline.py:76-77,115-123,151-153) on one node, with exact
    whole-contig rows.  Collectives: one all_reduce of the per-contig row counts (RCCL; PRF_DIST_BACKEND=gloo for CPU
    tensors) and an error flag; the rows themselves travel through the part files."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import multi_gpu
    backend = os.environ.get("PRF_DIST_BACKEND", "nccl")
    started_here = not dist.is_initialized()
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    if started_here:
        dist.init_process_group(backend)
    device = "cuda" if backend == "nccl" else "cpu"
    rank, world = dist.get_rank(), dist.get_world_size()
    part_path = f"{bed_path}.part{rank}"
    try:
        index, whole = prf_native.fasta_index(path)          # [(name, length)]; whole: the parsed file if there was no index
        lens = [n for _name, n in index]
        if fs.min_repeats < 2:
            # the literal lane works on whole sequences (its rows depend on where a sequence begins and ends): whole contigs
            shares = multi_gpu.plan_whole_contigs(lens, world)
        else:
            shares = multi_gpu.plan_parts(lens, world, prf_native.tile_positions(),
                                          _tile_classes_from_the_index(path, index, whole, dist, rank, world))
        mine = shares[rank]
        needed = sorted({c for c, _b, _e in mine})
        counts = np.zeros(len(index), dtype=np.int64)
        error = None
        try:
            if whole is not None:
                entries = [whole[c] for c in needed]
            else:
                readers = [prf_native.Fasta(path, only=index[c][0]) for c in needed]    # one record each, by seeking
                entries = [r[0] if len(r) else None for r in readers]
            # the shares were planned from the index: what was read must be what the index says, or the ranks would cut the
            # genome differently from what they scan (a stale .fai) -- an error on every rank, never a short BED
            for c, e in zip(needed, entries):
                if e is None or e.length != index[c][1]:
                    raise ValueError(f"{path}: contig {index[c][0]} has {'no record' if e is None else f'{e.length} bases'} in the file, "
                                     f"its index says {index[c][1]}: rebuild {path}.fai (samtools faidx) or remove it")
            local = {c: i for i, c in enumerate(needed)}
            settings = (fs.min_motif_size, fs.max_motif_size, fs.min_repeats, fs.min_span)
            scan = scan_fn or (_gpu_scan_whole_contigs if fs.min_repeats < 2 else _gpu_scan_parts)
            rows = scan(entries, settings, [(local[c], b, e) for c, b, e in mine]) if mine else []
            rows = np.asarray(rows, dtype=multi_gpu.ROW_DTYPE)
            per_local = prf_native.write_bed(part_path, entries, rows)                  # motif text from this rank's own contigs
            for c, n_rows in zip(needed, per_local):
                counts[c] = n_rows
        except Exception as exc:      # noqa: BLE001 -- whatever it is, the peers must hear about it
            error = exc
        try:
            multi_gpu.agree_or_raise(error, dist, torch, device)
        except multi_gpu.ShardError as exc:
            if exc.kind == "PrfError" and exc.code in (prf_native.PRF_EINVAL, prf_native.PRF_ESYMBOL):
                raise ValueError(exc.message) from None     # what the single-process path raises
            raise
        total = torch.from_numpy(counts).to(device)
        dist.all_reduce(total)
        dist.barrier()                                        # every part file is complete
        # (one node, one working directory: the part files are read by rank 0 where the other ranks wrote them)
        error = None
        if rank == 0:
            try:
                with open(bed_path, "wb") as out:
                    for r in range(world):
                        with open(f"{bed_path}.part{r}", "rb") as f:
                            while True:
                                chunk = f.read(1 << 24)
                                if not chunk:
                                    break
                                out.write(chunk)
            except Exception as exc:      # noqa: BLE001 -- disk full, a part file that is not there: the peers must not wait for ever
                error = exc
        multi_gpu.agree_or_raise(error, dist, torch, device)
        if rank == 0:
            class _E:                                         # what report() reads: name and length
                def __init__(self, name, n):
                    self.name, self._n = name, n
                def __len__(self):
                    return self._n
            for (name, n), c in zip(index, total.cpu().tolist()):
                report(_E(name, n), int(c))
        dist.barrier()
        return rank == 0
    finally:
        try:
            os.remove(part_path)
        except OSError:
            pass
        if started_here:
            dist.destroy_process_group()


def _scan_fasta(args, parser):
    if not args.output_prefix:
        args.output_prefix = re.sub(".fa(sta)?(.gz)?", "", args.input_sequence)   # same unanchored pattern as reference :114
    bed_path = f"{os.path.basename(args.output_prefix)}.bed"
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch  # noqa: F401  before libprf.so is loaded: PyTorch brings its own HIP runtime (load order)
    if not args.interval:
        # the reference crashes here without --interval (:139); scan every contig whole instead
        def report(entry, n_rows):
            print(f"Processing {entry.name} ({len(entry):,d} bp)")
            print(f"Found {n_rows:,d} repeats")
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:                # one process per GPU: the genome sharded over the ranks
            _check_settings(args)
            if _scan_whole_fasta_sharded(args.input_sequence, bed_path, args, report):
                print(f"Wrote results to {bed_path}")
            return
        entries = prf_native.Fasta(args.input_sequence)   # libprf's reader (plain or gzip); pyfastx in the reference
        _check_settings(args)
        stats = _scan_whole_fasta(entries, bed_path, args, report)
        print(f"Wrote results to {bed_path}")
        if getattr(args, "stats", False) and stats is not None:
            import json
            alg = int(stats.packed_bytes) + 24 * int(stats.n_hits)          # 2-bit input once for all k + the rows
            ms = float(stats.scan_ms)
            print(json.dumps({"positions": int(stats.positions), "rows": int(stats.n_hits), "scan_ms": round(ms, 4),
                              "Gbp_per_s": round(int(stats.positions) / ms / 1e6, 2) if ms else None,
                              "algorithmic_bytes": alg, "algorithmic_GB_per_s": round(alg / ms / 1e6, 2) if ms else None,
                              "kernel_path": ["generic", "fused", "literal"][int(stats.path)],
                              "rows_sorted_on_device": bool(stats.sorted_on_device), "kernel_launches": int(stats.n_launches)}))
        return
    parts = re.split("[:-]", args.interval)
    if len(parts) != 3:
        parser.error("Invalid --interval format. Must be chrom:start_0based-end")
    args.interval_chrom = parts[0]
    args.interval_start_0based = int(parts[1])
    args.interval_end = int(parts[2])
    entries = prf_native.Fasta(args.input_sequence, only=args.interval_chrom)   # one record; by seeking if there is a .fai
    if args.interval_chrom not in entries:
        parser.error(f"Chromosome {args.interval_chrom} not found in the input FASTA file")
    entry = entries[args.interval_chrom]
    seq = entry.seq
    args.interval_end = min(args.interval_end, len(seq))
    with open(bed_path, "wt") as bed:
        print(f"Processing {entry.name} ({args.interval_end - args.interval_start_0based:,d} bp)")
        rows = detect_repeats(seq, args, verbose=args.verbose, show_progress_bar=args.show_progress_bar, debug=args.debug)
        print(f"Found {len(rows):,d} repeats")
        bed.writelines(f"{entry.name}\t{s}\t{e}\t{m}\n" for s, e, m in rows)
    print(f"Wrote results to {bed_path}")


def _scan_literal(args, parser):
    if args.interval:
        parser.error("The --interval option is only supported for FASTA files.")
    if not args.output_prefix:
        args.output_prefix = "repeats"
    tsv_path = f"{args.output_prefix}.tsv"
    rows = detect_repeats(args.input_sequence, args)
    print(f"Found {len(rows):,d} repeats")
    with open(tsv_path, "wt") as tsv:
        tsv.write("start_0based\tend\tmotif\n")
        tsv.writelines(f"{s}\t{e}\t{m}\n" for s, e, m in rows)
    print(f"Wrote results to {tsv_path}")
    if args.plot:
        print("Warning: --plot is not implemented in this build. Skipping plot...")


def main(argv=None):
    parser = _build_parser()
    args = parser.parse_args(argv)
    if args.min_motif_size < 1:
        parser.error(f"--min-motif-size is set to {args.min_motif_size}. It must be at least 1.")
    if args.max_motif_size < args.min_motif_size:
        parser.error(f"--max-motif-size is set to {args.max_motif_size}. It must be at least --min-motif-size.")
    if args.min_repeats < 1:
        parser.error(f"--min-repeats is set to {args.min_repeats}. It must be at least 1.")
    if args.min_span < 1:
        parser.error(f"--min-span is set to {args.min_span}. It must be at least 1.")
    if os.path.isfile(args.input_sequence):
        _scan_fasta(args, parser)
    elif set(args.input_sequence.upper()) <= set("ACGTN"):
        _scan_literal(args, parser)
    else:
        parser.error(f"Invalid input: {args.input_sequence}. This should be a FASTA file path or a string of nucleotides.")


if __name__ == "__main__":
    main()
