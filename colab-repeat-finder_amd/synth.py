"""Synthetic inputs for benchmarks and full-size parity tests (no genome FASTA exists offline).

synth_bases()      SURVEY 8(d) counter-based generator: base i = "ACGT"[splitmix64(seed + (i+1)*phi) >> 62];
                   reproducible in Python, C (oracle/prf_oracle.c: prf_oracle_synth) and on any device.
standin2()         the stand-in recipe in a form that every position can compute for itself (integer arithmetic only), so
                   that a whole hg38-shaped genome is generated ON the device (libprf: prf_genome_standin) and any
                   window of it on the host: uniform background, N blocks at both ends and a centromere-like gap,
                   one planted perfect tandem repeat per 588-position slot (1 700 / Mbp), motif sizes after the
                   reference's golden chr22 BED, motif = the background at the repeat's own position.
chr22_real()       the chr22 stand-in background with EVERY cluster of the reference's golden chr22 BED planted at its real
                   coordinate (tests/golden/chr22_clusters_all.tsv.gz, reconstructed from the BED and re-run through the
                   reference by oracle/gen_golden.py): a workload with the real genome's row clustering -- 582 occupied
                   65536-position tiles, up to 748 rows in one -- and 63 651 known-answer windows inside it.
chr_standin()      (older recipe, kept for the pinned chr22 inputs) stand-in for a human chromosome: the same uniform background, N blocks where hg38 has
                   them (chr22: the first 10.51 Mb and the last 10 kb), and planted perfect tandem repeats at
                   ~1.7 k/Mbp whose motif sizes follow the reference's golden chr22 BED
                   (benchmark/repeat_finder/chr22_repeats.bed: 1:20000 2:12110 3:21917 4:9475 5:3207 6:929)
                   plus a thin tail of longer motifs up to 50 bp.
"""
import numpy as np

_PHI = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)

CHR22_LEN = 50_818_468
CHR1_LEN = 248_956_422


def _splitmix(z):
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def synth_codes(n, seed, start=0, chunk=1 << 22):
    """uint8 array of 2-bit draws (0..3)."""
    out = np.empty(n, dtype=np.uint8)
    seed = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        idx = np.arange(start + lo + 1, start + hi + 1, dtype=np.uint64)
        with np.errstate(over="ignore"):
            z = _splitmix(seed + idx * _PHI)
        out[lo:hi] = (z >> np.uint64(62)).astype(np.uint8)
    return out


def synth_bases(n, seed, start=0):
    """ASCII uint8 array of length n."""
    return _ACGT[synth_codes(n, seed, start)]


def _draws(count, seed, stream):
    """count uniform uint64 draws, independent per `stream`."""
    idx = np.arange(1, count + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return _splitmix(np.uint64(seed) + np.uint64(stream) * np.uint64(0xD1B54A32D192ED03) + idx * _PHI)


def chr_standin(length=CHR22_LEN, seed=22, n_head=10_510_000, n_tail=10_000, repeats_per_mbp=1700.0):
    """ASCII uint8 array; deterministic in (length, seed, n_head, n_tail, repeats_per_mbp)."""
    seq = synth_bases(length, seed)
    n_head = min(n_head, length)
    n_tail = min(n_tail, length - n_head)
    body_lo, body_hi = n_head, length - n_tail
    body = body_hi - body_lo
    n_rep = int(body / 1e6 * repeats_per_mbp)
    if n_rep > 0 and body > 1000:
        pos = body_lo + (_draws(n_rep, seed, 1) % np.uint64(body - 700)).astype(np.int64)
        u = (_draws(n_rep, seed, 2) >> np.uint64(11)).astype(np.float64) / float(1 << 53)
        # motif size: golden-BED histogram for 1..6 (95 %), geometric tail 7..50 (5 %)
        hist = np.array([20000, 12110, 21917, 9475, 3207, 929], dtype=np.float64)
        cdf = np.cumsum(hist / hist.sum()) * 0.95
        k = np.searchsorted(cdf, u, side="right") + 1
        tail = u >= 0.95
        k[tail] = 7 + np.minimum(43, np.floor(-np.log1p(-(u[tail] - 0.95) / 0.05 * 0.999) * 9.0)).astype(np.int64)
        v = (_draws(n_rep, seed, 3) >> np.uint64(11)).astype(np.float64) / float(1 << 53)
        copies = 3 + np.floor(-np.log1p(-v * 0.999) * 2.5).astype(np.int64)
        span = np.maximum(k * copies, 9 + (k > 1) * 0)
        span = np.minimum(span, 600)
        for p, kk, sp in zip(pos.tolist(), k.tolist(), span.tolist()):
            motif = seq[p:p + kk].copy()
            reps = -(-sp // kk)
            seq[p:p + sp] = np.tile(motif, reps)[:sp]
    if n_head:
        seq[:n_head] = ord("N")
    if n_tail:
        seq[length - n_tail:] = ord("N")
    return seq


# ---- stand-in recipe 2: position-computable (numpy here, HIP in csrc/pack.hip::prf_standin2_kernel) ----------------
SLOT2 = 588                      # positions per planted repeat: 1e6 / 588 = 1 700.7 repeats per Mbp
_SLOT_SALT = np.uint64(0xD1B54A32D192ED03)
# motif size 1..6: cumulative golden-BED histogram (20000, 12110, 21917, 9475, 3207, 929) scaled to 95 % of 2^16
_K_CUM = np.array([18412, 29561, 49738, 58461, 61413, 62268], dtype=np.int64)
# copies = 3 + geometric(ratio 0.67): v (24 bits) below floor(2^24 * 0.67^(i+1)) adds one copy, i = 0..15
_COPY_TH = np.array([11240734, 7531292, 5045965, 3380797, 2265134, 1517639, 1016818, 681268, 456449, 305821, 204900, 137283,
                     91979, 61626, 41289, 27664], dtype=np.int64)   # the same table in csrc/pack.hip


def standin2_layout(n):
    """(n_head, n_tail, gap_lo, gap_hi): N blocks of a contig of length n (gap_lo == gap_hi: no inner gap)."""
    n_head = n_tail = min(10_000, n // 100)
    if n >= 1_000_000:
        gap_lo = n // 8 * 3 + 12_345
        return n_head, n_tail, gap_lo, gap_lo + n // 25
    return n_head, n_tail, 0, 0


def _slot_params(seed, slots):
    """k, span, offset-in-slot of the planted repeat of every slot index in `slots` (int64 arrays)."""
    with np.errstate(over="ignore"):
        d = _splitmix((np.uint64(seed & 0xFFFFFFFFFFFFFFFF) ^ _SLOT_SALT) + (slots.astype(np.uint64) + np.uint64(1)) * _PHI)
        d2 = _splitmix(d + _PHI)
    u = (d >> np.uint64(48)).astype(np.int64)
    v = ((d >> np.uint64(24)) & np.uint64(0xFFFFFF)).astype(np.int64)
    w = (d & np.uint64(0xFFFFFF)).astype(np.int64)
    x = (d2 >> np.uint64(32)).astype(np.int64)
    k = np.searchsorted(_K_CUM, u, side="right") + 1
    t = u - int(_K_CUM[-1])
    k = np.where(u >= _K_CUM[-1], 7 + (t * t * 44) // (3268 * 3268), k)
    copies = 3 + (v[:, None] < _COPY_TH[None, :]).sum(axis=1)
    span = np.maximum(np.minimum(k * copies + w % k, 500), 10)
    off = x % (SLOT2 - span)
    return k, span, off


def standin2(n, seed, start=0, count=None):
    """ASCII uint8 array: positions [start, start+count) of the recipe-2 stand-in contig (n, seed)."""
    count = n - start if count is None else count
    assert 0 <= start and start + count <= n
    seq = synth_bases(count, seed, start)
    n_head, n_tail, gap_lo, gap_hi = standin2_layout(n)
    body_lo, body_hi = n_head, n - n_tail
    n_slots = (body_hi - body_lo) // SLOT2
    if n_slots > 0 and count > 0:
        s0 = max(0, (start - body_lo) // SLOT2)
        s1 = min(n_slots, (start + count - 1 - body_lo) // SLOT2 + 1)
        if s1 > s0:
            slots = np.arange(s0, s1, dtype=np.int64)
            k, span, off = _slot_params(seed, slots)
            p = body_lo + slots * SLOT2 + off
            tot = int(span.sum())
            rep = np.repeat(np.arange(len(slots)), span)
            j = np.arange(tot, dtype=np.int64) - np.repeat(np.cumsum(span) - span, span)     # 0 .. span-1 per repeat
            dst = p[rep] + j
            src = p[rep] + j % k[rep]
            keep = (dst >= start) & (dst < start + count)
            dst, src = dst[keep], src[keep]
            seq[dst - start] = _ACGT[synth_codes_at(src, seed)]
    lo, hi = start, start + count
    for a, b in ((0, n_head), (n - n_tail, n), (gap_lo, gap_hi)):
        a, b = max(a, lo), min(b, hi)
        if b > a:
            seq[a - lo:b - lo] = ord("N")
    return seq


def synth_codes_at(idx, seed):
    """2-bit draws of the background generator at arbitrary positions (int64 array)."""
    with np.errstate(over="ignore"):
        z = _splitmix(np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + (idx.astype(np.uint64) + np.uint64(1)) * _PHI)
    return (z >> np.uint64(62)).astype(np.uint8)


def chr22_real(clusters_path=None, seed=22, n_head=10_510_000, n_tail=10_000, length=CHR22_LEN):
    """-> (ASCII uint8 array of `length` bases, planted rows).  N blocks where hg38 chr22 has them, uniform background, and
    every cluster of the golden-BED fixture written over it at its real coordinate, one flank base either side (the flanks
    are what makes the reference reproduce exactly the cluster's rows).  planted rows: structured array (start, end, k) in
    contig coordinates, sorted -- every one of them must be a row of the scan (the background may add rows of its own).
    A cluster whose flank collides with its predecessor's (different base at the same position) is left out."""
    import gzip
    import os
    if clusters_path is None:
        clusters_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                     "chr22_clusters_all.tsv.gz")
    seq = synth_bases(length, seed)
    seq[:n_head] = ord("N")
    seq[length - n_tail:] = ord("N")
    rows = []
    prev_end = -1
    with gzip.open(clusters_path, "rt") as f:
        for line in f:
            if line.startswith("#"):
                continue
            s0, body, spec = line.rstrip("\n").split("\t")
            at = int(s0) - 1                                   # the fixture's sequence begins one flank base before the cluster
            b = np.frombuffer(body.encode(), dtype=np.uint8)
            if at < prev_end:                                  # shares its first base(s) with the previous cluster's flank
                ov = prev_end - at
                if not np.array_equal(seq[at:prev_end], b[:ov]):
                    continue
            if at < n_head or at + len(b) > length - n_tail:
                continue
            seq[at:at + len(b)] = b
            prev_end = at + len(b)
            for r in spec.split(","):
                a, e, m = r.split(":")
                rows.append((at + int(a), at + int(e), len(m)))
    rows = np.array(sorted(rows), dtype=[("start", "<u8"), ("end", "<u8"), ("k", "<u4")])
    return seq, rows
