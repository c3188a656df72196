"""Synthetic inputs for benchmarks and full-size parity tests (no genome FASTA exists offline).

synth_bases()      SURVEY 8(d) counter-based generator: base i = "ACGT"[splitmix64(seed + (i+1)*phi) >> 62];
                   reproducible in Python, C (oracle/prf_oracle.c: prf_oracle_synth) and on any device.
chr_standin()      stand-in for a human chromosome: the same uniform background, N blocks where hg38 has
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
