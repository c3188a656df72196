"""GPU (-m gpu): the HIP path, called through the C ABI, against the golden fixtures and the oracle.
Integer / index work: the bar is bit-exact rows."""
import argparse
import glob
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from helpers import expected, outcome

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    # torch first: one test hands rows to a torch tensor, and torch's bundled HIP runtime must be the one the
    # process loads (bench.py imports in the same order); libprf alone runs on the system runtime.
    import torch
    assert torch.cuda.is_available()
    import prf_native
    c = prf_native.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def detect(ctx):
    import perfect_repeat_finder as prf

    def run(seq, fs):
        return prf.detect_repeats(seq, fs, context=ctx)
    return run


def prf_native_flags():
    import prf_native
    return prf_native


def rows_as_tuples(rows):
    return [(int(r["contig"]), int(r["start"]), int(r["end"]), int(r["k"])) for r in rows]


def oracle_rows(seq_bytes, kmin, kmax, r, span):
    from oracle import prf_oracle
    return [(s, e, k) for s, e, _ml, k in prf_oracle.detect_rows(seq_bytes, kmin, kmax, r, span)]


def test_reference_unit_vectors(detect, golden_unit):
    for case in golden_unit:
        assert outcome(detect, case["seq"], case["settings"]) == ("ok", case["rows"]), case["tag"]


def test_fuzz_small(detect, golden_fuzz):
    n = n_one = 0
    for case in golden_fuzz:
        n += 1
        n_one += case["settings"]["min_repeats"] < 2
        assert outcome(detect, case["seq"], case["settings"]) == expected(case), case
    assert n >= 4000 and n_one > 600


def test_min_repeats_one(detect, golden_min_repeats_one):
    """The regime outside the closed form, served by the literal lane (csrc/scan_literal.hip): 3000 reference-generated
    cases with interval mode, N at both ends, lower case, IUPAC letters, motif sizes beyond the sequence (IndexError)."""
    statuses = set()
    for case in golden_min_repeats_one:
        statuses.add(case["status"])
        assert outcome(detect, case["seq"], case["settings"]) == expected(case), case
    assert {"ok", "IndexError", "AssertionError"} <= statuses


def test_odd_intervals(detect, golden_odd_intervals):
    """Interval bounds reversed, outside the sequence or on N, min_repeats 1-3: 1500 reference-generated cases."""
    for case in golden_odd_intervals:
        assert outcome(detect, case["seq"], case["settings"]) == expected(case), case


def test_literal_lane_vs_oracle_and_vs_packed_kernels(ctx):
    """Seeded inputs at sizes the oracle finishes in seconds: (a) min_repeats == 1 against the oracle, incl. a long
    homopolymer and N blocks; (b) for min_repeats >= 2 the literal lane, the fused kernel and the oracle agree row for row;
    (c) prf_scan on several contigs with min_repeats == 1 = the oracle per contig (N-trimming of reference :40-46)."""
    import synth
    from oracle import prf_oracle
    seq = synth.chr_standin(length=150_000, seed=31, n_head=3_000, n_tail=700, repeats_per_mbp=4000).tobytes()
    contig_like = seq
    seq = seq.strip(b"Nn")     # prf_scan_literal is the bare lane: the N-trimming of reference :40-46 is the caller's
    seq = seq[:60_000] + b"A" * 20_000 + seq[60_000:90_000] + b"N" * 777 + seq[90_000:]
    # (motif sizes up to 63 take the 64-positions-per-thread kernel, larger ones the byte routine: (1, 70) and (60, 66) cross the
    # split; span 150 is beyond what a thread's 128 positions can decide)
    for kmin, kmax, span in [(1, 12, 9), (3, 30, 40), (7, 7, 1), (1, 4, 2), (1, 70, 9), (5, 63, 150), (60, 66, 1), (2, 40, 64)]:
        rows, stats = ctx.scan_literal(seq, kmin, kmax, 1, span)
        assert stats.path == 2
        got = [(int(r["start"]), int(r["end"]), int(r["k"])) for r in rows]
        want = [(s, e, ml) for s, e, ml, _k in prf_oracle.detect_rows(seq, kmin, kmax, 1, span)]
        assert got == want, (kmin, kmax, span, len(got), len(want))
        assert len(got) >= 20 or kmin >= 5
    for kmin, kmax, r, span in [(1, 50, 3, 9), (2, 6, 2, 1), (1, 100, 4, 30)]:
        lit, _ = ctx.scan_literal(seq, kmin, kmax, r, span)
        fused, _ = ctx.scan([seq], kmin, kmax, r, span)
        a = [(int(x["start"]), int(x["end"]), int(x["k"])) for x in lit]
        assert a == [(int(x["start"]), int(x["end"]), int(x["k"])) for x in fused]
        assert a == oracle_rows(seq, kmin, kmax, r, span)
        assert len(a) >= 20
    contigs = [b"NNNN" + seq[3_000:9_000] + b"nn", b"", b"NNN", seq[70_000:70_500], b"ACGTRYACGTRYACGTRYnn", contig_like[:12_000]]
    rows, stats = ctx.scan(contigs, 1, 8, 1, 5)
    got = {}
    for c, s, e, k in rows_as_tuples(rows):
        got.setdefault(c, []).append((s, e, k))
    for i, cs in enumerate(contigs):
        want = [(s, e, ml) for s, e, ml, _k in prf_oracle.detect_rows(cs, 1, 8, 1, 5)]
        assert got.get(i, []) == want, i
    # (d) the same on a RESIDENT genome (prf_scan_genome): the bytes are rebuilt on the device from the packed planes
    g = ctx.load(contigs, 8)
    try:
        rows_g, stats_g = g.scan(1, 8, 1, 5)
        assert stats_g.path == 2 and rows_as_tuples(rows_g) == rows_as_tuples(rows) and len(rows_g) > 100
    finally:
        g.free()


def test_adversarial(detect, golden_adversarial):
    for case in golden_adversarial:
        assert outcome(detect, case["seq"], case["settings"]) == expected(case), case["tag"]


def test_chr22_clusters_as_one_multi_contig_genome(ctx, golden_clusters):
    """8000 real-genome known-answer clusters, loaded as 8000 contigs of one resident genome."""
    seqs = [seq.encode() for _pos, seq, _want in golden_clusters]
    rows, stats = ctx.scan(seqs, 1, 6, 3, 9)
    got = {}
    for c, s, e, k in rows_as_tuples(rows):
        got.setdefault(c, []).append((s, e, golden_clusters[c][1][s:s + k]))
    for i, (_pos, _seq, want) in enumerate(golden_clusters):
        assert got.get(i, []) == want, i
    assert stats.n_hits == sum(len(w) for _p, _s, w in golden_clusters)


def test_synthetic_golden(detect):
    from oracle import prf_oracle
    files = sorted(glob.glob(os.path.join(GOLDEN, "synth_*.json")))
    assert len(files) >= 5
    for path in files:
        with open(path) as f:
            g = json.load(f)
        seq = prf_oracle.synth(g["n"], g["seed"]).decode()
        assert outcome(detect, seq, g["settings"]) == ("ok", g["rows"]), path


def test_generic_kernel_param_sweep_vs_oracle(ctx):
    """Seeded inputs at sizes the oracle finishes in seconds, many parameter sets, forced generic kernel."""
    import prf_native
    import synth
    seq = synth.chr_standin(length=400_000, seed=11, n_head=30_000, n_tail=2_000, repeats_per_mbp=4000).tobytes()
    g = ctx.load([seq], 130)
    try:
        for kmin, kmax, r, span in [(1, 50, 3, 9), (2, 6, 3, 9), (1, 6, 3, 9), (1, 100, 3, 9), (5, 64, 2, 1), (1, 130, 4, 30),
                                    (63, 65, 2, 5), (1, 20, 3, 12), (1, 10, 3, 7), (3, 3, 2, 100)]:
            rows, stats = g.scan(kmin, kmax, r, span, flags=prf_native.SCAN_FORCE_GENERIC)
            assert stats.path == 0
            got = [(s, e, k) for _c, s, e, k in rows_as_tuples(rows)]
            assert got == oracle_rows(seq, kmin, kmax, r, span), (kmin, kmax, r, span)
    finally:
        g.free()


VSPECS = [(1, 50, 3, 9), (1, 100, 3, 9), (2, 6, 3, 9), (1, 6, 3, 9), (1, 7, 3, 6), (2, 10, 3, 6), (1, 10, 3, 6),
          (1, 20, 3, 12), (1, 10, 3, 7), (5, 64, 2, 1), (1, 30, 2, 20), (3, 3, 2, 100), (17, 17, 5, 1), (1, 130, 4, 30),
          (63, 65, 2, 5), (7, 9, 2, 1), (1, 14, 2, 2), (30, 130, 2, 200), (2, 40, 6, 60)]


def test_fused_bit_sliced_kernel_param_sweep_vs_oracle_and_generic(ctx):
    """The fused bit-sliced (vertical) kernel over many parameter sets (exact tasks for every M 1..14, group
    tasks, ranges that mix both), against the oracle and against the generic kernel; the input has N blocks
    that start and end inside tiles, so both the clean and the not-ACGT tile variants run, and planted
    repeats that cross stream and tile edges."""
    import prf_native
    import synth
    seq = bytearray(synth.chr_standin(length=700_000, seed=5, n_head=70_000, n_tail=3_000, repeats_per_mbp=5000).tobytes())
    seq[200_000:200_400] = b"N" * 400                       # N block inside a tile
    seq[65_536 * 4 - 40:65_536 * 4 + 60] = b"ACG" * 33 + b"A"  # run across a tile edge
    seq[65_536 * 5 - 9:65_536 * 5 + 9] = b"T" * 18          # homopolymer across a tile edge
    seq[300_000:300_000 + 2048 * 3] = (b"CAGT" * 2048)[:2048 * 3]  # run longer than several streams
    seq[400_000] = ord("N")                                  # single N
    seq = bytes(seq)
    g = ctx.load([seq], 130)
    try:
        for kmin, kmax, r, span in VSPECS:
            rows, stats = g.scan(kmin, kmax, r, span)
            assert stats.path == 1, (kmin, kmax, r, span)
            got = [(s, e, k) for _c, s, e, k in rows_as_tuples(rows)]
            assert got == oracle_rows(seq, kmin, kmax, r, span), (kmin, kmax, r, span)
            rows2, stats2 = g.scan(kmin, kmax, r, span, flags=prf_native.SCAN_FORCE_GENERIC)
            assert stats2.path == 0
            assert rows_as_tuples(rows2) == rows_as_tuples(rows)
    finally:
        g.free()


def test_widest_lds_image_with_n_blocks_kmax_260_to_480(ctx):
    """Motif sizes above ~260 take the widest LDS image of the fused kernel (80 virtual lanes: 16 extra lanes x 8 row
    groups x 3 planes = 384 staged slots on a tile with N in reach, more than one round of the 256 threads).  N blocks
    inside tiles, at tile edges and at the contig's end, long-period repeats next to them: fused == oracle == generic."""
    import prf_native
    import synth
    n = 330_000
    seq = bytearray(synth.chr_standin(length=n, seed=41, n_head=9_000, n_tail=700, repeats_per_mbp=3000).tobytes())
    rng = np.random.default_rng(41)
    for p, ln in ((65_536 - 3, 9), (100_000, 1), (131_072 + 2_000, 300), (200_000, 5_000), (262_144 - 1_000, 1_001)):
        seq[p:p + ln] = b"N" * ln
    for p, k, copies in ((20_000, 261, 3.2), (64_000, 300, 4), (99_000, 333, 3), (130_500, 470, 3.1), (150_000, 480, 5),
                         (196_000, 279, 3.5), (258_000, 400, 3), (n - 3_000, 264, 3.3), (40_000, 128, 9), (300_000, 97, 4)):
        motif = bytes(rng.choice(list(b"ACGT"), size=k).astype(np.uint8))
        body = (motif * (int(copies) + 2))[:int(k * copies)]
        seq[p:p + len(body)] = body
    seq = bytes(seq)
    assert b"N" * 300 in seq
    g = ctx.load([seq], 480)
    try:
        for kmin, kmax, r, span in ((1, 260, 3, 9), (250, 300, 3, 9), (1, 480, 3, 9), (257, 480, 2, 700), (470, 480, 4, 1)):
            assert prf_native.plan_describe(kmin, kmax, r, span)["nc"] == 80
            rows, stats = g.scan(kmin, kmax, r, span)
            assert stats.path == 1, (kmin, kmax, r, span)
            got = [(s, e, k) for _c, s, e, k in rows_as_tuples(rows)]
            assert got == oracle_rows(seq, kmin, kmax, r, span), (kmin, kmax, r, span)
            rows2, stats2 = g.scan(kmin, kmax, r, span, flags=prf_native.SCAN_FORCE_GENERIC)
            assert stats2.path == 0 and rows_as_tuples(rows2) == rows_as_tuples(rows)
        assert max(k for _s, _e, k in got) >= 470
    finally:
        g.free()


def test_symbols_other_than_acgtn_are_ordinary_symbols(ctx, detect, golden_iupac):
    """The reference compares characters: R == R matches, R != A does not, only the literal N never matches
    (utils/perfect_repeat_tracker.py:53, :83).  (1) The reference's own outputs on 277 sequences with IUPAC letters
    (tests/golden/iupac.jsonl.gz).  (2) A multi-contig genome with such letters sprinkled in -- single ones, runs, repeats whose
    motifs hold them, at tile edges, next to N blocks -- on the fused path (tiles with such a symbol in reach go to the generic
    kernels, the rest stays on the fused kernel), on the forced generic path, and cut into shares: all equal the oracle."""
    import multi_gpu
    import prf_native
    import synth
    for case in golden_iupac:
        assert outcome(detect, case["seq"], case["settings"]) == expected(case), case["tag"]
    rng = np.random.default_rng(8)
    seqs = []
    for ci, n in enumerate((3_500_000, 70_000, 300_000)):
        s = bytearray(synth.standin2(n, 90 + ci).tobytes())
        for _ in range(12 if ci != 2 else 0):                        # contig 2 stays plain: whole contigs without such symbols
            p = int(rng.integers(20_000, n - 5_000))
            kind = int(rng.integers(0, 5))
            if kind == 0:
                s[p] = ord("R")
            elif kind == 1:
                s[p:p + 30] = b"Y" * 30
            elif kind == 2:
                s[p:p + 60] = (b"ACR" * 20)
            elif kind == 3:
                s[p:p + 90] = (b"GATTAKMCW" * 10)
            else:
                s[p:p + 40] = b"wsWS" * 10
        if ci == 0:
            for edge in (65_536, 131_072, 196_608, 10 * 65_536):     # at and across tile edges; next to an N block
                s[edge - 7:edge + 8] = b"RY" * 7 + b"R"
            s[3 * 65_536 + 100:3 * 65_536 + 130] = b"N" * 30
            s[3 * 65_536 + 130:3 * 65_536 + 190] = b"AYG" * 20
            s[5 * 65_536 - 1] = ord("K")
            s[7 * 65_536] = ord("M")
            s[8 * 65_536 - 40:8 * 65_536 + 40] = b"CAGR" * 20
        seqs.append(bytes(s))
    want = [(c, a, b, k) for c, s in enumerate(seqs) for a, b, k in oracle_rows(s, 1, 50, 3, 9)]
    g = ctx.load(seqs, 50)
    try:
        rows, st = g.scan(1, 50, 3, 9)
        assert st.path == 1 and st.sorted_on_device == 0            # two row sets (fused tiles, generic tiles): merged on the host
        assert rows_as_tuples(rows) == want
        assert any(b"R"[0] in seqs[c][a:a + k].upper() or b"Y"[0] in seqs[c][a:a + k].upper() for c, a, _b, k in want)
        cls0 = g.tile_classes(0)
        assert 3 in cls0 and 0 in cls0 and 3 not in g.tile_classes(2)
        rows2, st2 = g.scan(1, 50, 3, 9, flags=prf_native.SCAN_FORCE_GENERIC)
        assert st2.path == 0 and rows_as_tuples(rows2) == want
        for world in (2, 5):
            shares = multi_gpu.plan_parts([len(s) for s in seqs], world, prf_native.tile_positions(),
                                          [g.tile_classes(c) for c in range(len(seqs))])
            got = []
            for share in shares:
                g.select(share)
                r, _ = g.scan(1, 50, 3, 9)
                got += rows_as_tuples(r)
            assert got == want, world
        g.select(None)
        with pytest.raises(prf_native.PrfError):
            g.scan_async(1, 50, 3, 9)                               # the pipelined call has no second pass: refused, not wrong
    finally:
        g.free()


def test_very_long_runs_flush_the_candidate_lists(ctx):
    """An all-A contig is one run for every k: every stream start inside it is a (spurious) candidate, the
    per-wave candidate lists of the fused kernel fill up over and over and are verified on the spot; the
    single real row per k is found by a walk of 300 000 positions."""
    seq = b"A" * 300_000 + b"C" + b"GT" * 40_000
    rows, stats = ctx.scan([seq], 1, 50, 3, 9)
    assert stats.path == 1
    got = [(s, e, k) for _c, s, e, k in rows_as_tuples(rows)]
    assert got == [(0, 300_000, 1), (300_001, 380_001, 2)]
    assert stats.n_candidates > 10_000   # (stream, exact task) flags + records of the group tasks, whose lists fill up and flush


def test_row_slab_overflow_grows_the_slabs(ctx):
    """More rows in one tile than the default per-tile row slab holds (1024): the scan must grow the slabs and retry on the
    fused path."""
    import collections
    unit = b"ACACACACACAC" + b"GTTGCAGATCCGTAGCTAGGCTAACGTTAG"
    seq = unit * 5000
    rows, stats = ctx.scan([seq], 1, 6, 3, 9)
    assert stats.path == 1
    got = [(s, e, k) for _c, s, e, k in rows_as_tuples(rows)]
    assert got == oracle_rows(seq, 1, 6, 3, 9)
    per_tile = collections.Counter(s // 65536 for s, _e, _k in got)
    assert max(per_tile.values()) > 1024


def test_errors_cross_the_boundary_cleanly(ctx):
    import prf_native
    with pytest.raises(prf_native.PrfError) as info:
        ctx.scan([b"ACGT-*ACGT"], 1, 5, 3, 9)                # letters only: anything else is refused, never guessed
    assert info.value.code == prf_native.PRF_ESYMBOL and "position 4" in info.value.message
    rows, _ = ctx.scan([b"ACGTRYACGT"], 1, 5, 3, 9)          # IUPAC letters are ordinary symbols (see the test below)
    assert len(rows) == 0
    g = ctx.load([b"ACGT" * 40000], 5)
    try:
        g.select([(0, 0, 65536)])
        with pytest.raises(prf_native.PrfError) as info:     # min_repeats == 1 scans whole contigs: not with a selection of parts
            g.scan(1, 5, 1, 9)
        assert info.value.code == prf_native.PRF_EUNSUPPORTED
        g.select([])
        rows, stats = g.scan(1, 5, 1, 9)
        assert rows_as_tuples(rows) == [(0, 0, 160000, 4)] and stats.path == 2
    finally:
        g.free()
    rows, stats = ctx.scan([b"ACGT"], 1, 5, 1, 9)            # prf_scan serves it on the literal lane
    assert len(rows) == 0 and stats.path == 2
    rows, _ = ctx.scan([b"NNACGTACGTAC"], 1, 5, 1, 9)
    assert rows_as_tuples(rows) == [(0, 2, 12, 4)]
    with pytest.raises(prf_native.PrfError) as info:         # the reference's IndexError (tracker :87): motif size > len + 1
        ctx.scan_literal(b"ACG", 1, 9, 1, 1)
    assert info.value.code == prf_native.PRF_EINDEX
    with pytest.raises(prf_native.PrfError) as info:
        ctx.scan_literal(b"ACGT-ACGT", 1, 5, 1, 1)
    assert info.value.code == prf_native.PRF_ESYMBOL and "position 4" in info.value.message
    with pytest.raises(prf_native.PrfError) as info:         # the position names the contig's own coordinate, N trimmed or not
        ctx.scan([b"NNNACGT-ACGT"], 1, 5, 1, 1)
    assert info.value.code == prf_native.PRF_ESYMBOL and "position 7" in info.value.message
    with pytest.raises(prf_native.PrfError) as info:
        ctx.scan([b"ACGT"], 0, 5, 3, 9)
    assert info.value.code == prf_native.PRF_EINVAL
    rows, _ = ctx.scan([], 1, 5, 3, 9)
    assert len(rows) == 0
    rows, _ = ctx.scan([b"", b"ACACACACACAC", b""], 1, 5, 3, 9)
    assert rows_as_tuples(rows) == [(1, 0, 12, 2)]


def test_full_size_chr22_standin_vs_oracle(ctx):
    """BASELINE config C2 at full size (50 818 468 bp, motif 1-50): rows bit-exact against the oracle."""
    import synth
    seq = synth.chr_standin().tobytes()
    rows, stats = ctx.scan([seq], 1, 50, 3, 9)
    got = [(s, e, k) for _c, s, e, k in rows_as_tuples(rows)]
    # size-independent properties first (cheap, and they localise a failure)
    starts = np.array([g[0] for g in got])
    ends = np.array([g[1] for g in got])
    ks = np.array([g[2] for g in got])
    assert len(got) > 50_000
    assert np.all((starts[1:] > starts[:-1]) | ((starts[1:] == starts[:-1]) & (ends[1:] > ends[:-1])))  # strictly sorted
    assert np.all(ends - starts >= np.maximum(3 * ks, 9))
    assert starts.min() >= 10_510_000 and ends.max() <= len(seq) - 10_000
    want = oracle_rows(seq, 1, 50, 3, 9)
    assert got == want


def test_device_generated_contigs_equal_the_oracle_generator(ctx):
    """prf_genome_synth writes the SURVEY 8(d) generator's bases on the device: the rows of a generated genome
    equal the oracle's rows on the oracle's own generator output (two contigs, lengths that are not multiples
    of the 16-base store, different seeds)."""
    from oracle import prf_oracle
    lens, seeds = [333_337, 70_001], [99, 12345678901234567]
    g = ctx.synth(lens, seeds, 100)
    try:
        assert g.positions == sum(lens)
        rows, stats = g.scan(1, 100, 3, 9)
        assert stats.path == 1
        want = [(c, s, e, k) for c, (n, sd) in enumerate(zip(lens, seeds))
                for s, e, _m, k in prf_oracle.detect_rows(prf_oracle.synth(n, sd), 1, 100, 3, 9)]
        assert rows_as_tuples(rows) == want and len(want) > 100
    finally:
        g.free()


def test_config_c5_share_random_1250mbp_motif_1_100(ctx):
    """BASELINE config C5, one GPU's share (10 Gbp / 8 = 1.25 Gbp of random ACGT generated on the device,
    motif 1-100).  Too long for the oracle as a whole, so: (1) size-independent properties of the row set,
    (2) the fused and the generic kernel -- independent implementations -- agree row for row, (3) on three
    2 Mbp windows the oracle's rows that lie strictly inside the window equal the GPU's rows there (a maximal
    run strictly inside a window is maximal in the whole contig)."""
    import prf_native
    from oracle import prf_oracle
    n, seed = 1_250_000_000, 5
    g = ctx.synth([n], [seed], 100)
    try:
        rows, stats = g.scan(1, 100, 3, 9)
        assert stats.path == 1 and stats.positions == n
        starts = rows["start"].astype(np.int64)
        ends = rows["end"].astype(np.int64)
        ks = rows["k"].astype(np.int64)
        assert len(rows) > 200_000
        assert np.all((starts[1:] > starts[:-1]) | ((starts[1:] == starts[:-1]) & (ends[1:] > ends[:-1])))
        assert np.all(ends - starts >= np.maximum(3 * ks, 9)) and ends.max() <= n and ks.max() <= 100
        rows2, stats2 = g.scan(1, 100, 3, 9, flags=prf_native.SCAN_FORCE_GENERIC)
        assert stats2.path == 0 and np.array_equal(rows, rows2)
        win, margin = 2_000_000, 1_000
        for off in (0, 617_283_951, n - win):
            chunk = prf_oracle.synth(win, seed, start=off)
            want = [(s + off, e + off, k) for s, e, _m, k in prf_oracle.detect_rows(chunk, 1, 100, 3, 9)
                    if s >= margin and e <= win - margin]
            sel = (starts >= off + margin) & (ends <= off + win - margin)
            got = list(zip(starts[sel].tolist(), ends[sel].tolist(), ks[sel].tolist()))
            assert got == want and len(want) > 300, off
    finally:
        g.free()


def test_config_c5_as_specified_one_sequence_of_10_gbp_seed_2026(ctx):
    """BASELINE config C5 exactly as SURVEY 8(d) specifies it: ONE sequence of 10^10 bp uniform ACGT, seed 2026, motif 1-100,
    generated on the device, scanned on ONE GPU.  Far too long for the oracle as a whole, so: (1) size-independent properties of
    the row set (count in the expected band, sorted, thresholds, in range, rows leave the device sorted); (2) the fused and the
    generic kernel -- independent implementations -- agree row for row; (3) on windows at the start, in the middle (beyond
    2^32) and at the end the oracle's rows that lie strictly inside the window equal the GPU's rows there; (4) the 8-GPU
    sharding rehearsed on this one GPU: the eight shares of multi_gpu.plan_parts, scanned one after the other, concatenate
    to exactly the whole scan."""
    import multi_gpu
    import prf_native
    from oracle import prf_oracle
    n, seed = 10_000_000_000, 2026
    g = ctx.synth([n], [seed], 100)
    try:
        rows, stats = g.scan(1, 100, 3, 9)
        assert stats.path == 1 and stats.positions == n and stats.sorted_on_device == 1
        starts, ends, ks = rows["start"].astype(np.int64), rows["end"].astype(np.int64), rows["k"].astype(np.int64)
        assert 2_000_000 < len(rows) < 2_500_000                       # ~210-235 rows / Mbp
        assert np.all((starts[1:] > starts[:-1]) | ((starts[1:] == starts[:-1]) & (ends[1:] > ends[:-1])))
        assert np.all(ends - starts >= np.maximum(3 * ks, 9)) and ends.max() <= n and ks.max() <= 100 and starts.max() > 9_990_000_000
        rows2, stats2 = g.scan(1, 100, 3, 9, flags=prf_native.SCAN_FORCE_GENERIC)
        assert stats2.path == 0 and np.array_equal(rows, rows2)
        del rows2
        win, margin = 1_500_000, 1_000
        for off in (0, 4_294_967_296 - win // 2, n - win):
            chunk = prf_oracle.synth(win, seed, start=off)
            want = [(s + off, e + off, k) for s, e, _m, k in prf_oracle.detect_rows(chunk, 1, 100, 3, 9)
                    if (s >= margin or off == 0) and (e <= win - margin or off + win == n)]
            sel = ((starts >= off + margin) | (off == 0)) & ((ends <= off + win - margin) | (off + win == n)) & (starts >= off) & (ends <= off + win)
            got = list(zip(starts[sel].tolist(), ends[sel].tolist(), ks[sel].tolist()))
            assert got == want and len(want) > 250, off
        shares = multi_gpu.plan_parts([n], 8, prf_native.tile_positions(), [g.tile_classes(0)])
        parts = []
        for share in shares:
            g.select(share)
            r8, st8 = g.scan(1, 100, 3, 9)
            assert st8.path == 1 and abs(int(st8.positions) - n // 8) < 2 * 65_536
            parts.append(r8)
        assert np.array_equal(np.concatenate(parts), rows)
    finally:
        g.free()


def test_full_size_properties_translation_and_reverse_complement(ctx):
    """Size-independent properties at BASELINE C2's full size (no oracle needed, so they also run where the oracle
    would take minutes): (1) translation -- the same contig behind 12 345 + 31 leading N has the same rows, shifted;
    every position changes its tile, stream, row and bit, so this is a test of all edge handling at once; (2) the
    rows of the reverse complement are the mirror image (maximal runs, thresholds and primitivity are symmetric;
    the motif text differs, the (start, end, k) set may not)."""
    import synth
    seq = synth.chr_standin().tobytes()
    n = len(seq)
    base, _ = ctx.scan([seq], 1, 50, 3, 9)
    assert len(base) > 50_000
    pad = 12_345 + 31
    moved, _ = ctx.scan([b"N" * pad + seq], 1, 50, 3, 9)
    assert len(moved) == len(base)
    assert np.array_equal(moved["start"], base["start"] + pad) and np.array_equal(moved["end"], base["end"] + pad)
    assert np.array_equal(moved["k"], base["k"])
    comp = bytes.maketrans(b"ACGTNacgtn", b"TGCANtgcan")
    rc, _ = ctx.scan([seq.translate(comp)[::-1]], 1, 50, 3, 9)
    mirror = np.zeros(len(base), dtype=base.dtype)
    mirror["start"], mirror["end"], mirror["k"] = n - base["end"], n - base["start"], base["k"]
    mirror = mirror[np.lexsort((mirror["end"], mirror["start"]))]
    assert np.array_equal(rc, mirror)


def test_multi_round_launch_with_n_blocks_takes_the_four_per_cu_build(ctx):
    """More than 768 tiles -> the 4-workgroups-per-CU build of the fused kernel, in which a tile with N in reach keeps its
    not-ACGT plane where clean tiles keep their linear window and verifies on the global planes.  70 Mbp stand-in with
    N blocks of many sizes sprinkled in (tile-aligned, straddling tile edges, single N): fused == generic row for row,
    and the oracle on windows around N-block edges."""
    import prf_native
    import synth
    from oracle import prf_oracle
    n = 70_000_000
    seq = bytearray(synth.chr_standin(length=n, seed=77, n_head=1_000_000, n_tail=5_000, repeats_per_mbp=2500).tobytes())
    rng = np.random.default_rng(5)
    blocks = []
    for i in range(60):
        p = int(rng.integers(1_100_000, n - 200_000))
        ln = int(rng.choice([1, 2, 7, 50, 1000, 65_536, 100_000]))
        if i % 5 == 0:
            p = (p // 65_536) * 65_536 - int(rng.integers(0, 3))      # at / just before a tile edge
        seq[p:p + ln] = b"N" * ln
        blocks.append((p, ln))
    for p, ln in blocks[:20]:                                            # repeats that run into an N block, or out of one
        seq[p - 30:p] = (b"CAG" * 10)
        seq[p + ln:p + ln + 24] = (b"AT" * 12)
    seq = bytes(seq)
    g = ctx.load([seq], 50)
    try:
        rows, st = g.scan(1, 50, 3, 9)
        assert st.path == 1 and len(rows) > 100_000
        rows2, st2 = g.scan(1, 50, 3, 9, flags=prf_native.SCAN_FORCE_GENERIC)
        assert st2.path == 0 and np.array_equal(rows, rows2)
        starts, ends, ks = rows["start"].astype(np.int64), rows["end"].astype(np.int64), rows["k"].astype(np.int64)
        for p, ln in blocks[:12]:
            lo, hi = max(0, p - 150_000), min(n, p + ln + 150_000)
            want = [(s + lo, e + lo, k) for s, e, _m, k in prf_oracle.detect_rows(seq[lo:hi], 1, 50, 3, 9)
                    if s >= 1_000 and e <= hi - lo - 1_000]
            sel = (starts >= lo + 1_000) & (ends <= hi - 1_000)
            assert list(zip(starts[sel].tolist(), ends[sel].tolist(), ks[sel].tolist())) == want, (p, ln)
    finally:
        g.free()


HG38_LENS = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717,
             133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285,
             58617616, 64444167, 46709983, 50818468, 156040895, 57227415, 16569, 0, 4262]


def test_config_c4_hg38_sized_genome_on_one_gpu(ctx):
    """BASELINE config C4's input shape on ONE GPU: 27 contigs with the hg38 primary-assembly lengths (plus chrM, an
    empty and a tiny one), 3.09 Gbp of device-generated random ACGT, motif 1-50, one resident genome, one launch.
    Checks: row-set properties per contig, fused == generic row for row, and the oracle on a window at the START
    and at the END of several contigs (contig edges are where the guard gaps and the end-of-contig rule act)."""
    import prf_native
    from oracle import prf_oracle
    seeds = [1000 + i for i in range(len(HG38_LENS))]
    g = ctx.synth(HG38_LENS, seeds, 50)
    try:
        rows, stats = g.scan(1, 50, 3, 9)
        assert stats.path == 1 and stats.positions == sum(HG38_LENS) and stats.n_launches == 2 and stats.sorted_on_device == 1
        contig = rows["contig"].astype(np.int64)
        starts = rows["start"].astype(np.int64)
        ends = rows["end"].astype(np.int64)
        ks = rows["k"].astype(np.int64)
        assert len(rows) > 500_000
        order_ok = (contig[1:] > contig[:-1]) | ((contig[1:] == contig[:-1]) & (
            (starts[1:] > starts[:-1]) | ((starts[1:] == starts[:-1]) & (ends[1:] > ends[:-1]))))
        assert order_ok.all()
        assert np.all(ends <= np.array(HG38_LENS, dtype=np.int64)[contig]) and np.all(ends - starts >= np.maximum(3 * ks, 9))
        rows2, stats2 = g.scan(1, 50, 3, 9, flags=prf_native.SCAN_FORCE_GENERIC)
        assert stats2.path == 0 and np.array_equal(rows, rows2)
        win = 1_500_000
        for c in (0, 7, 21, 23, 24, 26):
            n = HG38_LENS[c]
            for off in sorted({0, max(0, n - win)}):
                m = min(win, n - off)
                chunk = prf_oracle.synth(m, seeds[c], start=off)
                # rows that touch the window's inner edge are cut by the window, not by the contig: leave them out
                lo = off + (1_000 if off > 0 else 0)
                hi = off + m - (1_000 if off + m < n else 0)
                want = [(s + off, e + off, k) for s, e, _m, k in prf_oracle.detect_rows(chunk, 1, 50, 3, 9)
                        if s + off >= lo and e + off <= hi]
                sel = (contig == c) & (starts >= lo) & (ends <= hi)
                got = list(zip(starts[sel].tolist(), ends[sel].tolist(), ks[sel].tolist()))
                assert got == want, (c, off)
    finally:
        g.free()


BENCH_DEFAULT_ROWS = 5_699_373
BENCH_DEFAULT_ROWS_SHA256 = "0a363624c734e86942f2d8ccfc2822b2b4e3dcd4ced115f46fd503af8f865d7b"   # printed by bench.py as config.rows_sha256_rank0


def test_bench_default_workload_at_full_size(ctx):
    """The workload bench.py times by default: 25 contigs with the hg38 primary-assembly lengths, the stand-in
    recipe generated on the device (seeds 1000...), motif 1-50.  The number in BENCH_r03 is tied to THIS row set: fused ==
    generic row for row; the oracle on windows of the host recipe at both N-block edges and both gap edges of three contigs;
    the eight shares of an 8-GPU run (multi_gpu.plan_parts) concatenate to the whole scan; the SHA-256 of the row array is the
    one bench.py prints (config.rows_sha256_rank0)."""
    import hashlib
    import multi_gpu
    import prf_native
    import synth
    from oracle import prf_oracle
    seeds = [1000 + i for i in range(len(HG38_LENS[:25]))]
    lens = HG38_LENS[:25]
    g = ctx.standin(lens, seeds, 50)
    try:
        rows, st = g.scan(1, 50, 3, 9)
        assert st.path == 1 and st.sorted_on_device == 1 and st.n_launches == 2 and len(rows) == BENCH_DEFAULT_ROWS
        assert hashlib.sha256(np.ascontiguousarray(rows).tobytes()).hexdigest() == BENCH_DEFAULT_ROWS_SHA256
        gen, stg = g.scan(1, 50, 3, 9, flags=prf_native.SCAN_FORCE_GENERIC)
        assert stg.path == 0 and np.array_equal(rows, gen)
        contig, starts, ends, ks = (rows[f].astype(np.int64) for f in ("contig", "start", "end", "k"))
        win, margin = 400_000, 1_000
        for c in (0, 11, 22):
            n = lens[c]
            n_head, n_tail, gap_lo, gap_hi = synth.standin2_layout(n)
            for off in (0, gap_lo - win // 2, gap_hi - win // 2, n - win):      # both N blocks' edges, both gap edges
                chunk = synth.standin2(n, seeds[c], off, win).tobytes()
                want = [(s + off, e + off, k) for s, e, _m, k in prf_oracle.detect_rows(chunk, 1, 50, 3, 9)
                        if s >= margin and e <= win - margin]
                sel = (contig == c) & (starts >= off + margin) & (ends <= off + win - margin)
                got = list(zip(starts[sel].tolist(), ends[sel].tolist(), ks[sel].tolist()))
                assert got == want and len(want) > 100, (c, off)
        shares = multi_gpu.plan_parts(lens, 8, prf_native.tile_positions(), [g.tile_classes(c) for c in range(len(lens))])
        pieces = []
        for share in shares:
            g.select(share)
            r, s_ = g.scan(1, 50, 3, 9)
            assert s_.sorted_on_device == 1
            pieces.append(r)
        g.select([])
        assert min(len(p) for p in pieces) > 0.8 * len(rows) / 8 and np.array_equal(np.concatenate(pieces), rows)
    finally:
        g.free()


def test_config_c3_chr1_sized_stand_in_generated_on_the_device(ctx):
    """BASELINE config C3's workload as bench.py --workload chr1 times it: a chr1-sized contig (248 956 422 bp) of the stand-in
    recipe (N blocks at both ends, a 10 Mbp centromere-like gap, one planted repeat per 588 positions), generated on the
    device.  Too long for the oracle as a whole: row-set properties, fused == generic row for row, and the oracle on windows
    of the host recipe at every N-block edge (where tiles with N in reach, their late-staged windows and the tile
    ownership rule act) and in the middle."""
    import prf_native
    import synth
    from oracle import prf_oracle
    n, seed = synth.CHR1_LEN, 1
    g = ctx.standin([n], [seed], 50)
    try:
        rows, st = g.scan(1, 50, 3, 9)
        assert st.path == 1 and st.positions == n and st.sorted_on_device == 1
        starts, ends, ks = rows["start"].astype(np.int64), rows["end"].astype(np.int64), rows["k"].astype(np.int64)
        assert 400_000 < len(rows) < 520_000
        assert np.all((starts[1:] > starts[:-1]) | ((starts[1:] == starts[:-1]) & (ends[1:] > ends[:-1])))
        assert np.all(ends - starts >= np.maximum(3 * ks, 9))
        n_head, n_tail, gap_lo, gap_hi = synth.standin2_layout(n)
        assert starts.min() >= n_head and ends.max() <= n - n_tail and not np.any((starts < gap_hi) & (ends > gap_lo))
        rows2, st2 = g.scan(1, 50, 3, 9, flags=prf_native.SCAN_FORCE_GENERIC)
        assert st2.path == 0 and np.array_equal(rows, rows2)
        win, margin = 600_000, 1_000
        for off in (0, gap_lo - win // 2, gap_hi - win // 2, n // 2, n - win):
            chunk = synth.standin2(n, seed, off, win).tobytes()
            want = [(s + off, e + off, k) for s, e, _m, k in prf_oracle.detect_rows(chunk, 1, 50, 3, 9)
                    if s >= margin and e <= win - margin]
            sel = (starts >= off + margin) & (ends <= off + win - margin)
            got = list(zip(starts[sel].tolist(), ends[sel].tolist(), ks[sel].tolist()))
            assert got == want and len(want) > 400, off
    finally:
        g.free()


def test_deferred_timings_and_device_hand_off_with_count_record(ctx):
    """PRF_SCAN_DEFER_TIMING + prf_scan_timings (what bench.py's timed loop uses) and the device-to-device
    hand-off of rows with the trailing count record (what its multi-GPU gather ships)."""
    import prf_native
    import synth
    import torch
    seq = synth.chr_standin(length=400_000, seed=9, n_head=30_000, n_tail=2_000, repeats_per_mbp=3000).tobytes()
    g = ctx.load([seq], 50)
    try:
        rows, st = g.scan(1, 50, 3, 9)
        assert st.path == 1 and st.n_launches == 2 and st.phase1_ms > 0 and st.seq > 0
        seqs = []
        for _ in range(5):
            none, st2 = g.scan(1, 50, 3, 9, flags=prf_native.SCAN_DEFER_TIMING, fetch=False)
            assert none is None and st2.phase1_ms == 0 and st2.n_hits == len(rows)
            seqs.append(st2.seq)
        assert seqs == list(range(seqs[0], seqs[0] + 5))
        ms = ctx.scan_timings(seqs[0], 5)
        assert len(ms) == 5 and all(0 < m < 50 for m in ms)
        with pytest.raises(prf_native.PrfError):
            ctx.scan_timings(seqs[-1] + 1, 1)               # not issued yet
        cap = len(rows) + 7
        buf = torch.full((cap + 1, 3), -1, dtype=torch.int64, device="cuda")
        n = ctx.last_hits_to_device(buf.data_ptr(), cap, count_row=True)
        host = buf.cpu().numpy()
        assert n == len(rows) and host[cap].tolist() == [n, 0, 0] and (host[n:cap] == -1).all()
        got = np.ascontiguousarray(host[:n]).view(rows.dtype).reshape(-1)
        if st.sorted_on_device:                              # (a tile this dense may flush a record list: then the host sorts)
            assert np.array_equal(got, rows)                 # the rows leave the device sorted: no host sort in between
        got = got[np.lexsort((got["end"], got["start"], got["contig"]))]
        assert np.array_equal(got, rows)
        # pipelined scans: two in flight, collected one step late; rows of a collected scan through the hand-off
        pending, done = None, []
        for _ in range(7):
            sq = g.scan_async(1, 50, 3, 9)
            if pending is not None:
                done.append(ctx.scan_wait(pending))
            pending = sq
        done.append(ctx.scan_wait(pending))
        assert [int(d.n_hits) for d in done] == [len(rows)] * 7 and [d.seq for d in done] == sorted(d.seq for d in done)
        assert all(0 < m < 50 for m in ctx.scan_timings(done[0].seq, 7))
        buf.fill_(-1)
        torch.cuda.synchronize()
        assert ctx.last_hits_to_device(buf.data_ptr(), cap) == n
        got = np.ascontiguousarray(buf.cpu().numpy()[:n]).view(rows.dtype).reshape(-1)
        assert np.array_equal(got[np.lexsort((got["end"], got["start"], got["contig"]))], rows)
        with pytest.raises(prf_native.PrfError):
            a1, a2 = g.scan_async(1, 50, 3, 9), g.scan_async(1, 50, 3, 9)
            try:
                g.scan_async(1, 50, 3, 9)                    # a third one in flight is refused
            finally:
                ctx.scan_wait(a1), ctx.scan_wait(a2)
        rows5, _ = g.scan(1, 50, 3, 9)                       # the synchronous path is untouched by all that
        assert np.array_equal(rows5, rows)
        # row sink: the kernel compacts the rows straight into caller-owned memory, count record behind them;
        # both kernel paths, a fetch through the sink, and a sink that is too small
        for fl in (prf_native.SCAN_DEFAULT, prf_native.SCAN_FORCE_GENERIC):
            buf.fill_(-1)
            torch.cuda.synchronize()
            ctx.set_row_sink(buf.data_ptr(), cap)
            rows3, st3 = g.scan(1, 50, 3, 9, flags=fl)
            assert np.array_equal(rows3, rows)
            host = buf.cpu().numpy()
            assert host[cap].tolist() == [n, 0, 0] and (host[n:cap] == -1).all()
            got = np.ascontiguousarray(host[:n]).view(rows.dtype).reshape(-1)
            assert np.array_equal(got[np.lexsort((got["end"], got["start"], got["contig"]))], rows)
        ctx.set_row_sink(buf.data_ptr(), 10)
        with pytest.raises(prf_native.PrfError):
            g.scan(1, 50, 3, 9)
        ctx.set_row_sink(buf.data_ptr(), cap)                # the literal lane hands its rows over on the host: refused with a sink
        with pytest.raises(prf_native.PrfError) as info:
            ctx.scan([b"ACGTACGTACGT"], 1, 5, 1, 9)
        assert info.value.code == prf_native.PRF_EUNSUPPORTED
        ctx.set_row_sink(None, 0)
        rows4, _ = g.scan(1, 50, 3, 9)
        assert np.array_equal(rows4, rows)
    finally:
        ctx.set_row_sink(None, 0)
        g.free()


def test_rows_leave_the_device_sorted_and_dense_tiles_fall_back(ctx):
    """The fused path sorts a tile's rows in LDS and the gather concatenates the tiles in position order (reference
    perfect_repeat_finder.py:81 returns sorted rows): the raw device array equals the fetched rows.  Runs that start in the
    last positions of a tile and whose first examined group lies in the next tile are reported by the tile that holds the
    start (boundary pass): planted at many tile edges here.  A tile with more rows than its LDS list sorts on the host
    (stats.sorted_on_device == 0) -- same rows."""
    import synth
    import torch
    n = 3_000_000
    seq = bytearray(synth.standin2(n, 3).tobytes())
    rng = np.random.default_rng(12)
    for t in range(1, 45):
        k = int(rng.choice([8, 9, 12, 16, 17, 24, 31, 40, 50]))
        back = int(rng.integers(1, 8 * (4 if 2 * k >= 39 else (2 if 2 * k >= 23 else 1))))
        p = t * 65_536 - back
        motif = bytes(rng.choice(list(b"ACGT"), size=k).astype(np.uint8))
        seq[p:p + 4 * k] = (motif * 4)
    seq = bytes(seq)
    g = ctx.load([seq, seq[1_000_000:1_400_000]], 50)
    try:
        rows, st = g.scan(1, 50, 3, 9)
        assert st.path == 1 and st.sorted_on_device == 1
        want = [(0, s, e, k) for s, e, k in oracle_rows(seq, 1, 50, 3, 9)] + \
               [(1, s, e, k) for s, e, k in oracle_rows(seq[1_000_000:1_400_000], 1, 50, 3, 9)]
        assert rows_as_tuples(rows) == want
        buf = torch.zeros((len(rows) + 1, 3), dtype=torch.int64, device="cuda")
        assert ctx.last_hits_to_device(buf.data_ptr(), len(rows)) == len(rows)
        raw = np.ascontiguousarray(buf.cpu().numpy()[:len(rows)]).view(rows.dtype).reshape(-1)
        assert np.array_equal(raw, rows)
        # the 8-byte wire format of the multi-GPU gather: packed on the device, decoded on the host
        import multi_gpu
        import prf_native
        cap, side = len(rows) + 5, 4
        words = torch.zeros(cap + 1 + 3 * side, dtype=torch.int64, device="cuda")
        assert ctx.last_hits_packed_to_device(g, words.data_ptr(), cap, side) == len(rows)
        got = multi_gpu.unpack_rows(words.cpu().numpy(), cap, side, g.contig_bases(), prf_native.tile_positions())
        assert np.array_equal(got, rows)
        # the host reference encoder (what the 2-rank gloo test of the product path sends) writes the same words
        assert np.array_equal(multi_gpu.pack_rows(rows, cap, side, g.contig_bases(), prf_native.tile_positions()),
                              words.cpu().numpy().view(np.uint64))
        with pytest.raises(prf_native.PrfError):
            ctx.last_hits_packed_to_device(g, words.data_ptr(), len(rows) - 1, side)       # too small: refused
    finally:
        g.free()
    long_run = [b"ACGT" * 10 + b"A" * 70_000 + b"C", b"GT" * 40_000 + b"ACGTTGCA"]         # spans that do not fit 16 bits
    g = ctx.load(long_run, 50)
    try:
        rows, _ = g.scan(1, 50, 3, 9)
        assert rows_as_tuples(rows) == [(c, a, b, k) for c, s_ in enumerate(long_run) for a, b, k in oracle_rows(s_, 1, 50, 3, 9)]
        assert (0, 40, 70_040, 1) in rows_as_tuples(rows) and (1, 0, 80_000, 2) in rows_as_tuples(rows)
        words = torch.zeros(len(rows) + 1 + 3 * 4, dtype=torch.int64, device="cuda")
        ctx.last_hits_packed_to_device(g, words.data_ptr(), len(rows), 4)
        assert np.array_equal(multi_gpu.unpack_rows(words.cpu().numpy(), len(rows), 4, g.contig_bases(), prf_native.tile_positions()), rows)
        host_words = multi_gpu.pack_rows(rows, len(rows), 4, g.contig_bases(), prf_native.tile_positions())
        dev_words = words.cpu().numpy().view(np.uint64)
        n_side = int(dev_words[len(rows)]) >> 40                                           # (the device fills the side list in any order)
        assert n_side == 2 and np.array_equal(host_words[:len(rows) + 1], dev_words[:len(rows) + 1])
        assert sorted(map(tuple, host_words[len(rows) + 1:len(rows) + 1 + 3 * n_side].reshape(-1, 3).tolist())) == \
            sorted(map(tuple, dev_words[len(rows) + 1:len(rows) + 1 + 3 * n_side].reshape(-1, 3).tolist()))
    finally:
        g.free()
    # motif sizes above 511 do not fit the wire row: the hand-off is refused (it used to truncate k silently), the whole rows leave
    import synth
    wide = synth.synth_bases(600, 77).tobytes() * 3 + b"TTGACCA"
    g = ctx.load([wide], 600)
    try:
        rows, st = g.scan(590, 600, 2, 9)
        assert st.path == 0 and rows_as_tuples(rows) == [(0, a, b, k) for a, b, k in oracle_rows(wide, 590, 600, 2, 9)] and len(rows) >= 1
        words = torch.zeros(len(rows) + 1 + 3 * 4, dtype=torch.int64, device="cuda")
        with pytest.raises(prf_native.PrfError) as info:
            ctx.last_hits_packed_to_device(g, words.data_ptr(), len(rows), 4)
        assert info.value.code == prf_native.PRF_EUNSUPPORTED
        buf = torch.zeros((len(rows) + 1, 3), dtype=torch.int64, device="cuda")
        assert ctx.last_hits_to_device(buf.data_ptr(), len(rows)) == len(rows)
    finally:
        g.free()
    # dense tiles (round 3): 799 rows per tile -- more than the LDS row list holds (the densest tile of the reference's golden
    # chr22 BED has 748) -- are collected and ranked in the dead image region: still sorted on the device, no second scan
    dense = (b"ACACACACACAC" + b"GTTGCAGATCCGTAGCTAGGCTAACGTTAGCCATGGATCAAGCTTGCATGCCTGCAGGTCGACTCTAGAG") * 2500
    rows, st = ctx.scan([dense], 1, 6, 3, 9)
    assert st.sorted_on_device == 1 and [(s, e, k) for _c, s, e, k in rows_as_tuples(rows)] == oracle_rows(dense, 1, 6, 3, 9)
    # 2048 rows per tile: beyond the default slab (1024): the slabs grow, the scan is repeated, the rows are still sorted on the device
    denser = (b"ACACACACACAC" + b"GTTGCAGATCCGTAGCTAGG") * 6000
    rows, st = ctx.scan([denser], 1, 6, 3, 9)
    assert st.sorted_on_device == 1 and [(s, e, k) for _c, s, e, k in rows_as_tuples(rows)] == oracle_rows(denser, 1, 6, 3, 9)
    # 3277 rows per tile: more than the image region holds -- no genome does that -- the host sorts
    densest = (b"ACACACACACAC" + b"GTTGCAGA") * 9000
    rows, st = ctx.scan([densest], 1, 6, 3, 9)
    assert st.sorted_on_device == 0 and [(s, e, k) for _c, s, e, k in rows_as_tuples(rows)] == oracle_rows(densest, 1, 6, 3, 9)


def test_chr22_real_golden_bed_clusters_at_their_real_coordinates(ctx):
    """The workload with the REAL genome's row clustering: every cluster of the reference's golden
    benchmark/repeat_finder/chr22_repeats.bed planted at its real coordinate in the chr22 stand-in (synth.chr22_real; the
    clusters were re-run through the reference itself by oracle/gen_golden.py).  At the BED's own settings (motif 1-6, r 3,
    span 9) the whole scan equals the oracle's, holds every planted golden row, leaves the device sorted -- the densest tile has
    759 rows, more than the LDS row list: ranked in the dead image region -- and takes ONE pass (no slab overflow, no second
    scan).  At motif 1-50 fused == generic row for row."""
    import synth
    seq, planted = synth.chr22_real()
    lo, hi = 10_510_000, len(seq) - 10_000
    g = ctx.load([seq.tobytes()], 50)
    try:
        rows, st = g.scan(1, 6, 3, 9)
        assert st.path == 1 and st.sorted_on_device == 1 and st.n_launches == 2       # one scan kernel + one gather: no retry
        want = oracle_rows(seq[lo:hi].tobytes(), 1, 6, 3, 9)
        got = [(s, e, k) for _c, s, e, k in rows_as_tuples(rows)]
        assert got == [(s + lo, e + lo, k) for s, e, k in want] and len(got) == 76_570
        have = set(got)
        assert all((int(s), int(e), int(k)) in have for s, e, k in planted.tolist())
        import collections
        assert max(collections.Counter(s // 65_536 for s, _e, _k in got).values()) == 759
        rows50, st50 = g.scan(1, 50, 3, 9)
        assert st50.path == 1 and st50.sorted_on_device == 1 and st50.n_launches == 2
        gen50, stg = g.scan(1, 50, 3, 9, flags=prf_native_flags().SCAN_FORCE_GENERIC)
        assert stg.path == 0 and np.array_equal(rows50, gen50) and len(rows50) > len(rows)
    finally:
        g.free()


def test_parts_of_one_genome_scanned_separately_add_up_to_the_whole_scan(ctx):
    """prf_genome_select: every rank holds the genome and scans its parts (position ranges cut at tile multiples); a row
    belongs to the part that holds its first position.  The union over 2, 3, 8 and 13 shares == the whole scan, on the
    fused and on the generic path, with runs planted across every cut and N blocks around; a part list may be empty."""
    import multi_gpu
    import prf_native
    import synth
    tile = prf_native.tile_positions()
    assert tile == 65_536
    lens = [1_500_000, 70_000, 0, 900_001, 65_536, 131_072]
    seqs = [bytearray(synth.standin2(n, 50 + i).tobytes()) for i, n in enumerate(lens)]
    rng = np.random.default_rng(3)
    for s in seqs:
        for cut in range(tile, len(s), tile):
            for k, back in ((1, 5), (3, 2), (7, 30), (12, 3), (20, 12), (50, 31), (33, 140)):
                if rng.random() < 0.5:
                    motif = bytes(rng.choice(list(b"ACGT"), size=k).astype(np.uint8))
                    body = motif * (300 // k + 4)
                    s[cut - back:cut - back + len(body)] = body[:max(0, len(s) - (cut - back))]
    seqs[0][5 * tile - 10:5 * tile + 10] = b"N" * 20
    seqs = [bytes(s) for s in seqs]
    g = ctx.load(seqs, 60)
    try:
        for flags in (prf_native.SCAN_DEFAULT, prf_native.SCAN_FORCE_GENERIC):
            g.select(None)
            whole, st = g.scan(1, 60, 3, 9, flags=flags)
            assert st.positions == sum(lens) and len(whole) > 3_000
            classes = [g.tile_classes(c) for c in range(len(lens))]
            assert [len(c) for c in classes] == [-(-n // tile) for n in lens] and classes[0][4] == 1 and classes[0][0] == 1
            for world, cls in ((2, None), (3, classes), (8, classes), (13, None)):
                shares = multi_gpu.plan_parts(lens, world, tile, cls)
                assert len(shares) == world
                got, covered = [], 0
                for parts in shares:
                    g.select(parts)
                    rows, st = g.scan(1, 60, 3, 9, flags=flags)
                    assert st.positions == sum(e - b for _c, b, e in parts)
                    covered += st.positions
                    # rows start inside the parts that produced them
                    assert all(any(c == int(r["contig"]) and b <= int(r["start"]) < e for c, b, e in parts) for r in rows)
                    got.append(rows)
                assert covered == sum(lens)
                got = np.concatenate(got)
                assert np.array_equal(got, whole), (world, flags)       # shares are in genome order: concatenation is sorted
        g.select([(0, 0, tile)])
        with pytest.raises(prf_native.PrfError):
            g.select([(0, 100, tile)])                                    # not on a tile multiple
        with pytest.raises(prf_native.PrfError):
            g.select([(0, 0, 2 * tile), (0, tile, 3 * tile)])             # overlapping
        g.select(None)
        rows, _ = g.scan(1, 60, 3, 9)
        assert np.array_equal(rows, whole)
    finally:
        g.free()


def test_stand_in_genome_generated_on_the_device_equals_the_host_recipe(ctx):
    """prf_genome_standin writes synth.standin2's contigs on the device (N blocks, one planted repeat per 588-position
    slot): same rows as the host-generated bytes loaded over PCIe, and as the oracle on the small contig."""
    import synth
    lens, seeds = [2_300_017, 400_000, 12_345], [7, 8, 9]
    g = ctx.standin(lens, seeds, 50)
    try:
        rows, st = g.scan(1, 50, 3, 9)
        assert st.path == 1 and st.positions == sum(lens)
    finally:
        g.free()
    host = [synth.standin2(n, s).tobytes() for n, s in zip(lens, seeds)]
    rows2, _ = ctx.scan(host, 1, 50, 3, 9)
    assert np.array_equal(rows, rows2) and len(rows) > 4_000
    assert [(s, e, k) for c, s, e, k in rows_as_tuples(rows) if c == 1] == oracle_rows(host[1], 1, 50, 3, 9)


def test_cli_fasta_and_literal_on_the_gpu(tmp_path, monkeypatch, capsys):
    """The drop-in command line end to end on the GPU: multi-contig FASTA -> BED, literal sequence -> TSV."""
    import argparse
    import perfect_repeat_finder as prf
    import synth
    from oracle import prf_oracle
    monkeypatch.chdir(tmp_path)
    contigs = {"chrA": synth.chr_standin(length=90_000, seed=31, n_head=5_000, n_tail=500, repeats_per_mbp=3000).tobytes().decode(),
               "chrB": "acgt" * 10 + "N" * 7 + "CAGCAGCAGCAGCAGCAG" + "TTTTTTTTTTTTT",
               "chrC": ""}
    with open("toy.fasta", "wt") as f:
        for name, seq in contigs.items():
            f.write(f">{name} test\n")
            for i in range(0, len(seq), 60):
                f.write(seq[i:i + 60] + "\n")
    prf.main(["toy.fasta"])
    fs = argparse.Namespace(min_motif_size=1, max_motif_size=50, min_repeats=3, min_span=9)
    want = "".join(f"{name}\t{s}\t{e}\t{m}\n" for name, seq in contigs.items() for s, e, m in prf_oracle.detect_repeats(seq, fs))
    assert open("toy.bed").read() == want and want.count("\n") > 100
    # the same command under torch.distributed.run with two ranks (both on this GPU, gloo for the gather): contigs
    # sharded over the ranks, rank 0 writes the BED
    import subprocess
    import sys
    from conftest import PKG
    env = dict(os.environ, PRF_DIST_BACKEND="gloo", PRF_DEVICE="0")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", "29533", os.path.join(PKG, "perfect_repeat_finder.py"), "-o", "two_ranks",
                          "toy.fasta"], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    assert open("two_ranks.bed").read() == want and res.stdout.count("Wrote results to two_ranks.bed") == 1
    # min_repeats == 1 (the literal lane): whole FASTA in one process, and the same under two ranks (whole contigs per rank)
    fs1 = argparse.Namespace(min_motif_size=1, max_motif_size=12, min_repeats=1, min_span=10)
    want1 = "".join(f"{name}\t{s}\t{e}\t{m}\n" for name, seq in contigs.items() for s, e, m in prf_oracle.detect_repeats(seq, fs1))
    prf.main(["-max", "12", "--min-repeats", "1", "--min-span", "10", "-o", "one", "toy.fasta"])
    assert open("one.bed").read() == want1 and want1.count("\n") > 100 and want1 != want
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", "29534", os.path.join(PKG, "perfect_repeat_finder.py"), "-max", "12",
                          "--min-repeats", "1", "--min-span", "10", "-o", "one_two_ranks", "toy.fasta"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    assert open("one_two_ranks.bed").read() == want1
    prf.main(["-min", "2", "-max", "6", "CACACACACACAGGGTTTTTTTTTTT"])
    assert open("repeats.tsv").read() == "start_0based\tend\tmotif\n0\t12\tCA\n"
    # --stats (not in the reference): the same BED, the reference's stdout lines, then one JSON line
    import json
    assert "Found 1 repeats" in capsys.readouterr().out
    prf.main(["--stats", "-o", "with_stats", "toy.fasta"])
    lines = capsys.readouterr().out.strip().split("\n")
    st = json.loads(lines[-1])
    assert open("with_stats.bed").read() == want and lines[-2] == "Wrote results to with_stats.bed"
    assert st["positions"] == sum(len(v) for v in contigs.values()) and st["rows"] == want.count("\n") and st["kernel_path"] == "fused"
    assert st["algorithmic_bytes"] == (st["positions"] + 3) // 4 + 24 * st["rows"] and st["rows_sorted_on_device"] and st["scan_ms"] > 0


def test_resident_genome_footprint(ctx):
    """prf_genome_footprint: a resident genome of A, C, G, T, N holds 0.625 bytes per position of its coordinate space (three
    linear planes, two bit-sliced ones) plus tables of a few bytes per 65 536 positions; letters outside ACGTN add five planes."""
    import prf_native
    import synth
    seq = synth.chr_standin(length=3_000_000, seed=5, n_head=1000, n_tail=1000).tobytes()
    g = ctx.load([seq, seq[:700_000]], 50)
    try:
        dev_bytes, positions = g.footprint()
        assert positions % prf_native.tile_positions() == 0 and positions >= 3_700_000
        assert 0.625 <= dev_bytes / positions < 0.64
    finally:
        g.free()
    g = ctx.load([seq[:500_000] + b"RYK" + seq[500_000:900_000]], 50)
    try:
        dev_bytes, positions = g.footprint()
        assert 1.25 <= dev_bytes / positions < 1.27
    finally:
        g.free()


def test_randomised_differential_stress():
    """tools/stress_gpu.py for ~25 s: random multi-contig inputs x random parameter sets, GPU rows == oracle rows
    (a 240 s run of the same script covered 3 924 scans without a mismatch in round 1)."""
    import subprocess
    import sys
    from conftest import ROOT
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_gpu.py"), "25", "11"], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "0 mismatches" in res.stdout
