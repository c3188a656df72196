#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE itself.

Runs only in the build container (it needs /root/reference); the fixtures it
writes are plain data (inputs + the reference's outputs) and are what travels.
The reference's module-level ``import pyfastx`` (perfect_repeat_finder.py:2,
used only inside main()) is satisfied with an empty placeholder module because
pyfastx is not installed here; detect_repeats() never touches it.

    python3 oracle/gen_golden.py [--quick]

Fixture files (all under tests/golden/):
  ref_unit_tests.json     the reference's own known-answer vectors
                          (perfect_repeat_finder_tests.py:21-143), re-run here
  fuzz_small.jsonl.gz     random small cases incl. interval mode / min_repeats=1 / exceptions
  adversarial.jsonl.gz    tile/word-boundary, all-A, all-N, k > L, lowercase ...
  min_repeats_one.jsonl.gz  min_repeats == 1 (the regime outside the closed form): random cases incl. interval mode, lower
                          case, symbols other than ACGTN, motif sizes beyond the sequence (IndexError), N at both ends
  odd_intervals.jsonl.gz  interval bounds reversed / outside the sequence / on N, min_repeats 1-3
  iupac.jsonl.gz          symbols other than ACGTN (ordinary symbols to the reference), incl. at tile edges
  synth_*.json            SURVEY 8(d) synthetic sequences (by seed/length) + reference rows
  chr22_clusters.tsv.gz   known-answer clusters mined from the reference's golden BED
                          (benchmark/repeat_finder/chr22_repeats.bed), each re-run through the reference
  chr22_clusters_all.tsv.gz  (--only clusters_all) EVERY consistent cluster of that BED with its real coordinate
"""
import argparse
import gzip
import json
import os
import random
import sys
import time
import types

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, REF)
sys.modules.setdefault("pyfastx", types.ModuleType("pyfastx"))
import perfect_repeat_finder as ref  # noqa: E402
from utils.plot_utils import shift_string_by  # noqa: E402


def ns(kmin, kmax, r, span, interval=None):
    d = dict(min_motif_size=kmin, max_motif_size=kmax, min_repeats=r, min_span=span)
    if interval is not None:
        d["interval_start_0based"], d["interval_end"] = interval
    return d


def run_ref(seq, settings):
    fs = argparse.Namespace(**settings)
    try:
        rows = ref.detect_repeats(seq, fs)
        return {"status": "ok", "rows": [[s, e, m] for s, e, m in rows]}
    except (AssertionError, IndexError, ValueError) as exc:
        return {"status": type(exc).__name__}


def synth(n, seed, start=0):
    """SURVEY 8(d) generator in pure Python ints (slow, only used for <= few Mbp)."""
    mask = (1 << 64) - 1
    out = bytearray(n)
    acgt = b"ACGT"
    for j in range(n):
        z = (seed + (start + j + 1) * 0x9E3779B97F4A7C15) & mask
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
        z ^= z >> 31
        out[j] = acgt[z >> 62]
    return bytes(out)


# ----------------------------------------------------------------------------------------
def gen_ref_unit_tests():
    """The inputs of perfect_repeat_finder_tests.py, re-run through the reference here and
    stored with the expected values the reference's test file asserts."""
    cases = []

    def add(tag, seq, settings, expected):
        got = run_ref(seq, settings)
        exp = [[s, e, m] for s, e, m in expected]
        assert got["status"] == "ok" and got["rows"] == exp, (tag, seq, got, exp)
        cases.append({"tag": tag, "seq": seq, "settings": settings, "rows": exp})

    base = ns(1, 7, 3, 6)
    for motif in "A", "CA", "CAG", "CAGA", "CAGAT", "CAGATT", "CAGATTA", "CAGATTAG":
        seq = 6 * motif
        add("T1", seq, base, [(0, len(seq), motif)] if len(motif) <= 7 else [])
    for motif in "A", "CA", "CAG", "CAGA", "CAGAT", "CAGATT", "CAGATAT", "CAGATTAG":
        seq = 6 * motif + 10 * "TA"
        ta = (6 * len(motif) - 1, len(seq), "AT") if motif.endswith("A") else (6 * len(motif), len(seq), "TA")
        add("T2", seq, base, [(0, 6 * len(motif), motif), ta] if len(motif) <= 7 else [ta])
    add("T3", "A" * 9 + "C" * 11 + "G" * 10 + "T" * 9, base, [(0, 9, "A"), (9, 20, "C"), (20, 30, "G"), (30, 39, "T")])
    add("T4", "CA" * 9 + "GT" * 11 + "CG" * 10 + "TA" * 9, base,
        [(0, 18, "CA"), (18, 40, "GT"), (40, 60, "CG"), (60, 78, "TA")])
    add("T5", 7 * "A", ns(2, 7, 3, 6), [])
    seq = 7 * "A" + 7 * "AGAC" + 10 * "AAC" + "N" * 10
    add("T6", seq, ns(2, 10, 3, 6), [(7, 36, "AGAC"), (33, 65, "ACA")])
    add("T7", seq, ns(1, 10, 3, 6), [(0, 8, "A"), (7, 36, "AGAC"), (33, 65, "ACA")])
    for ob in range(1, 10):
        left = "T" + "A" * ob
        right = (ob + 1) * "A" + "T"
        seq = 9 * left + "T" + 11 * right
        add("T8", seq, ns(1, 20, 3, 12),
            [(0, len(left) * 10, left), (len(left) * 8 + 1, len(seq), shift_string_by(right, -1))])
    seq = "N" * 10 + 7 * "A" + "N" + 7 * "AGAC" + "NNNN" + 10 * "AAC" + "N" * 10
    add("T9", seq, ns(1, 10, 3, 7), [(10, 17, "A"), (18, 46, "AGAC"), (50, 80, "AAC")])
    seq = "GATGG" + "GGG" + "TGACATGACA" + "CAG" * 5 + "ACAGTTTTTTTTTT"
    add("T10", seq, ns(1, 100, 3, 3, interval=(5, 20)), [(5, 8, "G"), (18, 33, "CAG")])
    with open(os.path.join(OUT, "ref_unit_tests.json"), "w") as f:
        json.dump({"source": "reference perfect_repeat_finder_tests.py:21-143, re-run through the reference",
                   "cases": cases}, f, indent=0)
    print("ref_unit_tests:", len(cases))


def rand_seq(rng, L, alpha):
    seq = ""
    while len(seq) < L:
        if rng.random() < 0.3:
            m = "".join(rng.choice(alpha) for _ in range(rng.randint(1, 14)))
            seq += m * rng.randint(1, 9)
        else:
            seq += "".join(rng.choice(alpha) for _ in range(rng.randint(1, 25)))
    return seq[:L]


def gen_fuzz_small(n_cases):
    rng = random.Random(20261003)
    path = os.path.join(OUT, "fuzz_small.jsonl.gz")
    stats = {}
    with gzip.open(path, "wt") as f:
        for it in range(n_cases):
            alpha = rng.choice(["ACGT", "ACGT", "AC", "ACGTN", "ACGTacgtn", "A", "ACGTNN"])
            L = rng.randint(0, 400)
            seq = rand_seq(rng, L, alpha)
            kmin = rng.randint(1, 5)
            kmax = kmin + rng.randint(0, 40)
            mode = rng.random()
            r = rng.randint(2, 6)
            span = rng.randint(1, 40)
            interval = None
            if mode < 0.15:
                r = 1
            elif mode < 0.40 and L > 0:
                a = rng.randint(0, L)
                b = rng.randint(a, L)
                interval = (a, b)
            st = ns(kmin, kmax, r, span, interval)
            res = run_ref(seq, st)
            stats[res["status"]] = stats.get(res["status"], 0) + 1
            f.write(json.dumps({"seq": seq, "settings": st, **res}) + "\n")
    print("fuzz_small:", n_cases, stats)


def gen_min_repeats_one(n_cases):
    """min_repeats == 1: the rows depend on the text in front of a run, on the slice clamp at the end of the sequence and
    on negative-index wrap-around (utils/perfect_repeat_tracker.py:82-91), and N-trimming (:40-46) is not a no-op."""
    rng = random.Random(20261004)
    path = os.path.join(OUT, "min_repeats_one.jsonl.gz")
    stats = {}
    cases = []
    fixed = [("", 1, 1, 1), ("", 1, 3, 1), ("N", 1, 2, 1), ("NNNN", 1, 5, 1), ("A", 1, 1, 1), ("A", 1, 4, 1), ("AC", 1, 6, 2),
             ("NNACGTNN", 1, 10, 1), ("NNACGTNN", 6, 6, 1), ("ACACACAC", 1, 5, 3), ("ACACACAC", 2, 12, 1), ("acgtacgtNNacgt", 1, 8, 4),
             ("AAAAAAAAAA", 1, 12, 1), ("ANANANANA", 1, 4, 1), ("ACGTRYACGTRY", 1, 8, 2), ("NACGTACGTACGTN", 3, 20, 5)]
    for seq, kmin, kmax, span in fixed:
        cases.append((seq, ns(kmin, kmax, 1, span)))
        if seq:
            cases.append((seq, ns(kmin, kmax, 1, span, (0, len(seq)))))
            cases.append((seq, ns(kmin, kmax, 1, span, (len(seq) // 3, 2 * len(seq) // 3))))
    while len(cases) < n_cases:
        alpha = rng.choice(["ACGT", "ACGT", "AC", "ACGTN", "ACGTacgtn", "A", "ACGTNN", "ACGTRYN", "AN"])
        L = rng.choice([rng.randint(0, 12), rng.randint(0, 60), rng.randint(0, 300)])
        seq = rand_seq(rng, L, alpha)
        if rng.random() < 0.25:
            seq = "N" * rng.randint(0, 5) + seq + "n" * rng.randint(0, 5)
        L = len(seq)
        kmin = rng.randint(1, 6)
        kmax = kmin + rng.choice([0, rng.randint(0, 8), rng.randint(0, 40)])
        if rng.random() < 0.1:
            kmax = max(kmax, L + rng.randint(-2, 4))
        span = rng.choice([1, 1, rng.randint(1, 12), rng.randint(1, 40)])
        interval = None
        if rng.random() < 0.45 and L > 0:
            a = rng.randint(0, L)
            b = rng.randint(a, L)
            interval = (a, b)
        cases.append((seq, ns(kmin, max(kmax, kmin), 1, span, interval)))
    with gzip.open(path, "wt") as f:
        for seq, st in cases:
            res = run_ref(seq, st)
            stats[res["status"]] = stats.get(res["status"], 0) + 1
            f.write(json.dumps({"seq": seq, "settings": st, **res}) + "\n")
    print("min_repeats_one:", len(cases), stats)


def gen_odd_intervals(n_cases):
    """Intervals the command line never produces but detect_repeats() accepts: end in front of start, bounds outside the
    sequence (negative indices wrap, IndexError past the ends), N at the interval edges; min_repeats 1-3."""
    rng = random.Random(20261005)
    path = os.path.join(OUT, "odd_intervals.jsonl.gz")
    stats = {}
    with gzip.open(path, "wt") as f:
        for it in range(n_cases):
            L = rng.randint(0, 200)
            body = "".join(rng.choice("ACGTNn" if rng.random() < 0.5 else "ACac") for _ in range(L))
            seq = "N" * rng.randint(0, 6) + body + "N" * rng.randint(0, 6)
            L = len(seq)
            kmin = rng.randint(1, 4)
            st = ns(kmin, kmin + rng.randint(0, 12), rng.randint(1, 3), rng.randint(1, 12),
                    (rng.randint(-3, L + 3), rng.randint(-3, L + 3)))
            res = run_ref(seq, st)
            stats[res["status"]] = stats.get(res["status"], 0) + 1
            f.write(json.dumps({"seq": seq, "settings": st, **res}) + "\n")
    print("odd_intervals:", n_cases, stats)


def gen_adversarial():
    """Cases aimed at the GPU design's seams: 32/64-bit word edges, 2048-base stream
    edges and 65536-base tile edges, very long runs, N placement, k vs L."""
    rng = random.Random(7)
    cases = []

    def add(tag, seq, st):
        cases.append({"tag": tag, "seq": seq, "settings": st, **run_ref(seq, st)})

    d = ns(1, 50, 3, 9)
    add("empty", "", d)
    add("one", "A", d)
    add("allN", "N" * 300, d)
    add("allA_5000", "A" * 5000, ns(1, 12, 3, 9))
    add("k_gt_L", "ACGTACGTACGT", ns(1, 50, 3, 9))
    add("lowercase", "acacacacacacacacGTgtGTgtGTgtgtgt", ns(1, 6, 3, 9))
    add("N_inside_run", "CA" * 10 + "N" + "CA" * 10, ns(1, 6, 3, 9))
    add("N_is_never_equal", "N" * 40 + "ACGT" * 8 + "N" * 40, ns(1, 8, 3, 9))
    # runs starting / ending exactly on word, stream (32 bases x n) and tile (65536) edges
    for edge in (31, 32, 33, 63, 64, 65, 127, 128, 129, 2047, 2048, 2049, 4096):
        for motif in ("A", "CA", "CAG", "ACGTT", "AACCGGTTAC"):
            for reps in (3, 4, 9, 40):
                flank_l = rand_seq(rng, edge, "ACGT")
                body = motif * reps
                # make sure the flank does not extend the run
                seq = flank_l[:-1] + ("G" if motif[-1] != "G" else "T") + body + ("C" if motif[0] != "C" else "A") \
                    + rand_seq(rng, 70, "ACGT")
                add(f"edge{edge}_{motif}x{reps}", seq, ns(1, 12, 3, 9))
    # long runs crossing several 2048-base streams, every k a multiple of the period also matches
    for motif, total in (("A", 7000), ("CA", 9000), ("CAG", 6500), ("ACGTTGCA" + "T", 8000)):
        seq = rand_seq(rng, 1000, "ACGT") + (motif * (total // len(motif) + 1))[:total] + rand_seq(rng, 1200, "ACGT")
        add(f"long_{motif}_{total}", seq, ns(1, 30, 3, 9))
    # around the 65536 tile edge (kept short in k to bound the reference's run time)
    for motif in ("A", "GT", "AAC", "ACGTAGC"):
        pre = rand_seq(rng, 65536 - 25, "ACGT")
        seq = pre[:-1] + "C" + (motif * 40)[:50] + "G" + rand_seq(rng, 300, "ACGT")
        add(f"tile_edge_{motif}", seq, ns(1, 8, 3, 9))
    # repeats touching the contig end / start
    add("end_touch", rand_seq(rng, 200, "ACGT") + "CAG" * 12, ns(1, 10, 3, 9))
    add("start_touch", "CAG" * 12 + rand_seq(rng, 200, "ACGT"), ns(1, 10, 3, 9))
    add("whole", "ACGGT" * 30, ns(1, 50, 3, 9))
    # thresholds: exactly at / one below min_repeats and min_span for several settings
    for r in (2, 3, 5):
        for span in (1, 9, 20):
            for k in (1, 2, 3, 5, 7, 16, 33):
                motif = ("ACGT" * 10)[:k - 1] + "C" if k > 1 else "A"
                for extra in (-1, 0, 1):
                    need = max(r * k, span) + extra
                    body = (motif * (need // k + 2))[:max(need, 0)]
                    seq = "GTTG" + body + ("T" if body[-1:] != "T" else "G") + "GATTACA"
                    add(f"thr_r{r}_s{span}_k{k}_{extra}", seq, ns(1, 40, r, span))
    path = os.path.join(OUT, "adversarial.jsonl.gz")
    with gzip.open(path, "wt") as f:
        for c in cases:
            f.write(json.dumps(c) + "\n")
    print("adversarial:", len(cases))


def gen_iupac():
    """Symbols other than A, C, G, T, N are ORDINARY symbols to the reference: R == R matches, R != A does not, only the
    literal N never matches (utils/perfect_repeat_tracker.py:53, :83); .upper() folds case (perfect_repeat_finder.py:33).
    Cases: hand-made ones, random sequences over a mixed alphabet with planted repeats whose motifs hold IUPAC letters,
    and long ones with such repeats at and across the 65536-base tile edges of the GPU design (kept short in k to bound
    the reference's run time)."""
    rng = random.Random(11)
    cases = []

    def add(tag, seq, st):
        cases.append({"tag": tag, "seq": seq, "settings": st, **run_ref(seq, st)})

    add("R_run", "ACGT" + "R" * 12 + "ACGT", ns(1, 6, 3, 9))
    add("R_vs_N", "R" * 10 + "N" * 10 + "R" * 10, ns(1, 6, 3, 9))
    add("motif_with_Y", "GATTACA" + "ACY" * 7 + "TTGCA", ns(1, 10, 3, 9))
    add("motif_all_iupac", "ACGT" * 3 + "RYKM" * 6 + "ACGT" * 3, ns(1, 12, 3, 9))
    add("lowercase_iupac", "acgt" + "ryRYry" * 4 + "wsWS" * 5 + "n" * 4 + "acg", ns(1, 8, 3, 9))
    add("iupac_breaks_a_run", "CA" * 6 + "R" + "CA" * 6, ns(1, 6, 3, 9))
    add("iupac_next_to_N", "CAG" * 5 + "N" + "RAG" * 5 + "NRNRNRNRNRNR", ns(1, 6, 3, 9))
    add("every_letter", "".join(chr(c) * 11 for c in range(ord("A"), ord("Z") + 1)), ns(1, 4, 3, 9))
    add("non_primitive_iupac", "RR" * 10 + "A" + "RYRY" * 6, ns(1, 8, 3, 9))
    add("interval_iupac", "GATGG" + "RYR" * 9 + "ACAGTTTTTTTTTT", ns(1, 20, 3, 3, interval=(5, 20)))
    alpha = "ACGTACGTACGTRYKMSWBDHVN"
    for i in range(260):
        n = rng.choice([30, 80, 200, 600])
        seq = list(rand_seq(rng, n, alpha if i % 3 else "ACGTRYN"))
        for _ in range(rng.randint(0, 4)):
            k = rng.choice([1, 1, 2, 3, 4, 5, 7, 9, 12])
            motif = rand_seq(rng, k, "ACGTRYKMSW")
            copies = rng.choice([2, 3, 3, 4, 6, 9])
            p = rng.randrange(max(1, n - 1))
            body = (motif * copies + motif[:rng.randrange(k)])[:max(0, n - p)]
            seq[p:p + len(body)] = body
        seq = "".join(seq)
        if i % 5 == 0:
            seq = "".join(c.lower() if rng.random() < 0.4 else c for c in seq)
        add(f"rand{i}", seq, ns(rng.choice([1, 1, 2]), rng.choice([6, 12, 20]), rng.choice([2, 3, 3, 4]), rng.choice([1, 6, 9, 14])))
    # around the 65536-base tile edges: IUPAC motifs ending before / starting at / crossing the edge, a lone IUPAC letter in
    # the last and first positions of a tile, a plain repeat next to one
    for j, (motif, where) in enumerate((("R", -5), ("AY", -21), ("ACR", 0), ("ACGTR", -1), ("KM", -65), ("CAG", -40), ("WSW", 3))):
        pre = list(rand_seq(rng, 65536 + 400, "ACGT"))
        p = 65536 + where
        body = (motif * 30)[:48]
        pre[p - 1] = "C" if body[0] != "C" else "A"
        pre[p:p + len(body)] = body
        pre[p + len(body)] = "G" if body[-1] != "G" else "T"
        if motif == "CAG":
            pre[65535] = "R"
            pre[65536] = "Y"
        add(f"tile_edge_iupac_{j}_{motif}", "".join(pre), ns(1, 6, 3, 9))
    path = os.path.join(OUT, "iupac.jsonl.gz")
    with gzip.open(path, "wt") as f:
        for c in cases:
            f.write(json.dumps(c) + "\n")
    ok = sum(1 for c in cases if c["status"] == "ok")
    print("iupac:", len(cases), "cases,", ok, "ok,", sum(len(c.get("rows") or []) for c in cases), "rows")


def gen_synth(quick):
    specs = [(200_000, 22, ns(2, 6, 3, 9)), (200_000, 22, ns(1, 50, 3, 9))]
    if not quick:
        specs += [(1_000_000, 22, ns(2, 6, 3, 9)), (1_000_000, 22, ns(1, 50, 3, 9)), (400_000, 2026, ns(1, 100, 3, 9))]
    for n, seed, st in specs:
        t0 = time.time()
        seq = synth(n, seed).decode()
        res = run_ref(seq, st)
        dt = time.time() - t0
        name = f"synth_n{n}_seed{seed}_k{st['min_motif_size']}-{st['max_motif_size']}.json"
        with open(os.path.join(OUT, name), "w") as f:
            json.dump({"n": n, "seed": seed, "settings": st, "rows": res["rows"],
                       "head": seq[:64], "ref_seconds": round(dt, 2)}, f)
        print(name, len(res["rows"]), "rows", round(dt, 1), "s")


def gen_chr22_clusters(max_clusters, out_name="chr22_clusters.tsv.gz"):
    """Rows of the reference's golden BED that overlap or abut determine the sequence under
    them; each cluster + one flank base per side is a real-genome known-answer vector.
    max_clusters None: every cluster (chr22_clusters_all.tsv.gz -- the fixture synth.chr22_real() plants at the real
    coordinates to get a workload with the real genome's row clustering)."""
    bed = os.path.join(REF, "benchmark", "repeat_finder", "chr22_repeats.bed")
    rows = []
    with open(bed) as f:
        for line in f:
            c, s, e, m = line.rstrip("\n").split("\t")
            rows.append((int(s), int(e), m))
    clusters, cur, cur_end = [], [], -1
    for s, e, m in rows:
        if cur and s > cur_end:
            clusters.append(cur)
            cur = []
        cur.append((s, e, m))
        cur_end = max(cur_end, e) if len(cur) > 1 else e
    if cur:
        clusters.append(cur)
    st = ns(1, 6, 3, 9)
    rng = random.Random(5)
    big = [c for c in clusters if len(c) >= 3]
    small = [c for c in clusters if len(c) < 3]
    rng.shuffle(small)
    chosen = clusters if max_clusters is None else big + small[:max(0, max_clusters - len(big))]
    chosen.sort(key=lambda c: c[0][0])
    n_ok = 0
    with gzip.open(os.path.join(OUT, out_name), "wt") as f:
        f.write("#genome_start\tseq_with_1bp_flanks\trows(start:end:motif,...) relative to seq; k1-6 r3 span9; "
                "source: reference benchmark/repeat_finder/chr22_repeats.bed\n")
        for cl in chosen:
            s0 = cl[0][0]
            e0 = max(e for _, e, _ in cl)
            buf = [None] * (e0 - s0)
            ok = True
            for s, e, m in cl:
                for i in range(s, e):
                    ch = m[(i - s) % len(m)]
                    if buf[i - s0] not in (None, ch):
                        ok = False
                    buf[i - s0] = ch
            if not ok or None in buf:
                continue
            body = "".join(buf)
            want = [[s - s0 + 1, e - s0 + 1, m] for s, e, m in cl]
            found = None
            for x in "ACGT":
                for y in "ACGT":
                    got = run_ref(x + body + y, st)
                    if got["status"] == "ok" and got["rows"] == want:
                        found = x + body + y
                        break
                if found:
                    break
            if not found:
                continue
            n_ok += 1
            f.write(f"{s0}\t{found}\t" + ",".join(f"{a}:{b}:{m}" for a, b, m in want) + "\n")
    print("chr22 clusters: total", len(clusters), "chosen", len(chosen), "written", n_ok)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    todo = a.only.split(",") if a.only else ["unit", "fuzz", "adv", "odd", "mr1", "iupac", "clusters", "synth"]
    if "unit" in todo:
        gen_ref_unit_tests()
    if "fuzz" in todo:
        gen_fuzz_small(600 if a.quick else 4000)
    if "adv" in todo:
        gen_adversarial()
    if "odd" in todo:
        gen_odd_intervals(300 if a.quick else 1500)
    if "mr1" in todo:
        gen_min_repeats_one(500 if a.quick else 3000)
    if "iupac" in todo:
        gen_iupac()
    if "clusters" in todo:
        gen_chr22_clusters(1500 if a.quick else 8000)
    if "clusters_all" in todo:      # (not part of the default set: ~10 minutes of the reference)
        gen_chr22_clusters(None, "chr22_clusters_all.tsv.gz")
    if "synth" in todo:
        gen_synth(a.quick)


if __name__ == "__main__":
    main()
