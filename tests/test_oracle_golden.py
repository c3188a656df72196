"""CPU: the oracle (oracle/prf_oracle.c) is pinned against every golden vector -- the reference's own
unit-test vectors, the reference's outputs on fuzz/adversarial/synthetic inputs, and the clusters mined
from the reference's chr22 golden BED."""
import glob
import json
import os

from conftest import GOLDEN
from helpers import expected, outcome
from oracle import prf_oracle


def test_reference_unit_vectors(golden_unit):
    assert len(golden_unit) >= 30
    for case in golden_unit:
        assert outcome(prf_oracle.detect_repeats, case["seq"], case["settings"]) == ("ok", case["rows"]), case["tag"]


def test_fuzz_small(golden_fuzz):
    statuses = set()
    for case in golden_fuzz:
        statuses.add(case["status"])
        assert outcome(prf_oracle.detect_repeats, case["seq"], case["settings"]) == expected(case), case
    assert {"ok", "IndexError", "AssertionError"} <= statuses


def test_adversarial(golden_adversarial):
    for case in golden_adversarial:
        assert outcome(prf_oracle.detect_repeats, case["seq"], case["settings"]) == expected(case), case["tag"]


def test_min_repeats_one(golden_min_repeats_one):
    """min_repeats_one.jsonl.gz: the reference run with min_repeats == 1 (interval mode, N at the ends, k > len, IUPAC)."""
    statuses = set()
    assert len(golden_min_repeats_one) >= 3000
    for case in golden_min_repeats_one:
        statuses.add(case["status"])
        assert outcome(prf_oracle.detect_repeats, case["seq"], case["settings"]) == expected(case), case
    assert {"ok", "IndexError", "AssertionError"} <= statuses


def test_odd_intervals(golden_odd_intervals):
    """odd_intervals.jsonl.gz: interval bounds reversed, outside the sequence, on N (reference perfect_repeat_finder.py:35-46)."""
    assert len(golden_odd_intervals) >= 1500
    for case in golden_odd_intervals:
        assert outcome(prf_oracle.detect_repeats, case["seq"], case["settings"]) == expected(case), case


def test_symbols_other_than_acgtn_are_ordinary_symbols(golden_iupac):
    """iupac.jsonl.gz: the reference run on sequences with IUPAC letters (R == R matches, only N never does)."""
    assert len(golden_iupac) >= 270 and sum(len(c.get("rows") or []) for c in golden_iupac) > 2500
    for case in golden_iupac:
        assert outcome(prf_oracle.detect_repeats, case["seq"], case["settings"]) == expected(case), case["tag"]


def test_chr22_clusters(golden_clusters):
    st = dict(min_motif_size=1, max_motif_size=6, min_repeats=3, min_span=9)
    assert len(golden_clusters) >= 8000
    for pos, seq, want in golden_clusters:
        assert outcome(prf_oracle.detect_repeats, seq, st) == ("ok", [list(w) for w in want]), pos


def test_synthetic_sequences():
    files = sorted(glob.glob(os.path.join(GOLDEN, "synth_*.json")))
    assert len(files) >= 5
    for path in files:
        with open(path) as f:
            g = json.load(f)
        seq = prf_oracle.synth(g["n"], g["seed"]).decode()
        assert seq[:64] == g["head"]
        assert outcome(prf_oracle.detect_repeats, seq, g["settings"]) == ("ok", g["rows"]), path
