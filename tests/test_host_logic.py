"""CPU: the host-side logic of the drop-in detect_repeats() -- argument validation, interval mode
(early-break rule, trimming, assertion), coordinate/motif materialisation -- checked against the golden
fixtures with the GPU scan replaced by closed-form rows computed in this file (pure Python).
The closed form itself is checked against the oracle here as well."""
import argparse

import pytest

import perfect_repeat_finder as prf
from helpers import expected, outcome
from utils.plot_utils import shift_string_by


def closed_form_rows(seq, fs, context=None):
    """SURVEY 3.4 in ~15 lines; stands in for libprf in these CPU tests."""
    if fs.min_repeats < 2:
        raise NotImplementedError
    s = seq.upper()
    n = len(s)
    rows = []
    for k in range(fs.min_motif_size, fs.max_motif_size + 1):
        need = max((fs.min_repeats - 1) * k, fs.min_span - k)
        j = 0
        while j < n - k:
            if s[j] == s[j + k] and s[j] != "N":
                a = j
                while j < n - k and s[j] == s[j + k] and s[j] != "N":
                    j += 1
                motif = s[a:a + k]
                primitive = not any(k % d == 0 and motif == motif[:d] * (k // d) for d in range(1, k))
                if j - a >= need and primitive:
                    rows.append((a, j + k, k))
            else:
                j += 1
    return sorted(rows)


@pytest.fixture()
def cpu_rows(monkeypatch):
    monkeypatch.setattr(prf, "_gpu_rows", closed_form_rows)


def test_shift_string_by():
    # reference perfect_repeat_finder_tests.py:11-18
    assert shift_string_by("A", 1) == "A"
    for shift, want in enumerate(["TTTCG", "GTTTC", "CGTTT", "TCGTT", "TTCGT", "TTTCG"]):
        assert shift_string_by("TTTCG", shift) == want
    assert shift_string_by("AAAT", -1) == "AATA"


def test_validation_messages():
    ok = dict(min_motif_size=1, max_motif_size=5, min_repeats=3, min_span=9)
    for key, val, msg in (("min_motif_size", 0, "min_motif_size is set to 0. It must be at least 1."),
                          ("max_motif_size", 0, "max_motif_size is set to 0. It must be at least min_motif_size."),
                          ("min_repeats", 0, "min_repeats is set to 0. It must be at least 1."),
                          ("min_span", -2, "min_span is set to -2. It must be at least 1.")):
        bad = dict(ok)
        bad[key] = val
        with pytest.raises(ValueError) as info:
            prf.detect_repeats("ACGT", argparse.Namespace(**bad))
        assert str(info.value) == msg
    with pytest.raises(AttributeError):
        prf.detect_repeats("ACGT", argparse.Namespace(min_motif_size=1, max_motif_size=5, min_repeats=3))


def test_unit_vectors_through_host_logic(cpu_rows, golden_unit):
    for case in golden_unit:
        assert outcome(prf.detect_repeats, case["seq"], case["settings"]) == ("ok", case["rows"]), case["tag"]


def test_fuzz_through_host_logic(cpu_rows, golden_fuzz):
    n_interval = 0
    for case in golden_fuzz:
        if case["settings"]["min_repeats"] < 2:
            continue
        n_interval += "interval_end" in case["settings"]
        assert outcome(prf.detect_repeats, case["seq"], case["settings"]) == expected(case), case
    assert n_interval > 500


def test_min_repeats_one_is_refused_loudly(cpu_rows):
    with pytest.raises(NotImplementedError):
        prf.detect_repeats("ACACACAC", argparse.Namespace(min_motif_size=1, max_motif_size=5, min_repeats=1, min_span=3))


def test_find_repeats_alias():
    assert prf.find_repeats is prf.detect_repeats
