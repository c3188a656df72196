"""CPU: the host-side logic of the drop-in detect_repeats() -- argument validation, interval mode
(early-break rule, trimming, assertion), coordinate/motif materialisation -- checked against the golden
fixtures with the GPU scan replaced by closed-form rows computed in this file (pure Python).
The closed form itself is checked against the oracle here as well."""
import argparse

import pytest

import perfect_repeat_finder as prf
from helpers import expected, outcome
from utils.plot_utils import shift_string_by


def event_rows(seq, fs, stop=None):
    """The decomposition csrc/scan_literal.hip uses, in pure Python: every flush call of a tracker (a failed comparison at
    i < min(stop, len - k), and done() at min(stop, len - k)) is evaluated on its own as reference
    utils/perfect_repeat_tracker.py:81-101 reads; per (start, end) the shortest motif stays.  Stands in for the literal lane
    in these CPU tests, so that the host logic around it (N-trimming, cutoff, assertion order) is checked against the
    reference's outputs without a GPU; the kernel itself is checked on the GPU against the same fixtures."""
    s = seq.upper()
    n = len(s)
    stop = n if stop is None else min(stop, n)
    best = {}
    for k in range(fs.min_motif_size, fs.max_motif_size + 1):
        nk = max(n - k, 0)
        pos_f = min(stop, nk)

        def match(j):
            return s[j] == s[j + k] and s[j] != "N"
        for i0 in range(pos_f + 1):
            if i0 < pos_f and match(i0):
                continue
            start = i0
            while start > 0 and match(start - 1):
                start -= 1
            run = i0 - start + 1
            motif = s[start:start + k]
            if "N" in motif:
                continue
            i = i0
            if run + k - 1 >= fs.min_span and run + k - 1 >= fs.min_repeats * k:
                while i < n - 1 and s[i + 1] == s[i + 1 - k]:      # IndexError as in the reference
                    run += 1
                    i += 1
            if run < fs.min_span or run < fs.min_repeats * k:
                continue
            m = len(motif)
            if any(m % d == 0 and motif == motif[:d] * (m // d) for d in range(1, m // 2 + 1)):
                continue
            key = (start, i + 1)
            best[key] = min(best.get(key, m), m)
    return sorted((a, b, m) for (a, b), m in best.items())


def closed_form_rows(seq, fs, context=None, stop=None):
    """SURVEY 3.4 in ~15 lines; stands in for libprf in these CPU tests."""
    if fs.min_repeats < 2:
        return event_rows(seq, fs, stop)
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


def cpu_scan_whole_fasta(fasta, bed_path, fs, report):
    """CPU stand-in for the one-scan whole-FASTA path (which needs a GPU): per contig, closed form, Python writer."""
    plain = argparse.Namespace(min_motif_size=fs.min_motif_size, max_motif_size=fs.max_motif_size,
                               min_repeats=fs.min_repeats, min_span=fs.min_span)
    with open(bed_path, "wt") as bed:
        for entry in fasta:
            rows = prf.detect_repeats(entry.seq, plain)
            bed.writelines(f"{entry.name}\t{s}\t{e}\t{m}\n" for s, e, m in rows)
            report(entry, len(rows))


@pytest.fixture()
def cpu_rows(monkeypatch):
    monkeypatch.setattr(prf, "_gpu_rows", closed_form_rows)
    monkeypatch.setattr(prf, "_scan_whole_fasta", cpu_scan_whole_fasta)


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


def test_min_repeats_one_through_host_logic(cpu_rows, golden_min_repeats_one, golden_fuzz):
    """min_repeats == 1: N-trimming, the cutoff handed to the literal lane, assertion-before-IndexError order."""
    statuses = set()
    for case in golden_min_repeats_one + [c for c in golden_fuzz if c["settings"]["min_repeats"] < 2]:
        statuses.add(case["status"])
        assert outcome(prf.detect_repeats, case["seq"], case["settings"]) == expected(case), case
    assert {"ok", "IndexError", "AssertionError"} <= statuses


def test_odd_intervals_through_host_logic(cpu_rows, golden_odd_intervals):
    """Interval bounds reversed / outside the sequence / on N: trimming fast path vs the reference's loops, cutoff, assertion."""
    statuses = set()
    for case in golden_odd_intervals:
        statuses.add(case["status"])
        assert outcome(prf.detect_repeats, case["seq"], case["settings"]) == expected(case), case
    assert {"ok", "IndexError"} <= statuses


def test_event_decomposition_equals_closed_form_for_two_or_more_repeats(golden_fuzz):
    """For min_repeats >= 2 and no early stop the literal lane's event model and the closed form give the same rows."""
    n = 0
    for case in golden_fuzz[:1500]:
        st = case["settings"]
        if st["min_repeats"] < 2 or case["status"] != "ok":
            continue
        fs = argparse.Namespace(**st)
        assert event_rows(case["seq"], fs) == closed_form_rows(case["seq"], fs), case
        n += 1
    assert n > 800


def test_find_repeats_alias():
    assert prf.find_repeats is prf.detect_repeats


# ---- command line (reference perfect_repeat_finder.py:83-183) ----

def _write_fasta(path, gz=False):
    import gzip
    text = ">chrA first contig\n" + "ACGT" * 5 + "\n" + "CAG" * 8 + "\nTTGA\n" + ">chrB\n" + "gatc" + "a" * 12 + "CT\n"
    (gzip.open(path, "wt") if gz else open(path, "wt")).write(text)
    return {"chrA": "ACGT" * 5 + "CAG" * 8 + "TTGA", "chrB": "gatc" + "a" * 12 + "CT"}


def test_cli_literal_sequence_writes_tsv(cpu_rows, tmp_path, monkeypatch, capsys):
    monkeypatch.chdir(tmp_path)
    prf.main(["-min", "1", "-max", "6", "--min-repeats", "3", "--min-span", "9", "-o", "out", "ACGCAGCAGCAGCAGCAGTT"])
    assert (tmp_path / "out.tsv").read_text() == "start_0based\tend\tmotif\n2\t18\tGCA\n"
    assert "Found 1 repeats" in capsys.readouterr().out


def test_cli_fasta_scans_every_contig_and_interval(cpu_rows, tmp_path, monkeypatch, capsys):
    monkeypatch.chdir(tmp_path)
    for gz in (False, True):
        fa = tmp_path / ("toy.fa.gz" if gz else "toy.fa")
        seqs = _write_fasta(str(fa), gz)
        prf.main([str(fa)])
        want = ""
        for name, seq in seqs.items():
            fs = argparse.Namespace(min_motif_size=1, max_motif_size=50, min_repeats=3, min_span=9)
            for s, e, k in closed_form_rows(seq, fs):
                want += f"{name}\t{s}\t{e}\t{seq.upper()[s:s + k]}\n"
        assert (tmp_path / "toy.bed").read_text() == want
        assert want.count("\n") == 3
    out = capsys.readouterr().out
    assert "Processing chrA (48 bp)" in out and "Wrote results to toy.bed" in out
    prf.main(["-i", "chrA:18-40", "-o", "iv", str(tmp_path / "toy.fa")])
    from oracle import prf_oracle
    fs = argparse.Namespace(min_motif_size=1, max_motif_size=50, min_repeats=3, min_span=9, interval_start_0based=18, interval_end=40)
    want = "".join(f"chrA\t{s}\t{e}\t{m}\n" for s, e, m in prf_oracle.detect_repeats(seqs["chrA"], fs))
    assert want and (tmp_path / "iv.bed").read_text() == want
    with pytest.raises(SystemExit):
        prf.main(["-i", "chrZ:1-5", str(tmp_path / "toy.fa")])
