"""CPU: the kept API surface (PerfectRepeatTracker, consists_of_perfect_repeats) behaves like the reference's:
driving one tracker per motif size by hand reproduces the golden rows."""
from helpers import settings_ns
from utils.perfect_repeat_tracker import PerfectRepeatTracker, consists_of_perfect_repeats


def drive(seq, fs):
    seq = seq.upper()
    out = {}
    trackers = [PerfectRepeatTracker(k, fs.min_repeats, fs.min_span, seq, out)
                for k in range(fs.min_motif_size, fs.max_motif_size + 1)]
    for _ in range(len(seq)):
        for t in trackers:
            t.advance()
    for t in trackers:
        assert not t.advance()
        t.done()
    return [[s, e, m] for (s, e), m in sorted(out.items())]


def test_consists_of_perfect_repeats():
    assert consists_of_perfect_repeats("CAGCAGCAG") == "CAG"
    assert consists_of_perfect_repeats("AAAA") == "A"
    assert consists_of_perfect_repeats("ACAC") == "AC"
    assert consists_of_perfect_repeats("ACGTACGTACGTACGT") == "ACGT"
    for w in ("A", "", "AC", "ACA", "CAGCA", "ACGTACGA"):
        assert consists_of_perfect_repeats(w) is None


def test_tracker_reproduces_golden_rows(golden_unit, golden_fuzz):
    n = 0
    for case in golden_unit + golden_fuzz[:600]:
        st = case["settings"]
        if "interval_end" in st or st["min_repeats"] < 2 or case.get("status", "ok") != "ok":
            continue
        n += 1
        assert drive(case["seq"], settings_ns(st)) == case["rows"], case
    assert n > 300


def test_tracker_properties():
    out = {}
    t = PerfectRepeatTracker(2, 3, 6, "ACACACACGT", out)
    assert t.current_position == 0 and not t.is_in_middle_of_repeat()
    for _ in range(4):
        t.advance()
    assert t.is_in_middle_of_repeat() and t.current_position == 4
