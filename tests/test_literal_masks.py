"""The bit arithmetic of the literal lane's 64-positions kernel (csrc/scan_literal.hip::prf_lit_events64_kernel), restated
with Python integers and checked against the events' definition (utils/perfect_repeat_tracker.py:53, :86-91 of the
reference) by brute force -- no GPU.  Two claims carry the kernel's exactness:
  (a) the bit-parallel candidate set contains every event that passes both filters (and every event whose run or extension
      leaves the 128 positions a thread sees);
  (b) for a candidate, run length and extension read off the masks with clz / ctz are the tracker's.
The GPU tests check the kernel itself against the reference's cases and the oracle."""
import random

M128 = (1 << 128) - 1


def masks(seq, k, p0):
    """128-bit masks of the positions p0 - 64 .. p0 + 63 (bit n = position p0 - 64 + n): mm = failed comparison under the
    tracker's rule (letters differ, or the letter is N), pm = equal letters."""
    mm = pm = 0
    for n in range(128):
        q = p0 - 64 + n
        a, b = seq[q], seq[q + k]
        if a != b or a == "N":
            mm |= 1 << n
        if a == b:
            pm |= 1 << n
    return mm, pm


def shr1(v, s):                      # right shift filling with ones: positions past the window count as matching
    return (v >> s) | (M128 & ~(M128 >> s))


def window_down(nm, r):              # bit n: the r positions directly below n all set in nm (shift-and-AND doubling)
    v = (nm << 1) & M128
    have = 1
    while 2 * have <= r:
        v &= (v << have) & M128
        have *= 2
    if have < r:
        v &= (v << (r - have)) & M128
    return v


def window_up(pm, b):                # bit n: the b positions n .. n + b - 1 all set in pm
    v = pm
    have = 1
    while 2 * have <= b:
        v &= shr1(v, have)
        have *= 2
    if have < b:
        v &= shr1(v, b - have)
    return v


def candidate_bits(mm, pm, k, T):
    """The kernel's candidate set for the 64 own positions (bits 64 .. 127), as its four-window test computes it."""
    if T <= 1:
        return mm >> 64
    av = min(T // 2 + 1, 60)
    bv = max(1, min(T - av + 1, 60))
    tr = min(T, 61) - 1
    te = min(T - 1, 60)
    nm = ~mm & M128
    d1 = (nm << 1) & M128
    dn = window_down(nm, av - 1)
    dt = dn & ((dn << (tr - (av - 1))) & M128) if tr > av - 1 else dn
    ub = window_up(pm, bv)
    ut = ub & shr1(ub, te - bv) if te > bv else ub
    al = lambda v: (v << (k - 1)) & M128      # bit n <- bit n + 1 - k
    cand = mm & ((dn & al(pm)) | dt | (al(ub) & d1) | al(ut))
    return cand >> 64


def brute(seq, k, i, lo):
    """(run, ext) of the event at position i as the tracker computes them, None where the walk leaves [lo, lo + 128)."""
    run, q = 1, i - 1
    while True:
        if q < lo:
            run = None
            break
        if seq[q] == seq[q + k] and seq[q] != "N":
            run, q = run + 1, q - 1
        else:
            break
    ext, j = 0, i + 1 - k
    while True:
        if j >= lo + 128:
            ext = None
            break
        if seq[j] == seq[j + k]:
            ext, j = ext + 1, j + 1
        else:
            break
    return run, ext


def exact_from_masks(mm, pm, k, i):
    """run and extension of the event at own position i (0 .. 63), read off the masks as the kernel does."""
    n = 64 + i
    below = mm & ((1 << n) - 1)
    run = n - (below.bit_length() - 1) if below else None
    m = n + 1 - k
    v = ~(pm >> m) & ((1 << (128 - m)) - 1)
    ext = (v & -v).bit_length() - 1 if v else None
    return run, ext


def make_seq(rng, n):
    s = [rng.choice("ACGT") for _ in range(n)]
    for _ in range(n // 40):                     # planted repeats of random period and span, some N
        p, k, span = rng.randrange(n), rng.randint(1, 40), rng.randint(2, 150)
        for t in range(p + k, min(n, p + span)):
            s[t] = s[t - k]
    for _ in range(n // 300):
        p = rng.randrange(n)
        for t in range(p, min(n, p + rng.randint(1, 5))):
            s[t] = "N"
    return "".join(s)


def test_candidate_set_and_exact_walks():
    rng = random.Random(20251005)
    checked = passing = unknown = 0
    for _ in range(12):
        seq = make_seq(rng, 1600)
        for k in (1, 2, 3, 5, 8, 13, 21, 34, 50, 63):
            for T in sorted({1, 2, 3, 9, max(9, k), 2 * k, 3 * k, 64, 121, 200}):
                R1 = T - k + 1
                for p0 in range(64, len(seq) - 64 - k - 64, 64):
                    mm, pm = masks(seq, k, p0)
                    cand = candidate_bits(mm, pm, k, T)
                    for i in range(64):
                        if not (mm >> (64 + i)) & 1:
                            continue
                        run, ext = brute(seq, k, p0 + i, p0 - 64)
                        assert (run, ext) == exact_from_masks(mm, pm, k, i)
                        checked += 1
                        if run is None or ext is None:
                            unknown += 1
                            must = run is None or run >= R1          # a known run below R1 fails :86 whatever the extension
                        else:
                            must = run >= R1 and run + ext >= T
                            passing += must
                        if must:
                            assert (cand >> i) & 1, (k, T, p0, i, run, ext)
    assert checked > 500_000 and passing > 2_000 and unknown > 100
