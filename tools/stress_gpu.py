"""Randomised differential test of the GPU path against the oracle: random multi-contig inputs (N blocks, planted
repeats of random period, runs across tile and stream edges, short contigs) x random parameter sets.
usage (GPU box): python tools/stress_gpu.py [seconds] [seed]"""
import random, sys, time
import numpy as np
sys.path.insert(0, 'colab-repeat-finder_amd'); sys.path.insert(0, '.')
import prf_native, synth
from oracle import prf_oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = prf_native.Context(0)

def make_contig(n):
    seq = bytearray(synth.synth_bases(n, rng.randrange(1 << 30)).tobytes())
    for _ in range(rng.randint(0, 4)):                       # N blocks of assorted sizes, some at the ends
        a = rng.choice([0, rng.randrange(max(1, n)), max(0, n - rng.randint(1, 3000))])
        b = min(n, a + rng.choice([1, 2, 7, 40, 500, 70000, 140000]))
        seq[a:b] = b'N' * (b - a)
    for _ in range(int(n / 1e6 * rng.choice([200, 2000, 8000])) + 3):   # planted repeats
        k = rng.choice([1, 1, 2, 2, 3, 3, 4, 5, 6, 7, 8, 11, 12, 16, 23, 31, 32, 33, 47, 64, 65, 100, 127])
        copies = rng.choice([2, 3, 3, 4, 5, 8, 20, 60]) + rng.random()
        span = min(int(k * copies), 4000)
        p = rng.choice([rng.randrange(max(1, n)), (rng.randrange(1 + n // 65536)) * 65536 - rng.randint(0, 80),
                        (rng.randrange(1 + n // 2048)) * 2048 - rng.randint(0, 40)])
        p = max(0, min(n - 1, p))
        motif = bytes(rng.choice(b'ACGT') for _ in range(k))
        body = (motif * (span // k + 2))[:span]
        seq[p:p + len(body)] = body[:max(0, n - p)]
    return bytes(seq[:n])

t_end = time.time() + budget
t_report = time.time() + 60
cases = bad = 0
while time.time() < t_end:
    if time.time() > t_report:   # a line a minute: a silent job looks hung to the GPU runner
        print(f'... {cases} scans so far, {bad} mismatches', flush=True)
        t_report = time.time() + 60
    contigs = [make_contig(rng.choice([0, 1, 50, 3000, 70000, 200000, 400000])) for _ in range(rng.randint(1, 4))]
    kmax_all = rng.choice([6, 20, 50, 100, 150])
    g = ctx.load(contigs, kmax_all)
    for _ in range(3):
        kmin = rng.randint(1, 6)
        kmax = min(kmax_all, kmin + rng.choice([0, 3, 10, 40, 140]))
        r = rng.choice([2, 2, 3, 3, 4, 6])
        span = rng.choice([1, 5, 9, 9, 12, 16, 30, 100])
        rows, st = g.scan(kmin, kmax, r, span)
        got = [(int(x['contig']), int(x['start']), int(x['end']), int(x['k'])) for x in rows]
        want = []
        for ci, s in enumerate(contigs):
            want += [(ci, a, b, k) for a, b, _ml, k in prf_oracle.detect_rows(s, kmin, kmax, r, span)]
        cases += 1
        if got != want or st.path != 1:
            bad += 1
            print('MISMATCH', dict(kmin=kmin, kmax=kmax, r=r, span=span, lens=[len(c) for c in contigs], path=st.path),
                  'got', len(got), 'want', len(want), 'first diff', next((x for x in zip(got, want) if x[0] != x[1]), None))
            if bad > 5: sys.exit(1)
    g.free()
    # the literal lane (min_repeats == 1) on the same contigs through prf_scan: N-trimming, slice clamp, wrap-around, IndexError
    small = [c[:rng.choice([0, 1, 7, 300, 20000, 90000])] for c in contigs]
    kmin = rng.randint(1, 6)
    kmax = kmin + rng.choice([0, 3, 10, 40, 62])   # (62: across the literal lane's 63 / 64 split)
    span = rng.choice([1, 5, 9, 12, 30, 70, 150])
    try:
        want = [(ci, a, b, ml) for ci, s in enumerate(small) for a, b, ml, _k in prf_oracle.detect_rows(s, kmin, kmax, 1, span)]
    except IndexError:
        want = 'IndexError'
    try:
        if rng.random() < 0.5:
            rows, st = ctx.scan(small, kmin, kmax, 1, span)
        else:                                    # the same on a resident genome: bytes rebuilt on the device from the planes
            gs = ctx.load(small, kmax)
            try:
                rows, st = gs.scan(kmin, kmax, 1, span)
            finally:
                gs.free()
        got = [(int(x['contig']), int(x['start']), int(x['end']), int(x['k'])) for x in rows]
    except prf_native.PrfError as exc:
        got = 'IndexError' if exc.code == prf_native.PRF_EINDEX else repr(exc)
    cases += 1
    if got != want:
        bad += 1
        print('MISMATCH (literal lane)', dict(kmin=kmin, kmax=kmax, span=span, lens=[len(c) for c in small]),
              'got', got if isinstance(got, str) else len(got), 'want', want if isinstance(want, str) else len(want))
        if bad > 5: sys.exit(1)
print(f'stress: {cases} scans, {bad} mismatches')
sys.exit(1 if bad else 0)
