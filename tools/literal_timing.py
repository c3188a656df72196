#!/usr/bin/env python3
"""Kernel time of the literal lane (csrc/scan_literal.hip) on random sequence: python tools/literal_timing.py [Mbp] [kmax]."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "colab-repeat-finder_amd"))
import numpy as np  # noqa: E402
import prf_native  # noqa: E402

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 50.0
kmax = int(sys.argv[2]) if len(sys.argv) > 2 else 50
n = int(mbp * 1e6)
rng = np.random.default_rng(22)
seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n, dtype=np.uint8)].tobytes()
ctx = prf_native.Context(0)
for r, span in ((1, 9), (3, 9)):
    for _ in range(2):
        rows, st = ctx.scan_literal(seq, 1, kmax, r, span)
    gb = n * kmax * 2 / 1e9          # two byte reads per (position, motif size)
    print(f"{mbp:g} Mbp, k 1-{kmax}, min_repeats {r}, min_span {span}: {st.scan_ms:.3f} ms kernel, {len(rows)} rows, "
          f"{n / st.scan_ms / 1e6:.1f} Gbp/s, {gb / st.scan_ms * 1e3:.0f} GB/s of byte reads (L2-served)")
    if r == 3:
        fused, fst = ctx.scan([seq], 1, kmax, r, span)
        assert len(fused) == len(rows) and all((fused[x] == rows[x]).all() for x in ("start", "end", "k"))
        print(f"  rows equal the fused kernel's ({fst.scan_ms:.3f} ms there)")
ctx.close()
