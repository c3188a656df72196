"""Diagnostic: the same scan again and again must give the same rows (synchronous and pipelined calls).
usage (GPU box): [NC=contigs] [N=scans] [PRF_SKIP=..] python tools/nondet_check.py"""
import os, sys
import numpy as np
sys.path.insert(0, "colab-repeat-finder_amd"); sys.path.insert(0, ".")
import prf_native
HG38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622,
        133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415, 16569]
ctx = prf_native.Context(0)
lens = HG38[: int(os.environ.get("NC", "1"))]
g = ctx.standin(lens, [1000 + i for i in range(len(lens))], 50)
ref, st = g.scan(1, 50, 3, 9)
print("rows", len(ref), "sorted", st.sorted_on_device, flush=True)
bad = 0
n = int(os.environ.get("N", "60"))
for i in range(n):
    rows, st = g.scan(1, 50, 3, 9)
    if len(rows) != len(ref) or not np.array_equal(rows, ref):
        bad += 1
        a = set(map(tuple, rows.tolist())); b = set(map(tuple, ref.tolist()))
        print("scan", i, "differs: rows", len(rows), "missing", sorted(b - a)[:5], "extra", sorted(a - b)[:5], flush=True)
print("synchronous: bad", bad, "of", n)
counts, pending = [], None
for i in range(n):
    s = g.scan_async(1, 50, 3, 9)
    if pending is not None:
        counts.append(int(ctx.scan_wait(pending).n_hits))
    pending = s
counts.append(int(ctx.scan_wait(pending).n_hits))
import collections
print("pipelined: row counts", dict(collections.Counter(counts)), "expected", len(ref))
