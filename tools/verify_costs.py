"""Diagnostic (PRF_STAMPS build): per-record cost of the cooperative verify pass.
Usage (GPU box): PRF_LIB=colab-repeat-finder_amd/libprf_stamps.so PRF_STAMPS_OUT=/tmp/st.bin python tools/verify_costs.py"""
import os, sys
import numpy as np
sys.path.insert(0, 'colab-repeat-finder_amd'); sys.path.insert(0, '.')
import prf_native, synth
seq = synth.chr_standin().tobytes()
ctx = prf_native.Context(0)
g = ctx.load([seq], 50)
for _ in range(2):
    g.scan(1, 50, 3, 9, fetch=False)
v = np.fromfile(os.environ['PRF_STAMPS_OUT'] + '.verify', dtype=np.uint64).reshape(-1, 1024, 2)
cyc = v[:, :, 0].astype(np.int64)
info = v[:, :, 1]
used = cyc > 0
kind = (info & np.uint64(0xFF)).astype(np.int64)
k = ((info >> np.uint64(8)) & np.uint64(0xFFFF)).astype(np.int64)
slow = ((info >> np.uint64(24)) & np.uint64(0xFF)).astype(np.int64)
walk = ((info >> np.uint64(32)) & np.uint64(0xFFFFFF)).astype(np.int64)
nb = ((info >> np.uint64(56)) & np.uint64(0xFF)).astype(np.int64)
pop = np.array([bin(i).count('1') for i in range(256)])[nb]
print('records per tile: mean %.1f max %d' % (used.sum(1).mean(), used.sum(1).max()))
print('cycles per record: p50 %d p90 %d p99 %d max %d' % tuple(np.percentile(cyc[used], q) for q in (50, 90, 99, 100)))
for kd in range(4):
    m = used & (kind == kd)
    if m.any():
        print('kind %d: n/tile %.1f cycles p50 %d p90 %d p99 %d; mask bits mean %.2f; slow-path share %.3f' % (
            kd, m.sum() / len(cyc), *np.percentile(cyc[m], (50, 90, 99)), pop[m].mean(), (slow[m] > 0).mean()))
m = used & (slow > 0)
print('slow-path records: n/tile %.2f cycles p50 %d p90 %d' % (m.sum() / len(cyc), *np.percentile(cyc[m], (50, 90))))
# cost of a wave iteration = max over its 64 lanes
wmax = cyc.reshape(len(cyc), 16, 64).max(2)
for w in range(5):
    print('wave-iteration %d (idx %d..%d): max-lane cycles p50 %d p90 %d' % (w, 64 * w, 64 * w + 63, *np.percentile(wmax[:, w], (50, 90))))
# what is the max lane?
arg = cyc.reshape(len(cyc), 16, 64)[:, 0, :].argmax(1)
sel = (np.arange(len(cyc)), arg)
print('slowest lane of iteration 0: kind histogram', np.bincount(kind[:, :64][sel], minlength=4), 'slow share %.2f' % (slow[:, :64][sel] > 0).mean(),
      'walk p50 %d' % np.percentile(walk[:, :64][sel], 50), 'k p50 %d' % np.percentile(k[:, :64][sel], 50), 'bits mean %.2f' % pop[:, :64][sel].mean())
for kk in range(1, 9):
    m = used & (kind == 0) & (k == kk)
    if m.any():
        print('START k=%d: n/tile %.1f cycles p50 %d p90 %d' % (kk, m.sum() / len(cyc), *np.percentile(cyc[m], (50, 90))))
