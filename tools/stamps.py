"""Diagnostic: per-wave phase times of the fused kernel from the PRF_STAMPS build (libprf_stamps.so).
Usage (GPU box): PRF_LIB=colab-repeat-finder_amd/libprf_stamps.so PRF_STAMPS_OUT=/tmp/st.bin python tools/stamps.py [length]"""
import os, sys
import numpy as np
sys.path.insert(0, 'colab-repeat-finder_amd'); sys.path.insert(0, '.')
import prf_native, synth
L = int(sys.argv[1]) if len(sys.argv) > 1 else 50_818_468
seq = synth.chr_standin(length=L, seed=22, n_head=10_510_000 if L > 2e7 else L // 10, n_tail=10_000).tobytes()
ctx = prf_native.Context(0)
g = ctx.load([seq], 50)
for _ in range(3):
    g.scan(1, 50, 3, 9, fetch=False)
_, st = g.scan(1, 50, 3, 9, fetch=False)
print('kernel ms', st.phase1_ms)
d = np.fromfile(os.environ['PRF_STAMPS_OUT'], dtype=np.uint64).reshape(-1, 4, 16).astype(np.int64)
t0 = d[:, :, 0].min()
names = ['stage', 'bar1', 'scan', 'bar2', 'verify', 'bar3', 'rows']
for w in range(4):
    seg = [np.median(d[:, w, i + 1] - d[:, w, i]) for i in range(7)]
    print('wave', w, ' '.join(f'{n}={int(v)}' for n, v in zip(names, seg)), 'total', int(np.median(d[:, w, 7] - d[:, w, 0])))
print('WG start spread (cycles): min', 0, 'median', int(np.median(d[:, 0, 0] - t0)), 'max', int((d[:, 0, 0] - t0).max()))
print('WG end   (cycles): median', int(np.median(d[:, :, 7].max(axis=1) - t0)), 'max', int((d[:, :, 7].max(axis=1) - t0).max()))
tot = (d[:, :, 7].max(axis=1) - d[:, :, 0].min(axis=1))
print('WG total cycles: p50', int(np.percentile(tot, 50)), 'p90', int(np.percentile(tot, 90)), 'p99', int(np.percentile(tot, 99)), 'max', int(tot.max()), 'argmax', int(tot.argmax()), 'of', len(tot))
print('last 4 WGs (mixed tiles are last):', tot[-4:])
ends = np.sort(d[:, :, 7].max(axis=1) - t0)
print('WG end percentiles (cycles from first start): p10 %d p50 %d p90 %d p99 %d max %d' % tuple(int(np.percentile(ends, q)) for q in (10, 50, 90, 99, 100)))
order = np.argsort(tot)[-6:]
for i in order:
    print('slow WG', int(i), 'total', int(tot[i]), 'phases w0', [int(d[i, 0, j + 1] - d[i, 0, j]) for j in range(6)])

# per-task durations (cycles, median over workgroups): slot 8+i = start of the wave's i-th task, slot 3 = end of scan
for w in range(4):
    starts = [d[:, w, 8 + i] for i in range(8)]
    out = []
    for i in range(8):
        if np.median(starts[i]) == 0: break
        nxt = starts[i + 1] if i + 1 < 8 and np.median(starts[i + 1]) != 0 else d[:, w, 3]
        out.append(int(np.median(nxt - starts[i])))
    print('wave', w, 'task cycles', out)

for w in range(4):
    print('wave', w, 'verify split: collect flags=%d barrier=%d flags body=%d group+boundary=%d' % tuple(int(np.median(x)) for x in (
        d[:, w, 12] - d[:, w, 4], d[:, w, 13] - d[:, w, 12], d[:, w, 14] - d[:, w, 13], d[:, w, 5] - d[:, w, 14])))
