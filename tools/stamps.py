"""Diagnostic: per-wave phase times of the fused kernel from the PRF_STAMPS build (libprf_stamps.so).
Usage (GPU box): PRF_LIB=colab-repeat-finder_amd/libprf_stamps.so PRF_STAMPS_OUT=/tmp/st.bin python tools/stamps.py [length]"""
import os, sys
import numpy as np
sys.path.insert(0, 'colab-repeat-finder_amd'); sys.path.insert(0, '.')
import prf_native, synth
L = int(sys.argv[1]) if len(sys.argv) > 1 else 50_818_468
ctx = prf_native.Context(0)
if len(sys.argv) > 2 and sys.argv[2] == 'standin2':   # the default bench workload's recipe, generated on the device
    g = ctx.standin([L], [1], 50)
else:
    seq = synth.chr_standin(length=L, seed=22, n_head=10_510_000 if L > 2e7 else L // 10, n_tail=10_000).tobytes()
    g = ctx.load([seq], 50)
for _ in range(3):
    g.scan(1, 50, 3, 9, fetch=False)
_, st = g.scan(1, 50, 3, 9, fetch=False)
print('kernel ms', st.phase1_ms)
st_ms_global = st.phase1_ms
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

nf = d[:, 0, 11] & 0xFFFFFFFF; nr = d[:, 0, 11] >> 32
print('flags per tile: mean %.1f p10 %d p50 %d p90 %d p99 %d max %d; share <=64 %.3f <=128 %.3f <=192 %.3f <=256 %.3f' % (
    nf.mean(), *[int(np.percentile(nf, q)) for q in (10, 50, 90, 99, 100)], *[(nf <= c).mean() for c in (64, 128, 192, 256)]))
print('records per tile: mean %.1f p10 %d p50 %d p90 %d p99 %d max %d; share <=64 %.3f <=128 %.3f <=192 %.3f' % (
    nr.mean(), *[int(np.percentile(nr, q)) for q in (10, 50, 90, 99, 100)], *[(nr <= c).mean() for c in (64, 128, 192)]))
import collections
print('joint (ceil(flags/64), ceil(recs/64)) shares:', sorted(((k, round(v / len(nf), 3)) for k, v in collections.Counter(zip(((nf + 63) // 64).tolist(), ((nr + 63) // 64).tolist())).items()), key=lambda kv: -kv[1])[:12])
for w in range(4):
    print('wave', w, 'verify: flags part %d, records / boundary part %d' % (int(np.median(d[:, w, 14] - d[:, w, 4])), int(np.median(d[:, w, 5] - d[:, w, 14]))))
for w in range(4):
    print('wave', w, 'cycles inside the exact-task functions (sum over the wave\'s exact tasks): median', int(np.median(d[:, w, 15])))

# slot utilisation from the chip-wide 100 MHz counter (slots 12 / 13 = workgroup start / end)
rs = d[:, :, 12].min(axis=1); re = d[:, :, 13].max(axis=1)
span = re.max() - rs.min()
print('kernel span on the 100 MHz counter: %.1f us; workgroup time: p50 %.2f us; slot utilisation (256 CUs x 4): %.3f' % (
    span / 100.0, np.median(re - rs) / 100.0, (re - rs).sum() / (span * 1024.0)))
print('core cycles per 10 ns tick (p50 over workgroups): %.2f' % np.median((d[:, 0, 7] - d[:, 0, 0]) / np.maximum(1, d[:, 0, 13] - d[:, 0, 12])))
# how many workgroups are resident over time (per 5 us bucket)
edges = np.arange(rs.min(), re.max() + 500, 500)
res = [(int(((rs < e + 500) & (re > e)).sum())) for e in edges[:-1]]
print('resident workgroups per 5 us bucket:', res[:8], '...', res[len(res)//2 - 2: len(res)//2 + 2], '...', res[-6:])
