"""Host time of the packed-row hand-off (prf_last_hits_packed_to_device) behind a scan of the default workload's contig 0.
usage (GPU box): python tools/pack_timing.py"""
import sys, time
sys.path.insert(0, 'colab-repeat-finder_amd'); sys.path.insert(0, '.')
import torch, prf_native
ctx = prf_native.Context(0)
g = ctx.standin([248_956_422], [1000], 50)
rows, st = g.scan(1, 50, 3, 9, fetch=False)
n = int(st.n_hits)
words = torch.zeros(n + 1 + 3 * 1024, dtype=torch.int64, device="cuda")
for _ in range(5):
    ctx.last_hits_packed_to_device(g, words.data_ptr(), n, 1024)
t0 = time.perf_counter()
for _ in range(200):
    ctx.last_hits_packed_to_device(g, words.data_ptr(), n, 1024)
dt = (time.perf_counter() - t0) / 200
print(f"{n} rows: {dt * 1e6:.1f} us per packed hand-off")
