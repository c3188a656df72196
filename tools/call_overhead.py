"""Diagnostic: host-side cost of one scan call (Python wrapper + ctypes + library) on a genome with no tile to launch,
and the split of a real step.  Usage (GPU box): python tools/call_overhead.py"""
import sys, time, ctypes
import torch  # before libprf (load order)
sys.path.insert(0, 'colab-repeat-finder_amd'); sys.path.insert(0, '.')
import prf_native, synth
ctx = prf_native.Context(0)
g0 = ctx.load([b"N" * 1000], 50)          # nothing but N: the scan returns before any launch
N = 20000
t0 = time.perf_counter()
for _ in range(N):
    g0.scan(1, 50, 3, 9, flags=prf_native.SCAN_DEFER_TIMING, fetch=False)
dt = (time.perf_counter() - t0) / N
print('empty scan through Genome.scan: %.2f us' % (dt * 1e6))
lib = ctx.lib
stats = prf_native.ScanStats()
f = lib.prf_scan_genome
args = (ctx._h, g0._h, 1, 50, 3, 9, prf_native.SCAN_DEFER_TIMING | prf_native.SCAN_NO_FETCH, None, ctypes.byref(stats))
t0 = time.perf_counter()
for _ in range(N):
    f(*args)
dt = (time.perf_counter() - t0) / N
print('empty scan, bare ctypes call with prebuilt arguments: %.2f us' % (dt * 1e6))
seq = synth.chr_standin().tobytes()
g = ctx.load([seq], 50)
for _ in range(5):
    g.scan(1, 50, 3, 9, flags=prf_native.SCAN_DEFER_TIMING, fetch=False)
N = 2000
t0 = time.perf_counter()
for _ in range(N):
    g.scan(1, 50, 3, 9, flags=prf_native.SCAN_DEFER_TIMING, fetch=False)
dt = (time.perf_counter() - t0) / N
print('chr22 scan through Genome.scan: %.2f us' % (dt * 1e6))
args = (ctx._h, g._h, 1, 50, 3, 9, prf_native.SCAN_DEFER_TIMING | prf_native.SCAN_NO_FETCH, None, ctypes.byref(stats))
t0 = time.perf_counter()
for _ in range(N):
    f(*args)
dt = (time.perf_counter() - t0) / N
print('chr22 scan, bare ctypes call: %.2f us' % (dt * 1e6))
ms = ctx.scan_timings(stats.seq - 99, 100)
print('kernel (events) mean %.2f us' % (1e3 * sum(ms) / len(ms)))
import torch
buf = torch.zeros((100001, 3), dtype=torch.int64, device="cuda")
ctx.set_row_sink(buf.data_ptr(), 100000)
for _ in range(5):
    g.scan(1, 50, 3, 9, flags=prf_native.SCAN_DEFER_TIMING, fetch=False)
t0 = time.perf_counter()
for _ in range(N):
    g.scan(1, 50, 3, 9, flags=prf_native.SCAN_DEFER_TIMING, fetch=False)
dt = (time.perf_counter() - t0) / N
print('chr22 scan into a row sink (what a rank of the multi-GPU bench does per step): %.2f us' % (dt * 1e6))
ctx.set_row_sink(None, 0)
