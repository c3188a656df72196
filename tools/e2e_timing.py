"""End-to-end (PCIe-inclusive) timing of one scan: host ASCII -> rows on the host.  Never the headline `value`
(bench.py times the scan with the genome resident in HBM); reported in DESIGN.md for completeness.
usage: python tools/e2e_timing.py [length]"""
import sys, time
sys.path.insert(0, 'colab-repeat-finder_amd'); sys.path.insert(0, '.')
import prf_native, synth
L = int(sys.argv[1]) if len(sys.argv) > 1 else synth.CHR22_LEN
seq = synth.chr_standin(length=L, seed=22, n_head=10_510_000 if L > 2e7 else L // 10, n_tail=10_000).tobytes()
ctx = prf_native.Context(0)
for rep in range(3):
    t0 = time.perf_counter()
    g = ctx.load([seq], 50)
    t1 = time.perf_counter()
    rows, st = g.scan(1, 50, 3, 9)
    t2 = time.perf_counter()
    g.free()
    print(f"rep {rep}: load+pack {1e3*(t1-t0):.2f} ms, scan+fetch+sort {1e3*(t2-t1):.2f} ms (device scan {st.scan_ms:.3f} ms), "
          f"total {1e3*(t2-t0):.2f} ms = {L/(t2-t0)/1e9:.2f} Gbp/s end to end, {len(rows)} rows")
