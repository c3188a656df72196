"""End-to-end timing of the drop-in command line on a synthetic multi-contig FASTA (written to /tmp).
usage: python tools/cli_timing.py [total_bp]"""
import os, sys, time
sys.path.insert(0, 'colab-repeat-finder_amd'); sys.path.insert(0, '.')
import synth
total = int(sys.argv[1]) if len(sys.argv) > 1 else 120_000_000
lens = [int(total * f) for f in (0.45, 0.3, 0.15, 0.07, 0.03)]
path = '/tmp/prf_cli_timing.fa'
t0 = time.perf_counter()
with open(path, 'wb') as f:
    for i, n in enumerate(lens):
        seq = synth.chr_standin(length=n, seed=50 + i, n_head=n // 50, n_tail=n // 500).tobytes()
        f.write(b'>chr%d synthetic\n' % (i + 1))
        body = bytearray()
        for j in range(0, n, 60):
            body += seq[j:j + 60] + b'\n'
        f.write(body)
print('wrote', path, os.path.getsize(path) / 1e6, 'MB in %.1f s' % (time.perf_counter() - t0))
import perfect_repeat_finder as prf
os.chdir('/tmp')
t0 = time.perf_counter()
prf.main(['-o', 'prf_cli_timing', path])
dt = time.perf_counter() - t0
rows = sum(1 for _ in open('/tmp/prf_cli_timing.bed'))
print(f'CLI FASTA -> BED: {dt:.2f} s for {sum(lens)/1e6:.0f} Mbp = {sum(lens)/dt/1e9:.3f} Gbp/s end to end, {rows} rows')
