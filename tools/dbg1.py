import sys
sys.path.insert(0,'colab-repeat-finder_amd'); sys.path.insert(0,'.')
import prf_native, synth
from oracle import prf_oracle
ctx=prf_native.Context(0)
for name,seq in [('rand200k', synth.synth_bases(200_000, 3).tobytes()),
                 ('standin700k', synth.chr_standin(length=700_000, seed=5, n_head=70_000, n_tail=3_000, repeats_per_mbp=1800).tobytes())]:
    for spec in [(2,6,3,9),(1,50,3,9)]:
        rows,st=ctx.scan([seq],*spec)
        want=[(s,e,k) for s,e,_m,k in prf_oracle.detect_rows(seq,*spec)]
        got=[(int(r['start']),int(r['end']),int(r['k'])) for r in rows]
        print(name, spec, 'path',st.path,'hits',st.n_hits,'cand',st.n_candidates,'ok',got==want, len(want))
