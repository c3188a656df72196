#!/bin/bash
# one GPU round trip of the development loop: parity suite, three bench points, phase stamps.  usage: tools/gpu_cycle.sh <tag> [quick]
tag=$1
python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; tail -3 gpurun_out/${tag}_pytest.log
python bench.py --workload chr22 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_chr22.json 2>gpurun_out/${tag}_chr22.err
python bench.py --workload hg38-random --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_hg38.json 2>gpurun_out/${tag}_hg38.err
python bench.py --workload chr1 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_chr1.json 2>gpurun_out/${tag}_chr1.err
PRF_LIB=colab-repeat-finder_amd/libprf_stamps.so PRF_STAMPS_OUT=/tmp/st.bin python tools/stamps.py 2>/dev/null | grep -v "^\[prf\]" > gpurun_out/${tag}_stamps_chr22.txt
python bench.py --workload hg38 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_hg38s.json 2>gpurun_out/${tag}_hg38s.err
PRF_LIB=colab-repeat-finder_amd/libprf_stamps.so PRF_STAMPS_OUT=/tmp/st.bin python tools/stamps.py 400000000 2>/dev/null | grep -v "^\[prf\]" > gpurun_out/${tag}_stamps_400M.txt
echo cycle done
