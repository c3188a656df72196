#!/bin/bash
# Refresh profiles/r02 inputs on the GPU box (outputs under gpurun_out/final/): usage tools/final_profiles.sh <part>
set -e
mkdir -p gpurun_out/final
case "$1" in
  pmc_hg38)  tools/pmc_passes.sh final/pmc_hg38 --steps 8 --warmup 2 > gpurun_out/final/pmc_hg38.log 2>&1 ;;
  pmc_small) tools/pmc_passes.sh final/pmc_chr22 --workload chr22 --steps 20 --warmup 3 > gpurun_out/final/pmc_chr22.log 2>&1
             tools/pmc_passes.sh final/pmc_chr1 --workload chr1 --steps 20 --warmup 3 > gpurun_out/final/pmc_chr1.log 2>&1 ;;
  bench)     python bench.py --steps 20 --warmup 5 > gpurun_out/final/bench_n1_hg38_default.json 2> gpurun_out/final/bench_default.err
             python bench.py --workload hg38-random --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/final/bench_n1_hg38_random.json 2>/dev/null
             python bench.py --workload chr22 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/final/bench_n1_chr22.json 2>/dev/null
             python bench.py --workload chr1 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/final/bench_n1_chr1.json 2>/dev/null
             python bench.py --workload random --kmax 100 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/final/bench_n1_c5_random_10Gbp_k100.json 2> gpurun_out/final/bench_c5.err ;;
esac
echo final $1 done
