#!/bin/bash
# Refresh the judged measurements on the GPU box (outputs under gpurun_out/final/): usage tools/final_profiles.sh <part>
# Counter passes are separate rocprofv3 runs, each with --kernel-trace only (tools/pmc_passes.sh).
set -e
mkdir -p gpurun_out/final
case "$1" in
  pmc_hg38)  tools/pmc_passes.sh final/pmc_hg38 --steps 8 --warmup 2 > gpurun_out/final/pmc_hg38.log 2>&1 ;;
  pmc_small) tools/pmc_passes.sh final/pmc_chr22 --workload chr22 --steps 20 --warmup 3 > gpurun_out/final/pmc_chr22.log 2>&1
             tools/pmc_passes.sh final/pmc_chr1 --workload chr1 --steps 20 --warmup 3 > gpurun_out/final/pmc_chr1.log 2>&1
             tools/pmc_passes.sh final/pmc_chr22-real --workload chr22-real --steps 20 --warmup 3 > gpurun_out/final/pmc_chr22-real.log 2>&1 ;;
  bench)     python bench.py --steps 20 --warmup 5 > gpurun_out/final/bench_n1_hg38_default.json 2> gpurun_out/final/bench_default.err
             python bench.py --steps 20 --warmup 5 > gpurun_out/final/bench_n1_hg38_default_again.json 2>> gpurun_out/final/bench_default.err
             python bench.py --workload hg38-random --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/final/bench_n1_hg38_random.json 2>/dev/null
             python bench.py --workload chr22 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/final/bench_n1_chr22.json 2>/dev/null
             python bench.py --workload chr22-real --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/final/bench_n1_chr22-real.json 2>/dev/null
             python bench.py --workload chr22-real --kmax 6 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/final/bench_n1_chr22-real_k6.json 2>/dev/null
             python bench.py --workload chr1 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/final/bench_n1_chr1.json 2>/dev/null
             python bench.py --workload random --kmax 100 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/final/bench_n1_c5_random_10Gbp_k100.json 2> gpurun_out/final/bench_c5.err ;;
  rehearsal) export PRF_BENCH_BACKEND=gloo PRF_BENCH_ONE_GPU=1
             for n in 2 4; do python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2950$n bench.py --gpus $n --steps 10 --warmup 3 2> gpurun_out/final/rehearsal_n$n.err | grep '^{' > gpurun_out/final/rehearsal_n$n.json; done
             python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 10 --warmup 3 --gather-every 0 2> gpurun_out/final/rehearsal_n2_nogather.err | grep '^{' > gpurun_out/final/rehearsal_n2_nogather.json ;;
esac
echo final $1 done
