#!/bin/bash
# N ranks on ONE GPU (gloo for the collectives): rehearsal of bench.py's N > 1 path.  usage: tools/gpu_multi.sh <tag>
tag=$1
export PRF_BENCH_BACKEND=gloo PRF_BENCH_ONE_GPU=1
for n in 2 4; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2950$n bench.py --gpus $n --steps 5 --warmup 2 > gpurun_out/${tag}_n$n.json 2> gpurun_out/${tag}_n$n.err
done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --workload random --length 2000000000 > gpurun_out/${tag}_rand_n2.json 2> gpurun_out/${tag}_rand_n2.err
echo multi done
