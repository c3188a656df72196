#!/bin/bash
# Extra counter passes (instruction cache, scalar cache, branch and LDS wait counts) over one bench.py command line:
#   tools/pmc_icache.sh <tag> <bench.py arguments ...>   -> gpurun_out/<tag>/pmc_icache.json
set -e
tag=$1; shift
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
args=("$@" --no-cpu-baseline)
rocprofv3 --output-format csv --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -d "$out/ic1" -o run -- python3 bench.py "${args[@]}" > "$out/ic1.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_BUSY_CYCLES -d "$out/ic2" -o run -- python3 bench.py "${args[@]}" > "$out/ic2.log" 2>&1
python3 tools/pmc_summary.py prf_vscan "$out/pmc_icache.json" "$out/ic1" "$out/ic2" > /dev/null
cat "$out/pmc_icache.json"
