#!/bin/bash
# rocprofv3 passes over one bench.py command line, on the GPU box:
#   tools/pmc_passes.sh <tag> <bench.py arguments ...>
# writes gpurun_out/<tag>/{kt,sq1,sq2,fetch,write}/... and gpurun_out/<tag>/pmc.json (median per launch of the scan
# kernel, tools/pmc_summary.py).  Counter passes are separate runs, each with --kernel-trace only
# (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md, rocprofv3 PMC slots).
set -e
tag=$1; shift
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
args=("$@" --no-cpu-baseline)
rocprofv3 --output-format csv --kernel-trace --stats -d "$out/kt" -o run -- python3 bench.py "${args[@]}" > "$out/kt.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d "$out/sq1" -o run -- python3 bench.py "${args[@]}" > "$out/sq1.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d "$out/sq2" -o run -- python3 bench.py "${args[@]}" > "$out/sq2.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d "$out/fetch" -o run -- python3 bench.py "${args[@]}" > "$out/fetch.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d "$out/write" -o run -- python3 bench.py "${args[@]}" > "$out/write.log" 2>&1
python3 tools/pmc_summary.py prf_vscan "$out/pmc.json" "$out/sq1" "$out/sq2" "$out/fetch" "$out/write" > /dev/null
python3 tools/pmc_summary.py prf_vgather "$out/pmc_gather.json" "$out/sq1" "$out/sq2" "$out/fetch" "$out/write" > /dev/null
find "$out/kt" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats.csv" \;
cat "$out/pmc.json"
head -5 "$out/kernel_stats.csv"
