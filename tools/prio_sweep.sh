#!/bin/bash
# issue-priority sweep on the default workload (diagnostic): PRF_PRIO = stage | busy<<2 | slack<<4 | flags<<6 | records<<8 | rows<<10
mkdir -p gpurun_out/prio
for cfg in 0xA02 0x000 0xA00 0xE02 0xF02 0xA42 0xAC2 0xA82 0xA03 0xA06 0xA0A 0xA0E 0xE42 0x602 0x202 0xA02; do
  PRF_PRIO=$cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prio/$cfg.json 2>/dev/null
  PRF_PRIO=$cfg python bench.py --workload hg38-random --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prio/r$cfg.json 2>/dev/null
  python - <<PY
import json
a=json.load(open('gpurun_out/prio/$cfg.json')); b=json.load(open('gpurun_out/prio/r$cfg.json'))
print('$cfg', 'standin kernel_ms', a['roofline']['kernel_ms'], a['ms_per_step'], 'random kernel_ms', b['roofline']['kernel_ms'], b['ms_per_step'], flush=True)
PY
done
