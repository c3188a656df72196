#!/bin/bash
# issue-priority sweep on the default workload (diagnostic): PRF_PRIO = stage | busy<<2 | slack<<4 | flags<<6 | records<<8 | rows<<10
mkdir -p gpurun_out/prio
for cfg in 0x000 0xA00 0xB00 0xE00 0xF00 0xA40 0xA01 0xA02 0xA04 0xA08 0xA0C 0xE01 0xA00 0x000; do
  PRF_PRIO=$cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prio/$cfg.json 2>/dev/null
  PRF_PRIO=$cfg python bench.py --workload hg38-random --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prio/r$cfg.json 2>/dev/null
  python - <<PY
import json
a=json.load(open('gpurun_out/prio/$cfg.json')); b=json.load(open('gpurun_out/prio/r$cfg.json'))
print('$cfg', 'standin kernel_ms', a['roofline']['kernel_ms'], 'random kernel_ms', b['roofline']['kernel_ms'], flush=True)
PY
done
