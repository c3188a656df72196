import json, sys
tag = sys.argv[1]
for f in ("chr22", "chr1", "hg38", "hg38s"):
    try:
        d = json.load(open(f"gpurun_out/{tag}_{f}.json"))
        print(f, "ms/step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "Gbp/s", d["value"], "rows", d["config"]["rows_rank0"], "cand", d["config"]["candidate_records_rank0"], "gather_ms", d["roofline"]["gather_kernel_ms"])
    except Exception as e:
        print(f, "ERR", e, open(f"gpurun_out/{tag}_{f}.err").read()[-600:])
for f in ("chr22", "400M"):
    print(open(f"gpurun_out/{tag}_stamps_{f}.txt").read())
