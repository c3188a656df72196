"""Summarise rocprofv3 --pmc passes (one directory per pass) for one kernel: median counter value per launch.
usage: python tools/pmc_summary.py <kernel-substring> <out.json> <pass-dir> [<pass-dir> ...]"""
import collections, csv, glob, json, os, sys

kernel, out = sys.argv[1], sys.argv[2]
res = {}
for d in sys.argv[3:]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(path)):
            if kernel in r["Kernel_Name"]:
                agg[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for name, per in agg.items():
            vals = sorted(per.values())
            res[name] = {"launches": len(vals), "median_per_launch": vals[len(vals) // 2]}
json.dump(dict(sorted(res.items())), open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
