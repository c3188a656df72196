"""One line per bench JSON under gpurun_out/final/ (diagnostic)."""
import glob, json
for f in sorted(glob.glob("gpurun_out/final/*.json")):
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
        r = d.get("roofline", {})
        print(f.split("/")[-1], d["ms_per_step"], r.get("frac"), r.get("kernel_ms_min_median"), r.get("gather_kernel_ms"), r.get("traffic"), r.get("valu_busy"))
    except Exception as e:
        print(f, "unreadable", e)
