"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs) of bench.py into profiles/pmc_traffic.json.
usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <length> <kmin> <kmax>
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads, so it is
doubled (MI355X_MICROARCH.md, HBM section)."""
import collections, csv, json, os, sys

def per_launch(path, counter, kernel):
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]:
            agg[r["Dispatch_Id"]] += float(r["Counter_Value"])
    vals = sorted(agg.values())
    return vals[len(vals) // 2]

fetch_csv, write_csv, length, kmin, kmax = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
fetch_kb = per_launch(fetch_csv, "FETCH_SIZE", "prf_vscan_kernel")
write_kb = per_launch(write_csv, "WRITE_SIZE", "prf_vscan_kernel")
rec = {"length": length, "kmin": kmin, "kmax": kmax, "kernel": "prf_vscan_kernel",
       "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB": write_kb,
       "hbm_bytes_per_launch": int((2 * fetch_kb + write_kb) * 1024),
       "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 for wide coalesced reads); median over launches"}
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
recs = [r for r in (json.load(open(out)) if os.path.exists(out) else []) if not (r["length"] == length and r["kmin"] == kmin and r["kmax"] == kmax)]
recs.append(rec)
json.dump(recs, open(out, "w"), indent=1)
print(rec)
