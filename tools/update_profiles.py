"""Copy the judged summaries of tools/final_profiles.sh from gpurun_out/final/ into profiles/r02/ and rebuild
profiles/pmc_traffic.json (HBM bytes per launch of the scan kernel = 2 x FETCH_SIZE + WRITE_SIZE, in KB: FETCH_SIZE is
doubled as MI355X_MICROARCH.md prescribes for gfx950's wide coalesced reads)."""
import json, shutil, os
src, dst = "gpurun_out/final", "profiles/r02"
os.makedirs(dst, exist_ok=True)
entries = []
for w in ("hg38", "chr22", "chr1"):
    shutil.copy(f"{src}/pmc_{w}/pmc.json", f"{dst}/pmc_{w}.json")
    shutil.copy(f"{src}/pmc_{w}/kernel_stats.csv", f"{dst}/rocprofv3_kernel_stats_{w}.csv")
    d = json.load(open(f"{src}/pmc_{w}/pmc.json"))
    f, wr = d["FETCH_SIZE"]["median_per_launch"], d["WRITE_SIZE"]["median_per_launch"]
    entries.append({"workload": w, "kmin": 1, "kmax": 50, "kernel": "prf_vscan_kernel", "FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB": wr,
                    "hbm_bytes_per_launch": int((2 * f + wr) * 1024),
                    "note": f"round 2, final kernel; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 for wide coalesced reads); "
                            f"median over launches; separate --pmc passes (profiles/r02/pmc_{w}.json)"})
json.dump(entries, open("profiles/pmc_traffic.json", "w"), indent=1)
for f in os.listdir(src):
    if f.startswith("bench_n1_") and f.endswith(".json"):
        shutil.copy(f"{src}/{f}", f"{dst}/{f}")
for n in (2, 4):
    if os.path.exists(f"{src}/rehearsal_n{n}.json"):
        shutil.copy(f"{src}/rehearsal_n{n}.json", f"{dst}/rehearsal_gloo_one_gpu_n{n}.json")
if os.path.exists(f"{src}/rehearsal_rand_n2.json"):
    shutil.copy(f"{src}/rehearsal_rand_n2.json", f"{dst}/rehearsal_gloo_one_gpu_c5_2Gbp_n2.json")
if os.path.exists(f"{src}/ic_hg38/pmc_icache.json"):
    shutil.copy(f"{src}/ic_hg38/pmc_icache.json", f"{dst}/pmc_hg38_icache_scalar.json")
print(json.dumps(entries, indent=1))
