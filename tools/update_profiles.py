"""Copy the judged summaries of tools/final_profiles.sh from gpurun_out/final/ into profiles/r03/ and rebuild
profiles/pmc_traffic.json: HBM bytes per launch of the scan kernel and of the row gather = 2 x FETCH_SIZE + WRITE_SIZE (KB;
FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950's wide coalesced reads), the share of the VALU issue cycles the
scan kernel uses (VALU wave-instructions / 1024 SIMDs x 1.8 cycles / (GRBM_GUI_ACTIVE / 8 XCDs), the formula of VERDICT r2) and the
share of their life its waves are parked (SQ_WAIT_ANY / SQ_WAVE_CYCLES)."""
import json, shutil, os, glob
src, dst, rnd = "gpurun_out/final", "profiles/r03", "round 3"
os.makedirs(dst, exist_ok=True)
entries = []
for w in ("hg38", "chr22", "chr1", "chr22-real"):
    if not os.path.exists(f"{src}/pmc_{w}/pmc.json"):
        continue
    shutil.copy(f"{src}/pmc_{w}/pmc.json", f"{dst}/pmc_{w}.json")
    shutil.copy(f"{src}/pmc_{w}/pmc_gather.json", f"{dst}/pmc_{w}_gather.json")
    shutil.copy(f"{src}/pmc_{w}/kernel_stats.csv", f"{dst}/rocprofv3_kernel_stats_{w}.csv")
    d = json.load(open(f"{src}/pmc_{w}/pmc.json"))
    gth = json.load(open(f"{src}/pmc_{w}/pmc_gather.json"))
    m = lambda dd, k: dd[k]["median_per_launch"]
    f, wr = m(d, "FETCH_SIZE"), m(d, "WRITE_SIZE")
    entries.append({"workload": w, "kmin": 1, "kmax": 50, "kernel": "prf_vscan_kernel", "FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB": wr,
                    "hbm_bytes_per_launch": int((2 * f + wr) * 1024),
                    "gather_hbm_bytes_per_launch": int((2 * m(gth, "FETCH_SIZE") + m(gth, "WRITE_SIZE")) * 1024),
                    "valu_busy": round(m(d, "SQ_INSTS_VALU") / 1024 * 1.8 / (m(d, "GRBM_GUI_ACTIVE") / 8), 4),
                    "wait_any_over_wave_cycles": round(m(d, "SQ_WAIT_ANY") / m(d, "SQ_WAVE_CYCLES"), 4),
                    "note": f"{rnd}, final kernel; rocprofv3 --pmc passes of this bench.py command line, separate runs (profiles/r03/pmc_{w}.json, "
                            f"pmc_{w}_gather.json); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 for wide coalesced reads); median "
                            f"over launches; a committed measurement, not a live counter"})
json.dump(entries, open("profiles/pmc_traffic.json", "w"), indent=1)
for f in glob.glob(f"{src}/bench_n1_*.json") + glob.glob(f"{src}/rehearsal_*.json"):
    shutil.copy(f, f"{dst}/{os.path.basename(f).replace('rehearsal_', 'rehearsal_gloo_one_gpu_')}")
for f in ("stress_300s.txt", "pytest_gpu.log", "literal_timing.txt"):
    if os.path.exists(f"{src}/{f}"):
        shutil.copy(f"{src}/{f}", f"{dst}/{f}")
print(json.dumps(entries, indent=1))
