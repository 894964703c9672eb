"""Copy the judged artefacts of gpurun_out/prof_<tag> (profiles/run_profiles.sh) into profiles/<tag>/
and refresh profiles/icp_counters.json (what bounds k_icp, read by bench.py into `roofline`)."""
import collections, csv, gzip, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join("gpurun_out", "prof_" + tag)
dst = os.path.join("profiles", tag)
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "summary.md"), os.path.join(dst, "summary.md"))
shutil.copy(os.path.join(src, "kt", "kt_kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench_kt.json"), os.path.join(dst, "bench_under_kernel_trace.json"))
if os.path.exists(os.path.join(src, "bench_default.json")):
    shutil.copy(os.path.join(src, "bench_default.json"), os.path.join(dst, "bench_default.json"))
mean = {}
for sub in ("pmc_fetch", "pmc_tcc", "pmc_sq", "pmc_sq2", "pmc_mem"):
    p = os.path.join(src, sub, "pmc_counter_collection.csv")
    if not os.path.exists(p):
        continue
    with open(p, "rb") as f, gzip.open(os.path.join(dst, sub + ".csv.gz"), "wb") as g:
        g.write(f.read())
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if "k_icp" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        mean[k] = sum(v) / len(v)
        mean["_launches_" + k] = len(v)
kt = {r["Name"]: r for r in csv.DictReader(open(os.path.join(src, "kt", "kt_kernel_stats.csv")))}
icp = [r for n, r in kt.items() if "k_icp" in n][0]
avg_us = float(icp["AverageNs"]) / 1e3
bench = json.load(open(os.path.join(src, "bench_default.json")))
rf = bench["roofline"]
pairs_per_launch = rf["pairs_evaluated_frac"] * rf["candidates_per_query"] * rf["queries_per_launch"]
# VALU busy: every VALU wave-instruction holds its SIMD for >= 4 cycles (fp64 and DPP forms; the
# 2-cycle fp32 forms are a minority here); 256 CUs x 4 SIMDs; cycles = kernel duration x clock
cycles = avg_us * 2.3e3          # shader clock under this load: 2.3 GHz (profiles/phase_probe.py)
valu = mean.get("SQ_INSTS_VALU", 0.0)
out = {
    "kernel": "k_icp",
    "counters_source": "profiles/%s/pmc_*.csv.gz (rocprofv3 --pmc, separate passes, means per launch over %d launches)"
                       % (tag, int(mean.get("_launches_SQ_INSTS_VALU", 0))),
    "avg_launch_us_kernel_trace": round(avg_us, 2),
    "fetch_size_kb": mean.get("FETCH_SIZE"),
    "hbm_bytes_per_launch": int(round(mean["FETCH_SIZE"] * 1024 * 2)) if "FETCH_SIZE" in mean else None,
    "hbm_correction": "x2: on gfx950 FETCH_SIZE reports half the bytes of 16-B/lane reads (MI355X_MICROARCH.md, HBM); "
                      "Infinity-Cache hits are counted in it, so DRAM traffic is lower still",
    "valu_insts_per_launch": valu,
    "salu_insts_per_launch": mean.get("SQ_INSTS_SALU"),
    "valu_frac": round(valu * 4.0 / (1024.0 * cycles), 4) if valu else None,
    "lane_utilization": round(mean["SQ_THREAD_CYCLES_VALU"] / (mean["SQ_ACTIVE_INST_VALU"] * 64.0), 4)
                        if "SQ_THREAD_CYCLES_VALU" in mean and mean.get("SQ_ACTIVE_INST_VALU") else None,
    # the VALU instructions spent on one scanned (query, map point) pair — 12 in the fp32 filter of
    # the compact scan, 22 in the fp64 comparison of the full-record scan — 64 pairs per wave-instruction
    "useful_inst_frac": round(pairs_per_launch / 64.0 * (12.0 if "compact" in rf.get("scan_form", "") else 22.0) / valu, 4) if valu else None,
    "l2_hit_rate": round(mean["TCC_HIT_sum"] / (mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"]), 4)
                   if "TCC_HIT_sum" in mean else None,
}
json.dump(out, open(os.path.join("profiles", "icp_counters.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
