"""Copy the judged artefacts of gpurun_out/prof_<tag> (profiles/run_profiles.sh) into profiles/<tag>/
and refresh profiles/nn_traffic.json (HBM bytes per k_nn launch, read by bench.py)."""
import csv, gzip, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join("gpurun_out", "prof_" + tag)
dst = os.path.join("profiles", tag)
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "summary.md"), os.path.join(dst, "summary.md"))
shutil.copy(os.path.join(src, "kt", "kt_kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench_kt.json"), os.path.join(dst, "bench_under_kernel_trace.json"))
if os.path.exists(os.path.join(src, "bench_default.json")):
    shutil.copy(os.path.join(src, "bench_default.json"), os.path.join(dst, "bench_default.json"))
for sub in ("pmc_fetch", "pmc_tcc", "pmc_sq", "pmc_mem"):
    p = os.path.join(src, sub, "pmc_counter_collection.csv")
    if os.path.exists(p):
        with open(p, "rb") as f, gzip.open(os.path.join(dst, sub + ".csv.gz"), "wb") as g:
            g.write(f.read())
vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(os.path.join(src, "pmc_fetch", "pmc_counter_collection.csv")))
        if "k_nn" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
mean_kb = sum(vals) / len(vals)
json.dump({
    "kernel": "k_nn",
    "counter": "FETCH_SIZE (KB, rocprofv3 --pmc, separate pass)",
    "mean_per_launch_kb": mean_kb,
    "correction": "x2: on gfx950 FETCH_SIZE reports half the bytes of 16-B/lane reads (MI355X_MICROARCH.md, HBM section)",
    "hbm_bytes_per_launch": int(round(mean_kb * 1024 * 2)),
    "launches": len(vals),
    "source": "profiles/%s/pmc_fetch.csv.gz" % tag,
}, open(os.path.join("profiles", "nn_traffic.json"), "w"), indent=1)
print(open(os.path.join("profiles", "nn_traffic.json")).read())
