"""Copy the judged artefacts of gpurun_out/prof_<tag>_<workload>-<params> (profiles/run_profiles.sh)
into profiles/<tag>/<workload>-<params>/ and refresh that workload's entry of
profiles/icp_counters.json: what bounds k_icp, from the counter passes, tied to the device sources
they were collected on (bench.py uses an entry only when that hash matches the code it runs).
usage: python profiles/collect.py <tag> [workload] [params]"""
import collections
import csv
import glob
import gzip
import json
import os
import shutil
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "profiles"))
import summarize as S
from bench import device_source_hash, HBM_PEAK_GBS

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
wl = sys.argv[2] if len(sys.argv) > 2 else "c2"
prm = sys.argv[3] if len(sys.argv) > 3 else "cold"
key = "%s-%s" % (wl, prm)
src = os.path.join("gpurun_out", "prof_%s_%s" % (tag, key))
dst = os.path.join("profiles", tag, key)
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "summary.md"), os.path.join(dst, "summary.md"))
shutil.copy(os.path.join(src, "kt", "kt_kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench_kt.json"), os.path.join(dst, "bench_under_kernel_trace.json"))
if os.path.exists(os.path.join(src, "bench_default.json")):
    shutil.copy(os.path.join(src, "bench_default.json"), os.path.join(dst, "bench_default.json"))
trace = glob.glob(os.path.join(src, "kt", "**", "*kernel_trace.csv"), recursive=True)[0]
ex = S.executed_stats(trace)
with open(os.path.join(dst, "kernel_stats_executed.csv"), "w") as f:
    f.write('"Name","Launches","NoOpLaunches","AverageNs","MinNs","MaxNs","TotalDurationNs"\n')
    for k, v in sorted(ex.items(), key=lambda kv: -kv[1][5]):
        f.write('"%s",%d,%d,%.1f,%.0f,%.0f,%.0f\n' % (k, v[1], v[0] - v[1], v[2], v[3], v[4], v[5]))
bench = json.load(open(os.path.join(src, "bench_default.json")))
rf = bench["roofline"]
iters = int(bench["config"]["iterations_per_frame"])
# which kernel is the loop made of in this workload?  k_loop: one launch = one frame = `iters` iterations
# (its counters come from the counter-collection twin, profiles/run_profiles.sh); k_icp: one launch = one iteration
loop = [k for k in ex if "k_loop<" in k]
kname = "k_loop" if loop else "k_icp"
per = float(iters) if loop else 1.0
mean = {}
for sub in ("pmc_fetch", "pmc_tcc", "pmc_sq", "pmc_sq2", "pmc_mem"):
    p = os.path.join(src, sub, "pmc_counter_collection.csv")
    if not os.path.exists(p):
        continue
    with open(p, "rb") as f, gzip.open(os.path.join(dst, sub + ".csv.gz"), "wb") as g:
        g.write(f.read())
    acc = collections.defaultdict(list)
    dur = []
    for r in csv.DictReader(open(p)):
        if kname + "<" not in r["Kernel_Name"]:
            continue
        t = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        if t < S.noop_limit(r["Kernel_Name"]):
            continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]) / per)
        dur.append(t / per)
    for k, v in acc.items():
        mean[k] = sum(v) / len(v)
        mean["_launches_" + k] = len(v)
    if dur:
        mean["_us_" + sub] = sum(dur) / len(dur) / 1e3
icp = [v for k, v in ex.items() if kname + "<" in k][0]
fin = [v for k, v in ex.items() if "k_fin" in k]
avg_us = icp[2] / 1e3 / per              # per iteration
pairs_per_launch = rf["pairs_evaluated_frac"] * rf["candidates_per_query"] * rf["queries_per_launch"]
# VALU busy: every VALU wave-instruction holds its SIMD for >= 4 cycles (fp64 and DPP forms; the
# 2-cycle fp32 forms are a minority here); 256 CUs x 4 SIMDs; cycles = kernel duration x clock
cycles = avg_us * 2.3e3          # shader clock under this load: 2.3 GHz (profiles/phase_probe.py)
valu = mean.get("SQ_INSTS_VALU", 0.0)
hbm_bytes = int(round(mean["FETCH_SIZE"] * 1024 * 2)) if "FETCH_SIZE" in mean else None
# bytes and duration of the SAME (counter-pass) launches; the kernel-trace duration beside it
fetch_us = mean.get("_us_pmc_fetch", avg_us)
entry = {
    "kernel": kname,
    "workload": key,
    "iterations_per_launch": int(per),
    "source_sha256": device_source_hash(),
    "counters_source": "profiles/%s/%s/pmc_*.csv.gz (rocprofv3 --pmc, separate passes, means per iteration "
                       "over %d launches%s)"
                       % (tag, key, int(mean.get("_launches_SQ_INSTS_VALU", mean.get("_launches_FETCH_SIZE", 0))),
                          "; k_loop: counters of the counter-collection twin of the library (solving wave inside the grid: "
                          "--pmc runs one kernel at a time), divided by the launch's %d iterations; the duration is the "
                          "PRODUCT's, from --kernel-trace" % iters if loop else
                          "; no-op launches of a finished loop excluded"),
    # (the source hash covers the sources, not the -D flags or switches of the counter passes: said here)
    "counter_build": ("-DSAGE_LOOP_INGRID twin of the library (sage-icp_amd/_probe/libsageicp_ingrid.so): bytes and instruction counts of "
                      "the twin over the kernel-trace time of the product" if loop else
                      "the product library with SAGEICP_CHAIN=0 (k_fin between the launches instead of the resident solving wave: "
                      "the same k_icp)"),
    "avg_launch_us_kernel_trace": round(avg_us, 2),
    "k_fin_avg_us_kernel_trace": round(fin[0][2] / 1e3, 2) if fin and not loop else None,
    "avg_launch_us_fetch_pass": round(fetch_us, 2),
    "fetch_size_kb": mean.get("FETCH_SIZE"),
    "hbm_bytes_per_launch": hbm_bytes,
    "hbm_frac": round(hbm_bytes / (avg_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if hbm_bytes else None,
    "hbm_frac_definition": "FETCH_SIZE x 2 per iteration / kernel-trace duration per iteration / %.0f GB/s" % HBM_PEAK_GBS,
    "traffic_floor_us": round(hbm_bytes / 6.3e12 * 1e6, 2) if hbm_bytes else None,
    "x_over_traffic_floor": round(avg_us / (hbm_bytes / 6.3e12 * 1e6), 2) if hbm_bytes else None,
    "hbm_correction": "x2: on gfx950 FETCH_SIZE reports half the bytes of 16-B/lane reads (MI355X_MICROARCH.md, HBM); "
                      "Infinity-Cache hits are counted in it, so DRAM traffic is lower still",
    "valu_insts_per_launch": valu,
    "salu_insts_per_launch": mean.get("SQ_INSTS_SALU"),
    "valu_frac": round(valu * 4.0 / (1024.0 * cycles), 4) if valu else None,
    "wait_frac": round(mean["SQ_WAIT_INST_ANY"] / mean["SQ_WAVE_CYCLES"], 4) if mean.get("SQ_WAVE_CYCLES") and "SQ_WAIT_INST_ANY" in mean else None,
    "lane_utilization": round(mean["SQ_THREAD_CYCLES_VALU"] / (mean["SQ_ACTIVE_INST_VALU"] * 64.0), 4)
                        if "SQ_THREAD_CYCLES_VALU" in mean and mean.get("SQ_ACTIVE_INST_VALU") else None,
    # the VALU instructions spent on one scanned (query, map point) pair — 12 in the fp32 filter of
    # the compact scan, 22 in the fp64 comparison of the full-record scan — 64 pairs per wave-instruction
    "useful_inst_frac": round(pairs_per_launch / 64.0 * (12.0 if "compact" in rf.get("scan_form", "") else 22.0) / valu, 4) if valu else None,
    "l2_hit_rate": round(mean["TCC_HIT_sum"] / (mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"]), 4)
                   if "TCC_HIT_sum" in mean else None,
}
path = os.path.join("profiles", "icp_counters.json")
allc = {}
if os.path.exists(path):
    try:
        allc = json.load(open(path))
        if "kernel" in allc:          # round-2 layout (a single c2-cold entry, no source hash)
            allc = {}
    except Exception:
        allc = {}
allc[key] = entry
json.dump(allc, open(path, "w"), indent=1)
print(json.dumps(entry, indent=1))
