"""Turn a gpurun_out/prof_<tag> directory (profiles/run_profiles.sh) into a small markdown summary."""
import collections
import csv
import json
import os
import sys

d = sys.argv[1]


def short(name):
    n = name.split("(")[0]
    n = n.replace("void ", "").replace("sageicp::", "")
    return n[:60]


print("# rocprofv3 summary (%s)\n" % os.path.basename(d.rstrip("/")))
try:
    b = json.load(open(os.path.join(d, "bench_kt.json")))
    print("bench under --kernel-trace: %.2f frames/s, %.3f ms/frame, %d iterations/frame\n"
          % (b["value"], b["ms_per_step"], b["config"]["iterations_per_frame"]))
except Exception as e:   # noqa
    print("(no bench json: %s)\n" % e)
p = os.path.join(d, "kt", "kt_kernel_stats.csv")
if os.path.exists(p):
    print("## --kernel-trace --stats\n\n| kernel | calls | avg us | total ms | % |\n|---|---|---|---|---|")
    for r in csv.DictReader(open(p)):
        print("| %s | %s | %.2f | %.3f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                 float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
print("\n## PMC passes (mean per launch of each kernel)\n\n| pass | kernel | counter | mean/launch | launches |\n|---|---|---|---|---|")
for sub in sorted(os.listdir(d)):
    p = os.path.join(d, sub, "pmc_counter_collection.csv")
    if not os.path.exists(p):
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        acc[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print("| %s | %s | %s | %.4g | %d |" % (sub, k, c, sum(v) / len(v), len(v)))
