"""Turn a gpurun_out/prof_<tag>_<workload> directory (profiles/run_profiles.sh) into a small markdown
summary.  Launches that only found the loop finished (they return at once: duration under 6 us for
k_icp / under 5 us for k_fin; the host keeps a few iterations enqueued ahead) are counted apart,
not averaged in."""
import collections
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
NOOP_NS = {"k_icp": 6000.0, "k_fin": 5000.0}


def short(name):
    n = name.split("(")[0]
    n = n.replace("void ", "").replace("sageicp::", "")
    return n[:60]


def noop_limit(name):
    for k, v in NOOP_NS.items():
        if k in name:
            return v
    return 0.0


def executed_stats(path):
    """per kernel: (calls, executed, mean / min / max ns of the executed launches) from a
    kernel-trace CSV"""
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    out = {}
    for k, v in acc.items():
        ex = [x for x in v if x >= noop_limit(k)]
        if ex:
            out[k] = (len(v), len(ex), sum(ex) / len(ex), min(ex), max(ex), sum(ex))
    return out


if __name__ == "__main__":
    print("# rocprofv3 summary (%s)\n" % os.path.basename(d.rstrip("/")))
    try:
        b = json.load(open(os.path.join(d, "bench_kt.json")))
        print("bench under --kernel-trace (%s): %.2f frames/s, %.3f ms/frame, %d iterations/frame\n"
              % (b["config"]["workload"].split(":")[0], b["value"], b["ms_per_step"],
                 b["config"]["iterations_per_frame"]))
    except Exception as e:   # noqa
        print("(no bench json: %s)\n" % e)
    tr = glob.glob(os.path.join(d, "kt", "**", "*kernel_trace.csv"), recursive=True)
    if tr:
        st = executed_stats(tr[0])
        tot = sum(v[5] for v in st.values())
        print("## --kernel-trace, executed launches only (no-op launches of a finished loop apart)\n\n"
              "| kernel | launches | no-ops | avg us | min us | max us | total ms | % |\n|---|---|---|---|---|---|---|---|")
        for k, v in sorted(st.items(), key=lambda kv: -kv[1][5]):
            print("| %s | %d | %d | %.2f | %.2f | %.2f | %.3f | %.1f |"
                  % (short(k), v[1], v[0] - v[1], v[2] / 1e3, v[3] / 1e3, v[4] / 1e3, v[5] / 1e6, 100.0 * v[5] / tot))
    p = os.path.join(d, "kt", "kt_kernel_stats.csv")
    if os.path.exists(p):
        print("\n## --kernel-trace --stats as rocprofv3 prints it (every launch)\n\n| kernel | calls | avg us | total ms | % |\n|---|---|---|---|---|")
        for r in csv.DictReader(open(p)):
            print("| %s | %s | %.2f | %.3f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                     float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
    print("\n## PMC passes (mean per executed launch of each kernel; k_loop: one launch = one frame = all its iterations,\n"
          "collected on the counter-collection twin of the library, solving wave inside the grid)\n\n| pass | kernel | counter | mean/launch | launches |\n|---|---|---|---|---|")
    for sub in sorted(os.listdir(d)):
        p = os.path.join(d, sub, "pmc_counter_collection.csv")
        if not os.path.exists(p):
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(p)):
            if float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) < noop_limit(r["Kernel_Name"]):
                continue
            acc[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            print("| %s | %s | %s | %.4g | %d |" % (sub, k, c, sum(v) / len(v), len(v)))
