"""Inter-kernel gaps of the ICP loop from a rocprofv3 --kernel-trace CSV (start/end timestamps)."""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for i, (s, e, n) in enumerate(rows):
    dur[n].append(e - s)
    if i: gap[(rows[i - 1][2][-40:], n[-40:])].append(s - rows[i - 1][1])
print("kernel durations (us): n, mean")
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print("  %-60s %7d %8.2f" % (n[-60:], len(v), sum(v) / len(v) / 1e3))
print("gaps prev->next (us): n, median")
for k, v in sorted(gap.items(), key=lambda kv: -len(kv[1]))[:8]:
    v.sort(); print("  %-40s -> %-40s %7d %8.2f" % (k[0], k[1], len(v), v[len(v) // 2] / 1e3))
span = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e, _ in rows)
print("timeline span %.1f us, kernels busy %.1f us, idle %.1f us" % (span / 1e3, busy / 1e3, (span - busy) / 1e3))
print("idle by transition (us total, count, mean):")
for k, v in sorted(gap.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print("  %-40s -> %-40s %10.1f %6d %8.2f" % (k[0], k[1], sum(v) / 1e3, len(v), sum(v) / len(v) / 1e3))
