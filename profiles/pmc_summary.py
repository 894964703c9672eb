"""Mean per-dispatch counter values of one kernel from rocprofv3 --pmc CSVs (counter_collection)."""
import csv, glob, sys, collections
kern = sys.argv[2] if len(sys.argv) > 2 else "k_nn"
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("%-34s n=%5d mean=%16.1f" % (k, len(v), sum(v) / len(v)))
