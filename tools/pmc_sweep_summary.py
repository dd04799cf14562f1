"""Merge the per-pass counter CSVs written by tools/pmc_sweep.sh into one table per kernel."""
import collections
import csv
import glob
import sys

out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    kt = {}
    for r in csv.DictReader(open(f.replace("counter_collection", "kernel_trace"))):
        kt[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:48]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] in kt:
            agg[name]["dur_ns(pmc run)"].append(kt.pop(r["Dispatch_Id"]))
for name, ctrs in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("dur_ns(pmc run)", [0]))):
    print(name)
    for c, v in sorted(ctrs.items()):
        v2 = sorted(v)
        print("    %-34s mean %16.1f   max %16.1f   n=%d" % (c, sum(v) / len(v), v2[-1], len(v)))
