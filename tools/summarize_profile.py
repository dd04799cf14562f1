#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace + PMC passes) into a per-kernel table."""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(root, suffix):
    return sorted(glob.glob(os.path.join(root, "**", "*" + suffix), recursive=True))


def short(name):
    for tok in ("(anonymous namespace)::", "void "):
        name = name.replace(tok, "")
    name = name.split("(")[0]
    return name[:70]


def main(root):
    # timing
    for f in find(os.path.join(root, "trace"), "kernel_trace.csv"):
        dur = defaultdict(list)
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        print("== kernel trace: %s" % os.path.relpath(f, root))
        print("%-72s %8s %12s %12s %12s" % ("kernel", "calls", "avg_us", "min_us", "total_ms"))
        for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
            print("%-72s %8d %12.2f %12.2f %12.3f" % (k, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, sum(v) / 1e6))
    # counters
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
        for f in find(os.path.join(root, sub), "counter_collection.csv"):
            agg = defaultdict(lambda: defaultdict(list))
            for r in csv.DictReader(open(f)):
                agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
            print("== counters: %s (per-dispatch averages)" % os.path.relpath(f, root))
            for k, cs in sorted(agg.items()):
                parts = ["%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())]
                print("%-72s n=%d %s" % (k, len(next(iter(cs.values()))), " ".join(parts)))


def traffic_json(root, out_path):
    """Per-kernel average FETCH_SIZE / WRITE_SIZE (KiB per dispatch) keyed by template name."""
    import json
    res = {}
    for sub, key in (("pmc_fetch", "fetch_kb"), ("pmc_write", "write_kb")):
        for f in find(os.path.join(root, sub), "counter_collection.csv"):
            agg = defaultdict(list)
            for r in csv.DictReader(open(f)):
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
            for k, v in agg.items():
                res.setdefault(k, {})[key] = sum(v) / len(v)
    json.dump(res, open(out_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1])
    if len(sys.argv) > 2:
        traffic_json(sys.argv[1], sys.argv[2])
