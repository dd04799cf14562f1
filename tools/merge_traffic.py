#!/usr/bin/env python3
"""Merge the per-run traffic.json files tools/profile_bench.sh writes into profiles/latest_traffic.json, the table
bench.py's `roofline.traffic` is looked up in:  merge_traffic.py <workload key>=<traffic.json> ...
(workload key = bench.py --workload, plus "+first_<entry>" for a non-default first-layer entry)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(ROOT, "profiles", "latest_traffic.json")
d = json.load(open(path)) if os.path.exists(path) else {}
d.setdefault("by_workload", {})
for arg in sys.argv[1:]:
    key, f = arg.split("=", 1)
    d["by_workload"][key] = json.load(open(f))["by_tag"]
    print(key, sorted(d["by_workload"][key]))
json.dump(d, open(path, "w"), indent=1, sort_keys=True)
