#!/bin/bash
# usage: tools/pmc_pass.sh <tag> "<counters>" [bench args]  -- one rocprofv3 PMC pass over bench.py
TAG=$1; CTRS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $OUT/bench.log 2>&1 || { tail -5 $OUT/bench.log; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
root = sys.argv[1]
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    agg = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
        agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in sorted(agg.items()):
        if "k_conv" not in k: continue
        print(k, " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
PY
find $OUT -name "*.db" -delete
