#!/bin/bash
# Run on the GPU box: rocprofv3 kernel-trace durations of bench.py for the in-tree library and for
# any number of alternative builds csrc/libqnn_v_<tag>.so (built with -D switches for A/B timing).
# Usage: tools/trace_kernels.sh [tag ...]
D=quantizedneuralnetworks-keras-tensorflow_amd/csrc
ROOT=$(pwd)
cp $D/libqnn_hip.so /tmp/orig.so
cd /tmp && export TMPDIR=/tmp
run() { rm -rf /tmp/tr_$1; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$1 -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /tmp/tr_$1.log 2>&1; python3 - $1 <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob("/tmp/tr_%s/**/*kernel_trace.csv" % tag, recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"][:70]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:4]:
    v = sorted(v); print(tag, k[27:70], "n=%d avg=%.1f med=%.1f min=%.1f us" % (len(v), sum(v)/len(v)/1e3, v[len(v)//2]/1e3, v[0]/1e3))
PY
}
run base
for u in "$@"; do cp $ROOT/$D/libqnn_v_$u.so $ROOT/$D/libqnn_hip.so; run $u; done
cp /tmp/orig.so $ROOT/$D/libqnn_hip.so
