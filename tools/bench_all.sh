#!/bin/bash
# Run every bench workload once on the GPU box and print one line per workload (used to refresh profiles/).
for w in vgg64_full_qnn_w4a4 vgg64_full_bnn vgg_large_full_qnn_w8a8 imagenet224_resnet10_w4a4; do
  timeout -k 10 400 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_$w.json 2>gpurun_out/bench_$w.err || echo "$w FAILED"
  python3 - gpurun_out/bench_$w.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["config"]["workload"], round(d["value"]), "img/s", round(d["ms_per_step"], 4), "ms", [(k["kernel"], round(k["ms"] * 1e3, 1), round(k.get("TMACps", 0))) for k in d["kernels"]][:12])
PY
done
