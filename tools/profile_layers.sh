#!/bin/bash
# Run on the GPU box: rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes over tools/bench_layers.py (the layer ops
# behind the float32 surface, traffic model M0) -- the rocprof evidence for the 1-bit XNOR conv's HBM fraction.
# Usage: tools/profile_layers.sh <tag>
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_layers_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/bench_layers.py > $OUT/bench_trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/tools/bench_layers.py > $OUT/bench_fetch.log 2>&1 || exit 2
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ROOT/tools/bench_layers.py > $OUT/bench_write.log 2>&1 || exit 3
python3 $ROOT/tools/summarize_profile.py $OUT $OUT/traffic.json > $OUT/summary.txt 2>&1
cat $OUT/bench_trace.log | grep "^{" > $OUT/bench_layers.jsonl
head -40 $OUT/summary.txt
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete
