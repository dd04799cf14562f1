#!/bin/bash
# One PMC pass over bench.py on the GPU box: tools/pmc_one.sh <tag> "<counters>" [bench args]
TAG=$1; CTRS=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/p1 -- python3 $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline "$@" > $OUT/p1.log 2>&1 || echo "pass failed"
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete
python3 $ROOT/tools/pmc_sweep_summary.py $OUT | tee $OUT/summary.txt
