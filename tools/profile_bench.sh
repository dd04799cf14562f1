#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + PMC passes for bench.py.
# Usage: tools/profile_bench.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# --inflight 1: kernels of two overlapped batches share the CUs, which would inflate every per-kernel
# duration; the roofline figures are per kernel, so the profile is taken with one batch in flight
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-targets --no-alternatives --inflight 1 $*"
# 1) kernel trace + stats (timing)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/bench_trace.log 2>&1 || exit 1
# 2) PMC passes (separate runs, kernel-trace only beside them)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py $ARGS > $OUT/bench_fetch.log 2>&1 || exit 2
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py $ARGS > $OUT/bench_write.log 2>&1 || exit 3
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py $ARGS > $OUT/bench_sq.log 2>&1 || exit 4
python3 $ROOT/tools/summarize_profile.py $OUT $OUT/traffic.json $OUT/bench_trace.log > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep the merged-back payload small
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete
# the per-dispatch CSVs of a 63-kernel forward run to tens of MB: the summary above is what is kept
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*_counter_collection.csv" -delete
du -sh $OUT
