#!/bin/bash
# Run on the GPU box: several single-purpose PMC passes over bench.py (kernel-trace only beside
# them; SQ/GRBM blocks only: a TA_* pass hung the profiler on this pool), merged per kernel by tools/pmc_sweep_summary.py.  Usage: tools/pmc_sweep.sh <tag> [bench args]
TAG=${1:-sweep}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 8 --warmup 2 --no-cpu-baseline $*"
i=0
while read -r CTRS; do
  [ -z "$CTRS" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/p$i -- python3 $ROOT/bench.py $ARGS > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done <<'LIST'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS
SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VALU
LIST
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete
python3 $ROOT/tools/pmc_sweep_summary.py $OUT | tee $OUT/summary.txt
