#!/bin/bash
# usage: tools/pmc_conv.sh <tag> <bench_conv args...>   -- rocprofv3 kernel trace + PMC passes over ONE conv layer
# (tools/bench_conv.py, or PROG=tools/bench_first.py for the float-input first layer), summarised per kernel into gpurun_out/pmc_<tag>/summary.txt
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/${PROG:-tools/bench_conv.py} "$@" > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
i=0
while read -r CTRS; do
  [ -z "$CTRS" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/p$i -- python3 $ROOT/${PROG:-tools/bench_conv.py} "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done <<'LIST'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU
SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
FETCH_SIZE
WRITE_SIZE
LIST
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete
python3 $ROOT/tools/pmc_sweep_summary.py $OUT > $OUT/summary.txt 2>&1
python3 $ROOT/tools/summarize_profile.py $OUT >> $OUT/summary.txt 2>&1
grep -A40 "k_conv" $OUT/summary.txt | head -${HEAD:-80}
