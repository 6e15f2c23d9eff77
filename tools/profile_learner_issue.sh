#!/bin/bash
# Issue-side PMC passes of the learner GEMMs (what holds gemm_f32_128_kernel at 52-59 % MFMA-busy?), one --pmc pass per group, per-dispatch means.
# Run on the GPU box from the repo root: bash tools/profile_learner_issue.sh r05  ->  gpurun_out/prof_learner_<tag>/<tag>_learner_pmc_issue.txt
set -u
TAG=${1:-r05}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_learner_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
GROUPS_=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
 "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES"
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES"
 "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_BRANCH SQ_IFETCH SQ_WAIT_IFETCH SQ_THREAD_CYCLES_VALU"
)
SUM=$OUT/${TAG}_learner_pmc_issue.txt
echo "# rocprofv3 --pmc <group> -- python tools/t_pmc_learner.py, one pass per group (per-dispatch means)" > $SUM
i=0
for G in "${GROUPS_[@]}"; do
  i=$((i + 1))
  rocprofv3 --pmc $G --output-format csv -d $OUT/g$i -- python $ROOT/tools/t_pmc_learner.py > $OUT/g$i.log 2>&1 || true
  python $ROOT/tools/pmc_summary.py $OUT/g$i /dev/null "# group $i: $G" 2>/dev/null | grep -E "^# group|gemm_f32|mlp_fused|bwd_head|grad_reduce" >> $SUM
  rm -rf $OUT/g$i
done
cat $SUM
# HBM-side: FETCH_SIZE / WRITE_SIZE in separate passes (KB per dispatch)
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/h$C -- python $ROOT/tools/t_pmc_learner.py > $OUT/h$C.log 2>&1 || true
  python $ROOT/tools/pmc_summary.py $OUT/h$C /dev/null "# $C (KB per dispatch)" 2>/dev/null | grep -E "^# |gemm_f32|mlp_fused|bwd_head|grad_reduce" >> $SUM
  rm -rf $OUT/h$C
done
tail -20 $SUM
