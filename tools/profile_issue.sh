#!/bin/bash
# Issue-side PMC passes of env_step_kernel (round 4): what do the non-sweep stages wait for?
# Run on the GPU box from the repo root: bash tools/profile_issue.sh r04
#   - rocprofv3 -L                                   -> gpurun_out/prof_issue_<tag>/counters.txt (names are taken from this list; absent names are skipped)
#   - one --pmc pass per counter group (no trace domains, guide: collect counters in their own run)
#   - per-dispatch means of every group in ONE file  -> gpurun_out/prof_issue_<tag>/<tag>_env_step_pmc_issue.txt (stamped with the kernel-source hash)
#   - the I-cache micro-benchmark (tools/proto/icache_probe*)  -> <tag>_icache_probe.txt
set -u
TAG=${1:-r05}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_issue_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
have() { grep -qw "$1" $OUT/counters.txt; }
GROUPS_=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
 "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU"
 "SQ_WAVE_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_IFETCH SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN"
 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"
 "SQC_TC_INST_REQ SQC_ICACHE_INPUT_VALID_READY SQC_ICACHE_INPUT_VALID_READYB SQC_ICACHE_BUSY_CYCLES"
 "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INST_LEVEL_LDS"
 "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES"
)
HASH=$(cd $ROOT && python -c "import bench; print(bench.kernel_source_hash())")
SUM=$OUT/${TAG}_env_step_pmc_issue.txt
echo "# rocprofv3 --pmc <group> -- python tools/t_pmc.py, one pass per group (per-dispatch means); kernel sources sha1: $HASH" > $SUM
i=0
for g in "${GROUPS_[@]}"; do
  sel=""; for c in $g; do if have $c; then sel="$sel $c"; else echo "# not on this device: $c" >> $SUM; fi; done
  if [ -n "$sel" ]; then
    rm -rf $OUT/p$i
    timeout 600 rocprofv3 --pmc $sel --output-format csv -d $OUT/p$i -- python $ROOT/tools/t_pmc.py > $OUT/p$i.log 2>&1 || echo "# pass failed:$sel" >> $SUM
    python $ROOT/tools/pmc_summary.py $OUT/p$i /dev/null "# group:$sel" 2>/dev/null | grep -E "^#|env_step_kernel|env_reset_kernel" >> $SUM
    rm -rf $OUT/p$i
  fi
  i=$((i+1))
done
cd $ROOT
if [ -x tools/proto/icache_probe ]; then
  (echo "# tools/proto/icache_probe (v_fma_f32, 8-byte instructions)"; timeout 300 tools/proto/icache_probe; echo "# tools/proto/icache_probe_short (v_add_f32 e32, 4-byte instructions)"; timeout 300 tools/proto/icache_probe_short) > $OUT/${TAG}_icache_probe.txt 2>&1
fi
cat $SUM
cat $OUT/${TAG}_icache_probe.txt
