#!/bin/bash
# Profiles of one round, run on the GPU box from the repo root: bash tools/profile_round.sh r02
#   1. rocprofv3 --kernel-trace --stats of the default bench command      -> profiles/<tag>_bench_kernel_stats.txt + the bench line
#   2. SQ counters of the env kernel (own --pmc pass, no trace domains)    -> profiles/<tag>_env_step_pmc_sq.txt
#   3. FETCH_SIZE / WRITE_SIZE, two separate --pmc passes (guide: HBM)     -> profiles/<tag>_env_step_pmc_hbm.txt (+ hash of the kernel sources)
# Everything is written under gpurun_out/prof_<tag>/ (merged back by gpurun); copy the .txt / .json files into profiles/ afterwards.
set -e
TAG=${1:-r05}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -- python $ROOT/bench.py --steps 5 --warmup 1 > $OUT/bench_under_rocprof.log 2>&1 || true
grep '^{' $OUT/bench_under_rocprof.log | tail -1 > $OUT/${TAG}_bench_line_under_rocprof.json
python $ROOT/tools/rocprof_summary.py $(ls $OUT/kt/*/*.db | head -1) $OUT/${TAG}_bench_kernel_stats.txt > /dev/null
python $ROOT/tools/rocprof_timeline.py $(ls $OUT/kt/*/*.db | head -1) $OUT/${TAG}_timeline.txt > /dev/null || true      # what runs between two env steps / inside one learner minibatch
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/sq -- python $ROOT/tools/t_pmc.py > $OUT/sq.log 2>&1 || true
python $ROOT/tools/pmc_summary.py $OUT/sq $OUT/${TAG}_env_step_pmc_sq.txt "# rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -- python tools/t_pmc.py (per-dispatch means)" > /dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python $ROOT/tools/t_pmc.py > $OUT/fetch.log 2>&1 || true
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python $ROOT/tools/t_pmc.py > $OUT/write.log 2>&1 || true
mkdir -p $OUT/hbm; cp -r $OUT/fetch $OUT/hbm/; cp -r $OUT/write $OUT/hbm/
HASH=$(cd $ROOT && python -c "import bench; print(bench.kernel_source_hash())")
python $ROOT/tools/pmc_summary.py $OUT/hbm $OUT/${TAG}_env_step_pmc_hbm.txt "# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, KB per dispatch) -- python tools/t_pmc.py; kernel sources sha1: $HASH" > /dev/null
cd $ROOT
cp $OUT/${TAG}_env_step_pmc_hbm.txt $OUT/${TAG}_env_step_pmc_sq.txt $ROOT/profiles/ 2>/dev/null || true      # bench.py reads roofline.traffic from profiles/<tag>_env_step_pmc_hbm.txt (same kernel-source hash): the lines below carry it
python bench.py --steps 20 --warmup 2 2>/dev/null | tail -1 > $OUT/${TAG}_bench_line.json
python bench.py --workload cassietraj_recurrent --steps 3 --warmup 1 --no_cpu_baseline 2>/dev/null | tail -1 > $OUT/${TAG}_bench_line_recurrent.json
#   3b. the parity-run minibatch sizes (64 = the reference's CLI default apex.py:242, 2048 = the shipped run's) and the per-step-launch rollout for comparison
python bench.py --steps 3 --warmup 1 --no_cpu_baseline --minibatch 2048 2>/dev/null | tail -1 > $OUT/${TAG}_bench_line_mb2048.json || true
timeout 900 python bench.py --steps 2 --warmup 1 --no_cpu_baseline --minibatch 64 2>/dev/null | tail -1 > $OUT/${TAG}_bench_line_mb64.json || true
APX_ROLLOUT_STEPWISE=1 python bench.py --steps 10 --warmup 2 --no_cpu_baseline 2>/dev/null | tail -1 > $OUT/${TAG}_bench_line_stepwise_rollout.json || true
#   4. per-kernel time of the recurrent workload (persistent LSTM layer kernels)          -> profiles/<tag>_recurrent_kernel_stats.txt
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/ktr -- python $ROOT/bench.py --workload cassietraj_recurrent --steps 2 --warmup 1 --no_cpu_baseline > $OUT/rec_under_rocprof.log 2>&1 || true)
python $ROOT/tools/rocprof_summary.py $(ls $OUT/ktr/*/*.db | head -1) $OUT/${TAG}_recurrent_kernel_stats.txt > /dev/null || true
python $ROOT/tools/rocprof_timeline.py $(ls $OUT/ktr/*/*.db | head -1) $OUT/${TAG}_recurrent_timeline.txt > /dev/null || true      # one rollout step / one recurrent minibatch, kernel by kernel
rm -rf $OUT/ktr
#   4b. MFMA pipe counters of the learner kernels (fused fp32 forward, gemm_f32_kernel<*>, gemm_f32_128_kernel<*>), own --pmc passes  -> profiles/<tag>_learner_pmc_mfma.txt
(cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $OUT/mfma -- python $ROOT/tools/t_pmc_learner.py > $OUT/mfma.log 2>&1 || true)
python $ROOT/tools/pmc_summary.py $OUT/mfma $OUT/${TAG}_learner_pmc_mfma.txt "# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 -- python tools/t_pmc_learner.py (per-dispatch means; MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES)" > /dev/null || true
rm -rf $OUT/mfma
#   4c. TD3 workload: bench line + per-kernel time  -> profiles/<tag>_bench_line_td3.json, <tag>_td3_kernel_stats.txt
python bench.py --workload cassie_td3 --steps 3 --warmup 1 2>/dev/null | tail -1 > $OUT/${TAG}_bench_line_td3.json || true
python bench.py --workload cassie_td3 --td3_async --steps 3 --warmup 1 2>/dev/null | tail -1 > $OUT/${TAG}_bench_line_td3_async.json || true
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/ktt -- python $ROOT/bench.py --workload cassie_td3 --steps 2 --warmup 1 > $OUT/td3_under_rocprof.log 2>&1 || true)
python $ROOT/tools/rocprof_summary.py $(ls $OUT/ktt/*/*.db | head -1) $OUT/${TAG}_td3_kernel_stats.txt > /dev/null || true
rm -rf $OUT/ktt
#   4d. the RCCL path at world_size 1 (APX_FORCE_DIST=1 through torch.distributed.run): init_process_group("nccl"), gradient / moment / scalar all-reduces on device tensors  -> profiles/<tag>_bench_line_nccl_ws1.json
APX_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 1 --steps 5 --warmup 1 --no_cpu_baseline 2>$OUT/nccl_ws1.err | grep '^{' | tail -1 > $OUT/${TAG}_bench_line_nccl_ws1.json || true
#   4e. issue-side counters of the env kernel + the I-cache probe (tools/profile_issue.sh)  -> profiles/<tag>_env_step_pmc_issue.txt, <tag>_icache_probe.txt
bash tools/profile_issue.sh $TAG > $OUT/issue.log 2>&1 || true
cp gpurun_out/prof_issue_$TAG/${TAG}_env_step_pmc_issue.txt gpurun_out/prof_issue_$TAG/${TAG}_icache_probe.txt $OUT/ 2>/dev/null || true
#   5. -ffast-math A/B of the env kernel (needs lib/libapx_nofm.so: make -C apex_amd/csrc VARIANT=nofm FASTMATH=)  -> profiles/<tag>_fastmath_ab.json
if [ -f apex_amd/lib/libapx_nofm.so ]; then python tools/t_fastmath_ab.py > $OUT/${TAG}_fastmath_ab.json 2>$OUT/fastmath_ab.err || true; fi
#   6. stage profile of the env kernel (shader-clock probes; needs lib/libapx_prof.so and libapx_prof2.so)  -> profiles/<tag>_stage_profile.txt
if [ -f apex_amd/lib/libapx_prof.so ]; then
  { echo "# stage profile (APX_LIB=apex_amd/lib/libapx_prof.so / libapx_prof2.so python tools/t_prof.py), kernel sources sha1 $HASH"; APX_LIB=$ROOT/apex_amd/lib/libapx_prof.so python tools/t_prof.py 2>/dev/null;
    echo "# fine probes (-DAPX_PROF=2)"; APX_LIB=$ROOT/apex_amd/lib/libapx_prof2.so python tools/t_prof.py 2>/dev/null; } > $OUT/${TAG}_stage_profile.txt || true
fi
rm -rf $OUT/kt $OUT/sq $OUT/fetch $OUT/write $OUT/hbm
ls -la $OUT
