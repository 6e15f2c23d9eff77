#!/bin/bash
# Per-kernel time of the learner kernels: rocprofv3 --kernel-trace --stats over tools/t_pmc_learner.py (20 actor forwards at 16 384 rows + 4 PPO minibatch steps).
# Run on the GPU box from the repo root: bash tools/t_learner_kernels.sh [tag]  ->  gpurun_out/learner_kernels_<tag>.txt
TAG=${1:-x}
ROOT=$PWD
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/lk; rocprofv3 --kernel-trace --stats -d /tmp/lk -- python $ROOT/tools/t_pmc_learner.py > /tmp/lk.log 2>&1
python $ROOT/tools/rocprof_summary.py $(ls /tmp/lk/*/*.db | head -1) $ROOT/gpurun_out/learner_kernels_$TAG.txt > /dev/null
grep -E "gemm|mlp_fused|bwd_head|grad_reduce|ppo_|sumsq|adam" $ROOT/gpurun_out/learner_kernels_$TAG.txt | cut -c1-140
