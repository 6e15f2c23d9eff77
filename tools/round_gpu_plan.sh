#!/bin/bash
# What to run, in this order, with the first GPU minutes after a round without access (each line is one gpurun call; outputs under gpurun_out/, copy what is to be judged into profiles/):
#   1. gpurun --timeout 3300 -- 'bash tools/first_gpu_run.sh r06'        the whole -m gpu suite on the final sources with the log kept (incl. tests/test_gpu_zz_first_hardware_run.py:
#                                                                         barrier stress, the two persistent trainers, G24 on the kernel) + the three bench lines
#   2. gpurun --timeout 1500 -- 'bash tools/epoch_gpu_check.sh r06'      apx_ppo_epoch / apx_td3_updates: checks, minibatch 64 / 256 lines as launches and as one launch, grid sweep, rocprofv3
#   3. gpurun --timeout 2400 -- 'bash tools/t_train_eval.sh 3000'        training on the current physics -> trained_models/r06_cassie_v0_clock, profiles/r06_training_curve.json
#   4. gpurun --timeout 1800 -- 'bash tools/profile_round.sh r06'        bench lines, rocprofv3 kernel stats, PMC passes (then add the new source hash to profiles/kernel_identity.json if needed)
# Nothing here runs by itself.
sed -n 2,8p "$0"
