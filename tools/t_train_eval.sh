#!/bin/bash
# End-to-end acceptance run on one MI355X: train Cassie-v0 PPO on the HIP env, score the saved policy (deterministic episodes),
# then the push-recovery sweep of tools/eval_perturb.py as one batch and the transfer check in the fp64 oracle.  Output under gpurun_out/train_eval/.
set -e
OUT=gpurun_out/train_eval; mkdir -p $OUT
ITR=${1:-1000}
python apex.py ppo --reward clock --n_envs 4096 --n_itr $ITR --eval_every 50 --logdir $OUT/logs --seed 0 > $OUT/train.log 2>&1
RUN=$(ls -d $OUT/logs/Cassie-v0/*/ | head -1)
tail -5 $OUT/train.log
grep -E "Return \(|Iteration" $OUT/train.log | awk 'NR % 200 < 2' > $OUT/curve.txt
python apex.py eval --path $RUN --speed 0.5 | tee $OUT/eval.txt
python apex.py eval_perturb --path $RUN --n_sizes 40 | tee $OUT/eval_perturb.txt
python tools/t_transfer.py $RUN 1.0 | tee $OUT/transfer.txt
cp $RUN/eval_perturbs.npy $RUN/actor.pt $RUN/critic.pt $RUN/experiment.info $RUN/experiment.pkl $OUT/
