"""Sampling time of one PPO rollout (32 steps x 4096 envs): the one-launch rollout (env_rollout_kernel) against the per-step launches (APX_ROLLOUT_STEPWISE=1)."""
import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from apex_amd.vecenv import CassieVecEnv
from apex_amd.ppo import PPO
env = CassieVecEnv(n_envs=4096, seed=0)
args = dict(gamma=0.99, lam=0.95, lr=1e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=16384, epochs=3, num_steps=32 * 4096, max_traj_len=400, max_grad_norm=0.05,
            mirror=True, std_dev=-1.5, seed=0)
a = PPO(args, "/tmp/apx_unused", env); a.init_networks(0); a.normalization_params(10000)
for _ in range(2): a.iteration()
torch.cuda.synchronize(); ts = []
for _ in range(5):
    o = a.iteration(); ts.append(o["sample_time"])
print("mode", "stepwise" if os.environ.get("APX_ROLLOUT_STEPWISE") == "1" else "one launch", "sample_s", ["%.4f" % t for t in ts], "done frac", float((a.b_done != 0).float().mean()), "reset_miss", int(env.get_field("reset_miss")[0, 0]) if hasattr(env, "get_field") else -1)
