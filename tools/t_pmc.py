"""Env kernels for a PMC pass: three per-step launches (env_step_kernel: the substep code on its own, one env step per launch) and two one-launch rollouts
(env_rollout_kernel: 32 env steps + policy forward + restarts per launch, what the PPO headline runs)."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apex_amd.vecenv import CassieVecEnv
from apex_amd.ppo import PPO
env = CassieVecEnv(n_envs=4096, seed=0)
env.reset()
act = torch.randn(4096,10,device='cuda')*0.2
for _ in range(3): env.step(act, auto_reset=False)
torch.cuda.synchronize()
args = dict(gamma=0.99, lam=0.95, lr=1e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=16384, epochs=3, num_steps=32 * 4096, max_traj_len=400, max_grad_norm=0.05,
            mirror=True, std_dev=-1.5, seed=0)
a = PPO(args, "/tmp/apx_unused", env); a.init_networks(0)
for _ in range(2): a.sample()
torch.cuda.synchronize()
