"""Golden G15a: the reference's own PPO.sample (rl/algos/ppo.py:139-186) run in-process (ray stand-in) on a deterministic
toy env with the Cassie-v0 surface.  Pins the rollout control flow: episode boundaries (traj_idx, bit-exact), the
truncation rule at max_traj_len, the bootstrap rule last_val = (not done) * V(s_T) and the discounted returns."""
from common import setup_reference_path, GOLD
setup_reference_path()

import os
import numpy as np
import torch

from rl.algos.ppo import PPO
from rl.policies.actor import Gaussian_FF_Actor
from rl.policies.critic import FF_V


class ToyEnv:
    """Deterministic env: 50-d obs, 10-d act, episodes end (done=True) after a scripted number of steps."""
    def __init__(self):
        self.observation_space = np.zeros(50); self.action_space = np.zeros(10)
        self.simrate = 50; self.k = 0
        self.lens = [7, 400, 23, 55, 3, 61, 120, 9]      # 400 > max_traj_len: ends by truncation, not done

    def reset(self):
        self.t = 0; self.L = self.lens[self.k % len(self.lens)]; self.k += 1
        self.x = np.cos(np.arange(50) * 0.1 * self.k)
        return self.x.copy()

    def step(self, action, f_term=0):
        self.t += 1
        self.x = 0.9 * self.x + 0.1 * np.tile(action, 5) + 0.01
        return self.x.copy(), float(np.exp(-np.abs(self.x).mean())), self.t >= self.L, {}


def main():
    torch.manual_seed(15); np.random.seed(15)
    args = dict(env_name="Cassie-v0", gamma=0.99, lam=0.95, lr=1e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=64,
                epochs=3, num_steps=300, max_traj_len=50, use_gae=True, num_procs=1, max_grad_norm=0.05, recurrent=False)
    algo = PPO(args, "/tmp/unused")
    policy = Gaussian_FF_Actor(50, 10, layers=(32, 32), fixed_std=np.exp(-1.5)); critic = FF_V(50, layers=(32, 32))
    policy.obs_mean, policy.obs_std = torch.zeros(50), torch.ones(50)
    critic.obs_mean, critic.obs_std = policy.obs_mean, policy.obs_std
    policy.train(); critic.train()
    env = ToyEnv()
    buf = PPO.sample.remote(algo, lambda: env, policy, critic, 300, 50)
    rewards = np.array([float(np.asarray(r).reshape(-1)[0]) for r in buf.rewards])
    returns = np.array([float(np.asarray(r).reshape(-1)[0]) for r in buf.returns])
    values = np.array([float(np.asarray(v).reshape(-1)[0]) for v in buf.values], dtype=np.float32)
    traj_idx = np.array(buf.traj_idx)
    last_vals = np.array([(returns[e - 1] - rewards[e - 1]) / 0.99 for e in traj_idx[1:]])
    np.savez(os.path.join(GOLD, "g15a_ppo_sample.npz"), rewards=rewards, returns=returns, values=values, traj_idx=traj_idx,
             ep_lens=np.array(buf.ep_lens), ep_returns=np.array(buf.ep_returns), last_vals=last_vals, gamma=0.99, max_traj_len=50,
             scripted_lens=np.array(env.lens))
    print("trajectories:", buf.ep_lens, "steps", len(rewards))


if __name__ == "__main__":
    main()
