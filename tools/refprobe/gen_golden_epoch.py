"""Golden vectors G4b: a whole EPOCH of small-minibatch optimiser steps of the reference's PPO on the 2 x 256 networks.

Runs the reference's own minibatch loop (rl/algos/ppo.py:414-438: BatchSampler over a sample order, update_policy per minibatch, the
reference's CLI default minibatch_size 64, apex.py:242) in-process on seeded inputs and records every step's 6-tuple plus slim records
(sum, L2 norm, strided subsample: tests/golden_util.py) of the post-epoch parameters.  The inputs are NOT stored: parameters, batch
and sample order are regenerated on the test side from the stored seeds with epoch_case_inputs() below (numpy's frozen RandomState).
Replayed by tests/test_gpu_learner.py against apx_ppo_epoch (one launch) and against the per-step apx_ppo_minibatch loop.
"""
from common import setup_reference_path, GOLD, MIRRORED_OBS_FULL_CLOCK, MIRRORED_ACTS
setup_reference_path()

import os
from copy import deepcopy
import numpy as np
import torch
import torch.optim as optim

from rl.algos.ppo import PPO
from rl.policies.actor import Gaussian_FF_Actor
from rl.policies.critic import FF_V
from rl.envs.wrappers import SymmetricEnv
from golden_util import slim, epoch_case_inputs, EPOCH_CASES

torch.set_num_threads(1)


class _FakeEnv:
    clock_based = True
    clock_inds = [46, 47]
    mirrored_obs = MIRRORED_OBS_FULL_CLOCK
    mirrored_acts = MIRRORED_ACTS
    observation_space = np.zeros(50)
    action_space = np.zeros(10)
    simrate = 50


def _sym_env_fn():
    return SymmetricEnv(lambda: _FakeEnv(), mirrored_obs=MIRRORED_OBS_FULL_CLOCK, mirrored_act=MIRRORED_ACTS)


def main():
    args = dict(env_name="Cassie-v0", gamma=0.99, lam=0.95, lr=1e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2,
                minibatch_size=64, epochs=3, num_steps=5096, max_traj_len=400, use_gae=True, num_procs=2,
                max_grad_norm=0.05, recurrent=False)
    out = {"n_cases": len(EPOCH_CASES)}
    for c, (mirror, mb, nb, adam_t0) in enumerate(EPOCH_CASES):
        inp = epoch_case_inputs(c)
        algo = PPO(dict(args), save_path="/tmp/unused")
        policy = Gaussian_FF_Actor(50, 10, layers=(256, 256), fixed_std=np.exp(-1.5), env_name="Cassie-v0")
        critic = FF_V(50, layers=(256, 256))
        old = deepcopy(policy)
        with torch.no_grad():
            for p, v in zip(policy.parameters(), inp["actor"]): p.copy_(torch.tensor(v))
            for p, v in zip(old.parameters(), inp["old"]): p.copy_(torch.tensor(v))
            for p, v in zip(critic.parameters(), inp["critic"]): p.copy_(torch.tensor(v))
        policy.obs_mean = torch.tensor(inp["obs_mean"]); policy.obs_std = torch.tensor(inp["obs_std"])
        old.obs_mean, old.obs_std = policy.obs_mean, policy.obs_std
        critic.obs_mean, critic.obs_std = policy.obs_mean, policy.obs_std
        policy.train(); critic.train()
        algo.policy, algo.critic, algo.old_policy = policy, critic, old
        algo.actor_optimizer = optim.Adam(policy.parameters(), lr=args["lr"], eps=args["eps"])
        algo.critic_optimizer = optim.Adam(critic.parameters(), lr=args["lr"], eps=args["eps"])
        # the optimiser has already taken adam_t0 - 1 steps (bias correction of a later epoch): zero moments, advanced step counter
        if adam_t0 > 1:
            for opt in (algo.actor_optimizer, algo.critic_optimizer):
                for p in opt.param_groups[0]["params"]:
                    opt.state[p] = dict(step=torch.tensor(float(adam_t0 - 1)), exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p))
        env = _sym_env_fn()
        obs_mirr = env.mirror_clock_observation if mirror else None
        act_mirr = env.mirror_action if mirror else None
        obs, act, ret, adv = (torch.tensor(inp[k]) for k in ("obs", "act", "ret", "adv"))
        perm = inp["perm"]
        scal = []
        for k in range(nb):
            idx = torch.tensor(perm[k * mb:(k + 1) * mb])
            scal.append(algo.update_policy(obs[idx], act[idx], ret[idx].view(-1, 1), adv[idx].view(-1, 1), 1, _sym_env_fn,
                                           mirror_observation=obs_mirr, mirror_action=act_mirr))
        pre = f"c{c}_"
        out[pre + "scalars"] = np.array(scal, dtype=np.float64)
        for i, p in enumerate(policy.parameters()): out[pre + f"actor1.{i}"] = slim(p.detach().numpy())
        for i, p in enumerate(critic.parameters()): out[pre + f"critic1.{i}"] = slim(p.detach().numpy())
        print("case", c, "mirror", mirror, "mb", mb, "nb", nb, "scalars[-1]", scal[-1])
    np.savez_compressed(os.path.join(GOLD, "g4b_epoch_h256.npz"), **out)
    print("wrote", os.path.join(GOLD, "g4b_epoch_h256.npz"))


if __name__ == "__main__":
    main()
