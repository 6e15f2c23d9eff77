"""Golden G19 (next row f1, learner half): the reference's PPO.update_policy in RECURRENT mode (rl/algos/ppo.py:276-345 with the padded
batch of ppo.py:411-430: obs / act / ret / adv / mask as [T_max, B, .] tensors built by pad_sequence from trajectories of different
lengths), Gaussian_LSTM_Actor + LSTM_V (2 x 64 here), mirror loss on, two consecutive steps: the six scalars of every step and the
parameters afterwards."""
from common import setup_reference_path, GOLD
setup_reference_path()

import os
from copy import deepcopy
import numpy as np
import torch
import torch.optim as optim
from torch.nn.utils.rnn import pad_sequence

from rl.algos.ppo import PPO
from rl.policies.actor import Gaussian_LSTM_Actor
from rl.policies.critic import LSTM_V
from gen_golden_learner import _sym_env_fn


def main(H=64, name="g19_lstm_update", big=False):
    """big: LSTM 2 x 128 (BASELINE configs[3]); parameters from seeds, parameters after each step as slim records"""
    from golden_util import seeded_params, seeded_noise, slim
    torch.manual_seed(19)
    args = dict(env_name="Cassie-v0", gamma=0.99, lam=0.95, lr=1e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=4, epochs=1,
                num_steps=100, max_traj_len=400, use_gae=True, num_procs=1, max_grad_norm=0.05, recurrent=True)
    algo = PPO(args, save_path="/tmp/unused")
    policy = Gaussian_LSTM_Actor(50, 10, layers=(H, H), fixed_std=np.exp(-2.0)); critic = LSTM_V(50, layers=(H, H))
    g = torch.Generator().manual_seed(190)
    policy.obs_mean = torch.randn(50, generator=g) * 0.3; policy.obs_std = torch.rand(50, generator=g) + 0.5
    critic.obs_mean, critic.obs_std = policy.obs_mean, policy.obs_std
    policy.train(); critic.train()
    algo.policy, algo.critic = policy, critic
    if big:
        for net, seed in ((policy, 1901), (critic, 1902)):
            sd = net.state_dict()
            net.load_state_dict({k: torch.tensor(w) for k, w in zip(sd.keys(), seeded_params([v.shape for v in sd.values()], seed))})
    algo.old_policy = deepcopy(policy)
    with torch.no_grad():
        if big:
            for p, nz in zip(policy.state_dict().values(), seeded_noise([v.shape for v in policy.state_dict().values()], 1903, 0.02)):
                p.add_(torch.tensor(nz))
        else:
            for p in policy.parameters():
                p.add_(torch.randn(p.shape, generator=g) * 0.02)
    algo.actor_optimizer = optim.Adam(policy.parameters(), lr=args["lr"], eps=args["eps"])
    algo.critic_optimizer = optim.Adam(critic.parameters(), lr=args["lr"], eps=args["eps"])
    env = _sym_env_fn()
    out = {"hidden": H, "obs_mean": policy.obs_mean.numpy(), "obs_std": policy.obs_std.numpy(), "fixed_std": np.exp(-2.0),
           "actor_keys": np.array(list(policy.state_dict().keys())), "critic_keys": np.array(list(critic.state_dict().keys()))}
    if big:
        out["actor_seed"], out["critic_seed"], out["pert_seed"], out["pert_scale"] = 1901, 1902, 1903, 0.02      # old = seeded, actor0 = old + noise
        out["actor_shapes"] = np.array([list(v.shape) + [0] * (2 - v.dim()) for v in policy.state_dict().values()])
        out["critic_shapes"] = np.array([list(v.shape) + [0] * (2 - v.dim()) for v in critic.state_dict().values()])
    else:
        for k, v in policy.state_dict().items(): out["actor0." + k] = v.numpy().copy()
        for k, v in algo.old_policy.state_dict().items(): out["old." + k] = v.numpy().copy()
        for k, v in critic.state_dict().items(): out["critic0." + k] = v.numpy().copy()
    scal = []
    for s in range(2):
        lens = [7, 3, 11, 5] if s == 0 else [4, 9, 2, 6]
        obs_l, act_l, ret_l, adv_l = [], [], [], []
        for L in lens:
            o = torch.randn(L, 50, generator=g) * 0.6
            ph = torch.rand(L, generator=g) * 2 * np.pi
            o[:, 46] = torch.sin(ph); o[:, 47] = torch.cos(ph)
            obs_l.append(o); act_l.append(torch.randn(L, 10, generator=g) * 0.3)
            ret_l.append(torch.randn(L, 1, generator=g)); adv_l.append(torch.randn(L, 1, generator=g))
        mask_l = [torch.ones_like(r) for r in ret_l]
        obs = pad_sequence(obs_l, batch_first=False); act = pad_sequence(act_l, batch_first=False)
        ret = pad_sequence(ret_l, batch_first=False); adv = pad_sequence(adv_l, batch_first=False); mask = pad_sequence(mask_l, batch_first=False)
        p = "s%d_" % s
        out[p + "obs"] = obs.numpy(); out[p + "act"] = act.numpy(); out[p + "ret"] = ret.numpy(); out[p + "adv"] = adv.numpy()
        out[p + "mask"] = mask.numpy(); out[p + "lens"] = np.array(lens)
        scal.append([float(x) for x in algo.update_policy(obs, act, ret, adv, mask, _sym_env_fn, mirror_observation=env.mirror_clock_observation,
                                                          mirror_action=env.mirror_action)])
        for k, v in policy.state_dict().items(): out[p + "actor." + k] = slim(v.numpy()) if big else v.numpy().copy()
        for k, v in critic.state_dict().items(): out[p + "critic." + k] = slim(v.numpy()) if big else v.numpy().copy()
    out["scalars"] = np.array(scal)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print(np.array(scal))


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == "big":
        main(H=128, name="g19b_lstm_update_h128", big=True)
    else:
        main()
