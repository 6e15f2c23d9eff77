"""Golden G18 (next row f1, learner half): the reference's recurrent networks Gaussian_LSTM_Actor (rl/policies/actor.py:218-311) and
LSTM_V (rl/policies/critic.py:236-296) on a padded batch of trajectories [T, B, 50] (zero start state) and step by step with the
carried hidden state, plus the parameter gradients of a fixed scalar loss sum(w * y) through the whole sequence (BPTT by autograd)."""
from common import setup_reference_path, GOLD
setup_reference_path()

import os
import numpy as np
import torch

from rl.policies.actor import Gaussian_LSTM_Actor
from rl.policies.critic import LSTM_V


def main(H=64, name="g18_lstm", big=False):
    """big: BASELINE configs[3] size (2 x 128): parameters from golden_util.seeded_params (only the seed is stored), gradients as slim records"""
    from golden_util import seeded_params, slim
    torch.manual_seed(18)
    T, B, D = 9, 5, 50
    actor = Gaussian_LSTM_Actor(D, 10, layers=(H, H), fixed_std=np.exp(-2.0))
    critic = LSTM_V(D, layers=(H, H))
    if big:
        for net, seed in ((actor, 1801), (critic, 1802)):
            sd = net.state_dict()
            net.load_state_dict({k: torch.tensor(w) for k, w in zip(sd.keys(), seeded_params([v.shape for v in sd.values()], seed))})
    rs = np.random.RandomState(18)
    actor.obs_mean = torch.Tensor(rs.uniform(-0.2, 0.2, D)); actor.obs_std = torch.Tensor(rs.uniform(0.7, 1.4, D))
    critic.obs_mean, critic.obs_std = actor.obs_mean, actor.obs_std
    critic.train()
    x = torch.Tensor(rs.randn(T, B, D) * 0.7)
    wa = torch.Tensor(rs.randn(T, B, 10)); wc = torch.Tensor(rs.randn(T, B, 1))
    out = {"hidden": H, "x": x.numpy(), "wa": wa.numpy(), "wc": wc.numpy(), "obs_mean": actor.obs_mean.numpy(), "obs_std": actor.obs_std.numpy(),
           "actor_keys": np.array(list(actor.state_dict().keys())), "critic_keys": np.array(list(critic.state_dict().keys()))}
    if big:
        out["actor_seed"], out["critic_seed"] = 1801, 1802
        out["actor_shapes"] = np.array([list(v.shape) + [0] * (2 - v.dim()) for v in actor.state_dict().values()])
        out["critic_shapes"] = np.array([list(v.shape) + [0] * (2 - v.dim()) for v in critic.state_dict().values()])
    else:
        for k, v in actor.state_dict().items(): out["actor." + k] = v.numpy().copy()
        for k, v in critic.state_dict().items(): out["critic." + k] = v.numpy().copy()
    mu = actor(x, deterministic=True)                       # [T, B, 10], zero start state per batch (actor.py:260)
    v = critic(x)                                           # [T, B, 1], raw inputs in train mode
    (mu * wa).sum().backward(); (v * wc).sum().backward()
    out["mu"] = mu.detach().numpy(); out["v"] = v.detach().numpy()
    for k, p in actor.named_parameters(): out["actor_grad." + k] = slim(p.grad.numpy()) if big else p.grad.numpy().copy()
    for k, p in critic.named_parameters(): out["critic_grad." + k] = slim(p.grad.numpy()) if big else p.grad.numpy().copy()
    # step by step with the carried state (rollout, ppo.py:164-175): must equal the sequence pass column by column
    actor.init_hidden_state(); critic.init_hidden_state()
    with torch.no_grad():
        mu_step = torch.stack([torch.stack([actor(x[t, b], deterministic=True) for t in range(T)]) for b in [2]])   # one env, T steps
    out["mu_step_env2"] = mu_step[0].numpy()
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print("actor keys", list(actor.state_dict().keys())); print("mu", out["mu"].shape, "step vs seq", np.abs(out["mu_step_env2"] - out["mu"][:, 2]).max())


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == "big":
        main(H=128, name="g18b_lstm_h128", big=True)
    else:
        main()
