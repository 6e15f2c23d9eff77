"""Golden G20 (next row f2): the reference's TD3.train (rl/algos/sync_td3.py:133-209) for 4 iterations (two delayed policy updates) on
recorded replay batches: FF_Actor (tanh head) + Dual_Q_Critic (64-unit nets here), target-policy smoothing with the recorded noise,
clipped double-Q target, Adam steps, Polyak averaging.  Recorded: batches, noises, returned statistics, all four parameter sets after."""
from common import setup_reference_path, GOLD
setup_reference_path()

import os
import numpy as np
import torch

import sys, types
_tb = types.ModuleType("torch.utils.tensorboard"); _tb.SummaryWriter = object      # probe-only stand-in: remote_replay.py imports it at module level
sys.modules["torch.utils.tensorboard"] = _tb
_co = types.ModuleType("colorama"); _co.Fore = types.SimpleNamespace(); _co.Style = types.SimpleNamespace(); sys.modules["colorama"] = _co      # same, colour codes for log lines
import rl.algos.sync_td3 as td3mod
from rl.policies.actor import FF_Actor
from rl.policies.critic import Dual_Q_Critic


class FixedReplay:
    def __init__(self, batches): self.b = batches; self.i = 0
    def sample(self, n):
        x = self.b[self.i]; self.i += 1
        return x


def main(H=64, name="g20_td3", big=False):
    """big: the 256-unit nets of BASELINE configs[4]; parameters from seeds (golden_util), parameter sets afterwards as slim records"""
    from golden_util import seeded_params, seeded_noise, slim
    torch.manual_seed(20)
    B, iters = 64, 4
    algo = td3mod.TD3(50, 10, 1.0, 1e-3, 1e-3)
    algo.actor = FF_Actor(50, 10, layers=(H, H), max_action=1.0); algo.actor_target = FF_Actor(50, 10, layers=(H, H), max_action=1.0)
    algo.actor_target.load_state_dict(algo.actor.state_dict())
    algo.critic = Dual_Q_Critic(50, 10, hidden_size=H); algo.critic_target = Dual_Q_Critic(50, 10, hidden_size=H)
    algo.critic_target.load_state_dict(algo.critic.state_dict())
    if big:
        for net, seed in ((algo.actor, 2001), (algo.critic, 2002)):
            sd = net.state_dict()
            net.load_state_dict({k: torch.tensor(w) for k, w in zip(sd.keys(), seeded_params([v.shape for v in sd.values()], seed))})
        algo.actor_target.load_state_dict(algo.actor.state_dict()); algo.critic_target.load_state_dict(algo.critic.state_dict())
        with torch.no_grad():
            for net, seed in ((algo.actor_target, 2003), (algo.critic_target, 2004)):
                for p, nz in zip(net.state_dict().values(), seeded_noise([v.shape for v in net.state_dict().values()], seed, 0.01)):
                    p.add_(torch.tensor(nz))
    else:
        with torch.no_grad():           # make the targets differ from the live nets so that Polyak averaging is visible
            for p in list(algo.actor_target.parameters()) + list(algo.critic_target.parameters()):
                p.add_(torch.randn(p.shape) * 0.01)
    algo.actor_optimizer = torch.optim.Adam(algo.actor.parameters(), lr=1e-3)
    algo.critic_optimizer = torch.optim.Adam(algo.critic.parameters(), lr=1e-3)
    rs = np.random.RandomState(20)
    batches = []
    for _ in range(iters):
        x = rs.randn(B, 50) * 0.6; y = x + rs.randn(B, 50) * 0.1; u = np.tanh(rs.randn(B, 10)); r = rs.rand(B, 1); d = (rs.rand(B, 1) < 0.1).astype(np.float64)
        batches.append((x, y, u, r, d))
    out = {"hidden": H, "iters": iters, "lr": 1e-3, "discount": 0.99, "tau": 0.005, "policy_noise": 0.2, "noise_clip": 0.5, "policy_freq": 2}
    if big:
        out["seeds"] = np.array([2001, 2002, 2003, 2004]); out["target_noise"] = 0.01       # actor, critic, target perturbations
        out["actor_shapes"] = np.array([list(v.shape) + [0] * (2 - v.dim()) for v in algo.actor.state_dict().values()])
        out["critic_shapes"] = np.array([list(v.shape) + [0] * (2 - v.dim()) for v in algo.critic.state_dict().values()])
    else:
        for nm, net in (("actor0", algo.actor), ("actor_target0", algo.actor_target), ("critic0", algo.critic), ("critic_target0", algo.critic_target)):
            for k, v in net.state_dict().items(): out[nm + "." + k] = v.numpy().copy()
    out["actor_keys"] = np.array(list(algo.actor.state_dict().keys())); out["critic_keys"] = np.array(list(algo.critic.state_dict().keys()))
    for i, (x, y, u, r, d) in enumerate(batches):
        out["b%d_x" % i] = x.astype(np.float32); out["b%d_y" % i] = y.astype(np.float32); out["b%d_u" % i] = u.astype(np.float32)
        out["b%d_r" % i] = r.astype(np.float32); out["b%d_d" % i] = d.astype(np.float32)
    # the only RNG consumer inside train() is the smoothing noise: replay the generator to record it
    torch.manual_seed(2020)
    for i in range(iters):
        out["b%d_noise" % i] = torch.FloatTensor(batches[i][2]).data.normal_(0, 0.2).numpy().copy()
    torch.manual_seed(2020)
    ret = algo.train(FixedReplay(batches), iters, batch_size=B, discount=0.99, tau=0.005, policy_noise=0.2, noise_clip=0.5, policy_freq=2)
    out["ret_avg_q1"] = float(ret[0]); out["ret_q_loss"] = float(ret[2]); out["ret_pi_loss"] = float(ret[3])
    for nm, net in (("actor1", algo.actor), ("actor_target1", algo.actor_target), ("critic1", algo.critic), ("critic_target1", algo.critic_target)):
        for k, v in net.state_dict().items(): out[nm + "." + k] = slim(v.numpy()) if big else v.numpy().copy()
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print("avg_q1 %.6f q_loss %.6f pi_loss %.6f" % (out["ret_avg_q1"], out["ret_q_loss"], out["ret_pi_loss"]))


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == "big":
        main(H=256, name="g20b_td3_h256", big=True)
    else:
        main()
