"""Golden vectors G1-G5 for the learner half of the hot path.

Runs the reference's own rl/algos/ppo.py, rl/policies/*, rl/envs/wrappers.py in-process (ray stand-in)
and records inputs + outputs:
  G1 PPOBuffer.finish_path returns                      (rl/algos/ppo.py:73-89)
  G2 advantage normalisation                            (rl/algos/ppo.py:395-396)
  G3 Gaussian_FF_Actor / FF_V forward, init statistics  (rl/policies/actor.py:142-215, critic.py:37-77)
  G4 PPO.update_policy 6-tuple + post-step parameters   (rl/algos/ppo.py:276-345)
  G5 SymmetricEnv mirror matrices + clock mirroring     (rl/envs/wrappers.py:24-77)
"""
from common import setup_reference_path, GOLD, MIRRORED_OBS_FULL_CLOCK, MIRRORED_ACTS
setup_reference_path()

import os
from copy import deepcopy
import numpy as np
import torch
import torch.optim as optim

from rl.algos.ppo import PPO, PPOBuffer
from rl.policies.actor import Gaussian_FF_Actor
from rl.policies.critic import FF_V
from rl.envs.wrappers import SymmetricEnv

torch.set_num_threads(1)


# ----------------------------------------------------------------------------------------------- G1
def g1():
    rng = np.random.RandomState(1)
    out = {}
    cases = []
    for case in range(6):
        n_traj = [1, 3, 7, 12, 2, 5][case]
        gamma = [0.99, 0.99, 0.95, 0.99, 1.0, 0.9][case]
        buf = PPOBuffer(gamma, 0.95)
        lens, last_vals, rewards = [], [], []
        for _ in range(n_traj):
            T = int(rng.randint(1, 40))
            r = rng.uniform(-1, 1, size=T)
            done = bool(rng.randint(2))
            v = np.float32(rng.randn())
            for t in range(T):
                buf.store(np.zeros((1, 2)), np.zeros((1, 1)), np.array([r[t]]), np.zeros((1, 1), dtype=np.float32))
            lv = (not done) * np.array([v])
            buf.finish_path(last_val=lv)
            lens.append(T); last_vals.append(float(lv[0])); rewards.append(r)
        out[f"c{case}_gamma"] = gamma
        out[f"c{case}_lens"] = np.array(lens)
        out[f"c{case}_last_vals"] = np.array(last_vals)
        out[f"c{case}_rewards"] = np.concatenate(rewards)
        out[f"c{case}_returns"] = np.array([float(np.asarray(x).reshape(-1)[0]) for x in buf.returns])
        out[f"c{case}_traj_idx"] = np.array(buf.traj_idx)
        out[f"c{case}_ep_returns"] = np.array(buf.ep_returns)
        cases.append(case)
    out["n_cases"] = len(cases)
    np.savez(os.path.join(GOLD, "g1_finish_path.npz"), **out)


# ----------------------------------------------------------------------------------------------- G2
def g2():
    out = {}
    for i, n in enumerate([64, 1000, 4097]):
        g = torch.Generator().manual_seed(10 + i)
        returns = torch.randn(n, 1, generator=g) * 3 + 1
        values = torch.randn(n, 1, generator=g)
        adv = returns - values
        adv_n = (adv - adv.mean()) / (adv.std() + 1e-5)   # ppo.py:395-396 with eps = args.eps = 1e-5
        out[f"c{i}_returns"] = returns.numpy(); out[f"c{i}_values"] = values.numpy()
        out[f"c{i}_adv"] = adv_n.numpy()
    out["n_cases"] = 3
    np.savez(os.path.join(GOLD, "g2_adv_norm.npz"), **out)


def _params(mod):
    return {k: v.detach().numpy().copy() for k, v in mod.state_dict().items()}


# ----------------------------------------------------------------------------------------------- G3
def g3():
    torch.manual_seed(3)
    actor = Gaussian_FF_Actor(50, 10, fixed_std=np.exp(-1.5), env_name="Cassie-v0")
    critic = FF_V(50)
    g = torch.Generator().manual_seed(33)
    obs_mean = torch.randn(50, generator=g) * 0.5
    obs_std = torch.rand(50, generator=g) + 0.5
    actor.obs_mean, actor.obs_std = obs_mean, obs_std
    critic.obs_mean, critic.obs_std = obs_mean, obs_std
    # give biases non-zero values so the fixture exercises them
    with torch.no_grad():
        for p in list(actor.parameters()) + list(critic.parameters()):
            if p.dim() == 1:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    obs = torch.randn(37, 50, generator=g) * 1.5
    out = {"obs": obs.numpy(), "obs_mean": obs_mean.numpy(), "obs_std": obs_std.numpy()}
    for k, v in _params(actor).items():
        out["actor." + k] = v
    for k, v in _params(critic).items():
        out["critic." + k] = v
    out["actor_keys"] = np.array(list(actor.state_dict().keys()))
    out["critic_keys"] = np.array(list(critic.state_dict().keys()))
    with torch.no_grad():
        out["mean"] = actor(obs, deterministic=True).numpy()
        critic.train()
        out["value_train"] = critic(obs).numpy()          # training mode: NO input normalisation (critic.py:66-67)
        critic.eval()
        out["value_eval"] = critic(obs).numpy()
        critic.train()
        pdf = actor.distribution(obs)
        act = torch.randn(37, 10, generator=g) * 0.3
        out["act"] = act.numpy()
        out["logp"] = pdf.log_prob(act).sum(-1, keepdim=True).numpy()
        out["entropy"] = pdf.entropy().mean().item()
    out["fixed_std"] = float(np.exp(-1.5))
    # init statistics of normc (rows unit-norm, means layer *0.01, zero bias)
    torch.manual_seed(4)
    a2 = Gaussian_FF_Actor(50, 10, fixed_std=np.exp(-1.5))
    out["init_row_norm_l0"] = a2.actor_layers[0].weight.data.pow(2).sum(1).sqrt().numpy()
    out["init_row_norm_means"] = a2.means.weight.data.pow(2).sum(1).sqrt().numpy()
    out["init_bias_abs_max"] = max(float(p.abs().max()) for p in a2.parameters() if p.dim() == 1)
    np.savez(os.path.join(GOLD, "g3_policy_forward.npz"), **out)


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


# ----------------------------------------------------------------------------------------------- G4
def g4():
    args = dict(env_name="Cassie-v0", gamma=0.99, lam=0.95, lr=1e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2,
                minibatch_size=64, epochs=3, num_steps=5096, max_traj_len=400, use_gae=True, num_procs=2,
                max_grad_norm=0.05, recurrent=False)
    out = {}
    case = 0
    for mirror in (True, False):
        for ent in (0.0, 0.01):
            for nsteps in (1, 3):
                torch.manual_seed(100 + case)
                a = dict(args); a["entropy_coeff"] = ent
                algo = PPO(a, save_path="/tmp/unused")
                hid = 256 if case == 1 else 64     # one full-size case; the rest small to keep the fixture small
                policy = Gaussian_FF_Actor(50, 10, layers=(hid, hid), fixed_std=np.exp(-1.5), env_name="Cassie-v0")
                critic = FF_V(50, layers=(hid, hid))
                g = torch.Generator().manual_seed(200 + case)
                policy.obs_mean = torch.randn(50, generator=g) * 0.3
                policy.obs_std = torch.rand(50, generator=g) + 0.5
                critic.obs_mean, critic.obs_std = policy.obs_mean, policy.obs_std
                policy.train(); critic.train()
                algo.policy, algo.critic = policy, critic
                algo.old_policy = deepcopy(policy)
                # make new policy differ from old so ratio != 1 and the clip is exercised
                with torch.no_grad():
                    for p in policy.parameters():
                        p.add_(torch.randn(p.shape, generator=g) * 0.02)
                algo.actor_optimizer = optim.Adam(policy.parameters(), lr=a["lr"], eps=a["eps"])
                algo.critic_optimizer = optim.Adam(critic.parameters(), lr=a["lr"], eps=a["eps"])
                env_fn = _sym_env_fn
                env = env_fn()
                obs_mirr = env.mirror_clock_observation if mirror else None
                act_mirr = env.mirror_action if mirror else None
                pre = f"c{case}_"
                out[pre + "mirror"] = mirror; out[pre + "entropy_coeff"] = ent; out[pre + "nsteps"] = nsteps; out[pre + "hidden"] = hid
                out[pre + "obs_mean"] = policy.obs_mean.numpy(); out[pre + "obs_std"] = policy.obs_std.numpy()
                for k, v in _params(policy).items(): out[pre + "actor0." + k] = v
                for k, v in _params(algo.old_policy).items(): out[pre + "old." + k] = v
                for k, v in _params(critic).items(): out[pre + "critic0." + k] = v
                scal = []
                for s in range(nsteps):
                    obs = torch.randn(64, 50, generator=g)
                    # clock columns must be valid sines (arcsin in mirror_clock_observation)
                    ph = torch.rand(64, generator=g) * 2 * np.pi
                    obs[:, 46] = torch.sin(ph); obs[:, 47] = torch.cos(ph)
                    act = torch.randn(64, 10, generator=g) * 0.3
                    ret = torch.randn(64, 1, generator=g)
                    adv = torch.randn(64, 1, generator=g)
                    out[pre + f"s{s}_obs"] = obs.numpy(); out[pre + f"s{s}_act"] = act.numpy()
                    out[pre + f"s{s}_ret"] = ret.numpy(); out[pre + f"s{s}_adv"] = adv.numpy()
                    scal.append(algo.update_policy(obs, act, ret, adv, 1, env_fn, mirror_observation=obs_mirr, mirror_action=act_mirr))
                out[pre + "scalars"] = np.array(scal, dtype=np.float64)
                for k, v in _params(policy).items(): out[pre + "actor1." + k] = v
                for k, v in _params(critic).items(): out[pre + "critic1." + k] = v
                case += 1
    out["n_cases"] = case
    np.savez_compressed(os.path.join(GOLD, "g4_update_policy.npz"), **out)


# ----------------------------------------------------------------------------------------------- G5
def g5():
    env = _sym_env_fn()
    g = torch.Generator().manual_seed(5)
    obs = torch.randn(16, 50, generator=g)
    ph = torch.rand(16, generator=g) * 2 * np.pi
    obs[:, 46] = torch.sin(ph); obs[:, 47] = torch.cos(ph)
    act = torch.randn(16, 10, generator=g)
    np.savez(os.path.join(GOLD, "g5_mirror.npz"),
             mirrored_obs=np.array(MIRRORED_OBS_FULL_CLOCK), mirrored_acts=np.array(MIRRORED_ACTS),
             obs_mirror_matrix=env.obs_mirror_matrix.numpy(), act_mirror_matrix=env.act_mirror_matrix.numpy(),
             obs=obs.numpy(), act=act.numpy(),
             mirror_obs=env.mirror_clock_observation(obs.clone(), [46, 47]).numpy(),
             mirror_act=env.mirror_action(act).numpy())


if __name__ == "__main__":
    g1(); g2(); g3(); g4(); g5()
    print("wrote learner goldens to", GOLD)
