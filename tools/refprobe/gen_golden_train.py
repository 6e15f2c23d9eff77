"""Golden G15b: the reference's WHOLE PPO.train loop (rl/algos/ppo.py:347-505: sample_parallel -> merge -> returns ->
advantage normalisation -> epochs of SubsetRandomSampler minibatches -> update_policy (mirror loss on) -> KL early stop ->
evaluation pass -> save) run in-process under the ray stand-in on a deterministic toy env with the Cassie-v0 surface
(50-d obs with a sin/cos clock in columns 46/47, 10-d act, SymmetricEnv mirror lists of cassie/cassie.py:69,244).

Recorded per iteration: the merged batch (states, actions, rewards, values, returns, traj_idx, ep_lens, ep_returns), the
deterministic means of the sampling policy (so that the action noise can be replayed), every minibatch's index list and
6-tuple, the number of epochs run, and the parameters after the iteration.  The build's PPO driver replays the same
noise / index streams on the same toy dynamics (tests/test_gpu_ppo.py) and must reproduce all of it."""
from common import setup_reference_path, GOLD, MIRRORED_OBS_FULL_CLOCK, MIRRORED_ACTS
setup_reference_path()

import os
import numpy as np
import torch

import rl.algos.ppo as refppo
from rl.algos.ppo import PPO
from rl.envs.wrappers import SymmetricEnv
from rl.policies.actor import Gaussian_FF_Actor
from rl.policies.critic import FF_V

LENS = [7, 400, 23, 50, 3, 61, 120, 9]          # scripted episode lengths; > 50 ends by truncation at max_traj_len = 50
MAX_TRAJ = 50
PERIOD = sum(min(L, MAX_TRAJ) for L in LENS)    # 242 steps: every sample() call consumes whole periods


class ToyEnv:
    """x <- 0.9 x + 0.1 tile(a, 5) + 0.01, clock columns overwritten, reward exp(-mean|x|), done after a scripted length."""
    k = 0                                        # global episode counter (training and evaluation episodes share it)
    log = []                                     # (k of every episode started)

    def __init__(self):
        self.observation_space = np.zeros(50); self.action_space = np.zeros(10)
        self.simrate = 50; self.clock_based = True; self.clock_inds = [46, 47]
        self.mirrored_obs = MIRRORED_OBS_FULL_CLOCK; self.mirrored_acts = MIRRORED_ACTS

    def _obs(self):
        o = self.x.copy(); o[46] = np.sin(0.2 * self.t); o[47] = np.cos(0.2 * self.t)
        return o

    def reset(self):
        ToyEnv.k += 1; ToyEnv.log.append(ToyEnv.k)
        self.t = 0; self.L = LENS[(ToyEnv.k - 1) % len(LENS)]
        self.x = np.cos(np.arange(50) * 0.1 * ToyEnv.k)
        return self._obs()

    def step(self, action, f_term=0):
        self.t += 1
        self.x = 0.9 * self.x + 0.1 * np.tile(action, 5) + 0.01
        return self._obs(), float(np.exp(-np.abs(self.x).mean())), self.t >= self.L, {}


class FakeLogger:
    def __init__(self, d): self.dir = d; self.rows = []
    def add_scalar(self, name, val, itr): self.rows.append((name, float(val), int(itr)))


def main(name="g15b_ppo_train", lr=1e-4, H=256, n_itr=3, seed=151, recurrent=False, mb=64, big=False):
    from golden_util import seeded_params, slim
    torch.manual_seed(seed); np.random.seed(seed)
    ToyEnv.k = 0; ToyEnv.log = []
    epochs = 3
    args = dict(env_name="Cassie-v0", gamma=0.99, lam=0.95, lr=lr, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=mb,
                epochs=epochs, num_steps=2 * PERIOD, max_traj_len=MAX_TRAJ, use_gae=True, num_procs=1, max_grad_norm=0.05,
                recurrent=recurrent)
    os.makedirs("/tmp/g15b", exist_ok=True)
    algo = PPO(args, "/tmp/g15b")
    if recurrent:
        from rl.policies.actor import Gaussian_LSTM_Actor
        from rl.policies.critic import LSTM_V
        policy = Gaussian_LSTM_Actor(50, 10, layers=(H, H), fixed_std=np.exp(-2.0)); critic = LSTM_V(50, layers=(H, H))
    else:
        policy = Gaussian_FF_Actor(50, 10, layers=(H, H), fixed_std=np.exp(-1.5)); critic = FF_V(50, layers=(H, H))
    if big:        # BASELINE-size nets: parameters from a seed (golden_util), snapshots as slim records
        for net, sd_seed in ((policy, 1501), (critic, 1502)):
            sd = net.state_dict()
            net.load_state_dict({k: torch.tensor(w) for k, w in zip(sd.keys(), seeded_params([v.shape for v in sd.values()], sd_seed))})
    rs = np.random.RandomState(7)
    policy.obs_mean = torch.Tensor(rs.uniform(-0.2, 0.2, 50)); policy.obs_std = torch.Tensor(rs.uniform(0.7, 1.4, 50))
    critic.obs_mean, critic.obs_std = policy.obs_mean, policy.obs_std
    policy.train(); critic.train()
    out = {"obs_mean": policy.obs_mean.numpy(), "obs_std": policy.obs_std.numpy(), "lens": np.array(LENS), "max_traj_len": MAX_TRAJ,
           "n_itr": n_itr, "minibatch": mb, "epochs": epochs, "gamma": 0.99, "hidden": H, "num_steps": 2 * PERIOD}
    if big:
        out["actor_seed"], out["critic_seed"] = 1501, 1502
        out["actor_keys"] = np.array(list(policy.state_dict().keys())); out["critic_keys"] = np.array(list(critic.state_dict().keys()))
        out["actor_shapes"] = np.array([list(v.shape) + [0] * (2 - v.dim()) for v in policy.state_dict().values()])
        out["critic_shapes"] = np.array([list(v.shape) + [0] * (2 - v.dim()) for v in critic.state_dict().values()])
    else:
        for k, v in policy.state_dict().items(): out["actor0." + k] = v.numpy().copy()
        for k, v in critic.state_dict().items(): out["critic0." + k] = v.numpy().copy()

    env_fn = lambda: SymmetricEnv(lambda: ToyEnv(), mirrored_obs=MIRRORED_OBS_FULL_CLOCK, mirrored_act=MIRRORED_ACTS)
    rec = {"batches": [], "idx": [], "scal": [], "calls": 0}

    orig_sp = PPO.sample_parallel
    def sample_parallel(self, env_fn_, pol, cri, min_steps, max_traj_len, deterministic=False, anneal=1.0, term_thresh=0):
        k0 = ToyEnv.k
        buf = orig_sp(self, env_fn_, pol, cri, min_steps, max_traj_len, deterministic, anneal, term_thresh)
        if not deterministic:                    # the training batch (the evaluation pass asks for deterministic=True)
            st = torch.Tensor(np.array(buf.states)).view(-1, 50)
            with torch.no_grad():
                if recurrent:      # means of the sampling policy: every trajectory from zero hidden state, like the rollout
                    ti = buf.traj_idx
                    mu = np.concatenate([pol(st[ti[j]:ti[j + 1]].unsqueeze(1), deterministic=True).numpy()[:, 0] for j in range(len(ti) - 1)])
                else:
                    mu = pol(st, deterministic=True).numpy()
            rec["batches"].append(dict(states=st.numpy(), actions=np.array(buf.actions, dtype=np.float32).reshape(-1, 10), mu=mu,
                                       rewards=np.array(buf.rewards, dtype=np.float64).reshape(-1),
                                       values=np.array(buf.values, dtype=np.float32).reshape(-1),
                                       returns=np.array(buf.returns, dtype=np.float64).reshape(-1), traj_idx=np.array(buf.traj_idx),
                                       ep_lens=np.array(buf.ep_lens), ep_returns=np.array(buf.ep_returns), k0=k0))
            rec["idx"].append([]); rec["scal"].append([])
        return buf
    PPO.sample_parallel = sample_parallel

    class RecBatchSampler(refppo.BatchSampler):
        def __iter__(self):
            ep = []
            rec["idx"][-1].append(ep)
            for b in super().__iter__():
                ep.append(list(b)); yield b
    refppo.BatchSampler = RecBatchSampler

    orig_up = PPO.update_policy
    def update_policy(self, *a, **kw):
        r = orig_up(self, *a, **kw)
        rec["scal"][-1].append([float(x) for x in r])
        return r
    PPO.update_policy = update_policy

    orig_save = PPO.save
    saves = []
    def save(self, pol, cri):
        saves.append(len(rec["batches"]) - 1); orig_save(self, pol, cri)
    PPO.save = save

    snaps = []
    logger = FakeLogger("/tmp/g15b")
    # parameters after each iteration: hook the logger's last scalar of an iteration
    orig_add = logger.add_scalar
    def add_scalar(name, val, itr):
        orig_add(name, val, itr)
        if name == "Misc/Termination Threshold":
            snaps.append(({k: v.numpy().copy() for k, v in policy.state_dict().items()},
                          {k: v.numpy().copy() for k, v in critic.state_dict().items()}))
    logger.add_scalar = add_scalar
    algo.train(env_fn, policy, critic, n_itr, logger=logger)

    for i, b in enumerate(rec["batches"]):
        p = "it%d." % i
        for k, v in b.items(): out[p + k] = v
        nb = [len(e) for e in rec["idx"][i]]
        out[p + "epochs_run"] = len(nb)
        out[p + "idx"] = np.array([sum(e, []) for e in rec["idx"][i]], dtype=np.int64)       # [epochs_run, nb * mb] (recurrent: trajectory indices)
        out[p + "scal"] = np.array(rec["scal"][i], dtype=np.float64).reshape(len(nb), -1, 6)
        for k, v in snaps[i][0].items(): out[p + "actor." + k] = slim(v) if big else v
        for k, v in snaps[i][1].items(): out[p + "critic." + k] = slim(v) if big else v
    out["saved_after_itr"] = np.array(saves)
    out["scalar_names"] = np.array(sorted({r[0] for r in logger.rows}))
    out["train_return"] = np.array([r[1] for r in logger.rows if r[0] == "Train/Return"])
    out["mean_eplen"] = np.array([r[1] for r in logger.rows if r[0] == "Train/Mean Eplen"])
    out["timesteps"] = np.array([r[1] for r in logger.rows if r[0] == "Misc/Timesteps"])
    out["lr"] = lr; out["recurrent"] = int(recurrent); out["fixed_std"] = float(policy.fixed_std)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    for i, b in enumerate(rec["batches"]):
        print("itr", i, "B", len(b["rewards"]), "episodes", list(b["ep_lens"]), "k0", b["k0"], "epochs", out["it%d.epochs_run" % i],
              "mean scal last epoch", out["it%d.scal" % i][-1].mean(0))
    print("saved after iterations", saves, "files", os.listdir("/tmp/g15b"))


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == "recurrent_big":    # G15e: as G15d with the LSTM 2 x 128 of BASELINE configs[3]
        main("g15e_ppo_train_recurrent_h128", lr=1e-4, H=128, n_itr=2, seed=154, recurrent=True, mb=4, big=True)
    elif len(sys.argv) > 1 and sys.argv[1] == "recurrent":      # G15d: the whole loop in recurrent mode (LSTM 2 x 32, minibatches of 4 trajectories)
        main("g15d_ppo_train_recurrent", lr=1e-4, H=32, n_itr=2, seed=153, recurrent=True, mb=4)
    elif len(sys.argv) > 1 and sys.argv[1] == "earlystop":      # G15c: a learning rate large enough for the KL test (ppo.py:449) to cut epochs short; 64-unit nets
        main("g15c_ppo_train_earlystop", lr=float(sys.argv[2]) if len(sys.argv) > 2 else 1.2e-2, H=64, n_itr=2, seed=152)
    else:
        main()
