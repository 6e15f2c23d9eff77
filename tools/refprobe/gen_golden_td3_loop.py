"""Golden G20c: the reference's WHOLE synchronous TD3 loop body (rl/algos/sync_td3.py:304-313 of run_experiment: parallel_collect_experience ->
ReplayBuffer.add_parallel -> TD3.train for as many iterations as transitions were collected), three rounds, run in-process under the ray
stand-in on the deterministic toy env of G15b (50-d observation, 10-d action, scripted episode lengths).

What the loop does that the recorded-batch golden G20 does not see: whole-episode collection with ONE exploration scalar per step added to
all action dimensions and clipped (:75-77), done_bool = 1 at the time limit (:82), the merge order of the workers' episodes, the ring replay
(rl/utils/remote_replay.py:66-84), uniform sampling with replacement from everything collected so far, the iteration counter that restarts
at 0 in every train call (so the delayed policy update fires on the first batch of every round), and the returned statistics (pi_loss is
divided by ALL iterations, not by the number of policy updates).

Recorded per round: the exploration draws, the merged transitions, every iteration's sample indices and smoothing noise, the returned
statistics, the live nets' parameters; at the end the target nets.  run_experiment itself is not called (its imports of apex.py / the logger
are stale against the CLI); the three calls are made in its order with its arguments."""
from common import setup_reference_path, GOLD
setup_reference_path()

import os
import sys
import types
import numpy as np
import torch

_tb = types.ModuleType("torch.utils.tensorboard"); _tb.SummaryWriter = object      # probe-only stand-in: remote_replay.py imports it at module level
sys.modules["torch.utils.tensorboard"] = _tb
_co = types.ModuleType("colorama"); _co.Fore = types.SimpleNamespace(); _co.Style = types.SimpleNamespace(); sys.modules["colorama"] = _co
import rl.algos.sync_td3 as td3mod
from rl.utils.remote_replay import ReplayBuffer
from rl.policies.actor import FF_Actor
from rl.policies.critic import Dual_Q_Critic

LENS = [7, 60, 23, 50, 3, 31]          # scripted episode lengths; >= 50 ends at max_traj_len = 50
MAX_TRAJ = 50
ROUNDS, PROCS, B, H = 3, 2, 32, 64


class ToyEnv:
    """x <- 0.9 x + 0.1 tile(a, 5) + 0.01, clock columns overwritten, reward exp(-mean|x|), done after a scripted length (the env of G15b)."""
    k = 0

    def __init__(self):
        self.observation_space = np.zeros(50); self.action_space = np.zeros(10)

    def _obs(self):
        o = self.x.copy(); o[46] = np.sin(0.2 * self.t); o[47] = np.cos(0.2 * self.t)
        return o

    def reset(self):
        ToyEnv.k += 1
        self.t = 0; self.L = LENS[(ToyEnv.k - 1) % len(LENS)]
        self.x = np.cos(np.arange(50) * 0.1 * ToyEnv.k)
        return self._obs()

    def step(self, action):
        self.t += 1
        self.x = 0.9 * self.x + 0.1 * np.tile(action, 5) + 0.01
        return self._obs(), float(np.exp(-np.abs(self.x).mean())), self.t >= self.L, {}


def main(name="g20c_td3_loop"):
    torch.manual_seed(203); np.random.seed(203)
    ToyEnv.k = 0
    algo = td3mod.TD3(50, 10, 1.0, 1e-3, 1e-3)
    algo.actor = FF_Actor(50, 10, layers=(H, H), max_action=1.0); algo.actor_target = FF_Actor(50, 10, layers=(H, H), max_action=1.0)
    algo.actor_target.load_state_dict(algo.actor.state_dict())
    algo.critic = Dual_Q_Critic(50, 10, hidden_size=H); algo.critic_target = Dual_Q_Critic(50, 10, hidden_size=H)
    algo.critic_target.load_state_dict(algo.critic.state_dict())
    algo.actor_optimizer = torch.optim.Adam(algo.actor.parameters(), lr=1e-3)
    algo.critic_optimizer = torch.optim.Adam(algo.critic.parameters(), lr=1e-3)
    out = {"hidden": H, "rounds": ROUNDS, "procs": PROCS, "batch": B, "lens": np.array(LENS), "max_traj_len": MAX_TRAJ, "lr": 1e-3, "discount": 0.99,
           "tau": 0.005, "policy_noise": 0.2, "noise_clip": 0.5, "policy_freq": 2, "act_noise": 0.3}
    out["actor_keys"] = np.array(list(algo.actor.state_dict().keys())); out["critic_keys"] = np.array(list(algo.critic.state_dict().keys()))
    for nm, net in (("actor0", algo.actor), ("critic0", algo.critic)):
        for k, v in net.state_dict().items(): out[nm + "." + k] = v.numpy().copy()

    # ---- recorders: exploration draws (numpy), sample indices (numpy), smoothing noise (torch)
    rec = {"explore": [], "idx": [], "smooth": []}
    np_normal, np_randint, t_normal = np.random.normal, np.random.randint, torch.Tensor.normal_
    def normal(*a, **k):
        v = np_normal(*a, **k); rec["explore"].append(float(np.asarray(v).reshape(-1)[0])); return v
    def randint(*a, **k):
        v = np_randint(*a, **k); rec["idx"].append(np.asarray(v).copy()); return v
    def normal_(self, *a, **k):
        r = t_normal(self, *a, **k); rec["smooth"].append(self.detach().clone().numpy()); return r
    np.random.normal, np.random.randint, torch.Tensor.normal_ = normal, randint, normal_
    # parallel_collect_experience merges the workers' lists of (obs, new_obs, action, reward, done) tuples with np.concatenate (:52): the NumPy the
    # reference was written for returned an [N, 5] OBJECT array for such ragged input, NumPy 2 raises.  Probe-only shim with the old behaviour.
    np_concat = np.concatenate
    def concatenate(seqs, *a, **k):
        try:
            return np_concat(seqs, *a, **k)
        except ValueError:
            flat = [t for s_ in seqs for t in s_]
            arr = np.empty((len(flat), 5), dtype=object)
            for i, t in enumerate(flat):
                for j in range(5): arr[i, j] = t[j]
            return arr
    np.concatenate = concatenate
    np_array = np.array                      # ReplayBuffer.sample calls np.array(x, copy=False) (remote_replay.py:83-87): NumPy 1 semantics = np.asarray
    def array(obj, *a, **k):
        if k.get("copy", True) is False:
            k.pop("copy"); return np.asarray(obj, *a, **k)
        return np_array(obj, *a, **k)
    np.array = array
    try:
        replay = ReplayBuffer()
        env_fn = lambda: ToyEnv()
        for r in range(ROUNDS):
            rec["explore"], rec["idx"], rec["smooth"] = [], [], []
            merged, T = td3mod.parallel_collect_experience(algo, env_fn, 0.3, 10000, MAX_TRAJ, num_procs=PROCS)      # sync_td3.py:304
            replay.add_parallel(merged)                                                                                    # :305
            ret = algo.train(replay, T, B, 0.99, 0.005, 0.2, 0.5, 2)                                                      # :313
            p = "r%d_" % r
            out[p + "T"] = T
            out[p + "s"] = np.array([t[0] for t in merged], np.float32); out[p + "s2"] = np.array([t[1] for t in merged], np.float32)
            out[p + "a"] = np.array([t[2] for t in merged], np.float32); out[p + "rew"] = np.array([t[3] for t in merged], np.float64)
            out[p + "d"] = np.array([t[4] for t in merged], np.float32)
            out[p + "explore"] = np.array(rec["explore"], np.float64)
            out[p + "idx"] = np.array(rec["idx"], np.int64); out[p + "smooth"] = np.array(rec["smooth"], np.float32)
            assert out[p + "idx"].shape == (T, B) and out[p + "smooth"].shape == (T, B, 10) and len(rec["explore"]) == T
            out[p + "avg_q1"] = float(ret[0]); out[p + "q_loss"] = float(ret[2]); out[p + "pi_loss"] = float(ret[3])
            for nm, net in (("actor", algo.actor), ("critic", algo.critic)):
                for k, v in net.state_dict().items(): out[p + nm + "." + k] = v.numpy().copy()
            print("round %d: %d transitions, avg_q1 %.6f q_loss %.6f pi_loss %.6f" % (r, T, out[p + "avg_q1"], out[p + "q_loss"], out[p + "pi_loss"]))
    finally:
        np.random.normal, np.random.randint, torch.Tensor.normal_ = np_normal, np_randint, t_normal
        np.concatenate = np_concat; np.array = np_array
    for nm, net in (("actor_target", algo.actor_target), ("critic_target", algo.critic_target)):
        for k, v in net.state_dict().items(): out[nm + "." + k] = v.numpy().copy()
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print("wrote", name, os.path.getsize(os.path.join(GOLD, name + ".npz")) // 1024, "KB")


if __name__ == "__main__":
    main()
