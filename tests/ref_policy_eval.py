"""Replay of the reference's shipped Cassie-v0 policies and of its push sweep (tools/eval_perturb.py:97-160, test_policy.py:30-35) on the ORACLE's physics - the worker
side of tests/test_oracle_env.py::test_g24_* (fixture tests/golden/g24_ref_policy_push_sweep.npz; multiprocessing needs an importable module).

The policies were trained on the Cassie-v0 revision of their time: observation = the 46 estimator entries of today's env (cassie.py:839-850) + clock (sin, cos of
2 pi phase / phaselen) + commanded speed = 49, simrate 60, phase 0..phaselen with phaselen = 1682 // 60 - 1 = 27 (28 phases: the 28 columns of eval_perturbs.npy), PD
targets = action + the neutral offset (no_delta).  Test infrastructure only."""
import os

import numpy as np

SIMRATE, PHASELEN, NUM_ANGLES = 60, 1682 // 60 - 1, 100
_G = None


def fixture():
    global _G
    if _G is None:
        with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g24_ref_policy_push_sweep.npz")) as z:
            _G = {k: z[k] for k in z.files}      # (eagerly: the lazy NpzFile shares one file position between forked workers)
    return _G


def policy(tag):
    g = fixture()
    W = [g[f"{tag}_w{i}"].astype(np.float64) for i in range(6)]
    mean, std = g[f"{tag}_obs_mean"].astype(np.float64), g[f"{tag}_obs_std"].astype(np.float64)

    def act(obs):      # Gaussian_FF_Actor.forward(deterministic=True): the mean of Linear-ReLU-Linear-ReLU-Linear on the normalised input (rl/policies/actor.py:142-215)
        h = (np.asarray(obs, np.float32).astype(np.float64) - mean) / std
        h = np.maximum(W[0] @ h + W[1], 0.0); h = np.maximum(W[2] @ h + W[3], 0.0)
        return W[4] @ h + W[5]
    return act


class OldCassieEnv:
    def __init__(self):
        from oracle import sim as S
        self.e = S.OracleEnv(seed=0, env_id=0, dyn_rand=False, simrate=SIMRATE, max_traj_len=1000000)
        self.speed, self.phase, self.t = 0.5, 0, 0.0

    def obs(self):
        o = self.e.obs()
        return np.concatenate([o[:46], [np.sin(2 * np.pi * self.phase / PHASELEN), np.cos(2 * np.pi * self.phase / PHASELEN)], [self.speed]])

    def reset_for_test(self):
        self.e.reset_for_test(True); self.e.apply_force(np.zeros(6))
        self.phase, self.t = 0, 0.0
        return self.obs()

    def step(self, a):
        self.e.step_basic(np.asarray(a, dtype=np.float64))
        self.phase = 0 if self.phase + 1 > PHASELEN else self.phase + 1
        self.t += SIMRATE * 0.0005
        return self.obs()


def walk(tag, speed, steps=200):
    """(steps survived, final pelvis height, mean forward speed over the second half) of the policy at a commanded speed"""
    act, env = policy(tag), OldCassieEnv()
    o = env.reset_for_test(); env.speed = speed; o = env.obs()
    xs = []
    for t in range(steps):
        o = env.step(act(o))
        q = env.e.get("qpos"); xs.append(q[0])
        if q[2] < 0.4:
            break
    h = len(xs) // 2
    return len(xs), float(q[2]), float((xs[-1] - xs[h]) / ((len(xs) - h) * SIMRATE * 0.0005))


def push_cell(args):
    """the reference's sweep for one (policy, direction index, phase): the largest push survived [N] (eval_perturb.py:36-85)"""
    tag, ai, ph = args
    g = fixture()
    _, speed, wait, dur, first, incr = (float(x) for x in g["protocol"])
    act, env = policy(tag), OldCassieEnv()
    angle = -2 * np.pi * np.linspace(0, 1, NUM_ANGLES + 1)[ai]

    def reset_to_phase():
        o = env.reset_for_test(); env.speed = speed; o = env.obs()
        for _ in range(2 * (PHASELEN + 1) + ph):
            o = env.step(act(o))
        return o
    size, done = first - incr, False
    while not done and size < 600:
        size += incr
        o = reset_to_phase()
        env.e.apply_force([size * np.cos(angle), size * np.sin(angle), 0, 0, 0, 0])
        t0 = env.t
        while env.t < t0 + dur:
            o = env.step(act(o))
        env.e.apply_force(np.zeros(6))
        t0 = env.t
        while env.t < t0 + wait:
            o = env.step(act(o))
            if env.e.get("qpos")[2] < 0.4:
                done = True
                break
    return tag, ai, ph, size - incr
