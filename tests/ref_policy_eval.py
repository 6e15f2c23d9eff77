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


def _qmul(a, b):
    w1, x1, y1, z1 = a; w2, x2, y2, z2 = b
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def command_run(seed, tag="a", num_steps=200, num_commands=6, max_speed=3.0, min_speed=0.0, resample=0):
    """one iteration of the reference's command-following test (tools/test_commands.py:56-122 run_test, :131-140 the schedules): a new speed every 200 steps (previous
    +- U[0.4, 1.3], reflected into [0, 3]), half a period later a yaw command of +- U[pi/6, pi/3] applied to the policy's input, phase_add 1.5 above 1.4 m/s; failed when
    the pelvis drops below 0.4 m.  Returns the reference's row (passed, half-period kind, speed, yaw offset, last speed step, last yaw step).
    resample = R > 0: CassieEnv.step's own random command change, which the harness does not switch off - with probability 1 / R per step the commanded speed becomes
    U[min, max] (cassie.py:486-487: R = 100 today; the revision that produced the shipped tables is unknown)."""
    rng = np.random.RandomState(seed); rs = np.random.RandomState(100000 + seed)
    speeds = np.zeros(num_commands); speeds[0] = 0.5
    for i in range(num_commands - 1):
        add = rng.choice([-1, 1]) * rng.uniform(0.4, 1.3)
        if speeds[i] + add < min_speed or speeds[i] + add > max_speed:
            add = -add
        speeds[i + 1] = speeds[i] + add
    orients = rng.uniform(np.pi / 6, np.pi / 3, num_commands) * rng.choice([-1, 1], num_commands)
    act, env = policy(tag), OldCassieEnv()
    o = env.reset_for_test(); env.speed = 0.5; o = env.obs()
    phase, phase_add, count, orient_ind, speed_ind, orient_add, passed = 0.0, 1.0, 0, 0, 1, 0.0, 1
    while not (speed_ind == num_commands and orient_ind == num_commands and count == num_steps) and passed:
        if count == num_steps:
            count = 0
            env.speed = float(np.clip(speeds[speed_ind], min_speed, max_speed)); phase_add = 1.5 if env.speed > 1.4 else 1.0
            speed_ind += 1
        elif count == num_steps // 2:
            orient_add += orients[orient_ind]; orient_ind += 1
        iq = np.array([np.cos(orient_add / 2), 0.0, 0.0, -np.sin(orient_add / 2)])
        st = o.copy()
        no = _qmul(iq, st[1:5])
        st[1:5] = -no if no[0] < 0 else no
        st[15:18] = _qmul(_qmul(iq, np.array([0.0, *st[15:18]])), np.array([iq[0], -iq[1], -iq[2], -iq[3]]))[1:]
        st[46], st[47], st[48] = np.sin(2 * np.pi * phase / PHASELEN), np.cos(2 * np.pi * phase / PHASELEN), env.speed
        env.e.step_basic(np.asarray(act(st), dtype=np.float64))
        phase = 0.0 if phase + phase_add > PHASELEN else phase + phase_add
        if resample and rs.randint(resample) == 0:
            env.speed = float(rs.uniform(min_speed, max_speed))
        o = env.obs()
        passed = 0 if env.e.get("qpos")[2] < 0.4 else 1
        count += 1
    if passed:
        return [1.0, -1.0, 0.0, 0.0, 0.0, 0.0]
    return [0.0, float(count // (num_steps // 2)), env.speed, orient_add, env.speed - speeds[max(0, speed_ind - 2)], orients[orient_ind - 1]]


def command_run_300(seed):
    return command_run(seed, resample=300)


def mission_run(args):
    """one trial of the reference's "5k" stress test on the flat terrain (5k_test.py:26-70): the mission's per-step speed / yaw commands (cassie/missions/<name>/
    command_trajectory_<speed>.pkl, in the fixture) through step_basic with the shipped policy; optional floor friction and foot mass.  Returns (name, speed, passed)."""
    name, sp, fric, foot_mass = args
    from oracle import sim as S
    g = fixture()
    speeds, orients = g["mission_%s_%s_speed" % (name, sp)].astype(np.float64), g["mission_%s_%s_orient" % (name, sp)].astype(np.float64)
    act, env = policy("a"), OldCassieEnv()
    if fric is not None:
        m = env.e.get("mass")
        for b in ("left-foot", "right-foot"):
            m[S.BODY_NAMES.index(b)] = foot_mass
        env.e.set("mass", m); env.e.set("friction", [fric]); env.e.set_const()
    env.reset_for_test()
    phase = 0.0
    for i in range(len(speeds)):
        env.speed = float(np.clip(speeds[i], 0.0, 3.0))
        phase_add = 1.5 if env.speed > 1.4 else 1.0
        iq = np.array([np.cos(orients[i] / 2), 0.0, 0.0, -np.sin(orients[i] / 2)])
        st = env.obs()
        no = _qmul(iq, st[1:5])
        st[1:5] = -no if no[0] < 0 else no
        st[15:18] = _qmul(_qmul(iq, np.array([0.0, *st[15:18]])), np.array([iq[0], -iq[1], -iq[2], -iq[3]]))[1:]
        st[46], st[47], st[48] = np.sin(2 * np.pi * phase / PHASELEN), np.cos(2 * np.pi * phase / PHASELEN), env.speed
        env.e.step_basic(np.asarray(act(st), dtype=np.float64))
        phase = 0.0 if phase + phase_add > PHASELEN else phase + phase_add
        if env.e.get("qpos")[2] < 0.4:
            return name, sp, False
    return name, sp, True
