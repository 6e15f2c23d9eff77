"""Golden G22 (row f4, command_profile="phase"): the reference's CassieEnv built with command_profile="phase" on the RECORDING CassieSim
stand-in of G14 (MuJoCo is absent), numpy / random draws intercepted.
  (a) state space: observation size 55 = 46 + (sin, cos, swing, stance, one-hot stance mode, speed, side speed), clock indices, mirror
      index list (cassie.py:234-271);
  (b) reset: draw order and ranges of the phase profile (cassie.py:525-563: speed, side speed, swing = randint(1, 50) / 100,
      stance = randint(1, 30) / 100, stance mode = choice of 3, start phase) and of its "library" variant (:531-539), what they produce;
  (c) create_phase_reward on phase-profile durations (very short / asymmetric swing and stance, all three stance modes);
  (d) get_full_state for command_profile="phase" on synthetic state_out_t values (cassie.py:805-808,841-859)."""
from common import setup_reference_path, GOLD
setup_reference_path()

import os
import types
import numpy as np

import cassie.cassie as cc
from cassie.phase_function import create_phase_reward
from gen_golden_dynrand import RecSim


def run_reset(reward):
    cc.CassieSim = RecSim
    env = cc.CassieEnv(dynamics_randomization=False, reward=reward, command_profile="phase", config="unused")
    env.get_full_state = lambda: np.zeros(env._obs)
    log, ctr = [], [0]
    def unit():
        ctr[0] += 1
        return (ctr[0] * 0.61803398875) % 1.0
    def fake_uniform(a=0.0, b=1.0, size=None):
        n = 1 if size is None else int(size)
        log.append((0, float(a), float(b), n))
        v = np.array([a + (b - a) * unit() for _ in range(n)])
        return float(v[0]) if size is None else v
    def fake_randint(a, b):
        log.append((1, float(a), float(b), 1))
        return a + int((b - a + 1) * unit()) if b > a else a
    def fake_choice(seq):
        log.append((2, 0.0, float(len(seq)), 1))
        return seq[int(len(seq) * unit())]
    real = (cc.np.random.uniform, cc.random.randint, cc.np.random.choice)
    cc.np.random.uniform, cc.random.randint, cc.np.random.choice = fake_uniform, fake_randint, fake_choice
    try:
        env.reset()
    finally:
        cc.np.random.uniform, cc.random.randint, cc.np.random.choice = real
    modes = ["zero", "grounded", "aerial"]
    return env, dict(draws=np.array(log, dtype=np.float64), speed=np.array([env.speed, env.side_speed]), swing_stance=np.array([env.swing_duration, env.stance_duration]),
                     stance_mode=modes.index(env.stance_mode), phase=np.array([env.phase, env.phaselen], dtype=np.float64), reward_func=env.reward_func)


def main():
    out = {}
    env, r = run_reset("clock")
    for k, v in r.items(): out["plain_" + k] = v
    out["obs_size"] = env._obs; out["clock_inds"] = np.array(env.clock_inds); out["mirrored_obs"] = np.array(env.mirrored_obs, dtype=np.float64)
    out["mirrored_acts"] = np.array(env.mirrored_acts, dtype=np.float64)
    env2, r2 = run_reset("library_clock")
    for k, v in r2.items(): out["library_" + k] = v
    # (c) clock splines for phase-profile durations
    rs = np.random.RandomState(22)
    c = 0
    modes = ["zero", "grounded", "aerial"]
    for swing100, stance100 in [(1, 1), (50, 30), (1, 30), (50, 1), (7, 19), (33, 4), (15, 25), (2, 3)] + [(int(rs.randint(1, 51)), int(rs.randint(1, 31))) for _ in range(6)]:
        for mi, mode in enumerate(modes):
            swing, stance = swing100 / 100, stance100 / 100
            left, right, phaselen = create_phase_reward(swing, stance, 0.1, mode, True, FREQ=40)
            ph = np.arange(0, int(np.floor(phaselen)) + 2).astype(float)
            vals = np.stack([left[0](ph), left[1](ph), right[0](ph), right[1](ph)], axis=1)
            out[f"c{c}_params"] = np.array([swing, stance, 0.1, mi, 1, 40]); out[f"c{c}_phaselen"] = phaselen; out[f"c{c}_phases"] = ph; out[f"c{c}_vals"] = vals
            c += 1
    out["n_cases"] = c
    # (d) get_full_state, phase profile
    rng = np.random.RandomState(23)
    class _Sim:
        def qpos(self): return np.zeros(35)
        def qvel(self): return np.zeros(32)
    n = 6
    for k in range(n):
        s = types.SimpleNamespace()
        s.sim = _Sim(); s.command_profile = "phase"; s.input_profile = "full"
        s.phase = int(rng.randint(0, 30)); s.phaselen = rng.uniform(2.0, 60.0)
        s.speed = rng.uniform(-0.3, 4); s.side_speed = rng.uniform(-0.3, 0.3); s.orient_add = rng.uniform(-1.5, 1.5)
        s.swing_duration = rng.randint(1, 51) / 100; s.stance_duration = rng.randint(1, 31) / 100; s.stance_mode = modes[k % 3]
        q = rng.randn(4); q /= np.linalg.norm(q)
        pel = types.SimpleNamespace(position=[0, 0, rng.uniform(0.5, 1.1)], orientation=list(q), rotationalVelocity=list(rng.randn(3)),
                                    translationalVelocity=list(rng.randn(3)), translationalAcceleration=list(rng.randn(3)))
        s.cassie_state = types.SimpleNamespace(pelvis=pel, terrain=types.SimpleNamespace(height=rng.uniform(-0.05, 0.05)),
                                               motor=types.SimpleNamespace(position=list(rng.randn(10)), velocity=list(rng.randn(10))),
                                               joint=types.SimpleNamespace(position=list(rng.randn(6)), velocity=list(rng.randn(6))))
        s.joint_rand = True
        s.motor_encoder_noise = rng.uniform(-0.01, 0.01, 10); s.joint_encoder_noise = rng.uniform(-0.01, 0.01, 6)
        s.history = 0; s.state_history = [np.zeros(55)]
        s.rotate_to_orient = lambda v, s=s: cc.CassieEnv.rotate_to_orient(s, v)
        obs = cc.CassieEnv.get_full_state(s)
        pre = f"o{k}_"
        out[pre + "obs"] = obs
        out[pre + "scal"] = np.array([s.phase, s.phaselen, s.speed, s.side_speed, s.orient_add, pel.position[2], s.cassie_state.terrain.height,
                                      s.swing_duration, s.stance_duration, modes.index(s.stance_mode)])
        out[pre + "quat"] = q; out[pre + "rotvel"] = np.array(pel.rotationalVelocity); out[pre + "tvel"] = np.array(pel.translationalVelocity)
        out[pre + "tacc"] = np.array(pel.translationalAcceleration)
        out[pre + "mpos"] = np.array(s.cassie_state.motor.position); out[pre + "mvel"] = np.array(s.cassie_state.motor.velocity)
        out[pre + "jpos"] = np.array(s.cassie_state.joint.position); out[pre + "jvel"] = np.array(s.cassie_state.joint.velocity)
        out[pre + "mnoise"] = s.motor_encoder_noise; out[pre + "jnoise"] = s.joint_encoder_noise
    out["n_obs_cases"] = n
    np.savez_compressed(os.path.join(GOLD, "g22_phase_profile.npz"), **out)
    print("obs size", env._obs, "clock_inds", env.clock_inds, "plain draws", r["draws"][:8].tolist(), "library draws", r2["draws"][:8].tolist())
    print("plain:", r["speed"], r["swing_stance"], r["stance_mode"], r["phase"], "| library:", r2["speed"], r2["swing_stance"], r2["stance_mode"], r2["phase"], r2["reward_func"])


if __name__ == "__main__":
    main()
