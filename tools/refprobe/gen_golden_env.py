"""Golden vectors G6-G8 for the env-logic half: produced by calling the reference's own
cassie/phase_function.py, cassie/rewards/clock_rewards.py and CassieEnv.get_full_state / rotate_to_orient on
duck-typed `self` objects (CassieEnv itself cannot be constructed here: MuJoCo 2.0 is absent)."""
from common import setup_reference_path, GOLD
setup_reference_path()

import os
import types
import numpy as np

import cassie  # noqa: F401  (prints the MuJoCo load failure; harmless)
from cassie.phase_function import create_phase_reward
from cassie.rewards.clock_rewards import clock_reward, early_clock_reward, max_vel_clock_reward
from cassie.cassie import CassieEnv


def g6():
    out = {}
    c = 0
    modes = ["zero", "grounded", "aerial"]
    for speed in [-0.3, 0.0, 0.7, 1.9, 3.1, 4.0]:
        for mi, mode in enumerate(modes):
            for inc in (True, False):
                total = (0.9 - 0.25 / 3.0 * abs(speed)) / 2           # cassie.py:556-558
                swing = (0.30 + ((0.70 - 0.30) / 3) * abs(speed)) * total
                stance = (0.70 - ((0.70 - 0.30) / 3) * abs(speed)) * total
                left, right, phaselen = create_phase_reward(swing, stance, 0.1, mode, inc, FREQ=40)
                ph = np.arange(0, int(np.floor(phaselen)) + 2).astype(float)
                ph = np.concatenate([ph, ph[:-1] + 0.37])
                vals = np.stack([left[0](ph), left[1](ph), right[0](ph), right[1](ph)], axis=1)
                out[f"c{c}_params"] = np.array([swing, stance, 0.1, mi, int(inc), 40])
                out[f"c{c}_phaselen"] = phaselen; out[f"c{c}_phases"] = ph; out[f"c{c}_vals"] = vals
                c += 1
    out["n_cases"] = c
    np.savez_compressed(os.path.join(GOLD, "g6_clock_splines.npz"), **out)


class _Sim:
    def __init__(self, qpos, qvel): self._q, self._v = qpos, qvel
    def qpos(self): return self._q
    def qvel(self): return self._v


def g7():
    rng = np.random.RandomState(7)
    out = {}
    n = 24
    for c in range(n):
        speed = rng.uniform(-0.3, 4.0)
        total = (0.9 - 0.25 / 3.0 * abs(speed)) / 2
        swing = (0.30 + ((0.70 - 0.30) / 3) * abs(speed)) * total
        stance = (0.70 - ((0.70 - 0.30) / 3) * abs(speed)) * total
        left, right, phaselen = create_phase_reward(swing, stance, 0.1, "zero", True, FREQ=40)
        s = types.SimpleNamespace()
        qpos = rng.randn(35) * 0.2; qpos[2] = rng.uniform(0.6, 1.1); q = rng.randn(4); qpos[3:7] = q / np.linalg.norm(q)
        qvel = rng.randn(32)
        s.sim = _Sim(qpos, qvel)
        s.l_foot_frc, s.r_foot_frc = rng.uniform(0, 400, 2)
        s.l_foot_vel, s.r_foot_vel = rng.randn(3) * 1.5, rng.randn(3) * 1.5
        s.l_foot_orient_cost, s.r_foot_orient_cost = rng.uniform(0, 0.05, 2)
        s.speed = rng.uniform(-0.3, 4.0) if c % 2 else speed
        cs = types.SimpleNamespace(pelvis=types.SimpleNamespace(rotationalVelocity=list(rng.randn(3) * 0.3),
                                                                translationalAcceleration=list(rng.randn(3))),
                                   motor=types.SimpleNamespace(torque=list(rng.randn(10) * 20)))
        s.cassie_state = cs
        s.left_clock, s.right_clock = left, right
        s.phase = int(rng.randint(0, int(np.floor(phaselen)) + 2))
        s.prev_torque = rng.randn(10) * 20
        s.prev_action = rng.randn(10) * 0.2
        s.debug = False
        action = rng.randn(10) * 0.2
        r = clock_reward(s, action)
        r_early = early_clock_reward(s, action)
        r_maxvel = max_vel_clock_reward(s, action)
        pre = f"c{c}_"
        out[pre + "qpos"] = qpos; out[pre + "qvel"] = qvel
        out[pre + "scal"] = np.array([s.l_foot_frc, s.r_foot_frc, s.l_foot_orient_cost, s.r_foot_orient_cost, s.speed,
                                      s.phase, swing, stance])
        out[pre + "foot_vel"] = np.concatenate([s.l_foot_vel, s.r_foot_vel])
        out[pre + "rotvel"] = np.array(cs.pelvis.rotationalVelocity); out[pre + "tacc"] = np.array(cs.pelvis.translationalAcceleration)
        out[pre + "torque"] = np.array(cs.motor.torque); out[pre + "prev_torque"] = s.prev_torque
        out[pre + "prev_action"] = s.prev_action; out[pre + "action"] = action
        out[pre + "reward"] = r
        out[pre + "reward_early"] = r_early
        out[pre + "reward_max_vel"] = r_maxvel
    out["n_cases"] = n
    np.savez_compressed(os.path.join(GOLD, "g7_clock_reward.npz"), **out)


def g8():
    rng = np.random.RandomState(8)
    out = {}
    n = 12
    for c in range(n):
        s = types.SimpleNamespace()
        s.sim = _Sim(np.zeros(35), np.zeros(32))
        s.command_profile = "clock"; s.input_profile = "full"
        s.phase = int(rng.randint(0, 30)); s.phaselen = rng.uniform(22.0, 36.0)
        s.speed = rng.uniform(-0.3, 4); s.side_speed = rng.uniform(-0.3, 0.3)
        s.orient_add = rng.uniform(-1.5, 1.5) if c else 0.0
        q = rng.randn(4); q /= np.linalg.norm(q)
        pel = types.SimpleNamespace(position=[0, 0, rng.uniform(0.5, 1.1)], orientation=list(q),
                                    rotationalVelocity=list(rng.randn(3)), translationalVelocity=list(rng.randn(3)),
                                    translationalAcceleration=list(rng.randn(3)))
        s.cassie_state = types.SimpleNamespace(
            pelvis=pel, terrain=types.SimpleNamespace(height=rng.uniform(-0.05, 0.05)),
            motor=types.SimpleNamespace(position=list(rng.randn(10)), velocity=list(rng.randn(10))),
            joint=types.SimpleNamespace(position=list(rng.randn(6)), velocity=list(rng.randn(6))))
        s.joint_rand = True
        s.motor_encoder_noise = rng.uniform(-0.01, 0.01, 10); s.joint_encoder_noise = rng.uniform(-0.01, 0.01, 6)
        s.history = 0; s.state_history = [np.zeros(50)]
        s.rotate_to_orient = lambda v, s=s: CassieEnv.rotate_to_orient(s, v)
        obs = CassieEnv.get_full_state(s)
        pre = f"c{c}_"
        out[pre + "obs"] = obs
        out[pre + "scal"] = np.array([s.phase, s.phaselen, s.speed, s.side_speed, s.orient_add, pel.position[2],
                                      s.cassie_state.terrain.height])
        out[pre + "quat"] = q; out[pre + "rotvel"] = np.array(pel.rotationalVelocity)
        out[pre + "tvel"] = np.array(pel.translationalVelocity); out[pre + "tacc"] = np.array(pel.translationalAcceleration)
        out[pre + "mpos"] = np.array(s.cassie_state.motor.position); out[pre + "mvel"] = np.array(s.cassie_state.motor.velocity)
        out[pre + "jpos"] = np.array(s.cassie_state.joint.position); out[pre + "jvel"] = np.array(s.cassie_state.joint.velocity)
        out[pre + "mnoise"] = s.motor_encoder_noise; out[pre + "jnoise"] = s.joint_encoder_noise
    out["n_cases"] = n
    # input_profile "min": foot positions / orientations of the estimator instead of the 46 joint-level entries (cassie.py:246-256,829-837)
    out["n_min"] = 6
    for cp in ("clock", "phase"):
        key = "min_" + cp
        space, clock_inds, mirrored = CassieEnv.set_up_state_space(None, cp, "min")
        out[key + "_mirror"] = np.array(mirrored, dtype=np.float64); out[key + "_dim"] = len(space)
        for c in range(6):
            s = types.SimpleNamespace()
            s.command_profile = cp; s.input_profile = "min"; s.sim = _Sim(np.zeros(35), np.zeros(32))
            s.phase = int(rng.randint(0, 30)); s.phaselen = rng.uniform(22.0, 36.0)
            s.speed = rng.uniform(-0.3, 4); s.side_speed = rng.uniform(-0.3, 0.3); s.orient_add = rng.uniform(-1.5, 1.5)
            s.swing_duration = rng.uniform(0.1, 0.5); s.stance_duration = rng.uniform(0.1, 0.3); s.stance_mode = "grounded"
            q = rng.randn(4); q /= np.linalg.norm(q)
            fq = rng.randn(2, 4); fq /= np.linalg.norm(fq, axis=1, keepdims=True)
            foot = lambda k: types.SimpleNamespace(position=list(rng.randn(3)), orientation=list(fq[k]))
            lf, rf = foot(0), foot(1)
            s.cassie_state = types.SimpleNamespace(
                pelvis=types.SimpleNamespace(position=[0, 0, 1.0], orientation=list(q), rotationalVelocity=list(rng.randn(3)), translationalVelocity=list(rng.randn(3)), translationalAcceleration=list(rng.randn(3))),
                terrain=types.SimpleNamespace(height=0.0), leftFoot=lf, rightFoot=rf,
                motor=types.SimpleNamespace(position=list(rng.randn(10)), velocity=list(rng.randn(10))), joint=types.SimpleNamespace(position=list(rng.randn(6)), velocity=list(rng.randn(6))))
            s.joint_rand = False
            s.history = 0; s.state_history = [np.zeros(len(space))]
            s.rotate_to_orient = lambda v, s=s: CassieEnv.rotate_to_orient(s, v)
            pre = f"{key}{c}_"
            out[pre + "obs"] = CassieEnv.get_full_state(s)
            out[pre + "scal"] = np.array([s.phase, s.phaselen, s.speed, s.side_speed, s.orient_add, s.swing_duration, s.stance_duration])
            out[pre + "quat"] = q; out[pre + "rotvel"] = np.array(s.cassie_state.pelvis.rotationalVelocity)
            out[pre + "foot_pos"] = np.array(lf.position + rf.position); out[pre + "foot_quat"] = fq.reshape(-1)
    np.savez_compressed(os.path.join(GOLD, "g8_full_state.npz"), **out)


if __name__ == "__main__":
    g6(); g7(); g8()
    print("wrote env goldens")
