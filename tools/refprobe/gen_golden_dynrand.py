"""Golden G14: the reference's CassieEnv.__init__ + CassieEnv.reset (cassie/cassie.py:27-180, 523-680) run in-process on a
RECORDING stand-in for CassieSim (MuJoCo is absent), with numpy / random draws intercepted.  Pins, independently of the
physics: the ORDER of the random draws of a reset, the range of every draw as a function of the model defaults
(dynamics-randomisation tables cassie.py:569-632, slope / encoder noise :645-657, command / phase draws :525-563, :667-669)
and what is handed to the simulator (clipping, friction triple replication, floor quaternion)."""
from common import setup_reference_path, GOLD
setup_reference_path()

import os
import random
import numpy as np

import cassie.cassie as cc

NV, NB, NG = 32, 26, 9


class RecSim:
    """Stand-in for cassiemujoco.CassieSim: fake model defaults out, every setter recorded."""
    def __init__(self, *a, **k):
        self.calls = {}
        self.damp = 0.5 + 0.25 * np.arange(NV)
        self.mass = np.concatenate([[0.0], 1.0 + 0.5 * np.arange(NB - 1)])
        self.ipos = 0.01 * np.arange(3 * NB)
        self.fric = np.tile([1.0, 5e-3, 1e-4], NG)
        self.quat = np.tile([1.0, 0.0, 0.0, 0.0], NG)
    def qpos(self): return [0.0, 0.0, 1.01, 1.0, 0.0, 0.0, 0.0] + [0.0] * 28
    def qvel(self): return [0.0] * 32
    def get_dof_damping(self): return self.damp.copy()
    def get_body_mass(self): return self.mass.copy()
    def get_body_ipos(self): return self.ipos.copy()
    def get_geom_friction(self): return self.fric.copy()
    def get_geom_rgba(self): return np.ones(4 * NG)
    def get_geom_quat(self): return self.quat.copy()
    def set_dof_damping(self, v): self.calls["damping"] = np.array(v, dtype=np.float64)
    def set_body_mass(self, v): self.calls["mass"] = np.array(v, dtype=np.float64)
    def set_body_ipos(self, v): self.calls["ipos"] = np.array(v, dtype=np.float64)
    def set_geom_friction(self, v): self.calls["friction"] = np.array(v, dtype=np.float64)
    def set_geom_quat(self, v): self.calls["geom_quat"] = np.array(v, dtype=np.float64)
    def set_const(self): self.calls["set_const"] = self.calls.get("set_const", 0) + 1
    def step_pd(self, u): self.calls["step_pd"] = self.calls.get("step_pd", 0) + 1; return None
    def full_reset(self): pass


def main():
    cc.CassieSim = RecSim
    env = cc.CassieEnv(dynamics_randomization=True, reward="clock", config="unused")
    env.get_full_state = lambda: np.zeros(env._obs)
    log = []                      # (kind, lo, hi, count): kind 0 = np.random.uniform, 1 = random.randint (inclusive)
    ctr = [0]
    def unit():                   # deterministic "uniform" stream so that the recorded outputs can be re-derived
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
    cc.np.random.uniform = fake_uniform
    cc.random.randint = fake_randint
    env.reset()
    sim = env.sim
    out = dict(draws=np.array(log, dtype=np.float64), default_damping=sim.damp, default_mass=sim.mass, default_fric=sim.fric,
               default_quat=sim.quat, set_damping=sim.calls["damping"], set_mass=sim.calls["mass"], set_ipos=sim.calls["ipos"],
               set_friction=sim.calls["friction"], set_geom_quat=sim.calls["geom_quat"],
               motor_noise=np.array(env.motor_encoder_noise), joint_noise=np.array(env.joint_encoder_noise),
               speed=np.array([env.speed, env.side_speed, env.orient_add]), phase=np.array([env.phase, env.phaselen, env.time, env.counter], dtype=np.float64),
               swing_stance=np.array([env.swing_duration, env.stance_duration]),
               n_set_const=np.array([sim.calls["set_const"]]), n_step_pd=np.array([sim.calls.get("step_pd", 0)]),
               consts=np.array([env.damping_low, env.damping_high, env.mass_low, env.mass_high, env.fric_low, env.fric_high,
                                env.max_roll_incline, env.max_pitch_incline, env.encoder_noise, env.min_speed, env.max_speed,
                                env.min_side_speed, env.max_side_speed]))
    np.savez_compressed(os.path.join(GOLD, "g14_dynrand.npz"), **out)
    print("draws:", len(log), "scalars:", int(sum(d[3] for d in log)))
    for d in log[:6]: print(d)


if __name__ == "__main__":
    main()
