"""Golden G16: the evaluation-side env API of the reference (next row f3): CassieEnv.update_speed (cassie/cassie.py:757-775) on
random (speed, phase) states, and CassieEnv.reset_for_test (cassie.py:682-742) on a recording CassieSim stand-in (call order,
commands, clock, what is restored)."""
from common import setup_reference_path, GOLD
setup_reference_path()

import os
import numpy as np

import cassie.cassie as cc
from gen_golden_dynrand import RecSim


class OrderSim(RecSim):
    def __init__(self, *a, **k):
        super().__init__(*a, **k); self.order = []
    def set_dof_damping(self, v): self.order.append("damping"); super().set_dof_damping(v)
    def set_body_mass(self, v): self.order.append("mass"); super().set_body_mass(v)
    def set_body_ipos(self, v): self.order.append("ipos"); super().set_body_ipos(v)
    def set_geom_friction(self, v): self.order.append("friction"); super().set_geom_friction(v)
    def set_geom_quat(self, v, name=None): self.order.append("geom_quat:%s" % name); self.calls["geom_quat"] = np.array(v, dtype=np.float64)
    def set_const(self): self.order.append("set_const"); super().set_const()
    def step_pd(self, u): self.order.append("step_pd"); return super().step_pd(u)
    def full_reset(self): self.order.append("full_reset")


def main():
    cc.CassieSim = OrderSim
    rng = np.random.RandomState(16)
    n = 200
    rec = np.zeros((n, 10))
    for k in range(n):
        env = cc.CassieEnv(dynamics_randomization=True, reward="clock", config="unused")
        env.get_full_state = lambda: np.zeros(env._obs)
        # the state a training reset leaves behind (cassie.py:553-563), for a random first speed / phase
        sp0 = rng.uniform(-0.3, 4.0)
        total = (0.9 - 0.25 / 3.0 * abs(sp0)) / 2
        env.swing_duration = (0.30 + ((0.70 - 0.30) / 3) * abs(sp0)) * total
        env.stance_duration = (0.70 - ((0.70 - 0.30) / 3) * abs(sp0)) * total
        env.stance_mode = "zero"; env.strict_relaxer = 0.1; env.have_incentive = True
        env.left_clock, env.right_clock, env.phaselen = cc.create_phase_reward(env.swing_duration, env.stance_duration, env.strict_relaxer, env.stance_mode, env.have_incentive, FREQ=2000 // env.simrate)
        env.phase = int(rng.randint(0, int(np.floor(env.phaselen)) + 1))
        new_speed = float(np.float32(rng.uniform(-0.8, 4.6))); new_side = float(np.float32(rng.uniform(-0.5, 0.5)))
        ph0, pl0 = env.phase, env.phaselen
        env.update_speed(new_speed, new_side)
        rec[k] = [sp0, ph0, pl0, new_speed, new_side, env.speed, env.side_speed, env.swing_duration, env.stance_duration, env.phase]
    pl1 = np.array([rec[k, 7] * 2 + rec[k, 8] * 2 for k in range(n)]) * 40.0
    # reset_for_test on a recording sim
    env = cc.CassieEnv(dynamics_randomization=True, reward="clock", config="unused")
    env.get_full_state = lambda: np.zeros(env._obs)
    env.strict_relaxer = 0.1; env.have_incentive = True; env.stance_mode = "zero"; env.u = None
    env.speed = 2.5; env.side_speed = 0.2; env.phase = 17; env.time = 99; env.counter = 3; env.orient_add = 0.4
    env.motor_encoder_noise = np.full(10, 0.005); env.joint_encoder_noise = np.full(6, -0.004)
    env.reset_for_test()
    rft_scalars = np.array([env.phase, env.time, env.counter, env.orient_add, env.speed, env.side_speed,
                            env.swing_duration, env.stance_duration, env.phaselen, env.phase_add], dtype=np.float64)
    # step_basic bookkeeping after reset_for_test (cassie.py:498-521): 70 calls on the 32-step grounded clock
    n_pd0 = env.sim.order.count("step_pd")
    sb = []
    for k in range(70):
        env.step_basic(np.zeros(10))
        sb.append([env.time, env.phase, env.counter])
    n_pd = env.sim.order.count("step_pd") - n_pd0
    rft_order = list(env.sim.order[:env.sim.order.index("geom_quat:floor") + 1])
    out = dict(update_speed=rec, phaselen_new=pl1, step_basic=np.array(sb, dtype=np.float64), step_basic_pd_calls=np.array([n_pd]),
               rft_order=np.array(rft_order), rft_scalars_after_steps=np.array([env.speed, env.side_speed]), rft_scalars=rft_scalars,
               rft_stance_mode=np.array([env.stance_mode]), rft_damping=env.sim.calls["damping"], rft_mass=env.sim.calls["mass"],
               rft_friction=env.sim.calls["friction"], rft_floor=env.sim.calls["geom_quat"],
               rft_noise=np.concatenate([env.motor_encoder_noise, env.joint_encoder_noise]))
    # reset_for_test(full_reset=True) (tools/eval_perturb.py:31,109) with the REAL get_full_state on the reset cassie_state
    env2 = cc.CassieEnv(dynamics_randomization=True, reward="clock", config="unused")
    env2.strict_relaxer = 0.1; env2.have_incentive = True; env2.stance_mode = "zero"; env2.u = None
    env2.speed = 1.5; env2.side_speed = -0.1; env2.phase = 9; env2.time = 12; env2.counter = 2; env2.orient_add = -0.3
    env2.motor_encoder_noise = np.full(10, 0.005); env2.joint_encoder_noise = np.full(6, -0.004)
    obs_full = np.array(env2.reset_for_test(full_reset=True), dtype=np.float64)
    out["rft_full_order"] = np.array(list(env2.sim.order))
    out["rft_full_obs"] = obs_full
    out["rft_full_scalars"] = np.array([env2.phase, env2.time, env2.counter, env2.orient_add, env2.speed, env2.side_speed,
                                        env2.swing_duration, env2.stance_duration, env2.phaselen], dtype=np.float64)
    np.savez_compressed(os.path.join(GOLD, "g16_eval_api.npz"), **out)
    print("full order:", env2.sim.order); print("full obs:", obs_full)
    print("order:", env.sim.order); print("scalars:", out["rft_scalars"]); print(rec[:2])


if __name__ == "__main__":
    main()
