"""Golden G17 (next row f1, env half): CassieTraj-v0 with the CLI defaults (traj=walking, command_profile=clock, input_profile=full,
no_delta) differs from Cassie-v0 only in reset: the first speed draw is random.randint(0, 40) / 10, and after set_const the pose is
overwritten with the reference trajectory's state at the random start phase (cassie/cassie_traj.py:599-778, get_ref_state :926-972).
Recorded from the reference: get_ref_state on random (phase, phaselen, speed) and the reset call order / draws on a recording CassieSim
stand-in.  tools/gen_traj_table.py turns the same trajectory file into the 34-row table the build uses."""
from common import setup_reference_path, GOLD
setup_reference_path()

import os
import types
import numpy as np

import cassie.cassie_traj as ct
from gen_golden_dynrand import RecSim


class OrderSim(RecSim):
    def __init__(self, *a, **k):
        super().__init__(*a, **k); self.order = []
    def set_const(self): self.order.append("set_const"); super().set_const()
    def set_qpos(self, q): self.order.append("set_qpos"); self.calls["qpos"] = np.array(q, dtype=np.float64)
    def set_qvel(self, v): self.order.append("set_qvel"); self.calls["qvel"] = np.array(v, dtype=np.float64)
    def step_pd(self, u): self.order.append("step_pd"); return super().step_pd(u)
    def set_geom_quat(self, v, name=None): self.calls["geom_quat"] = np.array(v, dtype=np.float64)


def main():
    ct.CassieSim = OrderSim
    env = ct.CassieTrajEnv(traj="walking", dynamics_randomization=True, reward="clock", config="unused")
    traj = env.trajectory
    rng = np.random.RandomState(17)
    n = 300
    inp = np.zeros((n, 4)); pos = np.zeros((n, 35)); vel = np.zeros((n, 32))
    for k in range(n):
        speed = rng.randint(0, 41) / 10
        total = (0.9 - 0.25 / 3.0 * abs(speed)) / 2
        swing = (0.30 + ((0.70 - 0.30) / 3) * abs(speed)) * total; stance = (0.70 - ((0.70 - 0.30) / 3) * abs(speed)) * total
        phaselen = (2 * swing + 2 * stance) * 40
        phase = int(rng.randint(0, int(np.floor(phaselen)) + 2))            # + 1 over the top: exercises the wrap to 0
        counter = int(rng.randint(0, 3)) if k % 3 == 0 else 0
        s = types.SimpleNamespace(phase=phase, phaselen=phaselen, trajectory=traj, simrate=50, aslip_traj=False, speed=speed, counter=counter)
        p, v = ct.CassieTrajEnv.get_ref_state(s, phase)
        inp[k] = [phase, phaselen, speed, counter]; pos[k] = p; vel[k] = v
    # reset on the recording sim: draws + call order
    env.get_full_state = lambda: np.zeros(env._obs)
    log = []
    import random as pyrandom
    orig_randint = pyrandom.randint
    def rec_randint(a, b):
        v = orig_randint(a, b); log.append((a, b, v)); return v
    ct.random.randint = rec_randint
    pyrandom.seed(5); np.random.seed(5)
    env.reset()
    sim = env.sim
    first_speed = log[0][2] / 10
    total = (0.9 - 0.25 / 3.0 * abs(first_speed)) / 2
    pl = (2 * (0.30 + ((0.70 - 0.30) / 3) * abs(first_speed)) * total + 2 * (0.70 - ((0.70 - 0.30) / 3) * abs(first_speed)) * total) * 40
    s = types.SimpleNamespace(phase=env.phase, phaselen=env.phaselen, trajectory=traj, simrate=50, aslip_traj=False, speed=first_speed, counter=0)
    p0, v0 = ct.CassieTrajEnv.get_ref_state(s, env.phase)
    out = dict(inp=inp, pos=pos, vel=vel, traj_len=np.array([len(traj)]), traj_dx=np.array([traj.qpos[-1, 0] - traj.qpos[0, 0]]),
               reset_order=np.array(sim.order), reset_randint=np.array(log, dtype=np.float64), reset_phase=np.array([env.phase]),
               reset_phaselen=np.array([env.phaselen, pl]), reset_qpos=sim.calls["qpos"], reset_qvel=sim.calls["qvel"],
               reset_ref_qpos=p0, reset_ref_qvel=v0, reset_speed_after=np.array([env.speed, env.side_speed, env.orient_add]),
               offset=np.array(env.offset), phaselen_init=np.array([np.floor(len(traj) / 50) - 1]))
    np.savez_compressed(os.path.join(GOLD, "g17_traj_env.npz"), **out)
    print("order:", sim.order); print("randint log:", log); print("phase", env.phase, "phaselen", env.phaselen, pl, "speed after", env.speed)
    print("qpos set == ref state:", np.array_equal(sim.calls["qpos"], p0), np.array_equal(sim.calls["qvel"], v0))


if __name__ == "__main__":
    main()
