"""GPU: the reference's shipped Cassie-v0 policy (golden G24) on the batched HIP env against the two tables the reference generated with it UNDER MUJOCO.

    python tools/eval_ref_policy.py push        all 100 directions x 28 phases x 30 push sizes as ONE batch (apex_amd.eval.compute_perturbs) vs eval_perturbs.npy
    python tools/eval_ref_policy.py commands N  N random command schedules of tools/test_commands.py as one batch vs eval_commands.npy (pass rate, where the failures sit)
    python tools/eval_ref_policy.py missions    the 4 missions x 6 speeds of the "5k" stress test (5k_test.py) on the flat terrain, one env each, vs 5k_test.pkl's pass fractions

The oracle-side twins are tests/test_oracle_env.py::test_g24_* (CPU).  Open question this tool is for: at 2 - 3 m/s the oracle's robot passes 0.71 of
the command schedules, MuJoCo's 0.535 - does the kernel follow the oracle, and which modelling knob (friction, torque-speed limits, ...) moves the number?
Written at the end of round 5 with GPU access closed: not yet run."""
import math
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
PHASELEN = 27.0


class Policy49:
    """observation of the policies' Cassie-v0 revision: 46 estimator entries + clock (sin, cos of 2 pi phase / 27) + commanded speed; phase and speed per env"""

    def __init__(self, g, tag, dev, n):
        from apex_amd import engine
        self.net = engine.Mlp(49, 256, 10, dev)
        self.net.load_list([g[f"{tag}_w{i}"] for i in range(6)])
        self.mean, self.std = torch.tensor(g[f"{tag}_obs_mean"], device=dev), torch.tensor(g[f"{tag}_obs_std"], device=dev)
        self.phase = torch.zeros(n, device=dev); self.phase_add = torch.ones(n, device=dev); self.speed = torch.full((n,), 0.5, device=dev)

    def __call__(self, obs):
        c = 2.0 * math.pi * self.phase / PHASELEN
        x = torch.cat([obs[:, :46], torch.sin(c).view(-1, 1), torch.cos(c).view(-1, 1), self.speed.view(-1, 1)], 1).contiguous()
        nxt = self.phase + self.phase_add
        self.phase = torch.where(nxt > PHASELEN, torch.zeros_like(nxt), nxt)
        return self.net.forward(x, self.mean, self.std)


def main():
    from apex_amd.vecenv import CassieVecEnv
    from apex_amd import eval as E
    g = np.load(os.path.join(REPO, "tests", "golden", "g24_ref_policy_push_sweep.npz"))
    dev = torch.device("cuda:0")
    simrate, speed, wait, dur, first, incr = (float(x) for x in g["protocol"])
    make_env = lambda n: CassieVecEnv(n_envs=n, simrate=int(simrate), dynamics_randomization=False, seed=0, max_traj_len=100000)
    mode = sys.argv[1] if len(sys.argv) > 1 else "push"
    if mode == "push":
        holder = {}

        def make(n):
            holder["pol"] = Policy49(g, "a", dev, n)
            return make_env(n)
        mf, _ = E.compute_perturbs(lambda o: holder["pol"](o), make, wait_time=wait, perturb_duration=dur, perturb_size=first, perturb_incr=incr, num_angles=100, n_sizes=30, num_phases=28, speed=speed)
        mine, ref = mf.T.astype(np.float64), g["a_eval_perturbs"].astype(np.float64)
        print("kernel mean %.1f N | MuJoCo %.1f N | corr cells %.3f, direction means %.3f | mean |diff| %.1f N, max %.0f | identical %d of 2800" % (
            mine.mean(), ref.mean(), np.corrcoef(mine.ravel(), ref.ravel())[0, 1], np.corrcoef(mine.mean(1), ref.mean(1))[0, 1], np.abs(mine - ref).mean(), np.abs(mine - ref).max(), int((mine == ref).sum())))
        print("direction means kernel", np.round(mine.mean(1)[::10]), "\n                MuJoCo", np.round(ref.mean(1)[::10]))
        return
    if mode == "missions":
        names, msp = [str(x) for x in g["k5_missions"]], [float(x) for x in g["k5_mission_speeds"]]
        cells = [(m, sp) for m in names for sp in msp]
        n = 64
        L = max(len(g["mission_%s_%s_speed" % c]) for c in cells)
        spd = torch.zeros(L, n, device=dev); ori = torch.zeros(L, n, device=dev); ln = torch.zeros(n, dtype=torch.long, device=dev)
        for i, c in enumerate(cells):
            a, b = g["mission_%s_%s_speed" % c], g["mission_%s_%s_orient" % c]
            spd[:len(a), i] = torch.tensor(a, device=dev); ori[:len(b), i] = torch.tensor(b, device=dev); ln[i] = len(a)
        env = make_env(n); pol = Policy49(g, "a", dev, n)
        obs = env.reset_for_test(full_reset=True)
        fell = torch.zeros(n, dtype=torch.bool, device=dev)
        for t in range(L):
            pol.speed = spd[t].clamp(0.0, 3.0); pol.phase_add = torch.where(pol.speed > 1.4, 1.5, 1.0)
            obs = env.step_basic(pol(E._yaw_unrotate_obs(obs, ori[t])))
            fell |= (t < ln) & (env.get_field("qpos")[:, 2] < 0.4)
        ok = (~fell[:len(cells)]).cpu().numpy().reshape(len(names), len(msp))
        for m, row, ref in zip(names, ok, g["k5_flat_pass"]):
            print("%-9s kernel passes %d of 6 (%s) | MuJoCo pass fraction over 6 speeds x 19 x 19 friction x foot mass %.3f | oracle: straight 6, curvy 6, 90_left 5 (falls at 2.8), 90_right 6" % (
                m, int(row.sum()), " ".join("ok" if x else "F" for x in row), float(ref)))
        return
    # command following: apex_amd.eval.eval_commands' schedule logic with the policy's own clock / speed inputs
    n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    n = ((n_it + 63) // 64) * 64
    env = make_env(n)
    pol = Policy49(g, "a", dev, n)
    gen = torch.Generator(device=dev); gen.manual_seed(0)
    U = lambda lo, hi, *shape: lo + (hi - lo) * torch.rand(*shape, device=dev, generator=gen)
    sgn = lambda *shape: torch.where(torch.rand(*shape, device=dev, generator=gen) < 0.5, -1.0, 1.0)
    num_steps, num_commands, max_speed, min_speed = 200, 6, 3.0, 0.0
    speeds = torch.zeros(n, num_commands, device=dev); speeds[:, 0] = 0.5
    for i in range(num_commands - 1):
        add = sgn(n) * U(0.4, 1.3, n)
        nxt = speeds[:, i] + add
        add = torch.where((nxt < min_speed) | (nxt > max_speed), -add, add)
        speeds[:, i + 1] = speeds[:, i] + add
    orients = U(math.pi / 6, math.pi / 3, n, num_commands) * sgn(n, num_commands)
    obs = env.reset_for_test(full_reset=True)
    env.set_command(speed=0.5, side_speed=0.0, phase_add=1.0)
    orient_add = torch.zeros(n, device=dev)
    passed = torch.ones(n, dtype=torch.bool, device=dev)
    fail_speed = torch.zeros(n, device=dev); fail_kind = torch.zeros(n, device=dev)
    count, speed_ind, orient_ind = 0, 1, 0
    while not (speed_ind == num_commands and orient_ind == num_commands and count == num_steps):
        if count == num_steps:
            count = 0
            pol.speed = speeds[:, speed_ind].clamp(min_speed, max_speed)
            pol.phase_add = torch.where(pol.speed > 1.4, 1.5, 1.0)
            env.set_command(speed=pol.speed, phase_add=pol.phase_add)
            speed_ind += 1
        elif count == num_steps // 2:
            orient_add = orient_add + orients[:, orient_ind]
            orient_ind += 1
        obs = env.step_basic(pol(E._yaw_unrotate_obs(obs, orient_add)))
        count += 1
        fell = passed & (env.get_field("qpos")[:, 2] < 0.4)
        fail_speed = torch.where(fell, pol.speed, fail_speed); fail_kind = torch.where(fell, torch.full_like(fail_kind, float(count // (num_steps // 2))), fail_kind)
        passed &= ~fell
    p = passed[:n_it].cpu().numpy(); fs = fail_speed[:n_it].cpu().numpy()[~p]; fk = fail_kind[:n_it].cpu().numpy()[~p]
    ref = g["a_eval_commands"].astype(np.float64); fr = ref[ref[:, 0] == 0]
    bins = [0, 0.5, 1, 1.5, 2, 2.5, 3.01]
    print("pass rate kernel %.4f (n = %d) | MuJoCo %.4f (n = 10000) | oracle 0.713 (n = 240)" % (p.mean(), n_it, ref[:, 0].mean()))
    print("failures after a yaw change: kernel %.2f MuJoCo %.2f" % ((fk == 1).mean(), (fr[:, 1] == 1).mean()))
    print("failed-speed histogram share kernel", np.round(np.histogram(fs, bins=bins)[0] / max(len(fs), 1), 2), "MuJoCo", np.round(np.histogram(fr[:, 2], bins=bins)[0] / len(fr), 2))


if __name__ == "__main__":
    main()
