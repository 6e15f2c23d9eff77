"""Golden vectors G9 / G10: random I/O of the reference's own native blocks pd_input_step and
cassie_core_sim_step (libcassiemujoco.so, called through the reference's ctypes module; no MuJoCo involved)."""
import os
import numpy as np
from native_blocks import cm, make_out, set_motor, core_step, new_core, DRIVES
from common import GOLD


def g9():
    rng = np.random.RandomState(9)
    pd = cm.pd_input_alloc(); cm.pd_input_setup(pd)
    n = 64
    P = rng.uniform(0, 150, (n, 10)); D = rng.uniform(0, 15, (n, 10)); ff = rng.randn(n, 10) * 5
    pt = rng.randn(n, 10); dt = rng.randn(n, 10); q = rng.randn(n, 10); v = rng.randn(n, 10) * 3
    tau = np.zeros((n, 10))
    for k in range(n):
        u = cm.pd_in_t(); out = make_out(); uin = cm.cassie_user_in_t()
        for i in range(10):
            leg = u.leftLeg if i < 5 else u.rightLeg
            leg.motorPd.pGain[i % 5] = P[k, i]; leg.motorPd.dGain[i % 5] = D[k, i]; leg.motorPd.torque[i % 5] = ff[k, i]
            leg.motorPd.pTarget[i % 5] = pt[k, i]; leg.motorPd.dTarget[i % 5] = dt[k, i]
            set_motor(out, i, pos=q[k, i], vel=v[k, i])
        cm.pd_input_step(pd, u, out, uin)
        tau[k] = [uin.torque[i] for i in range(10)]
    np.savez(os.path.join(GOLD, "g9_pd_input.npz"), P=P, D=D, ff=ff, pTarget=pt, dTarget=dt, q=q, v=v, tau=tau)


def g10():
    rng = np.random.RandomState(10)
    n = 400
    # soft zones start 0.15 rad inside the drive limits [-15,20] [-22,22] [-50,80] [-156,-42] [-140,-35] deg (probe_safety*.py)
    lo = np.deg2rad([-15, -22, -50, -156, -140.0]) + 0.15; hi = np.deg2rad([20, 22, 80, -42, -35.0]) - 0.15
    q = np.zeros((n, 10)); v = rng.randn(n, 10) * 2; cmd = rng.randn(n, 10) * 60; radio = np.ones(n); tau = np.zeros((n, 10))
    for k in range(n):
        for i in range(10):
            j = i % 5
            lo_i, hi_i = (lo[j], hi[j]) if (i < 5 or j >= 2) else (-hi[j], -lo[j])      # roll / yaw mirror on the right leg
            mode = rng.randint(4)
            if mode == 0: q[k, i] = rng.uniform(lo_i, hi_i)                              # free interval
            elif mode == 1: q[k, i] = hi_i + rng.uniform(0, 0.25)                        # past the upper soft limit
            elif mode == 2: q[k, i] = lo_i - rng.uniform(0, 0.25)
            else: q[k, i] = rng.uniform(lo_i, hi_i) if rng.rand() < 0.8 else hi_i + rng.uniform(0, 0.02)
        for off in (0, 5):      # stay out of the coupled hip-pitch + knee zone (pitch + knee < ~-2.36), which is NOT modelled
            while q[k, off + 2] + q[k, off + 3] < -2.2:
                q[k, off + 2] = rng.uniform(lo[2], hi[2]); q[k, off + 3] = rng.uniform(lo[3], hi[3])
        if k % 50 == 49: radio[k] = rng.choice([0.0, -1.0])
        out = make_out(); out.pelvis.radio.channel[8] = radio[k]
        for i in range(10): set_motor(out, i, pos=q[k, i], vel=v[k, i])
        tau[k] = core_step(new_core(), out, cmd[k])
    np.savez(os.path.join(GOLD, "g10_core_sim.npz"), q=q, v=v, cmd=cmd, radio=radio, tau=tau, lo=lo, hi=hi)


def g10b():
    """Coupled zone of cassie_core_sim_step: hip pitch + knee below -135 deg (probe_safety5.py).  Random states around and inside
    it (both legs, with and without the single-joint zones active at the same time), random velocities and commands."""
    rng = np.random.RandomState(1010)
    n = 300
    lo = np.deg2rad([-15, -22, -50, -156, -140.0]) + 0.15; hi = np.deg2rad([20, 22, 80, -42, -35.0]) - 0.15
    q = np.zeros((n, 10)); v = rng.randn(n, 10) * 2; cmd = rng.randn(n, 10) * 60; tau = np.zeros((n, 10))
    T = -0.75 * np.pi
    for k in range(n):
        for i in range(10):
            j = i % 5
            lo_i, hi_i = (lo[j], hi[j]) if (i < 5 or j >= 2) else (-hi[j], -lo[j])
            q[k, i] = rng.uniform(lo_i, hi_i) if rng.rand() < 0.85 else (hi_i + rng.uniform(0, 0.1) if rng.rand() < 0.5 else lo_i - rng.uniform(0, 0.1))
        for off in (0, 5):
            mode = rng.randint(3)
            if mode == 0: continue                                                   # wherever the draw above put it
            depth = rng.uniform(-0.05, 0.20) if mode == 1 else rng.uniform(0.0, 0.14)
            qp = rng.uniform(lo[2] - (0.08 if mode == 2 and rng.rand() < 0.3 else 0.0), 0.2)
            q[k, off + 2] = qp; q[k, off + 3] = T - depth - qp
        out = make_out()
        for i in range(10): set_motor(out, i, pos=q[k, i], vel=v[k, i])
        tau[k] = core_step(new_core(), out, cmd[k])
    inside = ((q[:, 2] + q[:, 3] < T) | (q[:, 7] + q[:, 8] < T)).sum()
    np.savez(os.path.join(GOLD, "g10b_core_sim_coupled.npz"), q=q, v=v, cmd=cmd, tau=tau)
    print("g10b: %d of %d states inside the coupled zone" % (inside, n))


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    g9(); g10(); g10b()
    print("wrote g9, g10, g10b")
