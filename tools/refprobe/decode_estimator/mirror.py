"""Python mirror of the decoded state_output_step (velocity / position / terrain path), validated against the binary's state dumps."""
import numpy as np, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from fk import fk_leg, q2m, rotz, B, J, m as model
from lm import heel_solve, NAMES, PH, CO
np.set_printoptions(precision=9, suppress=True, linewidth=200)
OFF = np.array([0.017620176, 0.052189975, 0.0])
DT, G, HGT, MASS = 5e-4, 9.806, 1.0, 31.0
K_SHIN, K_HEEL = 1500.0, 1250.0
def grad_res(K, S, T, X):
    g = np.zeros(4)
    for n, p, c in zip(NAMES, PH, CO):
        u = n[0] * K + n[1] * S + n[2] * T + n[3] * X + p
        g += -c * np.sin(u) * np.array(n)
    return g
def foot_pos_and_derivs(side, q):
    """q = roll, yaw, pitch, knee, shin, tarsus, foot motor. returns p_rel (pelvis frame), dp/dS, dp/dT (analytic: axis x (p - joint origin))"""
    ang = dict(zip(("hip-roll", "hip-yaw", "hip-pitch", "knee", "shin", "tarsus", "foot"), q))
    fr = fk_leg(side, ang)
    Rf, pf = fr["foot"]; p = pf + Rf @ OFF
    d = {}
    for nm in ("shin", "tarsus"):
        Rj, pj = fr[nm]; d[nm] = np.cross(Rj[:, 2], p - pj)
    return p, d["shin"], d["tarsus"]
def mldivide23(Mm, tau):
    """MATLAB A\\b for a full-rank 2x3 system: QR with column pivoting, basic solution"""
    n0 = (Mm * Mm).sum(0); j1 = int(np.argmax(n0))
    a = Mm[:, j1]; e = a / np.linalg.norm(a)
    rest = [j for j in range(3) if j != j1]
    rem = [np.linalg.norm(Mm[:, j] - e * (e @ Mm[:, j])) for j in rest]
    j2 = rest[int(np.argmax(rem))]
    sol = np.linalg.solve(Mm[:, [j1, j2]], tau)
    x = np.zeros(3); x[j1] = sol[0]; x[j2] = sol[1]
    return x
class HFilter:
    def __init__(s): s.x = np.zeros(6); s.P = np.eye(6) * 1e-6
    def init(s, lfw, rfw): s.x = np.array([0, 0, -lfw, -rfw, 0.5, 0.0]); s.P = np.eye(6) * 1e-6
    def step(s, zL, zR, fl, fr, acc):
        tot = fl + fr; contact = not (1.0 > tot)
        alpha_m = fl / tot if contact else 0.5
        Q = np.diag([1e-8, 1e-8, 1e-6 if 50.0 > fl else 1e-10, 1e-6 if 50.0 > fr else 1e-10, 1e-5, 1e-2])
        p, v, pL, pR, al, fd = s.x; w2 = G / HGT
        A = np.eye(6); A[0, 1] = DT
        xp = s.x.copy(); xp[0] = p + DT * v
        if contact:
            xp[1] = v + DT * (w2 * (p - al * pL - (1 - al) * pR) + fd / MASS)
            A[1, 0] = DT * w2; A[1, 2] = -DT * w2 * al; A[1, 3] = -DT * w2 * (1 - al); A[1, 4] = -DT * w2 * (pL - pR); A[1, 5] = DT / MASS
        Pp = A @ s.P @ A.T + Q
        H = np.zeros((4, 6)); H[0, 0] = 1; H[0, 2] = -1; H[1, 0] = 1; H[1, 3] = -1; H[2, 4] = 1; H[3, 1] = 1
        R = np.diag([1e-6, 1e-6, 1e-6, 1.0])
        z = np.array([zL, zR, alpha_m, v + DT * acc])
        K = Pp @ H.T @ np.linalg.inv(H @ Pp @ H.T + R)
        s.x = xp + K @ (z - H @ xp); s.P = Pp - K @ H @ Pp
class ZFilter:
    def __init__(s): s.x = np.zeros(5); s.P = np.eye(5) * 1e-6
    def init(s, lfw, rfw): s.x = np.array([0, 0, -lfw, -rfw, MASS * G]); s.P = np.eye(5) * 1e-6
    def step(s, zL, zR, fl, fr):
        A = np.eye(5); A[0, 1] = DT; A[1, 4] = DT / MASS
        u = (fl + fr) / MASS - G
        Q = np.diag([1e-8, 1e-8, 1e-6 if 50.0 > fl else 1e-10, 1e-6 if 50.0 > fr else 1e-10, 0.01])
        x = A @ s.x; x[1] += DT * u
        P = A @ s.P @ A.T + Q
        H = np.zeros((2, 5)); H[0, 0] = 1; H[0, 2] = -1; H[1, 0] = 1; H[1, 3] = -1
        K = P @ H.T @ np.linalg.inv(H @ P @ H.T + np.eye(2) * 1e-6)
        s.x = x + K @ (np.array([zL, zR]) - H @ x); s.P = P - K @ H @ P
class Estimator:
    def __init__(s):
        s.heel = np.zeros(2); s.fx, s.fy, s.fz = HFilter(), HFilter(), ZFilter(); s.inited = False
        s.terr = 0.0
    def step(s, mp, jp, quat, gyro, accel, r_imu=np.array([0.03155, 0, -0.07996])):
        legL = (mp[3], jp[0], jp[1]); legR = (mp[8], jp[3], jp[4])
        s.heel = heel_solve(s.heel, legL, legR)
        Rp = q2m(quat)
        F = []; prel = []
        for side, o in (("left", 0), ("right", 1)):
            q = np.array([mp[5 * o], mp[5 * o + 1], mp[5 * o + 2], mp[5 * o + 3], jp[3 * o], jp[3 * o + 1], mp[5 * o + 4]])
            p, dS, dT = foot_pos_and_derivs(side, q)
            gK, gS, gT, gX = grad_res(q[3], q[4], q[5], s.heel[o])
            Av = dS - dT * gS / gT; Bv = -dT * gX / gT
            Mm = -np.stack([Av, Bv]); tau = np.array([K_SHIN * jp[3 * o], K_HEEL * s.heel[o]])
            F.append(Rp @ mldivide23(Mm, tau)); prel.append(p)
        s.prel = prel; s.F = F
        lfw, rfw = Rp @ prel[0], Rp @ prel[1]
        fl, fr = max(0.0, -F[0][2]), max(0.0, -F[1][2])
        acc_b = np.asarray(accel) - Rp.T @ np.array([0, 0, G]) - np.cross(gyro, np.cross(gyro, r_imu))
        s.acc_b = acc_b
        acc_w = Rp @ acc_b
        if not s.inited:
            s.fx.init(lfw[0], rfw[0]); s.fy.init(lfw[1], rfw[1]); s.fz.init(lfw[2], rfw[2]); s.inited = True
        s.fx.step(-lfw[0], -rfw[0], fl, fr, acc_w[0]); s.fy.step(-lfw[1], -rfw[1], fl, fr, acc_w[1]); s.fz.step(-lfw[2], -rfw[2], fl, fr)
        if fl + fr > 1.0:
            a = fl / (fl + fr); u = a * (s.fz.x[0] + lfw[2]) + (1 - a) * (s.fz.x[0] + rfw[2])
            s.terr = 0.0004997501249375313 * u + 0.9995002498750625 * s.terr
        s.pos = np.array([s.fx.x[0], s.fy.x[0], s.fz.x[0]]); s.vel = np.array([s.fx.x[1], s.fy.x[1], s.fz.x[1]])
        return s.pos, s.vel, s.terr
if __name__ == "__main__":
    d = np.load("/tmp/est_stream.npz"); st, out, inp = d["st"], d["out"], d["inp"]
    e = Estimator(); mx = np.zeros(8)
    for t in range(3000):
        x = inp[t]; s1 = st[t + 1]; o = out[t]
        pos, vel, terr = e.step(x[0:10], x[30:36], x[42:46], x[46:49], x[49:52])
        errs = np.array([np.abs(pos - o[0:3]).max(), np.abs(vel - o[10:13]).max(), abs(terr - o[60]), np.abs(e.prel[0] - s1[90:93]).max(), np.abs(e.prel[1] - s1[112:115]).max(),
                         np.abs(e.F[0] - s1[520:523]).max(), np.abs(e.F[1] - s1[523:526]).max(), np.abs(e.acc_b - s1[74:77]).max()])
        mx = np.maximum(mx, errs)
        if t < 3 or t % 500 == 0: print(t, errs)
    print("max errs [pos vel terr prelL prelR FL FR acc]:", mx)
