import numpy as np
np.set_printoptions(precision=6, suppress=True, linewidth=250)
d = np.load("/tmp/est_stream.npz"); st, out, inp = d["st"], d["out"], d["inp"]
def M(s, i, r, c): return s[i:i+r*c].reshape(c, r).T
dt, g, hgt, m = 5e-4, 9.806, 1.0, 31.0
def implied(x, P, xn, fl, fr):
    tot = fl + fr; contact = not (1.0 > tot)
    Q = np.diag([1e-8, 1e-8, 1e-6 if 50.0 > fl else 1e-10, 1e-6 if 50.0 > fr else 1e-10, 1e-5, 1e-2])
    p, v, pL, pR, al, fd = x; w2 = g / hgt
    A = np.eye(6); A[0, 1] = dt; xp = np.array(x); xp[0] = p + dt * v
    if contact:
        xp[1] = v + dt * (w2 * (p - al * pL - (1 - al) * pR) + fd / m)
        A[1, 0] = dt * w2; A[1, 2] = -dt * w2 * al; A[1, 3] = -dt * w2 * (1 - al); A[1, 4] = -dt * w2 * (pL - pR); A[1, 5] = dt / m
    Pp = A @ P @ A.T + Q
    H = np.zeros((4, 6)); H[0, 0] = 1; H[0, 2] = -1; H[1, 0] = 1; H[1, 3] = -1; H[2, 4] = 1; H[3, 1] = 1
    R = np.diag([1e-6, 1e-6, 1e-6, 1.0])
    K = Pp @ H.T @ np.linalg.inv(H @ Pp @ H.T + R)
    rhs = xn - xp + K @ H @ xp
    z, res, rk, sv = np.linalg.lstsq(K, rhs, rcond=None)
    return z, np.abs(K @ z - rhs).max(), xp
for base, ax in ((221, 0), (320, 1)):
    print("axis", ax)
    for t in (1, 2, 3, 50, 500, 1000, 1500, 2500):
        s, s1 = st[t], st[t+1]
        x, P, xn = s[base:base+6], M(s, base+58, 6, 6), s1[base:base+6]
        FL, FR = s1[520:523], s1[523:526]
        fl, fr = max(0, -FL[2]), max(0, -FR[2])
        z, r, xp = implied(x, P, xn, fl, fr)
        Rp = M(s1, 53, 3, 3)
        lfw, rfw = Rp @ s1[90:93], Rp @ s1[112:115]
        accw = Rp @ s1[74:77]
        print(t, "z", z, "resid %.1e" % r, " -lfw %.6f -rfw %.6f  alpha %.4f  (z4-v)/dt %.5f  accw %s acc_in %s" % (-lfw[ax], -rfw[ax], fl/(fl+fr) if fl+fr >= 1 else .5, (z[3]-x[1])/dt, accw, inp[t][49:52]))
