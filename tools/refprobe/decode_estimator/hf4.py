from hf import *
np.set_printoptions(precision=9, suppress=True, linewidth=250)
dt, g, hgt, m = 5e-4, 9.806, 1.0, 31.0
def hfilter(x, P, a, variant=0):
    a0, a1, a2, FL, FR, acc = a
    fl, fr = max(0.0, FL), max(0.0, FR)
    tot = fl + fr
    contact = not (1.0 > tot)
    alpha_m = fl / tot if contact else 0.5
    Q = np.diag([1e-8, 1e-8, 1e-6 if 50.0 > fl else 1e-10, 1e-6 if 50.0 > fr else 1e-10, 1e-5, 1e-2])
    p, v, pL, pR, al, fd = x
    w2 = g / hgt
    A = np.eye(6); A[0, 1] = dt
    xp = np.array(x, dtype=float)
    xp[0] = p + dt * v
    if contact:
        xp[1] = v + dt * (w2 * (p - al * pL - (1 - al) * pR) + fd / m)
        A[1, 0] = dt * w2; A[1, 2] = -dt * w2 * al; A[1, 3] = -dt * w2 * (1 - al); A[1, 4] = -dt * w2 * (pL - pR); A[1, 5] = dt / m
    Pp = A @ P @ A.T + Q
    H = np.zeros((4, 6)); H[0, 0] = 1; H[0, 2] = -1; H[1, 0] = 1; H[1, 3] = -1; H[2, 4] = 1; H[3, 1] = 1
    R = np.diag([1e-6, 1e-6, 1e-6, 1.0])
    z = np.array([a0 - a1, a0 - a2, alpha_m, v + dt * acc])
    S = H @ Pp @ H.T + R
    K = Pp @ H.T @ np.linalg.inv(S)
    xn = xp + K @ (z - H @ xp)
    Pn = Pp - K @ H @ Pp
    return xn, Pn
rng = np.random.default_rng(5)
for trial in range(8):
    x0 = rng.normal(size=6) * 0.3; x0[4] = rng.uniform(0, 1)
    L = rng.normal(size=(6, 6)) * 0.02; P0 = L @ L.T + np.eye(6) * 1e-5
    args = [rng.normal() * 0.1, rng.normal() * 0.3, rng.normal() * 0.3, rng.choice([0, 0.3, 30, 200]), rng.choice([0, 0.3, 30, 200]), rng.normal()]
    xb, Pb, _ = call(x0, P0, args)
    xm, Pm = hfilter(x0, P0, args)
    print(trial, "F", args[3], args[4], "x err", np.abs(xb - xm).max(), "P err", np.abs(Pb - Pm).max(), "  x diff", xb - xm)
