"""fp32 prototype of the kernel-side filters (sequential scalar measurement updates, P rows), compared with the binary's outputs"""
import numpy as np, sys
from mirror import *
f = np.float32
class HF32:
    def __init__(s): s.x = np.zeros(6, f); s.P = np.eye(6, dtype=f) * f(1e-6)
    def init(s, lfw, rfw): s.x = np.array([0, 0, -lfw, -rfw, 0.5, 0.0], f); s.P = np.eye(6, dtype=f) * f(1e-6)
    def step(s, zL, zR, fl, fr, acc):
        dt, w2 = f(DT), f(G / HGT)
        tot = fl + fr; contact = not (1.0 > tot)
        alpha_m = f(fl / tot) if contact else f(0.5)
        q = np.array([1e-8, 1e-8, 1e-6 if 50.0 > fl else 1e-10, 1e-6 if 50.0 > fr else 1e-10, 1e-5, 1e-2], f)
        p, v, pL, pR, al, fd = s.x
        a = np.zeros(6, f)       # row 1 of A minus e1
        if contact: a = np.array([dt * w2, 0, -dt * w2 * al, -dt * w2 * (1 - al), -dt * w2 * (pL - pR), dt / f(MASS)], f)
        xp = s.x.copy(); xp[0] = p + dt * v
        if contact: xp[1] = v + dt * (w2 * (p - al * pL - (1 - al) * pR) + fd / f(MASS))
        # P <- A P A^T + Q with A = I + e0 dt e1^T + e1 a^T
        P = s.P
        A = np.eye(6, dtype=f); A[0, 1] = dt; A[1, :] += a
        P = (A @ P @ A.T).astype(f) + np.diag(q)
        x = xp
        zv = v + dt * f(acc)
        for hidx, z, r in (((0, 2), f(zL), f(1e-6)), ((0, 3), f(zR), f(1e-6)), ((4, None), alpha_m, f(1e-6)), ((1, None), zv, f(1.0))):
            i, j = hidx
            Ph = P[:, i] - (P[:, j] if j is not None else 0)          # P h
            hx = x[i] - (x[j] if j is not None else 0)
            sden = Ph[i] - (Ph[j] if j is not None else 0) + r
            K = (Ph / sden).astype(f)
            x = (x + K * (z - hx)).astype(f)
            hP = P[i, :] - (P[j, :] if j is not None else 0)
            P = (P - np.outer(K, hP)).astype(f)
            P = ((P + P.T) * f(0.5)).astype(f)
        s.x, s.P = x, P
class ZF32:
    def __init__(s): s.x = np.zeros(5, f); s.P = np.eye(5, dtype=f) * f(1e-6)
    def init(s, lfw, rfw): s.x = np.array([0, 0, -lfw, -rfw, MASS * G], f); s.P = np.eye(5, dtype=f) * f(1e-6)
    def step(s, zL, zR, fl, fr):
        dt = f(DT)
        A = np.eye(5, dtype=f); A[0, 1] = dt; A[1, 4] = dt / f(MASS)
        u = f((fl + fr) / MASS - G)
        q = np.array([1e-8, 1e-8, 1e-6 if 50.0 > fl else 1e-10, 1e-6 if 50.0 > fr else 1e-10, 0.01], f)
        x = (A @ s.x).astype(f); x[1] += dt * u
        P = (A @ s.P @ A.T).astype(f) + np.diag(q)
        for (i, j), z in (((0, 2), f(zL)), ((0, 3), f(zR))):
            Ph = P[:, i] - P[:, j]; sden = Ph[i] - Ph[j] + f(1e-6); K = (Ph / sden).astype(f)
            x = (x + K * (z - (x[i] - x[j]))).astype(f)
            P = (P - np.outer(K, P[i, :] - P[j, :])).astype(f); P = ((P + P.T) * f(0.5)).astype(f)
        s.x, s.P = x, P
d = np.load("/tmp/est_stream.npz"); st, out, inp = d["st"], d["out"], d["inp"]
e = Estimator(); e.fx, e.fy, e.fz = HF32(), HF32(), ZF32()
mx = np.zeros(3)
for t in range(3000):
    x = inp[t]; o = out[t]
    pos, vel, terr = e.step(x[0:10], x[30:36], x[42:46], x[46:49], x[49:52])
    errs = np.array([np.abs(pos - o[0:3]).max(), np.abs(vel - o[10:13]).max(), abs((pos[2] - terr) - (o[2] - o[60]))]); mx = np.maximum(mx, errs)
    if t % 500 == 0: print(t, errs, vel, o[10:13])
print("fp32 filters: max err pos %.2e vel %.2e height %.2e" % tuple(mx))
