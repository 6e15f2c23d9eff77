import numpy as np, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from fk import *
from lm import res_jac, NAMES, PH, CO
np.set_printoptions(precision=6, suppress=True, linewidth=220)
OFF = np.array([0.017620176, 0.052189975, 0.0])
def footp(side, q):   # q: roll,yaw,pitch,knee,shin,tarsus,footmotor
    ang = dict(zip(("hip-roll", "hip-yaw", "hip-pitch", "knee", "shin", "tarsus", "foot"), q))
    Rf, pf = fk_leg(side, ang)["foot"]
    return pf + Rf @ OFF
def grad_res(K, S, T, X):
    g = np.zeros(4)
    for n, p, c in zip(NAMES, PH, CO):
        u = n[0] * K + n[1] * S + n[2] * T + n[3] * X + p
        g += -c * np.sin(u) * np.array(n)
    return g   # d/dK, dS, dT, dX
d = np.load("/tmp/est_stream.npz"); st, inp = d["st"], d["inp"]
for t in (500, 1500, 2500):
    x = inp[t]; s1 = st[t+1]; mp, jp = x[0:10], x[30:36]
    q = np.array([mp[0], mp[1], mp[2], mp[3], jp[0], jp[1], mp[4]])
    h = 1e-6
    def dp(i):
        e = np.zeros(7); e[i] = h; return (footp("left", q + e) - footp("left", q - e)) / (2 * h)
    dK, dS, dT = dp(3), dp(4), dp(5)
    gK, gS, gT, gX = grad_res(mp[3], jp[0], jp[1], s1[25])
    print(t, "struct A", s1[156:159], "B", s1[162:165])
    print("   dp/dS", dS, " dp/dT", dT, " dp/dK", dK)
    print("   dp/dS - dp/dT gS/gT", dS - dT * gS / gT, "  -dp/dT gX/gT", -dT * gX / gT, "  dK-dT gK/gT", dK - dT * gK / gT)
    print("   full struct 123..165:\n", s1[123:165].reshape(-1, 3))
