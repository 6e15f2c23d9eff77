import numpy as np, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from fk import *
from scipy.optimize import brentq
d = np.load("/tmp/est_stream.npz"); st, out, inp = d["st"], d["out"], d["inp"]
def M(s, i, r, c): return s[i:i+r*c].reshape(c, r).T
eq = m["equalities"][1]   # achilles (body 5) <-> heel spring (body 10)
rodlen = eq["anchor1"][0]; anc2 = np.array(eq["anchor2"])
def heel_defl(side, knee, shin, tarsus):
    # frames relative to hip-pitch body
    R, p = np.eye(3), np.zeros(3)
    for nm, ang in (("knee", knee), ("shin", shin), ("tarsus", tarsus)):
        b = B[side + "-" + nm]; j = J[side + "-" + nm]
        p = p + R @ np.array(b["pos"]); R = R @ q2m(b["quat"]) @ rotz(ang - j["ref"])
    bh = B[side + "-heel-spring"]; a2 = np.array(m["equalities"][1 if side == "left" else 3]["anchor2"])
    base = np.array(B[side + "-achilles-rod"]["pos"])
    def f(th):
        ph = p + R @ np.array(bh["pos"]); Rh = R @ q2m(bh["quat"]) @ rotz(th)
        return np.linalg.norm(ph + Rh @ a2 - base) - rodlen
    return brentq(f, -0.5, 0.5, xtol=1e-15)
for t in (0, 50, 500, 1000, 1500, 2500):
    x = inp[t]; s1 = st[t+1]; mp, jp = x[0:10], x[30:36]
    hl = heel_defl("left", mp[3], jp[0], jp[1]); hr = heel_defl("right", mp[8], jp[3], jp[4])
    print(t, "heel L mine %.9f est %.9f | R mine %.9f est %.9f   shin %.6f %.6f" % (hl, s1[208], hr, s1[210], jp[0], jp[3]), " idx25,26", s1[25], s1[26])
