import numpy as np
from native_blocks import *
np.set_printoptions(precision=4, suppress=True, linewidth=200)
def resp(idx, q, v=0.0, cmd=0.0, other=None):
    out = make_out(); set_motor(out, idx, pos=q, vel=v)
    t = core_step(new_core(), out, [cmd]*10)
    return t
# locate zone boundaries by bisection for each joint/side (left leg)
def boundary(idx, inside, outside):
    a, b = inside, outside
    for _ in range(60):
        m = 0.5*(a+b)
        if abs(resp(idx, m)[idx]) > 0: b = m
        else: a = m
    return 0.5*(a+b)
nom = NOMINAL
for idx in range(5):
    lo = boundary(idx, nom[idx], nom[idx]-2.0); hi = boundary(idx, nom[idx], nom[idx]+2.0)
    print("joint", idx, "free interval [%.6f, %.6f]" % (lo, hi))
    for side, b0, sg in (("hi", hi, 1), ("lo", lo, -1)):
        ds = np.array([0.001, 0.002, 0.005, 0.01, 0.02, 0.03, 0.05, 0.08, 0.10, 0.12, 0.15, 0.2])
        t0 = np.array([resp(idx, b0 + sg*d)[idx] for d in ds])
        tv = np.array([resp(idx, b0 + sg*d, v=1.0)[idx] for d in ds])
        tn = np.array([resp(idx, b0 + sg*d, v=-1.0)[idx] for d in ds])
        sc = np.array([(resp(idx, b0 + sg*d, cmd=10.0)[(idx+1) % 5] ) / 10 for d in ds])
        print(" side", side, "d:", ds)
        print("   tau(v=0):", t0)
        print("   tau/d   :", t0/ds)
        print("   dtau(v=+1):", tv - t0, " dtau(v=-1):", tn - t0)
        print("   scale others:", sc)
