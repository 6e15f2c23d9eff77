import sys, ctypes, numpy as np
sys.path.insert(0, "/root/repo/tools/refprobe"); sys.path.insert(0, "/root/repo"); sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from native_blocks import cm, make_out, DRIVES, JOINTS
from fk import fk_leg, q2m, rotz, B, J
np.set_printoptions(precision=9, suppress=True, linewidth=220)
def dump(est):
    p = ctypes.cast(est, ctypes.POINTER(ctypes.c_double)); return np.array([p[i] for i in range(526)])
def M(s, i, r, c): return s[i:i+r*c].reshape(c, r).T
def run(mp, jp, quat=(1,0,0,0), mv=None, jv=None, gyro=(0,0,0), acc=(0,0,9.806), n=1, est=None):
    if est is None:
        est = cm.state_output_alloc(); cm.state_output_setup(est)
    for _ in range(n):
        out = make_out()
        for i in range(10):
            d = getattr(out.leftLeg if i < 5 else out.rightLeg, DRIVES[i % 5]); d.position = mp[i]; d.velocity = 0 if mv is None else mv[i]
        for i in range(6):
            j = getattr(out.leftLeg if i < 3 else out.rightLeg, JOINTS[i % 3]); j.position = jp[i]; j.velocity = 0 if jv is None else jv[i]
        for k in range(4): out.pelvis.vectorNav.orientation[k] = quat[k]
        for k in range(3): out.pelvis.vectorNav.angularVelocity[k] = gyro[k]; out.pelvis.vectorNav.linearAcceleration[k] = acc[k]
        so = cm.state_out_t(); cm.state_output_step(est, out, so)
    return dump(est), so, est
if __name__ == "__main__":
    rng = np.random.default_rng(0)
    nomm = np.array([0.0045, 0, 0.4973, -1.1997, -1.5968] * 2); nomj = np.array([0, 1.4267, -1.5968] * 2)
    rows = []
    for trial in range(6):
        mp = nomm + rng.uniform(-0.2, 0.2, 10); jp = nomj + rng.uniform(-0.1, 0.1, 6)
        s, so, _ = run(mp, jp)
        for side, o, fr in (("left", 0, 81), ("right", 1, 103)):
            for footsrc in ("joint", "motor"):
                ang = {"hip-roll": mp[5*o], "hip-yaw": mp[5*o+1], "hip-pitch": mp[5*o+2], "knee": mp[5*o+3], "shin": jp[3*o], "tarsus": jp[3*o+1], "foot": jp[3*o+2] if footsrc == "joint" else mp[5*o+4]}
                r = fk_leg(side, ang); Rf, pf = r["foot"]; Rt, pt = r["tarsus"]
                Re, pe = M(s, fr, 3, 3), s[fr+9:fr+12]
                print(trial, side, footsrc, "C=Rf^T Re row0", (Rf.T @ Re)[0], "off", Rf.T @ (pe - pf), " off in tarsus", Rt.T @ (pe - pt))
