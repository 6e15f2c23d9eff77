import numpy as np, json
np.set_printoptions(precision=6, suppress=True, linewidth=220)
m = json.load(open("/root/repo/apex_amd/cassie_model.json"))
def q2m(q):
    w, x, y, z = q
    return np.array([[1-2*(y*y+z*z), 2*(x*y-w*z), 2*(x*z+w*y)], [2*(x*y+w*z), 1-2*(x*x+z*z), 2*(y*z-w*x)], [2*(x*z-w*y), 2*(y*z+w*x), 1-2*(x*x+y*y)]])
def rotz(a): c, s = np.cos(a), np.sin(a); return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
B = {b["name"]: b for b in m["bodies"]}
J = {j["name"]: j for j in m["joints"]}
def fk_leg(side, ang):
    """ang: dict joint name suffix -> qpos value; returns dict body -> (R, p) relative to pelvis frame"""
    R, p = np.eye(3), np.zeros(3)
    res = {}
    for nm in ("hip-roll", "hip-yaw", "hip-pitch", "knee", "shin", "tarsus", "foot"):
        b = B[side + "-" + nm]; j = J[side + "-" + nm]
        p = p + R @ np.array(b["pos"]); R = R @ q2m(b["quat"]) @ rotz(ang[nm] - j["ref"])
        res[nm] = (R.copy(), p.copy())
    return res
if __name__ == "__main__":
    d = np.load("/tmp/est_stream.npz"); st, inp = d["st"], d["inp"]
    def M(s, i, r, c): return s[i:i+r*c].reshape(c, r).T
    for t in (0, 1500, 2500):
        x = inp[t]; s1 = st[t+1]
        mp, jp = x[0:10], x[30:36]
        for side, o, fr in (("left", 0, 81), ("right", 1, 103)):
            ang = {"hip-roll": mp[5*o], "hip-yaw": mp[5*o+1], "hip-pitch": mp[5*o+2], "knee": mp[5*o+3], "shin": jp[3*o], "tarsus": jp[3*o+1], "foot": jp[3*o+2]}
            r = fk_leg(side, ang)
            Rf, pf = r["foot"]
            Re, pe = M(s1, fr, 3, 3), s1[fr+9:fr+12]
            print(t, side, "fk p", pf, "est p", pe, "diff", pf - pe)
            print("  Re^T Rf\n", Re.T @ Rf, "\n  Rf^T (pe - pf)", Rf.T @ (pe - pf))
