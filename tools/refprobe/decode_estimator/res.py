from hf import *
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from fk import *
f18e00 = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p)(base + 0x18e00)
def call18e00(mot5, jnt3, x):
    a = (ctypes.c_double * 5)(*mot5); b = (ctypes.c_double * 3)(*jnt3); r = (ctypes.c_double * 1)(0); gr = (ctypes.c_double * 8)(*([7.0] * 8))
    f18e00(ctypes.addressof(a), ctypes.addressof(b), x, ctypes.addressof(r), ctypes.addressof(gr))
    return r[0], np.array(gr[:])
def closure(side, knee, shin, tarsus, th):
    R, p = np.eye(3), np.zeros(3)
    for nm, ang in (("knee", knee), ("shin", shin), ("tarsus", tarsus)):
        b = B[side + "-" + nm]; j = J[side + "-" + nm]
        p = p + R @ np.array(b["pos"]); R = R @ q2m(b["quat"]) @ rotz(ang - j["ref"])
    bh = B[side + "-heel-spring"]; a2 = np.array(m["equalities"][1 if side == "left" else 3]["anchor2"])
    basep = np.array(B[side + "-achilles-rod"]["pos"])
    ph = p + R @ np.array(bh["pos"]); Rh = R @ q2m(bh["quat"]) @ rotz(th)
    dvec = ph + Rh @ a2 - basep
    return dvec @ dvec - 0.5012 ** 2
if __name__ == "__main__":
    rng = np.random.default_rng(0)
    rows = []
    for t in range(12):
        mot = np.array([0.0045, 0, 0.4973, -1.1997, -1.5968]) + rng.uniform(-0.3, 0.3, 5); jn = np.array([0, 1.4267, -1.5968]) + rng.uniform(-0.1, 0.1, 3); x = rng.uniform(-0.1, 0.1)
        r, gr = call18e00(mot, jn, x)
        c = closure("left", mot[3], jn[0], jn[1], x)
        rows.append((r, c))
        print("r %.9f  mine %.9f  ratio %.6f  grad %s" % (r, c, r / c, gr))
    rows = np.array(rows); A = np.stack([rows[:, 1], np.ones(len(rows))], 1)
    sol = np.linalg.lstsq(A, rows[:, 0], rcond=None); print("fit r = a*mine + b:", sol[0], "resid", np.abs(A @ sol[0] - rows[:, 0]).max())
    # dependence check: which inputs matter
    mot = np.array([0.0045, 0, 0.4973, -1.1997, -1.5968]); jn = np.array([0, 1.4267, -1.5968])
    r0, _ = call18e00(mot, jn, 0.0)
    for i in range(5):
        mm = mot.copy(); mm[i] += 0.1; print("d/dmot", i, call18e00(mm, jn, 0.0)[0] - r0)
    for i in range(3):
        jj = jn.copy(); jj[i] += 0.1; print("d/djnt", i, call18e00(mot, jj, 0.0)[0] - r0)
