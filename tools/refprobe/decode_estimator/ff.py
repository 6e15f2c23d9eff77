from hf import *
import scipy.linalg as sl
f21000 = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)(base + 0x21000)
def call21000(M6, tau2):
    a = (ctypes.c_double * 6)(*M6); b = (ctypes.c_double * 2)(*tau2); o = (ctypes.c_double * 3)(7, 7, 7)
    f21000(ctypes.addressof(a), ctypes.addressof(b), ctypes.addressof(o))
    return np.array(o[:])
def mldivide(Mm, tau):
    # MATLAB A\b for underdetermined: QR with column pivoting, basic solution
    Q, R, piv = sl.qr(Mm, pivoting=True)
    r = Mm.shape[0]
    y = Q.T @ tau
    xb = np.zeros(Mm.shape[1])
    xb[piv[:r]] = np.linalg.solve(R[:, :r], y)
    return xb
rng = np.random.default_rng(3)
for t in range(6):
    Mm = rng.normal(size=(2, 3)); tau = rng.normal(size=2)
    o = call21000(Mm.T.reshape(-1), tau)        # col-major 2x3
    print("bin", o, " minnorm", np.linalg.pinv(Mm) @ tau, " mldivide", mldivide(Mm, tau), " resid", Mm @ o - tau)
