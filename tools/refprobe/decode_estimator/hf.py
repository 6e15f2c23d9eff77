import sys, ctypes, numpy as np
sys.path.insert(0, "/root/repo/tools/refprobe"); sys.path.insert(0, "/root/repo")
from native_blocks import cm
np.set_printoptions(precision=6, suppress=True, linewidth=250)
lib = cm._libraries["./libcassiemujoco.so"]
base = ctypes.cast(lib.state_output_step, ctypes.c_void_p).value - 0x296b0
D = ctypes.c_double
f1cd10 = ctypes.CFUNCTYPE(None, ctypes.c_void_p, D, D, D, D, D, D)(base + 0x1cd10)
est = cm.state_output_alloc(); cm.state_output_setup(est)
estp = ctypes.cast(est, ctypes.c_void_p).value
obj0 = (ctypes.c_double * 99).from_address(estp + 0x6e8)
template = np.array(obj0[:])            # after setup
def call(x, P, args, R=None, Q=None):
    buf = (ctypes.c_double * 99)(*template)
    buf[0:6] = list(x)
    buf[58:94] = list(np.asarray(P).T.reshape(-1))
    if R is not None: buf[42:58] = list(np.asarray(R).T.reshape(-1))
    if Q is not None: buf[6:42] = list(np.asarray(Q).T.reshape(-1))
    f1cd10(ctypes.addressof(buf), *[float(a) for a in args])
    o = np.array(buf[:])
    return o[0:6], o[58:94].reshape(6, 6).T, o
if __name__ == "__main__":
    print("template x", template[:6]); print("Q\n", template[6:42].reshape(6,6).T); print("R\n", template[42:58].reshape(4,4).T); print("P\n", template[58:94].reshape(6,6).T); print("tail", template[94:99])
    rng = np.random.default_rng(1)
    x0 = rng.normal(size=6); P0 = np.eye(6) * 1e-3
    for name, args in (("contact both", [0.1, 0.2, 0.3, -200, -150, 0.7]), ("no contact", [0.1, 0.2, 0.3, 0, 0, 0.7]), ("left only", [0.1, 0.2, 0.3, -200, 0, 0.7])):
        Rbig = np.eye(4) * 1e12
        xb, Pb, _ = call(x0, P0, args, R=Rbig)
        A = np.zeros((6, 6))
        for i in range(6):
            e = np.zeros(6); e[i] = 1
            xi, _, _ = call(x0 + e, P0, args, R=Rbig); A[:, i] = xi - xb
        Bm = np.zeros((6, 6))
        for i in range(6):
            a = list(args); a[i] += 1.0 if i not in (3, 4) else -1.0
            xi, _, _ = call(x0, P0, a, R=Rbig); Bm[:, i] = xi - xb
        print("=====", name); print("A (pred)\n", A); print("d x'/d args (col i = arg i; cols 3,4 = per -1N)\n", Bm)
        print("P' - A P A'\n", Pb - A @ P0 @ A.T)
        print("x' - A x0", xb - A @ x0)
