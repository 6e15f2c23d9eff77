import numpy as np
np.set_printoptions(precision=6, suppress=True, linewidth=220)
d = np.load("/tmp/est_stream.npz"); st, out, inp = d["st"], d["out"], d["inp"]
def M(s, i, r, c): return s[i:i+r*c].reshape(c, r).T
s0 = st[0]
A = M(s0, 424, 5, 5); B = s0[449:454]; H = M(s0, 454, 2, 5); R = M(s0, 489, 2, 2)
g, m = s0[518], s0[519]
x = st[1][419:424].copy(); P = M(st[1], 493, 5, 5).copy()
maxe = 0
for t in range(1, 3000):
    s1 = st[t+1]
    FL, FR = s1[520:523], s1[523:526]
    fl, fr = max(0, -FL[2]), max(0, -FR[2])
    u = (fl + fr) / m - g
    Q = np.diag([1e-8, 1e-8, 1e-6 if fl < 50 else 1e-10, 1e-6 if fr < 50 else 1e-10, 0.01])
    x = A @ x + B * u
    P = A @ P @ A.T + Q
    Rp = M(s1, 53, 3, 3)
    lf, rf = Rp @ s1[90:93], Rp @ s1[112:115]
    zm = np.array([-lf[2], -rf[2]])
    S = H @ P @ H.T + R
    K = P @ H.T @ np.linalg.inv(S)
    x = x + K @ (zm - H @ x)
    P = P - K @ H @ P
    e = np.abs(x - s1[419:424]).max()
    maxe = max(maxe, e)
print("max err", maxe)
# is Rp = R(quat)?
def q2m(q):
    w, x, y, z = q
    return np.array([[1-2*(y*y+z*z), 2*(x*y-w*z), 2*(x*z+w*y)], [2*(x*y+w*z), 1-2*(x*x+z*z), 2*(y*z-w*x)], [2*(x*z-w*y), 2*(y*z+w*x), 1-2*(x*x+y*y)]])
t = 1500
print(M(st[t+1], 53, 3, 3) - q2m(inp[t][42:46]))
print("terrain lowpass", st[t+1][218:221], "out terrain", out[t][60], "pelvis z", out[t][2])
