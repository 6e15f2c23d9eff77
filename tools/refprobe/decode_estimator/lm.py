import numpy as np
from rodata import rd
C = rd
NAMES = [(0,0,1,-1), (1,1,1,0), (0,0,0,1), (1,1,1,-1), (1,0,0,0), (0,0,1,0), (1,1,1,1), (0,0,0,2), (0,1,1,1), (0,1,0,0), (1,1,0,0), (0,1,1,0), (0,1,1,-1), (0,0,1,1)]   # (K,S,T,X) multiples
PH = [C(0x2f7c0), C(0x2f7c8), C(0x2f7d0), C(0x2f7d8), C(0x2f7e0), C(0x2f7e8), -C(0x2f7f0), C(0x2f7f8), -C(0x2f800), -C(0x2f808), C(0x2f810), C(0x2f818), -C(0x2f820), -C(0x2f828)]
CO = [C(0x2f830) * C(0x2f838), -C(0x2f840) * C(0x2f848), -C(0x2f850) * C(0x2f858), -C(0x2f860) * C(0x2f868), C(0x2f870) * C(0x2f848), -C(0x2f878) * C(0x2f880), -C(0x2f888) * C(0x2f890),
      -C(0x2f898) * C(0x2f8a0), -C(0x2f8a8) * C(0x2f8b0), C(0x2f8b8) * C(0x2f880), C(0x2f8c0) * C(0x2f8c8), -C(0x2f8d0) * C(0x2f8d8), -C(0x2f8e0) * C(0x2f8e8), -C(0x2f8f0) * C(0x2f8f8)]
C0 = -C(0x2f900)
def res_jac(K, S, T, X):
    r = C0; j = 0.0
    for (nk, ns, nt, nx), p, c in zip(NAMES, PH, CO):
        u = nk * K + ns * S + nt * T + nx * X + p
        r += c * np.cos(u); j += -c * np.sin(u) * nx
    return r, j
TAU, KMAX, EPS1, EPS2 = 1e-3, 5.0, 1.4901161193847656e-08, 1.4901161193847656e-08
LB, UB = -np.pi / 4 + 1e-6, np.pi / 4 - 1e-6
def heel_solve(xprev, legL, legR):
    """legL/legR = (knee, shin, tarsus); xprev = previous (heelL, heelR)"""
    x = np.clip(np.array(xprev, dtype=float), LB, UB)
    def f(x):
        rL, jL = res_jac(*legL, x[0]); rR, jR = res_jac(*legR, x[1]); return np.array([rL, rR]), np.array([jL, jR])
    r, J = f(x)
    g = J * r; A = J * J
    if np.abs(g).max() <= EPS1: return x
    mu = TAU * A.max(); nu = 2.0; F = 0.5 * (r @ r); k = 0.0; stop = False
    while not stop:
        k += 1.0
        if k > KMAX: stop = True
        h = -g / (A + mu)
        if np.linalg.norm(h) <= EPS2 * (np.linalg.norm(x) + EPS2): break
        xn = np.clip(x + h, LB, UB)
        rn, Jn = f(xn); Fn = 0.5 * (rn @ rn)
        rho = (F - Fn) / (0.5 * (h @ (mu * h - g)))
        if rho > 0:
            x = xn; r, J, F = rn, Jn, Fn; g = J * r; A = J * J
            if np.abs(g).max() <= EPS1: break
            mu *= max(1.0 / 3.0, 1.0 - (2 * rho - 1) ** 3); nu = 2.0
        else:
            mu *= nu; nu *= 2.0
    return x
if __name__ == "__main__":
    d = np.load("/tmp/est_stream.npz"); st, inp = d["st"], d["inp"]
    mx = 0
    for t in range(3000):
        x = inp[t]; mp, jp = x[0:10], x[30:36]
        prev = st[t][25:27]
        sol = heel_solve(prev, (mp[3], jp[0], jp[1]), (mp[8], jp[3], jp[4]))
        e = np.abs(sol - st[t+1][25:27]).max(); mx = max(mx, e)
        if t < 5 or t % 500 == 0: print(t, sol, st[t+1][25:27], e)
    print("max err", mx)
