from res import *
from rodata import rd
C = lambda a: rd(a)
def resid(K, S, T, X):
    c = np.cos
    A9 = c(T - X + C(0x2f7c0)); A10 = c(T + S + K + C(0x2f7c8)); A11 = c(X + C(0x2f7d0)); A12 = c(T + S + K - X + C(0x2f7d8))
    A13 = c(K + C(0x2f7e0)); A14 = c(T + C(0x2f7e8)); A15 = c(T + S + K + X - C(0x2f7f0)); A16 = c(2 * X + C(0x2f7f8))
    A17 = c(X + T + S - C(0x2f800)); A18 = c(S - C(0x2f808)); A19 = c(K + S + C(0x2f810)); A20 = c(T + S + C(0x2f818))
    A21 = c(T + S - X - C(0x2f820)); A22 = c(T + X - C(0x2f828))
    r = (C(0x2f830) * A9 * C(0x2f838) - C(0x2f840) * A10 * C(0x2f848) - C(0x2f850) * A11 * C(0x2f858) - C(0x2f860) * A12 * C(0x2f868)
         + C(0x2f870) * A13 * C(0x2f848) - C(0x2f878) * A14 * C(0x2f880) - C(0x2f888) * A15 * C(0x2f890) - C(0x2f898) * A16 * C(0x2f8a0)
         - C(0x2f8a8) * A17 * C(0x2f8b0) + C(0x2f8b8) * A18 * C(0x2f880) + C(0x2f8c0) * A19 * C(0x2f8c8) - C(0x2f8d0) * A20 * C(0x2f8d8)
         - C(0x2f8e0) * A21 * C(0x2f8e8) - A22 * C(0x2f8f0) * C(0x2f8f8) - C(0x2f900))
    return r
rng = np.random.default_rng(0)
for t in range(6):
    mot = np.array([0.0045, 0, 0.4973, -1.1997, -1.5968]) + rng.uniform(-0.3, 0.3, 5); jn = np.array([0, 1.4267, -1.5968]) + rng.uniform(-0.1, 0.1, 3); x = rng.uniform(-0.1, 0.1)
    r, gr = call18e00(mot, jn, x)
    rm = resid(mot[3], jn[0], jn[1], x)
    h = 1e-6; dX = (resid(mot[3], jn[0], jn[1], x + h) - resid(mot[3], jn[0], jn[1], x - h)) / (2 * h)
    print("r %.12f mine %.12f diff %.2e   dX bin %.9f fd %.9f" % (r, rm, r - rm, gr[7], dX))
print("terms:")
names = ["T-X", "T+S+K", "X", "T+S+K-X", "K", "T", "T+S+K+X", "2X", "X+T+S", "S", "K+S", "T+S", "T+S-X", "T+X"]
ph = [C(0x2f7c0), C(0x2f7c8), C(0x2f7d0), C(0x2f7d8), C(0x2f7e0), C(0x2f7e8), -C(0x2f7f0), C(0x2f7f8), -C(0x2f800), -C(0x2f808), C(0x2f810), C(0x2f818), -C(0x2f820), -C(0x2f828)]
co = [C(0x2f830) * C(0x2f838), -C(0x2f840) * C(0x2f848), -C(0x2f850) * C(0x2f858), -C(0x2f860) * C(0x2f868), C(0x2f870) * C(0x2f848), -C(0x2f878) * C(0x2f880), -C(0x2f888) * C(0x2f890),
      -C(0x2f898) * C(0x2f8a0), -C(0x2f8a8) * C(0x2f8b0), C(0x2f8b8) * C(0x2f880), C(0x2f8c0) * C(0x2f8c8), -C(0x2f8d0) * C(0x2f8d8), -C(0x2f8e0) * C(0x2f8e8), -C(0x2f8f0) * C(0x2f8f8)]
for n, p, cc in zip(names, ph, co): print("  %-10s coef % .12e  phase % .15f" % (n, cc, p))
print("  const", -C(0x2f900))
for a in (0x2f830,0x2f838,0x2f840,0x2f848,0x2f850,0x2f858,0x2f860,0x2f868,0x2f870,0x2f878,0x2f880,0x2f888,0x2f890,0x2f898,0x2f8a0,0x2f8a8,0x2f8b0,0x2f8b8,0x2f8c0,0x2f8c8,0x2f8d0,0x2f8d8,0x2f8e0,0x2f8e8,0x2f8f0,0x2f8f8,0x2f900): print(hex(a), repr(C(a)))
