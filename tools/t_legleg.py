"""How often would leg-leg capsule pairs touch under a random policy?  (oracle rollouts, segment-segment distances)"""
import sys, os, json, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sim as S
m = json.load(open(os.path.join(os.path.dirname(__file__), "..", "apex_amd", "cassie_model.json")))
geoms = [g for g in m["geoms"] if g["type"] == "capsule"]
def q2m(q):
    w, x, y, z = q
    return np.array([[1-2*(y*y+z*z), 2*(x*y-w*z), 2*(x*z+w*y)], [2*(x*y+w*z), 1-2*(x*x+z*z), 2*(y*z-w*x)], [2*(x*z-w*y), 2*(y*z+w*x), 1-2*(x*x+y*y)]])
def segdist(p1, q1, p2, q2):
    d1, d2, r = q1 - p1, q2 - p2, p1 - p2
    a, e, f = d1 @ d1, d2 @ d2, d2 @ r
    c, b = d1 @ r, d1 @ d2
    den = a * e - b * b
    s = np.clip((b * f - c * e) / den, 0, 1) if den > 1e-12 else 0.0
    t = (b * s + f) / e
    if t < 0: t, s = 0.0, np.clip(-c / a, 0, 1)
    elif t > 1: t, s = 1.0, np.clip((b - c) / a, 0, 1)
    return np.linalg.norm(p1 + d1 * s - p2 - d2 * t)
rng = np.random.RandomState(0)
steps = pen = 0; mind = 9.0
for ep in range(24):
    e = S.OracleEnv(dyn_rand=True, seed=100 + ep, env_id=ep); e.reset()
    for t in range(150):
        _, r, d = e.step(rng.randn(10) * 0.25)
        xp = e.get("xpos").reshape(-1, 3); xq = e.get("xquat").reshape(-1, 4)
        segs = {}
        for g in geoms:
            b = g["body"]; R = q2m(xq[b]); c = xp[b] + R @ np.array(g["pos"]); ax = R @ np.array(g["axis"]) * g["half"]
            segs.setdefault(b >= 14, []).append((c - ax, c + ax, g["radius"]))
        dmin = min(segdist(a[0], a[1], b[0], b[1]) - a[2] - b[2] for a in segs.get(False, []) for b in segs.get(True, []))
        mind = min(mind, dmin); steps += 1; pen += int(dmin < 0)
        if d: break
print("env steps %d, steps with a penetrating left-right capsule pair %d (%.2f%%), min signed distance %.4f" % (steps, pen, 100.0 * pen / steps, mind))
print("capsule geoms:", [(g["body"], round(g["radius"], 3)) for g in geoms])
