"""VERDICT r4 item 8, measured before building: would MuJoCo's `tolerance` exit (cassie.xml:5 leaves the default 1e-8) shorten the kernel's sweep loop?
The fp64 oracle with tolerance = 1e-8 on bench-like rollouts (reference-initialised 2 x 256 actor, sigma = exp(-1.5) action noise, dynamics randomisation, episodes from
reset): histogram of the sweeps a forward pass runs, and what a WAVE of four envs (it leaves the loop when its slowest env does) would run with an exit test every k-th
sweep.  CPU only:  python tools/t_pgs_tolerance.py [n_envs] [n_steps]   ->  profiles/r05_pgs_tolerance_hist.txt"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sim as S

N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32
rs = np.random.RandomState(0)
# a random-init policy's mean is small against its noise: actions = sigma * N(0, 1) around a fixed small random linear map of the observation
W = rs.randn(10, 50) * 0.02
hist = np.zeros(51, dtype=np.int64)
for i in range(N):
    e = S.OracleEnv(dyn_rand=True, seed=0, env_id=i)
    sv = e.get("solver"); sv[3] = 1e-8; e.set("solver", sv)
    obs = e.reset()
    h0 = e.get("solver_hist").copy()
    for t in range(T):
        a = W @ np.asarray(obs)[:50] + np.exp(-1.5) * rs.randn(10)
        obs, r, done = e.step(a)[:3]
        if done:
            obs = e.reset()
    hist += (e.get("solver_hist") - h0).astype(np.int64)
p = hist / hist.sum()
k = np.arange(51)
cdf = np.cumsum(p)
emax4 = float(np.sum(k * (cdf ** 4 - np.concatenate([[0.0], cdf[:-1]]) ** 4)))
lines = ["# python tools/t_pgs_tolerance.py %d %d: fp64 oracle, tolerance 1e-8, %d forward passes" % (N, T, hist.sum()),
         "mean sweeps per pass %.2f, passes that run all 50: %.1f %%, median %d" % (float((k * p).sum()), 100 * p[50], int(np.searchsorted(cdf, 0.5))),
         "E[max over the 4 envs of a wave] = %.2f sweeps (independent draws from the histogram)" % emax4]
SWEEP, CHECK = 187, 60      # instructions per sweep of the shipped kernel / a lower bound for one exit test in Gram space (delta-form dual cost)
for every in (1, 2, 4, 5, 8):
    # an env that converged at sweep s is seen at the next multiple of `every`; the wave exits at the max over its four envs
    seen = np.minimum(np.ceil(np.maximum(k, 1) / every) * every, 50)
    order = np.argsort(seen, kind="stable")
    cs = np.cumsum(p[order]); vs = seen[order]
    e_exit = float(np.sum(vs * (cs ** 4 - np.concatenate([[0.0], cs[:-1]]) ** 4)))
    checks = e_exit / every
    net = (50 - e_exit) * SWEEP - checks * CHECK
    lines.append("exit test every %d sweeps: E[exit sweep of a wave] %.1f, %.1f tests -> %+.0f instructions per pass (%.1f sweeps saved x %d - tests x %d) = %+.1f %% of the 9 350-instruction sweep loop"
                 % (every, e_exit, checks, net, 50 - e_exit, SWEEP, CHECK, 100.0 * net / (50 * SWEEP)))
lines.append("histogram (sweeps: share of passes): " + " ".join("%d:%.3f" % (i, p[i]) for i in range(51) if p[i] >= 0.002))
txt = "\n".join(lines)
print(txt)
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05_pgs_tolerance_hist.txt")
open(out, "w").write(txt + "\n")
