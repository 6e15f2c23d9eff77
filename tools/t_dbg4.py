import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apex_amd.vecenv import CassieVecEnv
from oracle import sim as S
N = 64
for dr in (False, True):
    g = CassieVecEnv(n_envs=N, dynamics_randomization=dr, seed=7)
    o = [S.OracleEnv(dyn_rand=dr, seed=7, env_id=i) for i in range(4)]
    obs = g.reset().cpu().numpy()
    ref = np.stack([e.reset() for e in o])
    bad = np.isnan(obs)
    print("dyn_rand", dr, "nan envs", np.where(bad.any(1))[0][:20], "nan cols", np.where(bad.any(0))[0])
    d = np.abs(obs[:4] - ref)
    print(" max diff per env", np.nanmax(d, 1))
    for name in ("qpos", "qvel"):
        a = g.get_field(name).cpu().numpy()
        print(" ", name, "nan envs", np.where(np.isnan(a).any(1))[0][:10], "maxdiff env0", np.nanmax(np.abs(a[0] - o[0].get(name))))
