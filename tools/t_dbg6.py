import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apex_amd.vecenv import CassieVecEnv
from oracle import sim as S
g = CassieVecEnv(n_envs=64, dynamics_randomization=False, seed=0)
o = S.OracleEnv(dyn_rand=False, seed=0, env_id=0)
biw = g.get_field("body_invweight0").cpu().numpy()[0]
diw = g.get_field("dof_invweight0").cpu().numpy()[0]
rb = o.get("body_invweight0").reshape(26, 2)[:, 0]; rd = o.get("dof_invweight0")
np.set_printoptions(precision=5, suppress=True, linewidth=200)
print("biw gpu", biw); print("biw ref", rb)
print("diw gpu", diw); print("diw ref", rd)
