import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apex_amd.vecenv import CassieVecEnv
env = CassieVecEnv(n_envs=4096, seed=0)
env.reset()
act = torch.randn(4096,10,device='cuda')*0.2
for _ in range(3): env.step(act, auto_reset=False)
torch.cuda.synchronize()
