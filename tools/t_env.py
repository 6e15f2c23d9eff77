import sys, torch, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apex_amd.vecenv import CassieVecEnv
for iters in (50, 1):
    env = CassieVecEnv(n_envs=4096, seed=0, pgs_iters=iters)
    env.reset()
    act = torch.randn(4096,10,device='cuda')*0.2
    for _ in range(2): env.step(act)
    torch.cuda.synchronize(); t0=time.time()
    K=5
    for _ in range(K): env.step(act)
    torch.cuda.synchronize(); dt=(time.time()-t0)/K
    print("pgs_iters", iters, "env step ms %.1f"%(dt*1e3), "env-steps/s %.0f"%(4096/dt))
    env.close()
