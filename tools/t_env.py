import torch, time
from apex_amd.vecenv import CassieVecEnv
env = CassieVecEnv(n_envs=4096, seed=0)
env.reset()
act = torch.randn(4096,10,device='cuda')*0.2
for _ in range(2): env.step(act)
torch.cuda.synchronize(); t0=time.time()
K=10
for _ in range(K): env.step(act)
torch.cuda.synchronize(); dt=(time.time()-t0)/K
print("env step ms", dt*1e3, "env-steps/s", 4096/dt)
