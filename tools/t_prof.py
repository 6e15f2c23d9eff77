import sys, torch, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apex_amd.vecenv import CassieVecEnv
env = CassieVecEnv(n_envs=4096, seed=0)
env.reset()
torch.manual_seed(0); act = torch.randn(4096,10,device="cuda")*0.2
for _ in range(2): env.step(act, auto_reset=False)
buf = torch.zeros(4096*128, device='cuda')
from apex_amd import _lib; from apex_amd.engine import _p, _stream
lib=_lib.load()
lib.apx_env_get_field(env._h, b"prof", _p(buf), _stream())   # reset counters
torch.cuda.synchronize(); t0=time.time()
K=4
for _ in range(K): env.step(act, auto_reset=False)
torch.cuda.synchronize(); dt=(time.time()-t0)/K
lib.apx_env_get_field(env._h, b"prof", _p(buf), _stream())
p = buf[:12].cpu().numpy() / (K*50)
names=["io_model","tree_walk","factor","pgs_tail(z~)","finish+euler","rows(2 legs)","gram+warm","pgs_sweeps"]
print("env step ms %.1f"%(dt*1e3))
for n,v in zip(names,p): print("%-16s %9.0f cycles/substep"%(n,v))
print("total %.0f cycles/substep"%p[:8].sum()); print("whole sim_step_pd %.0f cycles/substep" % p[8])
