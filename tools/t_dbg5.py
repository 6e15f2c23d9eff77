import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apex_amd.vecenv import CassieVecEnv
g = CassieVecEnv(n_envs=64, dynamics_randomization=False, seed=7)
torch.cuda.synchronize(); print("created", flush=True)
g.substep(); torch.cuda.synchronize(); print("substep ok", flush=True)
obs = g.reset(); torch.cuda.synchronize(); print("reset ok", flush=True)
g.substep(); torch.cuda.synchronize(); print("substep2 ok", flush=True)
a = torch.zeros(64, 10, device='cuda')
g.step(a, auto_reset=False); torch.cuda.synchronize(); print("step ok", flush=True)
