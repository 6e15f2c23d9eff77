import sys, torch
sys.path.insert(0, "/root/repo")
from apex_amd.vecenv import CassieVecEnv
n, T = 1024, 300
env = CassieVecEnv(n_envs=n, seed=21); env.reset()
g = torch.Generator(device=env.device); g.manual_seed(0)
for t in range(T):
    env.step(0.2 * torch.randn(n, 10, device=env.device, generator=g))
flags, cnt = env.saturation()
print("passes with saturation / all passes:", float(cnt.sum()) / (n * T * 50), "envs ever saturated:", float((cnt > 0).float().mean()), "flag histogram:", [(int(f), int((flags == f).sum())) for f in flags.unique()])
