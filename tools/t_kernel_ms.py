import sys, torch
sys.path.insert(0, "/root/repo")
from apex_amd.vecenv import CassieVecEnv
e = CassieVecEnv(n_envs=4096, seed=1); e.reset()
a = torch.zeros(4096, 10, device=e.device)
for _ in range(3): e.step(a)
e.kernel_timing(True); e.kernel_timing_read()
for _ in range(20): e.step(torch.randn(4096, 10, device=e.device) * 0.1)
ms, n = e.kernel_timing_read(); print("env_step_kernel ms", round(ms / n, 4))
