"""Per-wave duration of env_step_kernel (experiment build: make -C apex_amd/csrc VARIANT=wavetime EXTRA=-DAPX_WAVETIME; APX_LIB=apex_amd/lib/libapx_wavetime.so).
The launch lasts as long as its slowest wave (one wave per SIMD, no second round): how far is the slowest wave from the typical one, and what does it have?"""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apex_amd.vecenv import CassieVecEnv
env = CassieVecEnv(n_envs=4096, seed=0)
env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(0)
for t in range(24):
    act = torch.randn(4096, 10, device="cuda", generator=g) * 0.2
    env.step(act)
    if t % 4 == 3:
        w = env.get_field("wavetime")[:, 0].cpu().numpy()[::4]
        sat = env.get_field("ints_bits").view(torch.int32)[:, 6].cpu().numpy()
        print("step %2d: waves %d  cycles min %.0f  median %.0f  p90 %.0f  p99 %.0f  max %.0f  (max / median %.3f)" % (t, len(w), w.min(), np.median(w), np.percentile(w, 90), np.percentile(w, 99), w.max(), w.max() / np.median(w)))
