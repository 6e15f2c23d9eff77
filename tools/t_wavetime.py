"""Per-wave duration of env_step_kernel (experiment build: make -C apex_amd/csrc VARIANT=wavetime EXTRA=-DAPX_WAVETIME; APX_LIB=apex_amd/lib/libapx_wavetime.so).
The launch lasts as long as its slowest wave (one wave per SIMD, no second round): how far is the slowest wave from the typical one, and what does it have?"""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apex_amd.vecenv import CassieVecEnv
env = CassieVecEnv(n_envs=4096, seed=0)
env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(0)
env.get_field("wavetime")
for t in range(24):
    act = torch.randn(4096, 10, device="cuda", generator=g) * 0.2
    env.step(act)
    if t % 4 == 3:
        w = env.get_field("wavetime").cpu().numpy()[:1024]      # rows = waves: cycles, substeps with leg-leg rows, with a limit row, sum of active contact slots (all since the last read: 4 launches, cycles of the last)
        c = w[:, 0]
        print("step %2d: cycles min %.0f  median %.0f  p90 %.0f  p99 %.0f  max %.0f  (max / median %.3f)" % (t, c.min(), np.median(c), np.percentile(c, 90), np.percentile(c, 99), c.max(), c.max() / np.median(c)))
        A = np.stack([np.ones(len(c)), w[:, 1], w[:, 2], w[:, 3]], 1)
        coef, *_ = np.linalg.lstsq(A, c, rcond=None)
        print("         least squares: cycles = %.0f + %.0f x (substeps with leg-leg rows) + %.0f x (substeps with a limit row) + %.0f x (active contact slots); mean features %.1f %.1f %.1f; slowest wave %s" % (
            coef[0], coef[1], coef[2], coef[3], w[:, 1].mean(), w[:, 2].mean(), w[:, 3].mean(), w[np.argmax(c)].tolist()))
