"""Saturation statistics per env step (bench-like random policy): how many env steps per launch hold a forward pass beyond the kernel's row caps, which caps, how many
passes of such a step, how long the bursts are."""
import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from apex_amd.vecenv import CassieVecEnv
n, T = 4096, 200
sig = float(sys.argv[1]) if len(sys.argv) > 1 else 0.223
env = CassieVecEnv(n_envs=n, seed=21); env.reset()
g = torch.Generator(device=env.device); g.manual_seed(0)
prev_f, prev_c = env.saturation(); prev_c = prev_c.clone(); prev_f = prev_f.clone()
steps_with = 0; env_steps = 0; passes = 0; kinds = np.zeros(16, dtype=np.int64); per_step = []; burst = np.zeros(n, dtype=np.int64); bursts = []
for t in range(T):
    o, r, d, _ = env.step(sig * torch.randn(n, 10, device=env.device, generator=g))
    f, c = env.saturation()
    dc = (c - prev_c).cpu().numpy(); newf = f.cpu().numpy()
    k = int((dc > 0).sum()); per_step.append(k); steps_with += k > 0; env_steps += k; passes += int(dc.sum())
    for e in np.nonzero(dc > 0)[0]: kinds[newf[e] & 15] += 1
    ended = (dc == 0) & (burst > 0)
    bursts += list(burst[ended]); burst[ended] = 0; burst[dc > 0] += 1
    prev_c = c.clone()
print("sigma", sig, "launches with >= 1 saturated env step: %d / %d" % (steps_with, T), "saturated env steps per launch: mean %.2f max %d" % (np.mean(per_step), max(per_step)))
print("saturated env steps / all env steps: %.2e, passes / all passes: %.2e, passes per saturated env step: %.1f" % (env_steps / (n * T), passes / (n * T * 50.0), passes / max(1, env_steps)))
print("cumulative flag sets of the saturated env steps (bit 0 contacts, 1 limits, 2 body-floor, 3 leg-leg):", {int(i): int(v) for i, v in enumerate(kinds) if v})
print("burst lengths (consecutive env steps):", np.bincount(np.array(bursts, dtype=np.int64))[:12] if bursts else [])
