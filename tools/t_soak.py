"""Soak + statistical parity: random-policy rollouts with auto-reset, HIP env (4096 envs) vs fp64 oracle (96 envs):
no NaN/inf, episode-length and per-step-reward statistics of the two must agree."""
import sys, os, numpy as np, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apex_amd.vecenv import CassieVecEnv
from oracle import sim as S
T, N, NO = 600, 4096, 96
g = CassieVecEnv(n_envs=N, seed=21)
g.reset()
gen = torch.Generator(device="cuda"); gen.manual_seed(5)
ep_len = torch.zeros(N, device="cuda"); lens = []; rsum = 0.0; bad = 0
t0 = time.time()
for t in range(T):
    act = 0.2 * torch.randn(N, 10, device="cuda", generator=gen)
    obs, rew, done, fin = g.step(act)
    ep_len += 1
    d = done != 0
    bad += int((~torch.isfinite(obs)).sum() + (~torch.isfinite(rew)).sum())
    rsum += float(rew.sum())
    if bool(d.any()):
        lens.append(ep_len[d].cpu().numpy()); ep_len[d] = 0
torch.cuda.synchronize()
lens = np.concatenate(lens)
print("HIP   : %d env steps in %.1fs, non-finite values %d, episodes %d, mean len %.2f (sd %.2f), mean reward/step %.4f" % (T * N, time.time() - t0, bad, lens.size, lens.mean(), lens.std(), rsum / (T * N)))
rng = np.random.RandomState(7); olens = []; orsum = 0.0; osteps = 0
for i in range(NO):
    e = S.OracleEnv(dyn_rand=True, seed=21, env_id=i); e.reset(); L = 0
    for t in range(T // 2):
        _, r, d = e.step(0.2 * rng.randn(10)); L += 1; orsum += r; osteps += 1
        if d: olens.append(L); L = 0; e.reset()
olens = np.array(olens)
print("oracle: %d env steps, episodes %d, mean len %.2f (sd %.2f), mean reward/step %.4f" % (osteps, olens.size, olens.mean(), olens.std(), orsum / osteps))
se = np.sqrt(lens.var() / lens.size + olens.var() / olens.size)
print("difference of mean episode length: %.2f (%.1f standard errors)" % (lens.mean() - olens.mean(), (lens.mean() - olens.mean()) / se))
