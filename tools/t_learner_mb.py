"""4 fp32 PPO minibatch steps at 16 384 rows (the bench shape), for kernel traces"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apex_amd import engine
from apex_amd.vecenv import MIRRORED_OBS, MIRRORED_ACTS, CLOCK_INDS
dev = torch.device("cuda", 0)
L = engine.PPOLearner(50, 10, 256, dev, 0.2231, mirrored_obs=MIRRORED_OBS, mirrored_acts=MIRRORED_ACTS, clock_inds=CLOCK_INDS)
g = torch.Generator(device="cpu"); g.manual_seed(0)
L.actor.params.copy_((torch.randn(L.actor.n, generator=g) * 0.05).to(dev)); L.critic.params.copy_((torch.randn(L.critic.n, generator=g) * 0.05).to(dev))
B = 16384
obs = torch.randn(B, 50, generator=g).to(dev); act = (torch.randn(B, 10, generator=g) * 0.3).to(dev); ret = torch.randn(B, generator=g).to(dev); adv = torch.randn(B, generator=g).to(dev)
mu = L.old_means(obs)
import time
for _ in range(4):
    L.minibatch(obs, act, ret, adv, mu, sync=False)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(24):
    L.minibatch(obs, act, ret, adv, mu, sync=False)
torch.cuda.synchronize(); print("ms per minibatch %.3f" % ((time.time() - t0) / 24 * 1e3))
