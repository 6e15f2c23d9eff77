"""Learner kernels for a PMC pass: 20 actor forwards at 16 384 rows (fused 3-layer kernel), 4 PPO minibatch steps (all fp32 GEMM flavours)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apex_amd import engine
from apex_amd.vecenv import MIRRORED_OBS, MIRRORED_ACTS, CLOCK_INDS
dev = torch.device("cuda", 0)
L = engine.PPOLearner(50, 10, 256, dev, 0.2231, mirrored_obs=MIRRORED_OBS, mirrored_acts=MIRRORED_ACTS, clock_inds=CLOCK_INDS)
g = torch.Generator(device="cpu"); g.manual_seed(0)
L.actor.params.copy_((torch.randn(L.actor.n, generator=g) * 0.05).to(dev)); L.critic.params.copy_((torch.randn(L.critic.n, generator=g) * 0.05).to(dev))
B = 16384
obs = torch.randn(B, 50, generator=g).to(dev); obs[:, 46] = torch.sin(torch.arange(B, device=dev) * 0.3); obs[:, 47] = torch.cos(torch.arange(B, device=dev) * 0.3)
act = (torch.randn(B, 10, generator=g) * 0.3).to(dev); ret = torch.randn(B, generator=g).to(dev); adv = torch.randn(B, generator=g).to(dev)
for _ in range(20):
    L.actor.forward(obs, L.obs_mean, L.obs_std)
mu = L.old_means(obs)
for _ in range(4):
    L.minibatch(obs, act, ret, adv, mu, sync=False)
torch.cuda.synchronize()
