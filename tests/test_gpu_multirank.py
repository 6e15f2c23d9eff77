"""GPU: the N > 1 data path of the HIP learner with two processes sharing cuda:0 over gloo: each rank runs
apx_ppo_minibatch(grad_only) on its half of a minibatch, the flat gradient is all-reduced (mean) and apx_clip_adam applies it; the
parameters must equal ONE rank stepping on the union minibatch (same maths as the reference's single-process SGD)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _mk_learner(dev):
    from apex_amd import engine
    from apex_amd.vecenv import MIRRORED_OBS, MIRRORED_ACTS, CLOCK_INDS
    L = engine.PPOLearner(50, 10, 256, dev, float(np.exp(-1.5)), mirrored_obs=MIRRORED_OBS, mirrored_acts=MIRRORED_ACTS, clock_inds=CLOCK_INDS)
    g = torch.Generator(device="cpu"); g.manual_seed(3)
    L.actor.params.copy_((torch.randn(L.actor.n, generator=g) * 0.05).to(dev)); L.critic.params.copy_((torch.randn(L.critic.n, generator=g) * 0.05).to(dev))
    L.obs_mean.copy_((torch.randn(50, generator=g) * 0.1).to(dev)); L.obs_std.copy_((0.8 + 0.4 * torch.rand(50, generator=g)).to(dev))
    return L


def _batch(dev, B):
    g = torch.Generator(device="cpu"); g.manual_seed(11)
    obs = torch.randn(B, 50, generator=g) * 0.5
    obs[:, 46] = torch.sin(torch.arange(B) * 0.3); obs[:, 47] = torch.cos(torch.arange(B) * 0.3)
    act = torch.randn(B, 10, generator=g) * 0.3; ret = torch.randn(B, generator=g); adv = torch.randn(B, generator=g)
    return [x.to(dev).contiguous() for x in (obs, act, ret, adv)]


def _steps(L, obs, act, ret, adv, n_steps, world=1):
    from apex_amd import dist as adist
    mu = L.old_means(obs)
    for _ in range(n_steps):
        if world > 1:
            L.minibatch(obs, act, ret, adv, mu, grad_only=True, sync=False)
            adist.allreduce_mean_(L.grad_flat, world=world)
            L.apply_grads(scale=1.0)
        else:
            L.minibatch(obs, act, ret, adv, mu, sync=False)
    torch.cuda.synchronize()


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0); torch.cuda.set_device(0)
    L = _mk_learner(dev)
    obs, act, ret, adv = _batch(dev, 512)
    sl = slice(rank * 256, (rank + 1) * 256)
    _steps(L, obs[sl].contiguous(), act[sl].contiguous(), ret[sl].contiguous(), adv[sl].contiguous(), 3, world=world)
    if rank == 0:
        q.put((L.actor.params.cpu().numpy(), L.critic.params.cpu().numpy()))
    dist.barrier(); dist.destroy_process_group()


def test_two_ranks_equal_union_minibatch_on_the_hip_learner():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    a2, c2 = q.get(timeout=300)
    [p.join(timeout=120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    dev = torch.device("cuda", 0)
    L = _mk_learner(dev)
    a0 = L.actor.params.cpu().numpy().copy()
    obs, act, ret, adv = _batch(dev, 512)
    _steps(L, obs, act, ret, adv, 3)
    a1, c1 = L.actor.params.cpu().numpy(), L.critic.params.cpu().numpy()
    assert np.abs(a1 - a0).max() > 1e-5                               # the steps moved the weights
    for x2, x1 in ((a2, a1), (c2, c1)):
        d = np.abs(x2 - x1)
        # Adam's first steps move every weight by ~lr * sign(g): identical up to fp32 summation order, except a vanishing fraction at g ~ 0
        assert (d > 2e-6).mean() < 2e-3 and d.max() < 4e-4, ((d > 2e-6).mean(), d.max())
