"""N>1 path on CPU: world_size-2 gloo processes run the collective protocol of apex_amd/dist.py with the numpy
oracle standing in for the GPU kernels, and the result must equal the single-process computation on the union
batch (gradient averaging == union-minibatch SGD; moment all-reduce == union normalisation)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from apex_amd import dist as adist
from oracle import learner as L
from tools.refprobe.common import MIRRORED_OBS_FULL_CLOCK, MIRRORED_ACTS


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class _Rec:
    def step(self, params, grads):
        self.g = grads
        return params


def _batch(seed, B, H=16):
    rng = np.random.RandomState(seed)
    shapes = [(H, 50), (H,), (H, H), (H,), (10, H), (10,)]
    cshapes = [(H, 50), (H,), (H, H), (H,), (1, H), (1,)]
    Wa = [rng.randn(*s) * 0.2 for s in shapes]; Wo = [w + rng.randn(*w.shape) * 0.01 for w in Wa]
    Wc = [rng.randn(*s) * 0.2 for s in cshapes]
    obs = rng.randn(B, 50); ph = rng.rand(B) * 2 * np.pi; obs[:, 46] = np.sin(ph); obs[:, 47] = np.cos(ph)
    return Wa, Wo, Wc, obs, rng.randn(B, 10) * 0.3, rng.randn(B, 1), rng.randn(B, 1)


def _grads(Wa, Wo, Wc, obs, act, ret, adv):
    ra, rc = _Rec(), _Rec()
    scal, _, _ = L.ppo_update(Wa, Wo, Wc, ra, rc, obs, act, ret, adv, np.zeros(50), np.ones(50), np.exp(-1.5), grad_clip=1e9,
                              M_obs=L.mirror_matrix(MIRRORED_OBS_FULL_CLOCK), M_act=L.mirror_matrix(MIRRORED_ACTS))
    return scal, np.concatenate([g.reshape(-1) for g in ra.g + rc.g])


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    Wa, Wo, Wc, obs, act, ret, adv = _batch(0, 64)
    sl = slice(rank * 32, (rank + 1) * 32)                       # this rank's half of the union minibatch
    scal, g = _grads(Wa, Wo, Wc, obs[sl], act[sl], ret[sl], adv[sl])
    flat = torch.tensor(g)
    two = flat.clone()
    adist.allreduce_mean_(flat)                                   # THE gradient collective
    # ... and in the two-half asynchronous form of PPO.update (round 5: the actor's half travels while the critic's backward runs): two async all-reduces on views
    # of the flat buffer, joined and averaged once
    na = two.numel() // 2
    h = [adist.allreduce_begin(two[:na])]
    h.append(adist.allreduce_begin(two[na:]))
    adist.allreduce_end(h, two, world)
    assert torch.equal(two, flat)
    sc = torch.tensor(scal); adist.allreduce_mean_(sc)
    a = (ret[sl] - adv[sl]).reshape(-1)
    mom = torch.tensor([a.sum(), (a * a).sum(), float(a.size)], dtype=torch.float64)
    adist.allreduce_moments(mom)
    if rank == 0:
        out.put((flat.numpy(), sc.numpy(), mom.numpy()))
    dist.barrier(); dist.destroy_process_group()


def test_two_rank_protocol_equals_union_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    flat, sc, mom = q.get(timeout=120)
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    Wa, Wo, Wc, obs, act, ret, adv = _batch(0, 64)
    scal, g = _grads(Wa, Wo, Wc, obs, act, ret, adv)
    np.testing.assert_allclose(flat, g, rtol=1e-9, atol=1e-12)                 # averaged grads == union-batch grads
    np.testing.assert_allclose(sc[[0, 2, 3, 4, 5]], scal[[0, 2, 3, 4, 5]], rtol=1e-9)   # mean-type scalars average exactly
    a = (ret - adv).reshape(-1)
    mean, std = adist.adv_stats_from_moments(mom)
    assert abs(mean - a.mean()) < 1e-12 and abs(std - a.std(ddof=1)) < 1e-12


def test_sharding_arithmetic():
    assert adist.shard_env_base(3, 4096) == 12288
    assert adist.rollout_len(5096, 4096, 1) == 2 and adist.rollout_len(131072 * 8, 4096, 8) == 32
    assert adist.rollout_len(10, 4096, 8) == 1
