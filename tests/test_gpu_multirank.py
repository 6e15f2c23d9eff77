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


def _rec_worker(rank, world, port, q):
    """one rank of a recurrent-PPO iteration: its own env shard (different trajectory counts per rank), shared cuda:0, gloo"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from apex_amd.vecenv import CassieVecEnv
    from apex_amd.ppo_recurrent import RecurrentPPO
    from apex_amd import dist as adist
    n = 64
    env = CassieVecEnv(n_envs=n, seed=5, max_traj_len=10 - 3 * rank, env_id_base=adist.shard_env_base(rank, n), env_name="CassieTraj-v0")
    args = dict(gamma=0.99, lam=0.95, lr=1e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=48, epochs=2, num_steps=20 * n * world, max_traj_len=10 - 3 * rank,
                max_grad_norm=0.05, mirror=True, seed=0, allow_short_rollout=True)
    algo = RecurrentPPO(args, "/tmp/apx_test_unused", env, rank=rank, world_size=world, group=dist.group.WORLD, hidden=64, layers=2)
    algo.init_networks(0); algo.normalization_params(1000)
    out = algo.iteration()
    q.put((rank, len(algo.trajectories()), algo.learner.actor.params.cpu().numpy(), algo.learner.obs_mean.cpu().numpy(), np.asarray(out["losses"])))
    dist.barrier(); dist.destroy_process_group()


def test_recurrent_ppo_two_ranks_agree_on_the_step_count():
    """Recurrent PPO with N > 1 (SURVEY.md section 8e for row f1): the ranks hold different numbers of trajectories, the number of optimiser steps
    per epoch is agreed by a MAX all-reduce and a rank that has run out contributes zero gradients: no hang, identical parameters and
    observation statistics on both ranks."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_rec_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda r: r[0])
    [p.join(timeout=120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (_, n0, a0, m0, l0), (_, n1, a1, m1, l1) = res
    assert n0 != n1                                                    # different episode limits -> different trajectory counts
    assert -(-n0 // 48) != -(-n1 // 48)                               # ... and different minibatch counts: the zero-gradient path ran
    np.testing.assert_array_equal(a0, a1); np.testing.assert_array_equal(m0, m1)
    assert np.isfinite(a0).all() and np.isfinite(l0).all() and np.isfinite(l1).all()


def test_eight_ranks_bench_driver_on_a_shared_gpu(tmp_path):
    """The driver's multi-GPU command line (python -m torch.distributed.run --nproc-per-node 8 bench.py --gpus 8 ...) with all eight ranks on
    cuda:0 over gloo (APX_BENCH_SHARE_GPU=1, small env shards): env shards with disjoint RNG streams, observation / advantage moments, one
    gradient all-reduce per optimiser step, max-over-ranks timing, ONE JSON line from rank 0."""
    import json, subprocess, sys
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ); env["APX_BENCH_SHARE_GPU"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(repo, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1", "--n_envs", "256", "--rollout_len", "8", "--minibatch", "512", "--no_cpu_baseline"]
    r = subprocess.run(cmd, cwd=repo, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["value"] > 0 and d["config"]["parallelism"].startswith("dp8")
    assert abs(d["value"] - 8 * 256 * 8 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3


def test_rccl_path_at_world_size_one(tmp_path):
    """the RCCL branch itself, executed on one GPU.  APX_FORCE_DIST=1 + torch.distributed.run --nproc-per-node 1 sends bench.py through
    init_process_group("nccl", device_id=...), the flat-gradient all-reduce on the 160 523-float device tensor (one per optimiser step), the advantage /
    observation moment all-reduces and the merged per-epoch scalar all-reduce, with the event timing of apex_amd/dist.py around them."""
    import json, subprocess, sys
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ); env["APX_FORCE_DIST"] = "1"; env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0"); env.pop("APX_BENCH_SHARE_GPU", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(repo, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--n_envs", "1024", "--rollout_len", "16", "--minibatch", "2048", "--no_cpu_baseline"]
    r = subprocess.run(cmd, cwd=repo, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    c = d["collectives"]
    assert c["backend"] == "nccl" and c["rccl_ranks_seen"] == 1 and c["gradient_floats"] == 160523
    # 3 epochs x 8 minibatches of 2048 gradient all-reduces + 3 per-epoch scalar all-reduces per iteration (fewer only if the KL test stops an epoch loop early)
    assert 9 <= c["allreduce_calls_per_step"] <= 27
    assert c["allreduce_ms_per_step"] > 0.0
    assert d["n_gpus"] == 1 and d["value"] > 0


@pytest.mark.gpu
def test_rccl_path_at_world_size_one_recurrent(tmp_path):
    """the RCCL branch of the RECURRENT workload (BASELINE configs[3]) on one GPU - APX_FORCE_DIST=1 + torch.distributed.run --nproc-per-node 1:
    init_process_group("nccl"), the MAX all-reduce that agrees on the number of optimiser steps of an epoch, the flat LSTM-gradient all-reduce per step, the advantage /
    observation moments and the per-epoch scalar all-reduce, on device tensors."""
    import json, subprocess, sys
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ); env["APX_FORCE_DIST"] = "1"; env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0"); env.pop("APX_BENCH_SHARE_GPU", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(repo, "bench.py"), "--workload", "cassietraj_recurrent", "--gpus", "1", "--steps", "1", "--warmup", "1", "--n_envs", "256", "--epochs", "1", "--no_cpu_baseline"]
    r = subprocess.run(cmd, cwd=repo, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    c = d["collectives"]
    assert c["backend"] == "nccl" and c["rccl_ranks_seen"] == 1 and c["gradient_floats"] > 400000      # two LSTM(2 x 128) networks
    assert c["allreduce_calls_per_step"] >= 2 and c["allreduce_ms_per_step"] > 0.0
    assert d["n_gpus"] == 1 and d["value"] > 0 and "recurrent" in d["metric"]
