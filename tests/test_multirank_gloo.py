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


# ---------------------------------------------------------------------------------------------------------------------------------
# apx_ppo_epoch (an epoch's optimiser steps as one launch) has no place for a gradient all-reduce: with a process group PPO.update must take the per-step launches even when
# the epoch kernel was asked for.  Two gloo ranks run PPO.update on the EMULATED kernel sources (tests/test_kernel_emulation_learner.py's redirection, set by hand in the
# worker) with epoch_kernel=True on different halves of a batch: no launch of ppo_small_epoch_kernel on either rank, the per-step kernels ran, both ranks end with
# bit-identical parameters (every step's gradient was all-reduced), and the decision the bench line prints says "not in use".
def _epoch_exclusion_worker(rank, world, port, out):
    import contextlib
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    from apex_amd import _lib, engine
    from apex_amd.ppo import PPO
    from tests import test_kernel_emulation_learner as E
    from golden_util import epoch_case_inputs
    lib = E.emulated_library(); lib.apx_emul_set_workgroups(0)
    _lib._lib = lib
    engine._need_gpu = lambda *ts: None; engine._stream = lambda: None
    ns = E._NoStream
    torch.cuda.current_stream = lambda *a, **k: ns(); torch.cuda.Stream = lambda *a, **k: ns(); torch.cuda.Event = lambda *a, **k: ns()
    torch.cuda.stream = lambda s: contextlib.nullcontext(); torch.cuda.synchronize = lambda *a, **k: None
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    inp = epoch_case_inputs(0)
    args = dict(gamma=0.99, lam=0.95, lr=1e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=64, epochs=1, num_steps=2 * 64 * world, max_traj_len=400,
                max_grad_norm=0.05, mirror=True, std_dev=-1.5, seed=0, epoch_kernel=True, prepare_resets=False)
    algo = PPO(args, "/tmp/apx_test_unused", E._GridOnlyEnv(dev), rank=rank, world_size=world, group=dist.group.WORLD)
    L_ = algo.learner
    L_.actor.load_list(inp["actor"]); L_.critic.load_list(inp["critic"])
    T, N = algo.T, algo.N
    assert (T, N) == (2, 64) and algo.epoch_kernel and algo.dist_on
    r2 = np.random.RandomState(40 + rank)                                    # each rank its own shard
    obs = r2.randn(T, N, 50).astype(np.float32); ph = r2.rand(T, N) * 2 * np.pi
    obs[..., 46] = np.sin(ph); obs[..., 47] = np.cos(ph)
    algo.b_obs.copy_(torch.tensor(obs))
    algo.b_mu.copy_(L_.actor.forward(algo.b_obs.view(T * N, 50), L_.obs_mean, L_.obs_std).view(T, N, 10))
    algo.b_act.copy_(algo.b_mu + torch.tensor(r2.randn(T, N, 10).astype(np.float32)) * algo.fixed_std)
    algo.b_val.copy_(torch.tensor(r2.randn(T, N).astype(np.float32)))
    ret = torch.tensor(r2.randn(T, N).astype(np.float32))
    perm = torch.tensor(np.random.RandomState(5).permutation(T * N).astype(np.int64))
    algo.perm_fn = lambda epoch: perm
    before = {k: E.launches(k) for k in ("ppo_small_epoch_kernel", "cooperative", "mlp_fused_fwd_kernel", "bwd_head_kernel")}
    in_use = algo.epoch_kernel_in_use(64)
    losses, kl, epochs_run = algo.update(ret)
    d = {k: E.launches(k) - v for k, v in before.items()}
    flat = torch.cat([L_.actor.params, L_.critic.params]).clone()
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    if rank == 0:
        out.put((in_use, d, bool(torch.equal(both[0], both[1])), float((flat - torch.tensor(np.concatenate([np.asarray(p, np.float32).ravel() for p in inp["actor"] + inp["critic"]]))).abs().max()), L_.t, epochs_run))
    dist.barrier(); dist.destroy_process_group()


def test_epoch_kernel_is_excluded_when_a_process_group_is_given():
    import pytest
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("no host clang++ (kernel emulation)")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_epoch_exclusion_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    in_use, d, same, moved, adam_t, epochs_run = q.get(timeout=900)
    [p.join(timeout=120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert in_use is False
    assert d["ppo_small_epoch_kernel"] == 0 and d["cooperative"] == 0, d            # the one-launch path was NOT taken ...
    assert d["mlp_fused_fwd_kernel"] > 0 and d["bwd_head_kernel"] >= 2, d             # ... the per-step launches were (2 optimiser steps of 64 rows)
    assert same and moved > 0 and adam_t == 2 and epochs_run == 1


# ---------------------------------------------------------------------------------------------------------------------------------
# bench.py at N = 2 on the CPU: two gloo ranks run bench.main() end to end on the EMULATED kernels at the tiny test shape (APX_BENCH_TINY; APX_BENCH_SHARE_GPU=1 = the gloo
# form of the launch contract): env shards by rank, gradient all-reduce per optimiser step, moments, the MAX-over-ranks time, the per-rank table, ONE JSON line on rank 0.
def _bench_worker(rank, world, port, workload, out):
    import contextlib, io, json, sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), APX_BENCH_SHARE_GPU="1", APX_BENCH_TINY="1")
    from apex_amd import _lib, engine, vecenv
    from tests import test_kernel_emulation_learner as E
    lib = E.emulated_library(); lib.apx_emul_set_workgroups(0)
    _lib._lib = lib
    engine._need_gpu = lambda *ts: None; engine._stream = lambda: None; vecenv._stream = lambda: None
    vecenv._device = lambda index: torch.device("cpu"); vecenv._on_device = lambda t: True
    ns = E._NoStream
    torch.cuda.current_stream = lambda *a, **k: ns(); torch.cuda.Stream = lambda *a, **k: ns(); torch.cuda.Event = lambda *a, **k: ns()
    torch.cuda.stream = lambda s: contextlib.nullcontext(); torch.cuda.synchronize = lambda *a, **k: None; torch.cuda.set_device = lambda *a, **k: None
    import bench
    extra = ["--n_envs", "64", "--rollout_len", "2", "--minibatch", "64", "--epochs", "1"] if workload == "cassie_ppo" else ["--epochs", "1"]
    sys.argv = ["bench.py", "--gpus", str(world), "--steps", "1", "--warmup", "0", "--workload", workload, "--no_cpu_baseline"] + extra
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    lines = [ln for ln in buf.getvalue().split("\n") if ln.startswith("{")]
    out.put((rank, lines))


def _bench_two_ranks(workload):
    import json
    import pytest
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("no host clang++ (kernel emulation)")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, workload, q)) for r in range(2)]
    [p.start() for p in procs]
    got = dict(q.get(timeout=900) for _ in range(2))
    [p.join(timeout=120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert got[1] == [] and len(got[0]) == 1                      # rank 0 prints the one line
    return json.loads(got[0][0])


def test_bench_headline_line_at_two_ranks():
    d = _bench_two_ranks("cassie_ppo")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["parallelism"].startswith("dp2") and d["config"]["tiny_test_shape"] is True
    assert abs(d["value"] - 2 * 64 * 2 / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]              # whole-job aggregate: both ranks' env steps over the MAX time
    c = d["collectives"]
    assert c["rccl_ranks_seen"] == 2 and c["backend"] == "gloo" and c["gradient_floats"] == 160523 and len(c["per_rank"]["sample_s"]) == 2
    assert d["config"]["optimiser_steps_as_one_launch_per_epoch"] is False


def test_bench_recurrent_line_at_two_ranks():
    d = _bench_two_ranks("cassietraj_recurrent")
    assert d["n_gpus"] == 2 and d["config"]["parallelism"].startswith("dp2") and abs(d["value"] - 2 * 64 * 3 / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]
    assert d["collectives"]["rccl_ranks_seen"] == 2 and len(d["collectives"]["per_rank"]["optimize_s"]) == 2 and d["optimiser_step_us"] > 0
