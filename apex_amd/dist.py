"""Collective protocol of the N>1 path, backend-agnostic so that it runs over RCCL on GPUs and over
gloo in the CPU tests.  Env shards never talk to each other; these are the only exchanges:
  * one all-reduce of the flat (actor | critic) gradient per optimiser step, averaged -> identical to single-process
    SGD on the union minibatch (equal per-rank minibatch sizes);
  * the three advantage moments (sum, sum of squares, count) before rl/algos/ppo.py:396's normalisation;
  * 101 observation moments at start-up (rl/envs/normalize.py:48);
  * per-epoch scalars (KL for the early stop, logged losses)."""
import torch
import torch.distributed as dist


def shard_env_base(rank, n_envs):
    """Global index of env 0 of this rank's shard = Philox stream offset (apx_env_cfg.env_id_base)."""
    return rank * n_envs


def rollout_len(num_steps, n_envs, world):
    """Steps per env per iteration so that the job samples >= num_steps in total (rl/algos/ppo.py:205)."""
    return max(1, -(-num_steps // (n_envs * world)))


# optional timing of the gradient all-reduce (bench.py at N > 1): event pairs around every allreduce_mean_ on the current stream, drained by timing_read()
_timing = {"on": False, "events": []}


def timing(enable):
    _timing["on"] = bool(enable); _timing["events"] = []


def timing_read():
    """(total ms, calls) of the all-reduces recorded since timing(True)"""
    ev = _timing["events"]; _timing["events"] = []
    if ev:
        ev[-1][1].synchronize()
    return sum(a.elapsed_time(b) for a, b in ev), len(ev)


def allreduce_mean_(flat, group=None, world=None):
    world = world or dist.get_world_size(group)
    rec = _timing["on"] and flat.is_cuda
    if rec:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    dist.all_reduce(flat, group=group)
    flat /= world
    if rec:
        e1.record(); _timing["events"].append((e0, e1))
    return flat


def allreduce_begin(flat, group=None):
    """Start the sum all-reduce of `flat` (async handle): the collective waits for the work already enqueued on the current stream and runs on the backend's own
    stream while the caller keeps enqueueing (PPO.update: the actor's gradient travels while the critic's backward runs)."""
    rec = _timing["on"] and flat.is_cuda
    e0 = None
    if rec:
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
    return dist.all_reduce(flat, group=group, async_op=True), e0


def allreduce_end(handles, flat, world):
    """Join the handles of allreduce_begin (the current stream waits for the collectives) and turn the sums in `flat` into means."""
    for h, _ in handles:
        h.wait()
    flat /= world
    if handles and handles[0][1] is not None:
        e1 = torch.cuda.Event(enable_timing=True); e1.record(); _timing["events"].append((handles[0][1], e1))
    return flat


def adv_stats_from_moments(mom):
    """(mean, unbiased std) from (sum a, sum a^2, n) — torch's .std() is unbiased (ppo.py:396)."""
    s, ss, n = (float(x) for x in mom)
    mean = s / n
    var = max(ss - n * mean * mean, 0.0) / max(n - 1.0, 1.0)
    return mean, var ** 0.5


def allreduce_moments(mom, group=None):
    if group is not None or (dist.is_available() and dist.is_initialized()):
        dist.all_reduce(mom, group=group)
    return mom
