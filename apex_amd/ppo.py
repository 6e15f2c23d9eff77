"""PPO driver for the batched engine — the host-side mirror of the reference's rl/algos/ppo.py.

Same control flow as PPO.train (ppo.py:347-505): sample -> returns -> normalised advantages -> `epochs` passes of
random minibatches with the clipped-ratio / value / mirror losses and KL early stop -> logging / best-checkpoint save,
but the Ray fan-out of one-env workers (ppo.py:188-237) becomes ONE lock-step batch of N envs per GPU and every array
stays in HBM.  One process per GPU; with world_size > 1 the env shards are independent and the only collectives are
one RCCL all-reduce of the 160 523-float gradient per optimiser step plus scalar moments.

Rollout layout: [T, N] grids (env-per-column).  An episode that ends inside the grid bootstraps with
(not done) * V(s_next) exactly like ppo.py:183-184; a column that is cut by the end of the grid bootstraps with
V(s_T) (the reference never cuts mid-episode because its workers sample whole episodes; with a fixed grid the cut is
treated like its max_traj_len truncation).
"""
import os
import time

import numpy as np
import torch

from . import dist as adist
from . import engine
from .vecenv import CassieVecEnv, MIRRORED_ACTS, MIRRORED_OBS, CLOCK_INDS


def episode_stats(rew, ended, ret0, len0):
    """Returns / lengths of the episodes that end inside a [T, N] rollout grid, without a per-step host round trip.
    rew [T, N] float, ended [T, N] bool, ret0 / len0 [N] = partial sums of the episodes still running at entry.
    -> (ep_returns, ep_lens) in (t, env) order, and the new partial sums.  Same numbers as accumulating per step and
    resetting at every end (rl/algos/ppo.py:166-186)."""
    T, N = rew.shape
    dev = rew.device
    cs = torch.cumsum(rew.double(), 0) + ret0.double()
    steps = torch.arange(1, T + 1, device=dev, dtype=torch.float64).view(T, 1) + len0.double()
    t_idx = torch.arange(T, device=dev).view(T, 1).expand(T, N)
    last_end = torch.cummax(torch.where(ended, t_idx, torch.full_like(t_idx, -1)), 0).values       # last end at or before t
    prev_end = torch.cat([torch.full((1, N), -1, device=dev, dtype=last_end.dtype), last_end[:-1]], 0)
    has_prev = prev_end >= 0
    gi = prev_end.clamp(min=0)
    zero = torch.zeros_like(cs)
    base_r = torch.where(has_prev, torch.gather(cs, 0, gi), zero)
    base_l = torch.where(has_prev, torch.gather(steps, 0, gi), zero)
    ep_rets = (cs - base_r)[ended].float()
    ep_lens = (steps - base_l)[ended].float()
    fin_end = last_end[-1]
    has = fin_end >= 0
    gl = fin_end.clamp(min=0).view(1, N)
    zn = torch.zeros(N, device=dev, dtype=torch.float64)
    ret1 = (cs[-1] - torch.where(has, torch.gather(cs, 0, gl).view(N), zn)).float()
    len1 = (steps[-1] - torch.where(has, torch.gather(steps, 0, gl).view(N), zn)).float()
    return ep_rets, ep_lens, ret1, len1


class PPO:
    def __init__(self, args, save_path, env, rank=0, world_size=1, group=None, hidden=256):
        self.gamma = args["gamma"]; self.lam = args["lam"]; self.lr = args["lr"]; self.eps = args["eps"]
        self.entropy_coeff = args["entropy_coeff"]; self.clip = args["clip"]
        self.minibatch_size = args["minibatch_size"]; self.epochs = args["epochs"]
        self.num_steps = args["num_steps"]; self.max_traj_len = args["max_traj_len"]
        self.grad_clip = args["max_grad_norm"]; self.mirror = args.get("mirror", True)
        self.fixed_std = float(np.exp(args.get("std_dev", -1.5)))
        self.env_name = args.get("env_name", "Cassie-v0")
        self.save_path = save_path
        self.env = env
        self.rank, self.world, self.group = rank, world_size, group
        self.dist_on = group is not None      # the collective path runs whenever a process group is given, world_size 1 included (APX_FORCE_DIST: the RCCL path on one GPU)
        self.device = env.device
        self.N = env.n_envs
        self.D = int(getattr(env, "obs_dim", 50))              # 50 (command_profile clock) or 55 (phase), x (history + 1)
        if getattr(env, "history", 0) and args.get("mirror", True):
            raise NotImplementedError("--history needs --not_mirror: the reference's mirror index lists cover a single frame (cassie.py:244,262-271)")
        # steps per env per iteration: the reference samples >= num_steps in total (ppo.py:205)
        self.T = adist.rollout_len(self.num_steps, self.N, self.world)
        self.learner = engine.PPOLearner(self.D, 10, hidden, self.device, self.fixed_std, lr=self.lr, eps=self.eps,
                                         clip=self.clip, entropy_coeff=self.entropy_coeff, grad_clip=self.grad_clip,
                                         mirrored_obs=list(getattr(env, "mirrored_obs", MIRRORED_OBS)) if self.mirror else None,
                                         mirrored_acts=MIRRORED_ACTS if self.mirror else None, clock_inds=list(getattr(env, "clock_inds", CLOCK_INDS)))      # env.clock_inds like rl/algos/ppo.py:307-310 (21, 22 with input_profile=min)
        self.total_steps = 0
        self.highest_reward = -1
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(int(args.get("seed", 0)) * 1000003 + rank)
        T, N = self.T, self.N
        f32 = dict(dtype=torch.float32, device=self.device)
        self.b_obs = torch.zeros(T, N, self.D, **f32); self.b_act = torch.zeros(T, N, 10, **f32)
        self.b_mu = torch.zeros(T, N, 10, **f32); self.b_rew = torch.zeros(T, N, **f32)
        self.b_val = torch.zeros(T, N, **f32); self.b_boot = torch.zeros(T, N, **f32)
        self.b_end = torch.zeros(T, N, dtype=torch.uint8, device=self.device)
        self.b_done = torch.zeros(T, N, dtype=torch.uint8, device=self.device)
        self.b_endb = torch.zeros(T, N, dtype=torch.bool, device=self.device)
        self.b_fin = torch.zeros(T, N, self.D, **f32)
        self.noise = torch.zeros(T, N, 10, **f32)
        self.prepare_resets = bool(args.get("prepare_resets", True)); self._side = None
        self.use_graph = bool(args.get("graph", False)) and not self.dist_on
        # small minibatches (the reference's CLI default 64, apex.py:242): the epoch's optimiser steps as ONE launch (apx_ppo_epoch) instead of 16 launches per step.
        # "auto" = whenever the minibatch is at most epoch_kernel_max_mb rows and this is a single-GPU run (a gradient all-reduce per step needs the per-step launches)
        ek = args.get("epoch_kernel") or os.environ.get("APX_PPO_EPOCH", "0")
        self.epoch_kernel = ek in (True, 1, "1", "auto", "on", "true")
        self.epoch_kernel_max_mb = int(args.get("epoch_kernel_max_mb", 256))
        self._graph = None
        if self.use_graph and hasattr(env, "set_refill"):
            env.set_refill(False)      # the captured rollout is a single-stream graph: no side-stream refill of the reset ring inside it (drains one in flight)
        self.ep_ret = torch.zeros(N, **f32); self.ep_len = torch.zeros(N, **f32)
        self.obs = None
        # parity mode: replay captured draw streams instead of the Philox generator.
        # noise_fn(t, out[N,10]) fills the action noise of rollout step t; perm_fn(epoch) -> int64[B] minibatch order;
        # trace, when a list, receives every minibatch's 6 scalars (device tensors).
        self.noise_fn = None; self.perm_fn = None; self.trace = None
        # evaluation pass + curriculum variables of PPO.train (ppo.py:374-386,456-470)
        self.eval_every = int(args.get("eval_every", 0)); self.eval_envs = int(args.get("eval_envs", 256))
        self.anneal_rate = float(args.get("anneal", 1.0))
        self.curr_anneal, self.curr_thresh, self.start_itr, self.ep_counter, self.do_term = 1.0, 0.0, 0, 0, False
        self._eval_env = None
        self.env_kwargs = dict(args.get("env_kwargs", {}))

    # ------------------------------------------------------------------------------------------ initialisation
    def init_networks(self, seed):
        """normc initialisation exactly as the reference constructs its nets (ppo.py:543-544, actor.py:175-178):
        built with the checkpoint classes on the host under torch.manual_seed, then uploaded."""
        from rl.policies.actor import Gaussian_FF_Actor
        from rl.policies.critic import FF_V
        torch.manual_seed(seed)
        H = self.learner.actor.H
        self.policy = Gaussian_FF_Actor(self.D, 10, layers=(H, H), fixed_std=np.exp(np.log(self.fixed_std)), env_name=self.env_name)
        self.critic = FF_V(self.D, layers=(H, H))
        self.upload()

    def upload(self):
        self.learner.actor.load_list([p.detach().numpy() for p in self.policy.parameters()])
        self.learner.critic.load_list([p.detach().numpy() for p in self.critic.parameters()])
        if torch.is_tensor(self.policy.obs_mean):
            self.learner.obs_mean.copy_(self.policy.obs_mean); self.learner.obs_std.copy_(self.policy.obs_std)

    def download(self):
        with torch.no_grad():
            for p, v in zip(self.policy.parameters(), self.learner.actor.views()):
                p.copy_(v.cpu())
            for p, v in zip(self.critic.parameters(), self.learner.critic.views()):
                p.copy_(v.cpu())
        self.policy.obs_mean = self.learner.obs_mean.cpu().clone(); self.policy.obs_std = self.learner.obs_std.cpu().clone()
        self.critic.obs_mean = self.policy.obs_mean; self.critic.obs_std = self.policy.obs_std

    def normalization_params(self, iters, noise_std=1.0):
        """get_normalization_params (rl/envs/normalize.py:11-48): policy(state) + N(0, noise_std) actions, mean and
        sqrt(var + 1e-8) of the raw observations; moments all-reduced over ranks."""
        steps = max(iters // (self.N * self.world), 50)
        obs = self.env.reset()
        D = self.D
        s = torch.zeros(D, dtype=torch.float64, device=self.device); ss = torch.zeros_like(s); n = 0
        nz = torch.zeros(self.N, 10, dtype=torch.float32, device=self.device)
        for t in range(steps):
            s += obs.double().sum(0); ss += (obs.double() ** 2).sum(0); n += obs.shape[0]
            mu = self.learner.actor.forward(obs, self.learner.obs_mean, self.learner.obs_std)
            if self.noise_fn is not None:      # parity mode: replay a captured draw stream (golden G21)
                self.noise_fn(t, nz)
            else:
                nz.normal_(generator=self.gen)
            act = mu + nz * noise_std
            obs, _, _, _ = self.env.step(act)
        mom = torch.cat([s, ss, torch.tensor([float(n)], dtype=torch.float64, device=self.device)])
        if self.group is not None:
            torch.distributed.all_reduce(mom, group=self.group)
        cnt = mom[2 * D]
        mean = mom[:D] / cnt
        var = (mom[D:2 * D] / cnt - mean * mean).clamp_min(0)
        self.learner.obs_mean.copy_(mean.float()); self.learner.obs_std.copy_(torch.sqrt(var + 1e-8).float())

    # ------------------------------------------------------------------------------------------ sampling
    @torch.no_grad()
    def _rollout_loop(self):
        L, env, T = self.learner, self.env, self.T
        self.b_obs[0].copy_(self.obs)
        if self.noise_fn is None:
            self.noise.normal_(generator=self.gen)                # the whole rollout's action noise in one launch
        if self.noise_fn is None and hasattr(env, "_h") and not getattr(env, "history", 0):          # the HIP env: the T-step loop is one C-ABI call (apx_rollout)
            from ._lib import load, check
            from .engine import _p, _stream
            check(load().apx_rollout(env._h, _p(L.actor.params), L.actor.H, _p(L.obs_mean), _p(L.obs_std), float(self.fixed_std * self.curr_anneal),
                                     _p(self.noise), T, _p(self.b_obs), _p(self.b_act), _p(self.b_mu), _p(self.b_rew), _p(self.b_done), _p(self.b_fin),
                                     _p(self.obs), _stream()))
            # The one-launch rollout (env_rollout_kernel) computes the policy means with a per-wave fp32 forward whose summation order is not the learner's MFMA forward:
            # the old-policy means the update compares against (ppo.py:287-291: the ratio is 1 at the first optimiser step) are recomputed with the learner's own forward
            # over the whole grid - one launch, 0.3 % of an iteration - so that old and new log-probabilities come from the same arithmetic
            L.actor.forward(self.b_obs.view(T * self.N, self.D), L.obs_mean, L.obs_std, out=self.b_mu.view(T * self.N, 10))
            L.critic.forward(self.b_obs.view(T * self.N, self.D), out=self.b_val.view(T * self.N, 1))
            return
        for t in range(T):      # every kernel writes straight into the rollout grids: no staging copies
            obs = self.b_obs[t]
            mu = L.actor.forward(obs, L.obs_mean, L.obs_std, out=self.b_mu[t])
            if self.noise_fn is not None:
                self.noise_fn(t, self.noise[t])
            torch.add(mu, self.noise[t], alpha=self.fixed_std * self.curr_anneal, out=self.b_act[t])
            nxt = self.b_obs[t + 1] if t + 1 < T else self.obs
            env.step(self.b_act[t], out=(nxt, self.b_rew[t], self.b_done[t], self.b_fin[t]))
        # V(s_t) is not on the stepping path: one batched critic pass over the whole grid instead of T small ones
        L.critic.forward(self.b_obs.view(T * self.N, self.D), out=self.b_val.view(T * self.N, 1))

    def sample(self):
        """T lock-step env steps for the N envs of this GPU (the batched PPO.sample, ppo.py:139-186).

        The loop enqueues work only: no host synchronisation inside the rollout (a `.any()` per step would expose every
        launch latency).  Bootstrap values V(s') of the time-limit truncations (ppo.py:184) are one batched critic pass
        over the recorded final observations, and the episode statistics are a scan over the [T, N] done/reward grids
        afterwards (same numbers as the per-step bookkeeping)."""
        L, env = self.learner, self.env
        if self.obs is None:
            self.obs = env.reset().clone()
        T, N = self.T, self.N
        if self.use_graph:
            # the T-step rollout is one captured HIP graph (12 launches per step, no host work between them on replay)
            if self._graph is None:
                self._rollout_loop()                              # one eager pass first: lazy initialisation, allocator warm-up
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                g.register_generator_state(self.gen)
                with torch.cuda.graph(g):
                    self._rollout_loop()
                self._graph = g
            self._graph.replay()
        else:
            self._rollout_loop()
        torch.ne(self.b_done, 0, out=self.b_endb)
        self.b_end.copy_(self.b_endb)
        # bootstrap value (not done) * V(s') of the time-limit truncations only (ppo.py:184): gather those rows, one critic pass
        self.b_boot.zero_()
        tr_idx = (self.b_done.view(-1) == 2).nonzero().view(-1)
        if tr_idx.numel():
            self.b_boot.view(-1)[tr_idx] = L.critic.forward(self.b_fin.view(T * N, self.D), idx=tr_idx).view(-1)
        last_val = L.critic.forward(self.obs).view(-1)
        ret = engine.returns_scan(self.b_rew, self.b_end, self.b_boot, last_val, self.gamma)
        ep_rets, ep_lens, self.ep_ret, self.ep_len = episode_stats(self.b_rew, self.b_endb, self.ep_ret, self.ep_len)
        return ret, ep_rets, ep_lens

    # ------------------------------------------------------------------------------------------ optimisation
    def epoch_kernel_in_use(self, mb):
        """whether update() runs an epoch's optimiser steps of `mb` rows as ONE launch (apx_ppo_epoch).  Never with a process group: the gradient all-reduce sits between
        the backward and the Adam step of every optimiser step, so an N > 1 run always takes the per-step launches - whatever --epoch_kernel says (bench.py prints this
        decision in config, tests/test_multirank_gloo.py holds the exclusion)."""
        return bool(self.epoch_kernel and not self.dist_on and mb <= self.epoch_kernel_max_mb and self.learner.epoch_supported(mb))

    def update(self, ret):
        L = self.learner
        B = self.T * self.N
        obs, act, mu = self.b_obs.view(B, self.D), self.b_act.view(B, 10), self.b_mu.view(B, 10)
        ret, val = ret.view(B), self.b_val.view(B)
        adv = engine.normalize_advantages(ret, val, self.eps, group=self.group)             # ppo.py:395-396
        mb = min(self.minibatch_size or B, B)
        losses, kl_last = None, 0.0
        epochs_run = 0
        for epoch in range(self.epochs):
            if self.perm_fn is None:
                perm = torch.randperm(B, device=self.device, generator=self.gen)             # SubsetRandomSampler
            else:
                perm = self.perm_fn(epoch)
            acc = torch.zeros(6, dtype=torch.float64, device=self.device)
            nb = B // mb                                                                     # drop_last=True, ppo.py:416
            if self.epoch_kernel_in_use(mb):
                scal_all = L.epoch(obs, act, ret, adv, mu, perm[:nb * mb].contiguous(), mb, mirror=self.mirror)      # [nb, 6]: every step's scalars, one launch
                acc, scal = scal_all.sum(0), scal_all[-1]
                if self.trace is not None:
                    self.trace.extend(scal_all[i].clone() for i in range(nb))
                nb_loop = 0
            else:
                nb_loop = nb
            for k in range(nb_loop):
                idx = perm[k * mb:(k + 1) * mb]
                if self.dist_on:
                    # the flat gradient travels in its two halves: the actor's (81 418 floats) is final before the critic's backward starts, so its all-reduce runs on
                    # the backend's stream NEXT TO that backward; the critic's (79 105) follows.  Same sums as one all-reduce of the whole buffer
                    scal = L.minibatch(obs, act, ret, adv, mu, idx=idx, mirror=self.mirror, grad_only=2, sync=False)
                    h = [adist.allreduce_begin(L.actor_g, group=self.group)]
                    L.minibatch(obs, act, ret, adv, mu, idx=idx, mirror=self.mirror, grad_only=3, sync=False)
                    h.append(adist.allreduce_begin(L.critic_g, group=self.group))
                    adist.allreduce_end(h, L.grad_flat, self.world)
                    L.apply_grads(scale=1.0)
                else:
                    scal = L.minibatch(obs, act, ret, adv, mu, idx=idx, mirror=self.mirror, sync=False)
                acc += scal
                if self.trace is not None:
                    self.trace.append(scal.clone())
            both = torch.cat([acc / max(nb, 1), scal.to(acc.dtype)])                         # epoch means + the last minibatch (KL test)
            if self.dist_on:
                adist.allreduce_mean_(both, group=self.group, world=self.world)              # ONE scalar all-reduce per epoch
            both = both.cpu().numpy()                                                        # the epoch's only host sync: the KL decision
            losses, kl_last = both[:6], float(both[10])
            if not np.isfinite(losses).all():
                raise FloatingPointError("non-finite PPO losses %r (a diverged env or exploding update)" % (losses,))
            epochs_run += 1
            if kl_last > 0.02:                                                               # ppo.py:449 (last minibatch's KL)
                break
        return losses, kl_last, epochs_run

    # ------------------------------------------------------------------------------------------ evaluation pass
    @torch.no_grad()
    def evaluate_pass(self):
        """The evaluation pass of PPO.train (ppo.py:464: sample_parallel(..., deterministic=True) on the TRAINING env_fn): whole
        episodes with the policy mean as action from CassieEnv.reset (dynamics randomisation as in training) on a separate small
        env batch, one episode per env.  Returns the mean episode return (rank-local; all-reduced by the caller when N > 1)."""
        if self._eval_env is None:
            kw = dict(self.env_kwargs)
            kw.update(n_envs=self.eval_envs, max_traj_len=self.max_traj_len, device=self.device.index or 0,
                      seed=int(kw.get("seed", 0)) + 7919, env_id_base=(self.rank + 1) * 1000003)
            self._eval_env = type(self.env)(**kw)
        env, L = self._eval_env, self.learner
        n = env.n_envs
        obs = env.reset()
        ret = torch.zeros(n, device=self.device); alive = torch.ones(n, dtype=torch.bool, device=self.device)
        for t in range(self.max_traj_len):
            obs, rew, done, _ = env.step(L.actor.forward(obs, L.obs_mean, L.obs_std), auto_reset=False)
            ret += torch.where(alive, rew, torch.zeros_like(rew))
            alive &= done == 0
            if t % 32 == 31 and not bool(alive.any()):
                break
        m = ret.double().mean().view(1)
        if self.group is not None:
            adist.allreduce_mean_(m, group=self.group, world=self.world)
        return float(m)

    # ------------------------------------------------------------------------------------------ training loop
    def iteration(self):
        t0 = time.time()
        ret, ep_rets, ep_lens = self.sample()
        torch.cuda.synchronize(self.device)
        t1 = time.time()
        # the next resets of every env (draws, set_const, forward pass: everything up to the settle step) are prepared on a side stream while the learner runs
        # (apx_env_prepare_resets): the in-rollout reset then only copies them and runs its one settle substep
        if self.prepare_resets and hasattr(self.env, "prepare_resets"):
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.device)
            with torch.cuda.stream(self._side):
                self.env.prepare_resets()
        losses, kl, epochs_run = self.update(ret)
        torch.cuda.synchronize(self.device)      # (all streams of the device: the prepared resets are complete before the next rollout)
        t2 = time.time()
        steps = self.T * self.N * self.world
        self.total_steps += steps
        return dict(steps=steps, sample_time=t1 - t0, optimize_time=t2 - t1, losses=losses, kl=kl, epochs=epochs_run,
                    ep_returns=ep_rets, ep_lens=ep_lens)

    def train(self, n_itr, logger=None):
        """PPO.train's outer loop (ppo.py:374-505).  The evaluation pass runs every `eval_every` iterations (the reference: every
        iteration; on the batched engine one whole-episode pass costs ~10 training iterations, so the default CLI setting is 10 and
        Test/Return holds its last value in between); eval_every = 0 disables it and Test/Return falls back to the batch return,
        labelled as such on stdout."""
        avg_eval = float("nan")
        for itr in range(n_itr):
            # curriculum variables exactly as ppo.py:378-381 / :456-460 (no-ops for Cassie-v0: anneal_rate 1.0, f_term ignored by the env)
            if self.highest_reward > (2 / 3) * self.max_traj_len and self.curr_anneal > 0.5:
                self.curr_anneal *= self.anneal_rate
            if self.do_term and self.curr_thresh < 0.35:
                self.curr_thresh = 0.1 * 1.0006 ** (itr - self.start_itr)
            out = self.iteration()
            er, el = out["ep_returns"], out["ep_lens"]
            stats = torch.stack([er.double().sum(), torch.tensor(float(er.numel()), dtype=torch.float64, device=self.device),
                                 el.double().sum()]) if er.numel() else torch.zeros(3, dtype=torch.float64, device=self.device)
            if self.group is not None:
                torch.distributed.all_reduce(stats, group=self.group)
            stats = stats.cpu().numpy()
            avg_ret = stats[0] / stats[1] if stats[1] > 0 else float("nan")
            avg_len = stats[2] / stats[1] if stats[1] > 0 else float("nan")
            if avg_len == avg_len and avg_len >= self.max_traj_len * 0.75:
                self.ep_counter += 1
            if not self.do_term and self.ep_counter > 50:
                self.do_term = True; self.start_itr = itr
            eval_time = 0.0
            if self.eval_every > 0 and itr % self.eval_every == 0:
                t0 = time.time()
                avg_eval = self.evaluate_pass()
                torch.cuda.synchronize(self.device); eval_time = time.time() - t0
            test_ret = avg_eval if self.eval_every > 0 else avg_ret
            if self.rank == 0:
                print("********** Iteration {} ************".format(itr))
                print("timesteps in batch: %i  sample %.2fs  optimize %.2fs  (%.0f env-steps/s)" % (
                    out["steps"], out["sample_time"], out["optimize_time"],
                    out["steps"] / (out["sample_time"] + out["optimize_time"])))
                print("Return (%s) %.3f  Return (batch) %.3f  Mean Eplen %.1f" % ("test" if self.eval_every > 0 else "batch, no eval pass", test_ret, avg_ret, avg_len))
                print(" ".join("%g" % x for x in out["losses"]))
                if logger is not None:
                    ml = out["losses"]
                    logger.add_scalar("Test/Return", test_ret, itr)
                    logger.add_scalar("Train/Return", avg_ret, itr)
                    logger.add_scalar("Train/Mean Eplen", avg_len, itr)
                    logger.add_scalar("Train/Mean KL Div", ml[4], itr)
                    logger.add_scalar("Train/Mean Entropy", ml[1], itr)
                    logger.add_scalar("Misc/Critic Loss", ml[2], itr)
                    logger.add_scalar("Misc/Actor Loss", ml[0], itr)
                    logger.add_scalar("Misc/Mirror Loss", ml[5], itr)
                    logger.add_scalar("Misc/Timesteps", self.total_steps, itr)
                    logger.add_scalar("Misc/Sample Times", out["sample_time"], itr)
                    logger.add_scalar("Misc/Optimize Times", out["optimize_time"], itr)
                    logger.add_scalar("Misc/Evaluation Times", eval_time, itr)
                    logger.add_scalar("Misc/Termination Threshold", self.curr_thresh, itr)
            if test_ret == test_ret and self.highest_reward < test_ret:            # ppo.py:502-504: best evaluation return so far
                self.highest_reward = test_ret
                if self.rank == 0:
                    self.save()

    def save(self):
        """PPO.save (ppo.py:129-137): whole-module pickles actor.pt / critic.pt."""
        os.makedirs(self.save_path, exist_ok=True)
        self.download()
        torch.save(self.policy, os.path.join(self.save_path, "actor.pt"))
        torch.save(self.critic, os.path.join(self.save_path, "critic.pt"))


def run_experiment(args):
    """rl/algos/ppo.py:507-584 for the batched engine.  One process per GPU (torchrun sets RANK/LOCAL_RANK/WORLD_SIZE)."""
    from .log import create_logger
    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    group = None
    if world > 1 or os.environ.get("APX_FORCE_DIST") == "1":      # (APX_FORCE_DIST: the RCCL path at world_size 1, launched through torch.distributed.run)
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
        group = torch.distributed.group.WORLD
    torch.manual_seed(args.seed); np.random.seed(args.seed)
    n_envs = getattr(args, "n_envs", 4096)
    env_kwargs = dict(simrate=args.simrate, dynamics_randomization=args.dyn_random, reward=args.reward, seed=args.seed,
                      command_profile=args.command_profile, input_profile=args.input_profile, history=args.history,
                      learn_gains=args.learn_gains, env_name=args.env_name, traj=args.traj, no_delta=args.no_delta,
                      ik_baseline=args.ik_baseline, est_lifetime=int(getattr(args, "est_lifetime", None) if getattr(args, "est_lifetime", None) is not None else 169))
    env = CassieVecEnv(n_envs=n_envs, max_traj_len=args.max_traj_len, device=local, env_id_base=adist.shard_env_base(rank, n_envs), **env_kwargs)
    logger = create_logger(args) if rank == 0 else None
    a = dict(vars(args)); a["mirror"] = args.mirror; a["env_kwargs"] = env_kwargs
    if getattr(args, "recurrent", False):
        from .ppo_recurrent import RecurrentPPO
        a.setdefault("std_dev", -2.0)
        a["std_dev"] = -2.0                                   # ppo.py:537: recurrent policies are built with fixed_std = exp(-2)
        algo = RecurrentPPO(a, logger.dir if logger else "/tmp/apx_unused", env, rank=rank, world_size=world, group=group)
    else:
        algo = PPO(a, logger.dir if logger else "/tmp/apx_unused", env, rank=rank, world_size=world, group=group)
    if args.previous is not None:
        algo.policy = torch.load(os.path.join(args.previous, "actor.pt"), weights_only=False)
        algo.critic = torch.load(os.path.join(args.previous, "critic.pt"), weights_only=False)
        algo.upload()
    else:
        algo.init_networks(args.seed)
        algo.normalization_params(args.input_norm_steps)
    algo.train(args.n_itr, logger=logger)
    return algo
