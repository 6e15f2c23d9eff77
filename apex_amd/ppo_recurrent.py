"""Recurrent PPO for the batched engine: the reference's `--recurrent` path (rl/algos/ppo.py:139-186 rollout
with per-episode hidden state, :411-430 minibatches of whole trajectories padded to [T_max, B, .] with a mask, :276-345 update) with
Gaussian_LSTM_Actor / LSTM_V in HIP (apex_amd.engine.Lstm, RecurrentPPOLearner).

Rollout: N envs in lock step for T steps; the hidden state of an env is zeroed when its episode ends (ppo.py:164-168).  Every iteration
starts from a reset of all envs, so that every trajectory in the batch starts at zero hidden state exactly like the padded training pass
assumes; the last (unfinished) trajectory of a column is cut at the grid end and bootstrapped with V(s_T) like a time-limit truncation.
A minibatch is `minibatch_size` TRAJECTORIES (ppo.py:412-413), not steps."""
import os
import time

import numpy as np
import torch

from . import dist as adist
from . import engine
from .vecenv import MIRRORED_ACTS, MIRRORED_OBS, CLOCK_INDS


class RecurrentPPO:
    def __init__(self, args, save_path, env, rank=0, world_size=1, group=None, hidden=128, layers=2):
        self.gamma = args["gamma"]; self.lr = args["lr"]; self.eps = args["eps"]; self.clip = args["clip"]
        self.minibatch_size = args["minibatch_size"]; self.epochs = args["epochs"]
        self.num_steps = args["num_steps"]; self.max_traj_len = args["max_traj_len"]
        self.grad_clip = args["max_grad_norm"]; self.mirror = args.get("mirror", True)
        self.fixed_std = float(np.exp(args.get("std_dev", -2.0)))                  # ppo.py:537: recurrent policies use exp(-2)
        self.env_name = args.get("env_name", "Cassie-v0")
        self.save_path, self.env = save_path, env
        self.rank, self.world, self.group = rank, world_size, group
        self.dist_on = group is not None      # the collective path runs whenever a process group is given, world_size 1 included (APX_FORCE_DIST: the RCCL path on one GPU)
        self.device, self.N = env.device, env.n_envs
        if getattr(env, "obs_dim", 50) != 50:
            raise NotImplementedError("recurrent PPO is built for the 50-entry observation (command_profile=clock, history 0); this env produces %d entries" % getattr(env, "obs_dim", 50))
        # the bootstrap of time-limit truncations is only looked for from step max_traj_len - 1 on (sample()): that threshold must be the one the ENV uses for done == 2
        if getattr(env, "max_traj_len", self.max_traj_len) != self.max_traj_len:
            raise ValueError("env.max_traj_len %d != args max_traj_len %d: time-limit truncations would be bootstrapped with 0" % (env.max_traj_len, self.max_traj_len))
        self.T = adist.rollout_len(self.num_steps, self.N, self.world)
        # every iteration restarts all envs (each trajectory must begin at an episode start with zero hidden state, like the padded
        # training pass assumes), so a grid shorter than max_traj_len would never show the policy the later part of an episode
        if self.T < self.max_traj_len and not args.get("allow_short_rollout", False):
            raise ValueError("recurrent PPO: %d steps per env and iteration < max_traj_len %d: the policy would only ever train on the first %d steps "
                             "of an episode; raise num_steps to n_envs x max_traj_len (or pass allow_short_rollout for a smoke run)" % (self.T, self.max_traj_len, self.T))
        self.H, self.L = hidden, layers
        self.learner = engine.RecurrentPPOLearner(50, 10, hidden, layers, self.device, self.fixed_std, lr=self.lr, eps=self.eps, clip=self.clip,
                                                  grad_clip=self.grad_clip, mirrored_obs=MIRRORED_OBS if self.mirror else None,
                                                  mirrored_acts=MIRRORED_ACTS if self.mirror else None, clock_inds=list(getattr(env, "clock_inds", CLOCK_INDS)))      # env.clock_inds like rl/algos/ppo.py:307-310 (21, 22 with input_profile=min)
        self.gen = torch.Generator(device=self.device); self.gen.manual_seed(int(args.get("seed", 0)) * 1000003 + rank)
        T, N = self.T, self.N
        f32 = dict(dtype=torch.float32, device=self.device)
        self.b_obs = torch.zeros(T, N, 50, **f32); self.b_act = torch.zeros(T, N, 10, **f32)
        self.b_rew = torch.zeros(T, N, **f32); self.b_val = torch.zeros(T, N, **f32); self.b_boot = torch.zeros(T, N, **f32)
        self.b_done = torch.zeros(T, N, dtype=torch.uint8, device=self.device); self.b_fin = torch.zeros(T, N, 50, **f32)
        self.b_end = torch.zeros(T, N, dtype=torch.uint8, device=self.device)
        self.total_steps = 0; self.highest_reward = -1
        self.noise_fn = None; self.perm_fn = None; self.trace = None              # parity hooks, as in apex_amd.ppo.PPO

    # ------------------------------------------------------------------------------------------ networks / checkpoints
    def init_networks(self, seed):
        from rl.policies.actor import Gaussian_LSTM_Actor
        from rl.policies.critic import LSTM_V
        torch.manual_seed(seed)
        self.policy = Gaussian_LSTM_Actor(50, 10, layers=(self.H,) * self.L, fixed_std=self.fixed_std, env_name=self.env_name)
        self.critic = LSTM_V(50, layers=(self.H,) * self.L)
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

    def save(self):
        os.makedirs(self.save_path, exist_ok=True)
        self.download()
        torch.save(self.policy, os.path.join(self.save_path, "actor.pt")); torch.save(self.critic, os.path.join(self.save_path, "critic.pt"))

    def normalization_params(self, iters, noise_std=1.0):
        """get_normalization_params (rl/envs/normalize.py:11-48) with the recurrent policy stepping its hidden state."""
        L = self.learner
        steps = max(iters // (self.N * self.world), 50)
        obs = self.env.reset()
        hc = torch.zeros(self.L, 2, self.N, self.H, device=self.device)
        s = torch.zeros(50, dtype=torch.float64, device=self.device); ss = torch.zeros_like(s); n = 0
        for _ in range(steps):
            s += obs.double().sum(0); ss += (obs.double() ** 2).sum(0); n += obs.shape[0]
            mu = L.actor.forward(((obs - L.obs_mean) / L.obs_std).contiguous(), hc=hc)
            obs, _, done, _ = self.env.step(mu + torch.randn(mu.shape, device=self.device, generator=self.gen) * noise_std)
            hc[:, :, done != 0] = 0
        mom = torch.cat([s, ss, torch.tensor([float(n)], dtype=torch.float64, device=self.device)])
        if self.group is not None:
            torch.distributed.all_reduce(mom, group=self.group)
        mean = mom[:50] / mom[100]
        var = (mom[50:100] / mom[100] - mean * mean).clamp_min(0)
        L.obs_mean.copy_(mean.float()); L.obs_std.copy_(torch.sqrt(var + 1e-8).float())

    # ------------------------------------------------------------------------------------------ sampling
    @torch.no_grad()
    def sample(self):
        L, env, T, N = self.learner, self.env, self.T, self.N
        obs = env.reset()
        hc_a = torch.zeros(self.L, 2, N, self.H, device=self.device); hc_c = torch.zeros_like(hc_a)
        noise = torch.zeros(N, 10, device=self.device)
        norm = lambda o: ((o - L.obs_mean) / L.obs_std).contiguous()
        self.b_boot.zero_()
        # No host round trip inside the loop (round 3): the Python loop runs ahead of the GPU, so launch latency hides behind the 3 ms env kernel.
        # The critic's one-step chain does not feed the env step: it runs on a side stream, next to the env kernel (2048 envs fill half the SIMDs).
        main = torch.cuda.current_stream(self.device)
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device)
        side = self._side
        side.wait_stream(main)
        # the policy step as ONE launch per network (apx_lstm_step: normalisation, init_hidden_state of the restarted rows, both cells, head, action noise) where the
        # shape allows it (2 x LSTMCell(128)); otherwise the per-launch chain
        fused = os.environ.get("APX_LSTM_STEP", "1") != "0" and L.actor.step_supported() and L.critic.step_supported()
        if fused:
            L.actor.pack_step(); L.critic.pack_step()
            side.wait_stream(main)
        # the env writes the next observation straight into the grid row of the next step (no per-step copy); the last one goes to a scratch row
        self.b_obs[0].copy_(obs)
        nxt_last = torch.empty_like(obs)
        noise_all = None
        if self.noise_fn is None:      # every step's action noise in one launch instead of one small launch per step in front of the policy step
            noise_all = torch.empty(T, N, 10, device=self.device).normal_(generator=self.gen)
            self._noise_all = noise_all      # (kept for the tests: act - fixed_std * noise = the rollout's means)
        # One launch for the whole rollout where the shape allows it (apx_rollout_lstm: 2 x LSTMCell(128) evaluated per wave inside env_rollout_kernel): no per-step launch, and a
        # forward pass on the complete-row path stalls its own wave instead of the launch of all envs.  The critic's values follow over the stored grid, step by step with its
        # carried state and the same bootstrap rule as below (it never fed the env step).
        one_launch = (fused and noise_all is not None and hasattr(env, "_h") and not getattr(env, "history", 0) and self.H == 128 and self.L == 2
                      and os.environ.get("APX_ROLLOUT_STEPWISE", "0") == "0")
        if one_launch:
            from ._lib import load, check
            from .engine import _p, _stream
            if getattr(self, "_b_mu", None) is None:
                self._b_mu = torch.empty(T, N, 10, device=self.device)
            check(load().apx_rollout_lstm(env._h, _p(L.actor.params), self.H, self.L, _p(L.obs_mean), _p(L.obs_std), float(self.fixed_std), _p(noise_all), T,
                                          _p(self.b_obs), _p(self.b_act), _p(self._b_mu), _p(self.b_rew), _p(self.b_done), _p(self.b_fin), _p(nxt_last), _stream()))
            for t in range(T):
                L.critic.step(self.b_obs[t], hc_c, reset=self.b_done[t - 1] if t > 0 else None, y_out=self.b_val[t].view(N, 1))
                last = t == T - 1
                if last or t + 1 >= self.max_traj_len:
                    nxt = self.b_obs[t + 1] if t + 1 < T else nxt_last
                    tr = self.b_done[t] == 2
                    rows = tr if not last else (tr | (self.b_done[t] == 0))
                    src = torch.where((self.b_done[t] != 0).view(N, 1), self.b_fin[t], nxt)
                    v_next = L.critic.forward(src.contiguous(), hc=hc_c.clone()).view(-1)
                    self.b_boot[t] = torch.where(rows, v_next, torch.zeros_like(v_next))
        for t in range(0 if not one_launch else T, T):
            nxt = self.b_obs[t + 1] if t + 1 < T else nxt_last
            ev_obs = main.record_event()
            prev_done = self.b_done[t - 1] if (fused and t > 0) else None      # uint8 0 / 1 / 2: non-zero rows start from a zero state
            with torch.cuda.stream(side):
                side.wait_event(ev_obs)
                if fused:
                    L.critic.step(self.b_obs[t], hc_c, reset=prev_done, y_out=self.b_val[t].view(N, 1))
                else:
                    self.b_val[t].copy_(L.critic.forward(self.b_obs[t], hc=hc_c).view(-1))
            if noise_all is not None:
                noise = noise_all[t]
            else:
                self.noise_fn(t, noise)
            if fused:
                L.actor.step(self.b_obs[t], hc_a, L.obs_mean, L.obs_std, reset=prev_done, noise=noise, sigma=self.fixed_std, act_out=self.b_act[t])
            else:
                mu = L.actor.forward(norm(self.b_obs[t]), hc=hc_a)
                torch.add(mu, noise, alpha=self.fixed_std, out=self.b_act[t])
            env.step(self.b_act[t], out=(nxt, self.b_rew[t], self.b_done[t], self.b_fin[t]))      # writes the next observation into the next grid row
            last = t == T - 1
            if fused and not (last or t + 1 >= self.max_traj_len):
                continue      # nothing for the side stream behind this step: the critic's next step is ordered by the next ev_obs (an event record is a barrier packet on the main stream)
            ev_step = main.record_event()
            with torch.cuda.stream(side):
                side.wait_event(ev_step)
                # a time-limit truncation needs max_traj_len steps since the env's last reset, and every env was reset at t = 0: before step
                # max_traj_len - 1 there is nothing to bootstrap (and no reason to ask the device whether there is)
                if last or t + 1 >= self.max_traj_len:      # V(s') with the critic's carried state (ppo.py:183-184), without advancing it
                    tr = self.b_done[t] == 2
                    rows = tr if not last else (tr | (self.b_done[t] == 0))
                    src = torch.where((self.b_done[t] != 0).view(N, 1), self.b_fin[t], nxt)      # the episode's own next observation
                    v_next = L.critic.forward(src.contiguous(), hc=hc_c.clone()).view(-1)
                    self.b_boot[t] = torch.where(rows, v_next, torch.zeros_like(v_next))
                if not fused:
                    hc_c.masked_fill_((self.b_done[t] != 0).view(1, 1, N, 1), 0.0)      # init_hidden_state at every episode start (ppo.py:164-168); an assignment, not a product: 0 * NaN stays NaN
            if not fused:
                hc_a.masked_fill_((self.b_done[t] != 0).view(1, 1, N, 1), 0.0)
        main.wait_stream(side)
        self.b_end.copy_((self.b_done != 0).to(torch.uint8)); self.b_end[T - 1] = 1      # the grid end cuts the last trajectory of every column
        ret = engine.returns_scan(self.b_rew, self.b_end, self.b_boot, torch.zeros(N, device=self.device), self.gamma)
        return ret

    def trajectories(self):
        """(column, t0, t1) of every trajectory in the grid, in (column, time) order -> int64 [n_traj, 3] on the host."""
        end = self.b_end.t().contiguous().cpu().numpy()                              # [N, T]
        col, tend = np.nonzero(end)                                                  # row-major: sorted by column, then time
        t1 = tend.astype(np.int64) + 1
        t0 = np.zeros_like(t1)
        if len(t1) > 1:
            t0[1:] = np.where(col[1:] == col[:-1], t1[:-1], 0)                       # a trajectory starts where the previous one of its column ended
        return np.stack([col.astype(np.int64), t0, t1], axis=1)

    def padded_index(self, trajs):
        """Flat grid indices t * N + n of a set of trajectories as a [T_max, B] tensor, -1 where padded (torch's pad_sequence layout).
        (Vectorised: as a Python loop over the 1024 trajectories of a minibatch it was 1.7 ms of host time in front of every 4 ms of learner kernels.)"""
        lens = trajs[:, 2] - trajs[:, 1]
        Tm = int(lens.max())
        tr = torch.as_tensor(np.ascontiguousarray(trajs), device=self.device)       # [B, 3]: 24 KB
        tt = torch.arange(Tm, device=self.device, dtype=torch.int64).view(Tm, 1)
        return torch.where(tt < (tr[:, 2] - tr[:, 1]).view(1, -1), (tr[:, 1].view(1, -1) + tt) * self.N + tr[:, 0].view(1, -1), -1)

    # ------------------------------------------------------------------------------------------ optimisation
    def update(self, ret):
        L, T, N = self.learner, self.T, self.N
        flat = lambda x, d: x.view(T * N, d)
        val = self.b_val.view(-1); retf = ret.view(-1)
        adv = engine.normalize_advantages(retf, val, self.eps, group=self.group)
        trajs = self.trajectories()
        mb = self.minibatch_size or len(trajs)
        # N > 1: the ranks hold different numbers of trajectories (their env shards end episodes at different times), so the number of
        # optimiser steps per epoch is agreed first (max over ranks); a rank that has run out of trajectories contributes a zero gradient to
        # the remaining all-reduces.  Every rank therefore issues the same sequence of collectives and ends with the same parameters.
        n_mb = -(-len(trajs) // mb)
        trajs_dev = torch.as_tensor(np.ascontiguousarray(trajs), device=self.device)      # [n_traj, 3] (column, t0, t1)
        lens = trajs[:, 2] - trajs[:, 1]
        if self.dist_on:
            cnt = torch.tensor([n_mb], dtype=torch.int64, device=self.device)
            torch.distributed.all_reduce(cnt, op=torch.distributed.ReduceOp.MAX, group=self.group)
            n_mb = int(cnt)
        L.sync_old()
        losses, kl_last, epochs_run = None, 0.0, 0
        self.optimiser_steps = 0      # of this update (bench.py: update time per optimiser step)
        for epoch in range(self.epochs):
            if self.perm_fn is not None:
                order = np.asarray(self.perm_fn(epoch))
                order_dev = torch.as_tensor(order.astype(np.int64), device=self.device)
            else:
                order_dev = torch.randperm(len(trajs), device=self.device, generator=self.gen)      # SubsetRandomSampler over trajectories
                order = order_dev.cpu().numpy()
            acc = torch.zeros(6, dtype=torch.float64, device=self.device); nb = 0
            for kk in range(n_mb):                                               # BatchSampler(..., drop_last=False), ppo.py:413
                k = kk * mb
                if k >= len(order):                                              # (N > 1 only) this rank has no trajectories left in this epoch
                    L.grad_flat.zero_()
                    adist.allreduce_mean_(L.grad_flat, group=self.group, world=self.world)
                    L.apply_grads()
                    continue
                # one launch (apx_rec_gather); the [T_max, B] index of padded_index() is formed inside it from the trajectory list
                sel = order_dev[k:k + mb]
                o_p, a_p, r_p, d_p, m_p, prep = L.gather(None, flat(self.b_obs, 50), flat(self.b_act, 10), retf, adv, mirror=self.mirror, traj=trajs_dev, sel=sel,
                                                         grid_cols=self.N, t_max=int(lens[order[k:k + mb]].max()))
                scal = L.minibatch(o_p, a_p, r_p, d_p, m_p, mirror=self.mirror, grad_only=self.dist_on, prepared=prep)
                if self.dist_on:
                    adist.allreduce_mean_(L.grad_flat, group=self.group, world=self.world)
                    L.apply_grads()
                acc += scal; nb += 1
                if self.trace is not None:
                    self.trace.append(scal.clone())
            both = torch.cat([acc / max(nb, 1), scal.to(acc.dtype)])
            if self.dist_on:
                adist.allreduce_mean_(both, group=self.group, world=self.world)          # the KL decision must be the same on every rank
            both = both.cpu().numpy()
            losses, kl_last = both[:6], float(both[10]); epochs_run += 1; self.optimiser_steps += n_mb
            if kl_last > 0.02:
                break
        return losses, kl_last, epochs_run

    def iteration(self):
        t0 = time.time()
        ret = self.sample()
        torch.cuda.synchronize(self.device); t1 = time.time()
        if hasattr(self.env, "prepare_resets"):      # next resets prepared on a side stream while the update runs (apx_env_prepare_resets)
            if getattr(self, "_prep_side", None) is None:
                self._prep_side = torch.cuda.Stream(device=self.device)
            with torch.cuda.stream(self._prep_side):
                self.env.prepare_resets()
        losses, kl, epochs_run = self.update(ret)
        torch.cuda.synchronize(self.device); t2 = time.time()
        ended = self.b_done != 0
        from .ppo import episode_stats
        z = torch.zeros(self.N, device=self.device)
        ep_rets, ep_lens, _, _ = episode_stats(self.b_rew, ended, z, z)
        steps = self.T * self.N * self.world
        self.total_steps += steps
        return dict(steps=steps, sample_time=t1 - t0, optimize_time=t2 - t1, losses=losses, kl=kl, epochs=epochs_run, optimiser_steps=self.optimiser_steps, ep_returns=ep_rets, ep_lens=ep_lens)

    def train(self, n_itr, logger=None):
        for itr in range(n_itr):
            out = self.iteration()
            er, el = out["ep_returns"], out["ep_lens"]
            avg_ret = float(er.mean()) if er.numel() else float("nan"); avg_len = float(el.mean()) if el.numel() else float("nan")
            if self.rank == 0:
                print("********** Iteration {} ************".format(itr))
                print("timesteps in batch: %i  sample %.2fs  optimize %.2fs  (%.0f env-steps/s)  mean eplen %.1f" % (
                    out["steps"], out["sample_time"], out["optimize_time"], out["steps"] / (out["sample_time"] + out["optimize_time"]), avg_len))
                print(" ".join("%g" % x for x in out["losses"]))
                if logger is not None:
                    ml = out["losses"]
                    for tag, v in (("Test/Return", avg_ret), ("Train/Return", avg_ret), ("Train/Mean Eplen", avg_len), ("Train/Mean KL Div", ml[4]),
                                   ("Train/Mean Entropy", ml[1]), ("Misc/Critic Loss", ml[2]), ("Misc/Actor Loss", ml[0]), ("Misc/Mirror Loss", ml[5]),
                                   ("Misc/Timesteps", self.total_steps), ("Misc/Sample Times", out["sample_time"]),
                                   ("Misc/Optimize Times", out["optimize_time"]), ("Misc/Evaluation Times", 0.0), ("Misc/Termination Threshold", 0.0)):
                        logger.add_scalar(tag, v, itr)
                if avg_ret == avg_ret and self.highest_reward < avg_ret:
                    self.highest_reward = avg_ret
                    self.save()
