"""TD3 for the batched engine (BASELINE configs[4]): the reference's synchronous TD3 (rl/algos/sync_td3.py)
with the replay buffer resident in HBM and the twin-critic update in HIP (apex_amd.engine.TD3Learner).

Reference loop (sync_td3.py:286-330): workers collect whole episodes with action = clip(actor(s) + N(0, act_noise), -1, 1) - ONE scalar
draw added to all action dimensions (np.random.normal(..., size=1), :77) -, transitions (s, s', a, r, done_bool) go to a ring replay
(done_bool is also 1 at the time limit, :82), then one TD3.train iteration per collected step on a uniform-with-replacement batch
(rl/utils/remote_replay.py:78-90).  Here N envs collect in lock step and every lock-step env step is followed by `updates_per_step`
updates; the replay is five device tensors written with a ring index."""
import os
import time

import numpy as np
import torch

from . import engine


class HbmReplay:
    """Ring replay of (s, s', a, r, notdone) in device memory; 10^6 x (50 + 50 + 10 + 1 + 1) fp32 = 448 MB."""

    def __init__(self, capacity, obs_dim, act_dim, device):
        f = dict(dtype=torch.float32, device=device)
        self.cap, self.device = int(capacity), device
        self.s = torch.zeros(self.cap, obs_dim, **f); self.s2 = torch.zeros(self.cap, obs_dim, **f); self.a = torch.zeros(self.cap, act_dim, **f)
        self.r = torch.zeros(self.cap, **f); self.nd = torch.zeros(self.cap, **f)
        self.ptr, self.size = 0, 0

    def add(self, s, s2, a, r, notdone):
        n = s.shape[0]
        idx = (torch.arange(n, device=self.device) + self.ptr) % self.cap          # remote_replay.py:70-74: overwrite the oldest
        self.s.index_copy_(0, idx, s); self.s2.index_copy_(0, idx, s2); self.a.index_copy_(0, idx, a)
        self.r.index_copy_(0, idx, r); self.nd.index_copy_(0, idx, notdone)
        self.ptr = (self.ptr + n) % self.cap; self.size = min(self.size + n, self.cap)

    def sample(self, batch_size, gen, ind=None):
        if ind is None:
            ind = torch.randint(0, self.size, (batch_size,), device=self.device, generator=gen)      # np.random.randint(0, len, size): with replacement
        g = lambda x: x.index_select(0, ind)
        return g(self.s), g(self.s2), g(self.a), g(self.r), g(self.nd)


class TD3:
    def __init__(self, env, save_path, hidden=256, a_lr=1e-3, c_lr=1e-3, discount=0.99, tau=0.005, policy_noise=0.2, noise_clip=0.5, policy_freq=2,
                 act_noise=0.3, batch_size=256, updates_per_step=1, replay_size=1_000_000, seed=0, param_noise=False, noise_scale=0.3, one_launch_updates=None):
        self.env, self.save_path, self.device, self.N = env, save_path, env.device, env.n_envs
        if getattr(env, "obs_dim", 50) != 50:
            raise NotImplementedError("TD3 is built for the 50-entry observation (command_profile=clock, history 0); this env produces %d entries" % getattr(env, "obs_dim", 50))
        self.learner = engine.TD3Learner(50, 10, hidden, self.device, 1.0, a_lr, c_lr)
        self.replay = HbmReplay(replay_size, 50, 10, self.device)
        self.discount, self.tau, self.policy_noise, self.noise_clip, self.policy_freq = discount, tau, policy_noise, noise_clip, policy_freq
        self.act_noise, self.batch_size, self.updates_per_step, self.hidden = act_noise, batch_size, updates_per_step, hidden
        self.gen = torch.Generator(device=self.device); self.gen.manual_seed(int(seed) * 1000003 + 17)
        self.it = 0; self.total_steps = 0; self.obs = None
        # the update block behind a one-launch collection as ONE launch as well (apx_td3_updates, td3_small.hip); on request until it has run on hardware
        self.one_launch_updates = bool(one_launch_updates) if one_launch_updates is not None else os.environ.get("APX_TD3_UPDATES", "0") == "1"
        # parameter-space exploration noise (rl/utils/param_noise.py AdaptiveParamNoiseSpec; TD3.perturb_actor_parameters, sync_td3.py:113-121).
        # The reference's synchronous loop builds the spec (initial 0.05, desired action stddev = --noise_scale, coefficient 1.05,
        # sync_td3.py:286) but never passes it to the collectors; here --param_noise wires it the way the asynchronous variant does: collect
        # with a perturbed copy of the actor, re-perturb and adapt the stddev after every collection round.
        self.param_noise = bool(param_noise)
        self.pn_std, self.pn_desired, self.pn_coef = 0.05, float(noise_scale), 1.05
        self.actor_perturbed = engine.Mlp(50, hidden, 10, self.device) if self.param_noise else None

    def perturb_actor_parameters(self):
        """actor_perturbed <- actor + N(0, current_stddev) on every parameter (sync_td3.py:113-121)"""
        self.actor_perturbed.params.copy_(self.learner.actor.params + torch.randn(self.learner.actor.n, device=self.device, generator=self.gen) * self.pn_std)

    def adapt_param_noise(self, states):
        """AdaptiveParamNoiseSpec.adapt(distance_metric(perturbed actions, actions)) on a batch of visited states (rl/utils/param_noise.py:19-50)"""
        a = self.learner.act(states); ap = torch.tanh(self.actor_perturbed.forward(states))
        dist = float(torch.sqrt(((a - ap) ** 2).mean(0).mean()))
        self.pn_std = self.pn_std / self.pn_coef if dist > self.pn_desired else self.pn_std * self.pn_coef
        return dist

    def init_networks(self, seed):
        from rl.policies.actor import FF_Actor
        from rl.policies.critic import Dual_Q_Critic
        torch.manual_seed(seed)
        self.policy = FF_Actor(50, 10, layers=(self.hidden, self.hidden), max_action=1.0)
        self.critic = Dual_Q_Critic(50, 10, hidden_size=self.hidden)
        L = self.learner
        L.actor.load_list([p.detach().numpy() for p in self.policy.parameters()]); L.actor_t.params.copy_(L.actor.params)
        ps = [p.detach().numpy() for p in self.critic.parameters()]
        L.q[0].load_list(ps[:6]); L.q[1].load_list(ps[6:]); L.critic_t_flat.copy_(L.critic_flat)

    def save(self):
        os.makedirs(self.save_path, exist_ok=True)
        with torch.no_grad():
            for p, v in zip(self.policy.parameters(), self.learner.actor.views()):
                p.copy_(v.cpu())
            for p, v in zip(self.critic.parameters(), self.learner.q[0].views() + self.learner.q[1].views()):
                p.copy_(v.cpu())
        torch.save(self.policy, os.path.join(self.save_path, "actor.pt")); torch.save(self.critic, os.path.join(self.save_path, "critic.pt"))

    @torch.no_grad()
    def collect_and_train(self, steps):
        """`steps` lock-step env steps and `steps x updates_per_step` updates.  Default (HIP env, hidden 256, no parameter noise): the schedule of sync_td3.py:300-313 -
        the whole collection runs first with the policy FIXED (one launch, apx_rollout_td3), its transitions enter the replay, then all updates run.  The fallback loop
        (parameter noise, other widths, envs without the C handle) interleaves `updates_per_step` updates behind every env step instead: same number of steps and
        updates, a different schedule - results of the two paths are not interchangeable."""
        L, env = self.learner, self.env
        if self.obs is None:
            self.obs = env.reset().clone()
        stats = torch.zeros(3, dtype=torch.float64, device=self.device); n_upd = 0
        if self.param_noise:
            self.perturb_actor_parameters()
        # The HIP env: the `steps` collection steps as ONE launch with the policy fixed (apx_rollout_td3 - what sync_td3.py does: collect_experience runs the current
        # policy to the step budget, :300-313, THEN policy.train runs one iteration per collected timestep), all transitions into the replay, then steps x updates_per_step
        # updates.  (The per-step loop below interleaves an update block behind every env step; it stays for envs without the C handle, parameter noise and non-256 widths.)
        if (hasattr(env, "_h") and not self.param_noise and self.hidden == 256 and not getattr(env, "history", 0) and os.environ.get("APX_ROLLOUT_STEPWISE", "0") == "0"
                and os.environ.get("APX_TD3_ROLLOUT", "1") != "0"):
            from ._lib import load, check
            from .engine import _p, _stream
            T, N = steps, self.N
            if getattr(self, "_g", None) is None or self._g["obs"].shape[0] != T:
                f32 = dict(dtype=torch.float32, device=self.device)
                self._g = dict(obs=torch.empty(T, N, 50, **f32), act=torch.empty(T, N, 10, **f32), mu=torch.empty(T, N, 10, **f32), rew=torch.empty(T, N, **f32),
                               done=torch.empty(T, N, dtype=torch.uint8, device=self.device), fin=torch.zeros(T, N, 50, **f32), nxt=torch.empty(N, 50, **f32))
            g = self._g
            g["obs"][0].copy_(self.obs)
            noise = torch.randn(T, N, device=self.device, generator=self.gen) if self.act_noise != 0 else None      # one scalar per env step (:77)
            self._noise_last = noise      # (kept for the tests)
            check(load().apx_rollout_td3(env._h, _p(L.actor.params), self.hidden, float(L.max_action), float(self.act_noise), _p(noise), 0, T, _p(g["obs"]), _p(g["act"]),
                                         _p(g["mu"]), _p(g["rew"]), _p(g["done"]), _p(g["fin"]), _p(g["nxt"]), _stream()))
            for t in range(T):
                nxt = g["obs"][t + 1] if t + 1 < T else g["nxt"]
                ended = g["done"][t] != 0
                self.replay.add(g["obs"][t], torch.where(ended.view(-1, 1), g["fin"][t], nxt), g["act"][t], g["rew"][t], (~ended).float())
            self.obs = g["nxt"].clone()
            n_todo = T * self.updates_per_step if self.replay.size >= self.batch_size else 0
            if n_todo and self.one_launch_updates and L.updates_supported(self.batch_size):
                # the whole update block as ONE launch (apx_td3_updates): the replay rows and the smoothing noise of every update are drawn up front (same distributions,
                # one draw each for the block instead of one per update), the kernel gathers the rows out of the replay tensors itself
                U, Bz = n_todo, self.batch_size
                ind = torch.randint(0, self.replay.size, (U, Bz), device=self.device, generator=self.gen)
                pn = torch.randn(U, Bz, 10, device=self.device, generator=self.gen) * self.policy_noise
                R = self.replay
                st = L.updates(R.s, R.s2, R.a, R.r, R.nd, ind, pn, self.it, self.discount, self.tau, self.noise_clip, self.policy_freq)
                if not bool(torch.isfinite(st).all()):      # the grid barrier's watchdog ends a launch that could not finish with NaN statistics AFTER parameters were partly updated
                    raise FloatingPointError("apx_td3_updates returned non-finite statistics (its workgroups were not resident together, or the update diverged): "
                                             "the networks are not valid any more; re-run from a checkpoint without --td3_one_launch")
                stats += st[:, :3].sum(0); n_upd += U; self.it += U
                n_todo = 0
            for _ in range(n_todo):
                s_, sn, ac, r, nd = self.replay.sample(self.batch_size, self.gen)
                pn = torch.randn(self.batch_size, 10, device=self.device, generator=self.gen) * self.policy_noise
                st, _ = L.train_step(s_, ac, sn, r, nd, pn, self.it, self.discount, self.tau, self.noise_clip, self.policy_freq)
                stats += st; n_upd += 1; self.it += 1
            steps = 0
            self.total_steps += T * N
        for _ in range(steps):
            a = torch.tanh(self.actor_perturbed.forward(self.obs)) if self.param_noise else L.act(self.obs)
            if self.act_noise != 0:
                a = (a + torch.randn(self.N, 1, device=self.device, generator=self.gen) * self.act_noise).clamp(-1, 1)      # one scalar per env step (:77)
            nxt, rew, done, fin = env.step(a)
            ended = done != 0
            s2 = torch.where(ended.view(-1, 1), fin, nxt)
            self.replay.add(self.obs, s2, a, rew, (~ended).float())
            self.obs = nxt.clone()
            for _ in range(self.updates_per_step):
                if self.replay.size < self.batch_size:
                    break
                s, sn, ac, r, nd = self.replay.sample(self.batch_size, self.gen)
                noise = torch.randn(self.batch_size, 10, device=self.device, generator=self.gen) * self.policy_noise
                st, _ = L.train_step(s, ac, sn, r, nd, noise, self.it, self.discount, self.tau, self.noise_clip, self.policy_freq)
                stats += st; n_upd += 1; self.it += 1
        self.total_steps += steps * self.N
        if self.param_noise:
            self.adapt_param_noise(self.obs)
        s = (stats / max(n_upd, 1)).cpu().numpy()
        return dict(q_loss=float(s[0]), avg_q1=float(s[1] / self.batch_size), avg_q2=float(s[2] / self.batch_size), updates=n_upd)

    # ------------------------------------------------------------------------------------------ asynchronous variant
    @torch.no_grad()
    def collect_and_train_async(self, steps, load_freq=10):
        """The reference's ASYNCHRONOUS TD3 (rl/algos/async_td3.py) on the batched env: collection and learning are decoupled.
        Reference: every Actor steps its env with a COPY of the global policy that it re-loads every `load_freq` of its own timesteps (:205-213), adds N(0, act_noise) PER
        ACTION DIMENSION (:253-256; the synchronous loop draws one scalar), asks the Learner for one update per collected step without waiting for it (:285) and ships
        its transitions to the replay; the Learner samples whatever the replay holds (:404-415).
        Here: the collection (behaviour-policy forward, env step, replay write) runs on the caller's stream, the `updates_per_step` updates per lock-step env step run on
        a second HIP stream NEXT TO THE FOLLOWING ENV STEP - the updates are latency chains of small launches (batch 1024: 32 workgroups), the env step is one long
        launch that issues at 60 % of its SIMDs' rate; serialised they cost 2.5 + 1.5 ms per lock step.  The behaviour copy is re-loaded every `load_freq` lock steps from
        the parameters as they were two steps earlier (so that the collector never waits for the learner).  Ordering that keeps the replay consistent: the updates of
        step t wait for the replay write of step t and draw all their batches first; the replay write of step t + 1 waits for those draws (a ring slot is never
        overwritten while a batch is being gathered from it)."""
        L, env = self.learner, self.env
        if int(load_freq) < 3:      # a snapshot is written two steps before it is used: with a period >= 3 the two buffers never hold two pending snapshots
            import warnings
            warnings.warn("td3_async: --initial_load_freq %d raised to 3 (the behaviour copy is double-buffered two lock steps ahead of its use)" % int(load_freq))
        if getattr(self, "param_noise", False):
            raise ValueError("td3_async: --param_noise is not supported by the asynchronous collector (the behaviour copy is a plain snapshot of the actor)")
        load_freq = max(3, int(load_freq))
        if self.obs is None:
            self.obs = env.reset().clone()
        main = torch.cuda.current_stream(self.device)
        if getattr(self, "_upd", None) is None:
            self._upd = torch.cuda.Stream(device=self.device)
            self.behav = [engine.Mlp(50, self.hidden, 10, self.device), engine.Mlp(50, self.hidden, 10, self.device)]
            self.behav[0].params.copy_(L.actor.params); self._cur = 0; self._async_t = 0; self._snap = {}
            self._stats = torch.zeros(3, dtype=torch.float64, device=self.device)
        upd = self._upd
        upd.wait_stream(main)
        self._stats.zero_(); n_upd = 0
        ev_sampled = None
        for _ in range(steps):
            t = self._async_t
            if t in self._snap:                                   # re-load the behaviour copy (async_td3.py:205-213): written by the learner stream two steps ago
                main.wait_event(self._snap.pop(t)); self._cur ^= 1
            a = torch.tanh(self.behav[self._cur].forward(self.obs))
            if self.act_noise != 0:
                a = (a + torch.randn(self.N, 10, device=self.device, generator=self.gen) * self.act_noise).clamp(-1, 1)      # per dimension (:253-256)
            nxt, rew, done, fin = env.step(a)
            ended = done != 0
            s2 = torch.where(ended.view(-1, 1), fin, nxt)
            if ev_sampled is not None:
                main.wait_event(ev_sampled)                       # the previous step's batches have been gathered
            self.replay.add(self.obs, s2, a, rew, (~ended).float())
            ev_added = main.record_event()
            self.obs = nxt.clone()
            if self.replay.size >= self.batch_size:
                with torch.cuda.stream(upd):
                    upd.wait_event(ev_added)
                    batches = []
                    for _k in range(self.updates_per_step):
                        b = self.replay.sample(self.batch_size, self.gen)
                        batches.append(b + (torch.randn(self.batch_size, 10, device=self.device, generator=self.gen) * self.policy_noise,))
                    ev_sampled = upd.record_event()
                    for s, sn, ac, r, nd, noise in batches:
                        st, _ = L.train_step(s, ac, sn, r, nd, noise, self.it, self.discount, self.tau, self.noise_clip, self.policy_freq)
                        self._stats += st; n_upd += 1; self.it += 1
                    if (t + 2) % load_freq == 0:                  # the snapshot the collector switches to at step t + 2
                        self.behav[self._cur ^ 1].params.copy_(L.actor.params)
                        self._snap[t + 2] = upd.record_event()
            self._async_t += 1
        main.wait_stream(upd)                                     # the caller (evaluation, checkpoint) sees the learner's parameters
        self.total_steps += steps * self.N
        s = (self._stats / max(n_upd, 1)).cpu().numpy()
        return dict(q_loss=float(s[0]), avg_q1=float(s[1] / self.batch_size), avg_q2=float(s[2] / self.batch_size), updates=n_upd)

    @torch.no_grad()
    def reference_round(self, max_traj_len, explore_fn=None, index_fn=None, smooth_fn=None):
        """ONE pass of the reference's synchronous loop body with its own collection semantics (sync_td3.py:304-313), for parity runs and for users
        who want the reference's schedule: every env is one of its workers and collects ONE whole episode from a fresh reset
        (collect_experience, :59-98: action = clip(actor(s) + one N(0, act_noise) scalar, -1, 1); done_bool = 1 at the time limit too, :82);
        the episodes are merged worker-major (np.concatenate of the workers' lists, :52) into the ring replay (remote_replay.py:66-74); then
        TD3.train runs for as many iterations as transitions were collected (:313), its iteration counter restarting at 0 (:137), every batch
        sampled uniformly with replacement from everything collected so far.  explore_fn(env, step) / index_fn(it) / smooth_fn(it) replace the
        random draws (golden G20c replays the reference's streams through them).  Returns the reference's train() statistics: avg_q1 (mean over
        iterations and batch), q_loss (mean over iterations), pi_loss (sum over the policy updates / ALL iterations, as :203 does)."""
        env, L, N, dev = self.env, self.learner, self.N, self.device
        obs = env.reset().clone()
        alive = torch.ones(N, dtype=torch.bool, device=dev)
        steps = []
        t = 0
        while bool(alive.any()) and t < max_traj_len:
            a = L.act(obs)
            if self.act_noise != 0:
                eps = (torch.tensor([[explore_fn(e, t)] for e in range(N)], dtype=torch.float32, device=dev) if explore_fn is not None
                       else torch.randn(N, 1, device=dev, generator=self.gen) * self.act_noise)
                a = (a + eps).clamp(-1, 1)
            nxt, rew, done, fin = env.step(a)
            term = done != 0
            ended = term | (t + 1 == max_traj_len)
            s2 = torch.where(term.view(-1, 1), fin, nxt)
            steps.append((obs.clone(), s2.clone(), a.clone(), rew.clone(), (~ended).float(), alive.clone()))
            alive = alive & ~term
            obs = nxt.clone(); t += 1
        n_new = 0
        for e in range(N):                                   # worker-major merge
            rows = [k for k in range(len(steps)) if bool(steps[k][5][e])]
            if not rows:
                continue
            g = lambda j: torch.stack([steps[k][j][e] for k in rows])
            self.replay.add(g(0), g(1), g(2), g(3), g(4))
            n_new += len(rows)
        self.total_steps += n_new
        q_loss = avg_q1 = pi_loss = 0.0
        for it in range(n_new):
            ind = None if index_fn is None else torch.as_tensor(index_fn(it), dtype=torch.long, device=dev)
            s, sn, ac, r, nd = self.replay.sample(self.batch_size, self.gen, ind)
            noise = (torch.as_tensor(smooth_fn(it), dtype=torch.float32, device=dev) if smooth_fn is not None
                     else torch.randn(self.batch_size, 10, device=dev, generator=self.gen) * self.policy_noise)
            st, pl = L.train_step(s, ac, sn, r, nd, noise, it, self.discount, self.tau, self.noise_clip, self.policy_freq)
            stc = st.cpu().numpy()
            q_loss += float(stc[0]); avg_q1 += float(stc[1]) / self.batch_size
            if pl is not None:
                pi_loss += float(pl)
        self.it += n_new
        k = max(n_new, 1)
        return dict(transitions=n_new, avg_q1=avg_q1 / k, q_loss=q_loss / k, pi_loss=pi_loss / k)

    @torch.no_grad()
    def evaluate(self, n_envs=256, max_traj_len=400):
        """evaluate_policy (sync_td3.py:23-46) as one batch of deterministic episodes on a fresh env batch."""
        from .vecenv import CassieVecEnv
        from .eval import evaluate
        ev = CassieVecEnv(n_envs=n_envs, max_traj_len=max_traj_len, dynamics_randomization=False, device=self.device.index or 0)
        ev.reset()
        out = evaluate(lambda o: self.learner.act(o), ev, max_steps=max_traj_len)
        ev.close()
        return float(out["returns"].mean()), float(out["lengths"].mean())


def run_experiment(args):
    from .vecenv import CassieVecEnv
    from .log import create_logger
    torch.manual_seed(args.seed); np.random.seed(args.seed)
    env = CassieVecEnv(n_envs=args.n_envs, reward=args.reward, max_traj_len=args.max_traj_len, seed=args.seed, env_name=args.env_name)
    logger = create_logger(args)
    algo = TD3(env, logger.dir, hidden=args.hidden, a_lr=args.a_lr, c_lr=args.c_lr, discount=args.discount, tau=args.tau, policy_noise=args.policy_noise,
               noise_clip=args.noise_clip, policy_freq=args.policy_freq, act_noise=args.act_noise, batch_size=args.batch_size,
               updates_per_step=args.updates_per_step, replay_size=args.replay_size, seed=args.seed, param_noise=getattr(args, "param_noise", False),
               noise_scale=getattr(args, "noise_scale", 0.3), one_launch_updates=getattr(args, "td3_one_launch", None))
    algo.init_networks(args.seed)
    updates = 0
    ret, eplen = algo.evaluate()
    logger.add_scalar("Test/Return", ret, updates); logger.add_scalar("Test/Eplen", eplen, updates)
    while algo.total_steps < args.max_timesteps:
        t0 = time.time()
        if getattr(args, "async_mode", False):
            out = algo.collect_and_train_async(args.collect_steps, load_freq=getattr(args, "initial_load_freq", 10))
        else:
            out = algo.collect_and_train(args.collect_steps)
        torch.cuda.synchronize(); dt = time.time() - t0
        updates += out["updates"]
        for k in ("avg_q1", "avg_q2", "q_loss"):
            logger.add_scalar("Train/" + k, out[k], updates)
        print("Total T: %d  updates %d  q_loss %.4f  avg_q1 %.3f  (%.0f env-steps/s, %.0f updates/s)" % (
            algo.total_steps, updates, out["q_loss"], out["avg_q1"], args.collect_steps * args.n_envs / dt, out["updates"] / dt))
        if (algo.total_steps // (args.collect_steps * args.n_envs)) % args.eval_every == 0:
            ret, eplen = algo.evaluate()
            logger.add_scalar("Test/Return", ret, updates); logger.add_scalar("Test/Eplen", eplen, updates); logger.add_scalar("Misc/Timesteps", algo.total_steps, updates)
            logger.add_scalar("Misc/ReplaySize", algo.replay.size, updates)
            print("Total T: %d\tEval Return: %.2f\t Eval Eplen: %.1f" % (algo.total_steps, ret, eplen))
            algo.save()
    return algo
