"""numpy restatement of the reference's learner arithmetic (TEST INFRASTRUCTURE, see oracle/__init__.py).

Pinned against golden vectors produced by the reference itself (tests/golden/g1..g5, generator:
tools/refprobe/gen_golden_learner.py); tests/test_oracle_learner.py holds the check.

Each function cites the reference file:line it follows (paths relative to osudrl/apex).
"""
import numpy as np

F32 = np.float32
LOG_SQRT_2PI = 0.5 * np.log(2.0 * np.pi)


# ------------------------------------------------------------------------------------------------ returns
def discounted_returns(rewards, lens, last_vals, gamma):
    """rl/algos/ppo.py:73-89 PPOBuffer.finish_path, for a concatenation of trajectories.

    R <- gamma*R + r backwards inside each trajectory, seeded with last_val = (not done)*V(s_T) (ppo.py:184).
    `lam`/`use_gae` are never read by the reference (ppo.py:50,103,112) => this IS its "GAE" (lambda = 1).
    fp64 scan (rewards are np.float64 in the reference), cast to fp32 only at ppo.py:393.
    """
    rewards = np.asarray(rewards, dtype=np.float64)
    out = np.empty_like(rewards)
    start = 0
    for T, lv in zip(lens, last_vals):
        R = np.float32(lv)      # last_val is a float32 array in the reference: the FIRST gamma*R is an fp32 product
        for t in range(start + T - 1, start - 1, -1):
            R = (np.float32(gamma) * R if isinstance(R, np.float32) else gamma * R) + rewards[t]
            R = float(R)
            out[t] = R
        start += T
    return out


def returns_scan_grid_boot(rew, end, boot, last_val, gamma):
    """[T,N] layout.  end[t,n] != 0 marks the last step of an episode; boot[t,n] is the bootstrap value used there
    ((not done)*V(s_{t+1}), ppo.py:183-184).  Column tail uses last_val[n].  fp64 like finish_path."""
    T, N = rew.shape
    out = np.empty((T, N), dtype=np.float64)
    R = np.asarray(last_val, dtype=np.float32).astype(np.float64)
    fresh = np.ones(N, dtype=bool)          # R currently holds an fp32 bootstrap value (first product is fp32)
    g32 = np.float32(gamma)
    for t in range(T - 1, -1, -1):
        is_end = end[t] != 0
        R = np.where(is_end, boot[t].astype(np.float32).astype(np.float64), R)
        fresh = fresh | is_end
        prod = np.where(fresh, (g32 * R.astype(np.float32)).astype(np.float64), gamma * R)
        R = prod + rew[t].astype(np.float64)
        fresh[:] = False
        out[t] = R
    return out


def normalize_advantages(returns, values, eps=1e-5):
    """rl/algos/ppo.py:395-396: adv = ret - val; (adv - mean) / (std_unbiased + eps), fp32 torch semantics."""
    adv = (np.asarray(returns, F32) - np.asarray(values, F32)).astype(F32)
    mean = adv.astype(np.float64).mean()
    n = adv.size
    var = ((adv.astype(np.float64) - mean) ** 2).sum() / (n - 1)
    return ((adv - F32(mean)) / (F32(np.sqrt(var)) + F32(eps))).astype(F32)


# ------------------------------------------------------------------------------------------------ networks
def mlp_forward(W, x, keep=False):
    """ReLU MLP: rl/policies/actor.py:186-191 / critic.py:69-72.  W = [W0,b0,W1,b1,W2,b2], torch [out,in]."""
    acts = [x]
    h = x
    nl = len(W) // 2
    for i in range(nl):
        h = h @ W[2 * i].T + W[2 * i + 1]
        if i < nl - 1:
            h = np.maximum(h, 0)
        acts.append(h)
    return (h, acts) if keep else h


def actor_mean(W, obs, obs_mean, obs_std):
    """Gaussian_FF_Actor._get_dist_params (actor.py:183-203): normalise, trunk, mean head."""
    return mlp_forward(W, ((obs - obs_mean) / obs_std).astype(obs.dtype))


def critic_value(W, obs, obs_mean=None, obs_std=None, training=True):
    """FF_V.forward (critic.py:65-74): normalises ONLY when not training."""
    if not training:
        obs = (obs - obs_mean) / obs_std
    return mlp_forward(W, obs)


def gaussian_logp(mu, sd, act):
    """torch.distributions.Normal.log_prob summed over the action dim (ppo.py:287-289)."""
    return (-((act - mu) ** 2) / (2 * sd * sd) - np.log(sd) - LOG_SQRT_2PI).sum(-1, keepdims=True)


def gaussian_entropy(sd):
    return 0.5 + LOG_SQRT_2PI + np.log(sd)


def mirror_matrix(mirrored):
    """rl/envs/wrappers.py:70-77 _get_symmetry_matrix (0.1 encodes '+index 0')."""
    n = len(mirrored)
    m = np.zeros((n, n))
    for i, j in zip(range(n), np.abs(np.array(mirrored).astype(int))):
        m[i, j] = np.sign(mirrored[i])
    return m


def mirror_clock_observation(obs, M_obs, clock_inds):
    """wrappers.py:59-67: obs @ M, then clock columns -> sin(arcsin(c) + pi) (== -c up to 2.4e-7)."""
    mo = obs @ M_obs
    for ci in clock_inds:
        mo[:, ci] = np.sin(np.arcsin(mo[:, ci]) + np.pi)
    return mo


def _mlp_backward(W, acts, dout):
    """Gradients of an MLP wrt its parameters given d(loss)/d(output); returns list like W and d(input)."""
    nl = len(W) // 2
    grads = [None] * len(W)
    d = dout
    for i in range(nl - 1, -1, -1):
        a_in = acts[i]
        grads[2 * i] = d.T @ a_in
        grads[2 * i + 1] = d.sum(0)
        d = d @ W[2 * i]
        if i > 0:
            d = d * (acts[i] > 0)
    return grads, d


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (ppo.py:326,335): scale by max_norm/(total+1e-6) when < 1."""
    total = np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads))
    coef = max_norm / (total + 1e-6)
    if coef < 1:
        grads = [g * coef for g in grads]
    return grads, total


class Adam:
    """torch.optim.Adam defaults (betas .9/.999, no amsgrad / weight decay); ppo.py:355-356."""

    def __init__(self, params, lr=1e-4, eps=1e-5):
        self.lr, self.eps, self.b1, self.b2 = lr, eps, 0.9, 0.999
        self.m = [np.zeros_like(p) for p in params]
        self.v = [np.zeros_like(p) for p in params]
        self.t = 0

    def step(self, params, grads):
        self.t += 1
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        out = []
        for i, (p, g) in enumerate(zip(params, grads)):
            self.m[i] = self.b1 * self.m[i] + (1 - self.b1) * g
            self.v[i] = self.b2 * self.v[i] + (1 - self.b2) * g * g
            denom = np.sqrt(self.v[i]) / np.sqrt(bc2) + self.eps
            out.append(p - (self.lr / bc1) * self.m[i] / denom)
        return out


def ppo_update(actor, old_actor, critic, opt_a, opt_c, obs, act, ret, adv, obs_mean, obs_std, fixed_std,
               clip=0.2, entropy_coeff=0.0, grad_clip=0.05, M_obs=None, M_act=None, clock_inds=(46, 47),
               dtype=np.float64):
    """rl/algos/ppo.py:276-345 PPO.update_policy, analytic gradients.

    Returns (scalars[6], new_actor, new_critic); scalars = actor_loss, entropy, critic_loss, ratio.mean, kl.mean,
    mirror_loss exactly as the reference returns them (ppo.py:345).
    """
    c = lambda L: [np.asarray(x, dtype) for x in L]
    actor, old_actor, critic = c(actor), c(old_actor), c(critic)
    obs, act, ret, adv = (np.asarray(x, dtype) for x in (obs, act, ret, adv))
    obs_mean, obs_std = np.asarray(obs_mean, dtype), np.asarray(obs_std, dtype)
    B, A = act.shape
    sd = dtype(fixed_std)

    # critic (train mode: raw obs, critic.py:66-67)
    v, acts_c = mlp_forward(critic, obs, keep=True)
    critic_loss = 0.5 * ((ret - v) ** 2).mean()
    dv = -(ret - v) / B
    g_c, _ = _mlp_backward(critic, acts_c, dv)

    # actor
    xn = (obs - obs_mean) / obs_std
    mu, acts_a = mlp_forward(actor, xn, keep=True)
    mu_old = mlp_forward(old_actor, xn)
    logp = gaussian_logp(mu, sd, act)
    logp_old = gaussian_logp(mu_old, sd, act)
    ratio = np.exp(logp - logp_old)
    cpi = ratio * adv
    clipped = np.clip(ratio, 1 - clip, 1 + clip) * adv
    actor_loss = -np.minimum(cpi, clipped).mean()
    inside = (ratio >= 1 - clip) & (ratio <= 1 + clip)
    w_cpi = np.where(cpi < clipped, 1.0, np.where(cpi == clipped, 0.5, 0.0))
    w_clip = 1.0 - w_cpi
    dsur_dratio = w_cpi * adv + w_clip * adv * inside          # torch.min ties split 1/2 + 1/2
    dlogp = -(dsur_dratio * ratio) / B                          # d(actor_loss)/d(logp)
    dmu = dlogp * (act - mu) / (sd * sd)
    entropy = gaussian_entropy(sd)
    # entropy penalty has zero gradient for a fixed std (ppo.py:299)

    mirror_loss = 0.0
    g_a, _ = _mlp_backward(actor, acts_a, dmu)
    if M_obs is not None:
        M_obs = np.asarray(M_obs, dtype); M_act = np.asarray(M_act, dtype)
        mobs = mirror_clock_observation(obs.copy(), M_obs, clock_inds)
        xm = (mobs - obs_mean) / obs_std
        mu_m, acts_m = mlp_forward(actor, xm, keep=True)
        diff = mu - mu_m @ M_act
        mirror_loss = 0.4 * (diff ** 2).mean()
        dd = 0.8 * diff / (B * A)
        g1, _ = _mlp_backward(actor, acts_a, dd)
        g2, _ = _mlp_backward(actor, acts_m, -(dd @ M_act.T))
        g_a = [a + b + cc for a, b, cc in zip(g_a, g1, g2)]

    g_a, _ = clip_grad_norm(g_a, grad_clip)
    new_actor = opt_a.step(actor, g_a)
    g_c, _ = clip_grad_norm(g_c, grad_clip)
    new_critic = opt_c.step(critic, g_c)

    kl = (0.5 * ((mu - mu_old) / sd) ** 2).mean()               # kl_divergence(Normal, Normal), equal std
    scal = np.array([actor_loss, entropy, critic_loss, ratio.mean(), kl, mirror_loss], dtype=np.float64)
    return scal, new_actor, new_critic


def normalization_params(policy_W, envs, noise, noise_std=1.0):
    """get_normalization_params (rl/envs/normalize.py:11-48): every worker records the state BEFORE each step, acts with
    policy(state) + noise_std * N(0, 1) (un-normalised policy), resets on done (the terminal state is dropped);
    returns mean and sqrt(var + 1e-8) over all recorded states.  envs: objects with reset() / step(a) -> (s, r, done, info);
    noise [workers, steps, act_dim] replays the workers' draws."""
    states = []
    for w, env in enumerate(envs):
        s = env.reset()
        for t in range(noise.shape[1]):
            states.append(np.asarray(s, dtype=np.float64).copy())
            a = mlp_forward(policy_W, np.asarray(s, dtype=np.float32)[None].astype(np.float64))[0] + noise_std * noise[w, t]
            s, _, done, _ = env.step(a.astype(np.float32).astype(np.float64))
            if done:
                s = env.reset()
    states = np.array(states)
    return states.mean(0), np.sqrt(states.var(0) + 1e-8)
