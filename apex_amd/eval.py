"""Batched deterministic evaluation on the HIP env: the build-side counterpart of the
reference's per-env eval loops (rl/algos/ppo.py:171 eval rollouts; tools/eval_*-style scoring on reset_for_test /
update_speed / step_basic): every env runs ONE episode with the policy mean as action."""
import torch


class RecurrentActor:
    """Callable wrapper that steps an apex_amd.engine.Lstm actor with one carried (h, c) per env (zero at construction / reset_state):
    what `policy(state)` does for a Gaussian_LSTM_Actor between init_hidden_state() calls (rl/policies/actor.py:270-281)."""

    def __init__(self, net, obs_mean=None, obs_std=None):
        self.net, self.mean, self.std, self.hc = net, obs_mean, obs_std, None

    def reset_state(self):
        self.hc = None

    def __call__(self, obs):
        if self.hc is None or self.hc.shape[2] != obs.shape[0]:
            self.hc = torch.zeros(self.net.L, 2, obs.shape[0], self.net.H, device=obs.device)
        x = obs if self.mean is None else (obs - self.mean) / self.std
        return self.net.forward(x.contiguous(), hc=self.hc)


@torch.no_grad()
def evaluate(actor, env, obs_mean=None, obs_std=None, speed=None, side_speed=0.0, max_steps=None, basic=False):
    """actor: apex_amd.engine.Mlp (or any callable obs -> action mean on device tensors).

    basic=False: training semantics (CassieEnv.step: reward, termination, command resampling) from CassieEnv.reset_for_test
    (+ update_speed when `speed` is given).  basic=True: CassieEnv.step_basic at the fixed command (no reward, no resampling);
    an env counts as fallen when the pelvis height qpos[2] drops below 0.4 m (cassie.py:462's threshold, as the harnesses test it).
    Returns dict(returns [N] (zeros when basic), lengths [N], terminated [N] bool, truncated [N] bool)."""
    n = env.n_envs
    max_steps = max_steps or env.max_traj_len
    if hasattr(actor, "reset_state"):
        actor.reset_state()
    env.reset_for_test()
    if speed is not None:
        env.update_speed(speed, side_speed)
    obs = env.obs
    fwd = (lambda o: actor.forward(o, obs_mean, obs_std)) if hasattr(actor, "forward") else actor
    ret = torch.zeros(n, device=env.device); length = torch.zeros(n, device=env.device)
    alive = torch.ones(n, dtype=torch.bool, device=env.device)
    term = torch.zeros(n, dtype=torch.bool, device=env.device); trunc = torch.zeros_like(term)
    for t in range(max_steps):
        if basic:
            obs = env.step_basic(fwd(obs))
            fell = env.get_field("qpos")[:, 2] < 0.4
            length += alive.float()
            term |= alive & fell
            alive &= ~fell
        else:
            obs, rew, done, _ = env.step(fwd(obs), auto_reset=False)
            ret += torch.where(alive, rew, torch.zeros_like(rew)); length += alive.float()
            term |= alive & (done == 1); trunc |= alive & (done == 2)
            alive &= done == 0
        if t % 16 == 15 and not bool(alive.any()):
            break
    if basic:
        trunc = alive.clone()
    return dict(returns=ret, lengths=length, terminated=term, truncated=trunc)


@torch.no_grad()
def compute_perturbs(actor, make_env, obs_mean=None, obs_std=None, wait_time=4.0, perturb_duration=0.2, perturb_size=100.0,
                     perturb_incr=10.0, num_angles=4, n_sizes=40, num_phases=33, speed=0.5, perturb_body="cassie-pelvis"):
    """The reference's push-recovery sweep (tools/eval_perturb.py:99-160 `compute_perturbs`, :15-85 `perturb_worker`) as ONE batch.

    Reference, per (angle, phase): grow the push by `perturb_incr` from `perturb_size` until the robot falls; every trial is
    reset_for_test(full_reset=True), `env.speed = 0.5`, 2 (phaselen + 1) + phase policy steps, then the wrench
    [F cos a, F sin a, 0, 0, 0, 0] on `perturb_body` (default cassie-pelvis, eval_perturb.py:104) for `perturb_duration` s of sim time, then up to `wait_time` s without it, failed
    when qpos[2] < 0.4 during the wait; result max_force[phase, angle] = (first failing size) - incr.
    Here: one env per (angle, phase, size) trial, all trials in lock step; `n_sizes` bounds the sweep (a cell that never fails
    reports the largest size tried).  make_env(n) -> CassieVecEnv with n envs (dynamics_randomization off).
    Returns (max_force [num_phases, num_angles] float32 numpy, fell [num_phases, num_angles, n_sizes] bool numpy)."""
    import math
    import numpy as np
    n_trials = num_angles * num_phases * n_sizes
    n = ((n_trials + 63) // 64) * 64                                      # the env handle wants a multiple of 64
    env = make_env(n)
    dev = env.device
    idx = torch.arange(n, device=dev)
    valid = idx < n_trials
    a_i = (idx // (num_phases * n_sizes)) % num_angles
    p_i = (idx // n_sizes) % num_phases
    s_i = idx % n_sizes
    angle = -2.0 * math.pi * a_i.float() / num_angles                   # perturb_dirs = -2 pi linspace(0, 1, num_angles + 1)
    size = perturb_size + perturb_incr * s_i.float()
    wrench = torch.zeros(n, 6, device=dev)
    wrench[:, 0] = size * torch.cos(angle); wrench[:, 1] = size * torch.sin(angle)
    fwd = (lambda o: actor.forward(o, obs_mean, obs_std)) if hasattr(actor, "forward") else actor
    dt = env.simrate * 0.0005
    n_push = int(math.ceil(perturb_duration / dt - 1e-9)); n_wait = int(math.ceil(wait_time / dt - 1e-9))
    obs = env.reset_for_test(full_reset=True)
    env.set_command(speed=speed)
    for _ in range(2 * num_phases):                                      # two cycles to settle into the gait
        obs, _, _, _ = env.step(fwd(obs), auto_reset=False)
    fell = torch.zeros(n, dtype=torch.bool, device=dev)
    zero = torch.zeros_like(wrench)
    for k in range(num_phases - 1 + n_push + n_wait):
        pushing = (k >= p_i) & (k < p_i + n_push)
        env.apply_force(torch.where(pushing.view(n, 1), wrench, zero), perturb_body)
        obs, _, _, _ = env.step(fwd(obs), auto_reset=False)
        waiting = (k >= p_i + n_push) & (k < p_i + n_push + n_wait)
        z = env.get_field("qpos")[:, 2]
        fell |= waiting & (z < 0.4)
    env.apply_force(zero, perturb_body)
    fell = (fell & valid)[:n_trials].view(num_angles, num_phases, n_sizes).permute(1, 0, 2).cpu().numpy()
    first = np.where(fell.any(-1), fell.argmax(-1), n_sizes)            # index of the first failing size (n_sizes = never)
    max_force = (perturb_size + perturb_incr * first - perturb_incr).astype(np.float32)
    return max_force, fell


def _yaw_unrotate_obs(obs, orient_add):
    """The harness-side command rotation of tools/test_commands.py:94-107: pelvis quaternion obs[1:5] <- q_yaw^-1 * q (sign fixed to
    w >= 0) and translational velocity obs[15:18] <- rotated by q_yaw^-1, on [N, 50] device tensors; orient_add [N]."""
    c, s = torch.cos(0.5 * orient_add), torch.sin(0.5 * orient_add)          # q_yaw = (c, 0, 0, s); inverse = (c, 0, 0, -s)
    w, x, y, z = obs[:, 1], obs[:, 2], obs[:, 3], obs[:, 4]
    nw, nx, ny, nz = c * w + s * z, c * x + s * y, c * y - s * x, c * z - s * w
    sign = torch.where(nw < 0, -torch.ones_like(nw), torch.ones_like(nw))
    out = obs.clone()
    out[:, 1], out[:, 2], out[:, 3], out[:, 4] = nw * sign, nx * sign, ny * sign, nz * sign
    cy, sy = torch.cos(orient_add), torch.sin(orient_add)
    vx, vy = obs[:, 15], obs[:, 16]
    out[:, 15], out[:, 16] = cy * vx + sy * vy, -sy * vx + cy * vy            # R_z(-yaw) v
    return out


@torch.no_grad()
def eval_commands(actor, make_env, obs_mean=None, obs_std=None, num_steps=200, num_commands=4, max_speed=3.0, min_speed=0.0,
                  num_iters=256, seed=0):
    """The reference's command-following test (tools/test_commands.py:56-122 `eval_worker.run_test`, :124-172 schedules) with every
    iteration as one env of a lock-step batch: reset_for_test(full_reset=True), speed 0.5, then every `num_steps` steps a new speed
    (previous +- U[0.4, 1.3], reflected into [min_speed, max_speed]) and, half a period later, a yaw command change of
    +- U[pi/6, pi/3] applied to the policy input; failed when qpos[2] < 0.4.  Returns save_data [num_iters, 6] float32 numpy with the
    reference's columns: passed, (-1 | 0 = failed in the half period after a speed change, 1 = after an orientation change), speed,
    orient_add, last speed change, last orientation change.  phase_add = 1.5 above 1.4 m/s (test_commands.py:85-88) is carried per env."""
    import numpy as np
    n = ((num_iters + 63) // 64) * 64
    env = make_env(n)
    dev = env.device
    g = torch.Generator(device=dev); g.manual_seed(seed)
    U = lambda lo, hi, *shape: lo + (hi - lo) * torch.rand(*shape, device=dev, generator=g)
    sign = lambda *shape: torch.where(torch.rand(*shape, device=dev, generator=g) < 0.5, -1.0, 1.0)
    speeds = torch.zeros(n, num_commands, device=dev); speeds[:, 0] = 0.5
    for i in range(num_commands - 1):
        add = sign(n) * U(0.4, 1.3, n)
        nxt = speeds[:, i] + add
        add = torch.where((nxt < min_speed) | (nxt > max_speed), -add, add)
        speeds[:, i + 1] = speeds[:, i] + add
    orients = U(np.pi / 6, np.pi / 3, n, num_commands) * sign(n, num_commands)
    fwd = (lambda o: actor.forward(o, obs_mean, obs_std)) if hasattr(actor, "forward") else actor
    obs = env.reset_for_test(full_reset=True)
    env.set_command(speed=0.5, side_speed=0.0, phase_add=1.0)
    orient_add = torch.zeros(n, device=dev); speed = torch.full((n,), 0.5, device=dev)
    passed = torch.ones(n, dtype=torch.bool, device=dev)
    data = torch.zeros(n, 6, device=dev)
    count, speed_ind, orient_ind = 0, 1, 0
    last_dspeed = torch.zeros(n, device=dev); last_dorient = torch.zeros(n, device=dev)
    while not (speed_ind == num_commands and orient_ind == num_commands and count == num_steps):
        if count == num_steps:
            count = 0
            new = speeds[:, speed_ind].clamp(min_speed, max_speed)
            last_dspeed = new - speeds[:, max(0, speed_ind - 1)]
            speed = new
            env.set_command(speed=speed, phase_add=torch.where(speed > 1.4, 1.5, 1.0))      # test_commands.py:85-88
            speed_ind += 1
        elif count == num_steps // 2:
            last_dorient = orients[:, orient_ind]
            orient_add = orient_add + last_dorient
            orient_ind += 1
        obs, _, _, _ = env.step(fwd(_yaw_unrotate_obs(obs, orient_add)), auto_reset=False)
        count += 1
        fell = passed & (env.get_field("qpos")[:, 2] < 0.4)
        if bool(fell.any()):
            row = torch.stack([torch.zeros(n, device=dev), torch.full((n,), float(count // (num_steps // 2)), device=dev), speed, orient_add,
                               last_dspeed, last_dorient], 1)
            data = torch.where(fell.view(n, 1), row, data)
            passed &= ~fell
        if not bool(passed.any()):
            break
    ok = torch.zeros(n, 6, device=dev); ok[:, 0] = 1.0; ok[:, 1] = -1.0
    data = torch.where(passed.view(n, 1), ok, data)
    return data[:num_iters].cpu().numpy()
