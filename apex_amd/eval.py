"""Batched deterministic evaluation on the HIP env: the build-side counterpart of the
reference's per-env eval loops (rl/algos/ppo.py:171 eval rollouts; tools/eval_*-style scoring on reset_for_test /
update_speed / step_basic): every env runs ONE episode with the policy mean as action."""
import torch


@torch.no_grad()
def evaluate(actor, env, obs_mean=None, obs_std=None, speed=None, side_speed=0.0, max_steps=None, basic=False):
    """actor: apex_amd.engine.Mlp (or any callable obs -> action mean on device tensors).

    basic=False: training semantics (CassieEnv.step: reward, termination, command resampling) from CassieEnv.reset_for_test
    (+ update_speed when `speed` is given).  basic=True: CassieEnv.step_basic at the fixed command (no reward, no resampling);
    an env counts as fallen when the estimated pelvis height obs[0] + 0.0818 drops below 0.4 m (cassie.py:462's threshold).
    Returns dict(returns [N] (zeros when basic), lengths [N], terminated [N] bool, truncated [N] bool)."""
    n = env.n_envs
    max_steps = max_steps or env.max_traj_len
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
            fell = (obs[:, 0] + 0.0818) < 0.4
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
                     perturb_incr=10.0, num_angles=4, n_sizes=40, num_phases=33, speed=0.5):
    """The reference's push-recovery sweep (tools/eval_perturb.py:99-160 `compute_perturbs`, :15-85 `perturb_worker`) as ONE batch.

    Reference, per (angle, phase): grow the push by `perturb_incr` from `perturb_size` until the robot falls; every trial is
    reset_for_test(full_reset=True), `env.speed = 0.5`, 2 (phaselen + 1) + phase policy steps, then the wrench
    [F cos a, F sin a, 0, 0, 0, 0] on cassie-pelvis for `perturb_duration` s of sim time, then up to `wait_time` s without it, failed
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
        env.apply_force(torch.where(pushing.view(n, 1), wrench, zero))
        obs, _, _, _ = env.step(fwd(obs), auto_reset=False)
        waiting = (k >= p_i + n_push) & (k < p_i + n_push + n_wait)
        z = env.get_field("qpos")[:, 2]
        fell |= waiting & (z < 0.4)
    env.apply_force(zero)
    fell = (fell & valid)[:n_trials].view(num_angles, num_phases, n_sizes).permute(1, 0, 2).cpu().numpy()
    first = np.where(fell.any(-1), fell.argmax(-1), n_sizes)            # index of the first failing size (n_sizes = never)
    max_force = (perturb_size + perturb_incr * first - perturb_incr).astype(np.float32)
    return max_force, fell
