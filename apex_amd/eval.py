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
