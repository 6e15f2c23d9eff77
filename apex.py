"""Command-line front-end reproducing the reference's `python apex.py ppo ...` (apex.py:16-39,214-255): same flag
names and defaults, same run-directory layout, same checkpoint files — backed by the MI355X engine in apex_amd/.
Extra flags (not in the reference): --n_envs (envs per GPU, default 4096) and --hidden."""
import argparse
import sys


def build_parser():
    p = argparse.ArgumentParser()
    # env flags, apex.py:16-39
    p.add_argument("--command_profile", default="clock", type=str.lower, choices=["clock", "phase", "traj"])
    p.add_argument("--input_profile", default="full", type=str.lower, choices=["full", "min"])
    p.add_argument("--simrate", default=50, type=int)
    p.add_argument("--not_dyn_random", default=True, action="store_false", dest="dyn_random")
    p.add_argument("--learn_gains", default=False, action="store_true", dest="learn_gains")
    p.add_argument("--traj", default="walking", type=str)
    p.add_argument("--not_no_delta", default=True, action="store_false", dest="no_delta")
    p.add_argument("--ik_baseline", default=False, action="store_true", dest="ik_baseline")
    p.add_argument("--not_mirror", default=True, action="store_false", dest="mirror")
    p.add_argument("--reward", default=None, type=str)
    p.add_argument("--env_name", default="Cassie-v0")
    p.add_argument("--run_name", default=None)
    p.add_argument("--exchange_reward", default=None)
    p.add_argument("--previous", type=str, default=None)
    # ppo flags, apex.py:224-250
    p.add_argument("--logdir", type=str, default="./trained_models/ppo/")
    p.add_argument("--seed", default=0, type=int)
    p.add_argument("--history", default=0, type=int)
    p.add_argument("--redis_address", type=str, default=None)
    p.add_argument("--viz_port", default=8097)
    p.add_argument("--input_norm_steps", type=int, default=10000)
    p.add_argument("--n_itr", type=int, default=10000)
    p.add_argument("--lr", type=float, default=1e-4)
    p.add_argument("--eps", type=float, default=1e-5)
    p.add_argument("--lam", type=float, default=0.95)
    p.add_argument("--gamma", type=float, default=0.99)
    p.add_argument("--anneal", default=1.0, action="store_true")
    p.add_argument("--learn_stddev", default=False, action="store_true")
    p.add_argument("--std_dev", type=int, default=-1.5)
    p.add_argument("--entropy_coeff", type=float, default=0.0)
    p.add_argument("--clip", type=float, default=0.2)
    p.add_argument("--minibatch_size", type=int, default=64)
    p.add_argument("--epochs", type=int, default=3)
    p.add_argument("--epoch_kernel", action="store_true", help="minibatches of at most 256 rows on one GPU: every epoch's optimiser steps as ONE launch (apx_ppo_epoch)")
    p.add_argument("--num_steps", type=int, default=5096)
    p.add_argument("--use_gae", type=bool, default=True)
    p.add_argument("--num_procs", type=int, default=30)
    p.add_argument("--max_grad_norm", type=float, default=0.05)
    p.add_argument("--max_traj_len", type=int, default=400)
    p.add_argument("--recurrent", action="store_true")
    p.add_argument("--bounded", type=bool, default=False)
    # engine flags (additions)
    p.add_argument("--n_envs", type=int, default=4096, help="lock-step envs per GPU (replaces Ray's num_procs)")
    p.add_argument("--eval_every", type=int, default=10, help="deterministic evaluation pass (ppo.py:464) every k iterations; 1 = the reference's cadence, 0 = off")
    p.add_argument("--eval_envs", type=int, default=256, help="envs (= episodes) of the evaluation pass")
    p.add_argument("--est_lifetime", type=int, default=None, help="env steps served by one state-estimator object: the reference builds a new CassieEnv (-> cassie_sim_init -> "
                   "state_output_setup) per PPO.sample call (rl/algos/ppo.py:152), i.e. every num_steps // num_procs env steps.  Default: 5096 // --num_procs "
                   "(5096 = the reference's --num_steps default, apex.py:244-246; this front-end's own --num_steps is rescaled to the lock-step batch and no longer "
                   "describes one worker's share) = 169 with the reference's defaults; 0 = one estimator for the whole run")
    return p


REFERENCE_NUM_STEPS = 5096      # the reference's --num_steps default (apex.py:244): what one PPO.sample fan-out collects in total


def resolve_est_lifetime(args):
    """est_lifetime of the run ("Estimator lifetime"): explicit flag, else the reference's per-worker share num_steps // num_procs
    (rl/algos/ppo.py:194: each of the n_proc workers samples min_steps // n_proc steps with ONE CassieEnv).  Stored in args, hence in experiment.pkl / .info."""
    if getattr(args, "est_lifetime", None) is None:
        args.est_lifetime = REFERENCE_NUM_STEPS // max(1, int(args.num_procs))
    if args.est_lifetime < 0:
        raise SystemExit("--est_lifetime must be >= 0")
    return args


def resolve_horizon(args, argv):
    """The reference's --num_steps default (5096) is a TOTAL per iteration for ~30 one-env workers sampling whole episodes.  On a
    lock-step batch of n_envs envs it would mean T = 2 steps per env, i.e. 2-step TD targets instead of episode returns.  When
    --num_steps is not given the horizon is therefore scaled to the batch: T = 32 steps per env (feed-forward; the headline
    configuration) or T = max_traj_len (recurrent: every trajectory must start at an episode start with zero hidden state).  An explicit
    --num_steps is honoured but a horizon below 8 steps per env is refused."""
    world = 1
    import os
    world = int(os.environ.get("WORLD_SIZE", 1))
    given = any(x == "--num_steps" or x.startswith("--num_steps=") for x in argv)
    if not given:
        per_env = args.max_traj_len if args.recurrent else 32
        args.num_steps = args.n_envs * world * per_env
        print("--num_steps not given: %d (= %d envs x %d steps per env and iteration)" % (args.num_steps, args.n_envs * world, per_env))
    if not any(x == "--minibatch_size" or x.startswith("--minibatch_size=") for x in argv) and not args.recurrent:
        # 5096 / 64 = 80 optimiser steps per epoch in the reference; on a 131 072-sample batch the headline run uses 8 steps of 16 384
        args.minibatch_size = max(64, min(16384, args.num_steps // world // 8))
        print("--minibatch_size not given: %d" % args.minibatch_size)
    T = -(-args.num_steps // (args.n_envs * world))
    if T < 8:
        raise SystemExit("--num_steps %d with --n_envs %d is a %d-step horizon per env: returns would be %d-step bootstraps, not episode returns; "
                         "raise --num_steps (>= 8 x n_envs) or lower --n_envs" % (args.num_steps, args.n_envs, T, T))
    return args


def eval_main(argv):
    """`apex.py eval --path <run dir>` (reference apex.py:257-280): score a saved actor.pt.  The reference opens a viewer /
    perturbation harness; here every env of the batch runs one deterministic episode from reset_for_test at the commanded speed
    (apex_amd/eval.py) and the statistics are printed."""
    import argparse, os
    import torch
    p = argparse.ArgumentParser()
    p.add_argument("--path", type=str, required=True)
    p.add_argument("--speed", type=float, default=None)
    p.add_argument("--side_speed", type=float, default=0.0)
    p.add_argument("--n_envs", type=int, default=256)
    p.add_argument("--max_traj_len", type=int, default=400)
    p.add_argument("--reward", type=str, default="clock")
    p.add_argument("--basic", action="store_true", help="CassieEnv.step_basic: fixed command, survival time only")
    p.add_argument("--terrain", default=None, type=str, help="height-field file (.npy, 500 x 500 like cassie/cassiemujoco/terrains/*.npy): reference apex.py:265, util/eval.py:73-76")
    a = p.parse_args(argv)
    from apex_amd.vecenv import CassieVecEnv
    from apex_amd.engine import Mlp
    from apex_amd.eval import evaluate
    env = CassieVecEnv(n_envs=a.n_envs, max_traj_len=a.max_traj_len, dynamics_randomization=False, **_run_env_kwargs(a.path, a.reward))
    if a.terrain is not None and ".npy" in a.terrain:            # env.sim = CassieSim("cassie_hfield.xml"); sim.set_hfield_data(np.load(terrain).flatten())
        import numpy as np
        path = a.terrain if os.path.exists(a.terrain) else os.path.join("./cassie/cassiemujoco/terrains/", a.terrain)
        env.set_hfield(np.load(path), size=(50.0, 50.0, 0.15))  # cassie_hfield.xml:69
    actor, mean, std = _load_actor(a.path, env.device)
    _check_obs_dim(actor, env)
    out = evaluate(actor, env, mean, std, speed=a.speed, side_speed=a.side_speed, max_steps=a.max_traj_len, basic=a.basic)
    ln, rt = out["lengths"].cpu(), out["returns"].cpu()
    print("episodes %d  mean length %.1f (min %d, max %d)  mean return %.3f  fell %d  reached the time limit %d" % (
        a.n_envs, float(ln.mean()), int(ln.min()), int(ln.max()), float(rt.mean()), int(out["terminated"].sum()), int(out["truncated"].sum())))
    return 0


def _run_env_kwargs(path, reward_flag):
    """env-defining arguments of the run that produced the checkpoint (experiment.pkl, util/log.py:57-64 / parse_previous :74-91): a policy trained with
    --command_profile phase or --history h expects that observation, not the 50-entry default"""
    import os, pickle
    kw = dict(reward=reward_flag)
    pkl = os.path.join(path, "experiment.pkl")
    if os.path.exists(pkl):
        with open(pkl, "rb") as f:
            run = pickle.load(f)
        for key, arg in (("reward", "reward"), ("command_profile", "command_profile"), ("input_profile", "input_profile"), ("history", "history"),
                         ("env_name", "env_name"), ("est_lifetime", "est_lifetime"), ("simrate", "simrate")):
            if getattr(run, arg, None) is not None:
                kw[key] = getattr(run, arg)
        if reward_flag != "clock":
            kw["reward"] = reward_flag                       # an explicit --reward wins
    return kw


def _check_obs_dim(actor, env):
    D = getattr(actor, "D", None) or getattr(getattr(actor, "net", None), "D", None)
    if D is not None and D != env.obs_dim:
        raise ValueError("the checkpoint expects %d observation entries, the env built from its experiment.pkl produces %d (command_profile / history mismatch)" % (D, env.obs_dim))


def _load_actor(path, device):
    import os
    import torch
    from apex_amd.engine import Mlp
    policy = torch.load(os.path.join(path, "actor.pt"), weights_only=False)
    params = [q.detach().numpy() for q in policy.parameters()]
    mean = torch.as_tensor(policy.obs_mean, dtype=torch.float32).to(device) if torch.is_tensor(policy.obs_mean) else None
    std = torch.as_tensor(policy.obs_std, dtype=torch.float32).to(device) if torch.is_tensor(policy.obs_std) else None
    if getattr(policy, "is_recurrent", False):        # Gaussian_LSTM_Actor: step the HIP LSTM with one carried (h, c) per env
        from apex_amd.engine import Lstm
        from apex_amd.eval import RecurrentActor
        H, L = policy.actor_layers[0].hidden_size, len(policy.actor_layers)
        net = Lstm(params[0].shape[1], H, L, params[-1].shape[0], device)
        net.load_list(params)
        return RecurrentActor(net, mean, std), None, None
    actor = Mlp(params[0].shape[1], params[0].shape[0], params[-1].shape[0], device)
    actor.load_list(params)
    return actor, mean, std


def eval_perturb_main(argv):
    """`apex.py eval_perturb --path <run dir>`: the reference's tools/eval_perturb.py sweep (largest survivable pelvis push per gait
    phase and direction), every (direction, phase, size) trial as one env of a single batch; writes eval_perturbs.npy next to
    actor.pt like the reference (test_policy.py:78-89)."""
    import argparse, os, time
    import numpy as np
    import torch
    p = argparse.ArgumentParser()
    p.add_argument("--path", type=str, required=True)
    p.add_argument("--num_angles", type=int, default=4)
    p.add_argument("--wait_time", type=float, default=4.0)
    p.add_argument("--perturb_duration", type=float, default=0.2)
    p.add_argument("--perturb_size", type=float, default=100.0)
    p.add_argument("--perturb_incr", type=float, default=10.0)
    p.add_argument("--n_sizes", type=int, default=40)
    p.add_argument("--perturb_body", type=str, default="cassie-pelvis")      # tools/eval_perturb.py:104
    p.add_argument("--reward", type=str, default="clock")
    a = p.parse_args(argv)
    from apex_amd.vecenv import CassieVecEnv
    from apex_amd.eval import compute_perturbs
    kw = _run_env_kwargs(a.path, a.reward)
    mk = lambda n: CassieVecEnv(n_envs=n, max_traj_len=100000, dynamics_randomization=False, **kw)
    actor, mean, std = _load_actor(a.path, torch.device("cuda", 0))
    _check_obs_dim(actor, mk(64))
    t0 = time.time()
    mf, fell = compute_perturbs(actor, mk, mean, std, wait_time=a.wait_time, perturb_duration=a.perturb_duration,
                                perturb_size=a.perturb_size, perturb_incr=a.perturb_incr, num_angles=a.num_angles, n_sizes=a.n_sizes,
                                perturb_body=a.perturb_body)
    dt = time.time() - t0
    np.save(os.path.join(a.path, "eval_perturbs.npy"), mf)
    print("push-recovery sweep: %d trials in %.1f s" % (fell.size, dt))
    for i in range(a.num_angles):
        print("direction %6.1f deg: max force over the %d phases  min %.0f  mean %.1f  max %.0f N" % (
            -360.0 * i / a.num_angles, mf.shape[0], mf[:, i].min(), mf[:, i].mean(), mf[:, i].max()))
    return 0


def eval_commands_main(argv):
    """`apex.py eval_commands --path <run dir>`: tools/test_commands.py's random speed / heading command schedules, one schedule per
    env of a single batch; writes eval_commands.npy next to actor.pt like test_policy.py:76 and prints report_stats-style totals."""
    import argparse, os, time
    import numpy as np
    import torch
    p = argparse.ArgumentParser()
    p.add_argument("--path", type=str, required=True)
    p.add_argument("--n_steps", type=int, default=200)
    p.add_argument("--n_commands", type=int, default=6)
    p.add_argument("--max_speed", type=float, default=3.0)
    p.add_argument("--min_speed", type=float, default=0.0)
    p.add_argument("--n_iter", type=int, default=1024)
    p.add_argument("--reward", type=str, default="clock")
    a = p.parse_args(argv)
    from apex_amd.vecenv import CassieVecEnv
    from apex_amd.eval import eval_commands
    kw = _run_env_kwargs(a.path, a.reward)
    mk = lambda n: CassieVecEnv(n_envs=n, max_traj_len=100000, dynamics_randomization=False, **kw)
    actor, mean, std = _load_actor(a.path, torch.device("cuda", 0))
    _check_obs_dim(actor, mk(64))
    t0 = time.time()
    d = eval_commands(actor, mk, mean, std, num_steps=a.n_steps, num_commands=a.n_commands, max_speed=a.max_speed,
                      min_speed=a.min_speed, num_iters=a.n_iter)
    np.save(os.path.join(a.path, "eval_commands.npy"), d)
    fail = d[d[:, 0] == 0]
    print("command test: %d schedules x %d commands in %.1f s: pass rate %.3f; failures after a speed change %d, after an orientation change %d" % (
        len(d), a.n_commands, time.time() - t0, float(d[:, 0].mean()), int((fail[:, 1] == 0).sum()), int((fail[:, 1] == 1).sum())))
    return 0


def td3_main(argv, async_mode=False):
    """`apex.py td3 ...` (row f2): synchronous TD3 (reference apex.py:140-166 flags of syncTD3 where they still apply) on the batched env with
    the replay buffer in HBM.  --n_envs / --collect_steps / --updates_per_step replace the Ray worker counts.
    `apex.py td3_async ...` (reference apex.py:160-211, rl/algos/async_td3.py): collection with a periodically re-loaded copy of the policy (--initial_load_freq lock
    steps), per-dimension action noise, the updates on a second HIP stream next to the following env step (apex_amd/td3.py::collect_and_train_async)."""
    import argparse
    p = argparse.ArgumentParser()
    p.add_argument("--env_name", default="Cassie-v0"); p.add_argument("--reward", default="clock", type=str)
    p.add_argument("--seed", type=int, default=0); p.add_argument("--logdir", type=str, default="./trained_models/td3_async/" if async_mode else "./trained_models/syncTD3/")
    p.add_argument("--initial_load_freq", type=int, default=10)      # (td3_async) lock steps between two re-loads of the behaviour copy (reference apex.py:188)
    p.add_argument("--run_name", type=str, default=None); p.add_argument("--previous", type=str, default=None)
    p.add_argument("--max_timesteps", type=float, default=1e8); p.add_argument("--max_traj_len", type=int, default=400)
    p.add_argument("--a_lr", type=float, default=1e-3); p.add_argument("--c_lr", type=float, default=1e-3)
    p.add_argument("--discount", type=float, default=0.99); p.add_argument("--tau", type=float, default=0.005)
    p.add_argument("--act_noise", type=float, default=0.3); p.add_argument("--policy_noise", type=float, default=0.2)
    p.add_argument("--noise_clip", type=float, default=0.5); p.add_argument("--policy_freq", type=int, default=2)
    p.add_argument("--batch_size", type=int, default=256); p.add_argument("--hidden", type=int, default=256)
    p.add_argument("--n_envs", type=int, default=4096); p.add_argument("--collect_steps", type=int, default=32)
    p.add_argument("--updates_per_step", type=int, default=1); p.add_argument("--replay_size", type=int, default=1000000)
    p.add_argument("--td3_one_launch", action="store_true", default=None, help="(td3) the update block behind a collection as ONE launch (apx_td3_updates)")
    p.add_argument("--eval_every", type=int, default=10)
    p.add_argument("--param_noise", type=bool, default=False); p.add_argument("--noise_scale", type=float, default=0.3)      # reference apex.py:143-144
    a = p.parse_args(argv)
    a.max_timesteps = int(a.max_timesteps)
    a.async_mode = bool(async_mode)
    from apex_amd.td3 import run_experiment
    run_experiment(a)
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if argv and argv[0] == "eval":
        return eval_main(argv[1:])
    if argv and argv[0] == "td3":
        return td3_main(argv[1:])
    if argv and argv[0] == "td3_async":
        return td3_main(argv[1:], async_mode=True)
    if argv and argv[0] == "eval_commands":
        return eval_commands_main(argv[1:])
    if argv and argv[0] == "eval_perturb":
        return eval_perturb_main(argv[1:])
    if not argv or argv[0] != "ppo":
        print("Usage: python apex.py ppo [flags] | python apex.py eval --path RUN_DIR [--speed S]   (only the PPO / Cassie-v0 path is built; see DESIGN.md)")
        return 2
    args = build_parser().parse_args(argv[1:])
    if args.env_name not in ("Cassie-v0", "CassieTraj-v0") or args.learn_stddev:
        raise NotImplementedError("only Cassie-v0 / CassieTraj-v0 PPO (feed-forward or --recurrent) with fixed std is built (SURVEY.md §8)")
    from apex_amd.log import parse_previous
    from apex_amd.ppo import run_experiment
    args = parse_previous(args)
    args = resolve_horizon(args, argv[1:])
    args = resolve_est_lifetime(args)
    run_experiment(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
