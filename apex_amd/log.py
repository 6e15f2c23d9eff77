"""Run-directory layout of the reference (util/log.py:11-91): <logdir>/<env_name>/<md5(args)[:6]>-seed<seed> (or
run_name), experiment.info (sorted `key: value` lines), experiment.pkl (pickled argparse.Namespace) and a scalar
writer with SummaryWriter's add_scalar surface.  TensorBoard is used when importable; otherwise scalars go to
scalars.jsonl with the same tags."""
import hashlib
import json
import os
import pickle
from collections import OrderedDict


class _JsonlWriter:
    def __init__(self, d):
        self._f = open(os.path.join(d, "scalars.jsonl"), "a")

    def add_scalar(self, tag, value, step):
        self._f.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")
        self._f.flush()

    def close(self):
        self._f.close()


def run_dir(args):
    arg_dict = OrderedDict(sorted(dict(args.__dict__).items(), key=lambda t: t[0]))
    run_name = arg_dict.pop("run_name")
    seed = str(arg_dict.pop("seed"))
    logdir = str(arg_dict.pop("logdir"))
    env_name = str(arg_dict["env_name"])
    if run_name is not None:
        return os.path.join(logdir, env_name, run_name), arg_dict
    if getattr(args, "previous", None) is not None:
        if getattr(args, "exchange_reward", None) is not None:
            return args.previous[0:-1] + "_NEW-" + args.reward, arg_dict
        return args.previous[0:-1] + "-cont", arg_dict
    arg_hash = hashlib.md5(str(arg_dict).encode("ascii")).hexdigest()[0:6] + "-seed" + seed
    return os.path.join(logdir, env_name, arg_hash), arg_dict


def create_logger(args):
    assert "seed" in args.__dict__ and "logdir" in args.__dict__ and "env_name" in args.__dict__
    output_dir, arg_dict = run_dir(args)
    os.makedirs(output_dir, exist_ok=True)
    with open(os.path.join(output_dir, "experiment.pkl"), "wb") as f:
        pickle.dump(args, f)
    with open(os.path.join(output_dir, "experiment.info"), "w") as f:
        for key, val in arg_dict.items():
            f.write("%s: %s" % (key, val))
            f.write("\n")
    try:
        from torch.utils.tensorboard import SummaryWriter
        logger = SummaryWriter(output_dir, flush_secs=0.1)
    except Exception:
        logger = _JsonlWriter(output_dir)
    logger.dir = output_dir
    return logger


def parse_previous(args):
    """util/log.py:74-91: env-defining args are copied from the previous run's experiment.pkl."""
    if args.previous is not None:
        run_args = pickle.load(open(os.path.join(args.previous, "experiment.pkl"), "rb"))
        for k in ("recurrent", "env_name", "command_profile", "input_profile", "learn_gains", "traj", "no_delta", "ik_baseline"):
            setattr(args, k, getattr(run_args, k))
        if getattr(run_args, "est_lifetime", None) is not None and getattr(args, "est_lifetime", None) is None:
            args.est_lifetime = run_args.est_lifetime      # (engine flag) the continued run keeps the estimator lifetime it was trained with
        if args.exchange_reward is not None:
            args.reward = args.exchange_reward
            args.run_name = run_args.run_name + "_NEW-" + args.reward
        else:
            args.reward = run_args.reward
            args.run_name = run_args.run_name + "--cont"
    return args
