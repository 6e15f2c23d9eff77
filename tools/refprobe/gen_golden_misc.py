"""Golden vectors G13 (run-directory layout / experiment.info text) produced by the reference's own util/log.py
create_logger under a stand-in SummaryWriter (tensorboard is not installed here)."""
from common import setup_reference_path, GOLD
setup_reference_path()

import argparse
import json
import os
import sys
import tempfile
import types

fake = types.ModuleType("torch.utils.tensorboard")
class SummaryWriter:                      # noqa: E302
    def __init__(self, d, flush_secs=0): self.d = d
fake.SummaryWriter = SummaryWriter
sys.modules["torch.utils.tensorboard"] = fake

from util.log import create_logger  # noqa: E402


def main():
    tmp = tempfile.mkdtemp()
    cases = []
    for seed, run_name in ((0, None), (7, None), (3, "myrun")):
        ns = argparse.Namespace(
            command_profile="clock", input_profile="full", simrate=50, dyn_random=True, learn_gains=False, traj="walking",
            no_delta=True, ik_baseline=False, mirror=True, reward="clock", env_name="Cassie-v0", run_name=run_name,
            exchange_reward=None, previous=None, logdir=tmp + "/", seed=seed, history=0, redis_address=None, viz_port=8097,
            input_norm_steps=10000, n_itr=10000, lr=1e-4, eps=1e-5, lam=0.95, gamma=0.99, anneal=1.0, learn_stddev=False,
            std_dev=-1.5, entropy_coeff=0.0, clip=0.2, minibatch_size=64, epochs=3, num_steps=5096, use_gae=True,
            num_procs=30, max_grad_norm=0.05, max_traj_len=400, recurrent=False, bounded=False)
        logger = create_logger(ns)
        rel = os.path.relpath(logger.dir, tmp)
        info = open(os.path.join(logger.dir, "experiment.info")).read().replace(tmp, "<LOGDIR>")
        cases.append(dict(args={k: v for k, v in vars(ns).items() if k != "logdir"}, rel_dir=rel, info=info,
                          files=sorted(os.listdir(logger.dir))))
    json.dump(cases, open(os.path.join(GOLD, "g13_logdir.json"), "w"), indent=1)
    print("wrote g13")


if __name__ == "__main__":
    main()
