"""Golden G21 (row a12): the reference's get_normalization_params (rl/envs/normalize.py:11-48) run in-process under the ray stand-in on
deterministic toy envs with the Cassie-v0 surface: `procs` workers of iter // procs steps each, action = policy(state) + N(0, noise_std)
with the un-normalised initial policy (ppo.py:546-547), the state BEFORE every step recorded, reset (and the terminal state dropped) on
done, mean and sqrt(var + 1e-8) over all recorded states.  Recorded: the policy, the noise each worker drew, the toy envs' schedules,
the returned (mean, std)."""
from common import setup_reference_path, GOLD
setup_reference_path()

import os
import numpy as np
import torch

from rl.envs.normalize import get_normalization_params
from rl.policies.actor import Gaussian_FF_Actor

LENS = [7, 40, 23, 55, 3, 61, 12, 9]


class ToyEnv:
    """x <- 0.9 x + 0.1 tile(a, 5) + 0.01; worker w starts its episode counter at 100 w (every worker builds its own env)."""
    built = 0

    def __init__(self):
        self.observation_space = np.zeros(50); self.action_space = np.zeros(10); self.simrate = 50
        self.k = 100 * ToyEnv.built; ToyEnv.built += 1

    def reset(self):
        self.k += 1; self.t = 0; self.L = LENS[(self.k - 1) % len(LENS)]
        self.x = np.cos(np.arange(50) * 0.1 * self.k)
        return self.x.copy()

    def step(self, action, f_term=0):
        self.t += 1
        self.x = 0.9 * self.x + 0.1 * np.tile(action, 5) + 0.01
        return self.x.copy(), 0.0, self.t >= self.L, {}


def main():
    torch.manual_seed(21)
    H, procs, iters, noise_std = 32, 4, 240, 1.0
    policy = Gaussian_FF_Actor(50, 10, layers=(H, H), fixed_std=np.exp(-1.5))
    out = {"hidden": H, "procs": procs, "iters": iters, "noise_std": noise_std, "lens": np.array(LENS)}
    for k, v in policy.state_dict().items(): out["actor." + k] = v.numpy().copy()
    # the only RNG consumer is the action noise, drawn worker by worker (the stand-in runs the workers one after the other)
    torch.manual_seed(2121)
    out["noise"] = torch.stack([torch.stack([torch.randn(1, 10) for _ in range(iters // procs)]) for _ in range(procs)]).numpy()[:, :, 0]   # [procs, steps, 10]
    torch.manual_seed(2121)
    with torch.no_grad():
        mean, std = get_normalization_params(iter=iters, noise_std=noise_std, policy=policy, env_fn=lambda: ToyEnv(), procs=procs)
    out["mean"] = mean; out["std"] = std
    np.savez_compressed(os.path.join(GOLD, "g21_normalization_params.npz"), **out)
    print("mean[:4]", mean[:4], "std[:4]", std[:4], "envs built", ToyEnv.built)


if __name__ == "__main__":
    main()
