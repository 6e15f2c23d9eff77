"""Gaussian_FF_Actor with the reference's pickle surface (rl/policies/actor.py:142-215): attributes actor_layers,
means, fixed_std, learn_std, action, action_dim, env_name, nonlinearity, obs_std, obs_mean, normc_init, bounded and
forward(state, deterministic=True, anneal=1.0).  On the GPU the weights live in apex_amd.engine.Mlp; this module is
the host-side mirror used for checkpoints and for running a saved policy with plain torch."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from rl.policies.base import Net, normc_fn


class Actor(Net):
    def __init__(self):
        super().__init__()

    def forward(self):
        raise NotImplementedError


class Gaussian_FF_Actor(Actor):
    def __init__(self, state_dim, action_dim, layers=(256, 256), env_name=None, nonlinearity=F.relu, fixed_std=None,
                 bounded=False, normc_init=True):
        super().__init__()
        self.actor_layers = nn.ModuleList()
        self.actor_layers += [nn.Linear(state_dim, layers[0])]
        for i in range(len(layers) - 1):
            self.actor_layers += [nn.Linear(layers[i], layers[i + 1])]
        self.means = nn.Linear(layers[-1], action_dim)
        if fixed_std is None:
            self.log_stds = nn.Linear(layers[-1], action_dim)
            self.learn_std = True
        else:
            self.fixed_std = fixed_std
            self.learn_std = False
        self.action = None
        self.action_dim = action_dim
        self.env_name = env_name
        self.nonlinearity = nonlinearity
        self.obs_std = 1.0
        self.obs_mean = 0.0
        self.normc_init = normc_init
        self.bounded = bounded
        self.init_parameters()

    def init_parameters(self):
        if self.normc_init:
            self.apply(normc_fn)
            self.means.weight.data.mul_(0.01)

    def _get_dist_params(self, state):
        x = (state - self.obs_mean) / self.obs_std
        for layer in self.actor_layers:
            x = self.nonlinearity(layer(x))
        mean = self.means(x)
        if self.bounded:
            mean = torch.tanh(mean)
        if self.learn_std:
            sd = (-2 + 0.5 * torch.tanh(self.log_stds(x))).exp()
        else:
            sd = self.fixed_std
        return mean, sd

    def forward(self, state, deterministic=True, anneal=1.0):
        mu, sd = self._get_dist_params(state)
        sd = sd * anneal
        self.action = mu if deterministic else torch.distributions.Normal(mu, sd).sample()
        return self.action

    def get_action(self):
        return self.action

    def distribution(self, inputs):
        mu, sd = self._get_dist_params(inputs)
        return torch.distributions.Normal(mu, sd)
