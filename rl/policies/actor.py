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


class Gaussian_LSTM_Actor(Actor):
    """Recurrent actor with the reference's pickle surface (rl/policies/actor.py:218-311): actor_layers = stacked nn.LSTMCell,
    network_out, fixed_std, obs_mean / obs_std, is_recurrent, hidden / cells lists, init_hidden_state(batch_size); a [T, B, D] input is a
    padded batch of trajectories run from zero state, a [D] / [B, D] input is one step with the carried state."""

    def __init__(self, state_dim, action_dim, layers=(128, 128), env_name=None, nonlinearity=torch.tanh, normc_init=False, max_action=1,
                 fixed_std=None):
        super().__init__()
        self.actor_layers = nn.ModuleList()
        self.actor_layers += [nn.LSTMCell(state_dim, layers[0])]
        for i in range(len(layers) - 1):
            self.actor_layers += [nn.LSTMCell(layers[i], layers[i + 1])]
        self.network_out = nn.Linear(layers[-1], action_dim)
        self.action = None
        self.action_dim = action_dim
        self.init_hidden_state()
        self.env_name = env_name
        self.nonlinearity = nonlinearity
        self.max_action = max_action
        self.obs_std = 1.0
        self.obs_mean = 0.0
        self.is_recurrent = True
        if fixed_std is None:
            self.log_stds = nn.Linear(layers[-1], action_dim)
            self.learn_std = True
        else:
            self.fixed_std = fixed_std
            self.learn_std = False
        if normc_init:
            self.initialize_parameters()

    def init_hidden_state(self, batch_size=1):
        self.hidden = [torch.zeros(batch_size, l.hidden_size) for l in self.actor_layers]
        self.cells = [torch.zeros(batch_size, l.hidden_size) for l in self.actor_layers]

    def _step(self, x):
        for idx, layer in enumerate(self.actor_layers):
            self.hidden[idx], self.cells[idx] = layer(x, (self.hidden[idx], self.cells[idx]))
            x = self.hidden[idx]
        return x

    def _get_dist_params(self, state):
        x = (state - self.obs_mean) / self.obs_std
        if x.dim() == 3:
            self.init_hidden_state(batch_size=x.size(1))
            x = torch.stack([self._step(x_t) for x_t in x])
        else:
            flat = x.dim() == 1
            x = self._step(x.view(1, -1) if flat else x)
            if flat:
                x = x.view(-1)
        mu = self.network_out(x)
        sd = (-2 + 0.5 * torch.tanh(self.log_stds(x))).exp() if self.learn_std else self.fixed_std
        return mu, sd

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


class FF_Actor(Actor):
    """Deterministic tanh actor of TD3 / DDPG with the reference's pickle surface (rl/policies/actor.py:43-72)."""

    def __init__(self, state_dim, action_dim, layers=(256, 256), env_name=None, nonlinearity=F.relu, max_action=1):
        super().__init__()
        self.actor_layers = nn.ModuleList()
        self.actor_layers += [nn.Linear(state_dim, layers[0])]
        for i in range(len(layers) - 1):
            self.actor_layers += [nn.Linear(layers[i], layers[i + 1])]
        self.network_out = nn.Linear(layers[-1], action_dim)
        self.action = None
        self.action_dim = action_dim
        self.env_name = env_name
        self.nonlinearity = nonlinearity
        self.initialize_parameters()
        self.max_action = max_action

    def forward(self, state, deterministic=True):
        x = state
        for layer in self.actor_layers:
            x = self.nonlinearity(layer(x))
        self.action = torch.tanh(self.network_out(x))
        return self.action * self.max_action

    def get_action(self):
        return self.action
