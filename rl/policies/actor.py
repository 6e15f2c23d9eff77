"""Host-side mirrors of the reference's actor classes, kept ONLY as the pickle surface: module path, class
names, attribute names, state_dict keys and forward signatures are dictated by the reference's checkpoints (rl/policies/actor.py:43-72,
142-215, 218-311), because torch.save(policy) pickles the class by name and the reference's eval scripts call policy(state,
deterministic=True).  The constructor bodies therefore follow the reference by necessity; everything the HIP engine does not train
(state-dependent log-std heads, tanh-bounded means) is absent: a reference checkpoint that uses them loads, and is refused at its first
forward.  On the GPU the weights live in apex_amd.engine.Mlp / Lstm; these classes only carry them to and from disk and run a saved
policy with plain torch."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from rl.policies.base import Net, normc_fn


def _stack(kind, sizes):
    """nn.ModuleList of `kind` layers over consecutive sizes (state_dict keys actor_layers.<i>.*)"""
    return nn.ModuleList([kind(a, b) for a, b in zip(sizes[:-1], sizes[1:])])


class Actor(Net):
    def forward(self):
        raise NotImplementedError


class _FixedStdGaussian(Actor):
    """forward / get_action / distribution shared by the two Gaussian actors: N(mean(state), fixed_std * anneal)"""

    def _std(self):
        if getattr(self, "learn_std", False) or getattr(self, "bounded", False):
            raise NotImplementedError("checkpoints with a learned log-std head or tanh-bounded means are not supported by this engine")
        return self.fixed_std

    def forward(self, state, deterministic=True, anneal=1.0):
        if getattr(self, "bounded", False):      # a reference checkpoint with tanh-bounded means: refuse at the first forward, deterministic or not
            raise NotImplementedError("checkpoints with tanh-bounded means are not supported by this engine")
        mu = self._mean(state)
        self.action = mu if deterministic else torch.distributions.Normal(mu, self._std() * anneal).sample()
        return self.action

    def get_action(self):
        return self.action

    def distribution(self, inputs):
        return torch.distributions.Normal(self._mean(inputs), self._std())


class Gaussian_FF_Actor(_FixedStdGaussian):
    def __init__(self, state_dim, action_dim, layers=(256, 256), env_name=None, nonlinearity=F.relu, fixed_std=None,
                 bounded=False, normc_init=True):
        super().__init__()
        assert fixed_std is not None and not bounded, "the engine trains fixed-std, unbounded Gaussian policies (apex.py --std_dev)"
        self.actor_layers = _stack(nn.Linear, (state_dim,) + tuple(layers))
        self.means = nn.Linear(layers[-1], action_dim)
        self.fixed_std, self.learn_std, self.bounded = fixed_std, False, False
        self.action, self.action_dim, self.env_name, self.nonlinearity = None, action_dim, env_name, nonlinearity
        self.obs_std, self.obs_mean, self.normc_init = 1.0, 0.0, normc_init
        if normc_init:
            self.apply(normc_fn)
            self.means.weight.data.mul_(0.01)

    def _mean(self, state):
        x = (state - self.obs_mean) / self.obs_std
        for layer in self.actor_layers:
            x = self.nonlinearity(layer(x))
        return self.means(x)


class Gaussian_LSTM_Actor(_FixedStdGaussian):
    """actor_layers = stacked nn.LSTMCell, network_out, hidden / cells lists, init_hidden_state(batch_size); a [T, B, D] input is a padded
    batch of trajectories run from zero state, a [D] / [B, D] input is one step with the carried state (rl/policies/actor.py:253-289)."""

    def __init__(self, state_dim, action_dim, layers=(128, 128), env_name=None, nonlinearity=torch.tanh, normc_init=False, max_action=1,
                 fixed_std=None):
        super().__init__()
        assert fixed_std is not None, "the engine trains fixed-std Gaussian policies (apex.py --std_dev)"
        self.actor_layers = _stack(nn.LSTMCell, (state_dim,) + tuple(layers))
        self.network_out = nn.Linear(layers[-1], action_dim)
        self.fixed_std, self.learn_std = fixed_std, False
        self.action, self.action_dim, self.env_name, self.nonlinearity, self.max_action = None, action_dim, env_name, nonlinearity, max_action
        self.obs_std, self.obs_mean, self.is_recurrent = 1.0, 0.0, True
        self.init_hidden_state()
        if normc_init:
            self.initialize_parameters()

    def init_hidden_state(self, batch_size=1):
        self.hidden = [torch.zeros(batch_size, l.hidden_size) for l in self.actor_layers]
        self.cells = [torch.zeros(batch_size, l.hidden_size) for l in self.actor_layers]

    def _step(self, x):
        for i, cell in enumerate(self.actor_layers):
            self.hidden[i], self.cells[i] = cell(x, (self.hidden[i], self.cells[i]))
            x = self.hidden[i]
        return x

    def _mean(self, state):
        x = (state - self.obs_mean) / self.obs_std
        if x.dim() == 3:
            self.init_hidden_state(batch_size=x.size(1))
            x = torch.stack([self._step(x_t) for x_t in x])
        else:
            x = self._step(x.view(1, -1)).view(-1) if x.dim() == 1 else self._step(x)
        return self.network_out(x)


class FF_Actor(Actor):
    """Deterministic tanh actor of TD3 / DDPG (rl/policies/actor.py:43-72)."""

    def __init__(self, state_dim, action_dim, layers=(256, 256), env_name=None, nonlinearity=F.relu, max_action=1):
        super().__init__()
        self.actor_layers = _stack(nn.Linear, (state_dim,) + tuple(layers))
        self.network_out = nn.Linear(layers[-1], action_dim)
        self.action, self.action_dim, self.env_name, self.nonlinearity, self.max_action = None, action_dim, env_name, nonlinearity, max_action
        self.initialize_parameters()

    def forward(self, state, deterministic=True):
        x = state
        for layer in self.actor_layers:
            x = self.nonlinearity(layer(x))
        self.action = torch.tanh(self.network_out(x))
        return self.action * self.max_action

    def get_action(self):
        return self.action
