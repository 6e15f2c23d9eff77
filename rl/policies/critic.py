"""Host-side mirrors of the reference's critic classes, kept ONLY as the pickle surface: module path, class and
attribute names, state_dict keys and forward signatures are dictated by the reference's checkpoints (rl/policies/critic.py:37-77 FF_V,
118-168 Dual_Q_Critic, 236-296 LSTM_V), so the constructor bodies follow the reference by necessity.  The value nets normalise their input
only when NOT in training mode (critic.py:66-67).  On the GPU the weights live in apex_amd.engine; these classes carry them to and from disk."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from rl.policies.base import Net, normc_fn
from rl.policies.actor import _stack


class Critic(Net):
    def __init__(self):
        super().__init__()
        self.welford_reward_mean = 0.0
        self.welford_reward_mean_diff = 1.0
        self.welford_reward_n = 1

    def forward(self):
        raise NotImplementedError


class FF_V(Critic):
    def __init__(self, state_dim, layers=(256, 256), env_name="NOT SET", nonlinearity=F.relu, normc_init=True,
                 obs_std=None, obs_mean=None):
        super().__init__()
        self.critic_layers = _stack(nn.Linear, (state_dim,) + tuple(layers))
        self.network_out = nn.Linear(layers[-1], 1)
        self.env_name, self.nonlinearity, self.obs_std, self.obs_mean, self.normc_init = env_name, nonlinearity, obs_std, obs_mean, normc_init
        if normc_init:
            self.apply(normc_fn)
        self.train()

    def forward(self, inputs):
        x = inputs if self.training else (inputs - self.obs_mean) / self.obs_std
        for layer in self.critic_layers:
            x = self.nonlinearity(layer(x))
        return self.network_out(x)

    def act(self, inputs):
        return self(inputs)


class LSTM_V(Critic):
    """Recurrent critic with the reference's pickle surface (rl/policies/critic.py:236-296): critic_layers = stacked nn.LSTMCell,
    network_out; raw inputs in training mode."""

    def __init__(self, input_dim, layers=(128, 128), env_name="NOT SET", normc_init=True):
        super().__init__()
        self.critic_layers = _stack(nn.LSTMCell, (input_dim,) + tuple(layers))
        self.network_out = nn.Linear(layers[-1], 1)
        self.init_hidden_state()
        self.is_recurrent, self.env_name, self.obs_std, self.obs_mean = True, env_name, 1.0, 0.0
        if normc_init:
            self.initialize_parameters()

    def get_hidden_state(self):
        return self.hidden, self.cells

    def init_hidden_state(self, batch_size=1):
        self.hidden = [torch.zeros(batch_size, l.hidden_size) for l in self.critic_layers]
        self.cells = [torch.zeros(batch_size, l.hidden_size) for l in self.critic_layers]

    def _step(self, x):
        for i, cell in enumerate(self.critic_layers):
            self.hidden[i], self.cells[i] = cell(x, (self.hidden[i], self.cells[i]))
            x = self.hidden[i]
        return self.network_out(x)

    def forward(self, state):
        x = state if self.training else (state - self.obs_mean) / self.obs_std
        if x.dim() == 3:
            self.init_hidden_state(batch_size=x.size(1))
            return torch.stack([self._step(x_t) for x_t in x])
        return self._step(x.view(1, -1)).view(-1) if x.dim() == 1 else self._step(x)


class Dual_Q_Critic(Critic):
    """Twin Q network of TD3 with the reference's pickle surface (rl/policies/critic.py:118-168): q1_layers / q1_out, q2_layers / q2_out."""

    def __init__(self, state_dim, action_dim, hidden_size=256, hidden_layers=2, env_name="NOT SET"):
        super().__init__()
        sizes = (state_dim + action_dim,) + (hidden_size,) * hidden_layers
        self.q1_layers, self.q1_out = _stack(nn.Linear, sizes), nn.Linear(hidden_size, 1)
        self.q2_layers, self.q2_out = _stack(nn.Linear, sizes), nn.Linear(hidden_size, 1)
        self.env_name = env_name

    @staticmethod
    def _q(layers, out, sa):
        for layer in layers:
            sa = F.relu(layer(sa))
        return out(sa)

    def Q1(self, state, action):
        return self._q(self.q1_layers, self.q1_out, torch.cat([state, action], state.dim() - 1))

    def forward(self, state, action):
        sa = torch.cat([state, action], state.dim() - 1)
        return self._q(self.q1_layers, self.q1_out, sa), self._q(self.q2_layers, self.q2_out, sa)
