"""FF_V with the reference's pickle surface (rl/policies/critic.py:37-77): critic_layers, network_out, obs_std,
obs_mean, normc_init; normalises its input only when NOT in training mode (critic.py:66-67)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from rl.policies.base import Net, normc_fn


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
        self.critic_layers = nn.ModuleList()
        self.critic_layers += [nn.Linear(state_dim, layers[0])]
        for i in range(len(layers) - 1):
            self.critic_layers += [nn.Linear(layers[i], layers[i + 1])]
        self.network_out = nn.Linear(layers[-1], 1)
        self.env_name = env_name
        self.nonlinearity = nonlinearity
        self.obs_std = obs_std
        self.obs_mean = obs_mean
        self.normc_init = normc_init
        self.init_parameters()
        self.train()

    def init_parameters(self):
        if self.normc_init:
            self.apply(normc_fn)

    def forward(self, inputs):
        if self.training is False:
            inputs = (inputs - self.obs_mean) / self.obs_std
        x = inputs
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
        self.critic_layers = nn.ModuleList()
        self.critic_layers += [nn.LSTMCell(input_dim, layers[0])]
        for i in range(len(layers) - 1):
            self.critic_layers += [nn.LSTMCell(layers[i], layers[i + 1])]
        self.network_out = nn.Linear(layers[-1], 1)
        self.init_hidden_state()
        self.is_recurrent = True
        self.env_name = env_name
        self.obs_std = 1.0
        self.obs_mean = 0.0
        if normc_init:
            self.initialize_parameters()

    def get_hidden_state(self):
        return self.hidden, self.cells

    def init_hidden_state(self, batch_size=1):
        self.hidden = [torch.zeros(batch_size, l.hidden_size) for l in self.critic_layers]
        self.cells = [torch.zeros(batch_size, l.hidden_size) for l in self.critic_layers]

    def _step(self, x):
        for idx, layer in enumerate(self.critic_layers):
            self.hidden[idx], self.cells[idx] = layer(x, (self.hidden[idx], self.cells[idx]))
            x = self.hidden[idx]
        return self.network_out(x)

    def forward(self, state):
        if self.training is False:
            state = (state - self.obs_mean) / self.obs_std
        if state.dim() == 3:
            self.init_hidden_state(batch_size=state.size(1))
            return torch.stack([self._step(s_t) for s_t in state])
        flat = state.dim() == 1
        x = self._step(state.view(1, -1) if flat else state)
        return x.view(-1) if flat else x


class Dual_Q_Critic(Critic):
    """Twin Q network of TD3 with the reference's pickle surface (rl/policies/critic.py:118-168): q1_layers / q1_out, q2_layers / q2_out."""

    def __init__(self, state_dim, action_dim, hidden_size=256, hidden_layers=2, env_name="NOT SET"):
        super().__init__()
        self.q1_layers = nn.ModuleList([nn.Linear(state_dim + action_dim, hidden_size)] + [nn.Linear(hidden_size, hidden_size) for _ in range(hidden_layers - 1)])
        self.q1_out = nn.Linear(hidden_size, 1)
        self.q2_layers = nn.ModuleList([nn.Linear(state_dim + action_dim, hidden_size)] + [nn.Linear(hidden_size, hidden_size) for _ in range(hidden_layers - 1)])
        self.q2_out = nn.Linear(hidden_size, 1)
        self.env_name = env_name

    def Q1(self, state, action):
        x = torch.cat([state, action], state.dim() - 1)
        for layer in self.q1_layers:
            x = F.relu(layer(x))
        return self.q1_out(x)

    def forward(self, state, action):
        x2 = torch.cat([state, action], state.dim() - 1)
        for layer in self.q2_layers:
            x2 = F.relu(layer(x2))
        return self.Q1(state, action), self.q2_out(x2)
