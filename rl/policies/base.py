"""Checkpoint-compatible base class (attribute names as pickled by the reference, rl/policies/base.py:16-26)."""
import torch
import torch.nn as nn


def normc_fn(m):
    """PPO-paper 'normc' initialisation (reference rl/policies/base.py:7-13): unit-norm rows, zero bias."""
    if m.__class__.__name__.find("Linear") != -1:
        m.weight.data.normal_(0, 1)
        m.weight.data *= 1 / torch.sqrt(m.weight.data.pow(2).sum(1, keepdim=True))
        if m.bias is not None:
            m.bias.data.fill_(0)


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.is_recurrent = False
        self.welford_state_mean = torch.zeros(1)
        self.welford_state_mean_diff = torch.ones(1)
        self.welford_state_n = 1
        self.env_name = None

    def initialize_parameters(self):
        self.apply(normc_fn)
