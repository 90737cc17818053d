"""Dueling Q head (rlpyt/models/dqn/dueling.py:8-44): Q = V + (A - mean A), shared bias
on the advantage stream, gradient scaled by 2^-1/2 going back into the conv trunk."""
import torch

from ..mlp import MlpModel
from ..utils import scale_grad


class DuelingHeadModel(torch.nn.Module):
    def __init__(self, input_size, hidden_sizes, output_size, grad_scale=2 ** (-1 / 2)):
        super().__init__()
        if isinstance(hidden_sizes, int):
            hidden_sizes = [hidden_sizes]
        self.advantage_hidden = MlpModel(input_size, hidden_sizes)
        self.advantage_out = torch.nn.Linear(hidden_sizes[-1], output_size, bias=False)
        self.advantage_bias = torch.nn.Parameter(torch.zeros(1))
        self.value = MlpModel(input_size, hidden_sizes, output_size=1)
        self._grad_scale = grad_scale

    def advantage(self, input):
        return self.advantage_out(self.advantage_hidden(input)) + self.advantage_bias

    def forward(self, input):
        x = scale_grad(input, self._grad_scale)
        adv = self.advantage(x)
        return self.value(x) + (adv - adv.mean(dim=-1, keepdim=True))


class DistributionalDuelingHeadModel(torch.nn.Module):
    """Dueling head with ``n_atoms`` outputs per action (dueling.py:48-84): value stream
    [B,1,P] + mean-centred advantage stream [B,A,P], one bias per atom."""

    def __init__(self, input_size, hidden_sizes, output_size, n_atoms,
                 grad_scale=2 ** (-1 / 2)):
        super().__init__()
        if isinstance(hidden_sizes, int):
            hidden_sizes = [hidden_sizes]
        self.advantage_hidden = MlpModel(input_size, hidden_sizes)
        self.advantage_out = torch.nn.Linear(hidden_sizes[-1], output_size * n_atoms,
                                             bias=False)
        self.advantage_bias = torch.nn.Parameter(torch.zeros(n_atoms))
        self.value = MlpModel(input_size, hidden_sizes, output_size=n_atoms)
        self._grad_scale = grad_scale
        self._output_size = output_size
        self._n_atoms = n_atoms

    def advantage(self, input):
        x = self.advantage_out(self.advantage_hidden(input))
        return x.view(-1, self._output_size, self._n_atoms) + self.advantage_bias

    def forward(self, input):
        x = scale_grad(input, self._grad_scale)
        adv = self.advantage(x)
        value = self.value(x).view(-1, 1, self._n_atoms)
        return value + (adv - adv.mean(dim=1, keepdim=True))
