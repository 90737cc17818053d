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
