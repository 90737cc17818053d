"""Small MLP policy/value model for vector observations (config #1 plumbing runs)."""
import torch
import torch.nn.functional as F

from ...utils.tensor import infer_leading_dims, restore_leading_dims
from ..mlp import MlpModel


class MlpPgModel(torch.nn.Module):
    def __init__(self, observation_shape, output_size, hidden_sizes=(64, 64)):
        super().__init__()
        self._obs_ndim = len(observation_shape)
        n_in = 1
        for s in observation_shape:
            n_in *= s
        self.body = MlpModel(n_in, list(hidden_sizes), nonlinearity=torch.nn.Tanh)
        self.pi = torch.nn.Linear(self.body.output_size, output_size)
        self.value = torch.nn.Linear(self.body.output_size, 1)

    def forward(self, observation, prev_action, prev_reward):
        lead_dim, T, B, _ = infer_leading_dims(observation, self._obs_ndim)
        x = self.body(observation.reshape(T * B, -1).float())
        pi = F.softmax(self.pi(x), dim=-1)
        v = self.value(x).squeeze(-1)
        return restore_leading_dims((pi, v), lead_dim, T, B)
