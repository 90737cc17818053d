"""AtariCatDqnModel: the AtariDqnModel trunk with ``n_atoms`` softmax outputs per action
(architecture and parameter names of rlpyt/models/dqn/atari_catdqn_model.py:10-81, so state
dicts interchange with the reference)."""
import torch
import torch.nn.functional as F

from ...utils.tensor import infer_leading_dims, restore_leading_dims
from ..conv2d import Conv2dModel
from ..mlp import MlpModel
from .dueling import DistributionalDuelingHeadModel


class DistributionalHeadModel(torch.nn.Module):
    """MLP head reshaped to [B, output_size, n_atoms] (atari_catdqn_model.py:10-21)."""

    def __init__(self, input_size, layer_sizes, output_size, n_atoms):
        super().__init__()
        self.mlp = MlpModel(input_size, layer_sizes, output_size * n_atoms)
        self._output_size = output_size
        self._n_atoms = n_atoms

    def forward(self, input):
        return self.mlp(input).view(-1, self._output_size, self._n_atoms)


class AtariCatDqnModel(torch.nn.Module):
    def __init__(self, image_shape, output_size, n_atoms=51, fc_sizes=512, dueling=False,
                 use_maxpool=False, channels=None, kernel_sizes=None, strides=None,
                 paddings=None):
        super().__init__()
        self.dueling = dueling
        c, h, w = image_shape
        self.conv = Conv2dModel(in_channels=c, channels=channels or [32, 64, 64],
                                kernel_sizes=kernel_sizes or [8, 4, 3],
                                strides=strides or [4, 2, 1], paddings=paddings or [0, 1, 1],
                                use_maxpool=use_maxpool)
        n = self.conv.conv_out_size(h, w)
        Head = DistributionalDuelingHeadModel if dueling else DistributionalHeadModel
        self.head = Head(n, fc_sizes, output_size=output_size, n_atoms=n_atoms)

    def refresh_step_weights(self):
        """Entering sample / eval mode: the conv stack packs its weights once for that phase."""
        self.conv.refresh_step_weights()

    uses_prev_inputs = False          # (see AtariDqnModel)

    def forward(self, observation, prev_action, prev_reward):
        """Probability masses [.., A, n_atoms] (softmax over atoms)."""
        lead_dim, T, B, img_shape = infer_leading_dims(observation, 3)
        p = F.softmax(self.head(self.conv.features(observation, T * B, img_shape)), dim=-1)
        return restore_leading_dims(p, lead_dim, T, B)
