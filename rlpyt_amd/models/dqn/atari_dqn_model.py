"""AtariDqnModel: conv 32-64-64 (k 8/4/3, s 4/2/1, p 0/1/1) + MLP(512) -> Q[A]
(architecture and names of rlpyt/models/dqn/atari_dqn_model.py:10-68); input preparation as
in ``AtariFfModel`` (one fused HIP kernel, channels-last) under autograd; no-grad forwards on the
device (sampling steps, target network) run the conv stack as ``rlpyt_dqn_convs_fwd_f32``
(``Conv2dModel.features``)."""
import torch

from ...utils.tensor import infer_leading_dims, restore_leading_dims
from ..conv2d import Conv2dModel
from ..mlp import MlpModel
from .dueling import DuelingHeadModel


class AtariDqnModel(torch.nn.Module):
    def __init__(self, image_shape, output_size, fc_sizes=512, dueling=False,
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
        self.head = (DuelingHeadModel(n, fc_sizes, output_size) if dueling
                     else MlpModel(n, fc_sizes, output_size))

    def refresh_step_weights(self):
        """Entering sample / eval mode: the conv stack packs its weights once for that phase."""
        self.conv.refresh_step_weights()

    # forward() never reads prev_action / prev_reward (as rlpyt/models/dqn/atari_dqn_model.py:53-64): agents
    # then hand them over as they are instead of building a one-hot nobody reads (a fill + a scatter
    # per forward pass: six launches per DQN update)
    uses_prev_inputs = False

    def forward(self, observation, prev_action, prev_reward):
        lead_dim, T, B, img_shape = infer_leading_dims(observation, 3)
        q = self.head(self.conv.features(observation, T * B, img_shape))
        return restore_leading_dims(q, lead_dim, T, B)
