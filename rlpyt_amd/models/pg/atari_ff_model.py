"""AtariFfModel: conv(4->16,k8,s4) ReLU conv(16->32,k4,s2,p1) ReLU FC 512 ReLU -> {pi: FC A +
softmax, value: FC 1} (architecture and parameter names of
rlpyt/models/pg/atari_ff_model.py:9-63; 1 785 911 parameters at A=6).

uint8 observations are scaled by 1/255 inside forward, as the reference does.  The
contraction itself runs on the MI355X through PyTorch-ROCm (MIOpen/hipBLASLt fp32);
everything around it on the training path is the fused HIP kernels of this package.
"""
import torch
import torch.nn.functional as F

from ...utils.tensor import infer_leading_dims, restore_leading_dims
from ..conv2d import Conv2dHeadModel


class AtariFfModel(torch.nn.Module):
    def __init__(self, image_shape, output_size, fc_sizes=512, use_maxpool=False,
                 channels=None, kernel_sizes=None, strides=None, paddings=None):
        super().__init__()
        self.conv = Conv2dHeadModel(
            image_shape=image_shape, channels=channels or [16, 32],
            kernel_sizes=kernel_sizes or [8, 4], strides=strides or [4, 2],
            paddings=paddings or [0, 1], use_maxpool=use_maxpool, hidden_sizes=fc_sizes)
        self.pi = torch.nn.Linear(self.conv.output_size, output_size)
        self.value = torch.nn.Linear(self.conv.output_size, 1)

    def forward(self, image, prev_action, prev_reward):
        """[T,B,C,H,W] / [B,C,H,W] / [C,H,W] uint8 -> (pi, value) with the same lead dims."""
        lead_dim, T, B, img_shape = infer_leading_dims(image, 3)
        img = image.reshape(T * B, *img_shape).float().mul_(1. / 255)
        fc_out = self.conv(img)
        pi = F.softmax(self.pi(fc_out), dim=-1)
        v = self.value(fc_out).squeeze(-1)
        return restore_leading_dims((pi, v), lead_dim, T, B)
