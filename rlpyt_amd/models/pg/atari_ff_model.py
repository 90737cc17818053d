"""AtariFfModel: conv(4->16,k8,s4) ReLU conv(16->32,k4,s2,p1) ReLU FC 512 ReLU -> {pi: FC A +
softmax, value: FC 1} (architecture and parameter names of
rlpyt/models/pg/atari_ff_model.py:9-63; 1 785 911 parameters at A=6).

Input handling on the MI355X: uint8 observations are converted by ONE HIP kernel
(``rlpyt_obs_to_nhwc_f32``: uint8 -> float32 * 1/255, CHW -> HWC) into channels-last
storage, which is the layout MIOpen's fp32 implicit-GEMM convolutions run in natively --
this replaces the reference's ``.type(float)`` + ``.mul_(1/255)`` (atari_ff_model.py:50-51)
and the NCHW<->NHWC transposes around every convolution.  A float input is taken as already
prepared (the PPO minibatch path fuses the gather into the same kernel).  On CPU tensors the
plain torch ops run (used by host-logic tests and example generation only).
"""
import torch
import torch.nn.functional as F

from ...utils.tensor import infer_leading_dims, restore_leading_dims
from ..conv2d import Conv2dHeadModel


def prepare_image(image, T_B, img_shape):
    """uint8 [..., C, H, W] -> float32 [T*B, C, H, W] scaled to [0,1] (channels-last on GPU)."""
    if image.dtype != torch.uint8:
        return image.reshape(T_B, *img_shape)
    if image.is_cuda:
        from ... import ops
        return ops.obs_to_nhwc_f32(image.contiguous().reshape(T_B, *img_shape))
    return image.reshape(T_B, *img_shape).float().mul_(1. / 255)


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
        """[T,B,C,H,W] / [B,C,H,W] / [C,H,W] -> (pi, value) with the same lead dims."""
        lead_dim, T, B, img_shape = infer_leading_dims(image, 3)
        img = prepare_image(image, T * B, img_shape)
        fc_out = self.conv(img)
        pi = F.softmax(self.pi(fc_out), dim=-1)
        v = self.value(fc_out).squeeze(-1)
        return restore_leading_dims((pi, v), lead_dim, T, B)
