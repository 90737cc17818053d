"""AtariFfModel: conv(4->16,k8,s4) ReLU conv(16->32,k4,s2,p1) ReLU FC 512 ReLU -> {pi: FC A +
softmax, value: FC 1} (architecture and parameter names of
rlpyt/models/pg/atari_ff_model.py:9-63; 1 785 911 parameters at A=6).

Input handling on the MI355X: with the default geometry (uint8 [4,104,80], conv 16-32,
kernels 8-4, strides 4-2, paddings 0-1) and the parameters on the device, the whole
convolution stack -- minibatch gather, uint8 -> float32 * 1/255, both convolutions, biases and
ReLUs, forward and backward -- runs in the hand-written fp32-MFMA kernels of csrc/conv.hip
(``ops.atari_conv_stack``); the f32 image is never materialised and the 3456 features come out
in the NCHW-flatten order the reference's ``nn.Linear`` weight expects, so state dicts
interchange with the reference.  Other geometries fall back to MIOpen: uint8 observations are
then converted by ``rlpyt_obs_to_nhwc_f32`` into channels-last storage.  On CPU tensors the
plain torch ops run (used by host-logic tests and example generation only).
"""
import os
from collections import namedtuple

import torch
import torch.nn.functional as F

from ...utils.tensor import infer_leading_dims, restore_leading_dims
from ..conv2d import Conv2dHeadModel


# A minibatch of observations named by index instead of by value: rows ``idx -> (idx % T,
# idx // T)`` of a [T,B,C,H,W] uint8 batch (rlpyt/algos/pg/ppo.py:94-100).  The fused conv
# kernels resolve the indices while staging each image, so the gathered copy never exists.
ObsGather = namedtuple("ObsGather", ["observation", "flat_idx"])


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
        self._default_geometry = (
            tuple(image_shape) == (4, 104, 80) and not use_maxpool
            and list(channels or [16, 32]) == [16, 32] and list(kernel_sizes or [8, 4]) == [8, 4]
            and list(strides or [4, 2]) == [4, 2] and list(paddings or [0, 1]) == [0, 1])
        self.use_fused_conv = True   # set False to force the MIOpen path (A/B tests)

    @property
    def fused_conv(self):
        """True when the hand-written MFMA conv stack applies (geometry + device)."""
        w = self.conv.conv.conv[0].weight
        return (self.use_fused_conv and self._default_geometry and w.is_cuda
                and w.dtype == torch.float32)

    def _single_fc(self):
        """The trunk's Linear when the head is exactly Linear -> ReLU (the default), else None."""
        mods = list(getattr(self.conv.head, "model", torch.nn.Sequential()).children())
        if (len(mods) == 2 and isinstance(mods[0], torch.nn.Linear)
                and isinstance(mods[1], torch.nn.ReLU) and mods[0].out_features % 16 == 0):
            return mods[0]
        return None

    # set False (or RLPYT_SPLIT_GEMM=0) for the library f32 GEMM in the update trunk (A/B tests)
    use_split_gemm = os.environ.get("RLPYT_SPLIT_GEMM", "1") != "0"

    def _trunk_matmul(self, feat, weight):
        """``feat @ weight.T`` of the update: update-size batches go through the bf16x6 GEMM
        (``ops.linear_nobias``), small ones through the library."""
        if self.use_split_gemm and feat.shape[0] >= 1024:
            from ... import ops
            return ops.linear_nobias(feat, weight)
        return F.linear(feat, weight)

    def _conv_features(self, obs, flat_idx):
        from ... import ops
        c1, c2 = self.conv.conv.conv[0], self.conv.conv.conv[2]
        return ops.atari_conv_stack(obs, flat_idx, c1.weight, c1.bias, c2.weight, c2.bias)

    uses_prev_inputs = False   # prev_action / prev_reward are ignored (as in the reference)

    @torch.no_grad()
    def sample_step(self, image, prev_action=None, prev_reward=None, generator=None,
                    uniforms=None):
        """Sampling forward on the device: (action, prob, value) for ``[B, C, H, W]`` uint8
        observations -- conv stack, FC trunk, then heads + softmax + categorical draw fused in
        one kernel (inverse-CDF on ``torch.rand``; same distribution as the reference's
        ``torch.multinomial``, rlpyt/distributions/categorical.py:28-31)."""
        from ... import ops
        lead_dim, T, B, img_shape = infer_leading_dims(image, 3)
        assert lead_dim == 1, "sample_step takes [B, C, H, W] observations"
        if image.dtype == torch.uint8 and self.fused_conv:
            feat = self._conv_features(image.contiguous(), None)
            lin = self._single_fc()
            if lin is not None and B <= 256 and feat.shape[1] % 16 == 0:
                # split-K MFMA kernel: a library GEMM is latency-bound at this batch size
                fc_out = ops.fc_small(feat, lin.weight, lin.bias, relu=True)
            else:
                fc_out = self.conv.head(feat)
        else:
            fc_out = self.conv(prepare_image(image, B, img_shape))
        if uniforms is not None:      # (table [T', B], device row index): no RNG op here
            u, u_row = uniforms
        else:
            u, u_row = torch.rand(B, device=image.device, generator=generator), None
        prob, value, action = ops.categorical_head(fc_out, self.pi.weight, self.pi.bias,
                                                   self.value.weight, self.value.bias, u, u_row)
        return action, prob, value

    @property
    def fused_head_loss(self):
        """True when ``ops.ppo_head_loss`` can take over the heads (device + sizes)."""
        w = self.pi.weight
        return (w.is_cuda and w.dtype == torch.float32 and w.shape[0] <= 8
                and w.shape[1] in (256, 512))

    @torch.no_grad()
    def sample_step_into(self, image, out, push=None):
        """Sampling forward that also performs the step's row writes (``out``: the sampler's
        ``StepBinding``): conv stack -> split-K trunk -> ONE kernel that finishes the trunk, runs
        the heads + softmax + draw and writes prob[t], value[t], action[t+1] and the host-bound
        action copy.  With ``push`` (the sampler's ``FramePush``: frame-stacked uint8 batch whose
        row ``t`` is rebuilt on the device) the frame push and both convolutions are one launch
        and ``image`` is not read.  Returns False when this fused path does not apply."""
        from ... import ops
        lin = self._single_fc()
        if push is not None:
            obs = push.obs
            ok = (obs.is_cuda and obs.dtype == torch.uint8 and obs.dim() == 5
                  and tuple(obs.shape[2:]) == (4, 104, 80) and push.new_frame.shape[0] <= 256)
        else:
            ok = (isinstance(image, torch.Tensor) and image.is_cuda
                  and image.dtype == torch.uint8 and image.dim() == 4 and image.shape[0] <= 256)
        if not (ok and self.fused_conv and self.fused_head_loss and lin is not None
                and ops.rollout_fc_ok(push.new_frame.shape[0] if push is not None else image.shape[0],
                                      *lin.weight.shape)):
            return False
        if push is not None:
            B = push.new_frame.shape[0]
            c1, c2 = self.conv.conv.conv[0], self.conv.conv.conv[2]
            feat = ops.atari_sample_convs(obs, out.t_dev, out.lo, push.new_frame, push.full_rows,
                                          push.slot, c1.weight, c1.bias, c2.weight, c2.bias,
                                          scalar_rows=push.scalar_rows)
        else:
            B = image.shape[0]
            feat = self._conv_features(image.contiguous(), None)
        self._trunk_and_head(feat, lin, out, B)
        return True

    def _trunk_and_head(self, feat, lin, out, B, bootstrap_out=None):
        """Split-K trunk partials + the head kernel that finishes them (see csrc/step.hip)."""
        from ... import ops
        partial, ksplit = ops.rollout_fc_partials(feat, lin.weight)
        ops.rollout_head(partial, ksplit, lin.bias, self.pi.weight, self.pi.bias,
                         self.value.weight, self.value.bias, out.uniforms, out.t_dev, B,
                         out.prob_rows, out.value_rows, out.action_rows, out.lo, out.action_out,
                         bootstrap_out=bootstrap_out)
        return True

    @torch.no_grad()
    def sample_value_into(self, out, push, dst_stage, bootstrap_out):
        """Bootstrap value of the observation AFTER the last step of a batch (``t = T``) through
        the step's own kernels: frame push (rebuilt stacks -> ``dst_stage``, reward / done rows
        ``T`` committed) + conv1 + conv2, split-K trunk, value head only -> ``bootstrap_out [Bg]``.
        Returns False when this path does not apply (the caller then runs ``agent.value``)."""
        from ... import ops
        lin = self._single_fc()
        obs = push.obs
        ok = (obs.is_cuda and obs.dtype == torch.uint8 and obs.dim() == 5
              and tuple(obs.shape[2:]) == (4, 104, 80) and push.new_frame.shape[0] <= 256)
        if not (ok and self.fused_conv and self.fused_head_loss and lin is not None):
            return False
        B = push.new_frame.shape[0]
        if not ops.rollout_fc_ok(B, *lin.weight.shape):
            return False
        c1, c2 = self.conv.conv.conv[0], self.conv.conv.conv[2]
        feat = ops.atari_sample_convs(obs, out.t_dev, out.lo, push.new_frame, push.full_rows,
                                      push.slot, c1.weight, c1.bias, c2.weight, c2.bias,
                                      scalar_rows=push.scalar_rows, dst_stage=dst_stage)
        return self._trunk_and_head(feat, lin, out, B, bootstrap_out=bootstrap_out)

    def forward(self, image, prev_action, prev_reward, features_only=False):
        """[T,B,C,H,W] / [B,C,H,W] / [C,H,W] -> (pi, value) with the same lead dims.
        ``features_only``: return the trunk output ``[T*B, fc]`` instead (for the fused
        head + loss kernel); ``features_only="pre"``: return ``(z, trunk_bias)`` with ``z`` the
        trunk's pre-activation without its bias when the trunk is one Linear + ReLU on the fused
        conv path (the head + loss kernel then applies bias and ReLU), else ``(trunk output,
        None)``."""
        lin = self._single_fc() if features_only == "pre" else None
        if isinstance(image, ObsGather):
            lead_dim, T, B = 1, 1, image.flat_idx.numel()
            if self.fused_conv:
                feat = self._conv_features(*image)
                if lin is not None:    # x W^T only: bias + ReLU belong to the head+loss kernel
                    return self._trunk_matmul(feat, lin.weight), lin.bias
                fc_out = self.conv.head(feat)
            else:
                from ... import ops
                fc_out = self.conv(ops.obs_to_nhwc_f32(*image))
        else:
            lead_dim, T, B, img_shape = infer_leading_dims(image, 3)
            if image.dtype == torch.uint8 and image.is_cuda and self.fused_conv:
                feat = self._conv_features(image.contiguous().reshape(T * B, *img_shape), None)
                if lin is not None:
                    return self._trunk_matmul(feat, lin.weight), lin.bias
                fc_out = self.conv.head(feat)
            else:
                fc_out = self.conv(prepare_image(image, T * B, img_shape))
        if features_only == "pre":
            return fc_out, None
        if features_only:
            return fc_out
        pi = F.softmax(self.pi(fc_out), dim=-1)
        v = self.value(fc_out).squeeze(-1)
        return restore_leading_dims((pi, v), lead_dim, T, B)
