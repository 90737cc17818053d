"""Conv stacks with the module/parameter naming of rlpyt/models/conv2d.py:8-117
(``conv.<i>`` inside ``Conv2dModel``; ``conv`` + ``head`` inside ``Conv2dHeadModel``), so
state dicts interchange with the reference.

``Conv2dModel.features`` is the entry the DQN-family models use: raw observations in, flattened
features out.  For the DQN geometry (4x104x80 uint8 frames, 32-64-64 channels, kernels 8/4/3,
strides 4/2/1, paddings 0/1/1, ReLU) a forward of up to ``FUSED_MAX_IMAGES`` (no-grad) /
``FUSED_MAX_IMAGES_GRAD`` (autograd) images on the device --
every sampling step, every target-network pass, and (round 6) the online network's pass under autograd
-- runs ``rlpyt_dqn_convs_fwd_f32`` (weight packing + one f32-MFMA kernel per layer,
``csrc/dqn_convs.hip``) instead of ~18 library launches; under autograd the kernels' channels-last
activations are kept for the backward pass (``ops.dqn_convs``).  Anything else (other geometries,
float images, larger batches, CPU) takes the ``torch.nn.Conv2d`` modules."""
import os

import torch

from .mlp import MlpModel
from .utils import conv2d_output_shape


class Conv2dModel(torch.nn.Module):
    def __init__(self, in_channels, channels, kernel_sizes, strides, paddings=None,
                 nonlinearity=torch.nn.ReLU, use_maxpool=False, head_sizes=None):
        super().__init__()
        paddings = [0] * len(channels) if paddings is None else paddings
        assert len(channels) == len(kernel_sizes) == len(strides) == len(paddings)
        ins = [in_channels] + list(channels[:-1])
        pool_strides = strides if use_maxpool else [1] * len(strides)
        conv_strides = [1] * len(strides) if use_maxpool else strides
        seq = []
        for ic, oc, k, s, p, ms in zip(ins, channels, kernel_sizes, conv_strides, paddings,
                                       pool_strides):
            seq += [torch.nn.Conv2d(ic, oc, kernel_size=k, stride=s, padding=p), nonlinearity()]
            if ms > 1:
                seq.append(torch.nn.MaxPool2d(ms))
        self.conv = torch.nn.Sequential(*seq)
        self._dqn_geometry = (
            in_channels == 4 and list(channels) == [32, 64, 64] and list(kernel_sizes) == [8, 4, 3]
            and list(strides) == [4, 2, 1] and list(paddings) == [0, 1, 1] and not use_maxpool
            and nonlinearity is torch.nn.ReLU)

    # set False (or RLPYT_DQN_CONVS=0) for the library convolutions in no-grad forwards too (A/B tests)
    use_fused_nograd_convs = os.environ.get("RLPYT_DQN_CONVS", "1") != "0"
    # The own kernels win at every size measured, with and without autograd (round 6, workgroups persistent
    # over the images beyond 256: no-grad forward 63 / 136 us at 128 images .. 1451 / 2197 us at 5440 own /
    # library, forward + backward 230 / 414 us .. 4438 / 5339 us; profiles/r6_dqn_convs_large_n.log) --
    # bounded only by the workspaces (88 KB of kept activations + 88 KB of gradients per image).
    FUSED_MAX_IMAGES = int(os.environ.get("RLPYT_DQN_CONVS_MAX_N", 1 << 15))     # (env: A/B runs)
    FUSED_MAX_IMAGES_GRAD = int(os.environ.get("RLPYT_DQN_CONVS_MAX_N_GRAD", 1 << 15))

    _packed = None                  # the weights in the kernels' register order (sampling only)

    def forward(self, input):
        return self.conv(input)

    def refresh_step_weights(self):
        """Called by the agent when it enters sample / eval mode: re-pack the weights for the fused
        kernels ONCE; the forwards of that phase (module in eval mode, parameters frozen until the
        next mode change) then skip the per-call packing launch.  Any other no-grad forward -- target
        network, double-DQN pass during training -- packs on the stream and is always current."""
        w = self.conv[0].weight if self._dqn_geometry else None
        if (w is None or not (self.use_fused_nograd_convs and w.is_cuda and w.dtype == torch.float32)
                or torch.cuda.is_current_stream_capturing()):
            return
        from .. import ops
        with torch.no_grad():
            self._packed = ops.dqn_convs_pack(self.conv[0].weight, self.conv[2].weight,
                                              self.conv[4].weight, out=self._packed)
        self._packed_versions = self._weight_versions()

    def _weight_versions(self):
        """(storage address, in-place version counter) of the three conv weights: any optimizer
        step, ``load_state_dict`` / ``copy_`` or ``.to()`` after the pack changes one of them."""
        return tuple((self.conv[i].weight.data_ptr(), self.conv[i].weight._version) for i in (0, 2, 4))

    def _current_pack(self, device):
        """The phase's packed weights if they still describe the live parameters, else None (the
        forward then packs on the stream): weights can change while the module stays in eval mode
        -- ``agent.load_state_dict`` after ``eval_mode``, a direct ``model.eval()``."""
        if (self.training or self._packed is None or self._packed.device != device
                or getattr(self, "_packed_versions", None) != self._weight_versions()):
            return None
        return self._packed

    # set False (or RLPYT_DQN_CONVS_GRAD=0) for the library convolutions in the forward under autograd
    use_fused_grad_convs = os.environ.get("RLPYT_DQN_CONVS_GRAD", "1") != "0"

    def _fused_ok(self, observation, T_B, img_shape):
        grad = torch.is_grad_enabled()
        on = self.use_fused_grad_convs if grad else self.use_fused_nograd_convs
        limit = self.FUSED_MAX_IMAGES_GRAD if grad else self.FUSED_MAX_IMAGES
        return (on and self._dqn_geometry
                and observation.is_cuda and observation.dtype == torch.uint8
                and tuple(img_shape) == (4, 104, 80) and 0 < T_B <= limit
                and self.conv[0].weight.dtype == torch.float32 and self.conv[0].weight.is_cuda
                and all(self.conv[i].weight.is_contiguous() for i in (0, 2, 4)))

    def features(self, observation, T_B, img_shape):
        """Flattened features ``[T*B, n]`` (the reference's ``conv(img).view(T * B, -1)``,
        rlpyt/models/dqn/atari_dqn_model.py:62-63) of raw observations with leading dims folded."""
        if self._fused_ok(observation, T_B, img_shape):
            from .. import ops
            c1, c2, c3 = self.conv[0], self.conv[2], self.conv[4]
            if torch.is_grad_enabled():          # online network of an update: own forward, kept activations
                return ops.dqn_convs(observation.reshape(T_B, *img_shape).contiguous(), c1.weight, c1.bias,
                                     c2.weight, c2.bias, c3.weight, c3.bias)
            packed = self._current_pack(observation.device)
            return ops.dqn_convs_fwd(observation.reshape(T_B, *img_shape).contiguous(), c1.weight, c1.bias,
                                     c2.weight, c2.bias, c3.weight, c3.bias, packed=packed)
        from .pg.atari_ff_model import prepare_image
        return self.conv(prepare_image(observation, T_B, img_shape)).reshape(T_B, -1)

    def conv_out_size(self, h, w, c=None):
        for m in self.conv.children():
            if isinstance(m, (torch.nn.Conv2d, torch.nn.MaxPool2d)):
                h, w = conv2d_output_shape(h, w, m.kernel_size, m.stride, m.padding)
            if isinstance(m, torch.nn.Conv2d):
                c = m.out_channels
        return h * w * c


class Conv2dHeadModel(torch.nn.Module):
    def __init__(self, image_shape, channels, kernel_sizes, strides, hidden_sizes,
                 output_size=None, paddings=None, nonlinearity=torch.nn.ReLU,
                 use_maxpool=False):
        super().__init__()
        c, h, w = image_shape
        self.conv = Conv2dModel(c, channels, kernel_sizes, strides, paddings=paddings,
                                nonlinearity=nonlinearity, use_maxpool=use_maxpool)
        n = self.conv.conv_out_size(h, w)
        if hidden_sizes or output_size:
            self.head = MlpModel(n, hidden_sizes, output_size=output_size,
                                 nonlinearity=nonlinearity)
            self._output_size = self.head.output_size
        else:
            self.head = lambda x: x
            self._output_size = n

    def forward(self, input):
        return self.head(self.conv(input).reshape(input.shape[0], -1))

    def refresh_step_weights(self):
        self.conv.refresh_step_weights()

    def from_observation(self, observation, T_B, img_shape):
        """``forward`` on raw observations (see ``Conv2dModel.features``)."""
        return self.head(self.conv.features(observation, T_B, img_shape))

    @property
    def output_size(self):
        return self._output_size
