"""AtariLstmModel: conv(4->16,k8,s4) ReLU conv(16->32,k4,s2,p1) ReLU FC 512 ReLU ->
LSTM(512 + A + 1 -> 512) -> {pi: FC A + softmax, value: FC 1} (architecture, argument and
parameter names of rlpyt/models/pg/atari_lstm_model.py:12-77, so state dicts interchange).

On the device the convolution stack is the same hand-written fp32-MFMA path as AtariFfModel's
(``ops.atari_conv_stack``: uint8 in, 3456 features out, forward and backward) when the geometry
is the default one; the LSTM runs through ``nn.LSTM`` (MIOpen RNN on ROCm) and always keeps the
B dimension in the returned state ``RnnState(h, c)`` of shape ``[N, B, H]``."""
import os

import torch
import torch.nn.functional as F

from ...utils.collections import namedarraytuple
from ...utils.tensor import infer_leading_dims, restore_leading_dims
from ..conv2d import Conv2dHeadModel
from .atari_ff_model import prepare_image

RnnState = namedarraytuple("RnnState", ["h", "c"])


class AtariLstmModel(torch.nn.Module):
    def __init__(self, image_shape, output_size, fc_sizes=512, lstm_size=512, use_maxpool=False,
                 channels=None, kernel_sizes=None, strides=None, paddings=None):
        super().__init__()
        self.conv = Conv2dHeadModel(
            image_shape=image_shape, channels=channels or [16, 32],
            kernel_sizes=kernel_sizes or [8, 4], strides=strides or [4, 2],
            paddings=paddings or [0, 1], use_maxpool=use_maxpool, hidden_sizes=fc_sizes)
        self.lstm = torch.nn.LSTM(self.conv.output_size + output_size + 1, lstm_size)
        self.pi = torch.nn.Linear(lstm_size, output_size)
        self.value = torch.nn.Linear(lstm_size, 1)
        self._default_geometry = (
            tuple(image_shape) == (4, 104, 80) and not use_maxpool
            and list(channels or [16, 32]) == [16, 32] and list(kernel_sizes or [8, 4]) == [8, 4]
            and list(strides or [4, 2]) == [4, 2] and list(paddings or [0, 1]) == [0, 1])
        self.use_fused_conv = True

    # set False (or RLPYT_LSTM_STEP=0) for the library RNN in the one-step sampling forward too
    use_fused_lstm_step = os.environ.get("RLPYT_LSTM_STEP", "1") != "0"
    _lstm_step = None

    def _fused_step_ok(self, T, B, fc_out, init_rnn_state):
        if not (self.use_fused_lstm_step and T == 1 and B <= 256 and init_rnn_state is not None
                and not torch.is_grad_enabled() and fc_out.is_cuda
                and fc_out.dtype == torch.float32 and self.lstm.hidden_size % 16 == 0):
            return False
        h0, _c0 = tuple(init_rnn_state)
        return h0.dim() == 3 and h0.shape[0] == 1 and h0.shape[1] == B

    def refresh_step_weights(self):
        """Bring the fused step's weight buffer up to date (captured step graphs read it by address).
        Unconditional: two small copies per iteration, independent of how the optimizer wrote the
        parameters (a raw-pointer kernel does not have to bump ``Tensor._version``)."""
        if self._lstm_step is not None:
            self._lstm_step.refresh(force=True)

    @property
    def fused_conv(self):
        w = self.conv.conv.conv[0].weight
        return (self.use_fused_conv and self._default_geometry and w.is_cuda
                and w.dtype == torch.float32)

    def forward(self, image, prev_action, prev_reward, init_rnn_state):
        """[T,B,C,H,W] / [B,C,H,W] / [C,H,W] uint8 -> (pi, value, RnnState [N,B,H]);
        ``prev_action`` one-hot."""
        lead_dim, T, B, img_shape = infer_leading_dims(image, 3)
        if image.dtype == torch.uint8 and image.is_cuda and self.fused_conv:
            from ... import ops
            c1, c2 = self.conv.conv.conv[0], self.conv.conv.conv[2]
            feat = ops.atari_conv_stack(image.contiguous().reshape(T * B, *img_shape), None,
                                        c1.weight, c1.bias, c2.weight, c2.bias)
            fc_out = self.conv.head(feat)
        else:
            fc_out = self.conv(prepare_image(image, T * B, img_shape))
        if self._fused_step_ok(T, B, fc_out, init_rnn_state):
            # sampling forward, one time step: gate GEMM + cell as two launches (ops.LstmStep)
            if self._lstm_step is None:
                from ... import ops
                self._lstm_step = ops.LstmStep(self.lstm)
            h0, c0 = tuple(init_rnn_state)
            hn, cn = self._lstm_step.step([fc_out, prev_action, prev_reward], h0[0], c0[0])
            flat, hn, cn = hn, hn.unsqueeze(0), cn.unsqueeze(0)
        else:
            lstm_input = torch.cat([fc_out.reshape(T, B, -1),
                                    prev_action.reshape(T, B, -1).to(fc_out.dtype),
                                    prev_reward.reshape(T, B, 1).to(fc_out.dtype)], dim=2)
            state = (None if init_rnn_state is None
                     else tuple(x.contiguous() for x in init_rnn_state))
            lstm_out, (hn, cn) = self.lstm(lstm_input, state)
            flat = lstm_out.reshape(T * B, -1)
        pi = F.softmax(self.pi(flat), dim=-1)
        v = self.value(flat).squeeze(-1)
        pi, v = restore_leading_dims((pi, v), lead_dim, T, B)
        return pi, v, RnnState(h=hn, c=cn)
