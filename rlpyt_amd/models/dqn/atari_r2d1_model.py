"""AtariR2d1Model: conv 32-64-64 -> FC 512 (ReLU) -> LSTM(512 + A + 1 -> 512) -> MLP / dueling head
-> Q[A] (architecture, argument and parameter names of
rlpyt/models/dqn/atari_r2d1_model.py:13-77, so state dicts interchange).

On the device the uint8 frames are converted by ``rlpyt_obs_to_nhwc_f32`` (one kernel, no f32
NCHW copy) under autograd, and go straight into ``rlpyt_dqn_convs_fwd_f32`` in no-grad forwards of
up to 1024 images (sampling steps; ``Conv2dModel.features``); the LSTM runs through torch's ``nn.LSTM`` (MIOpen RNN on ROCm) for sequences and for
anything under autograd, through ``ops.LstmStep`` (split-K gate GEMM + one cell kernel) for the
one-step sampling forward and through ``ops.lstm_sequence`` (one launch per time step) for the no-grad
sequence passes of the update; the returned state ``RnnState(h, c)`` always keeps the B dimension,
shape ``[N, B, H]``."""
import os

import torch

from ...utils.collections import namedarraytuple
from ...utils.tensor import infer_leading_dims, restore_leading_dims
from ..conv2d import Conv2dHeadModel
from ..mlp import MlpModel
from .dueling import DuelingHeadModel

RnnState = namedarraytuple("RnnState", ["h", "c"])


class AtariR2d1Model(torch.nn.Module):
    def __init__(self, image_shape, output_size, fc_size=512, lstm_size=512, head_size=512,
                 dueling=False, use_maxpool=False, channels=None, kernel_sizes=None,
                 strides=None, paddings=None):
        super().__init__()
        self.dueling = dueling
        self.conv = Conv2dHeadModel(
            image_shape=image_shape, channels=channels or [32, 64, 64],
            kernel_sizes=kernel_sizes or [8, 4, 3], strides=strides or [4, 2, 1],
            paddings=paddings or [0, 1, 1], use_maxpool=use_maxpool, hidden_sizes=fc_size)
        self.lstm = torch.nn.LSTM(self.conv.output_size + output_size + 1, lstm_size)
        self.head = (DuelingHeadModel(lstm_size, head_size, output_size) if dueling
                     else MlpModel(lstm_size, head_size, output_size=output_size))

    # set False (or RLPYT_LSTM_STEP=0) for the library RNN in the one-step sampling forward too (A/B tests)
    use_fused_lstm_step = os.environ.get("RLPYT_LSTM_STEP", "1") != "0"
    # ... and RLPYT_LSTM_SEQ=0 for the library RNN in no-grad SEQUENCE forwards
    use_fused_lstm_sequence = os.environ.get("RLPYT_LSTM_SEQ", "1") != "0"
    _lstm_step = None

    def _fused_step_ok(self, T, B, conv_out, init_rnn_state):
        if not (self.use_fused_lstm_step and T == 1 and B <= 256 and init_rnn_state is not None
                and not torch.is_grad_enabled() and conv_out.is_cuda
                and conv_out.dtype == torch.float32 and self.lstm.hidden_size % 16 == 0):
            return False
        h0, _c0 = tuple(init_rnn_state)     # (a namedarraytuple indexes its leaves: iterate for the fields)
        return h0.dim() == 3 and h0.shape[0] == 1 and h0.shape[1] == B

    def refresh_step_weights(self):
        """Bring the fused step's weight buffer up to date (captured step graphs read it by address).
        Unconditional: two small copies per iteration, independent of how the optimizer wrote the
        parameters (a raw-pointer kernel does not have to bump ``Tensor._version``)."""
        if self._lstm_step is not None:
            self._lstm_step.refresh(force=True)
        self.conv.refresh_step_weights()

    def sample_step_ok(self, observation, prev_action):
        """Whether ``sample_step`` serves this sampling step (else: ``forward`` through the agent's
        eager path)."""
        lin = getattr(self.conv.head, "model", None)
        return (self.use_fused_lstm_step and not torch.is_grad_enabled() and observation.is_cuda
                and observation.dim() == 4 and observation.shape[0] <= 256
                and isinstance(prev_action, torch.Tensor) and prev_action.dtype == torch.int64
                and prev_action.dim() == 1 and self.lstm.hidden_size % 16 == 0
                and self.lstm.weight_ih_l0.dtype == torch.float32
                and isinstance(lin, torch.nn.Sequential) and len(lin) == 2
                and isinstance(lin[0], torch.nn.Linear) and isinstance(lin[1], torch.nn.ReLU))

    def sample_step(self, observation, prev_action, prev_reward, done, h_state, c_state):
        """One sampling step with the collector's reset handling folded in: ``observation [B,C,H,W]``,
        ``prev_action [B]`` (indices), ``prev_reward [B]`` as stored -- NOT nulled --, ``done [B]``
        (bool; None: no resets) = environments that were reset before this step, ``h_state`` /
        ``c_state [B,H]`` the persistent recurrent state, UPDATED IN PLACE.  Conv stack -> trunk
        GEMM -> ``rlpyt_rnn_step_inputs_f32`` (trunk ReLU, one-hot, nulling, state zeroing, ``[x | h]``
        row, the state the step starts from) -> gate GEMM -> cell -> head.  Returns
        ``(q [B,A], prev_h [B,H], prev_c [B,H])``."""
        from ... import ops
        B, img_shape = observation.shape[0], observation.shape[1:]
        lin = self.conv.head.model[0]
        pre = torch.nn.functional.linear(self.conv.conv.features(observation, B, img_shape),
                                         lin.weight, lin.bias)
        if self._lstm_step is None:
            self._lstm_step = ops.LstmStep(self.lstm)
        step = self._lstm_step
        n_actions = self.lstm.input_size - self.conv.output_size - 1
        xh, prev_h, prev_c = ops.rnn_step_inputs(pre, prev_action, prev_reward.float(), done, h_state,
                                                 c_state, n_actions, step.Kp, relu=True)
        step.step_rows(xh, h_state, c_state)
        return self.head(h_state), prev_h, prev_c

    def forward(self, observation, prev_action, prev_reward, init_rnn_state):
        """Leading dims [T,B], [B] or []; prev_action one-hot; returns (q, RnnState [N,B,H])."""
        lead_dim, T, B, img_shape = infer_leading_dims(observation, 3)
        conv_out = self.conv.from_observation(observation, T * B, img_shape)
        if self._fused_step_ok(T, B, conv_out, init_rnn_state):
            # sampling forward, one time step: gate GEMM + cell as two launches (ops.LstmStep)
            if self._lstm_step is None:
                from ... import ops
                self._lstm_step = ops.LstmStep(self.lstm)
            h0, c0 = tuple(init_rnn_state)
            hn, cn = self._lstm_step.step([conv_out, prev_action, prev_reward], h0[0], c0[0])
            q = self.head(hn)
            return restore_leading_dims(q, lead_dim, T, B), RnnState(h=hn.unsqueeze(0),
                                                                     c=cn.unsqueeze(0))
        lstm_input = torch.cat([conv_out.reshape(T, B, -1),
                                prev_action.reshape(T, B, -1).to(conv_out.dtype),
                                prev_reward.reshape(T, B, 1).to(conv_out.dtype)], dim=2)
        state = None if init_rnn_state is None else tuple(x.contiguous() for x in init_rnn_state)
        if self.use_fused_lstm_sequence and T > 1:
            from ... import ops
            if ops.lstm_sequence_ok(self.lstm, lstm_input, None if state is None else state[0]):
                # no-grad sequence (target / warm-up / double-DQN passes of the update): one launch
                # per time step
                lstm_out, (hn, cn) = ops.lstm_sequence(self.lstm, lstm_input, *(state or (None, None)))
                q = self.head(lstm_out.reshape(T * B, -1))
                return restore_leading_dims(q, lead_dim, T, B), RnnState(h=hn, c=cn)
            if ops.lstm_sequence_train_ok(self.lstm, lstm_input, None if state is None else state[0]):
                # the online network's training pass: own forward (gates kept) + own BPTT
                lstm_out, (hn, cn) = ops.lstm_sequence_train(self.lstm, lstm_input, *(state or (None, None)))
                q = self.head(lstm_out.reshape(T * B, -1))
                return restore_leading_dims(q, lead_dim, T, B), RnnState(h=hn, c=cn)
        lstm_out, (hn, cn) = self.lstm(lstm_input, state)
        q = self.head(lstm_out.reshape(T * B, -1))
        return restore_leading_dims(q, lead_dim, T, B), RnnState(h=hn, c=cn)
