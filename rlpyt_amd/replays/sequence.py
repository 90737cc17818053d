"""Sequence replay buffers on the device (R2D1 path): classes and call signatures of
rlpyt/replays/sequence/{n_step,uniform,prioritized,frame}.py.

``extract_batch`` returns ``[T(+n), B]`` sequences assembled by the sequence gather kernels
(``rlpyt_gather_sequences`` for the small fields, ``rlpyt_frames_gather_seq`` for the frame
store: 266 MB per R2D2 batch written at HBM speed instead of a Python double loop of
strided host copies).  RNN state is stored every ``rnn_state_interval`` steps as in the
reference; tree geometry for the prioritized variant follows
rlpyt/replays/sequence/prioritized.py:56-68.
"""
import math

import numpy as np
import torch

from .. import ops
from ..utils.buffer import buffer_from_example, buffer_func
from ..utils.collections import namedarraytuple
from ..utils.quick_args import save__init__args
from .n_step import BaseNStepReturnBuffer, FrameBufferMixin, _to_device

SamplesFromReplay = namedarraytuple("SamplesFromReplay",
                                    ["all_observation", "all_action", "all_reward", "return_",
                                     "done", "done_n", "init_rnn_state"])
SamplesFromReplayPri = namedarraytuple("SamplesFromReplayPri",
                                       SamplesFromReplay._fields + ("is_weights",))
SamplesToBuffer = None


class SequenceNStepReturnBuffer(BaseNStepReturnBuffer):
    def __init__(self, example, size, B, rnn_state_interval, batch_T=None, **kwargs):
        self.rnn_state_interval = rnn_state_interval
        self.batch_T = batch_T
        self._rnn_example = None
        if rnn_state_interval <= 1:
            buffer_example = example
        else:
            field_names = [f for f in example._fields if f != "prev_rnn_state"]
            global SamplesToBuffer
            SamplesToBuffer = namedarraytuple("SamplesToBuffer", field_names)
            buffer_example = SamplesToBuffer(*(v for k, v in example.items()
                                               if k != "prev_rnn_state"))
            size = B * rnn_state_interval * math.ceil(math.ceil(size / B) / rnn_state_interval)
            self._rnn_example = example.prev_rnn_state
        super().__init__(example=buffer_example, size=size, B=B, **kwargs)
        if rnn_state_interval > 1:
            assert self.T % rnn_state_interval == 0
            self.rnn_T = self.T // rnn_state_interval
            self.samples_prev_rnn_state = buffer_from_example(
                self._rnn_example, (self.rnn_T, B), device=self.device)

    def append_samples(self, samples):
        """RNN state kept every ``rnn_state_interval`` steps
        (rlpyt/replays/sequence/n_step.py:49-66)."""
        t, rsi = self.t, self.rnn_state_interval
        if rsi <= 1:
            return super().append_samples(samples)
        buffer_samples = SamplesToBuffer(*(v for k, v in samples.items()
                                           if k != "prev_rnn_state"))
        T, idxs = super().append_samples(buffer_samples)
        start, stop = math.ceil(t / rsi), ((t + T - 1) // rsi) + 1
        offset = (rsi - t) % rsi
        if stop > self.rnn_T:
            rnn_idxs = torch.arange(start, stop, device=self.device) % self.rnn_T
        else:
            rnn_idxs = slice(start, stop)
        state = buffer_func(samples.prev_rnn_state, lambda x: _to_device(x, self.device))
        self.samples_prev_rnn_state[rnn_idxs] = state[offset::rsi]
        return T, idxs

    def _idx(self, x):
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(x)
        return torch.as_tensor(x).to(device=self.device, dtype=torch.int64)

    def extract_batch(self, T_idxs, B_idxs, T):
        """rlpyt/replays/sequence/n_step.py:68-100."""
        s, rsi = self.samples, self.rnn_state_interval
        T_idxs, B_idxs = self._idx(T_idxs), self._idx(B_idxs)
        if rsi > 1:
            init_rnn_state = buffer_func(self.samples_prev_rnn_state, ops.gather_rows,
                                         T_idxs // rsi, B_idxs)
        elif rsi == 1:
            init_rnn_state = buffer_func(self.samples.prev_rnn_state, ops.gather_rows,
                                         T_idxs, B_idxs)
        else:
            init_rnn_state = None
        Tn = T + self.n_step_return
        return SamplesFromReplay(
            all_observation=self.extract_observation(T_idxs, B_idxs, Tn),
            all_action=buffer_func(s.action, ops.extract_sequences, T_idxs - 1, B_idxs, Tn),
            all_reward=ops.extract_sequences(s.reward, T_idxs - 1, B_idxs, Tn),
            return_=ops.extract_sequences(self.samples_return_, T_idxs, B_idxs, T),
            done=ops.extract_sequences(s.done, T_idxs, B_idxs, T),
            done_n=ops.extract_sequences(self.samples_done_n, T_idxs, B_idxs, T),
            init_rnn_state=init_rnn_state)

    def extract_observation(self, T_idxs, B_idxs, T):
        return buffer_func(self.samples.observation, ops.extract_sequences, T_idxs, B_idxs, T)


class UniformSequenceReplay:
    def set_batch_T(self, batch_T):
        self.batch_T = batch_T

    def sample_batch(self, batch_B, batch_T=None):
        batch_T = self.batch_T if batch_T is None else batch_T
        T_idxs, B_idxs = self.sample_idxs(batch_B, batch_T)
        return self.extract_batch(T_idxs, B_idxs, batch_T)

    def sample_idxs(self, batch_B, batch_T):
        """rlpyt/replays/sequence/uniform.py:24-39 (same np.random draws)."""
        t, b, f = self.t, self.off_backward + batch_T, self.off_forward
        high = self.T - b - f if self._buffer_full else t - b - f
        T_idxs = np.random.randint(low=0, high=high, size=(batch_B,))
        T_idxs[T_idxs >= t - b] += min(t, b) + f
        if self.rnn_state_interval > 0:
            T_idxs = (T_idxs // self.rnn_state_interval) * self.rnn_state_interval
        B_idxs = np.random.randint(low=0, high=self.B, size=(batch_B,))
        return T_idxs, B_idxs


class PrioritizedSequenceReplay:
    def __init__(self, alpha=0.6, beta=0.4, default_priority=1, unique=False,
                 input_priorities=False, input_priority_shift=0, **kwargs):
        super().__init__(**kwargs)
        save__init__args(locals())
        assert self.batch_T is not None, "Must assign fixed batch_T for prioritized."
        if unique:
            raise NotImplementedError("unique=True sampling is not on the hot path")
        self.init_priority_tree()

    def init_priority_tree(self):
        rsi = max(1, self.rnn_state_interval)
        off_backward = math.ceil((1 + self.off_backward + self.batch_T) / rsi)
        self.priority_tree = ops.DeviceSumTree(
            T=self.T // rsi, B=self.B, off_backward=off_backward,
            off_forward=math.ceil(self.off_forward / rsi),
            default_value=self.default_priority ** self.alpha,
            enable_input_priorities=self.input_priorities,
            input_priority_shift=self.input_priority_shift, device=self.device)

    def set_beta(self, beta):
        self.beta = beta

    def append_samples(self, samples):
        """NB: input priorities are NOT raised to alpha here, as in the reference
        (sequence/prioritized.py:78-80 vs non_sequence/prioritized.py:52)."""
        if hasattr(samples, "priorities"):
            priorities = torch.as_tensor(samples.priorities, device=self.device).double()
            samples = samples.samples
        else:
            priorities = None
        t, rsi = self.t, self.rnn_state_interval
        T, idxs = super().append_samples(samples)
        if rsi <= 1:
            self.priority_tree.advance(T, priorities=priorities)
        else:
            if priorities is not None and priorities.dim() == 2:
                offset = (rsi - t) % rsi
                priorities = priorities[offset::rsi].contiguous()
            n = self.t // rsi - t // rsi
            if self.t < t:
                n += self.T // rsi
            self.priority_tree.advance(n, priorities=priorities)
        return T, idxs

    def sample_batch(self, batch_B):
        u = torch.from_numpy(np.random.rand(int(batch_B))).to(self.device, non_blocking=True)
        T_idxs, B_idxs, priorities = self.priority_tree.sample(u)
        if self.rnn_state_interval > 1:
            T_idxs = T_idxs * self.rnn_state_interval
        batch = self.extract_batch(T_idxs, B_idxs, self.batch_T)
        is_weights = (1. / priorities) ** self.beta  # no epsilon here (prioritized.py:108)
        is_weights = (is_weights / is_weights.max()).float()
        return SamplesFromReplayPri(*batch, is_weights=is_weights)

    def update_batch_priorities(self, priorities):
        p = priorities.detach().to(self.device) ** self.alpha
        self.priority_tree.update_batch_priorities(p)


class SequenceNStepFrameBuffer(FrameBufferMixin, SequenceNStepReturnBuffer):
    def extract_observation(self, T_idxs, B_idxs, T):
        """[T,B,C,H,W] with wrap and post-reset blanking
        (rlpyt/replays/sequence/frame.py:17-50) in one gather kernel."""
        return ops.frames_gather_seq(self.samples_frames, self.samples.done,
                                     self._idx(T_idxs), self._idx(B_idxs), self.n_frames, T)


class UniformSequenceReplayBuffer(UniformSequenceReplay, SequenceNStepReturnBuffer):
    pass


class PrioritizedSequenceReplayBuffer(PrioritizedSequenceReplay, SequenceNStepReturnBuffer):
    pass


class UniformSequenceReplayFrameBuffer(UniformSequenceReplay, SequenceNStepFrameBuffer):
    pass


class PrioritizedSequenceReplayFrameBuffer(PrioritizedSequenceReplay, SequenceNStepFrameBuffer):
    pass
