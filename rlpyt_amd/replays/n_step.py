"""HBM-resident n-step-return replay ring and unique-frame store.

Contracts restated from rlpyt/replays/n_step.py:11-108 and rlpyt/replays/frame.py:10-59
(SURVEY.md App. A): ring ``samples[T, B]`` with cursor ``t``; rows within ``off_backward``
behind and ``off_forward`` ahead of the cursor are invalid; with ``n_step_return > 1`` the
n-step return / done_n of row ``r`` are stored at row ``r``; the frame store is
``[T + C - 1, B, H, W]`` with the oldest frame of time ``r`` at row ``r`` and the first
``C - 1`` rows mirroring the last ``C - 1`` after a wrap.

MI355X design: the whole ring lives in device memory (the 1M-frame Atari store is 8.3 GB
of 288 GB), appends are device-to-device writes from the HBM-resident sampler batch, and
n-step returns are recomputed by the ``rlpyt_nstep_return_f32`` kernel on the touched
window -- nothing returns to the host.
"""
import math

import numpy as np
import torch

from .. import ops
from ..utils.buffer import buffer_from_example, get_leading_dims
from ..utils.collections import namedarraytuple


class BaseReplayBuffer:
    async_ = False

    def append_samples(self, samples):
        raise NotImplementedError

    def sample_batch(self, batch_B):
        raise NotImplementedError


def _to_device(x, device):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    return x if x.device == device else x.to(device, non_blocking=True)


class BaseNStepReturnBuffer(BaseReplayBuffer):
    def __init__(self, example, size, B, discount=1, n_step_return=1, device=None):
        self.T = T = math.ceil(size / B)
        self.B = B
        self.size = T * B
        self.discount = discount
        self.n_step_return = n_step_return
        self.t = 0
        self.device = torch.device(device if device is not None else
                                   f"cuda:{torch.cuda.current_device()}")
        self.samples = buffer_from_example(example, (T, B), device=self.device)
        if n_step_return > 1:
            self.samples_return_ = buffer_from_example(example.reward, (T, B),
                                                       device=self.device)
            self.samples_done_n = buffer_from_example(example.done, (T, B), device=self.device)
        else:
            self.samples_return_ = self.samples.reward
            self.samples_done_n = self.samples.done
        self._buffer_full = False
        self.off_backward = n_step_return
        self.off_forward = 1

    def _ring_idxs(self, start, stop):
        """slice when the range is inside the ring, else a device index vector."""
        if start >= 0 and stop <= self.T:
            return slice(start, stop)
        return torch.arange(start, stop, device=self.device) % self.T

    def append_samples(self, samples):
        """Ring write (wrap allowed anywhere) + n-step return refresh
        (rlpyt/replays/n_step.py:62-79)."""
        T, B = get_leading_dims(samples, n_dim=2)
        assert B == self.B
        t = self.t
        idxs = self._ring_idxs(t, t + T)
        dev = self.device
        moved = type(samples)(*(None if f is None else _to_device(f, dev) for f in samples))
        self.samples[idxs] = moved
        self.compute_returns(T)
        if not self._buffer_full and t + T >= self.T:
            self._buffer_full = True
        self.t = (t + T) % self.T
        return T, idxs

    def compute_returns(self, T):
        """n-step returns for rows t-(n-1) .. t+T-n (rlpyt/replays/n_step.py:81-108)."""
        if self.n_step_return == 1:
            return
        t, s, nm1 = self.t, self.samples, self.n_step_return - 1
        if t - nm1 >= 0 and t + T <= self.T:
            ops.discount_return_n_step(
                s.reward[t - nm1:t + T], s.done[t - nm1:t + T], self.n_step_return,
                self.discount, return_dest=self.samples_return_[t - nm1:t - nm1 + T],
                done_n_dest=self.samples_done_n[t - nm1:t - nm1 + T])
        else:
            idxs = torch.arange(t - nm1, t + T, device=self.device) % self.T
            ret, dn = ops.discount_return_n_step(s.reward[idxs], s.done[idxs],
                                                 self.n_step_return, self.discount)
            dest = idxs[:-nm1]
            self.samples_return_[dest] = ret
            self.samples_done_n[dest] = dn


BufferSamples = None


class FrameBufferMixin:
    """Stores only the newest frame of each observation (rlpyt/replays/frame.py:10-59)."""

    def __init__(self, example, **kwargs):
        field_names = [f for f in example._fields if f != "observation"]
        global BufferSamples
        BufferSamples = namedarraytuple("BufferSamples", field_names)
        buffer_example = BufferSamples(*(v for k, v in example.items() if k != "observation"))
        super().__init__(example=buffer_example, **kwargs)
        obs_ex = example.observation
        self.n_frames = n_frames = int(np.asarray(
            obs_ex.cpu() if isinstance(obs_ex, torch.Tensor) else obs_ex).shape[0])
        frame_ex = obs_ex[0]
        self.samples_frames = buffer_from_example(frame_ex, (self.T + n_frames - 1, self.B),
                                                  device=self.device)
        self.samples_new_frames = self.samples_frames[n_frames - 1:]
        self.off_forward = max(self.off_forward, n_frames - 1)

    def append_samples(self, samples):
        t, fm1 = self.t, self.n_frames - 1
        buffer_samples = BufferSamples(*(v for k, v in samples.items() if k != "observation"))
        T, idxs = super().append_samples(buffer_samples)
        obs = _to_device(samples.observation, self.device)
        self.samples_new_frames[idxs] = obs[:, :, -1]
        if t == 0:
            for f in range(fm1):
                self.samples_frames[f] = obs[0, :, f]
        elif self.t < t and fm1 > 0:
            self.samples_frames[:fm1] = self.samples_frames[-fm1:]
        return T, idxs
