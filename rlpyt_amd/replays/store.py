"""Storage parts of the HBM-resident replay: a cursor over a ``[T, B]`` ring, the field ring with its
n-step returns, the unique-frame store and the periodic RNN-state store.

The CONTRACT is the reference's (rlpyt/replays/n_step.py:11-108,
rlpyt/replays/frame.py:10-59, rlpyt/replays/sequence/n_step.py:26-66) because sampled index streams
and gathered batches must equal its bit for bit: ring ``[T = ceil(size / B), B]`` with a write
cursor; rows within ``guard_back`` behind and ``guard_fwd`` ahead of the cursor are not sampleable;
with ``n_step > 1`` the n-step return / done_n of row r live at row r; the frame store is
``[T + C - 1, B, H, W]`` with the oldest frame of time r at row r and its first ``C - 1`` rows
mirroring the last ``C - 1`` after a wrap; RNN states are kept every ``interval`` steps.

The STRUCTURE is this repo's: the reference stacks these behaviours as cooperating mixin classes on
host numpy arrays; here they are independent parts over device tensors that a ``ReplayBuffer``
(buffers.py) composes -- appends are device-to-device writes from the sampler's HBM batch, the
n-step refresh is one ``rlpyt_nstep_return_f32`` launch on the touched window, and nothing returns
to the host."""
import math
from collections import namedtuple

import numpy as np
import torch

from .. import ops
from ..utils.buffer import buffer_from_example, buffer_func

Claim = namedtuple("Claim", ["start", "count", "rows", "wrapped"])


def on_device(x, device):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    return x if x.device == device else x.to(device, non_blocking=True)


def as_index(x, device):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    return torch.as_tensor(x).to(device=device, dtype=torch.int64)


class RingCursor:
    """Write position of a ``[T, B]`` ring and the guard bands around it."""

    def __init__(self, size, B, guard_back, guard_fwd, device):
        self.T, self.B = math.ceil(size / B), B
        self.t, self.full = 0, False
        self.guard_back, self.guard_fwd = guard_back, guard_fwd
        self.device = device

    def span(self, start, stop):
        """Rows ``start .. stop - 1`` modulo T: a slice while they are inside the ring, else a
        device index vector (negative starts wrap to the ring's end)."""
        if 0 <= start and stop <= self.T:
            return slice(start, stop)
        return torch.arange(start, stop, device=self.device) % self.T

    def claim(self, n):
        """Reserve the next ``n`` rows for a write and move on."""
        start = self.t
        self.full = self.full or start + n >= self.T
        self.t = (start + n) % self.T
        return Claim(start, n, self.span(start, start + n), self.t < start)


class FieldRing:
    """The small per-step fields (action, reward, done, ...) and their n-step returns."""

    def __init__(self, example, cursor, discount, n_step):
        self.cursor, self.discount, self.n_step = cursor, discount, n_step
        shape, dev = (cursor.T, cursor.B), cursor.device
        self.data = buffer_from_example(example, shape, device=dev)
        if n_step > 1:
            self.return_ = buffer_from_example(example.reward, shape, device=dev)
            self.done_n = buffer_from_example(example.done, shape, device=dev)
        else:                      # 1-step: the reward IS the return (rlpyt/replays/n_step.py:88-89)
            self.return_, self.done_n = self.data.reward, self.data.done

    def write(self, samples, claim):
        dev = self.cursor.device
        self.data[claim.rows] = buffer_func(samples, on_device, dev)
        if self.n_step > 1:
            self._refresh_returns(claim)

    def _refresh_returns(self, claim):
        """Returns of the rows whose n-step window the new rows completed: ``start - (n - 1)``
        through ``start + count - n``."""
        back, d = self.n_step - 1, self.data
        lo, hi = claim.start - back, claim.start + claim.count
        if lo >= 0 and hi <= self.cursor.T:
            ops.discount_return_n_step(
                d.reward[lo:hi], d.done[lo:hi], self.n_step, self.discount,
                return_dest=self.return_[lo:lo + claim.count],
                done_n_dest=self.done_n[lo:lo + claim.count])
            return
        window = self.cursor.span(lo, hi)
        ret, dn = ops.discount_return_n_step(d.reward[window], d.done[window], self.n_step,
                                             self.discount)
        self.return_[window[:-back]] = ret
        self.done_n[window[:-back]] = dn


class FrameStore:
    """One copy of every frame of a frame-stacked observation stream."""

    def __init__(self, obs_example, cursor):
        probe = obs_example.cpu() if isinstance(obs_example, torch.Tensor) else obs_example
        self.C = int(np.asarray(probe).shape[0])
        self.cursor = cursor
        self.frames = buffer_from_example(obs_example[0], (cursor.T + self.C - 1, cursor.B),
                                          device=cursor.device)
        self.newest = self.frames[self.C - 1:]          # row r: the newest frame of time r

    @property
    def guard_fwd(self):
        return self.C - 1

    def write(self, observation, claim):
        obs = on_device(observation, self.cursor.device)
        self.newest[claim.rows] = obs[:, :, -1]
        older = self.C - 1
        if claim.start == 0:                 # the very first rows of a lap: the history of row 0
            for f in range(older):
                self.frames[f] = obs[0, :, f]
        elif older and self.cursor.t <= claim.start:
            # the lap closed: its tail is the next lap's history.  INTENTIONAL deviation from the
            # reference in one corner: rlpyt/replays/frame.py:57 tests the strict ``self.t < t``, so
            # an append of exactly T rows from a non-zero start (cursor lands back on its start)
            # leaves its mirror rows stale there; ``<=`` refreshes them.  Every other append takes
            # the same branch on both sides (a sampler batch is shorter than the ring in every
            # reference config); pinned by tests/test_host_logic.py
            # ::test_replay_store_parts_on_host_tensors.  The sum tree keeps the strict rule
            # (``claim.wrapped``, replays/sum_tree.py advance).
            self.frames[:older] = self.frames[-older:]


class RnnStateStore:
    """The recurrent state an agent entered every ``interval``-th step with."""

    def __init__(self, example, cursor, interval):
        assert cursor.T % interval == 0
        self.cursor, self.interval = cursor, interval
        self.rows = cursor.T // interval
        self.data = buffer_from_example(example, (self.rows, cursor.B), device=cursor.device)

    @staticmethod
    def padded_size(size, B, interval):
        """Ring size whose T is a multiple of ``interval``."""
        return B * interval * math.ceil(math.ceil(size / B) / interval)

    def write(self, prev_rnn_state, claim):
        k, dev = self.interval, self.cursor.device
        first, last = math.ceil(claim.start / k), (claim.start + claim.count - 1) // k + 1
        rows = (slice(first, last) if last <= self.rows
                else torch.arange(first, last, device=dev) % self.rows)
        phase = (k - claim.start) % k
        state = buffer_func(prev_rnn_state, lambda x: on_device(x, dev))
        self.data[rows] = state[phase::k]
