"""Which ring positions a training batch is made of: uniform draws and sum-tree priority draws, for
single steps and for sequences.

Index streams must equal the reference's on the same ``np.random`` state, so the ARITHMETIC of the
draws is its (rlpyt/replays/non_sequence/uniform.py:17-28, sequence/uniform.py:24-39,
non_sequence/prioritized.py:43-79, sequence/prioritized.py:36-115, replays/sum_tree.py); what is
different is where it runs -- the tree is an f64 array in HBM (``ops.DeviceSumTree``: descent, leaf
write and diff propagation are kernels) and only the ``n`` uniforms of a batch come from the host
RNG -- and how it is organised: one draw object per buffer, parameterised by a ``stride`` (1, or the
RNN-state interval of a sequence buffer: sequences start on stored states) and a ``reach`` (how
many rows a drawn start position must keep clear of the cursor)."""
import math

import numpy as np
import torch

from .. import ops


class UniformDraw:
    """Uniform start positions outside the guard band around the cursor."""

    def __init__(self, cursor, stride=1, reach=0, sequence=False):
        self.cursor, self.stride, self.reach, self.sequence = cursor, max(stride, 0), reach, sequence

    def on_append(self, claim, t_after, priorities=None):
        pass

    def draw(self, n, reach=None):
        c = self.cursor
        back = c.guard_back + (self.reach if reach is None else reach)
        fwd = c.guard_fwd
        if self.sequence:       # (a not-yet-full sequence ring also keeps the forward band clear)
            high = c.T - back - fwd if c.full else c.t - back - fwd
            low = 0
        else:
            high = c.T - back - fwd if c.full else c.t - back
            low = 0 if c.full else fwd
        T_idxs = np.random.randint(low=low, high=high, size=(n,))
        T_idxs[T_idxs >= c.t - back] += min(c.t, back) + fwd     # hop over the band
        if self.sequence and self.stride > 0:
            T_idxs = (T_idxs // self.stride) * self.stride
        B_idxs = np.random.randint(low=0, high=c.B, size=(n,))
        return T_idxs, B_idxs, None


class PriorityDraw:
    """Proportional draws from the device sum tree + importance weights.

    ``stride > 1``: one leaf per ``stride`` ring rows (sequence buffers: a leaf is a stored RNN
    state, i.e. a possible sequence start).  ``weight_eps``: the reference adds 1e-6 to the
    priorities of single-step batches before inverting them and nothing for sequences."""

    def __init__(self, cursor, alpha, beta, default_priority, input_priorities,
                 input_priority_shift, unique=False, stride=1, reach=0, sequence=False,
                 tree_cls=None):
        self.cursor, self.alpha, self.beta, self.unique = cursor, alpha, beta, bool(unique)
        self.stride, self.sequence = max(1, stride), sequence
        self.weight_eps = 0. if sequence else 1e-6
        self.input_alpha = not sequence     # fresh priorities ** alpha: single-step buffers only
        k = self.stride
        back = (math.ceil((1 + cursor.guard_back + reach) / k) if sequence else cursor.guard_back)
        fwd = math.ceil(cursor.guard_fwd / k) if sequence else cursor.guard_fwd
        self.band = (back, fwd)             # guard band around the tree's cursor, in leaf rows
        self.guard_stale = False            # asynchronous buffers: see ``update``
        self._appended = 0                  # leaf rows advanced so far
        self._drawn = None                  # (leaf T_idxs, B_idxs, tree cursor, _appended) of the last draw
        self.tree = (tree_cls or ops.DeviceSumTree)(
            T=cursor.T // k, B=cursor.B, off_backward=back, off_forward=fwd,
            default_value=default_priority ** alpha, enable_input_priorities=input_priorities,
            input_priority_shift=input_priority_shift, device=cursor.device)

    def on_append(self, claim, t_after, priorities=None):
        """Advance the tree past the rows just written (``priorities``: fresh input priorities of
        those rows, or None for the default)."""
        dev = self.cursor.device
        if priorities is not None:
            priorities = torch.as_tensor(priorities, device=dev).double()
            if self.input_alpha:
                priorities = priorities ** self.alpha
        k = self.stride
        if k == 1:
            self._appended += claim.count
            self.tree.advance(claim.count, priorities=priorities)
            return
        if priorities is not None and priorities.dim() == 2:      # per-step -> per stored state
            priorities = priorities[(k - claim.start) % k::k].contiguous()
        leaves = t_after // k - claim.start // k
        if claim.wrapped:
            leaves += self.cursor.T // k
        self._appended += leaves
        self.tree.advance(leaves, priorities=priorities)

    def draw(self, n, reach=None):
        if self.unique:
            T_idxs, B_idxs, pri = self.tree.sample_unique(int(n))
        else:
            u = torch.from_numpy(np.random.rand(int(n))).to(self.cursor.device, non_blocking=True)
            T_idxs, B_idxs, pri = self.tree.sample(u)
        if self.guard_stale:
            self._drawn = (T_idxs, B_idxs, self.tree.t, self._appended)
        if self.stride > 1:
            T_idxs = T_idxs * self.stride
        return T_idxs, B_idxs, self._is_weights(pri, self.beta)

    def _is_weights(self, pri, beta):
        """``(1 / (p + eps)) ** beta`` over its maximum, float64 inside, as float32
        (rlpyt/replays/non_sequence/prioritized.py:52-56) -- one launch on the device."""
        if pri.is_cuda and pri.dtype == torch.float64:
            return ops.is_weights(pri.contiguous(), self.weight_eps, beta)
        if isinstance(beta, torch.Tensor):
            beta = beta.to(pri.dtype)
        w = torch.pow(1. / (pri + self.weight_eps), beta)
        return (w / w.max()).float()

    def draw_device(self, uniforms, beta):
        """``draw`` from device-resident uniforms (f64 ``[n]``) with the importance exponent as a
        device scalar: no host work, capturable."""
        T_idxs, B_idxs, pri = self.tree.sample(uniforms)
        if self.guard_stale:
            self._drawn = (T_idxs, B_idxs, self.tree.t, self._appended)
        if self.stride > 1:
            T_idxs = T_idxs * self.stride
        return T_idxs, B_idxs, self._is_weights(pri, beta)

    def update(self, priorities):
        """New priorities of the last drawn batch: ``** alpha`` in the caller's dtype (as numpy does
        in the reference), then the f64 tree update.

        ``guard_stale`` (set by asynchronous buffers, replays/async_.py): rows may have been appended
        between the draw and this write-back.  The reference writes the batch's priorities regardless
        (rlpyt/replays/non_sequence/prioritized.py:60-68 under the write lock,
        replays/sum_tree.py:131-139) -- a leaf that meanwhile entered the guard band around the new
        cursor becomes drawable again although its frame stack / n-step return now straddles two
        laps, and a row rewritten meanwhile gets the old row's TD error.  Here such leaves keep the
        value the tree holds for them now (zero inside the band, the fresh priority for rewritten
        rows): a stated deviation that only exists where the reference's order is a race."""
        new = priorities.detach().to(self.cursor.device) ** self.alpha
        if self.guard_stale and self._drawn is not None and self._drawn[3] != self._appended:
            T_leaf, B_idxs, t_draw, n_draw = self._drawn
            n, T = self._appended - n_draw, self.tree.T
            back, fwd = self.band
            t_now = self.tree.t
            rewritten = ((T_leaf - t_draw) % T) < min(n, T)
            in_band = ((T_leaf - (t_now - back)) % T) < back + fwd
            current = self.tree.leaf_values(T_leaf * self.tree.B + B_idxs)
            new = torch.where(rewritten | in_band, current.to(new.dtype), new)
        self.tree.update_batch_priorities(new)
