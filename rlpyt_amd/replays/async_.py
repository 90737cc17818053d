"""Replay buffers for asynchronous sampling / optimisation (rlpyt/replays/async_.py:8-47,
rlpyt/replays/sum_tree.py:225-249, rlpyt/utils/synchronize.py:5-36), re-thought for a ring that
lives in HBM.

The reference puts the replay ring, the sum tree and a "universal cursor" into OS shared memory so
that a sampler-side process (the memory copier, rlpyt/runners/async_rl.py:574-610) can append while
optimizer processes draw batches, all under a multiple-reader / single-writer lock.  Here the ring,
the frame store and the f64 tree are device arrays already -- every thread of the process (and every
stream) sees the same bytes -- so what is left of "async" is ORDER:

* between host threads: ``RWLock``, the reference's lock on ``threading`` primitives (the sampler
  side and the optimizer side are threads of ONE process per GPU: the C serve loop, the env workers'
  hand-off and every kernel launch release the GIL); the cursor (``RingCursor``, the tree's own
  cursor in its C handle) is shared state under that lock -- the reference's ``_async_pull`` /
  ``_async_push`` of ``async_t`` have nothing left to carry;
* between HIP streams: a writer records an event behind its kernels and every later reader makes its
  stream wait for it (and the other way round), so an append issued on the sampler thread's stream
  and a batch gather issued on the optimizer thread's stream touch the ring in lock order even when
  the two threads use different streams.

``AsyncSumTree`` is the device tree with the reference's contract made checkable: "assumes that
writing to tree values is lock protected elsewhere, i.e. by the replay buffer" -- its mutators assert
that the calling thread holds the buffer's write lock.
"""
import threading

import torch

from .. import ops
from . import buffers as B


class RWLock:
    """Multiple simultaneous readers, one writer (rlpyt/utils/synchronize.py:5-36) -- for threads."""

    def __init__(self):
        self.write_lock = threading.Lock()
        self._read_lock = threading.Lock()
        self._read_count = 0
        self._writer = None                  # ident of the thread holding the write lock as a WRITER

    def __enter__(self):
        self.acquire_read()

    def __exit__(self, *args):
        self.release_read()

    def acquire_write(self):
        self.write_lock.acquire()
        self._writer = threading.get_ident()

    def release_write(self):
        self._writer = None
        self.write_lock.release()

    def acquire_read(self):
        with self._read_lock:
            self._read_count += 1
            if self._read_count == 1:
                self.write_lock.acquire()

    def release_read(self):
        with self._read_lock:
            self._read_count -= 1
            if self._read_count == 0:
                self.write_lock.release()

    def held_for_writing(self):
        return self._writer == threading.get_ident()

    class _Writing:
        def __init__(self, lock):
            self.lock = lock

        def __enter__(self):
            self.lock.acquire_write()

        def __exit__(self, *a):
            self.lock.release_write()

    def writing(self):
        """``with lock.writing():`` -- the write side as a context (the reference uses
        ``with rw_lock.write_lock:``; here the holder is also remembered for ``held_for_writing``)."""
        return RWLock._Writing(self)


class AsyncSumTree(ops.DeviceSumTree):
    """Device sum tree whose mutators insist on the replay buffer's write lock
    (rlpyt/replays/sum_tree.py:225-249: shared-memory tree, asynchronous cursor, "writing ... is lock
    protected elsewhere").  The tree array and its cursor are one object in this process; ``guard``
    is set by the buffer that owns the tree."""
    async_ = True
    guard = None

    def _check(self, what):
        g = self.guard
        assert g is None or g.held_for_writing(), \
            f"AsyncSumTree.{what} outside the replay buffer's write lock"

    def reset(self):
        self._check("reset")
        return super().reset()

    def advance(self, *args, **kwargs):
        self._check("advance")
        return super().advance(*args, **kwargs)

    def update_batch_priorities(self, *args, **kwargs):
        self._check("update_batch_priorities")
        return super().update_batch_priorities(*args, **kwargs)

    def leaf_values(self, *args, **kwargs):
        self._check("leaf_values")                 # (re-declares the sampled set: a write)
        return super().leaf_values(*args, **kwargs)


class AsyncReplayBufferMixin:
    """``append_samples`` / ``update_batch_priorities`` under the write lock, ``sample_batch`` under the
    read lock (rlpyt/replays/async_.py:24-39), device work chained through events (module docstring)."""
    async_ = True
    TREE_CLS = AsyncSumTree

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.rw_lock = RWLock()
        self._w_event = None                       # behind the last writer's kernels
        self._r_events = {}                        # thread ident -> behind that reader's last kernels
        self._ev_lock = threading.Lock()
        tree = getattr(getattr(self, "draws", None), "tree", None)
        if isinstance(tree, AsyncSumTree):
            tree.guard = self.rw_lock
            self.draws.guard_stale = True          # write-backs skip rows appended over meanwhile

    # ---- stream order --------------------------------------------------------------------------
    def _cuda(self):
        return self.device.type == "cuda"

    def _wait_for(self, events):
        if self._cuda():
            s = torch.cuda.current_stream(self.device)
            for e in events:
                if e is not None:
                    s.wait_event(e)

    def _mark(self):
        if not self._cuda():
            return None
        e = torch.cuda.Event()
        e.record(torch.cuda.current_stream(self.device))
        return e

    def can_sample_on_device(self):
        """No captured update graphs over an asynchronous buffer: the event chaining above cannot sit
        inside a capture, and the ring changes between replays."""
        return False

    # ---- protocol ------------------------------------------------------------------------------
    def append_samples(self, *args, **kwargs):
        with self.rw_lock.writing():
            with self._ev_lock:
                pending = [self._w_event] + list(self._r_events.values())
            self._wait_for(pending)                # every earlier reader's gathers, the last write
            ret = super().append_samples(*args, **kwargs)
            self._w_event = self._mark()
        return ret

    def sample_batch(self, *args, **kwargs):
        with self.rw_lock:                         # read lock
            self._wait_for([self._w_event])
            out = super().sample_batch(*args, **kwargs)
            e = self._mark()
            with self._ev_lock:
                self._r_events[threading.get_ident()] = e
        return out

    def sample_batch_device(self, *args, **kwargs):
        with self.rw_lock:
            self._wait_for([self._w_event])
            out = super().sample_batch_device(*args, **kwargs)
            e = self._mark()
            with self._ev_lock:
                self._r_events[threading.get_ident()] = e
        return out

    def update_batch_priorities(self, *args, **kwargs):
        with self.rw_lock.writing():
            with self._ev_lock:
                pending = [self._w_event] + list(self._r_events.values())
            self._wait_for(pending)
            ret = super().update_batch_priorities(*args, **kwargs)
            self._w_event = self._mark()
        return ret


# ---- the reference's class names (rlpyt/replays/{non_sequence,sequence}/*.py) ---------------------
class AsyncUniformReplayBuffer(AsyncReplayBufferMixin, B.UniformReplayBuffer):
    pass


class AsyncPrioritizedReplayBuffer(AsyncReplayBufferMixin, B.PrioritizedReplayBuffer):
    pass


class AsyncUniformReplayFrameBuffer(AsyncReplayBufferMixin, B.UniformReplayFrameBuffer):
    pass


class AsyncPrioritizedReplayFrameBuffer(AsyncReplayBufferMixin, B.PrioritizedReplayFrameBuffer):
    pass


class AsyncUniformSequenceReplayBuffer(AsyncReplayBufferMixin, B.UniformSequenceReplayBuffer):
    pass


class AsyncPrioritizedSequenceReplayBuffer(AsyncReplayBufferMixin, B.PrioritizedSequenceReplayBuffer):
    pass


class AsyncUniformSequenceReplayFrameBuffer(AsyncReplayBufferMixin, B.UniformSequenceReplayFrameBuffer):
    pass


class AsyncPrioritizedSequenceReplayFrameBuffer(AsyncReplayBufferMixin,
                                                B.PrioritizedSequenceReplayFrameBuffer):
    pass


_ASYNC = {(c.FRAMES, c.SEQUENCE, c.PRIORITIZED): c for c in (
    AsyncUniformReplayBuffer, AsyncUniformReplayFrameBuffer, AsyncUniformSequenceReplayBuffer,
    AsyncUniformSequenceReplayFrameBuffer, AsyncPrioritizedReplayBuffer,
    AsyncPrioritizedReplayFrameBuffer, AsyncPrioritizedSequenceReplayBuffer,
    AsyncPrioritizedSequenceReplayFrameBuffer)}


def async_replay_class(frames, sequence, prioritized):
    """The asynchronous buffer class for a combination of the three switches."""
    return _ASYNC[(bool(frames), bool(sequence), bool(prioritized))]
