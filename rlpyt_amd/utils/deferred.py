"""Diagnostics whose device -> host copy is still in flight.

``optimize_agent`` of the reference ends every minibatch / update with ``.item()`` calls
(rlpyt/algos/pg/ppo.py:106-109, rlpyt/algos/dqn/dqn.py:183-186): the host waits for the device, and
the device then waits for the host to get through the glue of the next sampling phase.  The algorithms
here keep the per-update scalars on the device and read them back ONCE per call -- and that one
read-back does not have to block either: ``PendingOptInfo`` starts the copy into pinned host memory,
records an event and hands the runner an object that LOOKS like the ``OptInfo`` namedtuple (field
attributes, ``_fields``, iteration, indexing, ``_asdict``); the first access waits for the event and
builds the lists.  A runner that stores diagnostics every iteration (the reference's
``store_diagnostics``: ``getattr(opt_info, k, [])``) sees exactly the values and the timing it always
saw; one that only touches them when it logs (``runners/minibatch_rl.py``, ``bench.py``) lets the host
run ahead into the next iteration's ``sample_mode`` / sampler set-up while the updates finish.
"""
import os

import torch

# RLPYT_LEAN_HOST=0: read back inside optimize_agent, redo the per-phase work on every mode call (A/B runs)
LEAN_HOST = os.environ.get("RLPYT_LEAN_HOST", "1") != "0"


def start_host_copy(t):
    """Device tensor -> pinned host tensor, asynchronous on the current stream (host tensors are
    returned as they are)."""
    if not t.is_cuda:
        return t
    host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    host.copy_(t, non_blocking=True)
    return host


class PendingOptInfo:
    """``build(host_tensors) -> OptInfoCls instance``, evaluated once the copies have landed."""

    __slots__ = ("_cls", "_fields", "_host", "_event", "_build", "_value")

    def __init__(self, OptInfoCls, device_tensors, build):
        self._cls, self._fields = OptInfoCls, tuple(OptInfoCls._fields)
        self._host = [start_host_copy(t) for t in device_tensors]
        self._event = None
        if any(t.is_cuda for t in device_tensors):
            self._event = torch.cuda.Event()
            self._event.record()
        self._build, self._value = build, None
        if not LEAN_HOST:
            self.resolve()

    @property
    def pending(self):
        return self._value is None and self._event is not None and not self._event.query()

    def resolve(self):
        if self._value is None:
            if self._event is not None:
                self._event.synchronize()
            self._value = self._build(self._host)
            self._host = self._build = None
        return self._value

    def __getattr__(self, name):
        # (only reached for names that are not slots)
        if name in object.__getattribute__(self, "_fields"):
            return getattr(self.resolve(), name)
        raise AttributeError(name)

    def __iter__(self):
        return iter(self.resolve())

    def __len__(self):
        return len(self._fields)

    def __getitem__(self, i):
        return self.resolve()[i]

    def _asdict(self):
        return self.resolve()._asdict()

    def __repr__(self):
        return repr(self.resolve()) if not self.pending else f"<pending {self._cls.__name__}>"


def resolve(opt_info):
    """The plain namedtuple behind ``opt_info`` (itself when it is one already)."""
    return opt_info.resolve() if isinstance(opt_info, PendingOptInfo) else opt_info
