"""Buffer helpers over (nested) namedarraytuples of arrays.

Contract restated from rlpyt/utils/buffer.py:11-205: leaves are numpy
arrays or torch tensors; structure = nested (named)tuples; ``None`` leaves pass through.
MI355X additions: ``buffer_from_example(..., device=...)`` allocates the leaves directly in
HBM, ``pinned=True`` in page-locked host memory (for async H2D of step buffers).
"""
import ctypes
import multiprocessing as mp

import numpy as np
import torch

from .collections import is_namedarraytuple, is_namedtuple, namedarraytuple_like


def _map(fn, buf, *rest):
    """Apply fn to every leaf (None passes through), preserving (named)tuple structure."""
    if buf is None:
        return None
    if isinstance(buf, tuple):
        vals = [_map(fn, b, *(tuple.__getitem__(r, i) if isinstance(r, tuple) else r for r in rest))
                for i, b in enumerate(buf)]
        return type(buf)(*vals) if is_namedtuple(buf) else type(buf)(vals)
    return fn(buf, *rest)


def np_mp_array(shape, dtype):
    """numpy array backed by fork-shared memory (rlpyt/utils/buffer.py:55-62)."""
    shape = (shape,) if isinstance(shape, int) else tuple(shape)
    size = int(np.prod(shape))
    nbytes = max(size * np.dtype(dtype).itemsize, 1)
    # 64-byte aligned (multiprocessing's heap only guarantees 8): kernels that read a page-locked
    # step buffer in place use 16-byte lanes, and a cache-line-aligned start keeps neighbouring
    # hand-off words of different arrays off one line
    raw = mp.RawArray(ctypes.c_char, nbytes + 64)
    off = (-ctypes.addressof(raw)) % 64
    return np.frombuffer(raw, dtype=dtype, count=size, offset=off).reshape(shape)


def buffer_from_example(example, leading_dims, share_memory=False, device=None, pinned=False):
    """Zero-filled buffer with ``leading_dims`` prepended to every leaf of ``example``.

    Default: numpy leaves (optionally fork-shared).  ``device``: torch leaves in HBM.
    ``pinned``: torch CPU leaves in page-locked memory."""
    if example is None:
        return None
    if isinstance(leading_dims, int):
        leading_dims = (leading_dims,)
    if is_namedtuple(example):
        cls = namedarraytuple_like(example)
        return cls(*(buffer_from_example(v, leading_dims, share_memory, device, pinned)
                     for v in example))
    if isinstance(example, torch.Tensor):
        example = example.detach().cpu().numpy()
    ex = np.asarray(example)
    if ex.dtype == np.object_:
        raise TypeError("Buffer example value cannot cast as np.dtype==object.")
    shape = tuple(leading_dims) + ex.shape
    if device is not None or pinned:
        tdtype = torch.from_numpy(np.zeros((), dtype=ex.dtype)).dtype
        if device is not None:
            return torch.zeros(shape, dtype=tdtype, device=device)
        t = torch.zeros(shape, dtype=tdtype)
        return t.pin_memory() if torch.cuda.is_available() else t
    if share_memory:
        arr = np_mp_array(shape, ex.dtype)
        arr[...] = 0
        return arr
    return np.zeros(shape, dtype=ex.dtype)


def torchify_buffer(buf):
    """numpy leaves -> zero-copy torch views (writes visible both ways)."""
    def f(x):
        if isinstance(x, np.ndarray):
            return torch.from_numpy(x)
        if isinstance(x, torch.Tensor):
            return x
        return torch.from_numpy(np.asarray(x))
    return _map(f, buf)


def numpify_buffer(buf):
    def f(x):
        if isinstance(x, torch.Tensor):
            return x.detach().cpu().numpy()
        return x
    return _map(f, buf)


def buffer_to(buf, device=None, non_blocking=False):
    """Move every torch leaf to ``device`` (numpy leaves raise, as in the reference)."""
    def f(x):
        if isinstance(x, np.ndarray):
            raise TypeError("Cannot move numpy array to device.")
        return x.to(device, non_blocking=non_blocking)
    return _map(f, buf)


def buffer_method(buf, method_name, *args, **kwargs):
    return _map(lambda x: getattr(x, method_name)(*args, **kwargs), buf)


def buffer_func(buf, func, *args, **kwargs):
    return _map(lambda x: func(x, *args, **kwargs), buf)


def buffer_pairs(buf, other):
    """``(leaf, matching leaf of other)`` for every non-None leaf of ``buf`` (same structure)."""
    out = []
    _map(lambda a, b: out.append((a, b)), buf, other)
    return out


def get_leading_dims(buf, n_dim=1):
    """Leading dims of the first leaf; asserts all leaves agree."""
    leaves = []

    def collect(x):
        leaves.append(tuple(x.shape[:n_dim]))
        return x
    _map(collect, buf)
    if not leaves:
        raise ValueError("Empty buffer.")
    if any(l != leaves[0] for l in leaves):
        raise ValueError(f"Found mismatched leading dimensions: {leaves}")
    return leaves[0]


def buffer_leaves(buf):
    out = []
    _map(lambda x: out.append(x) or x, buf)
    return out
