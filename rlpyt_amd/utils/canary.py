"""Guard bands around every device buffer this package allocates (debug mode, RLPYT_CANARY=1).

Parity tests check VALUES; a from-scratch kernel that writes a few bytes past its output, or reads
past its input, passes them as long as nobody looks at the neighbouring bytes.  With the canary on,
``torch.empty / zeros / empty_like / zeros_like`` for CUDA tensors (every kernel-facing allocation of
this package goes through them: ops.py outputs and workspaces, the sampler's HBM batch and staging
buffers, replay rings, optimizer scratch) return the interior of a larger allocation whose first and
last ``GUARD`` bytes hold 0xFF:

* an out-of-bounds WRITE of up to ``GUARD`` bytes lands in a guard and ``check()`` reports the buffer
  (shape, dtype, where it was allocated);
* an out-of-bounds float READ picks up 0xFFFFFFFF = NaN (int64: -1, uint8: 255), which the parity
  tests' comparisons then trip over.

``tests/conftest.py`` switches it on for the whole GPU suite when RLPYT_CANARY=1 and calls ``check()``
after every test; ``bench.py`` does the same after its timed region.  Not for production runs: every
allocation pays two fill launches."""
import os
import traceback
import weakref

import torch

GUARD = 4096
_orig = {}
_live = []          # [(weakref to the tensor handed out, padded uint8 tensor, nbytes, description)]
_stats = dict(allocs=0, checks=0)


def enabled():
    return bool(_orig)


def _site():
    for fr in reversed(traceback.extract_stack(limit=8)[:-3]):
        if "canary.py" not in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
    return "?"


def _padded(shape, dtype, device, zero):
    dtype = dtype or torch.get_default_dtype()
    if isinstance(shape, int):
        shape = (shape,)
    shape = tuple(int(s) for s in shape)
    n = 1
    for s in shape:
        n *= s
    nbytes = n * torch.empty((), dtype=dtype).element_size()
    pad = (-nbytes) % 16
    raw = _orig["empty"](GUARD + nbytes + pad + GUARD, dtype=torch.uint8, device=device)
    raw[:GUARD] = 0xFF
    raw[GUARD + nbytes:] = 0xFF
    inner = raw[GUARD:GUARD + nbytes].view(dtype).reshape(shape)
    if zero:
        inner.zero_()
    # the padded tensor is held HERE (a weak reference to it dies at once: a view keeps the storage
    # alive, not the base's Python object) and let go when the tensor handed out has died
    _live.append((weakref.ref(inner), raw, nbytes, f"{tuple(shape)} {dtype} @ {_site()}"))
    _stats["allocs"] += 1
    if _stats["allocs"] % 256 == 0:
        _live[:] = [e for e in _live if e[0]() is not None]
    return inner


def _is_cuda(device):
    return device is not None and torch.device(device).type == "cuda"


def _shape_of(args, kwargs):
    if "size" in kwargs:
        return kwargs["size"]
    if len(args) == 1 and isinstance(args[0], (tuple, list, torch.Size)):
        return args[0]
    return args


def _wrap_new(name, zero):
    def fn(*args, **kwargs):
        dev = kwargs.get("device")
        plain = set(kwargs) <= {"size", "dtype", "device", "requires_grad"}
        if not (_is_cuda(dev) and plain and not kwargs.get("requires_grad")):
            return _orig[name](*args, **kwargs)
        return _padded(_shape_of(args, kwargs), kwargs.get("dtype"), dev, zero)
    return fn


def _wrap_like(name, zero):
    def fn(x, **kwargs):
        dev = kwargs.get("device", x.device)
        ok = (set(kwargs) <= {"dtype", "device", "memory_format"} and x.is_contiguous()
              and kwargs.get("memory_format") in (None, torch.preserve_format,
                                                  torch.contiguous_format))
        if not (_is_cuda(dev) and ok):
            return _orig[name](x, **kwargs)
        return _padded(x.shape, kwargs.get("dtype", x.dtype), dev, zero)
    return fn


def enable():
    if _orig:
        return
    for name in ("empty", "zeros", "empty_like", "zeros_like"):
        _orig[name] = getattr(torch, name)
    torch.empty, torch.zeros = _wrap_new("empty", False), _wrap_new("zeros", True)
    torch.empty_like, torch.zeros_like = _wrap_like("empty_like", False), _wrap_like("zeros_like", True)


def disable():
    for name, fn in _orig.items():
        setattr(torch, name, fn)
    _orig.clear()
    _live.clear()


def check(what=""):
    """Synchronise and verify every live guard band; raises AssertionError naming the buffers whose
    guards were written.  Returns the number of buffers checked."""
    if not _orig:
        return 0
    torch.cuda.synchronize()
    bad, keep, n = [], [], 0
    for ref, raw, nbytes, desc in _live:
        if ref() is None:
            continue
        keep.append((ref, raw, nbytes, desc))
        n += 1
        head, tail = raw[:GUARD], raw[GUARD + nbytes:]
        if not (bool((head == 0xFF).all()) and bool((tail == 0xFF).all())):
            where = []
            for nm, g in (("before", head), ("after", tail)):
                idx = torch.nonzero(g != 0xFF).reshape(-1)
                if idx.numel():
                    where.append(f"{idx.numel()} bytes {nm} (first at offset {int(idx[0])}"
                                 f"{' from the end of the buffer' if nm == 'after' else ' of the guard'})")
            bad.append(f"{desc}: {', '.join(where)}")
            head.fill_(0xFF)
            tail.fill_(0xFF)
    _live[:] = keep
    _stats["checks"] += 1
    assert not bad, f"canary {what}: out-of-bounds device writes next to\n  " + "\n  ".join(bad)
    return n


def guards_of(tensor):
    """(guard before, guard after) of a tensor handed out by the canary -- uint8 views, for tests."""
    for ref, raw, nbytes, _ in _live:
        if ref() is tensor:
            return raw[:GUARD], raw[GUARD + nbytes:]
    raise KeyError("not a live canary allocation")


def stats():
    return dict(_stats, live=len(_live))
