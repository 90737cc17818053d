"""Optional HIP-event timing of individual kernel launches on torch's current stream.

``bench.py`` enables it over the timed region to obtain the average launch duration of
the path's kernels (BASELINE metric's ``roofline`` object); disabled it costs one branch.
The events are recorded on the SAME stream the C-ABI kernels are launched on
(``torch.cuda.current_stream()``), so the pair brackets exactly that launch.
"""
import torch

_enabled = False
_records = {}


def enable(flag=True):
    global _enabled
    _enabled = flag


def reset():
    _records.clear()


class _Region:
    __slots__ = ("name", "nbytes", "flops", "s", "e")

    def __init__(self, name, nbytes, flops=0):
        self.name, self.nbytes, self.flops = name, nbytes, flops

    def __enter__(self):
        self.s = None
        if _enabled and not torch.cuda.is_current_stream_capturing():
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *exc):
        if self.s is not None:
            self.e.record()
            _records.setdefault(self.name, []).append((self.s, self.e, self.nbytes, self.flops))
        return False


def region(name, nbytes=0, flops=0):
    return _Region(name, nbytes, flops)


def summary():
    """{name: dict(launches, avg_us, alg_bytes_per_launch, GBps[, alg_flops_per_launch,
    TFLOPs])} (synchronises)."""
    torch.cuda.synchronize()
    out = {}
    for name, recs in _records.items():
        ms = [s.elapsed_time(e) for s, e, _, _ in recs]
        nb = sum(r[2] for r in recs) / max(len(recs), 1)
        fl = sum(r[3] for r in recs) / max(len(recs), 1)
        avg = sum(ms) / len(ms) * 1e-3
        out[name] = dict(launches=len(recs), avg_us=avg * 1e6, alg_bytes_per_launch=nb,
                         GBps=(nb / avg / 1e9) if avg > 0 else 0.)
        if fl:
            out[name].update(alg_flops_per_launch=fl,
                             TFLOPs=(fl / avg / 1e12) if avg > 0 else 0.)
    return out
