"""Latency / bandwidth probe of the sampler's hand-off primitives on the GPU box: H2D copies from
hipHostRegister-ed fork-shared memory vs hipHostMalloc memory, kernels reading page-locked host
memory in place, graph launch round trips.  Prints one JSON object."""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlpyt_amd import _lib  # noqa: E402
from rlpyt_amd.utils.buffer import np_mp_array  # noqa: E402


def spin_until(ev):
    while not ev.query():
        pass


def timed(fn, stream, n=200):
    """Mean / min wall us of: call fn() on `stream`, record an event, poll it to completion."""
    ev = torch.cuda.Event()
    ts = []
    for i in range(n + 20):
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            fn()
            ev.record()
        t1 = time.perf_counter()
        spin_until(ev)
        t2 = time.perf_counter()
        if i >= 20:
            ts.append((t1 - t0, t2 - t0))
    a = np.array(ts) * 1e6
    return dict(issue_us=round(float(a[:, 0].mean()), 1), total_us=round(float(a[:, 1].mean()), 1),
                total_min_us=round(float(a[:, 1].min()), 1))


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    st = torch.cuda.Stream()
    out = {}
    for nbytes in (1024, 64 * 8320, 64 * 8320 + 1040, 4 * 64 * 8320):
        shared = np_mp_array(nbytes, np.uint8)
        shared[:] = 7
        assert _lib.lib.rlpyt_host_register(ctypes.c_void_p(shared.ctypes.data), int(shared.nbytes)) == 0
        h_reg = torch.from_numpy(shared)
        h_pin = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        d = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        mapped = _lib.host_mapped_tensor(shared, dev)
        key = f"{nbytes}B"
        out[key] = dict(
            h2d_registered=timed(lambda: d.copy_(h_reg, non_blocking=True), st),
            h2d_hostmalloc=timed(lambda: d.copy_(h_pin, non_blocking=True), st),
            kernel_reads_host=timed(lambda: d.copy_(mapped), st),
            d2d=timed(lambda: d.copy_(d.clone()), st))
    # graph with one tiny kernel / three tiny kernels
    x = torch.zeros(64, device=dev)
    for k in (1, 3):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(st):
            for _ in range(3):
                x.add_(1)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=st):
            for _ in range(k):
                x.add_(1)
        out[f"graph_{k}_tiny_kernels"] = timed(lambda: g.replay(), st)
    out["eager_tiny_kernel"] = timed(lambda: x.add_(1), st)
    # copy + graph, as the step does it
    shared = np_mp_array(64 * 8320 + 1040, np.uint8)
    assert _lib.lib.rlpyt_host_register(ctypes.c_void_p(shared.ctypes.data), int(shared.nbytes)) == 0
    h = torch.from_numpy(shared)
    d = torch.empty_like(h, device=dev)

    def step():
        d.copy_(h, non_blocking=True)
        g.replay()
    out["h2d_then_graph"] = timed(step, st)
    # host-visible completion flag written by a kernel vs the event
    flag_np = np_mp_array(16, np.int64)
    assert _lib.lib.rlpyt_host_register(ctypes.c_void_p(flag_np.ctypes.data), int(flag_np.nbytes)) == 0
    flag = _lib.host_mapped_tensor(flag_np, dev)
    one = torch.ones(16, dtype=torch.int64, device=dev)
    ts = []
    for i in range(220):
        want = int(flag_np[0]) + 1
        t0 = time.perf_counter()
        with torch.cuda.stream(st):
            flag.add_(one)
        while flag_np[0] < want:
            pass
        t2 = time.perf_counter()
        if i >= 20:
            ts.append(t2 - t0)
    out["kernel_writes_host_flag_total_us"] = round(float(np.mean(ts) * 1e6), 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
