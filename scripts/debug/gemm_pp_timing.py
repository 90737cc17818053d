"""Per-wave phase cycles of the ping-pong GEMM loop (needs a -DRLPYT_TIMING build of gemm_pp.hip):
load segment / barrier after it / compute segment (24 MFMAs) / barrier after it, per K-32 step."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rlpyt_amd import ops  # noqa: E402
from rlpyt_amd._lib import lib  # noqa: E402

lib.rlpyt_debug_timing_read_gemm_pp.argtypes = [ctypes.c_void_p, ctypes.c_int]
M, N, K = 8192, 512, 3456
x = torch.randn(M, K, device="cuda")
w = torch.randn(N, K, device="cuda")
g = torch.randn(M, N, device="cuda")
for name, fn, steps in (("fwd NT", lambda: ops.gemm_nt(x, w, pingpong=True), K // 32),
                        ("dgrad NN", lambda: ops.gemm_nn(g, w), N // 32)):
    # (the TN rows of profiles/r3_gemm_*_sweep*.log are of the producer / consumer TN instantiation,
    #  since replaced by gemm_tn_x6_kernel, which carries no phase counters)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); fn(); e.record(); torch.cuda.synchronize()
    buf = np.zeros(1024 * 8 * 4, dtype=np.float32)
    assert lib.rlpyt_debug_timing_read_gemm_pp(buf.ctypes.data, buf.size) == 0
    rows = buf.reshape(1024, 8, 4)
    used = rows.sum(axis=(1, 2)) > 0
    per = rows[used] / steps
    print(f"{name}: {s.elapsed_time(e) * 1e3:.0f} us (timing build); cycles per K-32 step over "
          f"{int(used.sum())} workgroup slots (last unit of each):")
    for wv in (0, 3, 4, 7):
        r = per[:, wv].mean(0)
        print(f"  wave {wv}: load {r[0]:6.0f}  bar {r[1]:6.0f}  compute {r[2]:6.0f}  bar {r[3]:6.0f}  total {r.sum():6.0f}")
