"""Per-wave phase cycles of gemm_nt_x6_kernel's K loop (needs a -DRLPYT_TIMING build of gemm.hip).
phases: p0 barrier wait, p1 fragment reads issued, p2 MFMAs + split/stage region, p3 fetch issue"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rlpyt_amd import ops  # noqa: E402
from rlpyt_amd._lib import lib  # noqa: E402

lib.rlpyt_debug_timing_read_gemm.argtypes = [ctypes.c_void_p, ctypes.c_int]
for name, (M, N, K) in {"fwd": (8192, 512, 3456), "dgrad": (8192, 3456, 512)}.items():
    a, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
    for _ in range(3):
        ops.gemm_nt(a, b)
    torch.cuda.synchronize()
    buf = np.zeros(512 * 16 * 8, dtype=np.float32)
    assert lib.rlpyt_debug_timing_read_gemm(buf.ctypes.data, buf.size) == 0
    rows = buf.reshape(512, 16, 8)[:, :8]
    used = rows.sum(axis=(1, 2)) > 0
    per_step = rows[used] / (K / 16)
    print(f"{name}: cycles per K-16 step (last tile of each of {int(used.sum())} workgroup slots):")
    for w in (0, 3, 4, 7):
        r = per_step[:, w].mean(0)
        print(f"  wave {w}: barrier {r[0]:6.0f}  frag-issue {r[1]:5.0f}  mfma+split {r[2]:6.0f}  fetch {r[3]:5.0f}  total {r[:4].sum():6.0f}")
