"""rlpyt_gemm_nt_f32 (bf16x6) beside torch / hipBLASLt f32 at the two trunk shapes of the update."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlpyt_amd import ops  # noqa: E402


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


res = {}
for name, (M, N, K) in {"fwd": (8192, 512, 3456), "dgrad": (8192, 3456, 512)}.items():
    a = torch.randn(M, K, device="cuda")
    b = torch.randn(N, K, device="cuda")
    us = timeit(lambda: ops.gemm_nt(a, b))
    ut = timeit(lambda: torch.mm(a, b.t()))
    fl = 2 * M * N * K
    res[name] = {"x6_us": round(us, 1), "torch_f32_us": round(ut, 1),
                 "x6_alg_TFLOPs": round(fl / us / 1e6, 1), "x6_issued_frac_bf16_peak": round(6 * fl / us / 1e6 / 2500, 3)}
print(json.dumps(res))
