"""A/B of the round-6 GEMM variants (RLPYT_GEMM_EXP: bit 0 PAIR, bit 1 DOT2) on the three trunk GEMMs of
the PPO update at M = 8192, interleaved (every round visits every variant) so that box-to-box and
clock-state differences cancel; HIP events around 20 back-to-back launches; one JSON line per round."""
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
    return round(s.elapsed_time(e) / iters * 1e3, 1)


M, N, K = 8192, 512, 3456
x = torch.randn(M, K, device="cuda")
w = torch.randn(N, K, device="cuda") * 0.02
g = torch.randn(M, N, device="cuda")
wt = w.t().contiguous()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for r in range(rounds):
    row = {"round": r}
    for exp in (0, 1, 2, 3):
        os.environ["RLPYT_GEMM_EXP"] = str(exp)
        row[f"fwd_exp{exp}"] = timeit(lambda: ops.gemm_nt(x, w))
        row[f"dgrad_exp{exp}"] = timeit(lambda: ops.gemm_nt(g, wt))
        if exp in (0, 2):
            row[f"wgrad_exp{exp}"] = timeit(lambda: ops.gemm_tn(g, x))
    print(json.dumps(row), flush=True)
os.environ.pop("RLPYT_GEMM_EXP", None)
