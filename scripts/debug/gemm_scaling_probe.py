"""Debug (GPU box): does the forward trunk GEMM's time follow the per-CU work or the number of busy CUs?
gemm_nt at K = 3456, N = 512 for M = 1024 .. 16384 (M = 8192: one 128 x 128 tile per CU)."""
import sys

import torch

sys.path.insert(0, ".")
from rlpyt_amd import ops  # noqa: E402
sys.path.insert(0, "scripts")
from gemm_bench import timeit  # noqa: E402

N, K = 512, 3456
w = torch.randn(N, K, device="cuda") * 0.02
for M in (256, 1024, 2048, 4096, 8192, 16384):
    x = torch.randn(M, K, device="cuda")
    us = timeit(lambda: ops.gemm_nt(x, w), iters=30)
    tiles = (M // 128) * 4
    print(f"M={M}: {us:.1f} us, {tiles} tiles, {2 * M * N * K / us / 1e6:.1f} TFLOP/s alg")
