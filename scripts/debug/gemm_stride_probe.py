"""Debug (GPU box): does the trunk GEMM's time per K-step depend on the operands' row stride (L2 channel
spread of the 128 x 64-byte row segments a workgroup fetches per step)?  gemm_nt [8192, K] x [512, K]^T and
the input-gradient shape [8192, K] x [3456, K]^T for K around 3456 / 512; ns per K-16 step."""
import sys

import torch

sys.path.insert(0, ".")
from rlpyt_amd import ops  # noqa: E402
sys.path.insert(0, "scripts")
from gemm_bench import timeit  # noqa: E402

for M, N, Ks in ((8192, 512, (3328, 3456, 3488, 3520, 3584, 3616, 3648)),
                 (8192, 3456, (448, 480, 512, 544, 576, 608))):
    for K in Ks:
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") * 0.02
        us = min(timeit(lambda: ops.gemm_nt(x, w), iters=20) for _ in range(3))
        print(f"M={M} N={N} K={K} (row stride {K * 4} B = {K * 4 / 256:.2f} x 256): {us:.1f} us, "
              f"{us * 1e3 / (K / 16):.1f} ns per K-16 step, {2 * M * N * K / us / 1e6:.1f} TFLOP/s")
