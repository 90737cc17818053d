"""The three trunk GEMMs of the PPO update at M = 8192 (forward x W^T, input gradient g W, weight
gradient g^T x) on the bf16x6 kernels, beside torch / hipBLASLt f32 on the same operands.
Ceiling of a bf16x6 GEMM: 6 bf16 MFMAs per MAC -> 2.5 PFLOP/s / 6 = 417 TFLOP/s algorithmic."""
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


def entry(us, fl):
    return {"us": round(us, 1), "alg_TFLOPs": round(fl / us / 1e6, 1),
            "frac_of_bf16x6_ceiling": round(6 * fl / us / 1e6 / 2500, 3)}


def main():
    M, N, K = 8192, 512, 3456          # batch, trunk width, conv features
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.02
    g = torch.randn(M, N, device="cuda")
    fl = 2 * M * N * K
    res = {"shape": {"M": M, "N": N, "K": K, "GFLOP": fl / 1e9}}
    res["fwd_nt_lockstep"] = entry(timeit(lambda: ops.gemm_nt(x, w)), fl)
    res["fwd_torch_f32"] = entry(timeit(lambda: torch.mm(x, w.t())), fl)
    wt = w.t().contiguous()
    res["dgrad_nt_lockstep_on_transposed_w"] = entry(timeit(lambda: ops.gemm_nt(g, wt)), fl)
    res["dgrad_torch_f32"] = entry(timeit(lambda: torch.mm(g, w)), fl)
    res["wgrad_tn_lockstep_split_k"] = entry(timeit(lambda: ops.gemm_tn(g, x)), fl)
    res["wgrad_torch_f32"] = entry(timeit(lambda: torch.mm(g.t(), x)), fl)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
