"""Fused conv1 -> conv2 update forward (rlpyt_atari_convs_fwd_f32) against the two separate launches:
bit identity of y1 / y2 / relu_mask at several M, and HIP-event timing at M = 8192."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rlpyt_amd._lib import check, lib, ptr, stream  # noqa: E402


def run(M, T=128, B=256, seed=0, time_it=False):
    g = torch.Generator(device="cuda").manual_seed(seed)
    obs = torch.randint(0, 256, (T, B, 4, 104, 80), dtype=torch.uint8, device="cuda", generator=g)
    idx = torch.randperm(T * B, device="cuda", generator=g)[:M]
    w1 = torch.randn(16, 4, 8, 8, device="cuda", generator=g) * 0.06
    b1 = torch.randn(16, device="cuda", generator=g) * 0.1
    w2 = torch.randn(32, 16, 4, 4, device="cuda", generator=g) * 0.06
    b2 = torch.randn(32, device="cuda", generator=g) * 0.1
    out = {}
    for tag in ("sep", "fused"):
        y1 = torch.full((M, 475, 16), float("nan"), device="cuda")
        y2 = torch.full((M, 3456), float("nan"), device="cuda")
        mk = torch.full((M, 128), -1, dtype=torch.int32, device="cuda")
        st = stream()
        if tag == "sep":
            def fn():
                check(lib.rlpyt_atari_conv1_fwd_f32(ptr(obs), ptr(idx), T, B, M, ptr(w1), ptr(b1), 1. / 255,
                                                    ptr(y1), st))
                check(lib.rlpyt_atari_conv2_fwd_f32(ptr(y1), M, ptr(w2), ptr(b2), ptr(y2), ptr(mk), st))
        else:
            def fn():
                check(lib.rlpyt_atari_convs_fwd_f32(ptr(obs), ptr(idx), T, B, M, ptr(w1), ptr(b1), ptr(w2),
                                                    ptr(b2), 1. / 255, ptr(y1), ptr(y2), ptr(mk), st))
        fn()
        torch.cuda.synchronize()
        us = None
        if time_it:
            for _ in range(3):
                fn()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                fn()
            e.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(e) / 20 * 1e3
        out[tag] = (y1, y2, mk, us)
    a, b = out["sep"], out["fused"]
    res = dict(M=M, y1_equal=bool(torch.equal(a[0], b[0])), y2_equal=bool(torch.equal(a[1], b[1])),
               mask_equal=bool(torch.equal(a[2], b[2])), finite=bool(torch.isfinite(b[1]).all()),
               us_sep=a[3], us_fused=b[3])
    if M <= 1024:      # against torch's own convolutions in float64
        import torch.nn.functional as F
        rows = obs.view(T * B, 4, 104, 80)[(idx % T) * B + idx // T].double() / 255
        r1 = F.relu(F.conv2d(rows, w1.double(), b1.double(), stride=4))
        r2 = F.relu(F.conv2d(r1, w2.double(), b2.double(), stride=2, padding=1)).reshape(M, -1)
        res["y1_err_vs_f64"] = float((b[0].double() - r1.permute(0, 2, 3, 1).reshape(M, 475, 16)).abs().max())
        res["y2_err_vs_f64"] = float((b[1].double() - r2).abs().max())
    if not res["y2_equal"]:
        d = (a[1] - b[1]).abs()
        res["y2_maxdiff"] = float(d.max())
        res["y2_nbad"] = int((d > 0).sum())
        res["y1_nbad"] = int((a[0] != b[0]).sum())
    return res


if __name__ == "__main__":
    for M in (257, 300, 1000, 2048):
        print(json.dumps(run(M, seed=M)), flush=True)
    print(json.dumps(run(8192, time_it=True)), flush=True)
    print(json.dumps(run(8192, seed=5, time_it=True)), flush=True)
