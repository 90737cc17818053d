"""Per-kernel timing of the fp32-MFMA conv stack at the PPO minibatch shape (M=8192) and the
A/B against the MIOpen path of the same AtariFfModel (HIP events on torch's stream)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlpyt_amd import ops  # noqa: E402
from rlpyt_amd._lib import check, lib, ptr, stream  # noqa: E402
from rlpyt_amd.models.pg.atari_ff_model import AtariFfModel, ObsGather  # noqa: E402

F32_PEAK_TF = 157.3


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3   # us


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    T, B = 128, 256
    obs = torch.randint(0, 256, (T, B, 4, 104, 80), dtype=torch.uint8, device="cuda")
    idx = torch.randperm(T * B, device="cuda")[:M]
    model = AtariFfModel((4, 104, 80), 6).cuda()
    c1, c2 = model.conv.conv.conv[0], model.conv.conv.conv[2]
    w1, b1, w2, b2 = (p.detach().contiguous() for p in (c1.weight, c1.bias, c2.weight, c2.bias))
    y1 = torch.empty((M, 475, 16), device="cuda")
    y2 = torch.empty((M, 3456), device="cuda")
    mask2 = torch.empty((M, 128), dtype=torch.int32, device="cuda")
    g2 = torch.randn(M, 3456, device="cuda")
    dy1 = torch.empty_like(y1)
    ws = torch.empty(lib.rlpyt_atari_conv_wgrad_workspace_bytes(), dtype=torch.uint8, device="cuda")
    dw1, db1 = torch.empty_like(w1), torch.empty_like(b1)
    dw2, db2 = torch.empty_like(w2), torch.empty_like(b2)
    st = stream()
    kern = {
        "conv1_fwd": (lambda: check(lib.rlpyt_atari_conv1_fwd_f32(
            ptr(obs), ptr(idx), T, B, M, ptr(w1), ptr(b1), 1. / 255, ptr(y1), st)), ops._FL_C1),
        "conv2_fwd": (lambda: check(lib.rlpyt_atari_conv2_fwd_f32(
            ptr(y1), M, ptr(w2), ptr(b2), ptr(y2), ptr(mask2), st)), ops._FL_C2),
        "convs_fwd": (lambda: check(lib.rlpyt_atari_convs_fwd_f32(
            ptr(obs), ptr(idx), T, B, M, ptr(w1), ptr(b1), ptr(w2), ptr(b2), 1. / 255, ptr(y1), ptr(y2),
            ptr(mask2), st)), ops._FL_C1 + ops._FL_C2),
        "conv2_bwd_x6": (lambda: check(lib.rlpyt_atari_conv2_bwd_x6_f32(
            ptr(g2), ptr(mask2), ptr(y1), M, ptr(w2), ptr(dy1), ptr(ws), ptr(dw2), ptr(db2), st)),
            ops._FL_C2D + ops._FL_C2),
        "conv1_wgrad": (lambda: check(lib.rlpyt_atari_conv1_wgrad_f32(
            ptr(obs), ptr(idx), T, B, M, ptr(dy1), 1. / 255, ptr(ws), ptr(dw1), ptr(db1), st)),
            ops._FL_C1),
    }
    res = {"M": M}
    only = [a.split("=", 1)[1].split(",") for a in sys.argv if a.startswith("--only=")]
    for name, (fn, fl) in kern.items():
        if only and name not in only[0]:
            continue
        us = timeit(fn)
        tf = M * fl / us / 1e6
        res[name] = {"us": round(us, 1), "TFLOPs": round(tf, 1), "frac_f32_peak": round(tf / F32_PEAK_TF, 3)}
    if "--no-model" not in sys.argv:
        for fused in (True, False):
            model.use_fused_conv = fused
            if not fused:
                model.to(memory_format=torch.channels_last)

            def fwd_bwd():
                model.zero_grad(set_to_none=True)
                pi, v = model(ObsGather(obs, idx), None, None)
                (pi.sum() + v.sum()).backward()
            res["model_fwd_bwd_us_" + ("fused" if fused else "miopen")] = round(timeit(fwd_bwd, 5, 2), 1)
            with torch.no_grad():
                o = obs[0]
                res["model_sample_fwd_B256_us_" + ("fused" if fused else "miopen")] = round(
                    timeit(lambda: model(o, None, None), 30, 5), 1)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
