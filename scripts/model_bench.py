"""A/B timing of the AtariFfModel training step pieces on the MI355X (HIP events)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlpyt_amd import ops  # noqa: E402
from rlpyt_amd.models.pg.atari_ff_model import AtariFfModel  # noqa: E402


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
    return s.elapsed_time(e) / iters


def main():
    T, B, M = 128, 256, 8192
    obs = torch.randint(0, 256, (T, B, 4, 104, 80), dtype=torch.uint8, device="cuda")
    idx = torch.randperm(T * B, device="cuda")[:M]
    res = {}
    variants = [("nchw", False), ("nhwc", True)]
    if len(sys.argv) > 1:
        variants = [v for v in variants if v[0] == sys.argv[1]]
    for name, cl in variants:
        model = AtariFfModel((4, 104, 80), 6).cuda()
        if cl:
            model = model.to(memory_format=torch.channels_last)

        def prep_ref():
            return ops.gather_tb(obs, idx).float().mul_(1. / 255)

        def prep_fused():
            return ops.obs_to_nhwc_f32(obs, idx)
        prep = prep_fused if cl else prep_ref

        def fwd_bwd():
            x = prep()
            pi, v = model(x, None, None)
            (pi.sum() + v.sum()).backward()
        res[f"{name}_prep_ms"] = timeit(prep)
        res[f"{name}_fwd_bwd_ms"] = timeit(fwd_bwd)
        with torch.no_grad():
            o = obs[0]
            res[f"{name}_sample_fwd_B256_ms"] = timeit(lambda: model(o, None, None), iters=50)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
