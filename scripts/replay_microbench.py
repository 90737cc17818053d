"""The replay-sample kernels in isolation, at the shapes BASELINE configs #3 / #5 issue them, on rings far
larger than the L2s (so a launch reads HBM, not cache): frames_gather_pair [2,128,4,104,80] on a
[62503,16] frame ring (8.3 GB), frames_gather_seq [125,64,4,104,80] on a [16387,64] ring (8.7 GB),
replay_step_fields (128 samples) and the 1M-leaf sum-tree draw.  Prints HIP-event us per launch;
scripts/replay_pmc.sh runs it under rocprofv3 --pmc for the HBM traffic per launch.
usage: replay_microbench.py [reps=30]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlpyt_amd import ops  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return round(s.elapsed_time(e) / reps * 1e3, 2)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    g = torch.Generator().manual_seed(0)
    C, H, W = 4, 104, 80
    out = {}
    # ---- DQN: [T=62500, B=16] ring, batch 128, agent + 1-step target stacks ------------------------
    T, B, n = 62500, 16, 128
    frames = torch.empty((T + C - 1, B, H, W), dtype=torch.uint8, device="cuda").random_(0, 256)
    done = (torch.rand(T, B, generator=g) < 0.005).cuda()
    # a fresh index set per launch: a repeated one would be served from the Infinity Cache
    idx = [(torch.randint(C, T - 8, (n,), generator=g).cuda(), torch.randint(0, B, (n,), generator=g).cuda())
           for _ in range(reps + 3)]
    pair = torch.empty((2, n, C, H, W), dtype=torch.uint8, device="cuda")
    it = iter(range(10 ** 9))

    def f_pair():
        t, b = idx[next(it) % len(idx)]
        ops.frames_gather_pair(frames, done, t, b, C, 1, out=pair)
    out["frames_gather_pair"] = {"us": timed(f_pair, reps), "alg_bytes": 2 * 2 * n * C * H * W}
    action = torch.randint(0, 6, (T, B), generator=g).cuda()
    reward = torch.randn(T, B, generator=g).cuda()
    ret = torch.randn(T, B, generator=g).cuda()
    done_n = (torch.rand(T, B, generator=g) < 0.01).cuda()

    def f_fields():
        t, b = idx[next(it) % len(idx)]
        ops.replay_step_fields(action, reward, done, ret, done_n, t, b, 1)
    out["replay_step_fields"] = {"us": timed(f_fields, reps), "alg_bytes": n * (8 * 3 + 4 * 4 + 3 + 16)}
    tree = ops.DeviceSumTree(T, B, 1, 3, default_value=1.0)
    tree.advance(T - 8)
    us = [torch.rand(n, generator=g, dtype=torch.float64).cuda() for _ in range(8)]
    out["sumtree_sample"] = {"us": timed(lambda: tree.sample(us[next(it) % 8]), reps),
                             "alg_bytes": n * 8 * (int(tree.tree_levels) - 1)}
    del frames, pair, tree
    torch.cuda.empty_cache()
    # ---- R2D1: [T=16384, B=64] ring, 64 sequences of 125 steps -------------------------------------
    T, B, n, seq_T = 16384, 64, 64, 125
    frames = torch.empty((T + C - 1, B, H, W), dtype=torch.uint8, device="cuda").random_(0, 256)
    done = (torch.rand(T, B, generator=g) < 0.005).cuda()
    idx2 = [((torch.randint(0, (T - seq_T - 8) // 40, (n,), generator=g) * 40).cuda(),
             torch.randint(0, B, (n,), generator=g).cuda()) for _ in range(min(reps, 10) + 3)]
    seq = torch.empty((seq_T, n, C, H, W), dtype=torch.uint8, device="cuda")

    def f_seq():
        t, b = idx2[next(it) % len(idx2)]
        ops.frames_gather_seq(frames, done, t, b, C, seq_T, out=seq)
    out["frames_gather_seq"] = {"us": timed(f_seq, min(reps, 10)),
                                "alg_bytes": n * (seq_T + C - 1) * H * W + seq_T * n * C * H * W}
    for v in out.values():
        v["GBps"] = round(v["alg_bytes"] / v["us"] * 1e-3, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
