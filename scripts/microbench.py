"""Kernel-level micro-benchmarks on the MI355X: HIP-event timing of each C-ABI kernel on
the synthetic inputs of SURVEY.md section 8(d); prints one JSON line per measurement with
the algorithmic bytes (section 8d figures) and achieved GB/s against the 8 TB/s HBM peak."""
import argparse
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlpyt_amd import ops  # noqa: E402

HBM_PEAK = 8000.0  # GB/s (MI355X_MICROARCH.md)


def timeit(fn, iters=50, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3  # seconds


def report(name, secs, nbytes, **kw):
    gbs = nbytes / secs / 1e9
    print(json.dumps(dict(kernel=name, us=round(secs * 1e6, 2), alg_bytes=nbytes,
                          GBps=round(gbs, 1), frac_hbm=round(gbs / HBM_PEAK, 4), **kw)), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-log2-n", type=int, default=20)
    args = ap.parse_args()
    dev = "cuda"
    T = 128
    for lg in [8, 12, 14, 16, 18, 20, 21, 22]:
        if lg > args.max_log2_n + 2:
            break
        N = 1 << lg
        r = torch.randn(T, N, device=dev) * 0.5
        v = torch.randn(T, N, device=dev)
        d = torch.rand(T, N, device=dev) < 0.01
        bv = torch.randn(1, N, device=dev)
        adv, ret = torch.empty_like(r), torch.empty_like(r)
        it = 200 if lg < 16 else 20
        s = timeit(lambda: ops.gae(r, v, d, bv, 0.99, 0.98, advantage_dest=adv, return_dest=ret), it)
        report("gae_exact", s, T * N * 17 + 4 * N, T=T, N=N)
        s = timeit(lambda: ops.gae(r, v, d, bv, 0.99, 0.98, advantage_dest=adv, return_dest=ret,
                                   with_valid=True), it)
        report("gae_exact+valid", s, T * N * 21 + 4 * N, T=T, N=N)
        if lg <= 16:
            s = timeit(lambda: ops.gae(r, v, d, bv, 0.99, 0.98, advantage_dest=adv,
                                       return_dest=ret, variant=ops.SCAN_SEGMENTED), it)
            report("gae_segmented", s, T * N * 17 + 4 * N, T=T, N=N)
        s = timeit(lambda: ops.discount_return(r, d, bv, 0.99, return_dest=ret), it)
        report("discount_return", s, T * N * 9 + 4 * N, T=T, N=N)
        s = timeit(lambda: ops.valid_from_done(d), it)
        report("valid_from_done", s, T * N * 5, T=T, N=N)
        a2 = adv.clone()
        s = timeit(lambda: ops.normalize_advantage_(a2), it)
        report("adv_normalize", s, T * N * 16, T=T, N=N)
        del r, v, d, bv, adv, ret, a2
    # PPO loss fwd+bwd
    for M in [8192, 1 << 20]:
        A = 6
        pn = torch.softmax(torch.randn(M, A, device=dev), -1)
        po = torch.softmax(torch.randn(M, A, device=dev), -1)
        act = torch.randint(0, A, (M,), device=dev)
        adv, ret, val = (torch.randn(M, device=dev) for _ in range(3))
        s = timeit(lambda: ops.ppo_loss(pn, val, po, act, adv, ret, None, 0.1, 1.0, 0.01), 100)
        report("ppo_loss_fwd_bwd", s, M * (12 * A + 28), M=M, A=A)
    # minibatch gather at the PPO config shape
    Tb, B = 128, 256
    obs = torch.randint(0, 256, (Tb, B, 4, 104, 80), dtype=torch.uint8, device=dev)
    idx = torch.randperm(Tb * B, device=dev)[:8192]
    out = torch.empty((8192, 4, 104, 80), dtype=torch.uint8, device=dev)
    s = timeit(lambda: ops.gather_tb(obs, idx, out=out), 20)
    report("gather_tb_obs", s, 2 * 8192 * 33280, M=8192)
    s = timeit(lambda: obs[idx % Tb, idx // Tb], 20)
    report("torch_index_obs(ref)", s, 2 * 8192 * 33280, M=8192)
    del obs, out
    # replay: frame gather + sum tree at the DQN config (1M leaves, n=128)
    Tr, Br, C = 62500, 16, 4
    frames = torch.randint(0, 256, (Tr + C - 1, Br, 104, 80), dtype=torch.uint8, device=dev)
    done = torch.rand(Tr, Br, device=dev) < 0.005
    Ti = torch.randint(0, Tr, (128,), device=dev)
    Bi = torch.randint(0, Br, (128,), device=dev)
    o = torch.empty((128, C, 104, 80), dtype=torch.uint8, device=dev)
    s = timeit(lambda: ops.frames_gather(frames, done, Ti, Bi, C, out=o), 100)
    report("frames_gather", s, 2 * 128 * C * 8320, n=128)
    Ti2 = torch.randint(0, Tr - 200, (64,), device=dev)
    Bi2 = torch.randint(0, Br, (64,), device=dev)
    o2 = torch.empty((125, 64, C, 104, 80), dtype=torch.uint8, device=dev)
    s = timeit(lambda: ops.frames_gather_seq(frames, done, Ti2, Bi2, C, 125, out=o2), 20)
    report("frames_gather_seq", s, 64 * 128 * 8320 + 125 * 64 * C * 8320, n=64, seq_T=125)
    del frames, o, o2
    tree = ops.DeviceSumTree(Tr, Br, 1, 3, default_value=1.0)
    for _ in range(60):
        tree.advance(1000)
    u = torch.rand(128, dtype=torch.float64, device=dev)
    s = timeit(lambda: tree.sample(u), 200)
    report("sumtree_sample", s, 128 * 20 * 8, n=128, ns_per_sample=round(s * 1e9 / 128, 1))
    newp = torch.rand(128, dtype=torch.float64, device=dev)
    s = timeit(lambda: tree.update_batch_priorities(newp), 200)
    report("sumtree_update", s, 128 * 21 * 16, n=128)
    s = timeit(lambda: tree.advance(2), 200)
    report("sumtree_advance", s, 2 * 2 * 16 * 21 * 16, T_new=2)


if __name__ == "__main__":
    main()
