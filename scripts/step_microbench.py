"""Steady-state time of each rollout-step kernel in isolation: N back-to-back launches of ONE kernel
captured in a hipGraph, replayed; HIP-event time / launches.  Separates the kernels' own time from
the hand-off chain around them.  usage: step_microbench.py [Bg=64]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlpyt_amd import ops  # noqa: E402
from rlpyt_amd.models.pg.atari_ff_model import AtariFfModel  # noqa: E402


def timed(fn, n_inner=20, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n_inner):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return round(s.elapsed_time(e) / (reps * n_inner) * 1e3, 2)


def main():
    Bg = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    T, B = 128, 256
    torch.manual_seed(0)
    m = AtariFfModel((4, 104, 80), 6).cuda().eval()
    c1, c2 = m.conv.conv.conv[0], m.conv.conv.conv[2]
    lin = m._single_fc()
    obs = torch.randint(0, 256, (T, B, 4, 104, 80), dtype=torch.uint8, device="cuda")
    new_frame = torch.randint(0, 256, (Bg, 104, 80), dtype=torch.uint8, device="cuda")
    full_rows = torch.zeros((Bg, 4, 104, 80), dtype=torch.uint8, device="cuda")
    slot = torch.full((Bg,), -1, dtype=torch.int32, device="cuda")
    t_dev = torch.tensor([5], dtype=torch.int64, device="cuda")
    rew, dn = torch.zeros(T + 1, B, device="cuda"), torch.zeros(T + 1, B, dtype=torch.bool, device="cuda")
    rs, ds = torch.zeros(Bg, device="cuda"), torch.zeros(Bg, dtype=torch.bool, device="cuda")
    feat = torch.randn(Bg, 3456, device="cuda")
    prob = torch.zeros((T, B, 6), device="cuda")
    value = torch.zeros((T, B), device="cuda")
    action = torch.zeros((T + 1, B), dtype=torch.int64, device="cuda")
    action_out = torch.zeros(Bg, dtype=torch.int64, device="cuda")
    u = torch.rand(T, Bg, device="cuda")
    out = {"Bg": Bg}
    y2 = torch.empty((Bg, 3456), device="cuda")
    out["sample_convs"] = timed(lambda: ops.atari_sample_convs(
        obs, t_dev, 0, new_frame, full_rows, slot, c1.weight, c1.bias, c2.weight, c2.bias,
        scalar_rows=(rew, rs, dn, ds), out=y2))
    out["rollout_fc"] = timed(lambda: ops.rollout_fc_partials(feat, lin.weight))
    p2, k2 = ops.rollout_fc_partials(feat, lin.weight)
    out["rollout_head"] = timed(lambda: ops.rollout_head(p2, k2, lin.bias, m.pi.weight, m.pi.bias,
                                                         m.value.weight, m.value.bias, u, t_dev, Bg,
                                                         prob, value, action, 0, action_out))
    x = torch.zeros(64, device="cuda")
    out["tiny_torch_add"] = timed(lambda: x.add_(1))
    q = torch.randn(Bg, 6, device="cuda")
    eps = torch.full((1,), 0.1, device="cuda")
    out["eps_greedy_tiny"] = timed(lambda: ops.eps_greedy(q, eps, u, t_dev))

    def chain():
        f = ops.atari_sample_convs(obs, t_dev, 0, new_frame, full_rows, slot, c1.weight, c1.bias,
                                   c2.weight, c2.bias, scalar_rows=(rew, rs, dn, ds), out=y2)
        p, k = ops.rollout_fc_partials(f, lin.weight)
        ops.rollout_head(p, k, lin.bias, m.pi.weight, m.pi.bias, m.value.weight, m.value.bias, u,
                         t_dev, Bg, prob, value, action, 0, action_out)
    out["chain_per_step"] = round(timed(chain, n_inner=8), 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
