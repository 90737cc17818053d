"""Isolated timing of the sampling-step kernels (single stream, no env workers): run under
`rocprofv3 --kernel-trace --stats` to get per-kernel durations at a given group size."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlpyt_amd import ops  # noqa: E402
from rlpyt_amd.models.pg.atari_ff_model import AtariFfModel  # noqa: E402


def main():
    Bg = int(sys.argv[1]) if len(sys.argv) > 1 else 85
    T, B = 128, 256
    model = AtariFfModel((4, 104, 80), 6).cuda().eval()
    obs = torch.randint(0, 256, (Bg, 4, 104, 80), dtype=torch.uint8, device="cuda")
    prob = torch.zeros((T, B, 6), device="cuda")
    value = torch.zeros((T, B), device="cuda")
    action = torch.zeros((T + 1, B), dtype=torch.int64, device="cuda")
    action_out = torch.zeros(Bg, dtype=torch.int64, device="cuda")
    u = torch.rand(T, Bg, device="cuda")
    t_dev = torch.zeros(1, dtype=torch.int64, device="cuda")
    from collections import namedtuple
    Out = namedtuple("Out", ["prob_rows", "value_rows", "action_rows", "action_out", "uniforms",
                             "t_dev", "lo"])
    out = Out(prob, value, action, action_out, u, t_dev, 0)
    for _ in range(5):
        model.sample_step_into(obs, out)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        model.sample_step_into(obs, out)
    s.record()
    for _ in range(200):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    print(f"Bg={Bg} graph replay (4 kernels) {s.elapsed_time(e) / 200 * 1e3:.1f} us", flush=True)


if __name__ == "__main__":
    main()
