"""Debug harness (GPU box): the DQN conv stack at R2D1's update sizes (thousands of images) -- own kernels
(csrc/dqn_convs.hip / dqn_convs_bwd.hip) against the library path of Conv2dModel.features, no-grad
forward and forward + backward.   python scripts/debug/dqn_convs_large_n.py [N ...]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from rlpyt_amd.models.conv2d import Conv2dModel  # noqa: E402


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


torch.manual_seed(0)
m = Conv2dModel(4, [32, 64, 64], [8, 4, 3], [4, 2, 1], paddings=[0, 1, 1]).cuda()
for N in [int(a) for a in sys.argv[1:]] or [128, 1024, 2560, 5440]:
    obs = torch.randint(0, 256, (N, 4, 104, 80), dtype=torch.uint8, device="cuda")
    cot = torch.randn(N, 6912, device="cuda")
    out = {}
    for name, limit in (("own", 1 << 20), ("lib", 0)):
        m.FUSED_MAX_IMAGES = m.FUSED_MAX_IMAGES_GRAD = limit

        def nograd():
            with torch.no_grad():
                return m.features(obs, N, (4, 104, 80))

        def grad():
            for p in m.parameters():
                p.grad = None
            m.features(obs, N, (4, 104, 80)).backward(cot)
        out[name] = (timed(nograd), timed(grad), nograd(), [p.grad.clone() for p in (grad(), m.parameters())[1]])
    err = (out["own"][2] - out["lib"][2]).abs().max().item() / out["lib"][2].abs().max().item()
    gerr = max((a - b).abs().max().item() / (b.abs().max().item() + 1e-30) for a, b in zip(out["own"][3], out["lib"][3]))
    print(f"N={N}: no-grad fwd own {out['own'][0]:.0f} us / lib {out['lib'][0]:.0f} us; fwd+bwd own "
          f"{out['own'][1]:.0f} us / lib {out['lib'][1]:.0f} us; rel diff fwd {err:.2e} grads {gerr:.2e}")
