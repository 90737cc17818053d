"""Debug harness (GPU box): the own backward of the DQN conv stack (csrc/dqn_convs_bwd.hip) against
the torch.nn.Conv2d modules in float64 / float32 and against the library backward on the same kept
activations, per parameter, plus the time of both backward paths at the update batch size.
  python scripts/debug/dqn_bwd_check.py [N ...]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from rlpyt_amd import ops  # noqa: E402


def stack(seed):
    torch.manual_seed(seed)
    cs = [torch.nn.Conv2d(4, 32, 8, stride=4), torch.nn.Conv2d(32, 64, 4, stride=2, padding=1),
          torch.nn.Conv2d(64, 64, 3, stride=1, padding=1)]
    with torch.no_grad():
        for c in cs:
            c.bias.uniform_(-0.2, 0.2)
    return cs


def module_path(convs, obs, cot, dtype, device):
    cs = [torch.nn.Conv2d(c.in_channels, c.out_channels, c.kernel_size, c.stride, c.padding)
          .to(device=device, dtype=dtype) for c in convs]
    for c, src in zip(cs, convs):
        c.load_state_dict({k: v.to(dtype) for k, v in src.state_dict().items()})
    x = obs.to(device=device, dtype=dtype) * (1. / 255)
    for c in cs:
        x = torch.relu(c(x))
    y = x.reshape(obs.shape[0], -1)
    y.backward(cot.to(device=device, dtype=dtype))
    return [p.grad.detach().cpu().double() for c in cs for p in c.parameters()]


def own_path(convs, obs, cot, own):
    ops.DQN_CONVS_OWN_BWD = own
    dev = [torch.nn.Conv2d(c.in_channels, c.out_channels, c.kernel_size, c.stride, c.padding).cuda()
           for c in convs]
    for c, src in zip(dev, convs):
        c.load_state_dict(src.state_dict())
    params = [p for c in dev for p in (c.weight, c.bias)]
    o, ct = obs.cuda(), cot.cuda()

    def run():
        for p in params:
            p.grad = None
        y = ops.dqn_convs(o, *params)
        y.backward(ct)
    run()
    torch.cuda.synchronize()
    got = [p.grad.detach().cpu().double() for p in params]
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        run()
    torch.cuda.synchronize()
    return got, (time.perf_counter() - t0) / 50 * 1e6


def rel(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)


def main():
    for N in [int(a) for a in sys.argv[1:]] or [1, 32, 128, 300]:
        convs = stack(40 + N)
        g = torch.Generator().manual_seed(N)
        obs = torch.randint(0, 256, (N, 4, 104, 80), dtype=torch.uint8, generator=g)
        cot = torch.randn(N, 6912, generator=g)
        g64 = module_path(convs, obs, cot, torch.float64, "cpu")
        g32 = module_path(convs, obs, cot, torch.float32, "cuda")
        own, t_own = own_path(convs, obs, cot, True)
        libp, t_lib = own_path(convs, obs, cot, False)
        names = ["dw1", "db1", "dw2", "db2", "dw3", "db3"]
        print(f"N={N}: fwd+bwd own {t_own:.1f} us, library backward {t_lib:.1f} us")
        for k, nm in enumerate(names):
            print(f"  {nm}: own {rel(own[k], g64[k]):.3e}  lib-on-kept {rel(libp[k], g64[k]):.3e}  "
                  f"module-f32 {rel(g32[k], g64[k]):.3e}")


if __name__ == "__main__":
    main()
