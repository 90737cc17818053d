"""Debug (GPU box): which conv-stack gradient differs between the own backward and the library at N >= 300,
against float64 modules on the CPU and on the device."""
import sys

import torch

sys.path.insert(0, ".")
from rlpyt_amd import ops  # noqa: E402
sys.path.insert(0, "scripts/debug")
from dqn_bwd_check import module_path, rel, stack  # noqa: E402

for N in [int(a) for a in sys.argv[1:]] or [300, 1024]:
    convs = stack(40 + N)
    g = torch.Generator().manual_seed(N)
    obs = torch.randint(0, 256, (N, 4, 104, 80), dtype=torch.uint8, generator=g)
    cot = torch.randn(N, 6912, generator=g)
    g64c = module_path(convs, obs, cot, torch.float64, "cpu")
    g64d = module_path(convs, obs, cot, torch.float64, "cuda")
    g32 = module_path(convs, obs, cot, torch.float32, "cuda")
    res = {}
    for own in (True, False):
        ops.DQN_CONVS_OWN_BWD = own
        dev = [torch.nn.Conv2d(c.in_channels, c.out_channels, c.kernel_size, c.stride, c.padding).cuda()
               for c in convs]
        for c, src in zip(dev, convs):
            c.load_state_dict(src.state_dict())
        params = [p for c in dev for p in (c.weight, c.bias)]
        ops.dqn_convs(obs.cuda(), *params).backward(cot.cuda())
        res[own] = [p.grad.detach().cpu().double() for p in params]
    print(f"N={N}")
    for k, nm in enumerate(["dw1", "db1", "dw2", "db2", "dw3", "db3"]):
        print(f"  {nm}: f64 cpu vs f64 device {rel(g64c[k], g64d[k]):.2e} | vs f64 device: own {rel(res[True][k], g64d[k]):.2e} "
              f"lib-on-kept {rel(res[False][k], g64d[k]):.2e} module-f32 {rel(g32[k], g64d[k]):.2e}")
