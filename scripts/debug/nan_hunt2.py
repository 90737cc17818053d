"""Minibatch-level hunt: loss -> backward -> which gradients are non-finite first."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rlpyt_amd import ops  # noqa: E402
from rlpyt_amd.agents.base import AgentInputs  # noqa: E402
from rlpyt_amd.agents.pg.atari import AtariFfAgent  # noqa: E402
from rlpyt_amd.algos.pg.ppo import PPO  # noqa: E402
from rlpyt_amd.envs import EnvSpaces  # noqa: E402
from rlpyt_amd.models.pg.atari_ff_model import ObsGather  # noqa: E402
from rlpyt_amd.samplers.collections import BatchSpec  # noqa: E402
from rlpyt_amd.spaces import IntBox  # noqa: E402
from rlpyt_amd.utils import logger  # noqa: E402

logger.set_quiet(True)
T, B, A, M = 128, 256, 6, 8192
n_mb = int(sys.argv[1]) if len(sys.argv) > 1 else 200
torch.manual_seed(0)
agent = AtariFfAgent()
agent.initialize(EnvSpaces(observation=IntBox(0, 256, shape=(4, 104, 80), dtype="uint8"), action=IntBox(0, A)))
agent.to_device(0)
algo = PPO(learning_rate=1e-4, clip_grad_norm=1., minibatches=4, epochs=4, linear_lr_schedule=False)
algo.initialize(agent=agent, n_itr=10, batch_spec=BatchSpec(T, B), mid_batch_reset=True, examples=None)
g = torch.Generator().manual_seed(1)
obs = torch.randint(0, 256, (T, B, 4, 104, 80), dtype=torch.uint8, generator=g).cuda()
po = torch.softmax(torch.randn(T, B, A, generator=g), -1).cuda()
act = torch.randint(0, A, (T, B), generator=g).cuda()
adv = torch.randn(T, B, generator=g).cuda()
ret = torch.randn(T, B, generator=g).cuda()
names = [n for n, _ in agent.model.named_parameters()]
for mb in range(n_mb):
    idx = torch.randperm(T * B, generator=g)[:M].cuda()
    algo.optimizer.zero_grad(set_to_none=True)
    loss, sc = algo.loss(AgentInputs(ObsGather(obs, idx), None, None), act, ret, adv, None, po, flat_idx=idx)
    loss.backward()
    torch.cuda.synchronize()
    badg = [n for n, p in agent.model.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
    big = {n: float(p.grad.abs().max()) for n, p in agent.model.named_parameters() if p.grad is not None}
    if badg or not torch.isfinite(sc).all() or max(big.values()) > 1e3:
        print(f"minibatch {mb}: scalars {sc.tolist()} non-finite grads: {badg}")
        print("  max |grad|:", {k: f"{v:.3e}" for k, v in big.items()})
        # which stage?  recompute the forward pieces
        with torch.no_grad():
            m = agent.model
            c1, c2 = m.conv.conv.conv[0], m.conv.conv.conv[2]
            feat = ops.atari_conv_stack(obs, idx, c1.weight, c1.bias, c2.weight, c2.bias)
            print("  features finite (recomputed):", torch.isfinite(feat).all().item(), float(feat.abs().max()))
        break
    gn = algo.clip_and_step()
    if mb % 20 == 0:
        print(f"minibatch {mb}: loss {sc[0].item():.4f} gradnorm {gn.item():.4f}", flush=True)
else:
    print("no failure in", n_mb, "minibatches")
