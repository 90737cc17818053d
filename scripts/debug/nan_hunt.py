"""Stress the PPO update on a fixed synthetic batch and report the first non-finite tensor."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rlpyt_amd.agents.pg.atari import AtariFfAgent  # noqa: E402
from rlpyt_amd.agents.pg.categorical import AgentInfo  # noqa: E402
from rlpyt_amd.algos.pg.ppo import PPO  # noqa: E402
from rlpyt_amd.distributions.categorical import DistInfo  # noqa: E402
from rlpyt_amd.envs import EnvSpaces  # noqa: E402
from rlpyt_amd.samplers.collections import AgentSamplesBsv, BatchSpec, EnvSamples, Samples  # noqa: E402
from rlpyt_amd.spaces import IntBox  # noqa: E402
from rlpyt_amd.utils import logger  # noqa: E402

logger.set_quiet(True)
T, B, A = 128, 256, 6
n_itr = int(sys.argv[1]) if len(sys.argv) > 1 else 12
torch.manual_seed(0)
np.random.seed(0)
agent = AtariFfAgent()
agent.initialize(EnvSpaces(observation=IntBox(0, 256, shape=(4, 104, 80), dtype="uint8"), action=IntBox(0, A)))
agent.to_device(0)
algo = PPO(discount=0.99, learning_rate=1e-3, value_loss_coeff=1., entropy_loss_coeff=0.01,
           clip_grad_norm=1., gae_lambda=0.98, minibatches=4, epochs=4, ratio_clip=0.1,
           linear_lr_schedule=True, normalize_advantage=False)
algo.initialize(agent=agent, n_itr=n_itr, batch_spec=BatchSpec(T, B), mid_batch_reset=True, examples=None)
g = torch.Generator().manual_seed(1)
bad = None
for itr in range(n_itr):
    obs = torch.randint(0, 256, (T, B, 4, 104, 80), dtype=torch.uint8, generator=g).cuda()
    all_action = torch.randint(0, A, (T + 1, B), generator=g).cuda()
    all_reward = torch.randint(-1, 2, (T + 1, B), generator=g).float().cuda()
    done = (torch.rand(T, B, generator=g) < 0.02).cuda()
    with torch.no_grad():
        pis, vs = [], []
        for t0 in range(0, T, 32):
            pi, v = agent(obs[t0:t0 + 32], None, None)
            pis.append(pi.prob)
            vs.append(v)
        prob, value = torch.cat(pis), torch.cat(vs)
        bv = agent.value(obs[-1], None, None).reshape(1, B)
    samples = Samples(
        agent=AgentSamplesBsv(action=all_action[1:], prev_action=all_action[:-1],
                              agent_info=AgentInfo(dist_info=DistInfo(prob=prob.clone()), value=value.clone()),
                              bootstrap_value=bv.clone()),
        env=EnvSamples(observation=obs, reward=all_reward[1:], prev_reward=all_reward[:-1], done=done,
                       env_info=()))
    agent.train_mode(itr)
    info = algo.optimize_agent(itr, samples)
    torch.cuda.synchronize()
    fin = all(torch.isfinite(p).all().item() for p in agent.parameters())
    print(f"itr {itr}: loss[0] {info.loss[0]:.4f} loss[-1] {info.loss[-1]:.4f} gradNorm max {max(info.gradNorm):.3f} "
          f"finite params {fin} sampling prob finite {torch.isfinite(prob).all().item()}", flush=True)
    if not fin or not np.isfinite(info.loss).all():
        firstbad = next((i for i, x in enumerate(info.loss) if not np.isfinite(x)), None)
        print("first non-finite loss at update", firstbad, "gradNorm there", info.gradNorm[firstbad] if firstbad is not None else None)
        for n, p in agent.model.named_parameters():
            if not torch.isfinite(p).all():
                print("  non-finite parameter:", n, int((~torch.isfinite(p)).sum()))
        break
