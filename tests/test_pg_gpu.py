"""End-to-end GPU runs of the policy-gradient configs through the runner (BASELINE config #2 at
test size, and A2C): rollout in HBM -> fused scan -> fused conv stack / head+loss kernels ->
optimizer, driven by MinibatchRl exactly as the reference's runner drives its classes."""
import numpy as np
import pytest
import torch

from rlpyt_amd.agents.pg.atari import AtariFfAgent
from rlpyt_amd.algos.pg.a2c import A2C
from rlpyt_amd.algos.pg.ppo import PPO
from rlpyt_amd.envs.synthetic import SyntheticPong
from rlpyt_amd.runners.minibatch_rl import MinibatchRl
from rlpyt_amd.samplers.collections import AtariTrajInfo
from rlpyt_amd.samplers.gpu import GpuSampler
from rlpyt_amd.utils import logger

pytestmark = pytest.mark.gpu
logger.set_quiet(True)


def _run(algo, T, B, n_itr, AgentCls=AtariFfAgent, **sampler_kw):
    sampler = GpuSampler(SyntheticPong, dict(points_to_end=2, max_steps=60), batch_T=T, batch_B=B,
                         n_workers=2, TrajInfoCls=AtariTrajInfo, max_decorrelation_steps=5,
                         **sampler_kw)
    agent = AgentCls()
    runner = MinibatchRl(algo=algo, agent=agent, sampler=sampler, n_steps=T * B * n_itr, seed=0,
                         affinity=dict(cuda_idx=0), log_interval_steps=T * B * n_itr)
    torch.cuda.set_device(0)
    before = None

    orig = algo.optimize_agent
    infos = []

    def spy(itr, samples):
        nonlocal before
        if before is None:
            before = torch.cat([p.detach().reshape(-1).clone() for p in agent.parameters()])
        out = orig(itr, samples)
        infos.append(out)
        return out
    algo.optimize_agent = spy
    runner.train()
    after = torch.cat([p.detach().reshape(-1) for p in agent.parameters()])
    assert torch.isfinite(after).all() and not torch.equal(before, after)
    assert runner.last_steps_per_second > 0
    return infos


@pytest.mark.parametrize("fused", [True, False])
def test_ppo_runner_end_to_end(fused):
    algo = PPO(learning_rate=3e-4, gae_lambda=0.95, minibatches=2, epochs=2, fused_head_loss=fused)
    infos = _run(algo, T=16, B=8, n_itr=4)
    assert algo.update_counter == 4 * 4
    for info in infos:
        assert len(info.loss) == 4 and np.all(np.isfinite(info.loss))
        assert np.all(np.isfinite(info.gradNorm)) and np.all(np.asarray(info.entropy) > 0)
        assert np.all(np.asarray(info.perplexity) <= 6.0 + 1e-4)


def test_ppo_wait_reset_with_valid_mask():
    """mid_batch_reset=False: the valid mask path (valid_from_done fused in the scan, masked
    loss means) end to end."""
    algo = PPO(learning_rate=3e-4, gae_lambda=0.95, minibatches=2, epochs=1,
               normalize_advantage=True)
    infos = _run(algo, T=16, B=8, n_itr=3, mid_batch_reset=False)
    for info in infos:
        assert np.all(np.isfinite(info.loss))


def test_a2c_runner_end_to_end():
    algo = A2C(learning_rate=3e-4, gae_lambda=1)
    infos = _run(algo, T=5, B=8, n_itr=4)
    assert algo.update_counter == 4
    assert all(np.isfinite(i.loss) and np.isfinite(i.gradNorm) for i in infos)


@pytest.mark.parametrize("algo_name", ["ppo", "a2c"])
def test_recurrent_pg_runner_end_to_end(algo_name):
    """AtariLstmAgent under the HBM sampler (persistent per-group LSTM state, prev_rnn_state rows)
    and the recurrent branches of PPO / A2C (whole columns, state of row 0, valid mask) through
    the runner (rlpyt/agents/pg/atari.py:27-30, rlpyt/algos/pg/ppo.py:84-99)."""
    from rlpyt_amd.agents.pg.atari import AtariLstmAgent
    if algo_name == "ppo":
        algo = PPO(learning_rate=3e-4, gae_lambda=0.95, minibatches=2, epochs=2)
        infos = _run(algo, T=12, B=8, n_itr=3, AgentCls=AtariLstmAgent, mid_batch_reset=False)
        assert algo.update_counter == 3 * 4
        for info in infos:
            assert len(info.loss) == 4 and np.all(np.isfinite(info.loss))
            assert np.all(np.asarray(info.perplexity) <= 6.0 + 1e-4)
    else:
        algo = A2C(learning_rate=3e-4, gae_lambda=1)
        infos = _run(algo, T=6, B=8, n_itr=3, AgentCls=AtariLstmAgent, mid_batch_reset=False)
        assert algo.update_counter == 3
        assert all(np.isfinite(i.loss) and np.isfinite(i.gradNorm) for i in infos)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs a second GPU")
def test_runner_on_second_device():
    """affinity['cuda_idx'] = 1 with the process's current device left at 0: the runner
    itself must make device 1 current before the sampler, the workspaces and every kernel launch --
    parameters, batch and update all on cuda:1, finite losses."""
    torch.cuda.set_device(0)
    sampler = GpuSampler(SyntheticPong, dict(points_to_end=2, max_steps=60), batch_T=16, batch_B=8,
                         n_workers=2, TrajInfoCls=AtariTrajInfo, max_decorrelation_steps=5)
    agent = AtariFfAgent()
    algo = PPO(learning_rate=3e-4, gae_lambda=0.95, minibatches=2, epochs=2)
    runner = MinibatchRl(algo=algo, agent=agent, sampler=sampler, n_steps=16 * 8 * 3, seed=0,
                         affinity=dict(cuda_idx=1), log_interval_steps=16 * 8 * 3)
    try:
        runner.train()
        assert torch.cuda.current_device() == 1
        assert all(p.device == torch.device("cuda", 1) for p in agent.parameters())
        assert sampler.samples.env.observation.device == torch.device("cuda", 1)
        assert all(torch.isfinite(p).all().item() for p in agent.parameters())
        assert algo.update_counter == 3 * 4
    finally:
        torch.cuda.set_device(0)
