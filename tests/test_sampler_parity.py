"""Rollout parity with the reference's own GPU sampler: tests/golden/sampler.npz
was recorded by running rlpyt's GpuSampler -- GpuResetCollector / GpuWaitResetCollector worker
processes + ActionServer.serve_actions -- on CPU over this repo's synthetic env under a
deterministic policy that depends on the observation and on the prev_action / prev_reward it is
handed.  This repo's GpuSampler must reproduce every field of every batch: observations (CRC),
reward / prev_reward, done, env_info, action / prev_action, agent_info, bootstrap_value, and the
completed-trajectory statistics, for any worker count and pipeline-group layout."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden
from rlpyt_amd.agents.base import AgentStep, BaseAgent
from rlpyt_amd.envs.synthetic import SyntheticPong
from rlpyt_amd.samplers.gpu import GpuSampler
from rlpyt_amd.utils import logger
from rlpyt_amd.utils.collections import namedarraytuple

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import sampler_cases as C  # noqa: E402

logger.set_quiet(True)
AgentInfo = namedarraytuple("AgentInfo", ["value"])


class DetAgent(BaseAgent):
    """The same deterministic policy the golden run used (tests/golden/sampler_cases.py)."""

    def initialize(self, env_spaces, share_memory=False, global_B=1, env_ranks=None):
        self.n = env_spaces.action.n
        self.env_spaces, self.share_memory = env_spaces, share_memory

    def step(self, observation, prev_action, prev_reward):
        a, v = C.det_policy(observation, prev_action, prev_reward, self.n)
        return AgentStep(action=a, agent_info=AgentInfo(value=v))

    def value(self, observation, prev_action, prev_reward):
        return C.det_policy(observation, prev_action, prev_reward, self.n)[1] + 1

    def sample_mode(self, itr):
        pass

    train_mode = eval_mode = sample_mode

    def parameters(self):
        return []


class RefSeededPong(SyntheticPong):
    """SyntheticPong seeded the way the reference's two-worker sampler seeds env i."""

    def seed(self, seed):
        i = seed - C.SEED
        if i >= 50000:      # this repo's evaluation envs are seeded from seed + 50000 + i
            i -= 50000
        super().seed(C.reference_env_seed(C.SEED, i))


@pytest.mark.parametrize("n_workers,n_groups,split", [(0, 1, False), (2, 2, False), (3, 1, False),
                                                       (0, 2, False), (4, 2, True)])
@pytest.mark.parametrize("case", C.CASES, ids=[c[0] for c in C.CASES])
def test_batches_match_reference_gpu_sampler(case, n_workers, n_groups, split):
    name, mode, T, n_batches = case
    g = load_golden("sampler")
    s = GpuSampler(RefSeededPong, C.ENV_KWARGS, batch_T=T, batch_B=C.B, n_workers=n_workers,
                   n_groups=n_groups, mid_batch_reset=(mode == "reset"), max_decorrelation_steps=0,
                   split_workers=split)
    assert s.n_workers == n_workers
    agent = DetAgent()
    s.initialize(agent, seed=C.SEED, bootstrap_value=True, traj_info_kwargs=dict(discount=0.9))
    assert s.split_workers == split      # dedicated workers per pipeline group when asked for
    got_infos, ref_infos = [], []
    for itr in range(n_batches):
        smp, infos = s.obtain_samples(itr)
        k = f"{name}{itr}_"
        assert np.array_equal(C.obs_crc(smp.env.observation.numpy()), g[k + "obs_crc"]), (itr, "obs")
        for field, got in [("reward", smp.env.reward), ("prev_reward", smp.env.prev_reward),
                          ("done", smp.env.done), ("action", smp.agent.action),
                          ("prev_action", smp.agent.prev_action),
                          ("value", smp.agent.agent_info.value),
                          ("bootstrap_value", smp.agent.bootstrap_value),
                          ("game_score", smp.env.env_info.game_score),
                          ("traj_done", smp.env.env_info.traj_done)]:
            got = got.numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
            assert np.array_equal(got, g[k + field]), (itr, field, got, g[k + field])
        got_infos += [(float(ti["Length"]), float(ti["Return"]), float(ti["NonzeroRewards"]),
                       round(float(ti["DiscountedReturn"]), 9)) for ti in infos]
        ref_infos += [tuple(r[:3]) + (round(r[3], 9),) for r in g[k + "traj_fields"].tolist()]
    # the reference drains its trajectory queue asynchronously (infos of the last batches may
    # still be in flight when the run stops): what it returned must be a sub-multiset of ours
    from collections import Counter
    got_c, ref_c = Counter(got_infos), Counter(ref_infos)
    assert len(ref_infos) > 0 and not (ref_c - got_c), (ref_c - got_c)
    assert len(got_infos) - len(ref_infos) <= 2 * C.B
    s.shutdown()


@pytest.mark.parametrize("n_workers,n_groups", [(0, 1), (2, 2)])
def test_atari_ff_agent_batches_match_reference_gpu_sampler(n_workers, n_groups):
    """``sampler_ff.npz``: the reference's GpuSampler + its own AtariFfAgent with a sharpened policy
    head (sampler_cases.ff_sharpen), reproduced by this repo's sampler and agent on CPU tensors
    (the GPU twin -- the fused rollout kernels -- is tests/test_sampler_gpu_parity.py)."""
    from rlpyt_amd.agents.pg.atari import AtariFfAgent
    from rlpyt_amd.models.pg.atari_ff_model import AtariFfModel
    g = load_golden("sampler_ff")
    s = GpuSampler(RefSeededPong, C.FF_ENV_KWARGS, batch_T=C.FF_T, batch_B=C.B, n_workers=n_workers,
                   n_groups=n_groups, mid_batch_reset=True, max_decorrelation_steps=0)
    agent = AtariFfAgent()
    s.initialize(agent, seed=C.SEED, bootstrap_value=True, traj_info_kwargs=dict(discount=0.9))
    torch.manual_seed(C.FF_INIT_SEED)
    agent.load_state_dict(AtariFfModel(image_shape=(4, 104, 80), output_size=6).state_dict())
    C.ff_sharpen(agent.model)
    assert np.array_equal(C.param_checksums(list(agent.parameters())), g["param_crc"])
    for itr in range(C.FF_BATCHES):
        agent.sample_mode(itr)
        smp, _infos = s.obtain_samples(itr)
        k = f"ff{itr}_"
        assert np.array_equal(C.obs_crc(smp.env.observation.numpy()), g[k + "obs_crc"]), itr
        for field, got in [("reward", smp.env.reward), ("done", smp.env.done),
                           ("action", smp.agent.action), ("prev_action", smp.agent.prev_action),
                           ("prob", smp.agent.agent_info.dist_info.prob)]:
            got = got.numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
            assert np.array_equal(got, g[k + field]), (itr, field)
        np.testing.assert_allclose(smp.agent.agent_info.value.numpy(), g[k + "value"],
                                   rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(smp.agent.bootstrap_value.numpy(), g[k + "bootstrap_value"],
                                   rtol=1e-5, atol=1e-6)
    s.shutdown()


@pytest.mark.parametrize("n_workers", [0, 2])
def test_evaluation_matches_reference_gpu_sampler(n_workers):
    """``evaluate_agent`` against the reference GpuSampler's evaluation (eval collectors in the
    workers + ``serve_actions_evaluation``): the same completed trajectories for a step budget,
    twice in a row (the eval envs live on between evaluations), and the training batches in
    between are the ones an evaluation-free run produces."""
    g = load_golden("sampler")
    s = GpuSampler(RefSeededPong, C.ENV_KWARGS, batch_T=5, batch_B=C.B, n_workers=n_workers,
                   max_decorrelation_steps=0, eval_n_envs=C.EVAL_N_ENVS,
                   eval_env_kwargs=C.EVAL_ENV_KWARGS, eval_max_steps=C.EVAL_MAX_STEPS)
    agent = DetAgent()
    s.initialize(agent, seed=C.SEED, bootstrap_value=True)
    for k in range(2):
        infos = s.evaluate_agent(k)
        got = sorted((float(ti["Length"]), float(ti["Return"])) for ti in infos)
        assert got == [tuple(r) for r in g[f"eval{k}_len_ret"].tolist()]
        smp, _ = s.obtain_samples(k)
        assert np.array_equal(smp.agent.action.numpy(), g[f"eval{k}_next_batch_action"])
    s.shutdown()


@pytest.mark.parametrize("T,B,mode,layout", [
    (3, 5, "reset", (2, 2, False)), (3, 5, "wait", (5, 1, False)), (7, 6, "reset", (3, 3, False)),
    (7, 6, "wait", (4, 2, True)), (4, 9, "reset", (6, 3, True)), (4, 9, "wait", (2, 4, False)),
    (1, 4, "reset", (2, 2, False)), (1, 4, "wait", (2, 2, False))])
def test_every_layout_reproduces_the_serial_batches(T, B, mode, layout):
    """Beyond the recorded shapes: for other [T, B] (incl. T = 1 and B not divisible by the
    worker / group counts) every worker / pipeline-group layout must produce exactly the batches
    of the serial single-group sampler -- the layout the reference vectors pin."""
    def run(n_workers, n_groups, split):
        s = GpuSampler(SyntheticPong, dict(points_to_end=1, max_steps=11), batch_T=T, batch_B=B,
                       n_workers=n_workers, n_groups=n_groups, split_workers=split,
                       mid_batch_reset=(mode == "reset"), max_decorrelation_steps=0)
        s.initialize(DetAgent(), seed=21, bootstrap_value=True)
        out, n_infos = [], 0
        for itr in range(40 // T + 6):
            smp, infos = s.obtain_samples(itr)
            n_infos += len(infos)
            out.append([x.numpy().copy() if isinstance(x, torch.Tensor) else np.array(x)
                        for x in (smp.env.reward, smp.env.done, smp.agent.action,
                                  smp.agent.prev_action, smp.env.prev_reward,
                                  smp.agent.agent_info.value, smp.agent.bootstrap_value,
                                  smp.env.env_info.traj_done)]
                       + [C.obs_crc(smp.env.observation.numpy())])
        s.shutdown()
        return out, n_infos
    ref, n_ref = run(0, 1, False)
    got, n_got = run(*layout)
    assert n_ref == n_got > 0
    for a, b in zip(ref, got):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
