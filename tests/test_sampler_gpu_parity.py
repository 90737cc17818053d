"""GPU-side rollout parity against the REFERENCE's recorded batches (reference
``rlpyt/samplers/parallel/gpu/action_server.py:17-74``, ``sampler.py:45-56``).

``tests/test_sampler_parity.py`` holds the host logic of this repo's GpuSampler to
``tests/golden/sampler.npz`` on CPU.  Here the PRODUCT device path of the MI355X -- batch resident in
HBM, newest-frame upload + on-device stack rebuild, captured step hipGraphs, actions written in place
into the page-locked step buffer, the C serve loop -- runs against recordings of the reference's own
GpuSampler:

* ``sampler.npz``: the deterministic policy of ``tests/golden/sampler_cases.py`` evaluated ON THE
  DEVICE (every field bit-identical; the policy's own float value to 1 ulp of the device's
  division), reset and wait-reset collectors;
* ``sampler_ff.npz``: the reference's own ``AtariFfAgent`` (sharpened policy head, see
  ``sampler_cases.ff_sharpen``) -- here the fused rollout kernels ``sample_convs_kernel`` ->
  ``rollout_fc_kernel`` -> ``rollout_head_kernel`` produce the batch: observations / actions /
  rewards / dones / env_info / one-hot probabilities bit-identical, value and bootstrap value within
  fp32 accumulation-order tolerance (rtol 1e-4, atol 2e-6).
"""
import os
import sys
from collections import Counter

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

pytestmark = pytest.mark.gpu
logger.set_quiet(True)
AgentInfo = namedarraytuple("AgentInfo", ["value"])


class DeviceDetAgent(BaseAgent):
    """The golden run's deterministic policy as a device "model": ``step`` is called by the sampler
    on HBM staging tensors inside the captured step graph (generic-agent path: frame push kernel
    -> this forward -> row-commit kernel)."""

    supports_sample_uniforms = True       # no RNG in the step: graphs + native serve loop engage

    def initialize(self, env_spaces, share_memory=False, global_B=1, env_ranks=None):
        self.n = env_spaces.action.n
        self.env_spaces, self.share_memory = env_spaces, share_memory

    def to_device(self, cuda_idx=None):
        self.device = torch.device("cuda", index=cuda_idx)

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
        super().seed(C.reference_env_seed(C.SEED, seed - C.SEED))


def _np(x):
    return x.cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def _traj_rows(infos):
    return [(float(ti["Length"]), float(ti["Return"]), float(ti["NonzeroRewards"]),
             round(float(ti["DiscountedReturn"]), 9)) for ti in infos]


def _assert_product_path(s, fused):
    """Captured graphs on every group, zero-copy action hand-off, C serve loop."""
    assert all(G.graph is not None for G in s.groups), "step graphs were not captured"
    assert all(G.zc for G in s.groups), "actions did not go through the page-locked step buffer"
    assert all(G.dedup for G in s.groups), "newest-frame upload not engaged"
    assert s._native is not None, "the time-step loop did not run in rlpyt_sampler_serve"


@pytest.mark.parametrize("n_workers,n_groups,split", [(2, 2, False), (2, 1, False), (4, 2, True)])
@pytest.mark.parametrize("case", C.CASES, ids=[c[0] for c in C.CASES])
def test_device_path_reproduces_reference_gpu_sampler_batches(case, n_workers, n_groups, split):
    name, mode, T, n_batches = case
    g = load_golden("sampler")
    s = GpuSampler(RefSeededPong, C.ENV_KWARGS, batch_T=T, batch_B=C.B, n_workers=n_workers,
                   n_groups=n_groups, mid_batch_reset=(mode == "reset"), max_decorrelation_steps=0,
                   split_workers=split)
    agent = DeviceDetAgent()
    s.initialize(agent, seed=C.SEED, bootstrap_value=True, traj_info_kwargs=dict(discount=0.9))
    torch.cuda.set_device(0)
    agent.to_device(0)
    got_infos, ref_infos = [], []
    try:
        for itr in range(n_batches):
            smp, infos = s.obtain_samples(itr)
            torch.cuda.synchronize()
            assert smp.env.observation.is_cuda and smp.agent.action.is_cuda
            k = f"{name}{itr}_"
            assert np.array_equal(C.obs_crc(_np(smp.env.observation)), g[k + "obs_crc"]), (itr, "obs")
            for field, got in [("reward", smp.env.reward), ("prev_reward", smp.env.prev_reward),
                               ("done", smp.env.done), ("action", smp.agent.action),
                               ("prev_action", smp.agent.prev_action),
                               ("game_score", smp.env.env_info.game_score),
                               ("traj_done", smp.env.env_info.traj_done)]:
                assert np.array_equal(_np(got), g[k + field]), (itr, field, _np(got), g[k + field])
            # the test policy's value is float arithmetic ((s % 97) / 97 + ...): the device's f32
            # division differs from the host's in the last bit -- 1 ulp, not a sampler property
            for field, got in [("value", smp.agent.agent_info.value),
                               ("bootstrap_value", smp.agent.bootstrap_value)]:
                np.testing.assert_allclose(_np(got), g[k + field], rtol=3e-7, atol=0, err_msg=field)
            got_infos += _traj_rows(infos)
            ref_infos += [tuple(r[:3]) + (round(r[3], 9),) for r in g[k + "traj_fields"].tolist()]
        _assert_product_path(s, fused=False)
    finally:
        s.shutdown()
    got_c, ref_c = Counter(got_infos), Counter(ref_infos)
    assert len(ref_infos) > 0 and not (ref_c - got_c), (ref_c - got_c)


@pytest.mark.parametrize("n_workers,n_groups,split", [(2, 2, False), (2, 1, False), (4, 2, True)])
def test_fused_rollout_kernels_reproduce_reference_atari_ff_batches(n_workers, n_groups, split):
    """The benchmarked chain itself (``sample_convs_kernel`` -> ``rollout_fc_kernel`` ->
    ``rollout_head_kernel`` in a captured graph, C serve loop, fused bootstrap tail) against the
    batches the reference's GpuSampler + AtariFfAgent recorded on CPU."""
    from rlpyt_amd import _lib
    from rlpyt_amd.agents.pg.atari import AtariFfAgent
    from rlpyt_amd.models.pg.atari_ff_model import AtariFfModel
    g = load_golden("sampler_ff")
    s = GpuSampler(RefSeededPong, C.FF_ENV_KWARGS, batch_T=C.FF_T, batch_B=C.B, n_workers=n_workers,
                   n_groups=n_groups, mid_batch_reset=True, max_decorrelation_steps=0,
                   split_workers=split)
    agent = AtariFfAgent()
    s.initialize(agent, seed=C.SEED, bootstrap_value=True, traj_info_kwargs=dict(discount=0.9))
    torch.cuda.set_device(0)
    agent.to_device(0)
    torch.manual_seed(C.FF_INIT_SEED)
    fresh = AtariFfModel(image_shape=(4, 104, 80), output_size=6)
    agent.load_state_dict(fresh.state_dict())
    C.ff_sharpen(agent.model)
    # same parameters as the reference's model (bit-identical initialisation: tests/test_models.py)
    assert np.array_equal(C.param_checksums(list(agent.parameters())), g["param_crc"])
    _lib.variant_reset()
    got_infos, ref_infos = [], []
    try:
        for itr in range(C.FF_BATCHES):
            agent.sample_mode(itr)
            smp, infos = s.obtain_samples(itr)
            torch.cuda.synchronize()
            k = f"ff{itr}_"
            assert np.array_equal(C.obs_crc(_np(smp.env.observation)), g[k + "obs_crc"]), (itr, "obs")
            for field, got in [("reward", smp.env.reward), ("prev_reward", smp.env.prev_reward),
                               ("done", smp.env.done), ("action", smp.agent.action),
                               ("prev_action", smp.agent.prev_action),
                               ("prob", smp.agent.agent_info.dist_info.prob),
                               ("game_score", smp.env.env_info.game_score),
                               ("traj_done", smp.env.env_info.traj_done)]:
                assert np.array_equal(_np(got), g[k + field]), (itr, field, _np(got), g[k + field])
            np.testing.assert_allclose(_np(smp.agent.agent_info.value), g[k + "value"],
                                       rtol=1e-4, atol=2e-6)
            np.testing.assert_allclose(_np(smp.agent.bootstrap_value), g[k + "bootstrap_value"],
                                       rtol=1e-4, atol=2e-6)
            got_infos += _traj_rows(infos)
            ref_infos += [tuple(r[:3]) + (round(r[3], 9),) for r in g[k + "traj_fields"].tolist()]
        _assert_product_path(s, fused=True)
        ran = {k for k, v in _lib.variant_counts().items() if v > 0}
        for kern in ("sample_convs_kernel", "rollout_fc_kernel", "rollout_head_kernel"):
            assert any(kern in k for k in ran), (kern, sorted(ran))
    finally:
        s.shutdown()
    got_c, ref_c = Counter(got_infos), Counter(ref_infos)
    assert len(ref_infos) > 0 and not (ref_c - got_c), (ref_c - got_c)
