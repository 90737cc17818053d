"""Alternating samplers (rlpyt/samplers/parallel/gpu/alternating_sampler.py:6-83,
action_server.py:123-363): tests/golden/sampler_alt.npz holds every field of every batch the
REFERENCE's ``AlternatingSampler`` and ``NoOverlapAlternatingSampler`` produced (their own collectors
and action servers, on CPU, ``make_golden.py sampler_alt``) under the deterministic policy of
``sampler_cases``; this repo's samplers of the same names must reproduce them, with the reference's
constructor contract (even B, even worker count, alternating affinity, agent.alternating)."""
import os
import sys
from collections import Counter

import numpy as np
import pytest
import torch

from conftest import load_golden
from rlpyt_amd.samplers.alternating import AlternatingSampler, NoOverlapAlternatingSampler
from test_sampler_parity import DetAgent, RefSeededPong

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import sampler_cases as C  # noqa: E402

AFFINITY = dict(workers_cpus=[0, 1], cuda_idx=None, set_affinity=False, alternating=True)


@pytest.mark.parametrize("tag,Cls", [("alt", AlternatingSampler), ("noalt", NoOverlapAlternatingSampler)])
@pytest.mark.parametrize("case", C.CASES, ids=[c[0] for c in C.CASES])
def test_batches_match_reference_alternating_samplers(case, tag, Cls):
    name, mode, T, n_batches = case
    g = load_golden("sampler_alt")
    s = Cls(RefSeededPong, C.ENV_KWARGS, batch_T=T, batch_B=C.B, mid_batch_reset=(mode == "reset"),
            max_decorrelation_steps=0)
    agent = DetAgent()
    assert not agent.alternating
    s.initialize(agent, affinity=dict(AFFINITY), seed=C.SEED, bootstrap_value=True,
                 traj_info_kwargs=dict(discount=0.9))
    assert s.alternating and agent.alternating and s.n_workers == 2 and s.n_groups == 2
    assert s.split_workers and s.half_B == C.B // 2 and [G.n_workers for G in s.groups] == [1, 1]
    got_infos, ref_infos = [], []
    for itr in range(n_batches):
        smp, infos = s.obtain_samples(itr)
        k = f"{tag}_{name}{itr}_"
        assert np.array_equal(C.obs_crc(smp.env.observation.numpy()), g[k + "obs_crc"]), (itr, "obs")
        for field, got in [("reward", smp.env.reward), ("prev_reward", smp.env.prev_reward),
                           ("done", smp.env.done), ("action", smp.agent.action),
                           ("prev_action", smp.agent.prev_action), ("value", smp.agent.agent_info.value),
                           ("bootstrap_value", smp.agent.bootstrap_value),
                           ("game_score", smp.env.env_info.game_score),
                           ("traj_done", smp.env.env_info.traj_done)]:
            got = got.numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
            assert np.array_equal(got, g[k + field]), (itr, field, got, g[k + field])
        got_infos += [(float(ti["Length"]), float(ti["Return"]), float(ti["NonzeroRewards"]),
                       round(float(ti["DiscountedReturn"]), 9)) for ti in infos]
        ref_infos += [tuple(r[:3]) + (round(r[3], 9),) for r in g[k + "traj_fields"].tolist()]
    # (the reference drains its trajectory queue asynchronously: a sub-multiset of ours)
    got_c, ref_c = Counter(got_infos), Counter(ref_infos)
    assert len(ref_infos) > 0 and not (ref_c - got_c), (ref_c - got_c)
    s.shutdown()


def test_reference_alternating_batches_equal_its_gpu_sampler_batches():
    """What the recordings themselves say: under one fixed policy the alternating samplers return the
    batches of the plain GpuSampler (sampler.npz) -- alternation changes WHEN a half steps, not what
    a column holds."""
    a, b = load_golden("sampler_alt"), load_golden("sampler")
    for name, _mode, _T, n in C.CASES:
        for itr in range(n):
            for f in ("obs_crc", "reward", "done", "action", "value", "bootstrap_value"):
                for tag in ("alt", "noalt"):
                    assert np.array_equal(a[f"{tag}_{name}{itr}_{f}"], b[f"{name}{itr}_{f}"]), (tag, name, itr, f)


def test_alternating_constructor_contract():
    """alternating_sampler.py:25-27,30-38,60-67: even batch_B, alternating affinity, even worker count,
    recurrent agents must be 'alternating'."""
    with pytest.raises(AssertionError, match="even number"):
        AlternatingSampler(RefSeededPong, C.ENV_KWARGS, batch_T=3, batch_B=5)
    s = AlternatingSampler(RefSeededPong, C.ENV_KWARGS, batch_T=3, batch_B=4, max_decorrelation_steps=0)
    with pytest.raises(AssertionError, match="alternating affinity"):
        s.initialize(DetAgent(), affinity=dict(workers_cpus=[0, 1], set_affinity=False), seed=1)
    s = AlternatingSampler(RefSeededPong, C.ENV_KWARGS, batch_T=3, batch_B=4, max_decorrelation_steps=0)
    with pytest.raises(AssertionError, match="even number workers"):
        s.initialize(DetAgent(), affinity=dict(workers_cpus=[0, 1, 2], set_affinity=False,
                                               alternating=True), seed=1)

    class Rec(DetAgent):
        recurrent = True
    s = NoOverlapAlternatingSampler(RefSeededPong, C.ENV_KWARGS, batch_T=3, batch_B=4,
                                    max_decorrelation_steps=0)
    with pytest.raises(TypeError, match="alternating"):
        s.initialize(Rec(), affinity=dict(AFFINITY), seed=1)
    assert s.native_loop is False


@pytest.mark.parametrize("Cls", [AlternatingSampler, NoOverlapAlternatingSampler])
def test_alternating_four_workers_and_inline(Cls):
    """Four workers (two per half) and the inline layout (n_workers=0) give the serial batches too."""
    from rlpyt_amd.envs.synthetic import SyntheticPong
    from rlpyt_amd.samplers.gpu import GpuSampler

    def run(make):
        s = make()
        s.initialize(DetAgent(), seed=21, bootstrap_value=True,
                     **({} if s._n_workers_arg is not None else dict(affinity=dict(
                         workers_cpus=[0, 1, 2, 3], set_affinity=False, alternating=True))))
        out = []
        for itr in range(8):
            smp, _ = s.obtain_samples(itr)
            out.append([x.numpy().copy() for x in (smp.env.reward, smp.env.done, smp.agent.action,
                                               smp.agent.agent_info.value, smp.agent.bootstrap_value)]
                       + [C.obs_crc(smp.env.observation.numpy())])
        s.shutdown()
        return out
    kw = dict(batch_T=5, batch_B=6, max_decorrelation_steps=0)
    ref = run(lambda: GpuSampler(SyntheticPong, dict(points_to_end=1, max_steps=11), n_workers=0,
                                 n_groups=1, **kw))
    for make in (lambda: Cls(SyntheticPong, dict(points_to_end=1, max_steps=11), **kw),
                 lambda: Cls(SyntheticPong, dict(points_to_end=1, max_steps=11), n_workers=0, **kw)):
        got = run(make)
        for a, b in zip(ref, got):
            for x, y in zip(a, b):
                assert np.array_equal(x, y)
