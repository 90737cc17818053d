"""Whole update iterations against the reference's own algorithms:
tests/golden/algos.npz holds, for PPO (two configurations) and A2C, what the reference's
``optimize_agent`` produced with its AtariFfAgent on CPU over two consecutive iterations on a fixed
sample batch -- per-update loss / gradNorm / entropy / perplexity and the parameters after each
iteration.  This repo's algorithms run the same iterations on the GPU (HIP scans, fused losses,
MFMA conv stack, fused Adam) from bit-identical initial parameters (same seed), the same shuffle
seed, and must land on the same numbers within fp32 tolerance."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import algo_cases as C  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", C.CASES, ids=[c[0] for c in C.CASES])
def test_iterations_match_reference(case):
    """Tolerances: the first update of the first iteration starts from identical parameters, so
    its diagnostics agree to 2e-5 relative.  Afterwards Adam moves every parameter by ~lr times
    the SIGN-like ratio m/sqrt(v): parameters whose gradient is at round-off level can step the
    other way than on the CPU, which shifts later diagnostics at the 1e-3 level and leaves a small
    fraction of parameters up to 2*lr*updates apart -- both bounded below."""
    from rlpyt_amd.agents.pg.atari import AtariFfAgent
    from rlpyt_amd.agents.pg.categorical import AgentInfo
    from rlpyt_amd.algos.pg.a2c import A2C
    from rlpyt_amd.algos.pg.ppo import PPO
    from rlpyt_amd.distributions.categorical import DistInfo
    from rlpyt_amd.envs.base import EnvSpaces
    from rlpyt_amd.samplers.collections import AgentSamplesBsv, BatchSpec, EnvSamples, Samples
    from rlpyt_amd.spaces import IntBox
    name, algo_name, kwargs, mbr = case
    g = load_golden("algos")
    spaces = EnvSpaces(observation=IntBox(0, 256, shape=(4, 104, 80), dtype="uint8"),
                       action=IntBox(0, C.A))
    inp = C.batch_inputs()
    torch.manual_seed(C.INIT_SEED)
    agent = AtariFfAgent()
    agent.initialize(spaces)
    agent.to_device(0)
    dev = lambda x: torch.as_tensor(x).cuda()  # noqa: E731
    all_action, all_reward = dev(inp["all_action"]), dev(inp["all_reward"])
    samples = Samples(
        agent=AgentSamplesBsv(
            action=all_action[1:], prev_action=all_action[:-1],
            agent_info=AgentInfo(dist_info=DistInfo(prob=dev(g[f"{name}_old_prob"])),
                                 value=dev(g[f"{name}_old_value"])),
            bootstrap_value=dev(g[f"{name}_bootstrap_value"])),
        env=EnvSamples(observation=dev(inp["observation"]), reward=all_reward[1:],
                       prev_reward=all_reward[:-1], done=dev(inp["done"]), env_info=()))
    # the behaviour policy the reference recorded is this agent's own initial policy
    with torch.no_grad():
        pi0, v0 = agent(samples.env.observation, None, None)
    np.testing.assert_allclose(pi0.prob.cpu().numpy(), g[f"{name}_old_prob"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(v0.cpu().numpy(), g[f"{name}_old_value"], rtol=1e-4, atol=2e-5)
    algo = (PPO if algo_name == "PPO" else A2C)(**kwargs)
    algo.initialize(agent=agent, n_itr=C.N_ITR, batch_spec=BatchSpec(C.T, C.B),
                    mid_batch_reset=mbr, examples=None, world_size=1, rank=0)
    np.random.seed(C.SHUFFLE_SEED)
    lr = kwargs["learning_rate"]
    n_updates = 0
    for itr in range(C.N_RUN):
        agent.train_mode(itr)
        info = algo.optimize_agent(itr, samples)
        for f in ("loss", "gradNorm", "entropy", "perplexity"):
            got = np.atleast_1d(np.array(getattr(info, f), dtype=np.float64))
            ref = g[f"{name}_itr{itr}_{f}"]
            assert got.shape == ref.shape, (f, got.shape, ref.shape)
            if itr == 0:
                np.testing.assert_allclose(got[0], ref[0], rtol=2e-5, atol=1e-6, err_msg=f)
            np.testing.assert_allclose(got, ref, rtol=1e-2, atol=2e-3, err_msg=f"{f} itr {itr}")
        n_updates += len(np.atleast_1d(info.loss))
        params = list(agent.parameters())
        sums, abs_sums = C.param_stats([p.cpu() for p in params])
        np.testing.assert_allclose(abs_sums, g[f"{name}_itr{itr}_param_abs_sums"], rtol=2e-4)
        for n, p in agent.model.named_parameters():
            key = f"{name}_itr{itr}_param__{n}"
            if key not in g:
                continue
            diff = np.abs(p.detach().cpu().numpy() - g[key])
            assert diff.max() <= 2.2 * lr * n_updates, (n, diff.max())
            assert (diff > 0.1 * lr).mean() <= 0.05, (n, (diff > 0.1 * lr).mean())
    assert algo.update_counter == n_updates
