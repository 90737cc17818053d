"""One whole reference iteration at the BENCHMARKED size: the unmodified
reference ``PPO.optimize_agent`` (rlpyt/algos/pg/ppo.py:59-115) + ``AtariFfAgent``, imported from
``oracle/_ref`` (tests may), runs on the host cores over one seeded ``[T=128, B=256]`` batch -- 4
epochs x 4 minibatches of M = 8192, SGD (linear in the gradient: no sign-like amplification of
round-off), fixed shuffle seed -- and the PRODUCT runs the same iteration on the device from
bit-identical initial parameters, through the kernels the bench line is measured on.

Tolerances (written here, as DESIGN section 2 states them): the first minibatch's loss / entropy /
perplexity rtol 2e-5; its gradNorm -- a norm over 1.79 M gradient entries, each an f32 sum over
8192 x up to 475 terms accumulated in a different order by the host's convolution library and by the
device kernels -- is held to a FLOAT64 statement of the same minibatch instead: the product may be at
most twice as far from float64 as the reference itself is (and within 1e-4 of the reference; measured
3.4e-5, reference vs float64 of the same order); every later update's diagnostics and all parameters
after the 16 updates rtol 2e-4 (the ``ppo_sgd`` tolerance: conv / GEMM reductions reorder f32 sums).

The reference side runs in its own process (its modules never mix with this suite's) and leaves an
``.npz`` in the test's tmp dir; nothing is read from /root/reference.
"""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
HAVE_REF = os.path.isfile(os.path.join(ROOT, "oracle", "_ref", "rlpyt", "__init__.py"))

T, B = 128, 256
KW = dict(discount=0.99, learning_rate=2e-2, value_loss_coeff=1., entropy_loss_coeff=0.01,
          clip_grad_norm=1., gae_lambda=0.98, minibatches=4, epochs=4, ratio_clip=0.1,
          linear_lr_schedule=True, normalize_advantage=False)
BATCH_SEED = 79

REFERENCE_SIDE = """
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {golden!r})
from oracle import ref_runner
assert ref_runner.load()
import numpy as np, torch
import algo_cases as C
from rlpyt.agents.pg.atari import AtariFfAgent
from rlpyt.agents.pg.base import AgentInfo
from rlpyt.algos.pg.ppo import PPO
from rlpyt.envs.base import EnvSpaces
from rlpyt.samplers.collections import AgentSamplesBsv, BatchSpec, EnvSamples, Samples
from rlpyt.spaces.int_box import IntBox
from rlpyt.distributions.categorical import DistInfo
torch.set_num_threads({threads})
T, B = {T}, {B}
spaces = EnvSpaces(observation=IntBox(0, 256, shape=(4, 104, 80), dtype="uint8"), action=IntBox(0, C.A))
inp = C.batch_inputs(T, B, seed={seed})
torch.manual_seed(C.INIT_SEED)
agent = AtariFfAgent()
agent.initialize(spaces)
obs = inp["observation"]
prev_action, action = inp["all_action"][:-1], inp["all_action"][1:]
prev_reward, reward = inp["all_reward"][:-1], inp["all_reward"][1:]
prob, value = [], []
with torch.no_grad():               # the behaviour policy = the reference agent's own initial one
    for t0 in range(0, T, 16):
        d, v = agent(obs[t0:t0 + 16], prev_action[t0:t0 + 16], prev_reward[t0:t0 + 16])
        prob.append(d.prob); value.append(v)
    _, bv = agent(obs[-1], action[-1], reward[-1])
    bv = (bv + 0.25).unsqueeze(0)
prob, value = torch.cat(prob), torch.cat(value)
samples = Samples(
    agent=AgentSamplesBsv(action=action, prev_action=prev_action,
                          agent_info=AgentInfo(dist_info=DistInfo(prob=prob), value=value),
                          bootstrap_value=bv),
    env=EnvSamples(observation=obs, reward=reward, prev_reward=prev_reward, done=inp["done"], env_info=()))
kw = dict({kw!r}, OptimCls=torch.optim.SGD)
algo = PPO(**kw)
algo.initialize(agent=agent, n_itr=4, batch_spec=BatchSpec(T, B), mid_batch_reset=True,
                examples=None, world_size=1, rank=0)
np.random.seed(C.SHUFFLE_SEED)
agent.train_mode(0)
info = algo.optimize_agent(0, samples)            # rlpyt/algos/pg/ppo.py:59-115, unmodified
out = dict(old_prob=prob.numpy(), old_value=value.numpy(), bootstrap_value=bv.numpy(),
           obs_sum=np.int64(int(obs.to(torch.int64).sum())))
for f in ("loss", "gradNorm", "entropy", "perplexity"):
    out[f] = np.array(getattr(info, f), dtype=np.float64)
for n, p in agent.model.named_parameters():
    out["param__" + n] = p.detach().numpy()
np.savez({out!r}, **out)
print("REFERENCE_ITERATION_DONE", algo.update_counter)
"""


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref absent (python oracle/make_ref.py where "
                                         "/root/reference exists)")
def test_reference_ppo_iteration_at_bench_size_matches_product(tmp_path):
    from rlpyt_amd.utils.misc import usable_cpus
    out_path = os.path.join(str(tmp_path), "ref_iteration.npz")
    threads = max(1, min(int(usable_cpus()), 32))
    code = textwrap.dedent(REFERENCE_SIDE).format(root=ROOT, golden=GOLDEN, threads=threads, T=T, B=B,
                                                  seed=BATCH_SEED, kw=KW, out=out_path)
    env = dict(os.environ, PYTHONPATH="", OMP_NUM_THREADS=str(threads))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1500,
                         cwd=str(tmp_path), env=env)
    assert res.returncode == 0 and "REFERENCE_ITERATION_DONE 16" in res.stdout, \
        res.stdout[-3000:] + res.stderr[-3000:]
    g = np.load(out_path)

    # ---- the product, same batch, same seeds, on the device --------------------------------
    sys.path.insert(0, GOLDEN)
    import algo_cases as C
    from rlpyt_amd import _lib
    from rlpyt_amd.agents.pg.atari import AtariFfAgent
    from rlpyt_amd.agents.pg.categorical import AgentInfo
    from rlpyt_amd.algos.pg.ppo import PPO
    from rlpyt_amd.distributions.categorical import DistInfo
    from rlpyt_amd.envs import EnvSpaces
    from rlpyt_amd.samplers.collections import AgentSamplesBsv, BatchSpec, EnvSamples, Samples
    from rlpyt_amd.spaces import IntBox
    spaces = EnvSpaces(observation=IntBox(0, 256, shape=(4, 104, 80), dtype="uint8"),
                       action=IntBox(0, C.A))
    inp = C.batch_inputs(T, B, seed=BATCH_SEED)
    assert int(inp["observation"].to(torch.int64).sum()) == int(g["obs_sum"])      # same bytes
    torch.manual_seed(C.INIT_SEED)
    agent = AtariFfAgent()
    agent.initialize(spaces)
    agent.to_device(0)
    dev = lambda x: torch.as_tensor(x).cuda()  # noqa: E731
    all_action, all_reward = dev(inp["all_action"]), dev(inp["all_reward"])
    obs = dev(inp["observation"])
    # the behaviour policy the reference recorded is this agent's own initial policy
    with torch.no_grad():
        for t0 in range(0, T, 32):
            pi0, v0 = agent(obs[t0:t0 + 32], None, None)
            np.testing.assert_allclose(pi0.prob.cpu().numpy(), g["old_prob"][t0:t0 + 32], rtol=1e-4,
                                       atol=2e-6)
            np.testing.assert_allclose(v0.cpu().numpy(), g["old_value"][t0:t0 + 32], rtol=1e-4,
                                       atol=2e-5)
    samples = Samples(
        agent=AgentSamplesBsv(
            action=all_action[1:], prev_action=all_action[:-1],
            agent_info=AgentInfo(dist_info=DistInfo(prob=dev(g["old_prob"])), value=dev(g["old_value"])),
            bootstrap_value=dev(g["bootstrap_value"])),
        env=EnvSamples(observation=obs, reward=all_reward[1:], prev_reward=all_reward[:-1],
                       done=dev(inp["done"]), env_info=()))
    algo = PPO(**dict(KW, OptimCls=torch.optim.SGD))
    algo.initialize(agent=agent, n_itr=4, batch_spec=BatchSpec(T, B), mid_batch_reset=True,
                    examples=None, world_size=1, rank=0)
    # ---- float64 arbiter of the FIRST minibatch (initial parameters): its gradient norm ---------
    import copy

    from rlpyt_amd.utils.misc import iterate_mb_idxs
    from test_bench_path_gpu import _f64_loss
    np.random.seed(C.SHUFFLE_SEED)
    idx0 = torch.from_numpy(next(iter(iterate_mb_idxs(T * B, T * B // 4, shuffle=True)))).cuda()
    ret_, adv_, _valid = algo.process_returns(samples)
    t_i, b_i = idx0 % T, idx0 // T
    m64 = copy.deepcopy(agent.model).double()
    m64.zero_grad(set_to_none=True)
    _f64_loss(m64, obs[t_i, b_i], dev(g["old_prob"])[t_i, b_i].double(), all_action[1:][t_i, b_i],
              adv_[t_i, b_i].double(), ret_[t_i, b_i].double())
    norm64 = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m64.parameters())))
    del m64
    np.random.seed(C.SHUFFLE_SEED)
    agent.train_mode(0)
    _lib.variant_reset()
    info = algo.optimize_agent(0, samples)
    torch.cuda.synchronize()
    ran = {k for k, v in _lib.variant_counts().items() if v > 0}
    for k in ("gemm_nt_x6_kernel", "gemm_tn_x6_kernel", "convs_fwd_fused_kernel",
              "conv2_bwd_x6_kernel", "conv1_wgrad_kernel", "ppo_head_loss_kernel", "scan_exact_kernel"):
        assert any(k in r for r in ran), (k, sorted(ran))
    assert algo.update_counter == 16
    report = []
    for f in ("loss", "gradNorm", "entropy", "perplexity"):
        got, ref = np.array(getattr(info, f), dtype=np.float64), g[f]
        assert got.shape == ref.shape == (16,)
        rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-12)
        report.append(f"{f}: first-minibatch rel diff {rel[0]:.2e}, max over the 16 updates {rel.max():.2e}")
        if f == "gradNorm":
            e_prod, e_ref = abs(got[0] - norm64) / norm64, abs(ref[0] - norm64) / norm64
            report.append(f"gradNorm of the first minibatch vs float64 ({norm64:.9g}): product {e_prod:.2e}, "
                          f"reference {e_ref:.2e}")
            assert e_prod <= max(2. * e_ref, 2e-5), report[-1]
            np.testing.assert_allclose(got[0], ref[0], rtol=1e-4, err_msg="gradNorm (first minibatch)")
        else:
            np.testing.assert_allclose(got[0], ref[0], rtol=2e-5, atol=1e-7, err_msg=f + " (first minibatch)")
        np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-6, err_msg=f)
    for n, p in agent.model.named_parameters():
        ref = g["param__" + n]
        got = p.detach().cpu().numpy()
        report.append(f"{n}: max |diff| {np.abs(got - ref).max():.2e} (max |p| {np.abs(ref).max():.2e})")
        np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-6, err_msg=n)
    print("\n".join(report))
