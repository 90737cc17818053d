"""CPU tests of the host-side mirror of the reference protocol: namedarraytuple / buffer
contract, models vs the golden state-dict layout, the HBM-layout sampler
driven on CPU tensors (serial and forked workers), and the runner loop.

The product algorithms have no CPU path (their arithmetic is HIP-only), so the runner tests
drive the loop with ``OraclePPO`` -- a TEST-ONLY subclass that swaps the two HIP touch
points for the CPU oracle."""
import os
import numpy as np
import pytest
import torch

from oracle import np_oracle as O
from rlpyt_amd.agents.pg.atari import AtariFfAgent, MlpCategoricalPgAgent
from rlpyt_amd.algos.pg.ppo import PPO
from rlpyt_amd.envs.synthetic import SyntheticPong, TinyDiscreteEnv
from rlpyt_amd.samplers.collections import AtariTrajInfo
from rlpyt_amd.samplers.gpu import GpuSampler
from rlpyt_amd.utils import logger
from rlpyt_amd.utils.buffer import (buffer_from_example, buffer_method, buffer_to,
                                    get_leading_dims, numpify_buffer, torchify_buffer)
from rlpyt_amd.utils.collections import namedarraytuple

logger.set_quiet(True)


# ---------------------------------------------------------------- namedarraytuple / buffers
def test_namedarraytuple_contract():
    P = namedarraytuple("P", ["x", "y"])
    p = P(np.arange(4), np.arange(4) * 10)
    assert p[1] == P(1, 10) and isinstance(p[1:3], P) and list(p[1:3].y) == [10, 20]
    assert "x" in p and "z" not in p
    assert np.array_equal(p.get(0), np.arange(4))
    assert dict(p.items()).keys() == {"x", "y"}
    p[0] = 7
    assert p.x[0] == 7 and p.y[0] == 7
    p[1] = P(3, 30)
    assert p.x[1] == 3 and p.y[1] == 30
    Q = namedarraytuple("Q", ["a", "p"])
    q = Q(np.zeros(3), P(np.zeros(3), None))
    q[2] = Q(1, P(2, None))          # nested, None fields respected
    assert q.a[2] == 1 and q.p.x[2] == 2 and q.p.y is None
    assert q[0].p.y is None
    with pytest.raises(ValueError):
        namedarraytuple("Bad", ["get"])
    with pytest.raises(Exception, match="field 'x'"):
        p[10]


def test_buffer_helpers():
    Ex = namedarraytuple("Ex", ["obs", "r"])
    ex = Ex(np.zeros((4, 3), dtype=np.uint8), np.float32(0))
    buf = buffer_from_example(ex, (5, 2))
    assert buf.obs.shape == (5, 2, 4, 3) and buf.obs.dtype == np.uint8 and buf.r.dtype == np.float32
    shared = buffer_from_example(ex, (5, 2), share_memory=True)
    assert shared.obs.shape == (5, 2, 4, 3) and not shared.obs.any()
    pyt = torchify_buffer(buf)
    pyt.r[1, 1] = 3.
    assert buf.r[1, 1] == 3.            # zero-copy view
    assert get_leading_dims(buf, 2) == (5, 2)
    assert numpify_buffer(pyt).r is not None
    moved = buffer_to(pyt, device="cpu")
    assert moved.obs.shape == pyt.obs.shape
    assert buffer_method(pyt, "float").obs.dtype == torch.float32
    with pytest.raises(TypeError):
        buffer_to(buf, device="cpu")    # numpy leaves cannot move
    dev = buffer_from_example(ex, (2,), device="cpu")
    assert isinstance(dev.obs, torch.Tensor) and dev.obs.dtype == torch.uint8


# -------------------------------------------------------------------------------- sampler
@pytest.mark.parametrize("n_workers", [0, 2])
def test_sampler_layout_and_determinism(n_workers):
    def run(seed):
        s = GpuSampler(SyntheticPong, dict(points_to_end=1), batch_T=6, batch_B=4,
                       n_workers=n_workers, TrajInfoCls=AtariTrajInfo,
                       max_decorrelation_steps=0)
        a = AtariFfAgent()
        torch.manual_seed(seed)
        np.random.seed(seed)
        s.initialize(a, seed=seed, bootstrap_value=True)
        torch.manual_seed(seed + 1)
        out = []
        for itr in range(3):
            smp, infos = s.obtain_samples(itr)
            out.append((smp.agent.action.clone(), smp.env.reward.clone(),
                        smp.env.observation.clone(), smp.env.done.clone()))
        # layout contract
        assert smp.env.observation.shape == (6, 4, 4, 104, 80)
        assert smp.env.observation.dtype == torch.uint8 and smp.env.done.dtype == torch.bool
        assert smp.agent.action.dtype == torch.int64 and smp.env.reward.dtype == torch.float32
        assert smp.agent.agent_info.dist_info.prob.shape == (6, 4, 6)
        assert smp.agent.bootstrap_value.shape == (1, 4)
        # prev_* are the shifted views of the same [T+1,B] storage
        assert torch.equal(smp.agent.prev_action[1:], smp.agent.action[:-1])
        assert torch.equal(smp.env.prev_reward[1:], smp.env.reward[:-1])
        assert smp.agent.prev_action.data_ptr() + 4 * 8 == smp.agent.action.data_ptr()
        s.shutdown()
        return out
    a, b = run(5), run(5)
    for x, y in zip(a, b):
        for u, v in zip(x, y):
            assert torch.equal(u, v)       # same seed => same batches
    c = run(6)
    assert not torch.equal(a[-1][0], c[-1][0]) or not torch.equal(a[-1][2], c[-1][2])


def test_sampler_batches_are_contiguous_in_time():
    """Observation at row 0 of a batch is what the env returned after the last action of the
    previous batch (no step is skipped or duplicated across batches)."""
    s = GpuSampler(SyntheticPong, dict(points_to_end=100, max_steps=10 ** 6), batch_T=4,
                   batch_B=2, n_workers=0, max_decorrelation_steps=0)
    a = AtariFfAgent()
    s.initialize(a, seed=1, bootstrap_value=True)
    smp, _ = s.obtain_samples(0)
    last = smp.env.observation[-1].clone()
    smp, _ = s.obtain_samples(1)
    # frame stack shifts by one per step: newest 3 frames of the old obs are the oldest 3
    # of the obs two steps later only if the steps are consecutive
    assert torch.equal(smp.env.observation[0][:, :3], last[:, 1:])
    s.shutdown()


@pytest.mark.parametrize("n_workers,n_groups", [(2, 2), (0, 2), (3, 1), (4, 2)])
def test_sampler_pipeline_groups_keep_trajectories_contiguous(n_workers, n_groups):
    """Pipeline groups are column ranges of the same [T,B] batch: every column stays one
    env's contiguous trajectory, within a batch and across batches, and prev_* rows are the
    shifted action / reward rows."""
    s = GpuSampler(SyntheticPong, dict(points_to_end=100, max_steps=10 ** 6), batch_T=5,
                   batch_B=6, n_workers=n_workers, n_groups=n_groups, split_workers=True,
                   max_decorrelation_steps=0)
    a = AtariFfAgent()
    s.initialize(a, seed=2, bootstrap_value=True)
    assert s.n_groups == n_groups
    assert s.split_workers == (n_workers >= 2 * n_groups and n_groups > 1)   # dedicated workers
    smp, _ = s.obtain_samples(0)
    obs = smp.env.observation.clone()
    for t in range(4):     # frame stack shifts by exactly one frame per step, every column
        assert torch.equal(obs[t + 1][:, :3], obs[t][:, 1:])
    last_a, last_r = smp.agent.action[-1].clone(), smp.env.reward[-1].clone()
    smp, _ = s.obtain_samples(1)
    assert torch.equal(smp.env.observation[0][:, :3], obs[-1][:, 1:])
    assert torch.equal(smp.agent.prev_action[0], last_a)
    assert torch.equal(smp.env.prev_reward[0], last_r)
    assert not smp.env.done.any()
    s.shutdown()


@pytest.mark.parametrize("B,n_workers,want", [(256, 20, 4), (192, 19, 3), (512, 20, 4), (128, 16, 2),
                                              (16, 2, 2), (16, 0, 1), (30, 20, 1)])
def test_sampler_default_pipeline_groups(B, n_workers, want):
    """Default layout rule (no ``n_groups`` argument): groups of ~64 environments from B = 192 on, at most
    four; two while every worker still serves >= 2 environments; one without worker processes."""
    from rlpyt_amd.utils.collections import AttrDict
    s = GpuSampler(SyntheticPong, {}, batch_T=4, batch_B=B, n_workers=n_workers)
    s.batch_spec = AttrDict(T=4, B=B)
    s._resolve_layout(None)
    assert s.n_groups == want


@pytest.mark.parametrize("n_workers", [0, 2])
def test_sampler_wait_reset_blanks_rows_after_done(n_workers):
    """GpuWaitResetCollector semantics (collectors.py:70-126): once an env is done it
    records done=True and blank action / reward / agent_info / observation for the rest of
    the batch, and starts the next batch from a fresh reset."""
    s = GpuSampler(TinyDiscreteEnv, dict(size=5, horizon=3), batch_T=8, batch_B=4,
                   n_workers=n_workers, mid_batch_reset=False, max_decorrelation_steps=0)
    a = MlpCategoricalPgAgent()
    s.initialize(a, seed=0, bootstrap_value=True)
    for itr in range(2):
        smp, infos = s.obtain_samples(itr)
        done = smp.env.done.numpy()
        assert done[-1].all()                       # horizon 3 < T: every env finishes
        for b in range(4):
            first = int(np.argmax(done[:, b]))
            assert done[first:, b].all() and not done[:first, b].any()
            assert (smp.agent.action[first + 1:, b] == 0).all()
            assert (smp.env.reward[first + 1:, b] == 0).all()
            assert (smp.agent.agent_info.value[first + 1:, b] == 0).all()
            assert (smp.agent.agent_info.dist_info.prob[first + 1:, b] == 0).all()
            assert (smp.env.observation[first + 1:, b] == 0).all()
            assert smp.env.observation[0, b, 2] == 1.   # fresh reset obs (bias feature)
        assert len(infos) == 4
    s.shutdown()


# --------------------------------------------------------------------------------- runner
class OraclePPO(PPO):
    """TEST ONLY: PPO with the HIP touch points replaced by the CPU oracle, to exercise the
    host control flow (minibatch indexing, optimiser, schedules, opt-info) without a GPU."""

    def process_returns(self, samples):
        r, d = samples.env.reward.numpy(), samples.env.done.numpy()
        v, bv = samples.agent.agent_info.value.numpy(), samples.agent.bootstrap_value.numpy()
        adv, ret = O.generalized_advantage_estimation(r, v, d, bv, self.discount,
                                                      self.gae_lambda)
        return torch.from_numpy(ret), torch.from_numpy(adv), None

    def optimize_agent(self, itr, samples):
        import rlpyt_amd.algos.pg.ppo as mod
        real = mod.ops.gather_tb
        mod.ops.gather_tb = lambda src, idx, out=None: src[idx % src.shape[0],
                                                           idx // src.shape[0]]
        try:
            return super().optimize_agent(itr, samples)
        finally:
            mod.ops.gather_tb = real

    def loss(self, agent_inputs, action, return_, advantage, valid, old_prob,
             init_rnn_state=None):
        dist_info, value = self.agent(*agent_inputs)
        res = O.ppo_loss_torch(dist_info.prob, value, old_prob, action, advantage, return_,
                               valid, self.ratio_clip, self.value_loss_coeff,
                               self.entropy_loss_coeff)
        return res[0], torch.stack([x.detach() for x in res])


def test_runner_loop_cpu_plumbing():
    """BASELINE config #1: serial sampling + policy-gradient update on a tiny discrete env,
    CPU only; the policy must improve on the chain env."""
    from rlpyt_amd.runners.minibatch_rl import MinibatchRl
    sampler = GpuSampler(TinyDiscreteEnv, dict(), batch_T=16, batch_B=8, n_workers=0,
                         max_decorrelation_steps=0)
    algo = OraclePPO(learning_rate=3e-3, gae_lambda=0.95, minibatches=2, epochs=2,
                     linear_lr_schedule=False)
    agent = MlpCategoricalPgAgent()
    runner = MinibatchRl(algo=algo, agent=agent, sampler=sampler, n_steps=16 * 8 * 40, seed=0,
                         log_interval_steps=16 * 8 * 20)
    runner.train()
    assert algo.update_counter == 40 * 4
    assert runner.last_steps_per_second > 0
    # greedy action in the middle of the chain should be "right"
    obs = torch.tensor([[0., 1., 1.]])
    pi, _ = agent.model(obs, None, None)
    assert pi[0, 1] > 0.6


def _ddp_rank(rank, world_size, port, ret):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world_size))
    from rlpyt_amd.runners.minibatch_rl import SyncRl
    logger.set_quiet(True)
    sampler = GpuSampler(TinyDiscreteEnv, dict(), batch_T=8, batch_B=4, n_workers=0,
                         max_decorrelation_steps=0)
    algo = OraclePPO(learning_rate=1e-3, gae_lambda=0.95, minibatches=2, epochs=1,
                     linear_lr_schedule=False)
    agent = MlpCategoricalPgAgent()
    runner = SyncRl(algo=algo, agent=agent, sampler=sampler, n_steps=8 * 4 * 2 * 3, seed=3,
                    log_interval_steps=8 * 4 * 2 * 3, backend="gloo",
                    init_method=f"tcp://127.0.0.1:{port}")
    runner.train()
    params = torch.cat([p.detach().reshape(-1) for p in agent.parameters()])
    obs0 = sampler.samples.env.observation[0].clone()
    ret[rank] = (params, obs0, runner.world_size, runner.seed)
    dist.destroy_process_group()


def test_sync_rl_two_ranks_gloo():
    """N>1 path on CPU: 2 ranks, gloo.  Ranks see different data (seed + 100*rank, disjoint
    env ranks) but hold identical parameters after DDP-averaged updates."""
    import torch.multiprocessing as tmp
    mgr = tmp.Manager()
    ret = mgr.dict()
    tmp.spawn(_ddp_rank, args=(2, 29533, ret), nprocs=2, join=True)
    (p0, o0, w0, s0), (p1, o1, w1, s1) = ret[0], ret[1]
    assert w0 == w1 == 2 and s1 == s0 + 100
    assert torch.allclose(p0, p1, atol=0, rtol=0)
    assert not torch.equal(o0, o1)


def test_sampler_recurrent_agent_state_bookkeeping():
    """Recurrent agent under the sampler (CPU): the stored ``prev_rnn_state[t]`` is the LSTM state
    the agent entered step t with, so re-running the model over a column's observation sequence
    from ``prev_rnn_state[0]`` reproduces the recorded Q-values, across batches and through
    env resets (state zeroed where the previous step was done)."""
    from rlpyt_amd.agents.dqn.r2d1_agent import AtariR2d1Agent
    T, B = 5, 3
    s = GpuSampler(SyntheticPong, dict(points_to_end=1, max_steps=7), batch_T=T, batch_B=B,
                   n_workers=0, max_decorrelation_steps=0)
    a = AtariR2d1Agent(model_kwargs=dict(fc_size=32, lstm_size=8, head_size=16), eps_final=0.5)
    torch.manual_seed(0)
    s.initialize(a, seed=3, bootstrap_value=False)
    carry = None
    for itr in range(4):
        a.sample_mode(itr)
        smp, _ = s.obtain_samples(itr)
        info = smp.agent.agent_info
        assert info.prev_rnn_state.h.shape == (T, B, 1, 8) and info.q.shape == (T, B, 6)
        onehot = torch.nn.functional.one_hot(smp.agent.prev_action, 6).float()
        done_prev = torch.cat([torch.zeros(1, B, dtype=torch.bool) if carry is None else carry[None],
                               smp.env.done[:-1]])
        with torch.no_grad():
            for b in range(B):
                h = info.prev_rnn_state.h[0, b].unsqueeze(1)      # [N,1,H]
                c = info.prev_rnn_state.c[0, b].unsqueeze(1)
                for t in range(T):
                    if done_prev[t, b]:
                        assert float(info.prev_rnn_state.h[t, b].abs().max()) == 0.
                        h, c = torch.zeros_like(h), torch.zeros_like(c)
                    np.testing.assert_allclose(info.prev_rnn_state.h[t, b].numpy(),
                                               h[:, 0].numpy(), rtol=1e-5, atol=1e-6)
                    # null prev_action after a reset = action index 0 (action_server.py:50)
                    pa = onehot.new_tensor([1., 0, 0, 0, 0, 0]) if done_prev[t, b] else onehot[t, b]
                    pr = torch.zeros(()) if done_prev[t, b] else smp.env.prev_reward[t, b]
                    q, (h, c) = a.model(smp.env.observation[t, b], pa, pr, (h, c))
                    np.testing.assert_allclose(q.numpy(), info.q[t, b].numpy(), rtol=1e-4,
                                               atol=1e-5)
        carry = smp.env.done[-1].clone()
    s.shutdown()


@pytest.mark.parametrize("B,n_groups", [(4, 2), (5, 2)])
def test_recurrent_pg_bootstrap_value_uses_its_own_groups_state(B, n_groups):
    """with several pipeline groups the bootstrap value of a recurrent PG agent
    must come from the LSTM state of THAT group (action_server.py:60-62 has one state for all
    envs).  No env finishes here, so row 0 of the next batch holds obs_T and the state after T
    steps: re-running the model on them must reproduce the recorded bootstrap value."""
    from rlpyt_amd.agents.pg.atari import AtariLstmAgent
    T = 5
    s = GpuSampler(SyntheticPong, dict(points_to_end=10 ** 6, max_steps=10 ** 6), batch_T=T,
                   batch_B=B, n_workers=0, n_groups=n_groups, max_decorrelation_steps=0)
    a = AtariLstmAgent(model_kwargs=dict(fc_sizes=32, lstm_size=8))
    torch.manual_seed(0)
    s.initialize(a, seed=3, bootstrap_value=True)
    assert s.n_groups == n_groups
    prev = None
    for itr in range(3):
        a.sample_mode(itr)
        smp, _ = s.obtain_samples(itr)
        assert not bool(smp.env.done.any())
        info = smp.agent.agent_info
        if prev is not None:
            bv, last_action, last_reward = prev
            with torch.no_grad():
                onehot = torch.nn.functional.one_hot(last_action, 6).float()
                state = tuple(x[0].transpose(0, 1).contiguous() for x in info.prev_rnn_state)
                _pi, v, _st = a.model(smp.env.observation[0], onehot, last_reward, state)
            np.testing.assert_allclose(bv.numpy(), v.numpy(), rtol=1e-5, atol=1e-6)
            # and it is NOT what another group's state would give (the states differ)
            assert float((state[0][:, 0] - state[0][:, -1]).abs().max()) > 0
        prev = (smp.agent.bootstrap_value[0].clone(), smp.agent.action[-1].clone(),
                smp.env.reward[-1].clone())
    s.shutdown()


# ---------------------------------------------------------------------- offline evaluation
@pytest.mark.parametrize("n_workers", [0, 2])
def test_sampler_evaluate_agent(n_workers):
    """Sampler.evaluate_agent (rlpyt/samplers/parallel/base.py:115-145 +
    gpu/action_server.py:76-120): separate eval envs, stops on the step budget or on the
    trajectory budget, leaves the training envs and the training batch sequence untouched."""
    def make(eval_max_trajectories=None, eval_max_steps=4 * 60):
        s = GpuSampler(TinyDiscreteEnv, dict(), batch_T=5, batch_B=4, n_workers=n_workers,
                       max_decorrelation_steps=0, eval_n_envs=4,
                       eval_env_kwargs=dict(horizon=12), eval_max_steps=eval_max_steps,
                       eval_max_trajectories=eval_max_trajectories)
        a = MlpCategoricalPgAgent()
        torch.manual_seed(3)
        np.random.seed(3)
        s.initialize(a, seed=3, bootstrap_value=True)
        return s, a

    # reference batches without any evaluation in between
    s, a = make()
    torch.manual_seed(11)
    plain = []
    for itr in range(3):
        a.sample_mode(itr)
        smp, _ = s.obtain_samples(itr)
        plain.append((smp.env.observation.clone(), smp.env.reward.clone(), smp.env.done.clone()))
    s.shutdown()

    s, a = make()
    assert s.eval_n_envs == 4 and s.eval_max_T == 60
    a.eval_mode(0)
    infos = s.evaluate_agent(0)
    # horizon 12 => every eval env finishes >= 5 trajectories in 60 steps; none longer than 12
    assert len(infos) >= 4 * 5 and all(1 <= ti["Length"] <= 12 for ti in infos)
    assert all(isinstance(ti, s.TrajInfoCls) for ti in infos)
    torch.manual_seed(11)
    for itr in range(3):
        a.sample_mode(itr)
        smp, _ = s.obtain_samples(itr)
        if itr == 1:     # evaluation between training batches must not disturb them
            state = torch.get_rng_state()
            a.eval_mode(itr)
            assert len(s.evaluate_agent(itr)) >= 20
            torch.set_rng_state(state)
        # env-side data depends on the actions drawn, so equality also checks the RNG stream
        assert torch.equal(smp.env.observation, plain[itr][0])
        assert torch.equal(smp.env.reward, plain[itr][1])
        assert torch.equal(smp.env.done, plain[itr][2])
    s.shutdown()

    # trajectory budget: stops early (checked every EVAL_TRAJ_CHECK = 20 steps)
    s, a = make(eval_max_trajectories=6, eval_max_steps=4 * 400)
    a.eval_mode(0)
    infos = s.evaluate_agent(0)
    assert 6 <= len(infos) <= 4 * 400 // 2
    total = sum(ti["Length"] for ti in infos)
    assert total <= 4 * 41, "evaluation should have stopped at the first or second check"
    # and can be repeated (hand-off counters stay consistent after an early stop)
    assert len(s.evaluate_agent(1)) >= 6
    a.sample_mode(0)
    s.obtain_samples(0)
    s.shutdown()


def test_sampler_evaluate_requires_eval_envs():
    with pytest.raises(ValueError, match="eval_max_steps"):
        GpuSampler(TinyDiscreteEnv, dict(), batch_T=2, batch_B=2, n_workers=0,
                   eval_n_envs=2).initialize(MlpCategoricalPgAgent(), seed=0)
    s = GpuSampler(TinyDiscreteEnv, dict(), batch_T=2, batch_B=2, n_workers=0)
    s.initialize(MlpCategoricalPgAgent(), seed=0)
    with pytest.raises(RuntimeError, match="eval_n_envs"):
        s.evaluate_agent(0)
    s.shutdown()


def test_minibatch_rl_eval_runner():
    """MinibatchRlEval (rlpyt/runners/minibatch_rl.py:286-357): evaluates at itr 0 and at every
    log interval, logs eval trajectories instead of training ones."""
    from rlpyt_amd.runners.minibatch_rl import MinibatchRlEval
    sampler = GpuSampler(TinyDiscreteEnv, dict(), batch_T=8, batch_B=4, n_workers=0,
                         max_decorrelation_steps=0, eval_n_envs=2, eval_max_steps=200,
                         eval_max_trajectories=10)
    calls = []
    orig = sampler.evaluate_agent
    sampler.evaluate_agent = lambda itr: calls.append(itr) or orig(itr)
    algo = OraclePPO(learning_rate=1e-3, minibatches=2, epochs=1, linear_lr_schedule=False)
    runner = MinibatchRlEval(algo=algo, agent=MlpCategoricalPgAgent(), sampler=sampler,
                             n_steps=8 * 4 * 6, seed=0, log_interval_steps=8 * 4 * 3)
    runner.train()
    assert calls == [0, 2, 5]
    assert runner._cum_eval_time > 0 and algo.update_counter == 6 * 2


@pytest.mark.parametrize("name,kw,global_B,env_ranks", [
    ("scalar", dict(eps_init=1., eps_final=0.05), 4, [0, 1, 2, 3]),
    ("vector_rank1", dict(eps_init=1., eps_final=0.1, eps_final_min=0.001), 8, [4, 5, 6, 7])])
def test_epsilon_schedule_matches_reference(name, kw, global_B, env_ranks):
    """Epsilon annealing and the rank-aware vector epsilon (log-spaced over the global env index,
    rlpyt/agents/dqn/epsilon_greedy.py:47-63,96-106) against values recorded from the
    reference's AtariDqnAgent."""
    from conftest import load_golden
    from rlpyt_amd.agents.dqn.dqn_agent import AtariDqnAgent
    from rlpyt_amd.envs import EnvSpaces
    from rlpyt_amd.spaces import IntBox
    g = load_golden("agents")
    agent = AtariDqnAgent(**kw)
    agent.initialize(EnvSpaces(observation=IntBox(0, 256, shape=(4, 104, 80), dtype="uint8"),
                               action=IntBox(0, 6)), global_B=global_B, env_ranks=env_ranks)
    agent.set_epsilon_itr_min_max(2, 10)
    for itr in range(13):
        agent.sample_mode(itr)
        eps = np.broadcast_to(np.asarray(agent.distribution.epsilon, dtype=np.float64),
                              (len(env_ranks),))
        np.testing.assert_allclose(eps, g[f"{name}_sample_eps"][itr], rtol=1e-6, atol=0)
    for k, itr in enumerate((0, 5)):
        agent.eval_mode(itr)
        assert float(agent.distribution.epsilon) == g[f"{name}_eval_eps"][k]


# ------------------------------------------------------------------- host-side primitives
def test_seq_words_handoff_between_processes():
    """rlpyt_seq_post / wait / arrive (csrc/hostsync.cpp): the futex sequence words behind every
    master <-> worker hand-off.  No HIP call involved, so this runs without a GPU."""
    import ctypes
    import multiprocessing as mp
    import time
    from rlpyt_amd import _lib
    from rlpyt_amd.utils.buffer import np_mp_array
    lib = _lib.lib
    words = np_mp_array(64, np.uint32)          # word 0: master -> workers, word 16: arrivals
    words[:] = 0
    post = ctypes.c_void_p(words.ctypes.data)
    arr = ctypes.c_void_p(words.ctypes.data + 64)
    n_workers, rounds = 3, 50

    def worker():
        for r in range(1, rounds + 1):
            assert lib.rlpyt_seq_wait(post, r, 50, 5000) == 0
            lib.rlpyt_seq_arrive(arr, r * n_workers)

    ctx = mp.get_context("fork")
    procs = [ctx.Process(target=worker) for _ in range(n_workers)]
    for p in procs:
        p.start()
    for r in range(1, rounds + 1):
        lib.rlpyt_seq_post(post, r)
        assert lib.rlpyt_seq_wait(arr, r * n_workers, 100, 5000) == 0
        assert int(words[16]) == r * n_workers
    for p in procs:
        p.join(timeout=5)
        assert p.exitcode == 0
    # a wait nobody answers times out with RLPYT_ETIMEOUT instead of hanging
    t0 = time.perf_counter()
    assert lib.rlpyt_seq_wait(post, rounds + 1, 10, 60) == -5
    assert 0.05 <= time.perf_counter() - t0 < 1.0
    # sequence comparison is wrap-around safe: a target "behind" the word counts as reached
    words[0] = 5
    assert lib.rlpyt_seq_wait(post, 0xfffffff0, 10, 50) == 0      # (int32)(5 - 0xfffffff0) > 0
    words[0] = 0xfffffffe
    assert lib.rlpyt_seq_wait(post, 3, 10, 50) == -5              # 3 is ahead of 0xfffffffe


def test_usable_cpus_honours_limits(monkeypatch):
    import builtins
    import io
    import os
    from rlpyt_amd.utils import misc
    n = misc.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            return io.StringIO("250000 100000\n")
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, "open", fake_open)
    assert misc.usable_cpus() == min(2.5, n if n < 2.5 else 2.5)

    def unlimited(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            return io.StringIO("max 100000\n")
        if str(path).startswith("/sys/fs/cgroup/cpu/"):
            raise OSError
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, "open", unlimited)
    assert misc.usable_cpus() == float(min(os.cpu_count() or 1, len(os.sched_getaffinity(0))))


@pytest.mark.parametrize("which", ["train", "eval"])
def test_runner_diagnostics_match_reference(which):
    """The runners log the tabular diagnostics of the reference's MinibatchRl / MinibatchRlEval --
    same names, same order (tests/golden/runner_keys.json, recorded from the reference runners);
    ``Diagnostics/StepsPerSecond`` is the metric BASELINE.json quotes."""
    import json
    import os
    from rlpyt_amd.runners.minibatch_rl import MinibatchRl, MinibatchRlEval
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                           "runner_keys.json")) as f:
        ref = json.load(f)[which]
    skw = {} if which == "train" else dict(eval_n_envs=2, eval_max_steps=200,
                                           eval_max_trajectories=10)
    sampler = GpuSampler(TinyDiscreteEnv, dict(), batch_T=8, batch_B=4, n_workers=0,
                         max_decorrelation_steps=0, **skw)
    Runner = MinibatchRl if which == "train" else MinibatchRlEval
    runner = Runner(algo=OraclePPO(minibatches=2, epochs=1), agent=MlpCategoricalPgAgent(),
                    sampler=sampler, n_steps=8 * 4 * 6, seed=0, log_interval_steps=8 * 4 * 3)
    rows = []
    orig = logger.dump_tabular
    logger.dump_tabular = lambda *a, **k: (rows.append(list(logger.get_tabular())), orig(*a, **k))
    try:
        runner.train()
    finally:
        logger.dump_tabular = orig
    assert rows and rows[-1] == ref, (sorted(set(ref) ^ set(rows[-1])), rows[-1][:14], ref[:14])


def test_trunk_pre_activation_contract_on_cpu():
    """``AtariFfModel.forward(features_only="pre")`` / ``CategoricalPgAgent.trunk_pre`` /
    ``ops.linear_nobias`` off the device: where the fused path does not apply (CPU tensors) the
    model hands out its ordinary trunk output and ``None`` for the bias (the caller then adds
    nothing), and ``linear_nobias`` is ``F.linear`` -- the same statements as the reference's
    Linear + ReLU (rlpyt/models/mlp.py:24-31)."""
    import torch.nn.functional as F

    from rlpyt_amd import ops
    from rlpyt_amd.models.pg.atari_ff_model import AtariFfModel
    torch.manual_seed(0)
    m = AtariFfModel(image_shape=(4, 104, 80), output_size=6)
    img = torch.randint(0, 256, (3, 4, 104, 80), dtype=torch.uint8)
    z, tb = m(img, None, None, features_only="pre")
    assert tb is None                                   # CPU: bias + ReLU already applied
    h = m(img, None, None, features_only=True)
    assert torch.equal(z, h)
    pi, v = m(img, None, None)
    assert torch.allclose(pi, torch.softmax(m.pi(h), -1)) and torch.allclose(v, m.value(h).squeeze(-1))
    lin = m._single_fc()
    assert lin is not None and lin.in_features == 3456 and lin.out_features == 512
    x, w = torch.randn(5, 64), torch.randn(32, 64)
    assert torch.equal(ops.linear_nobias(x, w), F.linear(x, w))   # CPU tensors: library path


def test_sampler_worker_dies_with_its_parent():
    """A forked worker waits for actions without a timeout; when the master process is killed the
    kernel must take the worker down (``_die_with_parent``), or a crashed run leaves hung workers
    behind (and hangs any launcher that waits for all children)."""
    import signal
    import subprocess
    import sys
    import time
    code = (
        "import multiprocessing as mp, sys, time\n"
        f"sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})\n"
        "from rlpyt_amd.samplers.gpu import _die_with_parent\n"
        "def child():\n"
        "    _die_with_parent()\n"
        "    time.sleep(120)\n"
        "if __name__ == '__main__':\n"
        "    p = mp.get_context('fork').Process(target=child); p.start()\n"
        "    print(p.pid, flush=True)\n"
        "    time.sleep(120)\n")
    pr = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True)
    pid = int(pr.stdout.readline())
    os.kill(pr.pid, signal.SIGKILL)
    pr.wait()
    for _ in range(50):
        try:
            with open(f"/proc/{pid}/status") as f:
                state = f.read().split("State:")[1].split()[0]
        except (FileNotFoundError, ProcessLookupError):
            state = "gone"
        if state in ("gone", "Z", "X"):
            break
        time.sleep(0.1)
    assert state in ("gone", "Z", "X"), f"worker {pid} still {state} after its parent was killed"


# ------------------------------------------------------------------- round-4 host structure
def test_runner_schedule_rounds_up_to_whole_logging_periods():
    """``schedule`` (runners/minibatch_rl.py) == the reference's get_n_itr arithmetic
    (rlpyt/runners/minibatch_rl.py:108-118): iterations rounded UP to a multiple of the logging
    period, at least one period of at least one iteration."""
    from rlpyt_amd.runners.minibatch_rl import schedule

    def reference(n_steps, itr_batch, log_steps):
        log_itrs = max(log_steps // itr_batch, 1)
        n_itr = n_steps // itr_batch
        if n_itr % log_itrs > 0:
            n_itr += log_itrs
            n_itr -= n_itr % log_itrs
        return max(n_itr, 1), log_itrs
    for n_steps, itr_batch, log_steps in [(1000, 32, 100), (32768 * 25, 32768, 10 ** 5), (5, 32, 100),
                                          (960, 32, 96), (961, 32, 96), (10 ** 6, 512, 10 ** 4),
                                          (100, 32, 10)]:
        plan = schedule(n_steps, itr_batch, log_steps)
        assert (plan.n_itr, plan.log_every) == reference(n_steps, itr_batch, log_steps)
        assert plan.n_itr % plan.log_every == 0 or plan.n_itr == 1


def test_update_plan_beta_anneal_and_replay_feed():
    """The DQN family's iteration arithmetic as data (algos/dqn/replay_algo.py) against the
    reference's formulas (rlpyt/algos/dqn/dqn.py:84-100,267-279)."""
    from collections import namedtuple
    from rlpyt_amd.algos.dqn.replay_algo import BetaAnneal, ReplayFeed, UpdateLog, plan_updates
    p = plan_updates(sampler_batch=32, train_batch=128, replay_ratio=8, min_steps_learn=5000,
                     eps_steps=10 ** 6, pri_beta_steps=5 * 10 ** 7)
    assert p.updates_per_itr == max(1, round(8 * 32 / 128)) == 2
    assert p.first_learn_itr == 5000 // 32 and p.eps_last_itr == 10 ** 6 // 32
    assert p.beta_last_itr == 5 * 10 ** 7 // 32
    assert plan_updates(7680, 7680, 1, 0, 1, 1).updates_per_itr == 1
    b = BetaAnneal(0.4, 1.0, first_itr=10, last_itr=110)
    assert b.at(0) == b.at(10) == 0.4 and abs(b.at(60) - 0.7) < 1e-12 and b.at(110) == 1.0
    assert b.at(111) is None                     # past the anneal: the buffer keeps its beta
    # feed: sampler batch / examples dict -> replay record
    Env = namedtuple("Env", ["observation", "reward", "done"])
    Info = namedtuple("Info", ["q", "prev_rnn_state"])
    Agent = namedtuple("Agent", ["action", "agent_info"])
    Samples = namedtuple("Samples", ["agent", "env"])
    smp = Samples(agent=Agent(action="A", agent_info=Info(q="Q", prev_rnn_state="S")),
                  env=Env(observation="O", reward="R", done="D"))
    step, rnn = ReplayFeed(ReplayFeed.STEP, "SamplesToBuffer"), ReplayFeed(ReplayFeed.RNN, "SamplesToBufferRnn")
    assert tuple(step.from_samples(smp)) == ("O", "A", "R", "D")
    assert tuple(rnn.from_samples(smp)) == ("O", "A", "R", "D", "S")
    ex = dict(observation="o", action="a", reward="r", done="d", agent_info=Info(q="q", prev_rnn_state="s"))
    assert tuple(step.from_examples(ex)) == ("o", "a", "r", "d")
    assert tuple(rnn.from_examples(ex)) == ("o", "a", "r", "d", "s")
    assert type(rnn.from_samples(smp)).__name__ == "SamplesToBufferRnn"
    # update log: one host copy at the end, scalar rows + vector fields in OptInfo order
    OptInfo = namedtuple("OptInfo", ["loss", "gradNorm", "tdAbsErr"])
    log = UpdateLog(OptInfo, ("loss", "gradNorm"))
    assert log.to_opt_info() == OptInfo([], [], [])
    log.add((torch.tensor(1.5), torch.tensor(2.0)), tdAbsErr=torch.tensor([0.1, 0.2]))
    log.add((torch.tensor(0.5), torch.tensor(3.0)), tdAbsErr=torch.tensor([[0.3]]))
    out = log.to_opt_info()
    assert out.loss == [1.5, 0.5] and out.gradNorm == [2.0, 3.0]
    assert np.allclose(out.tdAbsErr, [0.1, 0.2, 0.3])


def test_epsilon_schedule_object():
    """``EpsilonSchedule`` (agents/dqn/dqn_agent.py): anneal, hold, evaluation rate, and the rank's
    slice of the global log-spaced ladder."""
    from rlpyt_amd.agents.dqn.dqn_agent import EpsilonSchedule
    s = EpsilonSchedule(1.0, 0.1, None, itr_min=2, itr_max=12, eval_eps=0.001)
    assert s.sampling(0) == s.sampling(2) == 1.0 and abs(s.sampling(7) - 0.55) < 1e-12
    assert abs(s.sampling(12) - 0.1) < 1e-12 and abs(s.sampling(500) - 0.1) < 1e-12      # held
    assert s.evaluation(0) == 1.0 and s.evaluation(3) == 0.001
    v = EpsilonSchedule(1.0, 0.1, 0.001, 0, 10, 0.001)
    v.spread_over_envs(global_B=8, env_ranks=[4, 5, 6, 7])
    ladder = torch.logspace(-3, -1, 8)
    assert torch.allclose(v.final, ladder[4:]) and torch.equal(v.init, torch.ones(4))
    assert torch.allclose(v.sampling(10), ladder[4:]) and torch.allclose(v.sampling(5), 0.5 + 0.5 * ladder[4:])


def test_replay_class_factory_and_signatures():
    """``replay_class`` maps the three switches onto the reference's eight class names, and the
    classes keep the reference's constructor conventions (prioritized: keyword-only tail)."""
    import inspect
    from rlpyt_amd.replays import buffers as R
    assert R.replay_class(True, False, True) is R.PrioritizedReplayFrameBuffer
    assert R.replay_class(True, True, True) is R.PrioritizedSequenceReplayFrameBuffer
    assert R.replay_class(False, False, False) is R.UniformReplayBuffer
    assert R.replay_class(True, True, False) is R.UniformSequenceReplayFrameBuffer
    seen = {R.replay_class(f, s, p) for f in (0, 1) for s in (0, 1) for p in (0, 1)}
    assert len(seen) == 8
    for cls in seen:
        assert (cls.FRAMES, cls.SEQUENCE, cls.PRIORITIZED) == tuple(
            w in cls.__name__ for w in ("Frame", "Sequence", "Prioritized"))
    sig = inspect.signature(R.PrioritizedReplayBuffer.__init__).parameters
    assert list(sig)[:7] == ["self", "alpha", "beta", "default_priority", "unique",
                             "input_priorities", "input_priority_shift"]


def test_replay_store_parts_on_host_tensors():
    """The storage parts of the replay (replays/store.py) exercised on CPU tensors (1-step returns, so
    no kernel is involved): ring writes with wrap anywhere, the frame store's layout contract
    (oldest frame of time r at row r, the first C - 1 rows mirror the last C - 1 after
    a wrap), cursor / full flag, and uniform draws staying outside the guard band."""
    from rlpyt_amd.replays.buffers import UniformReplayFrameBuffer
    from rlpyt_amd.utils.collections import namedarraytuple
    S2B = namedarraytuple("SamplesToBuffer", ["observation", "action", "reward", "done"])
    C, H, W, B, size = 4, 3, 2, 2, 20          # ring T = 10
    ex = S2B(observation=np.zeros((C, H, W), np.uint8), action=np.int64(0),
             reward=np.float32(0), done=np.bool_(False))
    buf = UniformReplayFrameBuffer(example=ex, size=size, B=B, discount=0.99, n_step_return=1,
                                   device="cpu")
    assert (buf.T, buf.B, buf.t, buf._buffer_full) == (10, 2, 0, False)
    assert (buf.off_backward, buf.off_forward) == (1, C - 1)
    rng = np.random.RandomState(0)
    stream = rng.randint(1, 255, size=(64 + C - 1, B, H, W)).astype(np.uint8)   # frame k of env b

    def obs_at(k):                   # observation at global time k: frames k .. k + C - 1
        return np.stack([stream[k + f] for f in range(C)], axis=1)              # [B, C, H, W]
    k = 0
    for T_new in (3, 4, 3, 5, 2, 7):          # 10 rows = exactly one lap, then wraps inside appends
        obs = np.stack([obs_at(k + i) for i in range(T_new)])                   # [T, B, C, H, W]
        rec = S2B(observation=torch.from_numpy(obs),
                  action=torch.arange(k, k + T_new).repeat(B, 1).t().contiguous(),
                  reward=torch.zeros(T_new, B), done=torch.zeros(T_new, B, dtype=torch.bool))
        t_before = buf.t
        T_ret, rows = buf.append_samples(rec)
        k += T_new
        assert T_ret == T_new and buf.t == k % buf.T
        assert buf._buffer_full == (k >= buf.T)
        frames = buf.samples_frames.numpy()
        for j in range(max(0, k - buf.T), k):          # every step still in the ring
            r = j % buf.T
            assert np.array_equal(frames[r + C - 1], stream[j + C - 1]), (k, j)   # its newest frame
            assert int(buf.samples.action[r, 0]) == j
        if buf.t < t_before or k == T_new:             # lap closed (or very first rows): history rows
            lap0 = (k // buf.T) * buf.T
            for f in range(C - 1):
                assert np.array_equal(frames[f], stream[lap0 + f]), (k, f)
    # one append of EXACTLY T rows from a non-zero start lands back on its start: still a closed lap,
    # the history rows must be refreshed.  (Intentional fix of a reference corner case:
    # rlpyt/replays/frame.py:57 tests the strict ``self.t < t`` and would leave them stale here.)
    assert buf.t == 4
    obs = np.stack([obs_at(k + i) for i in range(buf.T)])
    buf.append_samples(S2B(observation=torch.from_numpy(obs),
                           action=torch.arange(k, k + buf.T).repeat(B, 1).t().contiguous(),
                           reward=torch.zeros(buf.T, B), done=torch.zeros(buf.T, B, dtype=torch.bool)))
    k += buf.T
    assert buf.t == 4
    frames = buf.samples_frames.numpy()
    lap0 = (k // buf.T) * buf.T
    for f in range(C - 1):
        assert np.array_equal(frames[f], stream[lap0 + f]), ("full-lap append", f)
    # uniform draws never land in the guard band around the cursor
    np.random.seed(1)
    T_idxs, B_idxs = buf.sample_idxs(4000)
    t, b, f = buf.t, buf.off_backward, buf.off_forward
    banned = {(t - 1 - i) % buf.T for i in range(b)} | {(t + i) % buf.T for i in range(f)}
    assert set(T_idxs.tolist()) == set(range(buf.T)) - banned
    assert set(B_idxs.tolist()) == {0, 1}


def test_conv_pack_guard_follows_weight_versions():
    """``Conv2dModel._current_pack``: the packed copy of the conv weights is only handed
    to the fused kernels while the module is in eval mode AND no weight changed since the pack
    (storage address + in-place version counter)."""
    from rlpyt_amd.models.conv2d import Conv2dModel
    m = Conv2dModel(4, [32, 64, 64], [8, 4, 3], [4, 2, 1], paddings=[0, 1, 1])
    m._packed = torch.zeros(1)                      # stands for the device pack
    m._packed_versions = m._weight_versions()
    dev = m._packed.device
    m.eval()
    assert m._current_pack(dev) is m._packed
    m.train()
    assert m._current_pack(dev) is None             # training: always pack on the stream
    m.eval()
    with torch.no_grad():
        m.conv[2].weight.mul_(0.5)                  # in-place edit in eval mode
    assert m._current_pack(dev) is None
    m._packed_versions = m._weight_versions()
    assert m._current_pack(dev) is m._packed
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.load_state_dict(sd)                           # copy_ bumps the version counters
    assert m._current_pack(dev) is None


def test_pending_opt_info_reads_like_the_namedtuple():
    """``utils/deferred.PendingOptInfo``: what ``optimize_agent`` returns while the diagnostics' copy
    to the host is in flight has to serve every way a runner reads an ``OptInfo`` -- field attributes
    and ``getattr(opt_info, k, [])`` (rlpyt/runners/minibatch_rl.py:139-142), ``_fields``, iteration."""
    from collections import namedtuple
    from rlpyt_amd.algos.dqn.replay_algo import UpdateLog
    from rlpyt_amd.utils.deferred import PendingOptInfo, resolve
    Info = namedtuple("OptInfo", ["loss", "gradNorm", "tdAbsErr"])
    log = UpdateLog(Info, ("loss", "gradNorm"))
    assert log.to_opt_info() == Info([], [], [])
    log.add((torch.tensor(1.5), torch.tensor(2.5)), tdAbsErr=torch.tensor([1., 2.]))
    log.add_rows(torch.tensor([[3., 4.], [5., 6.]]), tdAbsErr=torch.tensor([[7.], [8.]]))
    info = log.to_opt_info()
    assert isinstance(info, PendingOptInfo) and info._fields == Info._fields
    assert getattr(info, "nope", []) == []
    assert info.loss == [1.5, 3., 5.] and info.gradNorm == [2.5, 4., 6.]
    assert info.tdAbsErr == [1., 2., 7., 8.]
    assert list(info) == [info.loss, info.gradNorm, info.tdAbsErr] and len(info) == 3
    assert resolve(info) == Info(info.loss, info.gradNorm, info.tdAbsErr)
    assert info._asdict() == resolve(info)._asdict() and resolve(resolve(info)) is resolve(info)
    # the product runner's log reads pending infos when it emits, not when it absorbs
    from rlpyt_amd.runners.minibatch_rl import RunLog
    rl = RunLog(Info._fields)
    log2 = UpdateLog(Info, ("loss", "gradNorm"))
    log2.add((torch.tensor(9.), torch.tensor(10.)), tdAbsErr=torch.tensor([11.]))
    rl.absorb([], log2.to_opt_info())
    rl.absorb([], Info([1.], [2.], [3.]))
    assert rl.opt["loss"] == [1.] and len(rl.waiting) == 1
    rl.settle()
    assert rl.opt["loss"] == [1., 9.] and rl.opt["tdAbsErr"] == [3., 11.] and not rl.waiting


def test_repeated_mode_call_of_one_iteration_is_a_no_op():
    """Runner and sampler both call ``agent.sample_mode(itr)`` (minibatch_rl.py:237, gpu/sampler.py:33):
    the echo must not redo the per-phase work, a new iteration / another mode / a state-dict load must."""
    from rlpyt_amd.agents.dqn.dqn_agent import AtariDqnAgent
    from rlpyt_amd.envs.synthetic import SyntheticPong
    env = SyntheticPong()
    a = AtariDqnAgent(model_kwargs=dict(fc_sizes=16), eps_itr_max=10)
    a.initialize(env.spaces)
    calls = {"refresh": 0, "eps": 0}
    a._refresh_step_weights = lambda: calls.__setitem__("refresh", calls["refresh"] + 1)
    set_eps = a.distribution.set_epsilon
    a.distribution.set_epsilon = lambda e: (calls.__setitem__("eps", calls["eps"] + 1), set_eps(e))
    a.sample_mode(3)
    assert calls == {"refresh": 1, "eps": 1} and not a.model.training
    assert all(not m.training for m in a.model.modules())
    a.sample_mode(3)
    assert calls == {"refresh": 1, "eps": 1}
    a.sample_mode(4)
    assert calls == {"refresh": 2, "eps": 2}
    a.train_mode(4)
    assert all(m.training for m in a.model.modules()) and calls["refresh"] == 2
    a.sample_mode(4)                              # back from training: the weights changed
    assert calls == {"refresh": 3, "eps": 3} and all(not m.training for m in a.model.modules())
    a.load_state_dict(a.model.state_dict())
    a.sample_mode(4)
    assert calls["refresh"] == 4
    a.eval_mode(4)
    a.eval_mode(4)
    assert calls["refresh"] == 5
    # a flag somebody set through the module's own method is still brought back
    a.model.train()
    a.eval_mode(5)
    assert all(not m.training for m in a.model.modules())
