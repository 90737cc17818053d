"""End-to-end GPU run of BASELINE config #3 at test size: GpuSampler (HBM batch, step graphs)
-> DQN with the HBM-resident prioritized frame replay (sum tree + frame gather kernels) ->
fused DQN loss; checks the data path, not learning."""
import numpy as np
import pytest
import torch

from rlpyt_amd.agents.dqn.dqn_agent import AtariDqnAgent
from rlpyt_amd.algos.dqn.dqn import DQN
from rlpyt_amd.envs.synthetic import SyntheticPong
from rlpyt_amd.samplers.gpu import GpuSampler
from rlpyt_amd.utils import logger

pytestmark = pytest.mark.gpu
logger.set_quiet(True)


@pytest.mark.parametrize("prioritized,double", [(True, True), (False, False)])
def test_dqn_end_to_end(prioritized, double):
    T, B = 4, 8
    sampler = GpuSampler(SyntheticPong, dict(points_to_end=1, max_steps=11), batch_T=T, batch_B=B,
                         n_workers=2, max_decorrelation_steps=0, eval_n_envs=4,
                         eval_max_steps=4 * 30, eval_max_trajectories=8)
    agent = AtariDqnAgent(eps_final=0.1)
    algo = DQN(batch_size=16, min_steps_learn=2 * T * B, replay_size=512, replay_ratio=8,
               target_update_interval=4, n_step_return=2, prioritized_replay=prioritized,
               double_dqn=double, learning_rate=1e-4)
    torch.manual_seed(0)
    np.random.seed(0)
    examples = sampler.initialize(agent, seed=1, bootstrap_value=False)
    torch.cuda.set_device(0)
    agent.to_device(0)
    algo.initialize(agent=agent, n_itr=12, batch_spec=sampler.batch_spec,
                    mid_batch_reset=sampler.mid_batch_reset, examples=examples)
    losses, stored = [], []
    for itr in range(10):
        agent.sample_mode(itr)
        samples, _ = sampler.obtain_samples(itr)
        stored.append(samples.env.observation[:, :, -1].clone())   # newest frame of every step
        agent.train_mode(itr)
        info = algo.optimize_agent(itr, samples)
        losses += list(info.loss)
    assert algo.update_counter == (10 - algo.min_itr_learn) * algo.updates_per_optimize > 0
    assert len(losses) == algo.update_counter and np.all(np.isfinite(losses))
    rb = algo.replay_buffer
    # the replay ring holds exactly the frames the sampler produced, in time order
    # (newest frame of time r sits at row r + C - 1, rlpyt/replays/frame.py:27-59)
    frames = torch.cat(stored)                      # [10*T, B, H, W]
    assert torch.equal(rb.samples_frames[3:3 + frames.shape[0]], frames)
    batch = rb.sample_batch(16)
    obs = batch.agent_inputs.observation
    assert obs.is_cuda and obs.dtype == torch.uint8 and obs.shape == (16, 4, 104, 80)
    assert batch.return_.shape == (16,) and torch.isfinite(batch.return_).all()
    if prioritized:
        assert batch.is_weights.shape == (16,) and float(batch.is_weights.max()) <= 1.0 + 1e-6
    # offline evaluation on the device (MinibatchRlEval path), then training continues
    agent.eval_mode(10)
    infos = sampler.evaluate_agent(10)
    assert len(infos) >= 8 and all(1 <= ti["Length"] <= 11 for ti in infos)
    agent.sample_mode(10)
    samples, _ = sampler.obtain_samples(10)
    assert torch.isfinite(samples.env.reward).all()
    sampler.shutdown()


@pytest.mark.parametrize("prioritized", [True, False])
def test_r2d1_end_to_end(prioritized):
    """BASELINE config #5 at test size: recurrent agent under the HBM sampler (LSTM state per
    pipeline group, captured step graphs), sequence replay with stored RNN states and input
    priorities, warm-up + training segments, fused R2D1 loss kernel."""
    from rlpyt_amd.agents.dqn.r2d1_agent import AtariR2d1Agent
    from rlpyt_amd.algos.dqn.r2d1 import R2D1
    T, B = 8, 4
    sampler = GpuSampler(SyntheticPong, dict(points_to_end=1, max_steps=13), batch_T=T, batch_B=B,
                         n_workers=2, mid_batch_reset=False, max_decorrelation_steps=0,
                         eval_n_envs=2, eval_max_steps=2 * 40)
    agent = AtariR2d1Agent(model_kwargs=dict(fc_size=64, lstm_size=32, head_size=32),
                           eps_final=0.1)
    algo = R2D1(batch_T=8, batch_B=6, warmup_T=8, store_rnn_state_interval=8,
                min_steps_learn=5 * T * B, replay_size=T * B * 16, n_step_return=2,
                target_update_interval=2, prioritized_replay=prioritized,
                input_priorities=prioritized, learning_rate=1e-4)
    torch.manual_seed(0)
    np.random.seed(0)
    examples = sampler.initialize(agent, seed=2, bootstrap_value=False)
    torch.cuda.set_device(0)
    agent.to_device(0)
    algo.initialize(agent=agent, n_itr=12, batch_spec=sampler.batch_spec,
                    mid_batch_reset=sampler.mid_batch_reset, examples=examples)
    losses, pris = [], []
    for itr in range(10):
        agent.sample_mode(itr)
        samples, _ = sampler.obtain_samples(itr)
        info = samples.agent.agent_info
        assert info.prev_rnn_state.h.shape == (T, B, 1, 32) and info.q.shape == (T, B, 6)
        # wait-reset: blank rows after done within a batch
        done = samples.env.done
        first = done.int().argmax(0)
        for b in range(B):
            if done[:, b].any():
                f = int(first[b])
                assert bool(done[f:, b].all()) and bool((samples.agent.action[f + 1:, b] == 0).all())
        agent.train_mode(itr)
        opt = algo.optimize_agent(itr, samples)
        losses += list(opt.loss)
        pris += list(opt.priority)
    assert algo.update_counter == (10 - algo.min_itr_learn) * algo.updates_per_optimize > 0
    assert len(losses) == algo.update_counter and np.all(np.isfinite(losses))
    assert len(pris) == algo.update_counter * 6 and np.all(np.isfinite(pris)) and min(pris) >= 0
    batch = algo.replay_buffer.sample_batch(6)
    assert batch.all_observation.shape == (16 + 2, 6, 4, 104, 80) and batch.all_observation.is_cuda
    assert batch.init_rnn_state.h.shape == (6, 1, 32)
    # recurrent offline evaluation keeps its own LSTM state; the sampling state survives it
    agent.eval_mode(10)
    infos = sampler.evaluate_agent(10)
    assert len(infos) >= 2 * 3 and all(ti["Length"] <= 13 for ti in infos)
    agent.sample_mode(10)
    samples, _ = sampler.obtain_samples(10)
    assert samples.agent.agent_info.prev_rnn_state.h.shape == (T, B, 1, 32)
    sampler.shutdown()


@pytest.mark.parametrize("prioritized,double,dueling", [(True, True, True), (False, False, False)])
def test_catdqn_end_to_end(prioritized, double, dueling):
    """Categorical DQN through the same HBM sampler / replay path: the
    agent_info leaf is the [A, n_atoms] distribution, priorities come from the KL kernel."""
    from rlpyt_amd.agents.dqn.catdqn_agent import AtariCatDqnAgent
    from rlpyt_amd.algos.dqn.cat_dqn import CategoricalDQN
    T, B = 4, 8
    sampler = GpuSampler(SyntheticPong, dict(points_to_end=1, max_steps=11), batch_T=T, batch_B=B,
                         n_workers=2, max_decorrelation_steps=0)
    agent = AtariCatDqnAgent(n_atoms=21, eps_final=0.1,
                             model_kwargs=dict(fc_sizes=64, dueling=dueling))
    algo = CategoricalDQN(V_min=-2, V_max=2, batch_size=16, min_steps_learn=2 * T * B,
                          replay_size=512, replay_ratio=8, target_update_interval=4,
                          n_step_return=2, prioritized_replay=prioritized, double_dqn=double,
                          learning_rate=1e-4)
    assert algo.optim_kwargs["eps"] == 0.01 / 16
    torch.manual_seed(0)
    np.random.seed(0)
    examples = sampler.initialize(agent, seed=1, bootstrap_value=False)
    torch.cuda.set_device(0)
    agent.to_device(0)
    algo.initialize(agent=agent, n_itr=12, batch_spec=sampler.batch_spec,
                    mid_batch_reset=sampler.mid_batch_reset, examples=examples)
    assert agent.distribution.z.is_cuda and float(agent.distribution.z[0]) == -2.0
    losses, kls = [], []
    for itr in range(8):
        agent.sample_mode(itr)
        samples, _ = sampler.obtain_samples(itr)
        p = samples.agent.agent_info.p
        assert p.shape == (T, B, 6, 21)
        torch.testing.assert_close(p.sum(-1), torch.ones_like(p[..., 0]), rtol=0, atol=1e-5)
        agent.train_mode(itr)
        info = algo.optimize_agent(itr, samples)
        losses += list(info.loss)
        kls += list(info.tdAbsErr)
    assert algo.update_counter == (8 - algo.min_itr_learn) * algo.updates_per_optimize > 0
    assert len(losses) == algo.update_counter and np.all(np.isfinite(losses))
    assert np.all(np.isfinite(kls)) and min(kls) >= 1e-6 * (1 - 1e-6)   # KL clamp floor
    # one more loss evaluation compared with the CPU oracle on the same network outputs
    from oracle import np_oracle as O
    batch = algo.replay_buffer.sample_batch(16)
    with torch.no_grad():
        ps = agent(*batch.agent_inputs)
        tps = agent.target(*batch.target_inputs)
        nps = agent(*batch.target_inputs) if double else None
    loss, kl = algo.loss(batch)
    c = lambda x: None if x is None else x.detach().cpu()  # noqa: E731
    loss_c, kl_c = O.cat_dqn_loss_torch(c(ps), c(tps), c(nps), c(batch.action), c(batch.return_),
                                        c(batch.done_n),
                                        c(batch.is_weights) if prioritized else None, None,
                                        -2, 2, algo.discount, algo.n_step_return)
    np.testing.assert_allclose(loss.item(), loss_c.item(), rtol=2e-5)
    np.testing.assert_allclose(kl.cpu().numpy(), kl_c.numpy(), rtol=1e-4, atol=1e-6)
    sampler.shutdown()


@pytest.mark.parametrize("vector_eps", [False, True])
def test_epsilon_schedule_reaches_captured_step_graphs(vector_eps):
    """The sampler captures the per-step device work of every pipeline group in a hipGraph during
    the first batch.  Epsilon must not be frozen into those graphs: with a schedule that drops to
    zero after iteration 0, every later action equals argmax(q) of the recorded Q-values; with a
    per-environment (vector) epsilon of {0, 1} exactly the eps = 0 environments act greedily,
    whichever pipeline group they fall in (rlpyt/agents/dqn/epsilon_greedy.py:47-89)."""
    T, B = 6, 8
    sampler = GpuSampler(SyntheticPong, dict(points_to_end=1, max_steps=50), batch_T=T, batch_B=B,
                         n_workers=2, n_groups=2, max_decorrelation_steps=0)
    agent = AtariDqnAgent(eps_init=1., eps_final=0., eps_itr_min=0, eps_itr_max=1)
    torch.manual_seed(1)
    sampler.initialize(agent, seed=5, bootstrap_value=False)
    torch.cuda.set_device(0)
    agent.to_device(0)
    agent.eps_itr_min, agent.eps_itr_max = 0, 1
    if vector_eps:
        agent.eps_final = torch.tensor([0., 1., 0., 0., 1., 1., 0., 1.])   # spans both groups
    greedy_frac = []
    for itr in range(4):
        agent.sample_mode(itr)
        samples, _ = sampler.obtain_samples(itr)
        q, act = samples.agent.agent_info.q, samples.agent.action
        same = (q.argmax(-1) == act)                      # [T, B]
        greedy_frac.append(same.float().mean().item())
        if itr >= 1:
            if vector_eps:
                zero = agent.eps_final == 0
                assert bool(same[:, zero.cuda()].all()), "eps = 0 environments must act greedily"
                assert not bool(same[:, (~zero).cuda()].all()), "eps = 1 environments explore"
            else:
                assert bool(same.all()), f"itr {itr}: epsilon 0 but {1 - greedy_frac[-1]:.2f} of the actions are random"
    assert greedy_frac[0] < 0.6        # iteration 0 ran with epsilon 1 (uniform over 6 actions)
    assert all(G.graph is not None for G in sampler.groups)
    sampler.shutdown()


@pytest.mark.parametrize("B,I,H", [(5, 531, 512), (64, 531, 512), (192, 519, 512), (3, 71, 32)])
def test_lstm_step_matches_nn_lstm(B, I, H):
    """``ops.LstmStep`` (split-K gate GEMM + ``rlpyt_lstm_cell_f32``) == one step of
    ``torch.nn.LSTM`` (rlpyt/models/dqn/atari_r2d1_model.py:61-63 at T = 1): against float64 on the
    CPU, no further off than 2x the library RNN's own f32 error (or 2e-6 of the output scale)."""
    from rlpyt_amd import ops
    g = torch.Generator().manual_seed(B + I)
    lstm = torch.nn.LSTM(I, H)
    for p_ in lstm.parameters():
        p_.data = torch.randn(p_.shape, generator=g) * 0.2
    x = torch.randn(B, I, generator=g)
    h = torch.randn(B, H, generator=g) * 0.5
    c = torch.randn(B, H, generator=g)
    ref = torch.nn.LSTM(I, H).double()
    ref.load_state_dict({k: v.double() for k, v in lstm.state_dict().items()})
    with torch.no_grad():
        _, (h64, c64) = ref(x.double()[None], (h.double()[None], c.double()[None]))
        dl = lstm.cuda()
        _, (hl, cl) = dl(x.cuda()[None], (h.cuda()[None].contiguous(), c.cuda()[None].contiguous()))
        step = ops.LstmStep(dl)
        cut = [I - 19, I - 1] if I > 40 else [I - 7, I - 1]
        parts = [x[:, :cut[0]].cuda(), x[:, cut[0]:cut[1]].cuda(), x[:, cut[1]:].cuda()]
        h1, c1 = step.step(parts, h.cuda(), c.cuda())
    for mine, lib_, want in ((h1, hl[0], h64[0]), (c1, cl[0], c64[0])):
        e_mine = (mine.double().cpu() - want).abs().max().item()
        e_lib = (lib_.double().cpu() - want).abs().max().item()
        assert e_mine <= max(2 * e_lib, 2e-6 * want.abs().max().item()), (e_mine, e_lib)
    from rlpyt_amd import _lib
    assert _lib.variant_counts().get("lstm_cell_kernel", 0) > 0


def test_lstm_step_weight_buffer_follows_the_parameters_through_a_graph():
    """The fused step reads its concatenated weight by address: after an in-place parameter update
    ``refresh()`` (what ``BaseAgent.sample_mode`` triggers) must make a CAPTURED step compute with the
    new weights, without re-capturing."""
    from rlpyt_amd import ops
    torch.manual_seed(3)
    lstm = torch.nn.LSTM(71, 32).cuda()
    step = ops.LstmStep(lstm)
    x, h, c = (torch.randn(8, n, device="cuda") for n in (71, 32, 32))
    with torch.no_grad():
        for _ in range(3):
            step.step([x], h, c)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            h1, c1 = step.step([x], h, c)
        gr.replay()
        torch.cuda.synchronize()
        _, (hl, cl) = lstm(x[None], (h[None], c[None]))
        torch.testing.assert_close(h1, hl[0], rtol=1e-5, atol=1e-6)
        for p_ in lstm.parameters():
            p_.add_(0.05 * torch.randn_like(p_))        # an optimizer step
        gr.replay()
        torch.cuda.synchronize()
        stale = h1.clone()
        step.refresh()
        gr.replay()
        torch.cuda.synchronize()
        _, (hl2, cl2) = lstm(x[None], (h[None], c[None]))
        torch.testing.assert_close(h1, hl2[0], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(c1, cl2[0], rtol=1e-5, atol=1e-6)
        assert (stale - h1).abs().max() > 1e-4          # (the replay before refresh() was on the old weights)


def test_lstm_step_follows_clip_adam_raw_pointer_updates():
    """``ClipAdam.clip_and_step`` writes the parameters through raw pointers.  The
    fused step must still see the new weights -- eagerly (the optimizer bumps ``Tensor._version``)
    and through a captured graph after the per-iteration ``refresh_step_weights()`` (forced copy)."""
    from rlpyt_amd import ops
    from rlpyt_amd.optim import ClipAdam
    torch.manual_seed(11)
    lstm = torch.nn.LSTM(71, 32).cuda()
    opt = ClipAdam(lstm.parameters(), lr=5e-2)
    step = ops.LstmStep(lstm)
    x, h, c = (torch.randn(8, n, device="cuda") for n in (71, 32, 32))
    with torch.no_grad():
        for _ in range(3):
            step.step([x], h, c)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            h1, c1 = step.step([x], h, c)
        gr.replay()
        before = h1.clone()
    v0 = lstm.weight_ih_l0._version
    for _ in range(3):                                      # real optimizer updates
        opt.zero_grad()
        out, _ = lstm(x[None], (h[None], c[None]))
        out.square().sum().backward()
        opt.clip_and_step(1.0)
    assert lstm.weight_ih_l0._version > v0, "ClipAdam must mark the parameters it wrote as changed"
    with torch.no_grad():
        _, (hl, cl) = lstm(x[None], (h[None], c[None]))
        he, ce = step.step([x], h, c)                       # eager: picks the update up by itself
        torch.testing.assert_close(he, hl[0], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(ce, cl[0], rtol=1e-5, atol=1e-6)
        # captured graph + a buffer that an update reached without any version bump
        lstm.weight_ih_l0.data.view(-1)[:7].copy_(torch.randn(7, device="cuda"))     # .data: no bump
        step._key = (lstm.weight_ih_l0._version, lstm.weight_hh_l0._version,
                     lstm.weight_ih_l0.data_ptr(), lstm.weight_hh_l0.data_ptr())
        step.refresh(force=True)                            # what refresh_step_weights() does
        gr.replay()
        torch.cuda.synchronize()
        _, (hl2, _cl2) = lstm(x[None], (h[None], c[None]))
        torch.testing.assert_close(h1, hl2[0], rtol=1e-5, atol=1e-6)
        assert (before - h1).abs().max() > 1e-4


@pytest.mark.parametrize("kind", ["r2d1", "lstm_pg"])
def test_fused_sampling_forward_after_clip_adam_updates_equals_library_path(kind):
    """after real ClipAdam updates the one-step sampling forward with
    ``use_fused_lstm_step=True`` (after the agent-level ``refresh_step_weights``) equals the
    ``nn.LSTM`` path on the SAME trained parameters."""
    from rlpyt_amd.optim import ClipAdam
    torch.manual_seed(9)
    B = 6
    if kind == "r2d1":
        from rlpyt_amd.models.dqn.atari_r2d1_model import AtariR2d1Model
        m = AtariR2d1Model((4, 104, 80), 6).cuda()
    else:
        from rlpyt_amd.models.pg.atari_lstm_model import AtariLstmModel
        m = AtariLstmModel((4, 104, 80), 6).cuda()
    obs = torch.randint(0, 256, (B, 4, 104, 80), dtype=torch.uint8, device="cuda")
    pa = torch.nn.functional.one_hot(torch.randint(0, 6, (B,), device="cuda"), 6).float()
    pr = torch.randn(B, device="cuda")
    st = (torch.randn(1, B, 512, device="cuda") * 0.3, torch.randn(1, B, 512, device="cuda"))

    def one_step():
        with torch.no_grad():
            return m(obs, pa, pr, st)

    m.eval()
    one_step()                                              # builds the LstmStep buffer (old weights)
    opt = ClipAdam(m.parameters(), lr=1e-2)
    for _ in range(3):
        m.train()
        opt.zero_grad()
        out = m(obs[None], pa[None], pr[None], st)          # [T=1, B]: library path under autograd
        sum(o.square().sum() for o in out[:-1]).backward()
        opt.clip_and_step(10.0)
    m.eval()
    m._lstm_step._key = (m.lstm.weight_ih_l0._version, m.lstm.weight_hh_l0._version,
                         m.lstm.weight_ih_l0.data_ptr(), m.lstm.weight_hh_l0.data_ptr())
    m.refresh_step_weights()         # even with a key that claims "unchanged" the copy happens
    fused = one_step()
    m.use_fused_lstm_step = False
    lib_ = one_step()
    for a, b in zip(fused[:-1], lib_[:-1]):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(fused[-1].h, lib_[-1].h, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(fused[-1].c, lib_[-1].c, rtol=1e-4, atol=1e-5)


def test_r2d1_model_one_step_forward_fused_vs_library_rnn():
    """AtariR2d1Model's one-step no-grad forward through ``ops.LstmStep`` equals the nn.LSTM path;
    sequences and anything under autograd still take nn.LSTM."""
    from rlpyt_amd import _lib
    from rlpyt_amd.models.dqn.atari_r2d1_model import AtariR2d1Model
    torch.manual_seed(5)
    m = AtariR2d1Model((4, 104, 80), 6).cuda().eval()
    B = 12
    obs = torch.randint(0, 256, (B, 4, 104, 80), dtype=torch.uint8, device="cuda")
    pa = torch.nn.functional.one_hot(torch.randint(0, 6, (B,), device="cuda"), 6).float()
    pr = torch.randn(B, device="cuda")
    st = (torch.randn(1, B, 512, device="cuda") * 0.3, torch.randn(1, B, 512, device="cuda"))
    with torch.no_grad():
        _lib.variant_reset()
        q1, s1 = m(obs, pa, pr, st)
        assert _lib.variant_counts().get("lstm_cell_kernel", 0) == 1
        m.use_fused_lstm_step = False
        _lib.variant_reset()
        q0, s0 = m(obs, pa, pr, st)
        assert _lib.variant_counts().get("lstm_cell_kernel", 0) == 0
    torch.testing.assert_close(q1, q0, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(s1.h, s0.h, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(s1.c, s0.c, rtol=1e-4, atol=1e-5)
    assert s1.h.shape == (1, B, 512)
    m.use_fused_lstm_step = True
    _lib.variant_reset()
    q, _ = m(obs, pa, pr, st)                      # grad enabled: library path
    assert q.requires_grad and _lib.variant_counts().get("lstm_cell_kernel", 0) == 0


def test_eps_greedy_kernel_semantics():
    """``rlpyt_eps_greedy_f32``: epsilon 0 -> torch.argmax; epsilon 1 -> floor(u * A) (uniform over
    the actions); vector epsilon per environment; the uniforms row is picked by the device-side time
    index; P(random) = epsilon (rlpyt/distributions/epsilon_greedy.py:17-29 in distribution)."""
    from rlpyt_amd import _lib, ops
    from rlpyt_amd.distributions.epsilon_greedy import EpsilonGreedy
    g = torch.Generator().manual_seed(11)
    n, A, T = 4096, 6, 3
    q = torch.randn(n, A, generator=g).cuda()
    u = torch.rand(T, n, generator=g).cuda()
    t1 = torch.tensor([1], dtype=torch.int64, device="cuda")
    zero, one = torch.zeros(1, device="cuda"), torch.ones(1, device="cuda")
    assert torch.equal(ops.eps_greedy(q, zero, u, t1), q.argmax(-1))
    a = ops.eps_greedy(q, one, u, t1)
    assert torch.equal(a, (u[1] * A).long().clamp(max=A - 1))
    assert torch.equal(ops.eps_greedy(q, one, u), (u[0] * A).long().clamp(max=A - 1))      # no t_dev: row 0
    counts = torch.bincount(a, minlength=A).float() / n
    assert (counts - 1. / A).abs().max() < 0.03
    eps = torch.where(torch.arange(n, device="cuda") % 2 == 0, 0.25, 0.).float()
    a = ops.eps_greedy(q, eps, u, t1)
    greedy = q.argmax(-1)
    assert torch.equal(a[1::2], greedy[1::2])
    took = u[1][0::2] < 0.25
    assert abs(took.float().mean().item() - 0.25) < 0.03
    assert torch.equal(a[0::2][~took], greedy[0::2][~took])
    assert torch.equal(a[0::2][took], (u[1][0::2][took] / 0.25 * A).long().clamp(max=A - 1))
    # through the distribution object: device-bound epsilon + the sampler's (u_all, t) pair
    d = EpsilonGreedy(dim=A)
    d.bind_device("cuda", n_envs=n)
    d.set_epsilon(0.25)
    _lib.variant_reset()
    a2 = d.sample(q, uniforms=(u, t1))
    assert _lib.variant_counts().get("eps_greedy_kernel", 0) == 1
    assert abs((a2 != greedy).float().mean().item() - 0.25 * (A - 1) / A) < 0.03
    a3 = d.sample(q)                                  # no uniforms: the torch path
    assert a3.shape == a2.shape and _lib.variant_counts().get("eps_greedy_kernel", 0) == 1


def _r2d1_agent(model_kwargs=None):
    from rlpyt_amd.agents.dqn.r2d1_agent import AtariR2d1Agent
    env = SyntheticPong()
    agent = AtariR2d1Agent(model_kwargs=model_kwargs or {}, eps_final=0.1)
    torch.manual_seed(3)
    agent.initialize(env.spaces, global_B=8, env_ranks=list(range(8)))
    torch.cuda.set_device(0)
    agent.to_device(0)
    agent.sample_mode(0)
    return agent


@pytest.mark.parametrize("model_kwargs", [{}, dict(fc_size=64, lstm_size=32, head_size=32, dueling=True)])
@pytest.mark.parametrize("with_resets", [True, False])
def test_r2d1_fused_sampling_step_equals_eager_step(model_kwargs, with_resets):
    """``R2d1Agent.step_with_reset`` (conv stack, trunk GEMM, ``rlpyt_rnn_step_inputs_f32``, gate GEMM,
    cell, head; state updated in place) against the eager sequence it replaces -- null previous action
    / reward after a reset, ``reset_where``, ``step`` (rlpyt/samplers/parallel/gpu/action_server.py:49-53
    + rlpyt/agents/dqn/r2d1_agent.py:23-40): same actions (same pre-drawn uniforms), Q-values, stored
    ``prev_rnn_state`` and internal state over a run of steps with resets."""
    from rlpyt_amd import _lib
    agent = _r2d1_agent(model_kwargs)
    B, T, A = 8, 12, 6
    H = agent.model.lstm.hidden_size
    g = torch.Generator().manual_seed(11)
    u_all = torch.rand(T, B, generator=g).cuda()
    prev_a = torch.randint(0, A, (B,), generator=g).cuda()
    agent.select_envs(0, B)
    for t in range(T):
        obs = torch.randint(0, 256, (B, 4, 104, 80), dtype=torch.uint8, generator=g).cuda()
        prev_r = torch.randn(B, generator=g).cuda()
        done = ((torch.rand(B, generator=g) < 0.3) if with_resets and t > 0
                else torch.zeros(B, dtype=torch.bool)).cuda()
        t_dev = torch.tensor([t], device="cuda")
        mask = done if with_resets else None
        # fused
        agent.select_slot("fused")
        agent.sample_uniforms = (u_all, t_dev)
        _lib.variant_reset()
        out = agent.step_with_reset(obs, prev_a, prev_r, mask)
        assert out is not None
        assert any("rnn_step_inputs_kernel" in k and v > 0 for k, v in _lib.variant_counts().items())
        # eager, on its own state slot
        agent.select_slot("eager")
        pa, pr = prev_a, prev_r
        if mask is not None:
            pa = torch.where(mask, torch.zeros_like(pa), pa)
            pr = torch.where(mask, torch.zeros_like(pr), pr)
            agent.reset_where(mask)
        agent.sample_uniforms = (u_all, t_dev)
        ref = agent.step(obs, pa, pr)
        agent.sample_uniforms = None
        torch.cuda.synchronize()
        assert torch.equal(out.action, ref.action), t
        np.testing.assert_allclose(out.agent_info.q.cpu().numpy(), ref.agent_info.q.cpu().numpy(),
                                   rtol=2e-4, atol=2e-6)
        for f in ("h", "c"):
            a, b = getattr(out.agent_info.prev_rnn_state, f), getattr(ref.agent_info.prev_rnn_state, f)
            assert tuple(a.shape) == tuple(b.shape) == (B, 1, H)
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-4, atol=2e-6)
            if mask is not None and bool(mask.any()):
                assert float(a[mask].abs().max()) == 0.0          # reset rows start from zero
        sf, se = agent._rnn_states["fused"], agent._rnn_states["eager"]
        np.testing.assert_allclose(sf.h.cpu().numpy(), se.h.cpu().numpy(), rtol=2e-4, atol=2e-6)
        np.testing.assert_allclose(sf.c.cpu().numpy(), se.c.cpu().numpy(), rtol=2e-4, atol=2e-6)
        prev_a = out.action


@pytest.mark.parametrize("mid_batch_reset", [True, False])
def test_r2d1_sampler_batches_equal_with_and_without_fused_step(mid_batch_reset, monkeypatch):
    """Whole batches of the HBM sampler (captured step graphs, C serve loop) with the fused recurrent
    step against the same sampler with ``step_with_reset`` disabled: identical observations / rewards /
    dones / actions, Q-values and stored recurrent states to fp32 tolerance -- reset and wait-reset
    collectors."""
    from rlpyt_amd.agents.dqn.r2d1_agent import AtariR2d1Agent, R2d1AgentBase

    def run(fused):
        if not fused:
            monkeypatch.setattr(R2d1AgentBase, "step_with_reset", lambda self, *a: None)
        T, B = 6, 8
        sampler = GpuSampler(SyntheticPong, dict(points_to_end=1, max_steps=9), batch_T=T, batch_B=B,
                             n_workers=2, n_groups=2, mid_batch_reset=mid_batch_reset,
                             max_decorrelation_steps=0)
        agent = AtariR2d1Agent(model_kwargs=dict(fc_size=64, lstm_size=32, head_size=32), eps_final=0.1)
        torch.manual_seed(0)
        np.random.seed(0)
        sampler.initialize(agent, seed=5, bootstrap_value=False)
        torch.cuda.set_device(0)
        agent.to_device(0)
        out = []
        try:
            for itr in range(4):
                agent.sample_mode(itr)
                s, _ = sampler.obtain_samples(itr)
                torch.cuda.synchronize()
                out.append(dict(obs=s.env.observation.float().mean((2, 3, 4)).cpu().numpy(),
                                reward=s.env.reward.cpu().numpy(), done=s.env.done.cpu().numpy(),
                                action=s.agent.action.cpu().numpy(),
                                q=s.agent.agent_info.q.cpu().numpy(),
                                h=s.agent.agent_info.prev_rnn_state.h.cpu().numpy(),
                                c=s.agent.agent_info.prev_rnn_state.c.cpu().numpy()))
            graphs = all(G.graph is not None for G in sampler.groups)
        finally:
            sampler.shutdown()
        monkeypatch.undo()
        return out, graphs

    a, ga = run(True)
    b, gb = run(False)
    assert ga and gb
    assert any(x["done"].any() for x in a)
    for x, y in zip(a, b):
        for k in ("obs", "reward", "done", "action"):
            assert np.array_equal(x[k], y[k]), k
        for k in ("q", "h", "c"):
            np.testing.assert_allclose(x[k], y[k], rtol=5e-4, atol=5e-6, err_msg=k)


@pytest.mark.parametrize("n,K_in,hidden,A", [(1, 512, 512, 6), (8, 6912, 512, 6), (48, 512, 512, 18),
                                            (130, 6912, 512, 4), (5, 64, 256, 3), (256, 512, 256, 9)])
def test_mlp_q_head_matches_torch(n, K_in, hidden, A):
    """``Linear -> ReLU -> Linear`` Q head of a no-grad sampling / target pass (split-K hidden layer +
    ``rlpyt_q_head_f32``) against the modules it replaces (rlpyt/models/mlp.py:24-31), through
    ``MlpModel.forward``; the launch counters say which path ran, autograd keeps the modules."""
    from rlpyt_amd import _lib
    from rlpyt_amd.models.mlp import MlpModel
    torch.manual_seed(n + A)
    m = MlpModel(K_in, hidden, output_size=A).cuda()
    x = torch.randn(n, K_in, device="cuda")
    ref64 = m.double()(x.double())
    m = m.float()
    _lib.variant_reset()
    m.use_fused_q_head_train = False
    lib = m(x)                                           # autograd with the switch off: library GEMMs
    assert lib.requires_grad
    assert not any("q_head_kernel" in k and v > 0 for k, v in _lib.variant_counts().items())
    del m.use_fused_q_head_train
    with torch.no_grad():
        got = m(x)
    assert any("q_head_kernel" in k and v > 0 for k, v in _lib.variant_counts().items())
    scale = ref64.abs().max().item()
    err = (got.double() - ref64).abs().max().item() / scale
    err_lib = (lib.detach().double() - ref64).abs().max().item() / scale
    assert got.shape == (n, A) and err <= max(3 * err_lib, 3e-7), (err, err_lib)


@pytest.mark.parametrize("n,K_in,hidden,A", [(128, 6912, 512, 6), (32, 64, 256, 18), (7, 48, 256, 1)])
def test_mlp_q_head_under_autograd_matches_torch(n, K_in, hidden, A):
    """The same head in the online network's pass of an update (``ops.mlp_q_head_train``: own forward that
    keeps the hidden activations, ``rlpyt_q_head_bwd_f32`` for the output layer's gradients + ReLU mask +
    hidden bias gradient): Q-values and all five gradients against the modules in float64, held to the f32
    module path's own error (rlpyt/models/mlp.py:24-31 under rlpyt/algos/dqn/dqn.py:226-265)."""
    from rlpyt_amd import _lib
    from rlpyt_amd.models.mlp import MlpModel
    torch.manual_seed(n + A)
    m = MlpModel(K_in, hidden, output_size=A).cuda()
    with torch.no_grad():
        m.model[0].bias.add_(0.3)                        # (some units on, some off)
    x = torch.randn(n, K_in, device="cuda")
    g = torch.randn(n, A, device="cuda")

    def run(model, xin, gin):
        xin = xin.clone().requires_grad_(True)
        model.zero_grad(set_to_none=True)
        q = model(xin)
        q.backward(gin)
        return [q.detach(), xin.grad] + [p.grad for p in model.parameters()]

    ref64 = run(m.double(), x.double(), g.double())
    m = m.float()
    m.use_fused_q_head_train = False
    lib = run(m, x, g)
    del m.use_fused_q_head_train
    _lib.variant_reset()
    got = run(m, x, g)
    ran = {k for k, v in _lib.variant_counts().items() if v > 0}
    assert any("q_head_bwd_kernel" in k for k in ran) and any("q_head_kernel" in k for k in ran), ran
    for name, a, b, r in zip(("q", "dx", "dw1", "db1", "dw2", "db2"), got, lib, ref64):
        scale = r.abs().max().item() + 1e-30
        err = (a.double() - r).abs().max().item() / scale
        err_lib = (b.double() - r).abs().max().item() / scale
        assert a.shape == r.shape and err <= max(3 * err_lib, 1e-6), (name, err, err_lib)
    again = run(m, x, g)                                  # fixed summation order
    assert all(torch.equal(a, b) for a, b in zip(got, again))
