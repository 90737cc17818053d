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
                                                          (0, 1, True), (2, 2, False),
                                                          (4, 2, True)])
def test_sampler_rows_consistent_on_device(n_workers, n_groups, use_graph):
    T, B = 6, 8
    s = GpuSampler(SyntheticPong, dict(points_to_end=1, max_steps=9), batch_T=T, batch_B=B,
                   n_workers=n_workers, n_groups=n_groups, use_graph=use_graph,
                   split_workers=(n_workers == 4), max_decorrelation_steps=0)
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


def test_commit_rows_matches_indexing():
    """rlpyt_commit_rows: dst[t + dt, lo:hi] = src for several leaves in one launch, t from a
    device counter; bit-exact vs torch indexing, incl. unaligned byte counts."""
    from rlpyt_amd import ops
    T, B, lo, Bg = 7, 10, 3, 5
    g = torch.Generator().manual_seed(0)
    obs = torch.zeros((T, B, 4, 13, 5), dtype=torch.uint8, device="cuda")       # 260 B / col
    rew = torch.zeros((T + 1, B), dtype=torch.float32, device="cuda")
    done = torch.zeros((T + 1, B), dtype=torch.bool, device="cuda")
    act = torch.zeros((T + 1, B), dtype=torch.int64, device="cuda")
    out = torch.zeros(Bg, dtype=torch.int64, device="cuda")
    s_obs = torch.randint(0, 256, (Bg, 4, 13, 5), dtype=torch.uint8, generator=g).cuda()
    s_rew = torch.randn(Bg, generator=g).cuda()
    s_done = (torch.rand(Bg, generator=g) < 0.5).cuda()
    s_act = torch.randint(0, 6, (Bg,), generator=g).cuda()
    rc = ops.RowCommit(5, torch.device("cuda:0"))
    rc.set_entries([(obs, s_obs, lo, 0), (rew, s_rew, lo, 0), (done, s_done, lo, 0),
                    (act, s_act, lo, 1), (out, s_act, None, 0)])
    t_dev = torch.tensor([4], dtype=torch.int64, device="cuda")
    rc.launch(t_dev)
    torch.cuda.synchronize()
    exp_obs = torch.zeros_like(obs)
    exp_obs[4, lo:lo + Bg] = s_obs
    assert torch.equal(obs, exp_obs)
    exp = torch.zeros_like(rew); exp[4, lo:lo + Bg] = s_rew  # noqa: E702
    assert torch.equal(rew, exp)
    exp = torch.zeros_like(done); exp[4, lo:lo + Bg] = s_done  # noqa: E702
    assert torch.equal(done, exp)
    exp = torch.zeros_like(act); exp[5, lo:lo + Bg] = s_act  # noqa: E702
    assert torch.equal(act, exp)
    assert torch.equal(out, s_act)


def test_commit_rows_zero_where_blanks_finished_envs():
    """Entries with ``zero_where``: rows of the flagged environments are written as zeros (the wait-reset
    collector's blank action / agent_info, rlpyt/samplers/parallel/gpu/collectors.py:85-91), the others
    copied -- wide (16-byte-multiple units), narrow (8-byte and 4-byte units) and odd-sized units; the
    destination held other values before."""
    from rlpyt_amd import ops
    T, B, lo, Bg = 5, 9, 2, 6
    g = torch.Generator().manual_seed(1)
    done = torch.tensor([0, 1, 0, 0, 1, 1], dtype=torch.bool, device="cuda")
    dsts = [torch.full((T, B, 512), 7., device="cuda"), torch.full((T + 1, B), 7, dtype=torch.int64, device="cuda"),
            torch.full((T, B), 7., device="cuda"), torch.full((T, B, 3, 5), 7, dtype=torch.uint8, device="cuda"),
            torch.full((Bg,), 7, dtype=torch.int64, device="cuda")]
    srcs = [torch.randn(Bg, 512, generator=g).cuda(), torch.randint(1, 6, (Bg,), generator=g).cuda(),
            torch.randn(Bg, generator=g).cuda(),
            torch.randint(1, 256, (Bg, 3, 5), dtype=torch.uint8, generator=g).cuda(),
            torch.randint(1, 6, (Bg,), generator=g).cuda()]
    rc = ops.RowCommit(5, torch.device("cuda:0"))
    rc.set_entries([(dsts[0], srcs[0], lo, 0, done), (dsts[1], srcs[1], lo, 1, done),
                    (dsts[2], srcs[2], lo, 0, done), (dsts[3], srcs[3], lo, 0, done),
                    (dsts[4], srcs[4], None, 0, done)])
    want = [d.clone() for d in dsts]
    t = 3
    for k, (w, x) in enumerate(zip(want, srcs)):
        blank = x * (~done).reshape((-1,) + (1,) * (x.dim() - 1)).to(x.dtype)
        if k == 4:
            w.copy_(blank)
        else:
            w[t + (1 if k == 1 else 0), lo:lo + Bg] = blank
    rc.launch(torch.tensor([t], dtype=torch.int64, device="cuda"))
    torch.cuda.synchronize()
    for d, w in zip(dsts, want):
        assert torch.equal(d, w)


@pytest.mark.parametrize("n,K,A", [(256, 512, 6), (5, 64, 18), (1, 512, 2)])
def test_categorical_head_matches_torch(n, K, A):
    """Heads + softmax within f32 tolerance (rtol 1e-5) of torch; the drawn action is exactly
    the inverse-CDF index min{a: cumsum(prob)[a] > u} of the kernel's own probabilities, and
    over many uniforms the empirical frequencies match the probabilities."""
    from rlpyt_amd import ops
    g = torch.Generator().manual_seed(n + A)
    h = torch.randn(n, K, generator=g).cuda()
    w_pi, b_pi = (torch.randn(A, K, generator=g) * 0.05).cuda(), torch.randn(A, generator=g).cuda()
    w_v, b_v = (torch.randn(1, K, generator=g) * 0.05).cuda(), torch.randn(1, generator=g).cuda()
    u = torch.rand(n, generator=g).cuda()
    prob, value, action = ops.categorical_head(h, w_pi, b_pi, w_v, b_v, u)
    ref_p = torch.softmax(h.double() @ w_pi.double().t() + b_pi.double(), -1)
    ref_v = (h.double() @ w_v.double().t()).squeeze(-1) + b_v.double()
    np.testing.assert_allclose(prob.cpu().numpy(), ref_p.cpu().numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(value.cpu().numpy(), ref_v.cpu().numpy(), rtol=1e-5, atol=1e-5)
    cum = np.cumsum(prob.cpu().numpy().astype(np.float32), axis=1, dtype=np.float32)
    exp = np.minimum((cum <= u.cpu().numpy()[:, None]).sum(1), A - 1)
    assert np.array_equal(action.cpu().numpy(), exp)
    # distribution check on one row
    N = 200000
    hh = h[:1].expand(N, K).contiguous()
    uu = torch.rand(N, generator=g).cuda()
    p1, _, a1 = ops.categorical_head(hh, w_pi, b_pi, w_v, b_v, uu)
    freq = torch.bincount(a1, minlength=A).double().cpu().numpy() / N
    np.testing.assert_allclose(freq, p1[0].double().cpu().numpy(), atol=5e-3)
    # no-sampling / no-value variants
    p2, v2, a2 = ops.categorical_head(h, w_pi, b_pi)
    assert v2 is None and a2 is None and torch.equal(p2, prob)


@pytest.mark.parametrize("n_workers,n_groups,mbr", [(0, 1, True), (2, 2, True), (0, 1, False)])
def test_frame_dedup_upload_is_bit_identical(n_workers, n_groups, mbr):
    """Uploading only the newest frame and rebuilding the stack in HBM (rlpyt_frame_push) gives
    exactly the observations / actions / rewards / dones of the full-observation upload (values
    to f32 accumulation order), through env resets (short episodes) and under both reset modes."""
    def run(dedup):
        s = GpuSampler(SyntheticPong, dict(points_to_end=1, max_steps=7), batch_T=6, batch_B=8,
                       n_workers=n_workers, n_groups=n_groups, mid_batch_reset=mbr,
                       frame_dedup=dedup, max_decorrelation_steps=0)
        a = AtariFfAgent()
        torch.manual_seed(11)
        np.random.seed(11)
        s.initialize(a, seed=4, bootstrap_value=True)
        torch.cuda.set_device(0)
        a.to_device(0)
        torch.manual_seed(12)
        out = []
        for itr in range(5):
            smp, _ = s.obtain_samples(itr)
            torch.cuda.synchronize()
            out.append([x.clone() for x in (smp.env.observation, smp.agent.action,
                                            smp.env.reward, smp.env.done,
                                            smp.agent.agent_info.value,
                                            smp.agent.bootstrap_value)])
        assert s.groups[0].dedup == dedup
        s.shutdown()
        return out
    a, b = run(True), run(False)
    assert any(x[3].any() for x in a)          # resets really happened
    for x, y in zip(a, b):
        for k, (u, v) in enumerate(zip(x, y)):
            if k < 4:      # observation, action, reward, done
                assert torch.equal(u, v)
            else:          # values: conv1 is an f32-MFMA chain in the fused sampling kernel and
                           # exact bf16x3 in conv1_fwd -- same products, other accumulation order
                torch.testing.assert_close(u, v, rtol=1e-5, atol=1e-6)


def test_native_step_loop_matches_python_loop():
    """rlpyt_sampler_serve (the C time-step loop: futex waits, H2D, hipGraphLaunch, D2H, event
    sync, action hand-off) produces exactly the batches of the Python loop."""
    def run(native):
        s = GpuSampler(SyntheticPong, dict(points_to_end=1, max_steps=7), batch_T=6, batch_B=8,
                       n_workers=2, n_groups=2, native_loop=native, max_decorrelation_steps=0)
        a = AtariFfAgent()
        torch.manual_seed(21)
        np.random.seed(21)
        s.initialize(a, seed=6, bootstrap_value=True)
        torch.cuda.set_device(0)
        a.to_device(0)
        torch.manual_seed(22)
        out = []
        for itr in range(5):
            smp, infos = s.obtain_samples(itr)
            torch.cuda.synchronize()
            out.append([x.clone() for x in (smp.env.observation, smp.agent.action,
                                            smp.env.reward, smp.env.done,
                                            smp.agent.agent_info.dist_info.prob,
                                            smp.agent.bootstrap_value)])
        assert (s._native is not None) == native
        s.shutdown()
        return out
    a, b = run(True), run(False)
    for x, y in zip(a, b):
        for u, v in zip(x, y):
            assert torch.equal(u, v)


@pytest.mark.parametrize("M,K,N", [(128, 3456, 512), (256, 3456, 512), (1, 64, 16), (100, 160, 48)])
def test_fc_small_matches_torch(M, K, N):
    """Split-K MFMA trunk layer vs float64 torch: max err <= 2e-5 * max|ref| (f32 FMA chains of
    length K, different summation order); run-to-run deterministic."""
    from rlpyt_amd import ops
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * 0.05
    b = torch.randn(N, generator=g)
    ref = torch.relu(x.double() @ w.double().t() + b.double())
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
    y = ops.fc_small(xd, wd, bd, relu=True)
    err = (y.double().cpu() - ref).abs().max().item()
    assert err <= 2e-5 * ref.abs().max().item() + 1e-6, err
    assert torch.equal(y, ops.fc_small(xd, wd, bd, relu=True))
    y2 = ops.fc_small(xd, wd, None, relu=False)
    ref2 = x.double() @ w.double().t()
    assert (y2.double().cpu() - ref2).abs().max().item() <= 2e-5 * ref2.abs().max().item() + 1e-6


def test_fused_step_matches_unfused_step():
    """The fused step (frame_push + conv1 + conv2 + row commit in one launch; trunk finish + heads
    + draw + row writes in one launch) produces the batches of the node-per-op step graph:
    observations / actions / rewards / dones exactly, probabilities and values to f32 accumulation
    order (conv1 is an f32-MFMA chain in the fused kernel, exact bf16x3 in conv1_fwd)."""
    def run(fused):
        s = GpuSampler(SyntheticPong, dict(points_to_end=1, max_steps=7), batch_T=6, batch_B=8,
                       n_workers=2, n_groups=2, fused_step=fused, max_decorrelation_steps=0)
        a = AtariFfAgent()
        torch.manual_seed(31)
        np.random.seed(31)
        s.initialize(a, seed=8, bootstrap_value=True)
        torch.cuda.set_device(0)
        a.to_device(0)
        torch.manual_seed(32)
        out = []
        for itr in range(4):
            smp, _ = s.obtain_samples(itr)
            torch.cuda.synchronize()
            out.append([x.clone() for x in (smp.env.observation, smp.agent.action,
                                            smp.env.reward, smp.env.done,
                                            smp.agent.agent_info.dist_info.prob,
                                            smp.agent.agent_info.value,
                                            smp.agent.bootstrap_value)])
        s.shutdown()
        return out
    a, b = run(True), run(False)
    for x, y in zip(a, b):
        for k, (u, v) in enumerate(zip(x, y)):
            if k < 4:
                assert torch.equal(u, v)
            else:
                torch.testing.assert_close(u, v, rtol=1e-5, atol=1e-6)


def test_zero_copy_step_buffer_matches_dma():
    """The head kernel writing the actions in place in the page-locked step buffer (zero-copy, the
    default) gives exactly the batches of the all-DMA path (one H2D of the frames + misc block,
    one D2H of the actions)."""
    def run(zc):
        s = GpuSampler(SyntheticPong, dict(points_to_end=1, max_steps=7), batch_T=6, batch_B=8,
                       n_workers=2, n_groups=2, zero_copy=zc, max_decorrelation_steps=0)
        a = AtariFfAgent()
        torch.manual_seed(41)
        np.random.seed(41)
        s.initialize(a, seed=9, bootstrap_value=True)
        torch.cuda.set_device(0)
        a.to_device(0)
        torch.manual_seed(42)
        out = []
        for itr in range(5):
            smp, _ = s.obtain_samples(itr)
            torch.cuda.synchronize()
            out.append([x.clone() for x in (smp.env.observation, smp.agent.action,
                                            smp.env.reward, smp.env.done,
                                            smp.agent.agent_info.dist_info.prob,
                                            smp.agent.bootstrap_value)])
        assert all(G.zc == zc for G in s.groups)
        s.shutdown()
        return out
    a, b = run(True), run(False)
    for x, y in zip(a, b):
        for u, v in zip(x, y):
            assert torch.equal(u, v)


def test_sample_convs_kernel_matches_separate_launches():
    """rlpyt_atari_sample_convs_f32 (frame push + conv1 + conv2, one env per workgroup) against
    rlpyt_frame_push followed by the two forward conv kernels: bit-identical rows, features
    equal up to the f32 accumulation order of conv1 (an f32-MFMA chain here, exact bf16x3 in the
    training-time kernel), for shifted stacks, reset slots and the reward/done row commit."""
    from rlpyt_amd import ops
    T, B, lo, Bg = 5, 12, 3, 7
    g = torch.Generator().manual_seed(0)
    obs_a = torch.randint(0, 256, (T, B, 4, 104, 80), dtype=torch.uint8, generator=g).cuda()
    obs_b = obs_a.clone()
    new_frame = torch.randint(0, 256, (Bg, 104, 80), dtype=torch.uint8, generator=g).cuda()
    full_rows = torch.randint(0, 256, (Bg, 4, 104, 80), dtype=torch.uint8, generator=g).cuda()
    slot = torch.tensor([-1, 1, -1, -1, 0, -1, -1], dtype=torch.int32).cuda()
    w1 = (0.05 * torch.randn(16, 4, 8, 8, generator=g)).cuda()
    b1 = (0.1 * torch.randn(16, generator=g)).cuda()
    w2 = (0.05 * torch.randn(32, 16, 4, 4, generator=g)).cuda()
    b2 = (0.1 * torch.randn(32, generator=g)).cuda()
    rew_src = torch.randn(Bg, generator=g).cuda()
    done_src = (torch.rand(Bg, generator=g) < 0.5).cuda()
    for t in (2, 4):
        t_dev = torch.tensor([t], dtype=torch.int64).cuda()
        rows_a = (torch.zeros(T + 1, B).cuda(), rew_src, torch.zeros(T + 1, B, dtype=torch.bool).cuda(),
                  done_src)
        rows_b = (torch.zeros(T + 1, B).cuda(), rew_src, torch.zeros(T + 1, B, dtype=torch.bool).cuda(),
                  done_src)
        stage = torch.empty((Bg, 4, 104, 80), dtype=torch.uint8, device="cuda")
        ops.frame_push(obs_a, t_dev, lo, new_frame, full_rows, slot, stage=stage, scalar_rows=rows_a)
        ref = ops.atari_conv_stack(stage, None, w1, b1, w2, b2)
        got = ops.atari_sample_convs(obs_b, t_dev, lo, new_frame, full_rows, slot, w1, b1, w2, b2,
                                     scalar_rows=rows_b)
        torch.cuda.synchronize()
        assert torch.equal(obs_a, obs_b)
        assert torch.equal(rows_a[0], rows_b[0]) and torch.equal(rows_a[2], rows_b[2])
        assert float(rows_b[0][t, lo:lo + Bg].abs().sum()) > 0
        assert got.shape == (Bg, 3456)
        torch.testing.assert_close(got, ref.reshape(Bg, -1), rtol=2e-6, atol=2e-6)
        # against torch's own convolutions in float64 (the tolerance of tests/test_conv_gpu.py)
        x = stage.double() / 255
        y = torch.relu(torch.nn.functional.conv2d(x, w1.double(), b1.double(), stride=4))
        y = torch.relu(torch.nn.functional.conv2d(y, w2.double(), b2.double(), stride=2, padding=1))
        err = (got.double() - y.reshape(Bg, -1)).abs().max().item()
        assert err <= 2e-5 * y.abs().max().item() + 1e-6, err


def test_fused_push_step_matches_separate_push():
    """Sampler level: with the frame push folded into the agent's conv launch the batches are
    exactly those of the separate frame_push launch, through resets, over several batches."""
    def run(fused_push):
        s = GpuSampler(SyntheticPong, dict(points_to_end=1, max_steps=7), batch_T=6, batch_B=8,
                       n_workers=2, n_groups=2, fused_push=fused_push, max_decorrelation_steps=0)
        a = AtariFfAgent()
        torch.manual_seed(51)
        np.random.seed(51)
        s.initialize(a, seed=10, bootstrap_value=True)
        torch.cuda.set_device(0)
        a.to_device(0)
        torch.manual_seed(52)
        out = []
        for itr in range(5):
            smp, _ = s.obtain_samples(itr)
            torch.cuda.synchronize()
            out.append([x.clone() for x in (smp.env.observation, smp.agent.action,
                                            smp.env.reward, smp.env.done,
                                            smp.agent.agent_info.dist_info.prob,
                                            smp.agent.agent_info.value,
                                            smp.agent.bootstrap_value)])
        s.shutdown()
        return out
    a, b = run(True), run(False)
    assert any(x[3].any() for x in a)          # resets really happened
    for x, y in zip(a, b):
        for k, (u, v) in enumerate(zip(x, y)):
            if k < 4:      # observation, action, reward, done
                assert torch.equal(u, v)
            else:          # values: conv1 is an f32-MFMA chain in the fused sampling kernel and
                           # exact bf16x3 in conv1_fwd -- same products, other accumulation order
                torch.testing.assert_close(u, v, rtol=1e-5, atol=1e-6)


def test_fused_tail_matches_agent_value_tail(monkeypatch):
    """The bootstrap value computed by one more pass of the step's fused kernels (frame push at
    t = T into the staging buffer + convs, trunk, value head only) equals ``agent.value`` on the
    uploaded full observation (round 3's tail); reward / done rows T and everything else of the
    batches are identical, through resets on the last step of a batch."""
    from rlpyt_amd import _lib

    def run(fused_tail):
        if not fused_tail:
            from rlpyt_amd.samplers.device import DeviceBatch
            monkeypatch.setattr(DeviceBatch, "tail_fused", lambda self, G: False)
        s = GpuSampler(SyntheticPong, dict(points_to_end=1, max_steps=6), batch_T=6, batch_B=8,
                       n_workers=2, n_groups=2, max_decorrelation_steps=0)
        a = AtariFfAgent()
        torch.manual_seed(61)
        np.random.seed(61)
        s.initialize(a, seed=12, bootstrap_value=True)
        torch.cuda.set_device(0)
        a.to_device(0)
        torch.manual_seed(62)
        out = []
        _lib.variant_reset()
        for itr in range(5):
            smp, _ = s.obtain_samples(itr)
            torch.cuda.synchronize()
            out.append([x.clone() for x in (smp.env.observation, smp.agent.action, smp.env.reward,
                                            smp.env.done, smp.agent.agent_info.dist_info.prob,
                                            smp.agent.agent_info.value, smp.agent.bootstrap_value)])
        cnt = _lib.variant_counts()
        s.shutdown()
        monkeypatch.undo()
        return out, cnt
    (a, ca), (b, cb) = run(True), run(False)
    assert ca.get("rollout_head_kernel<2>", 0) > cb.get("rollout_head_kernel<2>", 0) > 0
    assert any(x[3][-1].any() for x in a)       # an env finished on the last step of a batch
    for x, y in zip(a, b):
        for k, (u, v) in enumerate(zip(x, y)):
            if k < 4:
                assert torch.equal(u, v)
            else:
                torch.testing.assert_close(u, v, rtol=1e-5, atol=1e-6)
