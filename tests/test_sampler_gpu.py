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
    default), and additionally the conv kernel reading the newest frames in place
    (``zero_copy_frames``), give exactly the batches of the all-DMA path (one H2D of the
    frames + misc block, one D2H of the actions)."""
    def run(zc, zcf=False):
        s = GpuSampler(SyntheticPong, dict(points_to_end=1, max_steps=7), batch_T=6, batch_B=8,
                       n_workers=2, n_groups=2, zero_copy=zc, zero_copy_frames=zcf,
                       max_decorrelation_steps=0)
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
        assert all(G.zc_out == zc and G.zc_in == zcf for G in s.groups)
        s.shutdown()
        return out
    a, b, c = run(True), run(False), run(True, True)
    for x, y, z in zip(a, b, c):
        for u, v, w in zip(x, y, z):
            assert torch.equal(u, v) and torch.equal(u, w)


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
            monkeypatch.setattr(GpuSampler, "_tail_fused", lambda self, G, cuda: False)
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


def test_rollout_fetch_kernel_matches_host_uploads():
    """``rlpyt_rollout_fetch`` (first node of a device-driven step) does on the device what the
    master's per-step uploads did on the host: newest frames, reward / done scalars, full stacks +
    slot for reset envs (or for every env at t = 0), and publishes t -- read in place from the
    page-locked step buffer."""
    import ctypes
    from rlpyt_amd import _lib, ops
    from rlpyt_amd.utils.buffer import np_mp_array
    Bg, C, H, W = 7, 4, 104, 80
    rng = np.random.RandomState(5)
    t_off = 8 * Bg + ((2 * Bg + 15) // 16) * 16
    fr_bytes = Bg * H * W
    blk = np_mp_array(fr_bytes + t_off + 16, np.uint8)
    obs = np_mp_array((Bg, C, H, W), np.uint8)
    pinned = []
    for a in (blk, obs):
        assert _lib.lib.rlpyt_host_register(ctypes.c_void_p(a.ctypes.data), int(a.nbytes)) == 0
        pinned.append(a.ctypes.data)
    try:
        frame = blk[:fr_bytes].reshape(Bg, H, W)
        misc = blk[fr_bytes:]
        dev = torch.device("cuda:0")
        h_frame = _lib.host_mapped_tensor(frame, dev)
        h_misc = _lib.host_mapped_tensor(misc, dev)
        h_obs = _lib.host_mapped_tensor(obs, dev)
        d_blk = torch.zeros(blk.size, dtype=torch.uint8, device=dev)
        d_frame = d_blk[:fr_bytes].view(Bg, H, W)
        d_misc = d_blk[fr_bytes:]
        full_rows = torch.zeros((Bg, C, H, W), dtype=torch.uint8, device=dev)
        t_ctr = torch.zeros(1, dtype=torch.int64, device=dev)
        for t, resets in ((0, []), (3, [1, 5]), (9, []), (4, list(range(Bg)))):
            frame[:] = rng.randint(0, 256, frame.shape)
            obs[:] = rng.randint(0, 256, obs.shape)
            misc[:4 * Bg].view(np.float32)[:] = rng.randn(Bg)
            misc[8 * Bg:9 * Bg] = rng.rand(Bg) < 0.4
            misc[9 * Bg:10 * Bg] = 0
            misc[9 * Bg:10 * Bg][resets] = 1
            misc[4 * Bg:8 * Bg] = 77                     # host slot area: must NOT be copied
            full_rows.zero_()
            d_blk.zero_()
            t_ctr.fill_(t)
            ops.rollout_fetch(h_frame, h_misc, h_obs, d_frame, d_misc, full_rows, t_off, t_ctr)
            torch.cuda.synchronize()
            got = d_blk.cpu().numpy()
            assert np.array_equal(got[:fr_bytes], blk[:fr_bytes])
            gm = got[fr_bytes:]
            assert np.array_equal(gm[:4 * Bg], misc[:4 * Bg])                   # reward
            assert np.array_equal(gm[8 * Bg:10 * Bg], misc[8 * Bg:10 * Bg])     # done, reset
            full = np.ones(Bg, bool) if t == 0 else np.isin(np.arange(Bg), resets)
            slot = gm[4 * Bg:8 * Bg].view(np.int32)
            assert np.array_equal(slot, np.where(full, np.arange(Bg), -1))
            assert int(gm[t_off:t_off + 8].view(np.int64)[0]) == t
            fr = full_rows.cpu().numpy()
            for b in range(Bg):
                assert np.array_equal(fr[b], obs[b] if full[b] else np.zeros_like(obs[b]))
            assert int(t_ctr.item()) == t               # the fetch kernel only reads the counter
    finally:
        for p_ in pinned:
            _lib.lib.rlpyt_host_unregister(ctypes.c_void_p(p_))


@pytest.mark.parametrize("agent_kind", ["ff", "dqn"])
def test_device_driven_stepping_matches_host_driven(agent_kind, monkeypatch):
    """Batches collected with the fetch kernel + device step counter + enqueue-ahead serve loop
    (``device_fetch=True``, the default) are bit-identical to round 3's host-issued uploads +
    event-driven loop, for the fused AtariFf step and for a non-fused agent (DQN), through resets,
    over several batches."""
    def run(dev_fetch):
        kw = dict(batch_T=6, batch_B=8, n_workers=2, n_groups=2, max_decorrelation_steps=0,
                  device_fetch=dev_fetch)
        s = GpuSampler(SyntheticPong, dict(points_to_end=1, max_steps=7), **kw)
        if agent_kind == "ff":
            a = AtariFfAgent()
        else:
            from rlpyt_amd.agents.dqn.dqn_agent import AtariDqnAgent
            a = AtariDqnAgent(eps_final=0.2)
        torch.manual_seed(71)
        np.random.seed(71)
        s.initialize(a, seed=13, bootstrap_value=(agent_kind == "ff"))
        torch.cuda.set_device(0)
        a.to_device(0)
        torch.manual_seed(72)
        out = []
        for itr in range(6):
            smp, _ = s.obtain_samples(itr)
            torch.cuda.synchronize()
            leaves = [smp.env.observation, smp.agent.action, smp.env.reward, smp.env.done]
            if agent_kind == "ff":
                leaves += [smp.agent.agent_info.dist_info.prob, smp.agent.agent_info.value,
                           smp.agent.bootstrap_value]
            else:
                leaves += [smp.agent.agent_info.q]
            out.append([x.clone() for x in leaves])
        assert all(G.dev_fetch == dev_fetch for G in s.groups)
        ahead = getattr(s, "_ahead", None) is not None
        s.shutdown()
        return out, ahead
    (a, ahead_a), (b, ahead_b) = run(True), run(False)
    assert ahead_a and not ahead_b          # the enqueue-ahead loop really served the batches
    assert any(x[3].any() for x in a)
    for x, y in zip(a, b):
        for u, v in zip(x, y):
            assert torch.equal(u, v)
