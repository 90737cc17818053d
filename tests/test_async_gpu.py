"""Asynchronous mode on the device (rlpyt/replays/async_.py:8-47, replays/sum_tree.py:225-249,
runners/async_rl.py): the asynchronous prioritized frame replay under a concurrent appender thread and
optimizer-side draws / priority write-backs on DIFFERENT HIP streams (rows stay internally consistent,
the f64 tree stays a sum tree), ``AsyncSumTree``'s write-lock contract, the device parameter mailbox, and
``AsyncRl`` end to end with the product ``DQN`` + ``AtariDqnAgent`` + ``GpuSampler``."""
import threading
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _record():
    from rlpyt_amd.utils.collections import namedarraytuple
    return namedarraytuple("SamplesToBuffer", ["observation", "action", "reward", "done"])


def test_async_prioritized_frame_replay_concurrent_streams():
    from rlpyt_amd.replays.async_ import AsyncPrioritizedReplayFrameBuffer, AsyncSumTree
    S2B = _record()
    C, H, W, B, T_ring, T_new, n_appends = 4, 104, 80, 8, 256, 4, 150
    ex = S2B(observation=np.zeros((C, H, W), np.uint8), action=np.int64(0), reward=np.float32(0),
             done=np.bool_(False))
    buf = AsyncPrioritizedReplayFrameBuffer(example=ex, size=T_ring * B, B=B, discount=0.99,
                                            n_step_return=1, alpha=0.6, beta=0.4, default_priority=1.,
                                            device="cuda")
    tree = buf.priority_tree
    assert isinstance(tree, AsyncSumTree) and tree.guard is buf.rw_lock
    with pytest.raises(AssertionError, match="write lock"):
        tree.advance(1)                                  # a mutator outside the buffer's write lock
    errors, counts = [], dict(draws=0)

    def writer():
        torch.cuda.set_device(0)
        with torch.cuda.stream(torch.cuda.Stream()):     # the sampler side's own stream
            k = 0
            for _ in range(n_appends):
                t = torch.arange(k, k + T_new, device="cuda")
                stamp = ((t.view(-1, 1, 1) + torch.arange(C, device="cuda").view(1, 1, -1)) % 251).to(torch.uint8)
                obs = stamp.view(T_new, 1, C, 1, 1).expand(T_new, B, C, H, W).contiguous()
                buf.append_samples(S2B(observation=obs, action=t.repeat(B, 1).t().contiguous(),
                                       reward=t.float().repeat(B, 1).t().contiguous(),
                                       done=torch.zeros(T_new, B, dtype=torch.bool, device="cuda")))
                k += T_new
                time.sleep(0.0003)

    w = threading.Thread(target=writer)
    np.random.seed(0)
    w.start()
    with torch.cuda.stream(torch.cuda.Stream()):         # the optimizer side's stream
        while w.is_alive():
            if buf.t < 16 and not buf._buffer_full:
                time.sleep(0.001)
                continue
            b = buf.sample_batch(32)
            a = b.action
            obs = b.agent_inputs.observation
            ok = (torch.equal(obs[:, -1, 0, 0].long(), (a + C - 1) % 251)
                  and torch.equal(obs[:, 0, 7, 9].long(), a % 251)
                  and torch.equal(b.return_, a.float())
                  and torch.equal(b.target_inputs.observation[:, -1, 3, 3].long(), (a + C) % 251)
                  and bool(torch.isfinite(b.is_weights).all()))
            if not ok:
                errors.append(a[:4].tolist())
            buf.update_batch_priorities(torch.rand(32, device="cuda") + 0.1)
            counts["draws"] += 1
    w.join()
    torch.cuda.synchronize()
    assert not errors, errors[:3]
    assert counts["draws"] > 20
    assert buf.t == (n_appends * T_new) % T_ring and buf._buffer_full
    # still a sum tree: every internal node is the sum of its children
    full = tree.tree_tensor().cpu().numpy()
    n_int = (len(full) - 1) // 2
    kids = full[1:2 * n_int + 1:2] + full[2:2 * n_int + 2:2]
    np.testing.assert_allclose(full[:n_int], kids, rtol=1e-12, atol=1e-12)
    assert full[0] > 0


def test_stale_priority_write_back_keeps_band_and_rewritten_rows():
    """Draw, then append, then write priorities back (the asynchronous order): leaves that entered the
    guard band around the new cursor stay zero, rows rewritten meanwhile keep their fresh default, every
    other drawn leaf takes its new priority -- and the tree is still a sum tree."""
    from rlpyt_amd.replays.async_ import AsyncPrioritizedReplayFrameBuffer
    S2B = _record()
    C, H, W, B, T_ring, T_new = 4, 8, 8, 4, 32, 4
    ex = S2B(observation=np.zeros((C, H, W), np.uint8), action=np.int64(0), reward=np.float32(0),
             done=np.bool_(False))
    buf = AsyncPrioritizedReplayFrameBuffer(example=ex, size=T_ring * B, B=B, discount=0.99,
                                            n_step_return=1, alpha=1., beta=0.4, default_priority=1.,
                                            device="cuda")

    def append():
        buf.append_samples(S2B(observation=torch.zeros(T_new, B, C, H, W, dtype=torch.uint8, device="cuda"),
                               action=torch.zeros(T_new, B, dtype=torch.int64, device="cuda"),
                               reward=torch.zeros(T_new, B, device="cuda"),
                               done=torch.zeros(T_new, B, dtype=torch.bool, device="cuda")))

    for _ in range(T_ring // T_new + 2):                     # full, cursor at row 8
        append()
    tree = buf.priority_tree
    low = tree.low_idx
    np.random.seed(3)
    for trial in range(20):
        t0 = buf.t
        before = tree.tree_tensor().cpu().numpy()[low:low + T_ring * B].reshape(T_ring, B)
        batch = buf.sample_batch(64)
        T_leaf, B_leaf = (x.cpu().numpy() for x in buf.draws._drawn[:2])
        n_app = 1 + trial % 2
        for _ in range(n_app):                               # rows t0 .. rewritten, the band moves on
            append()
        t1 = buf.t
        buf.update_batch_priorities(torch.full((64,), 7.0, device="cuda"))
        after = tree.tree_tensor().cpu().numpy()
        leaves = after[low:low + T_ring * B].reshape(T_ring, B)
        band = {(t1 - 1 + k) % T_ring for k in range(1 + (C - 1))}           # back 1, forward C - 1
        rewritten = {(t0 + k) % T_ring for k in range(T_new * n_app)}
        hit = dict(band=0, rewritten=0, plain=0)
        for t, b in zip(T_leaf, B_leaf):
            if t in band:
                assert leaves[t, b] == 0., (trial, t, b)
                hit["band"] += 1
            elif t in rewritten:
                assert leaves[t, b] == 1., (trial, t, b)
                hit["rewritten"] += 1
            else:
                assert leaves[t, b] == 7., (trial, t, b)
                hit["plain"] += 1
        assert (leaves[sorted(band)] == 0).all()
        n_int = (len(after) - 1) // 2
        kids = after[1:2 * n_int + 1:2] + after[2:2 * n_int + 2:2]
        np.testing.assert_allclose(after[:n_int], kids, rtol=1e-12, atol=1e-12)
        assert hit["plain"] > 0 and (n_app == 1 or hit["rewritten"] > 0)
    # no append in between: the plain write-back, nothing masked
    batch = buf.sample_batch(16)
    T_leaf, B_leaf = (x.cpu().numpy() for x in buf.draws._drawn[:2])
    buf.update_batch_priorities(torch.full((16,), 3.0, device="cuda"))
    leaves = tree.tree_tensor().cpu().numpy()[low:low + T_ring * B].reshape(T_ring, B)
    assert all(leaves[t, b] == 3. for t, b in zip(T_leaf, B_leaf))


def test_agent_twin_mailbox_on_device_across_streams():
    from rlpyt_amd.agents.dqn.dqn_agent import AtariDqnAgent
    from rlpyt_amd.envs.synthetic import SyntheticPong
    agent = AtariDqnAgent()
    agent.initialize(SyntheticPong().spaces)
    agent.to_device(0)
    twin = agent.async_twin()
    assert next(twin.parameters()).is_cuda and twin.model is not agent.sampling_model
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for k in range(5):
        with torch.cuda.stream(s1), torch.no_grad():
            for p in agent.parameters():
                p.add_(0.5)
            agent.send_shared_memory()
        with torch.cuda.stream(s2):
            assert twin.recv_shared_memory() is True
            assert twin.recv_shared_memory() is False
        torch.cuda.synchronize()
        for a, b in zip(agent.model.state_dict().values(), twin.model.state_dict().values()):
            assert torch.equal(a, b), k


def test_async_rl_dqn_end_to_end():
    """The product stack in asynchronous mode: sampler thread (GpuSampler stepping the twin, appends under
    the write lock) || optimizer loop (DQN updates drawing under the read lock), throttled to the replay
    ratio.  Finite losses, the async log rows, update count bounded by the throttle, replay filled by the
    sampler, and the update kernels of the product path ran."""
    from rlpyt_amd import _lib
    from rlpyt_amd.agents.dqn.dqn_agent import AtariDqnAgent
    from rlpyt_amd.algos.dqn.dqn import DQN
    from rlpyt_amd.envs.synthetic import SyntheticPong
    from rlpyt_amd.replays.async_ import AsyncPrioritizedReplayFrameBuffer
    from rlpyt_amd.runners.async_rl import AsyncRl
    from rlpyt_amd.samplers.gpu import GpuSampler
    from rlpyt_amd.utils import logger
    logger.set_quiet(True)
    T, B, n_itr = 2, 16, 120
    sampler = GpuSampler(SyntheticPong, dict(points_to_end=1, max_steps=40), batch_T=T, batch_B=B,
                         n_workers=2, n_groups=1, max_decorrelation_steps=5)
    algo = DQN(discount=0.99, batch_size=32, min_steps_learn=T * B * 10, replay_size=T * B * 64,
               replay_ratio=4, target_update_interval=10, n_step_return=1, learning_rate=1e-4,
               prioritized_replay=True, double_dqn=True, updates_per_sync=1, eps_steps=T * B * 50)
    agent = AtariDqnAgent()
    runner = AsyncRl(algo=algo, agent=agent, sampler=sampler, n_steps=T * B * n_itr,
                     log_interval_steps=T * B * 40, seed=0, affinity=dict(cuda_idx=0))
    rows = []
    orig = logger.dump_tabular
    logger.dump_tabular = lambda *a, **k: (rows.append(dict(logger._tabular)), orig(*a, **k))
    _lib.variant_reset()
    try:
        runner.train()
    finally:
        logger.dump_tabular = orig
    torch.cuda.synchronize()
    rb = algo.replay_buffer
    assert isinstance(rb, AsyncPrioritizedReplayFrameBuffer) and algo.updates_per_optimize == 1
    assert runner.ctrl.sampler_itr == n_itr - 1 and rb._buffer_full
    throttle0 = 1 + algo.min_steps_learn // (T * B)
    delta = algo.batch_size * algo.updates_per_optimize / (T * B * algo.replay_ratio)
    assert 1 <= algo.update_counter <= (n_itr - throttle0) / delta + 2, algo.update_counter
    assert all(bool(torch.isfinite(p).all()) for p in agent.parameters())
    assert sampler.agent is runner.twin
    last = {k.split("/")[-1]: v for k, v in rows[-1].items()}
    for k in ("SamplerIteration", "CumUpdates", "ReplayRatio", "OptThrottle", "lossAverage",
              "gradNormAverage", "tdAbsErrAverage"):
        assert k in last, (k, sorted(last))
    # (the final row, like the reference's "Final log" async_rl.py:126-131, may cover no update at all)
    losses = [float(v) for r in rows for k, v in r.items() if k.endswith("lossAverage")]
    assert any(np.isfinite(x) for x in losses), losses
    assert all(np.isfinite(x) for x in losses[:-1]), losses
    assert float(last["CumReplayRatio"]) <= algo.replay_ratio * 1.25
    ran = {k for k, v in _lib.variant_counts().items() if v > 0}
    for name in ("dqn_loss_kernel", "frames_gather_kernel", "replay_step_fields_kernel", "find_kernel",
                 "clip_adam_apply_kernel", "dqn_conv1"):
        assert any(name in k for k in ran), (name, sorted(ran))
