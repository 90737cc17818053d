"""Asynchronous mode off the device: the RW lock, the asynchronous replay buffers on
CPU tensors (one-step returns: no kernel involved) under concurrent appends and draws, the agent twin's
parameter mailbox, and the ``AsyncRl`` orchestration (throttle, sampler twin, log rows) over a stub
algorithm.  The device twins are in tests/test_async_gpu.py."""
import threading
import time

import numpy as np
import pytest
import torch

from rlpyt_amd.replays.async_ import (AsyncReplayBufferMixin, AsyncSumTree, AsyncUniformReplayFrameBuffer,
                                      RWLock, async_replay_class)
from rlpyt_amd.utils import logger
from rlpyt_amd.utils.collections import namedarraytuple

logger.set_quiet(True)
S2B = namedarraytuple("SamplesToBuffer", ["observation", "action", "reward", "done"])


def test_rw_lock_many_readers_one_writer():
    """rlpyt/utils/synchronize.py:5-36: readers share, a writer excludes everybody."""
    lock = RWLock()
    state = dict(readers=0, max_readers=0, writer=False, bad=0)
    stop = time.time() + 0.4

    def reader():
        while time.time() < stop:
            with lock:
                state["readers"] += 1
                state["max_readers"] = max(state["max_readers"], state["readers"])
                if state["writer"]:
                    state["bad"] += 1
                time.sleep(0.001)
                state["readers"] -= 1

    def writer():
        while time.time() < stop:
            with lock.writing():
                assert lock.held_for_writing()
                if state["readers"] or state["writer"]:
                    state["bad"] += 1
                state["writer"] = True
                time.sleep(0.001)
                state["writer"] = False
            time.sleep(0.0005)
    ts = [threading.Thread(target=reader) for _ in range(3)] + [threading.Thread(target=writer)
                                                                  for _ in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert state["bad"] == 0 and state["max_readers"] >= 2
    assert not lock.held_for_writing()
    with lock:                                   # a reader holds the underlying mutex, not "writing"
        assert not lock.held_for_writing() and lock.write_lock.locked()
    assert not lock.write_lock.locked()


def test_async_class_table_and_names():
    """The reference's eight asynchronous class names, each its synchronous buffer + the mixin."""
    import rlpyt_amd.replays.async_ as A
    import rlpyt_amd.replays.buffers as Bf
    for frames in (False, True):
        for seq in (False, True):
            for pri in (False, True):
                cls = async_replay_class(frames, seq, pri)
                base = Bf.replay_class(frames, seq, pri)
                assert cls.__name__ == "Async" + base.__name__ and getattr(A, cls.__name__) is cls
                assert issubclass(cls, AsyncReplayBufferMixin) and issubclass(cls, base) and cls.async_
                assert cls.TREE_CLS is AsyncSumTree


def test_async_uniform_frame_buffer_concurrent_append_and_draw():
    """A writer thread appends while two reader threads draw: every drawn row is INTERNALLY consistent
    (its frames, action and reward all carry the same global time stamp) and the cursor / full flag end
    where a serial run ends."""
    C, H, W, B, T_ring = 4, 2, 2, 3, 40
    ex = S2B(observation=np.zeros((C, H, W), np.uint8), action=np.int64(0), reward=np.float32(0),
             done=np.bool_(False))
    buf = AsyncUniformReplayFrameBuffer(example=ex, size=T_ring * B, B=B, discount=0.99, n_step_return=1,
                                        device="cpu")
    assert buf.async_ and not buf.can_sample_on_device()
    n_appends, T_new = 60, 5
    errors, drawn = [], [0]

    def frames_of(k):            # observation at global step k: frames k .. k + C - 1 (value = index % 251)
        return np.stack([np.full((B, H, W), (k + f) % 251, np.uint8) for f in range(C)], axis=1)

    def writer():
        k = 0
        for _ in range(n_appends):
            obs = np.stack([frames_of(k + i) for i in range(T_new)])
            steps = torch.arange(k, k + T_new)
            buf.append_samples(S2B(observation=torch.from_numpy(obs),
                                   action=steps.repeat(B, 1).t().contiguous(),
                                   reward=steps.float().repeat(B, 1).t().contiguous(),
                                   done=torch.zeros(T_new, B, dtype=torch.bool)))
            k += T_new
            time.sleep(0.0005)

    def reader():
        rng_done = time.time() + 5
        while w.is_alive() and time.time() < rng_done:
            if buf.t < 8 and not buf._buffer_full:
                time.sleep(0.001)
                continue
            # (the gather kernels need the device: tests/test_async_gpu.py draws through sample_batch;
            # here the rows are read by hand under the same read lock)
            with buf.rw_lock:
                T_idxs, B_idxs = buf.sample_idxs(16)
                a = buf.samples.action[T_idxs, B_idxs].numpy()
                r = buf.samples.reward[T_idxs, B_idxs].numpy()
                fr = buf.samples_frames.numpy()
                newest = fr[T_idxs + C - 1, B_idxs, 0, 0].astype(np.int64)
                oldest = fr[T_idxs, B_idxs, 0, 0].astype(np.int64)
            if not (np.array_equal(newest, (a + C - 1) % 251) and np.array_equal(oldest, a % 251)
                    and np.array_equal(r, a.astype(np.float32))):
                errors.append((a, newest, oldest))
            drawn[0] += 1
    w = threading.Thread(target=writer)
    rs = [threading.Thread(target=reader) for _ in range(2)]
    np.random.seed(0)
    w.start()
    for r in rs:
        r.start()
    w.join()
    for r in rs:
        r.join()
    assert not errors, errors[:2]
    assert drawn[0] > 10
    assert buf.t == (n_appends * T_new) % T_ring and buf._buffer_full


def test_agent_twin_mailbox():
    """``async_twin`` / ``send_shared_memory`` / ``recv_shared_memory``: the twin keeps its own
    parameters, sees published ones only after a receive, and reports whether anything was new."""
    from rlpyt_amd.agents.pg.atari import MlpCategoricalPgAgent
    from rlpyt_amd.envs.synthetic import TinyDiscreteEnv
    agent = MlpCategoricalPgAgent()
    agent.initialize(TinyDiscreteEnv().spaces)
    twin = agent.async_twin()
    assert twin.model is not agent.model and twin._mailbox is agent._mailbox
    p, q = next(agent.parameters()), next(twin.parameters())
    assert torch.equal(p, q) and p.data_ptr() != q.data_ptr()
    assert twin.recv_shared_memory() is False                 # nothing published yet
    with torch.no_grad():
        p.add_(1.0)
    assert not torch.equal(p, q)                              # the twin is frozen ...
    agent.send_shared_memory()
    assert not torch.equal(p, q)                              # ... until IT receives
    v = q._version
    assert twin.recv_shared_memory() is True and torch.equal(p, q) and q._version > v
    assert twin.recv_shared_memory() is False
    for a, b in zip(agent.model.state_dict().values(), twin.model.state_dict().values()):
        assert torch.equal(a, b)


class StubAsyncAlgo:
    """TEST ONLY: the asynchronous-algorithm protocol (rlpyt/algos/dqn/dqn.py:99-125,158-190) over an
    asynchronous uniform replay on CPU tensors, with a one-line 'update' that is visible in the policy."""
    opt_info_fields = ("loss",)
    bootstrap_value = False
    batch_size, replay_ratio, updates_per_optimize, min_steps_learn, discount = 8, 4, 1, 32, 0.99

    def __init__(self):
        self.update_counter = 0
        self.sampler_itrs_seen = []

    def async_initialize(self, agent, sampler_n_itr, batch_spec, mid_batch_reset, examples, world_size=1):
        self.agent = agent
        ex = S2B(observation=examples["observation"], action=examples["action"],
                 reward=examples["reward"], done=examples["done"])
        self.replay_buffer = async_replay_class(False, False, False)(
            example=ex, size=4096, B=batch_spec.B, discount=self.discount, n_step_return=1, device="cpu")
        return self.replay_buffer

    def optim_initialize(self, rank=0):
        self.rank = rank

    def samples_to_buffer(self, samples):
        return S2B(observation=samples.env.observation, action=samples.agent.action,
                   reward=samples.env.reward, done=samples.env.done)

    def optimize_agent(self, itr, samples=None, sampler_itr=None):
        from collections import namedtuple
        assert samples is None and sampler_itr is not None
        self.sampler_itrs_seen.append(sampler_itr)
        rb = self.replay_buffer
        with rb.rw_lock:                 # (sample_batch's gathers are device kernels: read by hand)
            T_idxs, B_idxs = rb.sample_idxs(self.batch_size)
            ret = rb.samples.reward[T_idxs, B_idxs]
        with torch.no_grad():
            for p in self.agent.parameters():
                p.add_(1e-3)
        self.update_counter += 1
        return namedtuple("OptInfo", ["loss"])(loss=[float(ret.mean())])

    def optim_state_dict(self):
        return {}


def test_async_rl_orchestration_on_cpu():
    """``AsyncRl``: the sampler thread fills the replay while the optimizer runs; the optimizer is
    throttled to the replay ratio (async_rl.py:187-192); the sampler steps the twin, which follows the
    published parameters; the reference's async log rows come out."""
    from rlpyt_amd.agents.pg.atari import MlpCategoricalPgAgent
    from rlpyt_amd.envs.synthetic import TinyDiscreteEnv
    from rlpyt_amd.runners.async_rl import AsyncRl
    from rlpyt_amd.samplers.gpu import GpuSampler
    T, B = 4, 4
    sampler = GpuSampler(TinyDiscreteEnv, dict(), batch_T=T, batch_B=B, n_workers=0,
                         max_decorrelation_steps=0)
    algo, agent = StubAsyncAlgo(), MlpCategoricalPgAgent()
    runner = AsyncRl(algo=algo, agent=agent, sampler=sampler, n_steps=T * B * 60,
                     log_interval_steps=T * B * 20, seed=0, affinity=dict(cuda_idx=None))
    rows = []
    orig = logger.dump_tabular
    logger.dump_tabular = lambda *a, **k: (rows.append(dict(logger._tabular)), orig(*a, **k))
    state = {}
    real_opt_init = algo.optim_initialize

    def opt_init(rank=0):            # (the model exists from sampler.initialize on)
        state["p0"] = next(agent.parameters()).detach().clone()
        return real_opt_init(rank)
    algo.optim_initialize = opt_init
    try:
        runner.train()
    finally:
        logger.dump_tabular = orig
    assert runner.n_itr == 60 and runner.ctrl.sampler_itr == 59
    # throttle: the first update waits for min_steps_learn, later ones for the replay ratio
    throttle0 = 1 + algo.min_steps_learn // (T * B)
    assert algo.sampler_itrs_seen[0] + 1 >= throttle0
    delta = algo.batch_size * algo.updates_per_optimize / (T * B * algo.replay_ratio)
    for k, seen in enumerate(algo.sampler_itrs_seen):
        assert seen + 1 >= throttle0 + k * delta - 1e-9, (k, seen)
    max_updates = (60 - throttle0) / delta + 2
    assert 1 <= algo.update_counter <= max_updates, (algo.update_counter, max_updates)
    # the sampler stepped the twin, and the twin followed the optimizer's parameters
    assert sampler.agent is runner.twin and runner.twin.model is not agent.model
    p, q = next(agent.parameters()), next(runner.twin.parameters())
    # (the sampler runs at full speed and may finish several updates before the optimizer stops: the
    # twin holds SOME published state -- not the initial parameters -- at most all updates behind)
    lag = float((p.detach() - q.detach()).abs().max())
    assert not torch.equal(q.detach(), state["p0"]) and lag <= algo.update_counter * 1e-3 + 1e-6
    assert abs(float((p.detach() - state["p0"]).abs().max()) - algo.update_counter * 1e-3) < 1e-5
    # replay holds what the sampler produced
    rb = algo.replay_buffer
    assert rb.t == (60 * T) % rb.T or rb._buffer_full
    keys = [k.split("/")[-1] for k in rows[-1]]
    assert keys[:13] == ["CumCompletedTrajs", "NewCompletedTrajs", "StepsInTrajWindow", "Iteration",
                         "SamplerIteration", "CumTime (s)", "CumSteps", "CumUpdates", "ReplayRatio",
                         "CumReplayRatio", "StepsPerSecond", "UpdatesPerSecond", "OptThrottle"]
    assert "lossAverage" in keys and "ReturnAverage" in keys and len(rows) >= 3
