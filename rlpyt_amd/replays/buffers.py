"""Replay buffers of the DQN family, resident in HBM.

Protocol (what ``DQN`` / ``R2D1`` and the reference's own algorithms call, SURVEY.md section 8(b);
constructor keywords and method signatures are pinned to the reference classes by
tests/golden/protocol.json): ``append_samples(samples) -> (T, rows)``, ``sample_batch(batch_B)``,
``update_batch_priorities(priorities)``, ``set_beta(beta)``.  The eight class names below are the
reference's (rlpyt/replays/non_sequence/{uniform,prioritized,frame}.py,
rlpyt/replays/sequence/{uniform,prioritized,frame}.py); each is one ``ReplayBuffer`` configured by
three switches -- ``FRAMES`` (observations stored once per frame), ``SEQUENCE`` (batches are
``[T, B]`` sequences with a stored RNN state), ``PRIORITIZED`` (sum-tree draws) -- over the parts of
``store.py`` and ``index.py``.  ``replay_class(frames, sequence, prioritized)`` picks by switches.

``sample_batch`` never leaves the device: tree descent, frame-stack re-assembly for agent and
target inputs (``rlpyt_frames_gather[_pair|_seq]``), the small-field gathers
(``rlpyt_gather_rows`` / ``rlpyt_gather_sequences``) and the importance weights are kernels; the
only host -> device traffic per batch is its ``n`` float64 uniforms (``np.random.rand``, so that a
seeded run draws the reference's stream)."""
import os

import torch

from .. import ops
from ..agents.base import AgentInputs
from ..utils.buffer import buffer_func, buffer_leaves, buffer_pairs, get_leading_dims
from ..utils.collections import namedarraytuple
from .index import PriorityDraw, UniformDraw
from .store import FieldRing, FrameStore, RingCursor, RnnStateStore, as_index

# (RLPYT_REPLAY_APPEND=0: the slice assignments of replays/store.py on the device too, for A/B runs)
ONE_LAUNCH_APPEND = os.environ.get("RLPYT_REPLAY_APPEND", "1") != "0"

StepBatch = namedarraytuple("SamplesFromReplay", ["agent_inputs", "action", "return_", "done",
                                                  "done_n", "target_inputs"])
StepBatchPri = namedarraytuple("SamplesFromReplayPri", StepBatch._fields + ("is_weights",))
SeqBatch = namedarraytuple("SamplesFromReplay", ["all_observation", "all_action", "all_reward",
                                                 "return_", "done", "done_n", "init_rnn_state"])
SeqBatchPri = namedarraytuple("SamplesFromReplayPri", SeqBatch._fields + ("is_weights",))


def _without(record, *names):
    """``record`` (a namedarraytuple instance) minus some fields, as a fresh namedarraytuple."""
    keep = [f for f in record._fields if f not in names]
    cls = namedarraytuple(type(record).__name__ + "Stored", keep)
    return cls(*(getattr(record, f) for f in keep))


class ReplayBuffer:
    FRAMES = SEQUENCE = PRIORITIZED = False
    async_ = False
    TREE_CLS = None         # sum-tree class of prioritized buffers (None: ops.DeviceSumTree)
    _appender = None        # ops.ReplayAppender over this buffer's rings, built at the first append

    def _build(self, example, size, B, discount=1, n_step_return=1, device=None,
               rnn_state_interval=0, batch_T=None, alpha=0.6, beta=0.4, default_priority=1,
               unique=False, input_priorities=False, input_priority_shift=0):
        dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.device, self.B = dev, B
        self.discount, self.n_step_return = discount, n_step_return
        self.rnn_state_interval, self.batch_T = rnn_state_interval, batch_T
        keeps_state = self.SEQUENCE and rnn_state_interval > 1
        if keeps_state:
            size = RnnStateStore.padded_size(size, B, rnn_state_interval)
        stored = example
        if self.FRAMES:
            stored = _without(stored, "observation")
        if keeps_state:
            stored = _without(stored, "prev_rnn_state")
        self._stored_fields = stored._fields
        cur = self.cursor = RingCursor(size, B, guard_back=n_step_return, guard_fwd=1, device=dev)
        self.T, self.size = cur.T, cur.T * B
        self.fields = FieldRing(stored, cur, discount, n_step_return)
        self._flat_record = all(isinstance(x, torch.Tensor) for x in self.fields.data)
        self.frame_store = FrameStore(example.observation, cur) if self.FRAMES else None
        if self.frame_store is not None:
            cur.guard_fwd = max(cur.guard_fwd, self.frame_store.guard_fwd)
        self.rnn_store = (RnnStateStore(example.prev_rnn_state, cur, rnn_state_interval)
                          if keeps_state else None)
        if self.PRIORITIZED:
            if self.SEQUENCE:
                assert batch_T is not None, "Must assign fixed batch_T for prioritized."
            self.draws = PriorityDraw(cur, alpha, beta, default_priority, input_priorities,
                                      input_priority_shift, unique=unique,
                                      stride=rnn_state_interval if self.SEQUENCE else 1,
                                      reach=batch_T if self.SEQUENCE else 0, sequence=self.SEQUENCE,
                                      tree_cls=self.TREE_CLS)
        else:
            self.draws = UniformDraw(cur, stride=rnn_state_interval, sequence=self.SEQUENCE)

    # -- views the algorithms, tests and bench read -------------------------------------------
    samples = property(lambda self: self.fields.data)
    samples_return_ = property(lambda self: self.fields.return_)
    samples_done_n = property(lambda self: self.fields.done_n)
    samples_frames = property(lambda self: self.frame_store.frames)
    samples_prev_rnn_state = property(lambda self: self.rnn_store.data)
    priority_tree = property(lambda self: self.draws.tree)
    t = property(lambda self: self.cursor.t)
    _buffer_full = property(lambda self: self.cursor.full)
    off_backward = property(lambda self: self.cursor.guard_back)
    off_forward = property(lambda self: self.cursor.guard_fwd)
    n_frames = property(lambda self: self.frame_store.C)

    # -- protocol -----------------------------------------------------------------------------
    def append_samples(self, samples):
        """Write ``[T, B]`` new steps at the cursor; returns ``(T, rows)`` (rows: slice or index
        vector).  ``samples`` may be a ``(priorities, samples)`` record carrying input priorities."""
        priorities = None
        if hasattr(samples, "priorities"):
            priorities, samples = samples.priorities, samples.samples
        news = None
        if self._flat_record:       # every stored field is one array: no walk over the record
            news = [getattr(samples, f) for f in self._stored_fields]
            T, B = news[0].shape[:2]
        else:
            T, B = get_leading_dims(samples, n_dim=2)
        assert B == self.B
        claim = self.cursor.claim(T)
        if self.device.type == "cuda" and T <= self.T and ONE_LAUNCH_APPEND:
            if news is None:
                stored = type(self.fields.data)(*(getattr(samples, f) for f in self._stored_fields))
                news = [new for _, new in buffer_pairs(self.fields.data, stored)]
            self._append_one_launch(news, samples, claim)
        else:       # host tensors (tests/test_host_logic.py) or an append longer than a lap
            stored = type(self.fields.data)(*(getattr(samples, f) for f in self._stored_fields))
            self.fields.write(stored, claim)
            if self.frame_store is not None:
                self.frame_store.write(samples.observation, claim)
        if self.rnn_store is not None:
            self.rnn_store.write(samples.prev_rnn_state, claim)
        self.draws.on_append(claim, self.cursor.t, priorities)
        return T, claim.rows

    def _append_one_launch(self, news, samples, claim):
        """Fields (``news``: one array per ring leaf) + newest frames + history / mirror rows of an
        append as ONE kernel (``rlpyt_replay_append``) instead of a slice assignment per leaf, then
        the n-step refresh."""
        fs = self.frame_store
        if self._appender is None:
            self._appender = ops.ReplayAppender(
                buffer_leaves(self.fields.data), self.T, self.B,
                frames=None if fs is None else fs.frames, n_frames=1 if fs is None else fs.C)
        self._appender(news, claim.start, None if fs is None else samples.observation)
        if self.fields.n_step > 1:
            self.fields._refresh_returns(claim)

    def sample_batch(self, batch_B, batch_T=None):
        if self.SEQUENCE:
            span = self.batch_T if batch_T is None else batch_T
            T_idxs, B_idxs, weights = self.draws.draw(batch_B, reach=span)
            batch = self.extract_batch(T_idxs, B_idxs, span)
        else:
            T_idxs, B_idxs, weights = self.draws.draw(batch_B)
            batch = self.extract_batch(T_idxs, B_idxs)
        if weights is None:
            return batch
        return (SeqBatchPri if self.SEQUENCE else StepBatchPri)(*batch, is_weights=weights)

    def can_sample_on_device(self):
        """True when a batch can be drawn without any host work inside the draw (captured update
        graphs): single-step batches, non-unique priority draws or plain index pairs."""
        return not self.SEQUENCE and not getattr(self.draws, "unique", False)

    def sample_batch_device(self, batch_B, uniforms=None, idxs=None, beta=None):
        """``sample_batch`` with the randomness handed in as DEVICE tensors at fixed addresses --
        ``uniforms`` (f64 ``[batch_B]``, prioritized: what ``np.random.rand`` would have produced)
        or ``idxs`` (``(T_idxs, B_idxs)`` int64, uniform replay) -- and the importance exponent as a
        device scalar ``beta``: every step is a kernel launch on the current stream, so the call
        can sit inside a captured hipGraph (``algos/dqn/captured.py``)."""
        assert self.can_sample_on_device()
        if self.PRIORITIZED:
            T_idxs, B_idxs, weights = self.draws.draw_device(uniforms, beta)
            return StepBatchPri(*self._steps(T_idxs, B_idxs), is_weights=weights)
        return self._steps(*idxs)

    def update_batch_priorities(self, priorities):
        self.draws.update(priorities)

    def set_beta(self, beta):
        self.beta = self.draws.beta = beta

    def set_batch_T(self, batch_T):
        self.batch_T = batch_T

    def sample_idxs(self, batch_B, batch_T=None):
        """The index draw alone (host arrays for uniform buffers), e.g. for inspection."""
        return self.draws.draw(batch_B, reach=batch_T)[:2]

    # -- gathers ------------------------------------------------------------------------------
    def extract_batch(self, T_idxs, B_idxs, T=None):
        T_idxs, B_idxs = as_index(T_idxs, self.device), as_index(B_idxs, self.device)
        return (self._sequences(T_idxs, B_idxs, T) if self.SEQUENCE
                else self._steps(T_idxs, B_idxs))

    def _steps(self, T_idxs, B_idxs):
        """Single-step training rows at ``(t, b)`` with the target network's inputs ``n_step`` rows
        later; the action / reward ENTERING a step are those of row ``t - 1`` (row -1 = the ring's
        last row), nulled where that row ended an episode."""
        d, row = self.fields.data, ops.gather_rows
        ret_ring, dn_ring = self.fields.return_, self.fields.done_n
        if self.frame_store is not None and self._fused_fields_ok(d, ret_ring, dn_ring):
            # product path: every small field in ONE launch, both 4-frame stacks in one more
            pa, pr, act, ret, dn, dnn, tpa, tpr = ops.replay_step_fields(
                d.action, d.reward, d.done, ret_ring, dn_ring, T_idxs, B_idxs, self.n_step_return)
            obs, nxt_obs = ops.frames_gather_pair(self.frame_store.frames, d.done, T_idxs, B_idxs,
                                                  self.frame_store.C, self.n_step_return)
            return StepBatch(
                agent_inputs=AgentInputs(observation=obs, prev_action=pa, prev_reward=pr),
                action=act, return_=ret, done=dn, done_n=dnn,
                target_inputs=AgentInputs(observation=nxt_obs, prev_action=tpa, prev_reward=tpr))
        nxt = (T_idxs + self.n_step_return) % self.T
        was_done = row(d.done, T_idxs - 1, B_idxs)
        pa, pr = row(d.action, T_idxs - 1, B_idxs), row(d.reward, T_idxs - 1, B_idxs)
        pa = torch.where(was_done.reshape((-1,) + (1,) * (pa.dim() - 1)), torch.zeros_like(pa), pa)
        pr = torch.where(was_done, torch.zeros_like(pr), pr)
        if self.frame_store is not None:      # both 4-frame stacks in ONE launch
            obs, nxt_obs = ops.frames_gather_pair(self.frame_store.frames, d.done, T_idxs, B_idxs,
                                                  self.frame_store.C, self.n_step_return)
        else:
            obs, nxt_obs = row(d.observation, T_idxs, B_idxs), row(d.observation, nxt, B_idxs)
        return StepBatch(
            agent_inputs=AgentInputs(observation=obs, prev_action=pa, prev_reward=pr),
            action=row(d.action, T_idxs, B_idxs),
            return_=row(ret_ring, T_idxs, B_idxs),
            done=row(d.done, T_idxs, B_idxs),
            done_n=row(dn_ring, T_idxs, B_idxs),
            target_inputs=AgentInputs(observation=nxt_obs,
                                      prev_action=row(d.action, nxt - 1, B_idxs),
                                      prev_reward=row(d.reward, nxt - 1, B_idxs)))

    @staticmethod
    def _fused_fields_ok(d, ret_ring, dn_ring):
        """The one-launch field gather covers the standard record: scalar int64 actions, float32
        reward / return, bool done flags, all plain contiguous ``[T, B]`` device tensors; anything
        else takes the row-by-row gathers below (measured on config #3: ``sample_batch`` 218 -> 92 us,
        profiles/r5_dqn_knobs.txt)."""
        t = torch.Tensor
        return (isinstance(d.action, t) and d.action.dtype == torch.int64 and d.action.dim() == 2
                and isinstance(d.reward, t) and d.reward.dtype == torch.float32 and d.reward.dim() == 2
                and d.done.dtype == torch.bool and ret_ring.dtype == torch.float32
                and dn_ring.dtype == torch.bool and d.action.is_cuda
                and all(x.is_contiguous() for x in (d.action, d.reward, d.done, ret_ring, dn_ring)))

    def _sequences(self, T_idxs, B_idxs, T):
        """``[T (+ n_step), B]`` sequences starting at ``(t, b)``: observations from row t, the
        action / reward streams from row t - 1 (so entry k is what ENTERED step k), plus the RNN
        state stored for the start row."""
        d, seq, k = self.fields.data, ops.extract_sequences, self.rnn_state_interval
        span = T + self.n_step_return
        if self.rnn_store is not None:
            state = buffer_func(self.rnn_store.data, ops.gather_rows, T_idxs // k, B_idxs)
        elif k == 1:
            state = buffer_func(d.prev_rnn_state, ops.gather_rows, T_idxs, B_idxs)
        else:
            state = None
        return SeqBatch(
            all_observation=self.extract_observation(T_idxs, B_idxs, span),
            all_action=buffer_func(d.action, seq, T_idxs - 1, B_idxs, span),
            all_reward=seq(d.reward, T_idxs - 1, B_idxs, span),
            return_=seq(self.fields.return_, T_idxs, B_idxs, T),
            done=seq(d.done, T_idxs, B_idxs, T),
            done_n=seq(self.fields.done_n, T_idxs, B_idxs, T),
            init_rnn_state=state)

    def extract_observation(self, T_idxs, B_idxs, T=None):
        """Observations at ``(t, b)`` (single steps) or ``[T, n]`` sequences from there: frame
        buffers re-assemble the C-frame stacks with post-reset blanking in one gather kernel."""
        T_idxs, B_idxs = as_index(T_idxs, self.device), as_index(B_idxs, self.device)
        fs, d = self.frame_store, self.fields.data
        if T is None:
            return (ops.frames_gather(fs.frames, d.done, T_idxs, B_idxs, fs.C) if fs is not None
                    else ops.gather_rows(d.observation, T_idxs, B_idxs))
        if fs is not None:
            return ops.frames_gather_seq(fs.frames, d.done, T_idxs, B_idxs, fs.C, T)
        return buffer_func(d.observation, ops.extract_sequences, T_idxs, B_idxs, T)


# ---- the reference's class names: constructor signatures as recorded in protocol.json -----------
class UniformReplayBuffer(ReplayBuffer):
    def __init__(self, example, size, B, discount=1, n_step_return=1, device=None):
        self._build(example, size, B, discount, n_step_return, device)


class UniformReplayFrameBuffer(ReplayBuffer):
    FRAMES = True

    def __init__(self, example, **kwargs):
        self._build(example, **kwargs)


class UniformSequenceReplayBuffer(UniformReplayFrameBuffer):
    FRAMES, SEQUENCE = False, True


class UniformSequenceReplayFrameBuffer(UniformReplayFrameBuffer):
    SEQUENCE = True


class PrioritizedReplayBuffer(ReplayBuffer):
    PRIORITIZED = True

    def __init__(self, alpha=0.6, beta=0.4, default_priority=1, unique=False,
                 input_priorities=False, input_priority_shift=0, **kwargs):
        self.alpha, self.beta, self.default_priority = alpha, beta, default_priority
        self._build(alpha=alpha, beta=beta, default_priority=default_priority, unique=unique,
                    input_priorities=input_priorities, input_priority_shift=input_priority_shift,
                    **kwargs)


class PrioritizedReplayFrameBuffer(PrioritizedReplayBuffer):
    FRAMES = True


class PrioritizedSequenceReplayBuffer(PrioritizedReplayBuffer):
    SEQUENCE = True


class PrioritizedSequenceReplayFrameBuffer(PrioritizedReplayBuffer):
    FRAMES = SEQUENCE = True


_BY_SWITCHES = {(c.FRAMES, c.SEQUENCE, c.PRIORITIZED): c for c in (
    UniformReplayBuffer, UniformReplayFrameBuffer, UniformSequenceReplayBuffer,
    UniformSequenceReplayFrameBuffer, PrioritizedReplayBuffer, PrioritizedReplayFrameBuffer,
    PrioritizedSequenceReplayBuffer, PrioritizedSequenceReplayFrameBuffer)}


def replay_class(frames, sequence, prioritized):
    """The buffer class for a combination of the three switches."""
    return _BY_SWITCHES[(bool(frames), bool(sequence), bool(prioritized))]
