"""Non-sequence replay buffers on the device: uniform and prioritized, plain and
frame-based (classes and call signatures of rlpyt/replays/non_sequence/{n_step,uniform,
prioritized,frame}.py).

``sample_batch`` never leaves HBM: tree descent (``rlpyt_sumtree_sample``), 4-frame
observation re-assembly for agent and target inputs (``rlpyt_frames_gather``), the small
field gathers (``rlpyt_gather_rows``) and the importance weights all run on the device;
the only host -> device traffic is the ``n`` float64 uniforms drawn from ``np.random.rand``
(kept on the host RNG so a seeded run draws the same stream as the reference,
rlpyt/replays/sum_tree.py:107).
"""
import numpy as np
import torch

from .. import ops
from ..agents.base import AgentInputs
from ..utils.collections import namedarraytuple
from ..utils.quick_args import save__init__args
from .n_step import BaseNStepReturnBuffer, FrameBufferMixin

SamplesFromReplay = namedarraytuple("SamplesFromReplay",
                                    ["agent_inputs", "action", "return_", "done", "done_n",
                                     "target_inputs"])
SamplesFromReplayPri = namedarraytuple("SamplesFromReplayPri",
                                       SamplesFromReplay._fields + ("is_weights",))
EPS = 1e-6  # rlpyt/replays/non_sequence/prioritized.py:9


class NStepReturnBuffer(BaseNStepReturnBuffer):
    def extract_batch(self, T_idxs, B_idxs):
        """Gather the training fields at [T_idxs, B_idxs] and the target inputs at
        T_idxs + n_step (rlpyt/replays/non_sequence/n_step.py:16-43)."""
        T_idxs = self._idx(T_idxs)
        B_idxs = self._idx(B_idxs)
        s = self.samples
        target_T = (T_idxs + self.n_step_return) % self.T
        prev_T = T_idxs - 1  # negative index wraps to the ring end (numpy rule)
        prev_action = ops.gather_rows(s.action, prev_T, B_idxs)
        prev_reward = ops.gather_rows(s.reward, prev_T, B_idxs)
        t_news = ops.gather_rows(s.done, prev_T, B_idxs)
        prev_action = torch.where(t_news.reshape((-1,) + (1,) * (prev_action.dim() - 1)),
                                  torch.zeros_like(prev_action), prev_action)
        prev_reward = torch.where(t_news, torch.zeros_like(prev_reward), prev_reward)
        obs, target_obs = self.extract_observation_pair(T_idxs, target_T, B_idxs)
        return SamplesFromReplay(
            agent_inputs=AgentInputs(observation=obs,
                                     prev_action=prev_action, prev_reward=prev_reward),
            action=ops.gather_rows(s.action, T_idxs, B_idxs),
            return_=ops.gather_rows(self.samples_return_, T_idxs, B_idxs),
            done=ops.gather_rows(s.done, T_idxs, B_idxs),
            done_n=ops.gather_rows(self.samples_done_n, T_idxs, B_idxs),
            target_inputs=AgentInputs(
                observation=target_obs,
                prev_action=ops.gather_rows(s.action, target_T - 1, B_idxs),
                prev_reward=ops.gather_rows(s.reward, target_T - 1, B_idxs)))

    def _idx(self, x):
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(x)
        return x.to(device=self.device, dtype=torch.int64)

    def extract_observation(self, T_idxs, B_idxs):
        return ops.gather_rows(self.samples.observation, T_idxs, B_idxs)

    def extract_observation_pair(self, T_idxs, target_T, B_idxs):
        """(agent observation at T_idxs, target observation at target_T = T_idxs + n_step)."""
        return (self.extract_observation(T_idxs, B_idxs),
                self.extract_observation(target_T, B_idxs))


class UniformReplay:
    def sample_batch(self, batch_B):
        T_idxs, B_idxs = self.sample_idxs(batch_B)
        return self.extract_batch(T_idxs, B_idxs)

    def sample_idxs(self, batch_B):
        """rlpyt/replays/non_sequence/uniform.py:17-28 (same np.random draws)."""
        t, b, f = self.t, self.off_backward, self.off_forward
        high = self.T - b - f if self._buffer_full else t - b
        low = 0 if self._buffer_full else f
        T_idxs = np.random.randint(low=low, high=high, size=(batch_B,))
        T_idxs[T_idxs >= t - b] += min(t, b) + f
        B_idxs = np.random.randint(low=0, high=self.B, size=(batch_B,))
        return T_idxs, B_idxs


class PrioritizedReplay:
    """Sum-tree prioritized replay (rlpyt/replays/non_sequence/prioritized.py:15-79)."""

    def __init__(self, alpha=0.6, beta=0.4, default_priority=1, unique=False,
                 input_priorities=False, input_priority_shift=0, **kwargs):
        super().__init__(**kwargs)
        save__init__args(locals())
        if unique:
            raise NotImplementedError("unique=True sampling is not on the hot path")
        self.init_priority_tree()

    def init_priority_tree(self):
        self.priority_tree = ops.DeviceSumTree(
            T=self.T, B=self.B, off_backward=self.off_backward, off_forward=self.off_forward,
            default_value=self.default_priority ** self.alpha,
            enable_input_priorities=self.input_priorities,
            input_priority_shift=self.input_priority_shift, device=self.device)

    def set_beta(self, beta):
        self.beta = beta

    def append_samples(self, samples):
        if hasattr(samples, "priorities"):
            priorities = torch.as_tensor(samples.priorities, device=self.device).double() \
                ** self.alpha
            samples = samples.samples
        else:
            priorities = None
        T, idxs = super().append_samples(samples)
        self.priority_tree.advance(T, priorities=priorities)
        return T, idxs

    def sample_batch(self, batch_B):
        u = torch.from_numpy(np.random.rand(int(batch_B))).to(self.device, non_blocking=True)
        T_idxs, B_idxs, priorities = self.priority_tree.sample(u)
        batch = self.extract_batch(T_idxs, B_idxs)
        is_weights = (1. / (priorities + EPS)) ** self.beta
        is_weights = (is_weights / is_weights.max()).float()
        return SamplesFromReplayPri(*batch, is_weights=is_weights)

    def update_batch_priorities(self, priorities):
        """priorities (e.g. |TD errors|, f32) ** alpha in the input dtype, as numpy does in
        the reference (prioritized.py:78-79), then the f64 tree update."""
        p = priorities.detach().to(self.device) ** self.alpha
        self.priority_tree.update_batch_priorities(p)


class NStepFrameBuffer(FrameBufferMixin, NStepReturnBuffer):
    def extract_observation(self, T_idxs, B_idxs):
        """4-frame stack + post-reset blanking
        (rlpyt/replays/non_sequence/frame.py:14-30) in one gather kernel."""
        return ops.frames_gather(self.samples_frames, self.samples.done, self._idx(T_idxs),
                                 self._idx(B_idxs), self.n_frames)


    def extract_observation_pair(self, T_idxs, target_T, B_idxs):
        """Both frame-stack gathers of a batch in one launch (8.5 MB each at batch 128 is
        latency-class: one launch of twice the rows moves them at about twice the rate)."""
        both = ops.frames_gather_pair(self.samples_frames, self.samples.done, self._idx(T_idxs),
                                      self._idx(B_idxs), self.n_frames, self.n_step_return)
        return both[0], both[1]


class UniformReplayBuffer(UniformReplay, NStepReturnBuffer):
    pass


class PrioritizedReplayBuffer(PrioritizedReplay, NStepReturnBuffer):
    pass


class UniformReplayFrameBuffer(UniformReplay, NStepFrameBuffer):
    pass


class PrioritizedReplayFrameBuffer(PrioritizedReplay, NStepFrameBuffer):
    pass
