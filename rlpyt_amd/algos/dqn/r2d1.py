"""R2D1 -- recurrent-replay DQN -- on the MI355X (API and hyper-parameters of
rlpyt/algos/dqn/r2d1.py:24-345).

The sequence replay (frames, small fields, stored LSTM states, f64 sum tree) lives in HBM
(``rlpyt_amd/replays/sequence.py``); per update the ``[T+n, B]`` sequences are assembled by the
sequence-gather kernels, the network passes run through PyTorch-ROCm (conv + MIOpen LSTM), and
everything after the network outputs -- action / double-DQN target selection, value rescaling
h / h^-1, n-step target, (Huber) TD loss, IS weights, the ``valid`` mask, ``dL/dq`` and the
sequence priorities eta*max + (1-eta)*mean -- is ONE kernel (``rlpyt_r2d1_loss_fwd_bwd_f32``).
Input priorities of fresh samples (r2d1.py:181-242) are computed on the device as well."""
from collections import namedtuple

import torch

from ... import ops
from ...agents.base import AgentInputs
from ...replays.sequence import (PrioritizedSequenceReplayFrameBuffer,
                                 UniformSequenceReplayFrameBuffer)
from ...utils import logger
from ...utils.buffer import buffer_method
from ...utils.collections import namedarraytuple
from ...utils.quick_args import save__init__args
from ...utils.tensor import select_at_indexes, valid_mean
from .dqn import DQN, SamplesToBuffer

OptInfo = namedtuple("OptInfo", ["loss", "gradNorm", "tdAbsErr", "priority"])
SamplesToBufferRnn = namedarraytuple("SamplesToBufferRnn",
                                     SamplesToBuffer._fields + ("prev_rnn_state",))
PrioritiesSamplesToBuffer = namedarraytuple("PrioritiesSamplesToBuffer",
                                            ["priorities", "samples"])


class R2D1(DQN):
    opt_info_fields = tuple(OptInfo._fields)

    def __init__(self, discount=0.997, batch_T=80, batch_B=64, warmup_T=40,
                 store_rnn_state_interval=40, min_steps_learn=int(1e5), delta_clip=None,
                 replay_size=int(1e6), replay_ratio=1, target_update_interval=2500,
                 n_step_return=5, learning_rate=1e-4, OptimCls=torch.optim.Adam,
                 optim_kwargs=None, initial_optim_state_dict=None, clip_grad_norm=80.,
                 eps_steps=int(1e6), double_dqn=True, prioritized_replay=True, pri_alpha=0.6,
                 pri_beta_init=0.9, pri_beta_final=0.9, pri_beta_steps=int(50e6), pri_eta=0.9,
                 default_priority=None, input_priorities=True, input_priority_shift=None,
                 value_scale_eps=1e-3, ReplayBufferCls=None, updates_per_sync=1):
        if optim_kwargs is None:
            optim_kwargs = dict(eps=1e-3)
        if default_priority is None:
            default_priority = delta_clip or 1.
        if input_priority_shift is None:
            input_priority_shift = warmup_T // store_rnn_state_interval
        target_update_tau = 1
        save__init__args(locals())
        self._batch_size = (self.batch_T + self.warmup_T) * self.batch_B
        self.update_counter = 0

    @property
    def batch_size(self):
        return self._batch_size

    def initialize_replay_buffer(self, examples, batch_spec, async_=False):
        example_to_buffer = SamplesToBuffer(observation=examples["observation"],
                                            action=examples["action"],
                                            reward=examples["reward"], done=examples["done"])
        if self.store_rnn_state_interval > 0:
            example_to_buffer = SamplesToBufferRnn(
                *example_to_buffer, prev_rnn_state=examples["agent_info"].prev_rnn_state)
        replay_kwargs = dict(example=example_to_buffer, size=self.replay_size, B=batch_spec.B,
                             discount=self.discount, n_step_return=self.n_step_return,
                             rnn_state_interval=self.store_rnn_state_interval,
                             batch_T=self.batch_T + self.warmup_T, device=self.agent.device)
        if self.prioritized_replay:
            replay_kwargs.update(dict(alpha=self.pri_alpha, beta=self.pri_beta_init,
                                      default_priority=self.default_priority,
                                      input_priorities=self.input_priorities,
                                      input_priority_shift=self.input_priority_shift))
            ReplayCls = PrioritizedSequenceReplayFrameBuffer
        else:
            ReplayCls = UniformSequenceReplayFrameBuffer
        if self.ReplayBufferCls is not None:
            ReplayCls = self.ReplayBufferCls
            logger.log(f"WARNING: ignoring internal selection logic and using input replay "
                       f"buffer class: {ReplayCls} -- compatibility not guaranteed.")
        self.replay_buffer = ReplayCls(**replay_kwargs)
        return self.replay_buffer

    def optimize_agent(self, itr, samples=None, sampler_itr=None):
        itr = itr if sampler_itr is None else sampler_itr
        if samples is not None:
            self.replay_buffer.append_samples(self.samples_to_buffer(samples))
        opt_info = OptInfo(*([] for _ in range(len(OptInfo._fields))))
        if itr < self.min_itr_learn:
            return opt_info
        stats, tds, pris = [], [], []
        for _ in range(self.updates_per_optimize):
            samples_from_replay = self.replay_buffer.sample_batch(self.batch_B)
            self.optimizer.zero_grad(set_to_none=True)
            loss, td_abs_errors, priorities = self.loss(samples_from_replay)
            loss.backward()
            grad_norm = self.clip_and_step()
            if self.prioritized_replay:
                self.replay_buffer.update_batch_priorities(priorities)
            stats.append(torch.stack([loss.detach(), grad_norm.to(loss.dtype)]))
            tds.append(td_abs_errors[::8].reshape(-1))     # every 8th time step (r2d1.py:166)
            pris.append(priorities)
            self.update_counter += 1
            if self.update_counter % self.target_update_interval == 0:
                self.agent.update_target()
        host = torch.stack(stats).cpu().tolist()    # one D2H per call for the diagnostics
        opt_info.loss.extend(r[0] for r in host)
        opt_info.gradNorm.extend(r[1] for r in host)
        opt_info.tdAbsErr.extend(torch.cat(tds).cpu().tolist())
        opt_info.priority.extend(torch.cat(pris).cpu().tolist())
        self.update_itr_hyperparams(itr)
        return opt_info

    def samples_to_buffer(self, samples):
        samples_to_buffer = super().samples_to_buffer(samples)
        if self.store_rnn_state_interval > 0:
            samples_to_buffer = SamplesToBufferRnn(
                *samples_to_buffer, prev_rnn_state=samples.agent.agent_info.prev_rnn_state)
        if self.input_priorities:
            samples_to_buffer = PrioritiesSamplesToBuffer(
                priorities=self.compute_input_priorities(samples), samples=samples_to_buffer)
        return samples_to_buffer

    @torch.no_grad()
    def compute_input_priorities(self, samples):
        """n-step TD errors from the Q-values recorded while sampling, value-rescaled, reduced
        over time to one priority per env column: eta*max + (1-eta)*mean over valid steps
        (r2d1.py:181-242), all on the device."""
        q = samples.agent.agent_info.q
        action = samples.agent.action
        q_max = torch.max(q, dim=-1).values
        q_at_a = select_at_indexes(action, q)
        return_n, done_n = ops.discount_return_n_step(samples.env.reward, samples.env.done,
                                                      self.n_step_return, self.discount,
                                                      do_truncated=False)
        nm1 = max(1, self.n_step_return - 1)
        y = self.value_scale(return_n + (1 - done_n.float()) * self.inv_value_scale(q_max[nm1:]))
        delta = torch.abs(q_at_a[:-nm1] - y)
        if self.delta_clip is not None:
            delta = torch.clamp(delta, 0, self.delta_clip)
        valid = ops.valid_from_done(samples.env.done[:-nm1].contiguous())
        max_d = torch.max(delta * valid, dim=0).values
        mean_d = valid_mean(delta, valid, dim=0)
        return self.pri_eta * max_d + (1 - self.pri_eta) * mean_d

    def loss(self, samples):
        """r2d1.py:244-334: warm the LSTM up on the first ``warmup_T`` steps (no grad), train on
        the next ``batch_T``; online and target nets start from the same stored state."""
        all_observation, all_action, all_reward = (samples.all_observation, samples.all_action,
                                                   samples.all_reward)
        wT, bT = self.warmup_T, self.batch_T
        agent_slice, target_slice = slice(wT, wT + bT), slice(wT, None)
        inputs = lambda sl: AgentInputs(all_observation[sl], all_action[sl],  # noqa: E731
                                        all_reward[sl])
        action = all_action[wT + 1:wT + 1 + bT]
        return_, done_n = samples.return_[wT:wT + bT], samples.done_n[wT:wT + bT]
        if self.store_rnn_state_interval == 0:
            init_rnn_state = None
        else:   # stored [B,N,H] -> [N,B,H]
            init_rnn_state = buffer_method(buffer_method(samples.init_rnn_state, "transpose", 0, 1),
                                           "contiguous")
        if wT > 0:
            with torch.no_grad():
                _, target_rnn_state = self.agent.target(*inputs(slice(None, wT)), init_rnn_state)
                _, init_rnn_state = self.agent(*inputs(slice(None, wT)), init_rnn_state)
                # an episode end inside the warm-up starts the training segment from zero state
                keep = ops.valid_from_done(samples.done[:wT].contiguous())[-1]      # [B] 1/0
                keep = keep.reshape(1, -1, 1)
                init_rnn_state = buffer_method(init_rnn_state, "mul", keep)
                target_rnn_state = buffer_method(target_rnn_state, "mul", keep)
        else:
            target_rnn_state = init_rnn_state
        qs, _ = self.agent(*inputs(agent_slice), init_rnn_state)               # [bT, B, A]
        with torch.no_grad():
            target_qs, _ = self.agent.target(*inputs(target_slice), target_rnn_state)
            next_qs = None
            if self.double_dqn:
                next_qs, _ = self.agent(*inputs(target_slice), init_rnn_state)
                next_qs = next_qs[-bT:]
            target_qs = target_qs[-bT:]
        valid = ops.valid_from_done(samples.done[wT:].contiguous())
        is_weights = samples.is_weights if self.prioritized_replay else None
        return ops.r2d1_loss(qs, target_qs, next_qs, action, return_, done_n, valid, is_weights,
                             self.discount ** self.n_step_return, self.delta_clip,
                             self.value_scale_eps, self.pri_eta)

    def value_scale(self, x):
        return torch.sign(x) * (torch.sqrt(torch.abs(x) + 1) - 1) + self.value_scale_eps * x

    def inv_value_scale(self, z):
        e = self.value_scale_eps
        return torch.sign(z) * (((torch.sqrt(1 + 4 * e * (torch.abs(z) + 1 + e)) - 1)
                                 / (2 * e)) ** 2 - 1)
