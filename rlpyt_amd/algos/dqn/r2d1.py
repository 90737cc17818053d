"""R2D1 -- recurrent-replay DQN -- on the MI355X (API and hyper-parameters of
rlpyt/algos/dqn/r2d1.py:24-345).

The sequence replay (frames, small fields, stored LSTM states, f64 sum tree) lives in HBM
(``rlpyt_amd/replays/sequence.py``); per update the ``[T+n, B]`` sequences are assembled by the
sequence-gather kernels, the network passes run through PyTorch-ROCm (conv + MIOpen LSTM), and
everything after the network outputs -- action / double-DQN target selection, value rescaling
h / h^-1, n-step target, (Huber) TD loss, IS weights, the ``valid`` mask, ``dL/dq`` and the
sequence priorities eta*max + (1-eta)*mean -- is ONE kernel (``rlpyt_r2d1_loss_fwd_bwd_f32``).
Input priorities of fresh samples (r2d1.py:181-242) are computed on the device as well."""
import os
from collections import namedtuple

import torch

from ... import ops
from ...agents.base import AgentInputs
from ...utils.buffer import buffer_method
from ...utils.tensor import select_at_indexes, valid_mean
from .dqn import DQN
from .replay_algo import Prioritised, ReplayFeed

OptInfo = namedtuple("OptInfo", ["loss", "gradNorm", "tdAbsErr", "priority"])


class R2D1(DQN):
    opt_info_fields = tuple(OptInfo._fields)
    OptInfoCls = OptInfo
    SEQUENCE_REPLAY = True
    CAPTURABLE = False      # sequence batches: three outputs of ``loss``, kernel-bound updates
    # double DQN's action-selection pass (r2d1.py:300-303: the ONLINE network, no grad, over the batch_T +
    # n_step steps behind the warm-up, from the warmed-up state) repeats the training pass on its first
    # batch_T steps -- same weights, same inputs, same initial state.  True: take those q-values from
    # the training pass and run only the n_step steps behind it from the state the training pass ends in
    # (one pass of the whole network over batch_T x B fewer steps per update); False: the reference's
    # statement sequence.
    share_online_pass = os.environ.get("RLPYT_R2D1_SHARE_ONLINE", "1") != "0"

    def __init__(self, discount=0.997, batch_T=80, batch_B=64, warmup_T=40,
                 store_rnn_state_interval=40, min_steps_learn=int(1e5), delta_clip=None,
                 replay_size=int(1e6), replay_ratio=1, target_update_interval=2500,
                 n_step_return=5, learning_rate=1e-4, OptimCls=torch.optim.Adam,
                 optim_kwargs=None, initial_optim_state_dict=None, clip_grad_norm=80.,
                 eps_steps=int(1e6), double_dqn=True, prioritized_replay=True, pri_alpha=0.6,
                 pri_beta_init=0.9, pri_beta_final=0.9, pri_beta_steps=int(50e6), pri_eta=0.9,
                 default_priority=None, input_priorities=True, input_priority_shift=None,
                 value_scale_eps=1e-3, ReplayBufferCls=None, updates_per_sync=1):
        hp = dict(locals())
        hp.pop("self")
        hp["optim_kwargs"] = dict(eps=1e-3) if optim_kwargs is None else optim_kwargs
        hp["default_priority"] = (delta_clip or 1.) if default_priority is None else default_priority
        if input_priority_shift is None:      # priorities of a batch belong to sequences that
            hp["input_priority_shift"] = warmup_T // store_rnn_state_interval   # started earlier
        hp["target_update_tau"] = 1
        self.__dict__.update(hp)
        # a "training batch" for the replay-ratio arithmetic: every stored step of every sequence
        self._batch_size = (batch_T + warmup_T) * batch_B
        self.update_counter = 0

    def make_feed(self):
        keeps_state = self.store_rnn_state_interval > 0
        return ReplayFeed(ReplayFeed.RNN if keeps_state else ReplayFeed.STEP,
                          "SamplesToBufferRnn" if keeps_state else "SamplesToBuffer")

    def replay_settings(self, batch_spec):
        kw = super().replay_settings(batch_spec)
        kw.update(rnn_state_interval=self.store_rnn_state_interval,
                  batch_T=self.batch_T + self.warmup_T)
        if self.prioritized_replay:
            kw.update(input_priorities=self.input_priorities,
                      input_priority_shift=self.input_priority_shift)
        return kw

    def samples_to_buffer(self, samples):
        """Sampler batch -> replay record, with input priorities when asked for (r2d1.py:167-179)."""
        record = self.feed.from_samples(samples)
        if self.input_priorities:      # fresh sequences enter the tree with their own TD errors
            record = Prioritised(priorities=self.input_priorities_of(samples), samples=record)
        return record

    def ingest(self, samples):
        self.replay_buffer.append_samples(self.samples_to_buffer(samples))

    def one_update(self, log):
        batch = self.replay_buffer.sample_batch(self.batch_B)
        self.optimizer.zero_grad(set_to_none=True)
        loss, td_abs, priorities = self.loss(batch)
        loss.backward(ops.unit_seed(loss.device) if loss.is_cuda else None)
        grad_norm = self.clip_and_step()
        if self.prioritized_replay:
            self.replay_buffer.update_batch_priorities(priorities)
        log.add((loss, grad_norm), tdAbsErr=td_abs[::8], priority=priorities)   # r2d1.py:166

    @torch.no_grad()
    def input_priorities_of(self, samples):
        """One priority per env column of a fresh sampler batch, on the device: the n-step TD
        errors implied by the Q-values recorded while sampling (value-rescaled targets), reduced
        over the valid time steps as ``eta * max + (1 - eta) * mean`` (r2d1.py:181-242)."""
        n, eta = self.n_step_return, self.pri_eta
        lag = max(1, n - 1)
        q = samples.agent.agent_info.q
        chosen = select_at_indexes(samples.agent.action, q)[:-lag]
        best_later = q.max(dim=-1).values[lag:]
        ret, done_n = ops.discount_return_n_step(samples.env.reward, samples.env.done, n,
                                                 self.discount, do_truncated=False)
        target = self.squash(ret + (1 - done_n.float()) * self.unsquash(best_later))
        err = (chosen - target).abs()
        if self.delta_clip is not None:
            err = err.clamp(0, self.delta_clip)
        live = ops.valid_from_done(samples.env.done[:-lag].contiguous())
        return eta * (err * live).max(dim=0).values + (1 - eta) * valid_mean(err, live, dim=0)

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
        qs, end_rnn_state = self.agent(*inputs(agent_slice), init_rnn_state)   # [bT, B, A]
        with torch.no_grad():
            target_qs, _ = self.agent.target(*inputs(target_slice), target_rnn_state)
            next_qs = None
            if self.double_dqn:
                n_tail = all_observation.shape[0] - (wT + bT)      # = n_step_return
                if self.share_online_pass and 0 < n_tail <= bT and end_rnn_state is not None:
                    # steps [n_tail, bT) of the training pass + the n_tail steps behind it
                    tail_qs, _ = self.agent(*inputs(slice(wT + bT, None)),
                                            buffer_method(end_rnn_state, "detach"))
                    next_qs = torch.cat([qs.detach()[n_tail:], tail_qs.to(qs.dtype)])
                else:
                    next_qs, _ = self.agent(*inputs(target_slice), init_rnn_state)
                    next_qs = next_qs[-bT:]
            target_qs = target_qs[-bT:]
        valid = ops.valid_from_done(samples.done[wT:].contiguous())
        is_weights = samples.is_weights if self.prioritized_replay else None
        return ops.r2d1_loss(qs, target_qs, next_qs, action, return_, done_n, valid, is_weights,
                             self.discount ** self.n_step_return, self.delta_clip,
                             self.value_scale_eps, self.pri_eta)

    def squash(self, x):
        return torch.sign(x) * (torch.sqrt(torch.abs(x) + 1) - 1) + self.value_scale_eps * x

    def unsquash(self, z):
        e = self.value_scale_eps
        return torch.sign(z) * (((torch.sqrt(1 + 4 * e * (torch.abs(z) + 1 + e)) - 1)
                                 / (2 * e)) ** 2 - 1)
