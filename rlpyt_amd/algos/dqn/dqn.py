"""DQN on the MI355X (API and hyper-parameters of rlpyt/algos/dqn/dqn.py:20-279).

The replay ring, the f64 sum tree, frame re-assembly, target selection, Huber TD loss and
its gradient all stay in HBM; per update the host only draws the uniforms and launches
kernels.  Diagnostics are read back once per ``optimize_agent`` call."""
from collections import namedtuple

import torch

from ... import ops
from ...replays.non_sequence import PrioritizedReplayFrameBuffer, UniformReplayFrameBuffer
from ...utils import logger
from ...utils.collections import namedarraytuple
from ...utils.quick_args import save__init__args
from ..base import RlAlgorithm

OptInfo = namedtuple("OptInfo", ["loss", "gradNorm", "tdAbsErr"])
SamplesToBuffer = namedarraytuple("SamplesToBuffer",
                                  ["observation", "action", "reward", "done"])


class DQN(RlAlgorithm):
    opt_info_fields = tuple(OptInfo._fields)

    def __init__(self, discount=0.99, batch_size=32, min_steps_learn=int(5e4), delta_clip=1.,
                 replay_size=int(1e6), replay_ratio=8, target_update_tau=1,
                 target_update_interval=312, n_step_return=1, learning_rate=2.5e-4,
                 OptimCls=torch.optim.Adam, optim_kwargs=None, initial_optim_state_dict=None,
                 clip_grad_norm=10., eps_steps=int(1e6), double_dqn=False,
                 prioritized_replay=False, pri_alpha=0.6, pri_beta_init=0.4,
                 pri_beta_final=1., pri_beta_steps=int(50e6), default_priority=None,
                 ReplayBufferCls=None, updates_per_sync=1):
        if optim_kwargs is None:
            optim_kwargs = dict(eps=0.01 / batch_size)
        if default_priority is None:
            default_priority = delta_clip
        self._batch_size = batch_size
        del batch_size
        save__init__args(locals())
        self.update_counter = 0

    def initialize(self, agent, n_itr, batch_spec, mid_batch_reset, examples, world_size=1,
                   rank=0):
        self.agent = agent
        self.n_itr = n_itr
        self.sampler_bs = sampler_bs = batch_spec.size
        self.mid_batch_reset = mid_batch_reset
        self.updates_per_optimize = max(1, round(self.replay_ratio * sampler_bs /
                                                 self.batch_size))
        logger.log(f"From sampler batch size {batch_spec.size}, training batch size "
                   f"{self.batch_size}, and replay ratio {self.replay_ratio}, computed "
                   f"{self.updates_per_optimize} updates per iteration.")
        self.min_itr_learn = int(self.min_steps_learn // sampler_bs)
        eps_itr_max = max(1, int(self.eps_steps // sampler_bs))
        agent.set_epsilon_itr_min_max(self.min_itr_learn, eps_itr_max)
        self.initialize_replay_buffer(examples, batch_spec)
        self.optim_initialize(rank)

    def optim_initialize(self, rank=0):
        self.rank = rank
        self.optimizer = self.make_optimizer(self.agent.parameters(), self.OptimCls,
                                             self.learning_rate, self.optim_kwargs)
        if self.initial_optim_state_dict is not None:
            self.optimizer.load_state_dict(self.initial_optim_state_dict)
        if self.prioritized_replay:
            self.pri_beta_itr = max(1, self.pri_beta_steps // self.sampler_bs)

    def initialize_replay_buffer(self, examples, batch_spec, async_=False):
        example_to_buffer = self.examples_to_buffer(examples)
        replay_kwargs = dict(example=example_to_buffer, size=self.replay_size, B=batch_spec.B,
                             discount=self.discount, n_step_return=self.n_step_return,
                             device=self.agent.device)
        if self.prioritized_replay:
            replay_kwargs.update(dict(alpha=self.pri_alpha, beta=self.pri_beta_init,
                                      default_priority=self.default_priority))
            ReplayCls = PrioritizedReplayFrameBuffer
        else:
            ReplayCls = UniformReplayFrameBuffer
        if self.ReplayBufferCls is not None:
            ReplayCls = self.ReplayBufferCls
            logger.log(f"WARNING: ignoring internal selection logic and using input replay "
                       f"buffer class: {ReplayCls} -- compatibility not guaranteed.")
        self.replay_buffer = ReplayCls(**replay_kwargs)

    def examples_to_buffer(self, examples):
        return SamplesToBuffer(observation=examples["observation"], action=examples["action"],
                               reward=examples["reward"], done=examples["done"])

    def samples_to_buffer(self, samples):
        return SamplesToBuffer(observation=samples.env.observation,
                               action=samples.agent.action, reward=samples.env.reward,
                               done=samples.env.done)

    def optimize_agent(self, itr, samples=None, sampler_itr=None):
        itr = itr if sampler_itr is None else sampler_itr
        if samples is not None:
            self.replay_buffer.append_samples(self.samples_to_buffer(samples))
        opt_info = OptInfo(*([] for _ in range(len(OptInfo._fields))))
        if itr < self.min_itr_learn:
            return opt_info
        stats, tds = [], []
        for _ in range(self.updates_per_optimize):
            samples_from_replay = self.replay_buffer.sample_batch(self.batch_size)
            self.optimizer.zero_grad(set_to_none=True)
            loss, td_abs_errors = self.loss(samples_from_replay)
            loss.backward()
            grad_norm = self.clip_and_step()
            if self.prioritized_replay:
                self.replay_buffer.update_batch_priorities(td_abs_errors)
            stats.append(torch.stack([loss.detach(), grad_norm.to(loss.dtype)]))
            tds.append(td_abs_errors[::8])      # per update, as dqn.py:186
            self.update_counter += 1
            if self.update_counter % self.target_update_interval == 0:
                self.agent.update_target(self.target_update_tau)
        host = torch.stack(stats).cpu().tolist()
        opt_info.loss.extend(r[0] for r in host)
        opt_info.gradNorm.extend(r[1] for r in host)
        opt_info.tdAbsErr.extend(torch.cat(tds).cpu().tolist())
        self.update_itr_hyperparams(itr)
        return opt_info

    def loss(self, samples):
        """dqn.py:211-265 with the arithmetic after the network outputs fused in one kernel."""
        if not self.mid_batch_reset:
            raise NotImplementedError  # as the reference (dqn.py:254-259)
        qs = self.agent(*samples.agent_inputs)
        with torch.no_grad():
            target_qs = self.agent.target(*samples.target_inputs)
            next_qs = self.agent(*samples.target_inputs) if self.double_dqn else None
        is_weights = samples.is_weights if self.prioritized_replay else None
        return ops.dqn_loss(qs, target_qs, next_qs, samples.action, samples.return_,
                            samples.done_n, is_weights, self.discount ** self.n_step_return,
                            self.delta_clip)

    def update_itr_hyperparams(self, itr):
        if self.prioritized_replay and itr <= self.pri_beta_itr:
            prog = min(1, max(0, itr - self.min_itr_learn) /
                       (self.pri_beta_itr - self.min_itr_learn))
            new_beta = prog * self.pri_beta_final + (1 - prog) * self.pri_beta_init
            self.replay_buffer.set_beta(new_beta)
