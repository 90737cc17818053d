"""DQN on the MI355X (constructor and runner protocol of rlpyt/algos/dqn/dqn.py:20-279).

The replay ring, the f64 sum tree, frame re-assembly, target selection, Huber TD loss and its
gradient all stay in HBM (``rlpyt_amd/replays``, ``csrc/{sumtree,gather,loss}.hip``); per update
the host draws the batch's uniforms and launches kernels.  Host-side structure (see
``replay_algo.py``): the iteration arithmetic is an ``UpdatePlan``, sampler batches reach the
buffer through a ``ReplayFeed`` table, the buffer class comes from ``replays.replay_class`` (or the
reference's own injection hook, ``ReplayBufferCls``), diagnostics from an ``UpdateLog`` that is read
back once per ``optimize_agent`` call."""
from collections import namedtuple

import torch

from ... import ops
from ...replays.buffers import replay_class
from ...utils import logger
from ..base import RlAlgorithm
from .replay_algo import BetaAnneal, ReplayFeed, UpdateLog, plan_updates

OptInfo = namedtuple("OptInfo", ["loss", "gradNorm", "tdAbsErr"])


class DQN(RlAlgorithm):
    opt_info_fields = tuple(OptInfo._fields)
    OptInfoCls = OptInfo
    SEQUENCE_REPLAY = False
    # one update = one captured hipGraph once the optimizer state exists (algos/dqn/captured.py);
    # subclasses whose update is not ``sample_batch -> loss -> clip_and_step`` switch it off
    CAPTURABLE = True
    _captured = None

    def __init__(self, discount=0.99, batch_size=32, min_steps_learn=int(5e4), delta_clip=1.,
                 replay_size=int(1e6), replay_ratio=8, target_update_tau=1,
                 target_update_interval=312, n_step_return=1, learning_rate=2.5e-4,
                 OptimCls=torch.optim.Adam, optim_kwargs=None, initial_optim_state_dict=None,
                 clip_grad_norm=10., eps_steps=int(1e6), double_dqn=False,
                 prioritized_replay=False, pri_alpha=0.6, pri_beta_init=0.4,
                 pri_beta_final=1., pri_beta_steps=int(50e6), default_priority=None,
                 ReplayBufferCls=None, updates_per_sync=1):
        hp = dict(locals())
        hp.pop("self")
        hp["optim_kwargs"] = dict(eps=0.01 / batch_size) if optim_kwargs is None else optim_kwargs
        hp["default_priority"] = delta_clip if default_priority is None else default_priority
        self._batch_size = hp.pop("batch_size")
        self.__dict__.update(hp)
        self.update_counter = 0
        self._captured = None

    # ------------------------------------------------------------------ set-up
    def initialize(self, agent, n_itr, batch_spec, mid_batch_reset, examples, world_size=1,
                   rank=0):
        self._setup(agent, n_itr, batch_spec, mid_batch_reset, examples, world_size, async_=False)
        self.optim_initialize(rank)

    def async_initialize(self, agent, sampler_n_itr, batch_spec, mid_batch_reset, examples,
                         world_size=1):
        """Asynchronous runners only (rlpyt/algos/dqn/dqn.py:99-115): builds the replay buffer in its
        asynchronous form (``replays/async_.py``: appends from the sampler side while the optimizer
        side draws batches) and returns it; the optimizer comes later (``optim_initialize``).  The
        number of updates per ``optimize_agent`` call is ``updates_per_sync`` there (the replay ratio
        is enforced by the runner's throttle instead)."""
        self._setup(agent, sampler_n_itr, batch_spec, mid_batch_reset, examples, world_size,
                    async_=True)
        self.updates_per_optimize = self.updates_per_sync
        return self.replay_buffer

    def optim_initialize(self, rank=0):
        """Called by ``initialize`` or, in asynchronous mode, by the runner once the sampler side is up
        (dqn.py:117-125)."""
        self.rank = rank
        self.optimizer = self.make_optimizer(self.agent.parameters(), self.OptimCls,
                                             self.learning_rate, self.optim_kwargs)
        if self.initial_optim_state_dict is not None:
            self.optimizer.load_state_dict(self.initial_optim_state_dict)

    def samples_to_buffer(self, samples):
        """Sampler batch -> replay record (dqn.py:192-200); what a memory copier appends."""
        return self.feed.from_samples(samples)

    def _setup(self, agent, n_itr, batch_spec, mid_batch_reset, examples, world_size, async_):
        self.agent, self.n_itr, self.world_size = agent, n_itr, world_size
        self.rank, self.async_ = 0, bool(async_)
        self.mid_batch_reset = mid_batch_reset
        self.sampler_bs = batch_spec.size
        plan = self.plan = plan_updates(batch_spec.size, self.batch_size, self.replay_ratio,
                                        self.min_steps_learn, self.eps_steps, self.pri_beta_steps)
        self.updates_per_optimize, self.min_itr_learn = plan.updates_per_itr, plan.first_learn_itr
        logger.log(f"DQN-family update plan: sampler batch {batch_spec.size}, training batch "
                   f"{self.batch_size}, replay ratio {self.replay_ratio} -> {plan.updates_per_itr} "
                   f"updates per iteration, learning from iteration {plan.first_learn_itr}.")
        agent.set_epsilon_itr_min_max(plan.first_learn_itr, plan.eps_last_itr)
        self.feed = self.make_feed()
        self.replay_buffer = self.make_replay(self.feed.from_examples(examples), batch_spec)
        self.beta = (BetaAnneal(self.pri_beta_init, self.pri_beta_final, plan.first_learn_itr,
                                plan.beta_last_itr) if self.prioritized_replay else None)

    def make_feed(self):
        return ReplayFeed(ReplayFeed.STEP, "SamplesToBuffer")

    def replay_settings(self, batch_spec):
        """Constructor keywords of the replay buffer beyond the example record."""
        kw = dict(size=self.replay_size, B=batch_spec.B, discount=self.discount,
                  n_step_return=self.n_step_return, device=self.agent.device)
        if self.prioritized_replay:
            kw.update(alpha=self.pri_alpha, beta=self.pri_beta_init,
                      default_priority=self.default_priority)
        return kw

    def make_replay(self, example, batch_spec):
        Cls = self.ReplayBufferCls
        if Cls is None:
            pick = replay_class
            if getattr(self, "async_", False):
                from ...replays.async_ import async_replay_class as pick
            Cls = pick(frames=True, sequence=self.SEQUENCE_REPLAY, prioritized=self.prioritized_replay)
        else:     # the reference's injection hook (dqn.py:56,151-156)
            logger.log(f"DQN: replay buffer class supplied by the caller ({Cls.__name__}); its "
                       "constructor gets the keywords of the built-in choice.")
        return Cls(example=example, **self.replay_settings(batch_spec))

    # ------------------------------------------------------------------ per iteration
    def ingest(self, samples):
        """New sampler batch -> replay ring."""
        self.replay_buffer.append_samples(self.samples_to_buffer(samples))

    def optimize_agent(self, itr, samples=None, sampler_itr=None):
        itr = itr if sampler_itr is None else sampler_itr
        if samples is not None:
            self.ingest(samples)
        log = UpdateLog(self.OptInfoCls, ("loss", "gradNorm"))
        if itr >= self.min_itr_learn:
            if self._captured is None:
                from .captured import CapturedUpdates
                self._captured = CapturedUpdates(self)
            if not self._captured.run(itr, log):
                for _ in range(self.updates_per_optimize):
                    self.one_update(log)
                    self.update_counter += 1
                    if self.update_counter % self.target_update_interval == 0:
                        self.agent.update_target(self.target_update_tau)
            if self.beta is not None:
                new_beta = self.beta.at(itr)
                if new_beta is not None:
                    self.replay_buffer.set_beta(new_beta)
        return log.to_opt_info()

    def one_update(self, log):
        batch = self.replay_buffer.sample_batch(self.batch_size)
        self.optimizer.zero_grad(set_to_none=True)
        loss, td_abs = self.loss(batch)
        loss.backward(ops.unit_seed(loss.device) if loss.is_cuda else None)
        grad_norm = self.clip_and_step()
        if self.prioritized_replay:
            self.replay_buffer.update_batch_priorities(td_abs)
        log.add((loss, grad_norm), tdAbsErr=td_abs[::8])      # every 8th, as dqn.py:186

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
