"""Training-loop drivers for this repo's Sampler / Algo / Agent classes.

The reference's own runners (rlpyt/runners/minibatch_rl.py:232-357, rlpyt/runners/sync_rl.py) drive
these classes unmodified wherever rlpyt is importable -- ``tests/test_protocol.py`` does exactly that
-- because the protocol they call is kept.  This module is the stand-in for
boxes without the reference (the GPU box, ``bench.py``, the GPU tests): the same constructor keywords
and ``train()``, the same tabular output (names and order pinned by tests/golden/runner_keys.json, which
was recorded from the reference's runners), its own structure:

* ``schedule()``     -- iteration count / logging period from ``n_steps`` and the global batch size;
* ``RunLog``         -- every number that reaches the log: optimisation-info accumulation, the
                        trajectory window, wall-clock rates; one ``emit`` per logging period whose
                        rows come from a table, not from a statement sequence;
* ``_Driver``        -- bring-up order (sampler forks its env workers BEFORE the first HIP call,
                        then device placement, DDP wrap, algorithm), the iteration, the report;
* ``MinibatchRl`` / ``MinibatchRlEval`` / ``SyncRl`` -- configurations of the driver.

``SyncRl``: one process PER GPU, launched from outside (``torch.distributed.run``, ``bench.py --gpus
N`` or ``launch_sync`` below) instead of forked from a master that already owns a HIP context; every
rank runs the same loop on its own ``[T, B]`` batch (weak scaling, sync_rl.py:40-45) and the only
coupling is DistributedDataParallel's gradient all-reduce -- RCCL over xGMI with the "nccl" backend.
"""
import os
import time
from collections import deque, namedtuple

import torch

from ..utils import logger
from ..utils.deferred import PendingOptInfo
from ..utils.seed import make_seed, set_seed

Schedule = namedtuple("Schedule", ["n_itr", "log_every", "itr_batch"])


def schedule(n_steps, itr_batch, log_interval_steps):
    """Whole logging periods covering ``n_steps`` env steps at ``itr_batch`` steps per iteration
    (the rounding of rlpyt/runners/minibatch_rl.py:108-118: up to the next multiple of the period)."""
    log_every = max(int(log_interval_steps) // itr_batch, 1)
    n_itr = -(-(int(n_steps) // itr_batch) // log_every) * log_every
    return Schedule(max(n_itr, 1), log_every, itr_batch)


class RunLog:
    """What a run reports.  ``absorb`` takes one iteration's trajectory records and optimisation
    info; ``emit`` writes one tabular row set.  Counter rows are listed in ``COUNTERS`` /
    ``TRAIN_HEAD`` / ``EVAL_HEAD`` as (key, field) pairs and filled from one dict of numbers."""

    TRAIN_HEAD = (("NewCompletedTrajs", "new_trajs"), ("StepsInTrajWindow", "window_steps"))
    EVAL_HEAD = (("StepsInEval", "eval_steps"), ("TrajsInEval", "eval_trajs"),
                 ("CumEvalTime", "eval_seconds"), ("CumTrainTime", "train_seconds"))
    COUNTERS = (("Iteration", "itr"), ("CumTime (s)", "seconds"), ("CumSteps", "steps"),
                ("CumCompletedTrajs", "trajs"), ("CumUpdates", "updates"),
                ("StepsPerSecond", "steps_per_s"), ("UpdatesPerSecond", "updates_per_s"),
                ("ReplayRatio", "replay_ratio"), ("CumReplayRatio", "cum_replay_ratio"))

    def __init__(self, opt_fields, traj_window=None, prefix="Diagnostics/"):
        self.prefix = prefix
        self.opt = {name: [] for name in opt_fields}
        self.waiting = []               # PendingOptInfo objects not read yet (utils/deferred.py)
        self.window = None if traj_window is None else deque(maxlen=int(traj_window))
        self.trajs = self.new_trajs = 0
        self.eval_seconds = 0.
        self.restart_clock()
        self.updates_seen = 0

    def restart_clock(self):
        self.t_start = self.t_mark = time.time()

    def absorb(self, traj_infos, opt_info):
        n = len(traj_infos)
        self.trajs += n
        self.new_trajs += n
        if self.window is not None:
            self.window.extend(traj_infos)
        if isinstance(opt_info, PendingOptInfo):      # its device -> host copy is in flight: read at
            self.waiting.append(opt_info)             # the period's end, not once per iteration
            if len(self.waiting) > 64:                # (bounded: pinned staging of the oldest goes back)
                self._settle(self.waiting.pop(0))
        else:
            self._settle(opt_info)

    def _settle(self, opt_info):
        for name, bucket in self.opt.items():
            got = getattr(opt_info, name, [])
            bucket += got if isinstance(got, list) else [got]

    def settle(self):
        """Bring the buckets up to date with every iteration absorbed so far."""
        for info in self.waiting:
            self._settle(info)
        self.waiting.clear()

    def drop_period(self):
        """Forget the period's accumulations without printing (non-logging ranks)."""
        self.waiting.clear()
        for bucket in self.opt.values():
            bucket.clear()
        self.new_trajs = 0

    def emit(self, itr, period_steps, cum_steps, algo, sampler_batch, world, eval_trajs=None,
             eval_seconds=0.):
        now = time.time()
        span = now - self.t_mark - eval_seconds
        d_updates = algo.update_counter - self.updates_seen
        first = itr == 0
        self.eval_seconds += eval_seconds
        num = dict(
            itr=itr, seconds=now - self.t_start, steps=cum_steps, trajs=self.trajs,
            updates=algo.update_counter, new_trajs=self.new_trajs,
            steps_per_s=float("nan") if first else period_steps / span,
            updates_per_s=float("nan") if first else d_updates / span,
            replay_ratio=d_updates * algo.batch_size * world / period_steps,
            cum_replay_ratio=algo.batch_size * algo.update_counter / ((itr + 1) * sampler_batch),
            eval_seconds=self.eval_seconds, train_seconds=now - self.t_start - self.eval_seconds)
        if eval_trajs is None:
            head, shown = self.TRAIN_HEAD, self.window
            num["window_steps"] = sum(t["Length"] for t in self.window)
        else:
            head, shown = self.EVAL_HEAD, eval_trajs
            num["eval_steps"] = sum(t["Length"] for t in eval_trajs)
            num["eval_trajs"] = len(eval_trajs)
        with logger.tabular_prefix(self.prefix):
            for key, field in head + self.COUNTERS:
                logger.record_tabular(key, num[field])
        if shown:
            for key in (k for k in shown[0] if not k.startswith("_")):
                logger.record_tabular_misc_stat(key, [t[key] for t in shown])
        self.settle()
        for name, bucket in self.opt.items():
            logger.record_tabular_misc_stat(name, bucket)
        logger.dump_tabular(with_prefix=False)
        self.drop_period()
        self.t_mark, self.updates_seen = now, algo.update_counter
        return num["steps_per_s"]


class _Driver:
    """Bring-up, iteration and report shared by the three runner configurations."""

    evaluates = False          # True: offline evaluation at itr 0 and at every logging period
    traj_window = None

    def __init__(self, algo, agent, sampler, n_steps, seed=None, affinity=None,
                 log_interval_steps=1e5):
        self.algo, self.agent, self.sampler = algo, agent, sampler
        self.n_steps, self.log_interval_steps = int(n_steps), int(log_interval_steps)
        self.seed = seed
        self.affinity = dict(affinity or {})
        self.min_itr_learn = getattr(algo, "min_itr_learn", 0)
        self.rank, self.world_size = 0, 1
        self.log = None

    # -- bring-up -----------------------------------------------------------------------------
    def _device(self):
        return self.affinity.get("cuda_idx", None)

    def _select_device(self):
        # kernel wrappers launch on torch's CURRENT device / stream and the replay / sum-tree
        # handles allocate there: the rank's GPU becomes current before anything touches it
        if self._device() is not None and torch.cuda.is_available():
            torch.cuda.set_device(self._device())

    def _bring_up(self):
        self._select_device()
        threads = self.affinity.get("master_torch_threads", None)
        if threads is not None:
            torch.set_num_threads(threads)
        self.seed = make_seed() if self.seed is None else self.seed
        set_seed(self.seed)
        algo, agent, sampler = self.algo, self.agent, self.sampler
        # the sampler first: it forks its env workers, which must happen before this process
        # creates a HIP context (the order of rlpyt/runners/minibatch_rl.py:74-96)
        examples = sampler.initialize(
            agent=agent, affinity=self.affinity, seed=self.seed + 1,
            bootstrap_value=getattr(algo, "bootstrap_value", False),
            traj_info_kwargs=dict(discount=getattr(algo, "discount", 1)),
            rank=self.rank, world_size=self.world_size)
        plan = schedule(self.n_steps, sampler.batch_spec.size * self.world_size,
                        self.log_interval_steps)
        self.n_itr, self.log_interval_itrs, self.itr_batch_size = plan
        logger.log(f"Running {plan.n_itr} iterations of minibatch RL.")
        agent.to_device(self._device())
        if self.world_size > 1:
            agent.data_parallel()
        algo.initialize(agent=agent, n_itr=plan.n_itr, batch_spec=sampler.batch_spec,
                        mid_batch_reset=sampler.mid_batch_reset, examples=examples,
                        world_size=self.world_size, rank=self.rank)
        if self.traj_window is not None:
            logger.log(f"Optimizing over {plan.log_every} iterations.")
        self.log = RunLog(algo.opt_info_fields, self.traj_window)
        return plan

    # -- one iteration / one report -----------------------------------------------------------------
    def _iterate(self, itr):
        logger.set_iteration(itr)
        self.agent.sample_mode(itr)
        samples, traj_infos = self.sampler.obtain_samples(itr)
        self.agent.train_mode(itr)
        self.log.absorb(traj_infos, self.algo.optimize_agent(itr, samples))

    def _evaluate(self, itr):
        if itr and itr < self.min_itr_learn - 1:
            logger.log("Evaluation runs complete.")
            return [], 0.
        logger.log("Evaluating agent...")
        self.agent.eval_mode(itr)
        t0 = time.time()
        trajs = self.sampler.evaluate_agent(itr)
        seconds = time.time() - t0
        logger.log("Evaluation runs complete.")
        if not trajs:
            logger.log("WARNING: had no complete trajectories in eval.")
        return trajs, seconds

    def _snapshot(self, itr):
        """Snapshot in the reference's layout (minibatch_rl.py:136-147): interchangeable files."""
        logger.save_itr_params(itr, dict(
            itr=itr, cum_steps=itr * self.sampler.batch_size * self.world_size,
            agent_state_dict=self.agent.state_dict(),
            optimizer_state_dict=self.algo.optim_state_dict()))

    def _logs(self):
        return True

    def _report(self, itr, eval_trajs=None, eval_seconds=0.):
        if not self._logs():
            self.log.drop_period()
            return
        if itr >= self.min_itr_learn - 1:
            self._snapshot(itr)
        per_itr = self.sampler.batch_size * self.world_size
        self.last_steps_per_second = self.log.emit(
            itr, per_itr * self.log_interval_itrs, (itr + 1) * per_itr, self.algo,
            self.sampler.batch_size, self.world_size, eval_trajs, eval_seconds)

    def train(self):
        plan = self._bring_up()
        if self.evaluates:
            self._report(0, *self._evaluate(0))
        for itr in range(plan.n_itr):
            self._iterate(itr)
            if (itr + 1) % plan.log_every == 0:
                self._report(itr, *(self._evaluate(itr) if self.evaluates else ()))
        logger.log("Training complete.")
        self.sampler.shutdown()

    @property
    def _cum_eval_time(self):
        return self.log.eval_seconds


class MinibatchRl(_Driver):
    """Online tracking: statistics over a window of the training trajectories themselves."""

    def __init__(self, log_traj_window=100, **kwargs):
        super().__init__(**kwargs)
        self.traj_window = int(log_traj_window)


class MinibatchRlEval(_Driver):
    """Offline tracking: at itr 0 and at every logging period the agent is evaluated through
    ``sampler.evaluate_agent`` and the statistics are those of the evaluation trajectories."""

    evaluates = True

    def __init__(self, algo, agent, sampler, n_steps, seed=None, affinity=None,
                 log_interval_steps=1e5):
        super().__init__(algo, agent, sampler, n_steps, seed, affinity, log_interval_steps)


class SyncRl(MinibatchRl):
    """Data-parallel training, one process per GPU (see the module docstring)."""

    def __init__(self, backend=None, init_method=None, **kwargs):
        super().__init__(**kwargs)
        self.backend, self.init_method = backend, init_method

    def _logs(self):
        return self.rank == 0

    def _bring_up(self):
        import torch.distributed as dist
        self._select_device()
        if not dist.is_initialized():
            env = os.environ
            extra = dict(init_method=self.init_method) if self.init_method else {}
            dist.init_process_group(
                backend=self.backend or ("gloo" if self._device() is None else "nccl"),
                rank=int(env.get("RANK", 0)), world_size=int(env.get("WORLD_SIZE", 1)), **extra)
        self.rank, self.world_size = dist.get_rank(), dist.get_world_size()
        if self.seed is None:
            # rank 0 draws, everybody derives (sync_rl.py:52,82): a run is reproducible from
            # rank 0's log
            drawn = [make_seed() if self.rank == 0 else None]
            dist.broadcast_object_list(drawn, src=0)
            self.seed = int(drawn[0])
        self.seed += 100 * self.rank
        if self.rank > 0:           # only rank 0 talks (sync_rl.py:178-179)
            logger.set_quiet(True)
        plan = super()._bring_up()
        dist.barrier()
        self.log.restart_clock()
        return plan


def _sync_entry(rank, world_size, port, backend, build_fn, args):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world_size), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    if torch.cuda.is_available():   # rank r <-> GPU r (all ranks on GPU 0 on a 1-GPU box)
        torch.cuda.set_device(rank % torch.cuda.device_count())
    dist.init_process_group(backend=backend, rank=rank, world_size=world_size,
                            init_method=f"tcp://127.0.0.1:{port}")
    try:
        build_fn(rank, world_size, *args)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def launch_sync(world_size, build_fn, args=(), backend="nccl", port=29511):
    """Spawn ``world_size`` ranks on this node; each calls ``build_fn(rank, world_size,
    *args)``, which builds its own sampler/algo/agent/SyncRl and trains."""
    import torch.multiprocessing as tmp
    tmp.spawn(_sync_entry, args=(world_size, port, backend, build_fn, args), nprocs=world_size,
              join=True)
