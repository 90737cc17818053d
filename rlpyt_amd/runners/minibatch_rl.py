"""Training-loop runners (protocol of rlpyt/runners/minibatch_rl.py:16-286 and
rlpyt/runners/sync_rl.py:11-193).

``MinibatchRl``: one process, one MI355X.  ``SyncRl``: one process PER GPU -- every rank runs
the same sampler->algo loop on its own ``[T, B]`` batch (weak scaling: global batch =
world_size x T x B, sync_rl.py:40-45); the only coupling is DistributedDataParallel's
gradient all-reduce, which on ROCm's "nccl" backend is RCCL over xGMI.  Unlike the
reference, ranks are not forked from a master that already holds a HIP context: each rank
is its own process launched by ``torch.distributed.run`` (or ``launch_sync`` below, which
spawns), reads RANK / WORLD_SIZE / LOCAL_RANK, and logs only on rank 0.
"""
import os
import time
from collections import deque

import torch

from ..utils import logger
from ..utils.quick_args import save__init__args
from ..utils.seed import make_seed, set_seed


class MinibatchRlBase:
    _eval = False

    def __init__(self, algo, agent, sampler, n_steps, seed=None, affinity=None,
                 log_interval_steps=1e5):
        n_steps = int(n_steps)
        log_interval_steps = int(log_interval_steps)
        affinity = dict() if affinity is None else affinity
        save__init__args(locals())
        self.min_itr_learn = getattr(self.algo, "min_itr_learn", 0)
        self.rank = 0
        self.world_size = 1

    def startup(self):
        # Every kernel wrapper launches on torch's CURRENT device / stream and the replay /
        # sum-tree handles allocate on it: make the rank's GPU current before anything touches
        # the device (sampler buffers, agent.to_device, workspaces).
        cuda_idx = self.affinity.get("cuda_idx", None)
        if cuda_idx is not None and torch.cuda.is_available():
            torch.cuda.set_device(cuda_idx)
        torch_threads = self.affinity.get("master_torch_threads", None)
        if torch_threads is not None:
            torch.set_num_threads(torch_threads)
        if self.seed is None:
            self.seed = make_seed()
        set_seed(self.seed)
        rank, world_size = self.rank, self.world_size
        examples = self.sampler.initialize(
            agent=self.agent, affinity=self.affinity, seed=self.seed + 1,
            bootstrap_value=getattr(self.algo, "bootstrap_value", False),
            traj_info_kwargs=self.get_traj_info_kwargs(), rank=rank, world_size=world_size)
        self.itr_batch_size = self.sampler.batch_spec.size * world_size
        n_itr = self.get_n_itr()
        self.agent.to_device(self.affinity.get("cuda_idx", None))
        if world_size > 1:
            self.agent.data_parallel()
        self.algo.initialize(agent=self.agent, n_itr=n_itr,
                             batch_spec=self.sampler.batch_spec,
                             mid_batch_reset=self.sampler.mid_batch_reset, examples=examples,
                             world_size=world_size, rank=rank)
        self.initialize_logging()
        return n_itr

    def get_traj_info_kwargs(self):
        return dict(discount=getattr(self.algo, "discount", 1))

    def get_n_itr(self):
        log_interval_itrs = max(self.log_interval_steps // self.itr_batch_size, 1)
        n_itr = self.n_steps // self.itr_batch_size
        if n_itr % log_interval_itrs > 0:
            n_itr += log_interval_itrs
            n_itr -= n_itr % log_interval_itrs
        self.log_interval_itrs = log_interval_itrs
        self.n_itr = max(n_itr, 1)
        logger.log(f"Running {self.n_itr} iterations of minibatch RL.")
        return self.n_itr

    def initialize_logging(self):
        self._opt_infos = {k: list() for k in self.algo.opt_info_fields}
        self._start_time = self._last_time = time.time()
        self._cum_time = 0.
        self._cum_completed_trajs = 0
        self._last_update_counter = 0

    def shutdown(self):
        logger.log("Training complete.")
        self.sampler.shutdown()

    def get_itr_snapshot(self, itr):
        return dict(itr=itr, cum_steps=itr * self.sampler.batch_size * self.world_size,
                    agent_state_dict=self.agent.state_dict(),
                    optimizer_state_dict=self.algo.optim_state_dict())

    def save_itr_snapshot(self, itr):
        logger.save_itr_params(itr, self.get_itr_snapshot(itr))

    def store_diagnostics(self, itr, traj_infos, opt_info):
        self._cum_completed_trajs += len(traj_infos)
        for k, v in self._opt_infos.items():
            new_v = getattr(opt_info, k, [])
            v.extend(new_v if isinstance(new_v, list) else [new_v])

    def log_diagnostics(self, itr, traj_infos=None, eval_time=0, prefix="Diagnostics/"):
        if itr >= self.min_itr_learn - 1:      # minibatch_rl.py:168-169
            self.save_itr_snapshot(itr)
        new_time = time.time()
        self._cum_time = new_time - self._start_time
        train_time_elapsed = new_time - self._last_time - eval_time
        new_updates = self.algo.update_counter - self._last_update_counter
        new_samples = self.sampler.batch_size * self.world_size * self.log_interval_itrs
        updates_per_second = (float("nan") if itr == 0 else new_updates / train_time_elapsed)
        samples_per_second = (float("nan") if itr == 0 else new_samples / train_time_elapsed)
        replay_ratio = (new_updates * self.algo.batch_size * self.world_size / new_samples)
        cum_replay_ratio = (self.algo.batch_size * self.algo.update_counter /
                            ((itr + 1) * self.sampler.batch_size))
        cum_steps = (itr + 1) * self.sampler.batch_size * self.world_size
        self.last_steps_per_second = samples_per_second
        with logger.tabular_prefix(prefix):
            if self._eval:
                logger.record_tabular("CumTrainTime", self._cum_time - getattr(self, "_cum_eval_time", 0))
            logger.record_tabular("Iteration", itr)
            logger.record_tabular("CumTime (s)", self._cum_time)
            logger.record_tabular("CumSteps", cum_steps)
            logger.record_tabular("CumCompletedTrajs", self._cum_completed_trajs)
            logger.record_tabular("CumUpdates", self.algo.update_counter)
            logger.record_tabular("StepsPerSecond", samples_per_second)
            logger.record_tabular("UpdatesPerSecond", updates_per_second)
            logger.record_tabular("ReplayRatio", replay_ratio)
            logger.record_tabular("CumReplayRatio", cum_replay_ratio)
        self._log_infos(traj_infos)
        logger.dump_tabular(with_prefix=False)
        self._last_time = new_time
        self._last_update_counter = self.algo.update_counter

    def _log_infos(self, traj_infos=None):
        if traj_infos is None:
            traj_infos = self._traj_infos
        if traj_infos:
            for k in traj_infos[0]:
                if not k.startswith("_"):
                    logger.record_tabular_misc_stat(k, [info[k] for info in traj_infos])
        if self._opt_infos:
            for k, v in self._opt_infos.items():
                logger.record_tabular_misc_stat(k, v)
        self._opt_infos = {k: list() for k in self._opt_infos}


class MinibatchRl(MinibatchRlBase):
    """Online tracking of training trajectories (minibatch_rl.py:232-283)."""

    def __init__(self, log_traj_window=100, **kwargs):
        super().__init__(**kwargs)
        self.log_traj_window = int(log_traj_window)

    def train(self):
        n_itr = self.startup()
        for itr in range(n_itr):
            logger.set_iteration(itr)
            self.agent.sample_mode(itr)
            samples, traj_infos = self.sampler.obtain_samples(itr)
            self.agent.train_mode(itr)
            opt_info = self.algo.optimize_agent(itr, samples)
            self.store_diagnostics(itr, traj_infos, opt_info)
            if (itr + 1) % self.log_interval_itrs == 0:
                self.log_diagnostics(itr)
        self.shutdown()

    def initialize_logging(self):
        self._traj_infos = deque(maxlen=self.log_traj_window)
        self._new_completed_trajs = 0
        logger.log(f"Optimizing over {self.log_interval_itrs} iterations.")
        super().initialize_logging()

    def store_diagnostics(self, itr, traj_infos, opt_info):
        self._new_completed_trajs += len(traj_infos)
        self._traj_infos.extend(traj_infos)
        super().store_diagnostics(itr, traj_infos, opt_info)

    def log_diagnostics(self, itr, prefix="Diagnostics/"):
        with logger.tabular_prefix(prefix):
            logger.record_tabular("NewCompletedTrajs", self._new_completed_trajs)
            logger.record_tabular("StepsInTrajWindow",
                                  sum(info["Length"] for info in self._traj_infos))
        super().log_diagnostics(itr, prefix=prefix)
        self._new_completed_trajs = 0


class MinibatchRlEval(MinibatchRlBase):
    """Offline tracking: pauses at every log interval to run evaluation trajectories through
    ``sampler.evaluate_agent`` (minibatch_rl.py:286-357)."""

    _eval = True

    def train(self):
        n_itr = self.startup()
        eval_traj_infos, eval_time = self.evaluate_agent(0)
        self.log_diagnostics(0, eval_traj_infos, eval_time)
        for itr in range(n_itr):
            logger.set_iteration(itr)
            self.agent.sample_mode(itr)
            samples, traj_infos = self.sampler.obtain_samples(itr)
            self.agent.train_mode(itr)
            opt_info = self.algo.optimize_agent(itr, samples)
            self.store_diagnostics(itr, traj_infos, opt_info)
            if (itr + 1) % self.log_interval_itrs == 0:
                eval_traj_infos, eval_time = self.evaluate_agent(itr)
                self.log_diagnostics(itr, eval_traj_infos, eval_time)
        self.shutdown()

    def evaluate_agent(self, itr):
        if itr >= self.min_itr_learn - 1 or itr == 0:
            logger.log("Evaluating agent...")
            self.agent.eval_mode(itr)
            eval_time = -time.time()
            traj_infos = self.sampler.evaluate_agent(itr)
            eval_time += time.time()
        else:
            traj_infos, eval_time = [], 0.0
        logger.log("Evaluation runs complete.")
        return traj_infos, eval_time

    def initialize_logging(self):
        super().initialize_logging()
        self._cum_eval_time = 0

    def log_diagnostics(self, itr, eval_traj_infos, eval_time, prefix="Diagnostics/"):
        if not eval_traj_infos:
            logger.log("WARNING: had no complete trajectories in eval.")
        steps_in_eval = sum(info["Length"] for info in eval_traj_infos)
        with logger.tabular_prefix(prefix):
            logger.record_tabular("StepsInEval", steps_in_eval)
            logger.record_tabular("TrajsInEval", len(eval_traj_infos))
            self._cum_eval_time += eval_time
            logger.record_tabular("CumEvalTime", self._cum_eval_time)
        super().log_diagnostics(itr, eval_traj_infos, eval_time, prefix=prefix)


class SyncRl(MinibatchRl):
    """Data-parallel training, one process per GPU (sync_rl.py:11-193 semantics)."""

    def __init__(self, backend=None, init_method=None, **kwargs):
        super().__init__(**kwargs)
        self.backend = backend
        self.init_method = init_method

    def startup(self):
        import torch.distributed as dist
        if not dist.is_initialized():
            rank = int(os.environ.get("RANK", 0))
            world_size = int(os.environ.get("WORLD_SIZE", 1))
            backend = self.backend or (
                "gloo" if self.affinity.get("cuda_idx", None) is None else "nccl")
            if self.affinity.get("cuda_idx", None) is not None:
                torch.cuda.set_device(self.affinity["cuda_idx"])
            kw = dict(init_method=self.init_method) if self.init_method else {}
            dist.init_process_group(backend=backend, rank=rank, world_size=world_size, **kw)
        self.rank, self.world_size = dist.get_rank(), dist.get_world_size()
        if self.affinity.get("cuda_idx", None) is not None and torch.cuda.is_available():
            torch.cuda.set_device(self.affinity["cuda_idx"])
        if self.seed is None:
            # the master draws the seed, the workers derive theirs from it (sync_rl.py:52,82):
            # rank 0's draw is broadcast so that a run is reproducible from rank 0's log
            box = [make_seed() if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            self.seed = int(box[0])
        self.seed = self.seed + 100 * self.rank  # sync_rl.py:82
        if self.rank > 0:
            logger.set_quiet(True)  # workers do no logging (sync_rl.py:178-179)
        n_itr = super().startup()
        dist.barrier()
        self._start_time = self._last_time = time.time()
        return n_itr

    def log_diagnostics(self, itr, prefix="Diagnostics/"):
        if self.rank == 0:
            super().log_diagnostics(itr, prefix=prefix)
        else:
            self._opt_infos = {k: list() for k in self._opt_infos}
            self._new_completed_trajs = 0

    def save_itr_snapshot(self, itr):
        if self.rank == 0:
            super().save_itr_snapshot(itr)


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
