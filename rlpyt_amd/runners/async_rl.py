"""Asynchronous sampling / optimisation on one GPU (rlpyt/runners/async_rl.py:20-140,524-610), as
THREADS of one process over device-resident buffers.

The reference forks a sampler process, two memory-copier processes and optimizer workers around OS
shared memory: sample double buffer -> copier -> replay ring, parameters back through a shared model.
With the sampler's batch, the replay ring, the sum tree and all model copies in HBM there is nothing to
share between address spaces -- what remains of the design is its ORDER and its throttle:

* sampler thread (``run_async_sampler``): ``recv_shared_memory`` -> ``obtain_samples`` ->
  ``replay.append_samples(algo.samples_to_buffer(batch))`` -> ``ctrl.sampler_itr = itr``.  The append
  is the memory copier's statement (async_rl.py:604-608) issued by the sampler thread itself: a
  device-to-device launch on its stream, under the replay's write lock (``replays/async_.py``); the
  batch is free for the next rollout once those launches are behind an event.  The C serve loop and
  every kernel launch release the GIL, so the optimizer thread runs meanwhile;
* optimizer (the calling thread, ``train``): waits until the sampler is ``throttle_itr`` batches in,
  runs ``algo.optimize_agent(itr, sampler_itr=...)`` (batches drawn under the read lock), publishes the
  parameters (``agent.send_shared_memory``), advances ``throttle_itr`` by ``delta_throttle_itr`` --
  the reference's replay-ratio bound, same formulas (async_rl.py:187-192);
* the sampler steps a TWIN of the agent (``BaseAgent.async_twin``: own parameters on the same device,
  refreshed between batches), so an update never changes the policy in the middle of a batch.

Constructor and ``train()`` follow the reference class; ``affinity`` may be the reference's structure
(``.sampler`` / ``.optimizer`` attributes) or one plain dict used for both sides.  Logged rows: the
reference's async diagnostics (async_rl.py:363-398).
"""
import threading
import time
from collections import deque

import torch

from ..utils import logger
from ..utils.collections import AttrDict
from ..utils.seed import make_seed, set_seed

THROTTLE_WAIT = 0.005      # (the reference sleeps 50 ms between polls of another PROCESS's counter)


def run_async_sampler(sampler, agent, algo, replay, ctrl, n_itr, device):
    """Sampler-thread body: async_rl.py:524-549 with the memory copier's append inlined."""
    try:
        if device is not None and device.type == "cuda":
            torch.cuda.set_device(device)
        for itr in range(n_itr):
            if ctrl.quit.is_set():
                break
            agent.recv_shared_memory()
            agent.sample_mode(itr)
            samples, traj_infos = sampler.obtain_samples(itr)
            replay.append_samples(algo.samples_to_buffer(samples))
            if device is not None and device.type == "cuda":
                # the rollout that follows overwrites the batch on the pipeline groups' streams:
                # the append's reads (this thread's stream) must be done first
                torch.cuda.current_stream(device).synchronize()
            with ctrl.lock:
                ctrl.traj_infos.extend(traj_infos)
                ctrl.sampler_itr = itr
        logger.log(f"Async sampler reached final itr: {ctrl.sampler_itr + 1}, quitting.")
    except BaseException as e:  # noqa: BLE001  (surface in the optimizer thread)
        ctrl.error = e
    finally:
        ctrl.quit.set()


class AsyncRl:
    """Asynchronous RL with online agent performance tracking (async_rl.py:401-434)."""

    def __init__(self, algo, agent, sampler, n_steps, affinity=None, seed=None,
                 log_interval_steps=1e5, log_traj_window=100):
        self.algo, self.agent, self.sampler = algo, agent, sampler
        self.n_steps, self.log_interval_steps = int(n_steps), int(log_interval_steps)
        self.affinity, self.seed = affinity, seed
        self.log_traj_window = int(log_traj_window)

    # ---- bring-up ------------------------------------------------------------------------------
    def _sides(self):
        aff = self.affinity or {}
        samp = getattr(aff, "sampler", None) or (aff.get("sampler") if isinstance(aff, dict) else None)
        opt = getattr(aff, "optimizer", None) or (aff.get("optimizer") if isinstance(aff, dict) else None)
        if isinstance(opt, (list, tuple)):
            assert len(opt) == 1, "one optimizer GPU per process (multi-GPU: SyncRl over RCCL)"
            opt = opt[0]
        if samp is None and opt is None:
            samp = opt = dict(aff)
        return dict(samp or {}), dict(opt or {})

    def startup(self):
        self.seed = make_seed() if self.seed is None else self.seed
        set_seed(self.seed)
        samp_aff, opt_aff = self._sides()
        cuda_idx = opt_aff.get("cuda_idx", samp_aff.get("cuda_idx", None))
        if cuda_idx is not None and torch.cuda.is_available():
            torch.cuda.set_device(cuda_idx)
        algo, agent, sampler = self.algo, self.agent, self.sampler
        examples = sampler.initialize(
            agent=agent, affinity=samp_aff, seed=self.seed + 1,
            bootstrap_value=getattr(algo, "bootstrap_value", False),
            traj_info_kwargs=dict(discount=getattr(algo, "discount", 1)))
        self.sampler_batch_size = sampler.batch_spec.size
        log_itrs = max(self.log_interval_steps // self.sampler_batch_size, 1)
        n_itr = -(-self.n_steps // self.log_interval_steps) * log_itrs          # async_rl.py:200-206
        self.log_interval_itrs, self.n_itr = log_itrs, n_itr
        logger.log(f"Running {n_itr} sampler iterations.")
        agent.to_device(cuda_idx)
        replay = algo.async_initialize(
            agent=agent, sampler_n_itr=n_itr, batch_spec=sampler.batch_spec,
            mid_batch_reset=sampler.mid_batch_reset, examples=examples, world_size=1)
        assert getattr(replay, "async_", False), "algo.async_initialize must return an async replay buffer"
        algo.optim_initialize(rank=0)
        self.twin = agent.async_twin()
        sampler.agent = self.twin                    # the sampler steps the twin from here on
        self.ctrl = AttrDict(quit=threading.Event(), lock=threading.Lock(), sampler_itr=-1,
                             traj_infos=[], error=None)
        self.sampler_thread = threading.Thread(
            target=run_async_sampler, name="async-sampler", daemon=True,
            args=(sampler, self.twin, algo, replay, self.ctrl, n_itr, agent.device))
        throttle_itr = 1 + getattr(algo, "min_steps_learn", 0) // self.sampler_batch_size
        delta_throttle_itr = (algo.batch_size * algo.updates_per_optimize /
                              (self.sampler_batch_size * algo.replay_ratio))
        self._opt_infos = {k: [] for k in algo.opt_info_fields}
        self._traj_infos = deque(maxlen=self.log_traj_window)
        self._cum_completed_trajs = self._new_completed_trajs = 0
        self._start_time = self._last_time = time.time()
        self._last_itr = self._last_sampler_itr = self._last_update_counter = 0
        self.sampler_thread.start()
        return throttle_itr, delta_throttle_itr

    # ---- the optimizer loop (async_rl.py:84-140) ---------------------------------------------------
    def train(self):
        throttle_itr, delta_throttle_itr = self.startup()
        ctrl = self.ctrl
        throttle_time, itr, log_counter = 0., 0, 0
        try:
            while True:
                logger.set_iteration(itr)
                while ctrl.sampler_itr + 1 < throttle_itr and not ctrl.quit.is_set():
                    time.sleep(THROTTLE_WAIT)
                    throttle_time += THROTTLE_WAIT
                if ctrl.quit.is_set():
                    break
                throttle_itr += delta_throttle_itr
                self.agent.train_mode(itr)
                opt_info = self.algo.optimize_agent(itr, sampler_itr=ctrl.sampler_itr)
                self.agent.send_shared_memory()          # to the sampler twin
                sampler_itr = ctrl.sampler_itr
                self.store_diagnostics(itr, sampler_itr, self._drain(), opt_info)
                if (sampler_itr + 1) // self.log_interval_itrs > log_counter:
                    self.log_diagnostics(itr, sampler_itr, throttle_time)
                    log_counter += 1
                    throttle_time = 0.
                itr += 1
        finally:
            ctrl.quit.set()
            self.sampler_thread.join(timeout=60)
        if ctrl.error is not None:
            raise ctrl.error
        self.store_diagnostics(itr, ctrl.sampler_itr, self._drain(), ())
        self.log_diagnostics(itr, ctrl.sampler_itr, throttle_time)
        self.optimizer_itrs = itr
        logger.log("Master optimizer shutting down; training complete.")
        self.sampler.shutdown()

    def _drain(self):
        with self.ctrl.lock:
            out, self.ctrl.traj_infos = self.ctrl.traj_infos, []
        return out

    # ---- diagnostics (async_rl.py:335-398,417-434) ---------------------------------------------------
    def store_diagnostics(self, itr, sampler_itr, traj_infos, opt_info):
        self._cum_completed_trajs += len(traj_infos)
        self._new_completed_trajs += len(traj_infos)
        self._traj_infos.extend(traj_infos)
        for k, v in self._opt_infos.items():
            new_v = getattr(opt_info, k, [])
            v.extend(new_v if isinstance(new_v, list) else [new_v])

    def log_diagnostics(self, itr, sampler_itr, throttle_time, prefix="Diagnostics/"):
        new_time = time.time()
        time_elapsed = max(new_time - self._last_time, 1e-9)
        algo = self.algo
        new_updates = algo.update_counter - self._last_update_counter
        new_samples = self.sampler_batch_size * (sampler_itr - self._last_sampler_itr)
        cum_steps = sampler_itr * self.sampler_batch_size          # (the reference's count, :356)
        nan = float("nan")
        with logger.tabular_prefix(prefix):
            logger.record_tabular("CumCompletedTrajs", self._cum_completed_trajs)
            logger.record_tabular("NewCompletedTrajs", self._new_completed_trajs)
            logger.record_tabular("StepsInTrajWindow", sum(t["Length"] for t in self._traj_infos))
            logger.record_tabular("Iteration", itr)
            logger.record_tabular("SamplerIteration", sampler_itr)
            logger.record_tabular("CumTime (s)", new_time - self._start_time)
            logger.record_tabular("CumSteps", cum_steps)
            logger.record_tabular("CumUpdates", algo.update_counter)
            logger.record_tabular("ReplayRatio", new_updates * algo.batch_size / max(1, new_samples))
            logger.record_tabular("CumReplayRatio", algo.update_counter * algo.batch_size / max(1, cum_steps))
            logger.record_tabular("StepsPerSecond", nan if itr == 0 else new_samples / time_elapsed)
            logger.record_tabular("UpdatesPerSecond", nan if itr == 0 else new_updates / time_elapsed)
            logger.record_tabular("OptThrottle", (time_elapsed - throttle_time) / time_elapsed)
        if self._traj_infos:
            for k in (k for k in self._traj_infos[0] if not k.startswith("_")):
                logger.record_tabular_misc_stat(k, [t[k] for t in self._traj_infos])
        for k, v in self._opt_infos.items():
            logger.record_tabular_misc_stat(k, v)
        self._opt_infos = {k: [] for k in self._opt_infos}
        logger.dump_tabular(with_prefix=False)
        self._last_time, self._last_itr = new_time, itr
        self._last_sampler_itr, self._last_update_counter = sampler_itr, algo.update_counter
        self._new_completed_trajs = 0
