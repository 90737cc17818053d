"""MI355X-native rollout sampler: the ``[T, B]`` sample batch lives in HBM.

Role of the reference's GpuSampler + ActionServer + GpuResetCollector /
GpuWaitResetCollector (rlpyt/samplers/parallel/gpu/{sampler,action_server,collectors}.py,
rlpyt/samplers/parallel/{base,worker}.py), re-designed for a 288 GB device:

* environments step on host cores, in forked worker processes (or inline when
  ``n_workers=0``), exactly as in the reference;
* workers write each step's observations into ONE fork-shared, page-locked step buffer;
  the master issues an asynchronous H2D of that step straight into row ``t`` of the
  HBM-resident observation batch, runs the batched action-selection forward on the
  device, samples the actions there, and copies only ``action[B]`` back;
* every other field of the batch (action, reward, done, dist_info, value, bootstrap) is
  written on the device, so ``algo.optimize_agent(samples)`` starts from HBM: the
  reference's 1.09 GB re-upload of the whole batch (rlpyt/algos/pg/ppo.py:72) and its
  per-step D2H of probabilities / values (agents/pg/categorical.py:42) are gone.

Buffer layout contract: ``action`` / ``prev_action`` are the ``[1:]`` /
``[:-1]`` views of one ``[T+1, B]`` array, likewise ``reward`` / ``prev_reward``;
``bootstrap_value`` is ``[1, B]``; ``done`` is bool; ``observation`` keeps the env dtype.
``env_info`` stays a host numpy buffer (only loggers read it).
"""
import ctypes
import multiprocessing as mp
import queue as queue_mod
import time

import numpy as np
import torch

from ..agents.base import AgentInputs
from ..utils import logger
from ..utils.buffer import buffer_from_example, torchify_buffer
from ..utils.collections import AttrDict, namedarraytuple
from ..utils.seed import set_seed
from .base import BaseSampler
from .collections import AgentSamplesBsv, AgentSamples, EnvSamples, Samples

StepBuffer = namedarraytuple("StepBuffer", ["observation", "action", "reward", "done"])


class EnvRunner:
    """Steps a slice of the environments against the shared step buffer (the worker side
    of rlpyt/samplers/parallel/gpu/collectors.py:18-126)."""

    def __init__(self, envs, step_np, env_info_np, TrajInfoCls, mid_batch_reset):
        self.envs = envs
        self.step = step_np            # views restricted to this runner's envs
        self.env_info = env_info_np    # [T, B_w] or None
        self.TrajInfoCls = TrajInfoCls
        self.mid_batch_reset = mid_batch_reset
        self.traj_infos = [TrajInfoCls() for _ in envs]
        self.need_reset = np.zeros(len(envs), dtype=bool)
        self.done_this_batch = np.zeros(len(envs), dtype=bool)

    def start(self, max_decorrelation_steps=0):
        """Reset (and optionally decorrelate with random actions,
        rlpyt/samplers/collectors.py:75-119); leaves obs / prev action / prev reward in
        the step buffer."""
        step = self.step
        for b, env in enumerate(self.envs):
            o = env.reset()
            a, r = env.action_space.null_value(), 0.
            if max_decorrelation_steps:
                n = 1 + int(np.random.rand() * max_decorrelation_steps)
                for _ in range(n):
                    a = env.action_space.sample()
                    o, r, d, info = env.step(a)
                    self.traj_infos[b].step(o, a, r, d, None, info)
                    if getattr(info, "traj_done", d):
                        o = env.reset()
                        self.traj_infos[b] = self.TrajInfoCls()
                    if d:
                        a, r = env.action_space.null_value(), 0.
            step.observation[b] = o
            step.action[b] = a
            step.reward[b] = r
            step.done[b] = False

    def begin_batch(self):
        """Between batches: reset envs that finished under wait-reset mode
        (collectors.py:117-126)."""
        if not self.mid_batch_reset:
            for b in np.where(self.need_reset)[0]:
                self.step.observation[b] = self.envs[b].reset()
                self.step.action[b] = 0
                self.step.reward[b] = 0
                self.step.done[b] = False
            self.need_reset[:] = False

    def step_all(self, t, completed):
        """Apply ``step.action`` to every env; write obs/reward/done for the next step."""
        step = self.step
        for b, env in enumerate(self.envs):
            if self.need_reset[b]:
                # wait-reset: a finished env idles with done=True, zero obs/reward
                # (collectors.py:85-93); the master zeroes its action row.
                step.reward[b] = 0
                step.done[b] = True
                continue
            a = step.action[b]
            o, r, d, info = env.step(a)
            self.traj_infos[b].step(step.observation[b], a, r, d, None, info)
            if getattr(info, "traj_done", d):
                completed.append(self.traj_infos[b].terminate(o))
                self.traj_infos[b] = self.TrajInfoCls()
                if self.mid_batch_reset:
                    o = env.reset()
                else:
                    self.need_reset[b] = True
            if d and not self.mid_batch_reset:
                o = 0 * o if not isinstance(o, tuple) else o
            step.observation[b] = o
            step.reward[b] = r
            step.done[b] = d
            if self.env_info is not None and info:
                self.env_info[t, b] = info


def _worker_loop(rank, runner, ctrl, batch_T, seed, cpus):
    """Forked sampler worker (rlpyt/samplers/parallel/worker.py:37-101)."""
    try:
        if cpus is not None:
            import psutil
            psutil.Process().cpu_affinity(cpus)
    except Exception:
        pass
    torch.set_num_threads(1)
    set_seed(seed)
    runner.start(ctrl.max_decorrelation_steps)
    ctrl.barrier_out.wait()
    while True:
        ctrl.barrier_in.wait()
        if ctrl.quit.value:
            break
        completed = []
        runner.begin_batch()
        ctrl.obs_ready[rank].release()
        for t in range(batch_T):
            ctrl.act_ready[rank].acquire()
            runner.step_all(t, completed)
            ctrl.obs_ready[rank].release()
        for info in completed:
            ctrl.traj_infos_queue.put(dict(info))
        ctrl.barrier_out.wait()


class GpuSampler(BaseSampler):
    """See module docstring.  ``mid_batch_reset=True`` behaves like GpuResetCollector,
    ``False`` like GpuWaitResetCollector."""

    def __init__(self, *args, n_workers=0, mid_batch_reset=True, pin_step_buffer=True,
                 **kwargs):
        super().__init__(*args, **kwargs)
        self.n_workers = int(n_workers)
        self.mid_batch_reset = bool(mid_batch_reset)
        self.pin_step_buffer = pin_step_buffer
        self._pinned_ptrs = []
        self.workers = []

    # ------------------------------------------------------------------------ initialize
    def initialize(self, agent, affinity=None, seed=None, bootstrap_value=False,
                   traj_info_kwargs=None, rank=0, world_size=1):
        T, B = self.batch_spec
        self.agent, self.rank, self.world_size = agent, rank, world_size
        self.seed = seed if seed is not None else 0
        affinity = affinity or dict()
        if traj_info_kwargs:
            for k, v in traj_info_kwargs.items():
                setattr(self.TrajInfoCls, "_" + k, v)
        global_B = B * world_size
        env_ranks = list(range(rank * B, (rank + 1) * B))
        envs = [self.EnvCls(**self.env_kwargs) for _ in range(B)]
        for i, env in enumerate(envs):
            env.seed(self.seed + env_ranks[i])
        agent.initialize(envs[0].spaces, share_memory=False, global_B=global_B,
                         env_ranks=env_ranks)
        # ---- examples (host, before any HIP call so that forking stays safe) ----------
        env0 = envs[0]
        o = env0.reset()
        a = env0.action_space.sample()
        o, r, d, env_info = env0.step(a)
        env0.reset()
        r = np.asarray(r, dtype="float32")
        agent.reset()
        a_t, agent_info = agent.step(*torchify_buffer(AgentInputs(o, np.asarray(a), r)))
        examples = dict(observation=o, reward=r, done=np.asarray(d, dtype=bool),
                        env_info=env_info, action=a_t, agent_info=agent_info)
        self.examples = examples
        # ---- fork-shared host step buffer + host env_info batch ------------------------
        shared = self.n_workers > 0
        self.step_np = StepBuffer(
            observation=buffer_from_example(o, (B,), share_memory=shared),
            action=buffer_from_example(a_t, (B,), share_memory=shared),
            reward=buffer_from_example(r, (B,), share_memory=shared),
            done=buffer_from_example(np.asarray(d, dtype=bool), (B,), share_memory=shared))
        self.env_info_np = (buffer_from_example(env_info, (T, B), share_memory=shared)
                            if env_info else None)
        self._bootstrap = bootstrap_value
        # ---- runners / workers ------------------------------------------------------------
        n_w = max(self.n_workers, 1)
        bounds = np.linspace(0, B, n_w + 1).astype(int)
        self.runners = []
        for w in range(n_w):
            lo, hi = int(bounds[w]), int(bounds[w + 1])
            self.runners.append(EnvRunner(
                envs[lo:hi], self.step_np[lo:hi],
                None if self.env_info_np is None else self.env_info_np[:, lo:hi],
                self.TrajInfoCls, self.mid_batch_reset))
        if self.n_workers > 0:
            self._launch_workers(affinity)
        else:
            set_state = np.random.get_state()
            for rn in self.runners:
                rn.start(self.max_decorrelation_steps)
            np.random.set_state(set_state)
        self._device_ready = False
        logger.log(f"GpuSampler initialized: B={B}, T={T}, workers={self.n_workers}.")
        return AttrDict(examples)

    def _launch_workers(self, affinity):
        ctx = mp.get_context("fork")
        n = self.n_workers
        self.ctrl = AttrDict(
            quit=ctx.RawValue(ctypes.c_bool, False),
            barrier_in=ctx.Barrier(n + 1), barrier_out=ctx.Barrier(n + 1),
            obs_ready=[ctx.Semaphore(0) for _ in range(n)],
            act_ready=[ctx.Semaphore(0) for _ in range(n)],
            traj_infos_queue=ctx.Queue(),
            max_decorrelation_steps=self.max_decorrelation_steps)
        cpus = affinity.get("workers_cpus", None)
        self.workers = []
        for w in range(n):
            wc = None if cpus is None else cpus[w % len(cpus)]
            wc = [wc] if isinstance(wc, int) else wc
            p = ctx.Process(target=_worker_loop, args=(
                w, self.runners[w], self.ctrl, self.batch_spec.T,
                self.seed + 1000 * (self.rank + 1) + w, wc), daemon=True)
            p.start()
            self.workers.append(p)
        self.ctrl.barrier_out.wait()  # decorrelation done, step buffer filled

    # ------------------------------------------------------------- device-side allocation
    def _ensure_device(self):
        """Allocate the HBM batch lazily: after the workers forked and after the runner
        moved the agent to its device (minibatch_rl.py:74-85 order)."""
        if self._device_ready:
            return
        T, B = self.batch_spec
        dev = self.agent.device
        ex = self.examples
        self.device = dev
        all_action = buffer_from_example(ex["action"], (T + 1, B), device=dev)
        all_reward = buffer_from_example(ex["reward"], (T + 1, B), device=dev)
        agent_info = buffer_from_example(ex["agent_info"], (T, B), device=dev)
        observation = buffer_from_example(ex["observation"], (T, B), device=dev)
        done = buffer_from_example(ex["done"], (T, B), device=dev)
        agent_buf = AgentSamples(action=all_action[1:], prev_action=all_action[:-1],
                                 agent_info=agent_info)
        if self._bootstrap:
            bv = buffer_from_example(ex["agent_info"].value, (1, B), device=dev)
            agent_buf = AgentSamplesBsv(*agent_buf, bootstrap_value=bv)
        env_buf = EnvSamples(observation=observation, reward=all_reward[1:],
                             prev_reward=all_reward[:-1], done=done,
                             env_info=self.env_info_np)
        self.samples = Samples(agent=agent_buf, env=env_buf)
        self._all_action, self._all_reward = all_action, all_reward
        self.step_pyt = torchify_buffer(self.step_np)
        self._next_obs_dev = torch.zeros((B,) + tuple(observation.shape[2:]),
                                         dtype=observation.dtype, device=dev)
        # pin the shared step buffer so the per-step copies are true async DMA
        if dev.type == "cuda" and self.pin_step_buffer:
            from .. import _lib
            for arr in self.step_np:
                rc = _lib.lib.rlpyt_host_register(ctypes.c_void_p(arr.ctypes.data),
                                                  int(arr.nbytes))
                if rc == 0:
                    self._pinned_ptrs.append(arr.ctypes.data)
                else:
                    logger.log(f"hipHostRegister failed ({_lib.last_error()}); "
                               "falling back to pageable copies.")
        self._device_ready = True

    # --------------------------------------------------------------------- obtain_samples
    def obtain_samples(self, itr):
        self._ensure_device()
        T, B = self.batch_spec
        agent, dev = self.agent, self.device
        s, step = self.samples, self.step_pyt
        nb = dev.type == "cuda"
        agent.sample_mode(itr)
        completed = []
        if self.n_workers > 0:
            self.ctrl.barrier_in.wait()
            self._wait_obs()
        else:
            for rn in self.runners:
                rn.begin_batch()
        # leading prev_action / prev_reward rows (collectors.py:23-24)
        self._all_action[0].copy_(step.action, non_blocking=nb)
        self._all_reward[0].copy_(step.reward, non_blocking=nb)
        done_prev = None
        for t in range(T):
            s.env.observation[t].copy_(step.observation, non_blocking=nb)
            prev_action, prev_reward = self._all_action[t], self._all_reward[t]
            if done_prev is not None:
                # after a reset the agent sees zero prev action/reward
                # (action_server.py:49-53); the stored rows stay untouched.
                prev_action = torch.where(done_prev, torch.zeros_like(prev_action), prev_action)
                prev_reward = torch.where(done_prev, torch.zeros_like(prev_reward), prev_reward)
            action, agent_info = agent.step(s.env.observation[t], prev_action, prev_reward)
            self._all_action[t + 1].copy_(action)
            s.agent.agent_info[t] = agent_info
            step.action.copy_(action, non_blocking=nb)
            if nb:
                torch.cuda.current_stream().synchronize()
            if self.n_workers > 0:
                for sem in self.ctrl.act_ready:
                    sem.release()
                self._wait_obs()
            else:
                for rn in self.runners:
                    rn.step_all(t, completed)
            # reward / done produced by this env step -> rows t of the HBM batch
            self._all_reward[t + 1].copy_(step.reward, non_blocking=nb)
            s.env.done[t].copy_(step.done, non_blocking=nb)
            done_prev = s.env.done[t] if self.mid_batch_reset else None
            if not self.mid_batch_reset:
                # finished envs record zero action for the rest of the batch
                # (collectors.py:85-93): handled by zeroing their action rows below.
                pass
        if "bootstrap_value" in s.agent:
            self._next_obs_dev.copy_(step.observation, non_blocking=nb)
            prev_action, prev_reward = self._all_action[T], self._all_reward[T]
            if done_prev is not None:
                prev_action = torch.where(done_prev, torch.zeros_like(prev_action), prev_action)
                prev_reward = torch.where(done_prev, torch.zeros_like(prev_reward), prev_reward)
            s.agent.bootstrap_value[0] = agent.value(self._next_obs_dev, prev_action,
                                                     prev_reward)
        if self.n_workers > 0:
            self.ctrl.barrier_out.wait()
            completed = self._drain_traj_infos()
        return self.samples, completed

    def _wait_obs(self):
        for sem in self.ctrl.obs_ready:
            sem.acquire()

    def _drain_traj_infos(self):
        out = []
        q = self.ctrl.traj_infos_queue
        while True:
            try:
                d = q.get(block=True, timeout=0.002 if not out else 0.0005)
            except queue_mod.Empty:
                break
            ti = self.TrajInfoCls()
            ti.update(d)
            out.append(ti)
        return out

    def evaluate_agent(self, itr):
        raise NotImplementedError("offline evaluation is not part of the hot path (SURVEY 8)")

    def shutdown(self):
        if self.n_workers > 0 and self.workers:
            self.ctrl.quit.value = True
            try:
                self.ctrl.barrier_in.wait(timeout=5)
            except Exception:
                pass
            for p in self.workers:
                p.join(timeout=5)
                if p.is_alive():
                    p.terminate()
            self.workers = []
        if self._pinned_ptrs:
            from .. import _lib
            for p in self._pinned_ptrs:
                _lib.lib.rlpyt_host_unregister(ctypes.c_void_p(p))
            self._pinned_ptrs = []
        t0 = time.time()
        logger.log(f"GpuSampler shut down ({time.time() - t0:.2f}s).")
