"""MI355X-native rollout sampler: the ``[T, B]`` sample batch lives in HBM.

Role of the reference's GpuSampler + ActionServer + GpuResetCollector /
GpuWaitResetCollector (rlpyt/samplers/parallel/gpu/{sampler,action_server,collectors}.py,
rlpyt/samplers/parallel/{base,worker}.py), re-designed for a 288 GB device:

* environments step on host cores, in forked worker processes (or inline when
  ``n_workers=0``), exactly as in the reference (``samplers/workers.py``);
* workers write each step's newest frame / reward / done into a fork-shared, page-locked step
  buffer; the master uploads that block with ONE DMA and replays ONE captured hipGraph per step
  which rebuilds the frame stacks into row ``t`` of the HBM-resident batch, runs the batched
  action-selection forward, samples the actions on the device and writes the action / agent_info
  rows; the sampled actions land directly in the page-locked step buffer (``samplers/device.py``);
* the environments are split into ``n_groups`` pipeline groups (default 2-4 with workers):
  while the device serves group g, the host cores step the environments of the other
  groups.  Groups are column ranges of the same ``[T, B]`` batch -- every column is still one
  environment's contiguous trajectory under one fixed policy, so the batch has exactly the
  reference's semantics (this is not the alternating sampler);
* the T steps of a batch are driven by a C loop once the graphs exist (``samplers/serve.py``);
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
import time

import numpy as np
import torch

from ..agents.base import AgentInputs
from ..utils import logger
from ..utils.buffer import _map, buffer_from_example, np_mp_array, torchify_buffer
from ..utils.collections import AttrDict
from ..utils.misc import usable_cpus
from .base import BaseSampler
from .collections import StepBuffer, StepBufferFs
from .device import DeviceBatch
from .evaluation import Evaluator
from .serve import NativeServe, serve_python
from .workers import EnvRunner, StepSync, die_with_parent, worker_loop  # noqa: F401  (re-exported)

_die_with_parent = die_with_parent      # (name the worker-death test imports)


class GpuSampler(BaseSampler):
    """See module docstring.  ``mid_batch_reset=True`` behaves like GpuResetCollector,
    ``False`` like GpuWaitResetCollector.

    ``n_groups``: pipeline groups (None: 4 for B >= 192 with worker processes, 2 for smaller
    batches when B allows it, else 1).  ``use_graph``: capture the per-step device work in a
    hipGraph (GPU only).  ``frame_dedup`` / ``fused_step`` / ``fused_push`` / ``zero_copy`` /
    ``native_loop``: the stages of the product path, individually switchable so that each has a
    bit-identity test against the path without it (tests/test_sampler_gpu.py)."""

    def __init__(self, *args, n_workers=None, mid_batch_reset=True, pin_step_buffer=True,
                 n_groups=None, use_graph=True, frame_dedup=True, native_loop=True, fused_step=True,
                 fused_push=True, split_workers=False, zero_copy=True, **kwargs):
        super().__init__(*args, **kwargs)
        # n_workers=None: one env worker per entry of affinity["workers_cpus"], the reference's
        # rule (rlpyt/samplers/parallel/base.py:157-172), resolved in initialize(); an explicit
        # count (0 = envs stepped in the master process) overrides the affinity.
        self._n_workers_arg = None if n_workers is None else int(n_workers)
        self._n_groups_arg = n_groups
        self.mid_batch_reset = bool(mid_batch_reset)
        self.native_loop = bool(native_loop)
        self._split_workers = bool(split_workers)
        self._opts = AttrDict(use_graph=bool(use_graph), fused_step=bool(fused_step),
                              fused_push=bool(fused_push), zero_copy=bool(zero_copy),
                              pin_step_buffer=bool(pin_step_buffer), frame_dedup=bool(frame_dedup))
        self._native = None
        self.dev = None
        self._resolve_layout(None)
        self.workers = []
        self.timing = dict(wait_env_s=0., device_issue_s=0., device_wait_s=0., batches=0,
                           pre_s=0., loop_s=0., tail_s=0., post_s=0.)

    def _resolve_layout(self, affinity):
        """Worker-process count and pipeline-group count.  Workers: the ctor's ``n_workers``, else
        ``len(affinity["workers_cpus"])`` capped at B (rlpyt/samplers/parallel/base.py:157-165),
        else 0."""
        B = self.batch_spec.B
        n = self._n_workers_arg
        if n is None:
            cpus = (affinity or {}).get("workers_cpus", None)
            n = 0 if cpus is None else len(cpus)
            if n > B:
                logger.log(f"WARNING: requested fewer envs ({B}) than available worker processes "
                           f"({n}). Using fewer workers.")
                n = B
        self.n_workers = int(n)
        n_groups = self._n_groups_arg
        if n_groups is None:
            # measured at B=256 on the bench host with the two-thread native step loop (~25 env
            # workers): 2 groups 516 K SPS, 3 groups 533-541 K, 4 groups 561 K, 5 groups 555 K,
            # 6 groups 541 K, 8 groups 496 K (with the earlier single-thread loop 3 was best)
            # Groups of ~64 environments: B = 192 (R2D1, round 6, device-bound recurrent step of 13
            # small kernels per group-step): 2 / 3 / 4 groups = 265-268 / 283-284 / 268-269 K SPS
            # interleaved on one box (profiles/r6_ab_r2d1_groups.txt)
            n_groups = 2 if (self.n_workers > 0 and B >= 2 * max(self.n_workers, 1)) else 1
            if n_groups == 2 and B >= 192:
                n_groups = min(4, B // 64)
        self.n_groups = max(1, min(int(n_groups), B))

    def _split_min_workers(self):
        """Fewest env workers for which ``split_workers`` gives every pipeline group its own set."""
        return 2 * self.n_groups

    # ------------------------------------------------------------------------ initialize
    def initialize(self, agent, affinity=None, seed=None, bootstrap_value=False,
                   traj_info_kwargs=None, world_size=1, rank=0, worker_process=None):
        """Signature of the reference's samplers (rlpyt/samplers/parallel/base.py:27-37;
        ``worker_process`` selects an alternative worker main there -- only the default exists
        here)."""
        if worker_process is not None:
            raise NotImplementedError("GpuSampler runs its own env-worker loop (worker_process)")
        T, B = self.batch_spec
        self.agent, self.rank, self.world_size = agent, rank, world_size
        self.seed = seed if seed is not None else 0
        affinity = affinity or dict()
        self._resolve_layout(affinity)
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
        # from a throw-away env instance, as the reference does (samplers/buffer.py:60-80): the
        # B training envs must start from their freshly seeded state
        env0 = self.EnvCls(**self.env_kwargs)
        o = env0.reset()
        a = env0.action_space.sample()
        o, r, d, env_info = env0.step(a)
        del env0
        r = np.asarray(r, dtype="float32")
        agent.reset()
        a_t, agent_info = agent.step(*torchify_buffer(AgentInputs(o, np.asarray(a), r)))
        if "prev_rnn_state" in agent_info:   # drop the B dim of the example (buffer.py:73-75)
            agent_info = agent_info._replace(
                prev_rnn_state=_map(lambda x: x[0], agent_info.prev_rnn_state))
        agent.reset()
        examples = dict(observation=o, reward=r, done=np.asarray(d, dtype=bool),
                        env_info=env_info, action=a_t, agent_info=agent_info)
        self.examples = examples
        # ---- fork-shared host step buffers (one per pipeline group) + host env_info ----
        shared = self.n_workers > 0
        self.env_info_np = (buffer_from_example(env_info, (T, B), share_memory=shared)
                            if env_info else None)
        self._bootstrap = bootstrap_value
        # frame-stacked uint8 image observations (newest frame last): only the newest frame
        # needs to cross PCIe each step
        self._dedup_capable = bool(
            self._opts.frame_dedup and getattr(self.EnvCls, "obs_newest_frame_last", False)
            and isinstance(o, np.ndarray) and o.dtype == np.uint8 and o.ndim == 3
            and (o[0].size % 16 == 0))
        gb = np.linspace(0, B, self.n_groups + 1).astype(int)
        n_w = max(self.n_workers, 1)
        # fork-shared switch "the master uploads newest frames only" (set in _ensure_device)
        self._lazy_obs = mp.get_context("fork").RawValue(ctypes.c_bool, False)
        self.split_workers = bool(self._split_workers and self.n_groups > 1
                                  and self.n_workers >= self._split_min_workers())
        self.groups = []
        runners = [[] for _ in range(n_w)]       # [worker][group]
        for g in range(self.n_groups):
            lo, hi = int(gb[g]), int(gb[g + 1])
            Bg = hi - lo
            # reward f32[Bg], slot i32[Bg], done bool[Bg], reset bool[Bg] share ONE block so
            # they travel in one H2D
            # ... followed by the int64 time index of the step (master-written)
            t_off = 8 * Bg + ((2 * Bg + 15) // 16) * 16
            nbytes = t_off + 16
            # frame-stacked envs: the newest frames and the misc block are ONE contiguous region
            # (frames first), so a group-step's whole upload is a single H2D transfer
            fr_bytes = Bg * int(o[-1].nbytes) if self._dedup_capable else 0
            blk = (np_mp_array(fr_bytes + nbytes, np.uint8) if shared
                   else np.zeros(fr_bytes + nbytes, np.uint8))
            blk[:] = 0
            misc = blk[fr_bytes:]
            fields = dict(
                observation=buffer_from_example(o, (Bg,), share_memory=shared),
                action=buffer_from_example(a_t, (Bg,), share_memory=shared),
                reward=misc[:4 * Bg].view(np.float32),
                done=misc[8 * Bg:9 * Bg].view(np.bool_))
            if self._dedup_capable:
                step_np = StepBufferFs(frame=blk[:fr_bytes].view(o.dtype).reshape((Bg,) + o[-1].shape),
                                       reset=misc[9 * Bg:10 * Bg].view(np.bool_), **fields)
            else:
                step_np = StepBuffer(**fields)
            G = AttrDict(idx=g, lo=lo, hi=hi, Bg=Bg, step_np=step_np, misc_np=misc, blk_np=blk,
                         fr_bytes=fr_bytes, calls=0,
                         graph=None, t_np=misc[t_off:t_off + 8].view(np.int64), t_off=t_off)
            self.groups.append(G)
            # workers of this group: every worker (each then serves all groups in turn; default),
            # or with split_workers a dedicated subset w = g (mod n_groups).  Measured on the
            # bench host (B=256, 2 groups): shared 32 workers 459 K SPS, dedicated 32 -> 432 K,
            # dedicated 64 -> 447 K: the shared layout keeps every core busy in both phases
            ws = ([w for w in range(n_w) if w % self.n_groups == g] if self.split_workers
                  else list(range(n_w)))
            wb = np.linspace(0, Bg, len(ws) + 1).astype(int)
            for k, w in enumerate(ws):
                l, h = int(wb[k]), int(wb[k + 1])
                rn = EnvRunner(
                    envs[lo + l:lo + h], step_np[l:h],
                    None if self.env_info_np is None else self.env_info_np[:, lo + l:lo + h],
                    self.TrajInfoCls, self.mid_batch_reset)
                rn.lazy_obs, rn.batch_T = self._lazy_obs, T
                runners[w].append((g, rn))
            G.n_workers = len(ws)
        self.runners = runners
        self.eval = (Evaluator(self, o, a_t, n_w, shared)
                     if self.eval_n_envs and self.eval_n_envs > 0 else None)
        if self.n_workers > 0:
            self._launch_workers(affinity)
        else:
            set_state = np.random.get_state()
            for _, rn in runners[0]:
                rn.start(self.max_decorrelation_steps)
            np.random.set_state(set_state)
        logger.log(f"GpuSampler initialized: B={B}, T={T}, workers={self.n_workers}, "
                   f"pipeline groups={self.n_groups}.")
        return AttrDict(examples)


    def _launch_workers(self, affinity):
        ctx = mp.get_context("fork")
        n = self.n_workers
        self.ctrl = AttrDict(
            quit=ctx.RawValue(ctypes.c_bool, False),
            barrier_out=ctx.Barrier(n + 1),
            do_eval=ctx.RawValue(ctypes.c_bool, False),
            stop_eval=ctx.RawValue(ctypes.c_bool, False),
            eval_traj_infos_queue=ctx.Queue(),
            # one pair of hand-off words per pipeline group + one for evaluation + batch words
            sync_words=np_mp_array(32 * (len(self.groups) + 2), np.uint32), n_workers=n,
            group_workers=[G.n_workers for G in self.groups] + [n],
            worker_spin=None, cpu_share=usable_cpus() / max(self.world_size, 1),
            traj_infos_queue=ctx.Queue(),
            max_decorrelation_steps=self.max_decorrelation_steps)
        # completed-trajectory statistics come back through a fork-shared float table
        # (one block of rows per worker) instead of a pickling queue
        proto = self.TrajInfoCls()
        self._ti_proto = dict(proto)
        keys = [k for k, v in proto.items() if isinstance(v, (int, float, bool, np.number))]
        if len(keys) != len(proto):
            keys = []        # non-numeric fields: everything goes through the queue
        envs_per_worker = max(sum(len(rn.envs) for _, rn in rs) for rs in self.runners)
        cap = self.batch_spec.T * envs_per_worker + 4 if keys else 0
        self.ctrl.ti_keys = keys
        self.ctrl.ti_table = np_mp_array((n, max(cap, 1), max(len(keys), 1)), np.float64)
        self.ctrl.ti_count = np_mp_array(n, np.int32)
        # per worker: ns waited for actions, ns stepping, group-steps (cumulative; diagnostics)
        self.ctrl.worker_timing = np_mp_array((n, 5), np.float64)
        # CPU pinning as the reference (parallel/base.py:236-237): worker w on workers_cpus[w],
        # unless affinity["set_affinity"] is False
        cpus = affinity.get("workers_cpus", None) if affinity.get("set_affinity", True) else None
        self.workers = []
        import gc
        gc.collect()       # (cyclic garbage of the master is collected HERE, where HIP is valid)
        for w in range(n):
            wc = None if cpus is None else cpus[w % len(cpus)]
            wc = [wc] if isinstance(wc, int) else wc
            p = ctx.Process(target=worker_loop, args=(
                w, self.runners[w], self.ctrl, self.batch_spec.T,
                self.seed + 1000 * (self.rank + 1) + w, wc,
                None if self.eval is None else self.eval.runners[w]), daemon=True)
            p.start()
            self.workers.append(p)
        self.sync = StepSync(self.ctrl.sync_words, self.ctrl.group_workers, n)
        self.ctrl.barrier_out.wait()  # decorrelation done, step buffers filled


    # --------------------------------------------------------------------- obtain_samples
    @property
    def samples(self):
        return None if self.dev is None else self.dev.samples

    @property
    def device(self):
        return None if self.dev is None else self.dev.device

    def _ensure_device(self):
        """Allocate the HBM batch lazily: after the workers forked and after the runner
        moved the agent to its device (minibatch_rl.py:74-85 order)."""
        if self.dev is not None:
            return
        opts = AttrDict(self._opts, dedup_capable=self._dedup_capable,
                        mid_batch_reset=self.mid_batch_reset, bootstrap=self._bootstrap)
        self.dev = DeviceBatch(self.agent, self.examples, self.batch_spec, self.groups,
                               self.env_info_np, opts, self.timing)
        self._lazy_obs.value = bool(all(G.dedup for G in self.groups))

    def _native_serve(self):
        """The C serve loop once it applies (all groups captured), else None."""
        if self._native is not None:
            return self._native
        if not (self.native_loop and self.n_workers > 0 and self.dev.cuda):
            return None
        try:
            self._native = NativeServe.build(self.dev, self.sync, self.world_size)
        except Exception as e:  # noqa: BLE001
            logger.log(f"GpuSampler: native step loop unavailable ({type(e).__name__}: {e}); "
                       "using the Python loop.")
            self.native_loop = False
            self._native = None
        return self._native

    def obtain_samples(self, itr):
        self._ensure_device()
        T, B = self.batch_spec
        dev, tm = self.dev, self.timing
        self.agent.sample_mode(itr)
        completed = []
        par = self.n_workers > 0
        tp0 = time.perf_counter()
        if par:
            self.sync.master_start_batch()
        else:
            for _, rn in self.runners[0]:
                rn.begin_batch()
        dev.begin_batch()
        tp1 = time.perf_counter()
        tail_done = self._serve_batch(dev, par, T, tm, completed)
        tp2 = time.perf_counter()
        if not tail_done:
            self._tail_batch(dev, par)
        dev.synchronize()
        # end of batch: null the prev action / reward the next batch starts from where the
        # env finished (action_server.py:63-68); ``done`` stays set as the carry flag.
        for G in self.groups:
            dn = G.step_np.done
            if np.any(dn):
                _map(lambda x: x.__setitem__(dn, 0), G.step_np.action)
                G.step_np.reward[dn] = 0
        tp3 = time.perf_counter()
        if par:
            self.sync.master_wait_batch_done()
            completed = self._collect_traj_infos()
        tp4 = time.perf_counter()
        tm["batches"] += 1
        tm["pre_s"] += tp1 - tp0
        tm["loop_s"] += tp2 - tp1
        tm["tail_s"] += tp3 - tp2
        tm["post_s"] += tp4 - tp3
        return dev.samples, completed

    def _serve_batch(self, dev, par, T, tm, completed):
        """The T time steps of a batch (``samplers/serve.py``); True when the bootstrap tail ran too."""
        native = self._native_serve()
        if native is not None:
            return native.serve(T, tm)            # (with captured tail graphs: the tail too)
        serve_python(dev, self.sync if par else None, self.runners[0], T, tm, completed)
        return False

    def _tail_batch(self, dev, par):
        """The pass on the observation after the last step (bootstrap value, end-of-batch rows)."""
        for G in self.groups:
            if par:
                self.sync.master_wait_obs(G.idx)
            dev.tail(G)

    def _collect_traj_infos(self):
        """Completed-trajectory statistics of this batch from the shared table (queue for
        the workers that could not use it)."""
        out = []
        keys, table, count = self.ctrl.ti_keys, self.ctrl.ti_table, self.ctrl.ti_count
        proto = self._ti_proto
        # which columns come back as int: those whose prototype value is an int (when integral)
        int_cols = [i for i, k in enumerate(keys) if isinstance(proto[k], int)]
        Cls = self.TrajInfoCls
        # records are rebuilt WITHOUT running the TrajInfo constructors (two levels of __init__ with
        # keyword packing: ~6 us per record, 0.5 ms per [128, 256] batch): a TrajInfo is a dict whose
        # __dict__ is itself (utils/collections.AttrDict), filled here with every key at once
        new, fill = Cls.__new__, dict.__init__
        n_queue = 0
        for w in range(self.n_workers):
            n = int(count[w])
            if n < 0:
                n_queue += -n
                continue
            for row in table[w, :n].tolist():          # one bulk conversion to Python floats
                for i in int_cols:
                    v = row[i]
                    if v.is_integer():
                        row[i] = int(v)
                ti = new(Cls)
                fill(ti, zip(keys, row))
                ti.__dict__ = ti
                out.append(ti)
        q = self.ctrl.traj_infos_queue
        for _ in range(n_queue):
            ti = self.TrajInfoCls()
            ti.update(q.get(block=True, timeout=5))
            out.append(ti)
        return out


    def evaluate_agent(self, itr):
        """Offline evaluation with the agent's current parameters (``samplers/evaluation.py``)."""
        if self.eval is None:
            raise RuntimeError("GpuSampler.evaluate_agent: construct the sampler with "
                               "eval_n_envs > 0 (and eval_max_steps) to evaluate offline.")
        return self.eval.run(itr)

    def shutdown(self):
        if self.n_workers > 0 and self.workers:
            self.ctrl.quit.value = True
            self.sync.master_start_batch()
            for p in self.workers:
                p.join(timeout=5)
                if p.is_alive():
                    p.terminate()
            self.workers = []
        self._native = None
        if self.dev is not None:
            self.dev.release()
        logger.log("GpuSampler shut down.")
