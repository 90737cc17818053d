"""MI355X-native rollout sampler: the ``[T, B]`` sample batch lives in HBM.

Role of the reference's GpuSampler + ActionServer + GpuResetCollector /
GpuWaitResetCollector (rlpyt/samplers/parallel/gpu/{sampler,action_server,collectors}.py,
rlpyt/samplers/parallel/{base,worker}.py), re-designed for a 288 GB device:

* environments step on host cores, in forked worker processes (or inline when
  ``n_workers=0``), exactly as in the reference;
* workers write each step's observations / reward / done into a fork-shared, page-locked
  step buffer; the master issues asynchronous H2D copies of that step into small device
  staging buffers and replays ONE captured hipGraph per step which (i) commits the staged
  observation / reward / done into row ``t`` of the HBM-resident batch (``t`` is a device
  counter, so the same graph serves every time step), (ii) runs the batched
  action-selection forward, (iii) samples the actions on the device and (iv) writes
  action / agent_info rows; only ``action[B]`` travels back to the host;
* the environments are split into ``n_groups`` pipeline groups (default 2-4 with workers):
  while the device serves group g, the host cores step the environments of the other
  group, so per time step the wall time is max(device, host) instead of their sum.
  Groups are column ranges of the same ``[T, B]`` batch -- every column is still one
  environment's contiguous trajectory under one fixed policy, so the batch has exactly the
  reference's semantics (this is not the alternating sampler: one agent, one model call
  per group, same parameters);
* every other field of the batch (action, reward, done, dist_info, value, bootstrap) is
  written on the device, so ``algo.optimize_agent(samples)`` starts from HBM: the
  reference's 1.09 GB re-upload of the whole batch (rlpyt/algos/pg/ppo.py:72) and its
  per-step D2H of probabilities / values (agents/pg/categorical.py:42) are gone.

Buffer layout contract: ``action`` / ``prev_action`` are the ``[1:]`` /
``[:-1]`` views of one ``[T+1, B]`` array, likewise ``reward`` / ``prev_reward``;
``bootstrap_value`` is ``[1, B]``; ``done`` is bool; ``observation`` keeps the env dtype.
``env_info`` stays a host numpy buffer (only loggers read it).
"""
import os
import ctypes
import multiprocessing as mp
from collections import namedtuple
import queue as queue_mod
import time

import numpy as np
import torch

from ..agents.base import AgentInputs
from ..utils import logger
from ..utils.buffer import (_map, buffer_from_example, buffer_leaves, np_mp_array,
                            torchify_buffer)
from ..utils.collections import AttrDict, namedarraytuple
from ..utils.misc import usable_cpus
from ..utils.seed import set_seed
from .base import BaseSampler
from .collections import (AgentSamples, AgentSamplesBsv, EnvSamples, FramePush, Samples,
                          StepBinding, StepBuffer, StepBufferFs)


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
        self.force_full = np.zeros(len(envs), dtype=bool)   # next obs must be uploaded whole
        # wait-reset: the observation returned with done is held back until the next
        # batch starts (collectors.py:65-68,103-104)
        self.temp_observation = None if mid_batch_reset else [None] * len(envs)
        self.frames = "frame" in step_np._fields
        # set by the sampler once it knows that the master uploads only newest frames: the full
        # observation is then written to the step buffer only when the master will read it (last
        # step of a batch, fresh stacks) -- a 33 KB copy less per env step on the host cores,
        # whose time is what bounds the rollout once the device side is fast
        self.lazy_obs = None          # object with a boolean ``.value`` (fork-shared) or None
        self.batch_T = None
        self.last_obs = [None] * len(envs)
        info_leaves = buffer_leaves(env_info_np) if env_info_np is not None else None
        # flat namedtuple env_info (the usual case): one array per field, written directly
        self._info_arrays = info_leaves if (
            info_leaves and not any(isinstance(v, tuple) for v in env_info_np)) else None
        self._native = None            # rlpyt_amd._envloop.EnvLoop once start() has armed it
        # False (or RLPYT_ENVLOOP=0): always the Python loop body (A/B, tests)
        self.use_native = os.environ.get("RLPYT_ENVLOOP", "1") != "0"

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
            self.last_obs[b] = o
        if self.envs and self._native_ok(self.last_obs[0]):
            self._native_begin()

    def begin_batch(self):
        """Between batches under wait-reset: reset finished envs, reinstate held observations,
        clear ``done`` (collectors.py:73-76,117-126).

        Reference behaviour kept on purpose: ``reset_if_needed`` writes the reset observation
        into the step buffer, then the next ``collect_batch`` overwrites it with the held
        terminal observation for EVERY env whose ``done`` flag is set -- so after a finished
        trajectory the first row of the next batch shows the last observation of the old episode
        while the env itself has been reset (the golden batches of the reference's
        GpuWaitResetCollector pin this, tests/test_sampler_parity.py)."""
        if self.mid_batch_reset:
            return
        step = self.step
        for b in np.where(step.done)[0]:
            if self.need_reset[b]:
                self.last_obs[b] = self.envs[b].reset()
                step.observation[b] = self.last_obs[b]
                step.action[b] = 0
                step.reward[b] = 0
                # the next observation does not continue the stack row 0 shows
                self.force_full[b] = True
            if self.temp_observation[b] is not None:
                step.observation[b] = self.temp_observation[b]
        self.need_reset[:] = False
        step.done[:] = False

    # ------------------------------------------------------------------ native loop body
    # With the device side of a time step at ~100 us the rollout is priced in host CPU-seconds per
    # env step under the box's CPU quota, and ~1/3 of them were the interpreter overhead of the loop
    # body below (TrajInfo dict updates, numpy scalar stores, attribute probing).  For the
    # mid-batch-reset collector with the stock trajectory statistics that body runs in C
    # (rlpyt_amd/_envloop, csrc/envloop.c); ``env.step`` stays the Python call it is.  The running
    # statistics live in numpy arrays typed as the reference's per-step updates leave them
    # (np.float32 rewards: float32 sums, float64 discount); a finished trajectory is turned back
    # into a ``TrajInfoCls`` record in ``_native_on_done``.
    def _native_ok(self, first_obs):
        from .collections import AtariTrajInfo, TrajInfo
        if not (self.use_native and self.mid_batch_reset
                and self.TrajInfoCls in (TrajInfo, AtariTrajInfo)
                and (self.env_info is None or self._info_arrays is not None)):
            return False
        step = self.step
        o = np.asarray(first_obs)
        return (step.reward.dtype == np.float32 and step.action.dtype == np.int64
                and step.action.ndim == 1 and step.done.dtype == np.bool_
                and o.dtype == step.observation.dtype and o.flags.c_contiguous
                and o.shape == step.observation.shape[1:]
                and all(a.ndim == 2 and a.dtype in _ENVLOOP_INFO_DTYPES
                        for a in (self._info_arrays or ())))

    def _native_begin(self):
        try:
            from .. import _envloop
        except ImportError:       # extension not built: the Python loop body does the same work
            return
        try:
            self._native_construct(_envloop)
        except (TypeError, ValueError):   # a buffer the C body does not handle: Python loop body
            self._native = None

    def _native_construct(self, _envloop):
        n = len(self.envs)
        ti = self.traj_infos
        st = self._nstats = AttrDict(
            length=np.array([int(x["Length"]) for x in ti], dtype=np.int64),
            nonzero=np.array([int(x["NonzeroRewards"]) for x in ti], dtype=np.int64),
            g=np.array([float(x["_cur_discount"]) for x in ti], dtype=np.float64),
            ret32=np.zeros(n, np.float32), disc32=np.zeros(n, np.float32),
            ret64=np.zeros(n, np.float64), disc64=np.zeros(n, np.float64),
            score=(np.array([float(x["GameScore"]) for x in ti], dtype=np.float64)
                   if "GameScore" in ti[0] else None))
        # sums so far (decorrelation steps of start()): float32 unless some reward was not np.float32
        vals = [x["Return"] for x in ti] + [x["DiscountedReturn"] for x in ti]
        f64 = any(isinstance(v, (float, np.float64)) for v in vals)
        for b, x in enumerate(ti):
            (st.ret64 if f64 else st.ret32)[b] = x["Return"]
            (st.disc64 if f64 else st.disc32)[b] = x["DiscountedReturn"]
        step = self.step
        self._native = _envloop.EnvLoop(
            envs=list(self.envs), action=step.action, reward=step.reward, done=step.done,
            frame=step.frame if self.frames else None, reset=step.reset if self.frames else None,
            observation=step.observation, info_arrays=self._info_arrays, length=st.length,
            ret32=st.ret32, nonzero=st.nonzero, disc32=st.disc32, ret64=st.ret64, disc64=st.disc64,
            cur_discount=st.g, score=st.score, discount=float(self.TrajInfoCls._discount),
            f64_mode=int(f64), on_done=self._native_on_done, float32_type=np.float32)
        self._completed = None

    def _native_on_done(self, b, final_obs):
        """Env ``b`` finished a trajectory: record it (fields typed as the reference's updates leave
        them), restart the statistics, reset the env; returns the first observation."""
        st = self._nstats
        f64 = self._native.f64_mode()
        info = self.TrajInfoCls()
        info["Length"] = int(st.length[b])
        info["Return"] = (st.ret64 if f64 else st.ret32)[b]
        info["NonzeroRewards"] = st.nonzero[b]
        info["DiscountedReturn"] = (st.disc64 if f64 else st.disc32)[b]
        info["_cur_discount"] = float(st.g[b])
        if st.score is not None:
            info["GameScore"] = float(st.score[b])
            st.score[b] = 0.
        self._completed.append(info.terminate(final_obs))
        st.length[b] = st.nonzero[b] = 0
        st.ret32[b] = st.disc32[b] = 0
        st.ret64[b] = st.disc64[b] = 0
        st.g[b] = 1
        o = self.envs[b].reset()
        self.last_obs[b] = o
        return o

    def step_synced(self, seq, g, t, completed):
        """``seq.worker_wait_act(g)`` + ``step_all`` + ``seq.worker_arrive(g)`` as one native call."""
        lazy = (self.frames and self.lazy_obs is not None and self.lazy_obs.value
                and self.batch_T is not None and t != self.batch_T - 1)
        self._completed = completed
        seq.acts[g] += 1
        seq.rounds[g] += 1
        self._native.step_synced(t, bool(lazy), seq.act[g].value, seq.acts[g] & 0xffffffff,
                                 seq.WORKER_SPIN, seq.obs[g].value,
                                 (seq.rounds[g] * seq.group_workers[g]) & 0xffffffff)

    def step_all(self, t, completed):
        """Apply ``step.action`` to every env; write obs/reward/done for the next step."""
        if self._native is not None:
            lazy = (self.frames and self.lazy_obs is not None and self.lazy_obs.value
                    and self.batch_T is not None and t != self.batch_T - 1)
            self._completed = completed
            self._native.step(t, bool(lazy))
            return
        step = self.step
        mbr = self.mid_batch_reset
        obs_buf, act_buf, rew_buf, done_buf = step.observation, step.action, step.reward, step.done
        frames = self.frames
        if frames:
            frame_buf, reset_buf = step.frame, step.reset
        lazy = (frames and self.lazy_obs is not None and self.lazy_obs.value
                and self.batch_T is not None and t != self.batch_T - 1)
        info_arrays, last_obs, traj_infos = self._info_arrays, self.last_obs, self.traj_infos
        for b, env in enumerate(self.envs):
            if not mbr and done_buf[b]:
                # wait-reset: a finished env idles with done=True and blank reward
                # (collectors.py:85-91); the master blanks its action / agent_info rows.
                rew_buf[b] = 0
                continue
            a = act_buf[b]
            o, r, d, info = env.step(a)
            traj_infos[b].step(last_obs[b], a, r, d, None, info)
            fresh = False     # True: the frame stack does not continue the previous one
            if getattr(info, "traj_done", d):
                completed.append(traj_infos[b].terminate(o))
                traj_infos[b] = self.TrajInfoCls()
                if mbr:
                    o = env.reset()
                    fresh = True
                else:
                    self.need_reset[b] = True
            if d and not mbr:
                self.temp_observation[b] = o
                o = 0
                fresh = True
            last_obs[b] = o
            if self.force_full[b]:
                fresh, self.force_full[b] = True, False
            if frames:
                frame_buf[b] = o[-1] if not isinstance(o, int) else 0
                reset_buf[b] = fresh
            if fresh or not lazy:
                obs_buf[b] = o
            rew_buf[b] = r
            done_buf[b] = d
            if info and self.env_info is not None:
                if info_arrays is not None:
                    for arr, v in zip(info_arrays, info):
                        arr[t, b] = v
                else:
                    self.env_info[t, b] = info


# env_info dtypes csrc/envloop.c:kind_of stores directly (anything else: Python loop body)
_ENVLOOP_INFO_DTYPES = tuple(np.dtype(x) for x in ("float32", "float64", "bool", "int32", "int64",
                                                   "uint8"))

EVAL_TRAJ_CHECK = 20    # time steps between checks of the completed-trajectory count


class EvalRunner:
    """Offline-evaluation env stepping against the eval step buffer (the worker side of
    rlpyt/samplers/parallel/gpu/collectors.py:129-161): separate env instances, every finished
    trajectory's info goes to ``sink`` at once, the env restarts immediately."""

    def __init__(self, envs, step_np, TrajInfoCls, max_T):
        self.envs, self.step, self.TrajInfoCls, self.max_T = envs, step_np, TrajInfoCls, max_T
        self.traj_infos = None

    def begin(self):
        step = self.step
        self.traj_infos = [self.TrajInfoCls() for _ in self.envs]
        for b, env in enumerate(self.envs):
            step.observation[b] = env.reset()
            step.action[b] = env.action_space.null_value()
        step.reward[:] = 0
        step.done[:] = False

    def step_all(self, sink):
        step = self.step
        for b, env in enumerate(self.envs):
            a = step.action[b]
            o, r, d, info = env.step(a)
            self.traj_infos[b].step(step.observation[b], a, r, d, None, info)
            if getattr(info, "traj_done", d):
                sink(self.traj_infos[b].terminate(o))
                self.traj_infos[b] = self.TrajInfoCls()
                o = env.reset()
            step.observation[b] = o
            step.reward[b] = r
            step.done[b] = d

    def collect(self, seq, ctrl, g_eval):
        """Worker-side evaluation run: one arrival up front, then one per action set received
        (also for the final "stop" message), so both sides always count the same rounds."""
        q = ctrl.eval_traj_infos_queue
        self.begin()
        seq.worker_arrive(g_eval)
        for _ in range(self.max_T):
            seq.worker_wait_act(g_eval)
            if ctrl.stop_eval.value:
                seq.worker_arrive(g_eval)
                break
            self.step_all(lambda info: q.put(dict(info)))
            seq.worker_arrive(g_eval)
        q.put(None)    # end sentinel of this worker


def _die_with_parent():
    """A worker waits for its next action set without a timeout; if the master is killed (a GPU
    fault aborts the process, an OOM kill) nobody would ever wake it -- under rocprofv3, which waits
    for every child, that hung the whole command.  Linux: have the kernel send SIGTERM to the
    worker when its parent dies."""
    try:
        import signal
        ppid = os.getppid()
        ctypes.CDLL(None, use_errno=True).prctl(1, int(signal.SIGTERM), 0, 0, 0)   # PR_SET_PDEATHSIG
        if os.getppid() != ppid:      # the parent died between fork and prctl
            os._exit(1)
    except Exception:  # noqa: BLE001  (not Linux: keep the reference's behaviour)
        pass


def _worker_loop(rank, runners, ctrl, batch_T, seed, cpus, eval_runner=None):
    """Forked sampler worker (rlpyt/samplers/parallel/worker.py:37-101).  ``runners`` =
    [(group index, EnvRunner)]: this worker's environments, served in group order (with
    dedicated workers per pipeline group there is exactly one entry)."""
    _die_with_parent()
    # Everything inherited from the master at fork time stays out of this process' garbage
    # collector: device tensors caught in reference cycles there would be "freed" here, in a
    # process without a HIP context (seen as a segfault inside gc under the guard-band debug mode).
    import gc
    gc.freeze()
    try:
        if cpus is not None:
            import psutil
            psutil.Process().cpu_affinity(cpus)
    except Exception:
        pass
    torch.set_num_threads(1)
    set_seed(seed)
    for _, rn in runners:
        rn.start(ctrl.max_decorrelation_steps)
    ctrl.barrier_out.wait()
    spin = ctrl.worker_spin
    if spin is None and ctrl.n_workers + 2 <= 1.5 * ctrl.cpu_share:
        # poll ~1 ms for the next action set before sleeping: the hand-off is a few tens of us, a
        # futex wake-up of 20 sleepers costs the poster ~7 us and the last sleeper ~10 us more
        # (profiles/r4_rollout_chain_spin.jsonl: +4..6 % SPS).  Only while this rank's workers and
        # its two serve threads roughly fit its share of the CPU quota: polling processes beyond
        # it only take time from the workers that have envs to step (8 ranks under a 16-CPU quota
        # keep the short poll)
        spin = 30000
    if os.environ.get("RLPYT_WORKER_SPIN"):           # A/B experiments (rollout sweep)
        spin = int(os.environ["RLPYT_WORKER_SPIN"])
    seq = _StepSync(ctrl.sync_words, ctrl.group_workers, ctrl.n_workers, spin)
    ti_keys, ti_table, ti_count = ctrl.ti_keys, ctrl.ti_table, ctrl.ti_count
    while True:
        seq.worker_wait_batch()
        if ctrl.quit.value:
            break
        if ctrl.do_eval.value:      # offline evaluation instead of a training batch
            eval_runner.collect(seq, ctrl, len(ctrl.group_workers) - 1)
            seq.worker_batch_done()
            continue
        completed = []
        for g, rn in runners:
            rn.begin_batch()
            seq.worker_arrive(g)
        for t in range(batch_T):
            for g, rn in runners:
                if rn._native is not None:     # wait -> step -> arrive in one C call
                    rn.step_synced(seq, g, t, completed)
                else:
                    seq.worker_wait_act(g)
                    rn.step_all(t, completed)
                    seq.worker_arrive(g)
        # completed-trajectory statistics -> this worker's rows of the shared table (numeric
        # TrajInfo fields); anything that does not fit goes through the queue instead
        n = len(completed)
        try:
            if n > ti_table.shape[1]:
                raise ValueError
            for i, info in enumerate(completed):
                if len(info) != len(ti_keys):
                    raise ValueError
                ti_table[rank, i] = [float(info[k]) for k in ti_keys]
            ti_count[rank] = n
        except (ValueError, TypeError, KeyError):
            ti_count[rank] = -n
            for info in completed:
                ctrl.traj_infos_queue.put(dict(info))
        # where this worker's batch went: waiting for actions vs stepping envs (native body only)
        wt = ctrl.worker_timing
        for _, rn in runners:
            if rn._native is not None:
                tm_ = rn._native.timing()
                w_ns, s_ns, calls, wake_ns, n_waited = tm_
                wt[rank, 0] += w_ns
                wt[rank, 1] += s_ns
                wt[rank, 2] += calls
                wt[rank, 3] += wake_ns
                wt[rank, 4] += n_waited
        seq.worker_batch_done()


class _StepSync:
    """Per-group step hand-off on two fork-shared 32-bit words (``rlpyt_seq_*`` in the C
    ABI): ``act`` = number of action sets the master has published, ``obs`` = running
    count of worker arrivals.  Both sides keep private copies of the expected values, so
    a hand-off is one atomic + at most one futex syscall instead of the reference's
    per-worker semaphore pair."""

    MASTER_SPIN = 4000     # ~40 us of polling before sleeping (hand-offs are ~100 us apart)
    # workers poll only briefly: letting 64 workers poll through the device phase of every step
    # (so that the master never has to wake them) measured 2-3x SLOWER on the bench host
    WORKER_SPIN = 300

    def __init__(self, words, group_workers, n_workers, worker_spin=None):
        from .. import _lib
        self._lib = _lib.lib
        base = words.ctypes.data
        n_groups = len(group_workers)
        # words[2g] = act sequence, words[2g+1] = arrival counter; 64 B apart per group
        self.act = [ctypes.c_void_p(base + 128 * g) for g in range(n_groups)]
        self.obs = [ctypes.c_void_p(base + 128 * g + 64) for g in range(n_groups)]
        self.n_workers = n_workers              # all workers (batch hand-off)
        self.group_workers = list(group_workers)  # workers serving each group (step hand-off)
        self.acts = [0] * n_groups       # action sets published / consumed so far
        self.rounds = [0] * n_groups     # arrival rounds completed so far
        # batch hand-off (replaces two n+1-party barriers per batch): word 0 of the extra
        # block = batches started, word 16 = workers finished
        if worker_spin is not None:
            self.WORKER_SPIN = int(worker_spin)
        self.batch_word = ctypes.c_void_p(base + 128 * n_groups)
        self.done_word = ctypes.c_void_p(base + 128 * n_groups + 64)
        self.batches = 0

    # -- worker side
    def worker_arrive(self, g):
        self.rounds[g] += 1
        self._lib.rlpyt_seq_arrive(self.obs[g],
                                   (self.rounds[g] * self.group_workers[g]) & 0xffffffff)

    def worker_wait_act(self, g):
        self.acts[g] += 1
        self._lib.rlpyt_seq_wait(self.act[g], self.acts[g] & 0xffffffff, self.WORKER_SPIN, 0)

    def worker_wait_batch(self):
        self.batches += 1
        self._lib.rlpyt_seq_wait(self.batch_word, self.batches & 0xffffffff, 300, 0)

    def worker_batch_done(self):
        self._lib.rlpyt_seq_arrive(self.done_word, (self.batches * self.n_workers) & 0xffffffff)

    # -- master side
    def master_start_batch(self):
        self.batches += 1
        self._lib.rlpyt_seq_post(self.batch_word, self.batches & 0xffffffff)

    def master_wait_batch_done(self, timeout_ms=120000):
        rc = self._lib.rlpyt_seq_wait(self.done_word, (self.batches * self.n_workers) & 0xffffffff,
                                      self.MASTER_SPIN, timeout_ms)
        if rc != 0:
            raise RuntimeError(f"GpuSampler: env workers did not finish the batch (rc={rc}).")

    def master_wait_obs(self, g, timeout_ms=120000):
        self.rounds[g] += 1
        rc = self._lib.rlpyt_seq_wait(self.obs[g],
                                      (self.rounds[g] * self.group_workers[g]) & 0xffffffff,
                                      self.MASTER_SPIN, timeout_ms)
        if rc != 0:
            raise RuntimeError("GpuSampler: env workers did not report within "
                               f"{timeout_ms / 1e3:.0f} s (rc={rc}); a worker process died?")

    def master_post_act(self, g):
        self.acts[g] += 1
        self._lib.rlpyt_seq_post(self.act[g], self.acts[g] & 0xffffffff)


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def _copy_leaves(dst, src, non_blocking=False):
    for d, s in zip(buffer_leaves(dst), buffer_leaves(src)):
        d.copy_(s, non_blocking=non_blocking)


class GpuSampler(BaseSampler):
    """See module docstring.  ``mid_batch_reset=True`` behaves like GpuResetCollector,
    ``False`` like GpuWaitResetCollector.

    ``n_groups``: pipeline groups (None: 4 for B >= 192 with worker processes, 2 for smaller
    batches when B allows it, else 1).  ``use_graph``: capture the per-step device work in a hipGraph (GPU only)."""

    GRAPH_WARMUP_CALLS = 3   # eager calls per group before capture (MIOpen/hipBLASLt find)

    def __init__(self, *args, n_workers=None, mid_batch_reset=True, pin_step_buffer=True,
                 n_groups=None, use_graph=True, frame_dedup=True, native_loop=True, fused_step=True,
                 fused_push=True, split_workers=False, zero_copy=True, zero_copy_frames=False,
                 device_fetch=False, **kwargs):
        super().__init__(*args, **kwargs)
        # n_workers=None: one env worker per entry of affinity["workers_cpus"], the reference's
        # rule (rlpyt/samplers/parallel/base.py:157-172), resolved in initialize(); an explicit
        # count (0 = envs stepped in the master process) overrides the affinity.
        self._n_workers_arg = None if n_workers is None else int(n_workers)
        self._n_groups_arg = n_groups
        self.mid_batch_reset = bool(mid_batch_reset)
        self.pin_step_buffer = pin_step_buffer
        self.use_graph = bool(use_graph)
        self.frame_dedup = bool(frame_dedup)
        self.native_loop = bool(native_loop)
        self.fused_step = bool(fused_step)
        self.fused_push = bool(fused_push)
        self._split_workers = bool(split_workers)
        # zero_copy: the step's head kernel writes the sampled actions straight into the page-locked
        # step buffer the workers read (no D2H copy node / launch); zero_copy_frames: the conv
        # kernel also READS the newest frames in place over PCIe (measured slower: off)
        self.zero_copy = bool(zero_copy)
        self.zero_copy_frames = bool(zero_copy_frames)
        # device_fetch: the step's first kernel pulls the newest frames / scalars / reset stacks out
        # of the page-locked step buffer itself and the time index lives in a device counter, so a
        # step needs NO host call besides its graph launch -- and the native loop enqueues whole
        # batches ahead of time behind stream waits on the workers' arrival counters
        # (rlpyt_sampler_serve_ahead).  False (or RLPYT_DEVICE_FETCH=0): round 3's host-issued
        # uploads + event-driven serve loop.
        self.device_fetch = bool(device_fetch) and os.environ.get("RLPYT_DEVICE_FETCH", "1") != "0"
        self._native = None
        self._resolve_layout(None)
        self._pinned_ptrs = []
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
            n_groups = 2 if (self.n_workers > 0 and B >= 2 * max(self.n_workers, 1)) else 1
            if n_groups == 2 and B >= 192:
                n_groups = 4
        self.n_groups = max(1, min(int(n_groups), B))

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
            self.frame_dedup and getattr(self.EnvCls, "obs_newest_frame_last", False)
            and isinstance(o, np.ndarray) and o.dtype == np.uint8 and o.ndim == 3
            and (o[0].size % 16 == 0))
        gb = np.linspace(0, B, self.n_groups + 1).astype(int)
        n_w = max(self.n_workers, 1)
        # fork-shared switch "the master uploads newest frames only" (set in _ensure_device)
        self._lazy_obs = mp.get_context("fork").RawValue(ctypes.c_bool, False)
        self.split_workers = bool(self._split_workers and self.n_workers >= 2 * self.n_groups
                                  and self.n_groups > 1)
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
        self._init_eval(o, a_t, n_w, shared)
        if self.n_workers > 0:
            self._launch_workers(affinity)
        else:
            set_state = np.random.get_state()
            for _, rn in runners[0]:
                rn.start(self.max_decorrelation_steps)
            np.random.set_state(set_state)
        self._device_ready = False
        logger.log(f"GpuSampler initialized: B={B}, T={T}, workers={self.n_workers}, "
                   f"pipeline groups={self.n_groups}.")
        return AttrDict(examples)

    def _init_eval(self, obs_example, action_example, n_w, shared):
        """Separate evaluation environments and their step buffer
        (rlpyt/samplers/parallel/base.py:88-98, worker.py:68-82): ``eval_n_envs`` is spread
        evenly over the workers (at least one each)."""
        self.eval = None
        if not self.eval_n_envs or self.eval_n_envs <= 0:
            return
        if not self.eval_max_steps:
            raise ValueError("GpuSampler: eval_n_envs > 0 needs eval_max_steps (total env steps "
                             "of one evaluation), as in the reference's samplers.")
        per = max(1, self.eval_n_envs // n_w)
        Be = per * n_w
        if Be != self.eval_n_envs:
            logger.log(f"GpuSampler: using {Be} evaluation environments ({per} per worker).")
        self.eval_n_envs = Be
        self.eval_max_T = max_T = max(1, int(self.eval_max_steps // Be))
        kwargs = self.eval_env_kwargs if self.eval_env_kwargs is not None else self.env_kwargs
        envs = [self.EnvCls(**kwargs) for _ in range(Be)]
        for i, env in enumerate(envs):
            env.seed(self.seed + 50000 + self.rank * Be + i)
        step_np = StepBuffer(
            observation=buffer_from_example(obs_example, (Be,), share_memory=shared),
            action=buffer_from_example(action_example, (Be,), share_memory=shared),
            reward=buffer_from_example(np.asarray(0, dtype=np.float32), (Be,), share_memory=shared),
            done=buffer_from_example(np.asarray(False), (Be,), share_memory=shared))
        runners = [EvalRunner(envs[w * per:(w + 1) * per], step_np[w * per:(w + 1) * per],
                              self.TrajInfoCls, max_T) for w in range(n_w)]
        self.eval = AttrDict(Be=Be, step_np=step_np, runners=runners, device_ready=False)

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
            p = ctx.Process(target=_worker_loop, args=(
                w, self.runners[w], self.ctrl, self.batch_spec.T,
                self.seed + 1000 * (self.rank + 1) + w, wc,
                None if self.eval is None else self.eval.runners[w]), daemon=True)
            p.start()
            self.workers.append(p)
        self.sync = _StepSync(self.ctrl.sync_words, self.ctrl.group_workers, n)
        self.ctrl.barrier_out.wait()  # decorrelation done, step buffers filled

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
        all_done = buffer_from_example(ex["done"], (T + 1, B), device=dev)
        agent_info = buffer_from_example(ex["agent_info"], (T, B), device=dev)
        observation = buffer_from_example(ex["observation"], (T, B), device=dev)
        agent_buf = AgentSamples(action=all_action[1:], prev_action=all_action[:-1],
                                 agent_info=agent_info)
        if self._bootstrap:
            bv = buffer_from_example(ex["agent_info"].value, (1, B), device=dev)
            agent_buf = AgentSamplesBsv(*agent_buf, bootstrap_value=bv)
        # done[t] lives in row t+1 of a [T+1,B] array whose row 0 carries the previous
        # batch's last ``done`` (the reset flag the first step of a batch sees).
        env_buf = EnvSamples(observation=observation, reward=all_reward[1:],
                             prev_reward=all_reward[:-1], done=all_done[1:],
                             env_info=self.env_info_np)
        self.samples = Samples(agent=agent_buf, env=env_buf)
        self._all_action, self._all_reward, self._all_done = all_action, all_reward, all_done
        cuda = dev.type == "cuda"
        for G in self.groups:
            Bg = G.Bg
            G.step_pyt = torchify_buffer(G.step_np)
            G.misc_h = torch.from_numpy(G.misc_np)
            G.obs_stage = buffer_from_example(ex["observation"], (Bg,), device=dev)
            G.blk_h = torch.from_numpy(G.blk_np)
            G.blk_stage = torch.zeros(G.blk_np.size, dtype=torch.uint8, device=dev)
            G.misc_stage = G.blk_stage[G.fr_bytes:]
            G.reward_stage = G.misc_stage[:4 * Bg].view(torch.float32)
            G.done_stage = G.misc_stage[8 * Bg:9 * Bg].view(torch.bool)
            G.dedup = self._dedup_capable and cuda
            if G.dedup:
                G.slot_np = G.misc_np[4 * Bg:8 * Bg].view(np.int32)
                G.slot_stage = G.misc_stage[4 * Bg:8 * Bg].view(torch.int32)
                G.frame_h = torch.from_numpy(G.step_np.frame)
                G.frame_stage = G.blk_stage[:G.fr_bytes].view((Bg,) + tuple(observation.shape[3:]))
                G.full_rows = torch.zeros((Bg,) + tuple(observation.shape[2:]),
                                          dtype=torch.uint8, device=dev)
                G.slot_all = np.arange(Bg, dtype=np.int32)
            G.action_out = buffer_from_example(ex["action"], (Bg,), device=dev)
            # the time index travels with the reward/done block: no counter kernel per step
            G.t_dev = G.misc_stage[G.t_off:G.t_off + 8].view(torch.int64)
            G.pre_commit = G.post_commit = None
            # uniforms for the whole batch are drawn once per batch (one RNG call instead of
            # one per step, and the captured step graph holds no RNG state)
            G.u_all = None
            if cuda and getattr(self.agent, "supports_sample_uniforms", False):
                G.u_all = torch.zeros((T, Bg), dtype=torch.float32, device=dev)
            if cuda:
                from .. import ops
                pre = [(all_reward, G.reward_stage, G.lo, 0), (all_done, G.done_stage, G.lo, 0)]
                if not G.dedup:
                    pre += [(d, x, G.lo, 0) for d, x in zip(buffer_leaves(observation),
                                                            buffer_leaves(G.obs_stage))]
                G.pre_commit = ops.RowCommit(len(pre), dev)
                G.pre_commit.set_entries(pre)
                n_post = 2 * len(buffer_leaves(all_action)) + len(buffer_leaves(agent_info))
                G.post_commit = ops.RowCommit(n_post, dev)
            G.event = torch.cuda.Event() if cuda else None
            # one HIP stream per pipeline group: the H2D of one group overlaps the forward of
            # the other (a single group keeps torch's current stream)
            G.stream = torch.cuda.Stream(device=dev) if (cuda and self.n_groups > 1) else None
            # one RNG stream per group: hipGraphs that replay concurrently must not share the
            # generator's device-side philox offset, or the draws depend on timing
            G.gen = None
            if cuda and self.n_groups > 1:
                G.gen = torch.Generator(device=dev)
                G.gen.manual_seed(int(torch.initial_seed() % (2 ** 31)) + 7919 * (G.idx + 1))
            # pin the shared step buffer so the per-step copies are true async DMA
            if cuda and self.pin_step_buffer:
                from .. import _lib
                arrs = buffer_leaves(G.step_np.observation) + buffer_leaves(G.step_np.action)
                for arr in arrs + [G.blk_np]:      # (frames + misc are one block)
                    rc = _lib.lib.rlpyt_host_register(ctypes.c_void_p(arr.ctypes.data),
                                                      int(arr.nbytes))
                    if rc == 0:
                        self._pinned_ptrs.append(arr.ctypes.data)
                    else:
                        logger.log(f"hipHostRegister failed ({_lib.last_error()}); "
                                   "falling back to pageable copies.")
        # zero-copy hand-off of the actions (default): the head kernel of the step writes them in
        # place in the page-locked step buffer the workers read -- no D2H launch per group-step (it
        # was a 3.8 us blit kernel on the device's serial chain plus one API call on the host's).
        # zero_copy_frames (off): also READ the workers' newest frames in place over PCIe; measured
        # slower in round 3 (the conv kernel then waits ~10 us for 532 KB of PCIe reads while it
        # holds every CU), so the frames + misc block keep travelling as ONE DMA.
        for G in self.groups:
            G.zc_out = G.zc_in = False
            pinned = self._pinned_ptrs
            if (cuda and self.zero_copy and self.pin_step_buffer
                    and isinstance(G.step_np.action, np.ndarray)
                    and G.step_np.action.ctypes.data in pinned):
                try:
                    from .. import _lib
                    G.action_out = _lib.host_mapped_tensor(G.step_np.action, dev)
                    G.zc_out = True
                except Exception as e:  # noqa: BLE001
                    logger.log(f"GpuSampler: zero-copy action hand-off unavailable ({e}); using DMA.")
            if (G.zc_out and self.zero_copy_frames and G.dedup and G.blk_np.ctypes.data in pinned):
                try:
                    from .. import _lib
                    G.frame_stage = _lib.host_mapped_tensor(G.step_np.frame, dev)
                    G.zc_in = True
                except Exception as e:  # noqa: BLE001
                    logger.log(f"GpuSampler: zero-copy frame reads unavailable ({e}); using DMA.")
            G.zc = G.zc_out        # (name kept for the tests / bench line)
        # device-driven stepping (see __init__): host-mapped views of the step buffer for the fetch
        # kernel + the device step counter
        for G in self.groups:
            G.dev_fetch = False
            G.t_ctr = torch.zeros(1, dtype=torch.int64, device=dev) if cuda else None
            pinned = self._pinned_ptrs
            obs_np = G.step_np.observation
            if (cuda and self.device_fetch and G.dedup and G.zc_out and not G.zc_in
                    and isinstance(obs_np, np.ndarray) and obs_np.ctypes.data in pinned
                    and G.blk_np.ctypes.data in pinned and obs_np[0].nbytes % 16 == 0
                    and G.blk_np.ctypes.data % 16 == 0 and obs_np.ctypes.data % 16 == 0):
                try:
                    from .. import _lib
                    G.h_frame = _lib.host_mapped_tensor(G.step_np.frame, dev)
                    G.h_misc = _lib.host_mapped_tensor(G.misc_np, dev)
                    G.h_obs = _lib.host_mapped_tensor(obs_np, dev)
                    G.dev_fetch = True
                except Exception as e:  # noqa: BLE001
                    logger.log(f"GpuSampler: device-side fetch unavailable ({e}); host uploads.")
        self._lazy_obs.value = bool(all(G.dedup for G in self.groups))
        self._device_ready = True

    # ------------------------------------------------------------------ per-step device work
    def _commit_rows(self, dst, src, G, t_idx):
        """``dst[t, lo:hi] = src`` for every leaf, ``t`` being a device index tensor."""
        lo, hi = G.lo, G.hi
        _map(lambda d, s: d[:, lo:hi].index_copy_(0, t_idx, s.unsqueeze(0)), dst, src)

    def _step_body(self, G, capturing=False):
        """Device work of one time step of group ``G`` (graph-capturable: fixed addresses,
        the time index ``G.t_dev`` arrives with the reward/done block of the step).

        Staging holds obs_t and the (reward, done) produced by env step t-1 (at t=0: the
        carry from the previous batch).  Commits obs -> row t, reward -> all_reward[t]
        (= reward[t-1] = prev_reward[t]), done -> all_done[t] (= done[t-1]); runs
        ``agent.step``; writes action -> all_action[t+1] (= action[t]) and agent_info[t].
        On the GPU the row writes are two ``rlpyt_commit_rows`` launches (all leaves at
        once); elsewhere torch ``index_copy_`` does the same thing leaf by leaf."""
        t_next = None
        if G.dev_fetch:
            # device-driven stepping: the step pulls its own inputs out of the page-locked step
            # buffer and reads / advances the device step counter -- no host upload precedes it
            from .. import ops
            ops.rollout_fetch(G.h_frame, G.h_misc, G.h_obs, G.frame_stage, G.misc_stage, G.full_rows,
                              G.t_off, G.t_ctr)
            t_next = G.t_ctr
        if self._step_core(G, capturing, t_next) and t_next is not None:
            # nobody downstream handed t + 1 to the counter (non-fused paths): one tiny launch
            t_next.add_(1)

    def _step_core(self, G, capturing, t_next):
        """The step proper; returns True when the device step counter (``t_next``) still has to be
        advanced by the caller."""
        s, t = self.samples, G.t_dev
        lo, hi = G.lo, G.hi
        if os.environ.get("RLPYT_NULL_STEP") == "1":
            # diagnostics only: no device work in the step (action 0 everywhere) -- what is left
            # of a time step is the host side (env stepping, hand-offs, launches, DMA)
            _map(lambda x: x.zero_(), G.action_out)
            G.post_entries = None
            return True
        fusable = (G.u_all is not None and self.mid_batch_reset and self.fused_step
                   and isinstance(self._all_action, torch.Tensor))
        if (fusable and G.dedup and G.pre_commit is not None and self.fused_push
                and not self.agent.recurrent and not getattr(self.agent, "uses_prev_inputs", True)):
            # frame push + forward + row writes all inside the agent's kernels
            binding = StepBinding(
                action_rows=self._all_action, agent_info_rows=s.agent.agent_info,
                action_out=G.action_out, uniforms=G.u_all, t_dev=t, lo=lo,
                push=FramePush(obs=s.env.observation, new_frame=G.frame_stage,
                               full_rows=G.full_rows, slot=G.slot_stage,
                               scalar_rows=(self._all_reward, G.reward_stage, self._all_done,
                                            G.done_stage)), t_next=t_next)
            if self.agent.step_into(None, None, None, binding):
                G.post_entries = None
                return False      # the head kernel advanced the step counter
        if G.pre_commit is not None:
            if G.dedup:
                # one launch: rebuild the frame stacks of row t + commit the reward/done rows
                from .. import ops
                ops.frame_push(s.env.observation, t, lo, G.frame_stage, G.full_rows,
                               G.slot_stage, stage=G.obs_stage,
                               scalar_rows=(self._all_reward, G.reward_stage, self._all_done,
                                            G.done_stage))
            else:
                G.pre_commit.launch(t)
        else:
            self._commit_rows(s.env.observation, G.obs_stage, G, t)
            self._all_reward[:, lo:hi].index_copy_(0, t, G.reward_stage.unsqueeze(0))
            self._all_done[:, lo:hi].index_copy_(0, t, G.done_stage.unsqueeze(0))
        if getattr(self.agent, "uses_prev_inputs", True):
            prev_action = _map(lambda x: x[:, lo:hi].index_select(0, t).squeeze(0),
                               self._all_action)
            prev_reward = G.reward_stage
            if self.mid_batch_reset:
                # after a reset the agent sees null prev action/reward
                # (action_server.py:49-53); the stored rows stay untouched.
                dn = G.done_stage
                prev_action = _map(lambda x: torch.where(
                    dn.reshape((-1,) + (1,) * (x.dim() - 1)), torch.zeros_like(x), x),
                    prev_action)
                prev_reward = torch.where(dn, torch.zeros_like(prev_reward), prev_reward)
        else:
            prev_action = prev_reward = None
        self.agent.select_envs(lo, hi)
        if self.agent.recurrent:
            # one persistent [N, B_g, H] state per pipeline group; after a reset the env starts
            # from a zero state (action_server.py:49-53)
            self.agent.select_slot(G.idx)
            if self.mid_batch_reset:
                self.agent.reset_where(G.done_stage)
        if fusable:
            # the agent runs the forward AND writes the step's rows (fused head kernel)
            binding = StepBinding(action_rows=self._all_action, agent_info_rows=s.agent.agent_info,
                                  action_out=G.action_out, uniforms=G.u_all, t_dev=t, lo=lo,
                                  push=None)
            if self.agent.step_into(G.obs_stage, prev_action, prev_reward, binding):
                G.post_entries = None
                return True
        self.agent.sample_generator = G.gen
        self.agent.sample_uniforms = None if G.u_all is None else (G.u_all, t)
        action, agent_info = self.agent.step(G.obs_stage, prev_action, prev_reward)
        self.agent.sample_generator = self.agent.sample_uniforms = None
        if not self.mid_batch_reset:
            # wait-reset: finished envs record blank action / agent_info
            # (collectors.py:85-91)
            keep = ~G.done_stage

            def blank(x):
                return x * keep.reshape((-1,) + (1,) * (x.dim() - 1)).to(x.dtype)
            action, agent_info = _map(blank, action), _map(blank, agent_info)
        if capturing:
            # one launch writes action[t], agent_info[t] and the host-bound action copy; the
            # sources live in the graph's private pool, so the table is filled after capture
            a_src = [x.contiguous() for x in buffer_leaves(action)]
            i_src = [x.contiguous() for x in buffer_leaves(agent_info)]
            G.post_entries = (
                [(d, x, lo, 1) for d, x in zip(buffer_leaves(self._all_action), a_src)]
                + [(d, x, lo, 0) for d, x in zip(buffer_leaves(s.agent.agent_info), i_src)]
                + [(d, x, None, 0) for d, x in zip(buffer_leaves(G.action_out), a_src)])
            G.post_commit.launch(t)
        else:
            self._commit_rows(self._all_action, action, G, t + 1)
            self._commit_rows(s.agent.agent_info, agent_info, G, t)
            _copy_leaves(G.action_out, action)
        return True

    def _tail_fused(self, G, cuda):
        """The tail as ONE more step of the group's fused kernels (non-recurrent agents that ignore
        prev inputs, frame-stacked uploads, mid-batch reset): upload the newest frames + misc
        block with t = T, rebuild obs_T into the staging buffer, commit reward / done rows T, and
        run trunk + VALUE head only -> bootstrap_value[0, lo:hi].  Round 3 went through the
        training-time conv kernels, three library GEMMs and a softmax per group here (2.3 ms of
        wall for the 4 groups).  Returns False when it does not apply."""
        T = self.batch_spec.T
        s = self.samples
        agent = self.agent
        if not (cuda and G.dedup and G.u_all is not None and self.mid_batch_reset
                and self.fused_step and self.fused_push
                and "bootstrap_value" in s.agent and not agent.recurrent
                and not getattr(agent, "uses_prev_inputs", True)
                and hasattr(agent, "value_into") and isinstance(self._all_action, torch.Tensor)):
            return False
        bv = s.agent.bootstrap_value
        if not (isinstance(bv, torch.Tensor) and bv.dtype == torch.float32 and bv.is_contiguous()):
            return False
        if G.dev_fetch:     # the device counter stands at T after the batch's T steps
            from .. import ops
            ops.rollout_fetch(G.h_frame, G.h_misc, G.h_obs, G.frame_stage, G.misc_stage, G.full_rows,
                              G.t_off, G.t_ctr)
        else:
            G.t_np[0] = T
            self._upload_special(G, cuda, first=False)
            self._upload_steady(G, cuda)
        binding = StepBinding(
            action_rows=self._all_action, agent_info_rows=s.agent.agent_info,
            action_out=G.action_out, uniforms=G.u_all, t_dev=G.t_dev, lo=G.lo,
            push=FramePush(obs=s.env.observation, new_frame=G.frame_stage,
                           full_rows=G.full_rows, slot=G.slot_stage,
                           scalar_rows=(self._all_reward, G.reward_stage, self._all_done,
                                        G.done_stage)))
        agent.select_envs(G.lo, G.hi)
        return bool(agent.value_into(binding, G.obs_stage, bv[0, G.lo:G.hi]))

    def _tail_body(self, G):
        """After the last env step of the batch: commit reward/done of step T-1 and compute
        the bootstrap value on obs_T (action_server.py:60-62)."""
        T = self.batch_spec.T
        s = self.samples
        lo, hi = G.lo, G.hi
        self._all_reward[T, lo:hi] = G.reward_stage
        self._all_done[T, lo:hi] = G.done_stage
        # THIS group's envs / recurrent state before any agent call: value() of a recurrent
        # agent reads the selected slot's LSTM state (with several pipeline groups the slot
        # still selected is the group stepped last)
        self.agent.select_envs(lo, hi)
        if self.agent.recurrent:
            self.agent.select_slot(G.idx)
        if "bootstrap_value" in s.agent:
            # as the reference: the value call sees the last action / reward as they are -- the
            # null-after-reset of prev inputs happens AFTER it (action_server.py:60-68); for an
            # env that just finished the bootstrap value is masked by (1 - done) anyway
            prev_action = _map(lambda x: x[T, lo:hi], self._all_action)
            prev_reward = G.reward_stage
            s.agent.bootstrap_value[0, lo:hi] = self.agent.value(G.obs_stage, prev_action,
                                                                 prev_reward)
        if self.agent.recurrent:     # end of batch: finished envs restart from a zero state
            self.agent.reset_where(G.done_stage)   # (action_server.py:63-68)

    def _upload_special(self, G, nb, first):
        """Host-dependent part of the upload (frame-stacked envs only): full stacks for the
        first step of a batch and for the few envs whose stack was reset."""
        if not G.dedup or G.dev_fetch:
            return
        if first:
            G.slot_np[:] = G.slot_all
            G.full_rows.copy_(G.step_pyt.observation, non_blocking=nb)
        else:
            G.slot_np[:] = -1
            rs = np.flatnonzero(G.step_np.reset)
            if rs.size:
                G.slot_np[rs] = G.slot_all[:rs.size]
                obs_h = G.step_pyt.observation
                for k, b in enumerate(rs):
                    G.full_rows[k].copy_(obs_h[b], non_blocking=nb)

    def _upload_steady(self, G, nb):
        """Fixed-address part of the upload: newest frames (or whole observations) + the
        reward/slot/done/reset block."""
        if G.dev_fetch:
            return          # the step's fetch kernel reads the page-locked buffer itself
        if G.dedup and not G.zc_in:
            G.blk_stage.copy_(G.blk_h, non_blocking=nb)      # newest frames + misc: one transfer
            return
        if not G.dedup:
            _copy_leaves(G.obs_stage, G.step_pyt.observation, non_blocking=nb)
        G.misc_stage.copy_(G.misc_h, non_blocking=nb)

    def _download(self, G, nb):
        if not G.zc_out:      # zero-copy: the step kernel already wrote the host buffer
            _copy_leaves(G.step_pyt.action, G.action_out, non_blocking=nb)

    def _on_stream(self, G):
        return torch.cuda.stream(G.stream) if G.stream is not None else _NullCtx()

    def _issue(self, G, t, first=False):
        """Enqueue H2D staging -> (graph of) step body -> D2H action on the group's stream."""
        cuda = self.device.type == "cuda"
        t0 = time.perf_counter()
        G.t_np[0] = t
        with self._on_stream(G):
            self._upload_special(G, cuda, first)
            if cuda and self.use_graph and G.graph is None and G.calls >= self.GRAPH_WARMUP_CALLS:
                try:
                    G.graph = self._capture(G)
                except Exception as e:  # noqa: BLE001  (keep sampling: eager step is correct)
                    logger.log(f"GpuSampler: hipGraph capture failed ({type(e).__name__}: {e}); "
                               "continuing with eager per-step launches.")
                    self.use_graph = False
                    G.graph = None
                    torch.cuda.synchronize()
            self._upload_steady(G, cuda)
            if G.graph is not None:
                G.graph.replay()
            else:
                self._step_body(G)
            self._download(G, cuda)
            G.calls += 1
            if cuda:
                G.event.record()
        self.timing["device_issue_s"] += time.perf_counter() - t0

    def _finish(self, G):
        """Block until the group's actions are visible to the host."""
        if G.event is not None:
            t0 = time.perf_counter()
            G.event.synchronize()
            self.timing["device_wait_s"] += time.perf_counter() - t0

    # ------------------------------------------------------------------ native step loop
    def _native_ready(self):
        """All groups captured, RNG-free graphs: the per-step loop can run in C
        (``rlpyt_sampler_serve``)."""
        if getattr(self, "_native", None) is not None:
            return True
        if not all(G.graph is not None and G.u_all is not None for G in self.groups):
            return False
        from .. import _lib
        arr = (_lib.StepGroup * len(self.groups))()
        for G, sg in zip(self.groups, arr):
            sg.act_word, sg.obs_word = self.sync.act[G.idx], self.sync.obs[G.idx]
            sg.n_workers = G.n_workers
            h2d = []
            if G.dedup and not G.zc_in:
                h2d.append((G.blk_stage, G.blk_h))
            else:
                if not G.dedup:
                    h2d += list(zip(buffer_leaves(G.obs_stage),
                                    buffer_leaves(G.step_pyt.observation)))
                h2d.append((G.misc_stage, G.misc_h))
            d2h = ([] if G.zc_out else
                   list(zip(buffer_leaves(G.step_pyt.action), buffer_leaves(G.action_out))))
            if len(h2d) > 8 or len(d2h) > 4:
                self.native_loop = False
                return False
            sg.n_h2d, sg.n_d2h = len(h2d), len(d2h)
            for i, (d, x) in enumerate(h2d):
                sg.h2d[i].dst, sg.h2d[i].src = d.data_ptr(), x.data_ptr()
                sg.h2d[i].nbytes = x.numel() * x.element_size()
            for i, (d, x) in enumerate(d2h):
                sg.d2h[i].dst, sg.d2h[i].src = d.data_ptr(), x.data_ptr()
                sg.d2h[i].nbytes = x.numel() * x.element_size()
            sg.dedup, sg.Bg = int(G.dedup), G.Bg
            if G.dedup:
                sg.reset_flags = G.step_np.reset.ctypes.data
                sg.slot_host = G.slot_np.ctypes.data
                sg.full_rows_dev = G.full_rows.data_ptr()
                sg.obs_host = G.step_np.observation.ctypes.data
                sg.row_bytes = G.step_np.observation[0].nbytes
            sg.t_host = G.t_np.ctypes.data
            sg.graph_exec = G.graph.raw_cuda_graph_exec()
            sg.stream = (G.stream or torch.cuda.current_stream(self.device)).cuda_stream
            G.event.record(G.stream or torch.cuda.current_stream(self.device))
            sg.event = G.event.cuda_event
        self._native = arr
        self._native_timing = (ctypes.c_double * 8)()
        logger.log("GpuSampler: time-step loop handed to rlpyt_sampler_serve (native).")
        return True

    def _ahead_ready(self):
        """Device-driven stepping for every group (fetch kernel first in each captured step graph,
        actions written in place): the native loop can enqueue the whole batch ahead of time
        (``rlpyt_sampler_serve_ahead``).  Builds the group table once."""
        if getattr(self, "_ahead", None) is not None:
            return True
        if getattr(self, "_ahead_failed", False) or os.environ.get("RLPYT_SERVE_AHEAD", "1") == "0":
            return False
        if not all(G.graph is not None and G.u_all is not None and G.dev_fetch and G.zc_out
                   for G in self.groups):
            return False
        from .. import _lib
        words = self.ctrl.sync_words
        if words.ctypes.data not in self._pinned_ptrs:
            rc = _lib.lib.rlpyt_host_register(ctypes.c_void_p(words.ctypes.data), int(words.nbytes))
            if rc != 0:
                logger.log(f"GpuSampler: cannot page-lock the hand-off words ({_lib.last_error()}); "
                           "host-driven step loop.")
                self._ahead_failed = True
                return False
            self._pinned_ptrs.append(words.ctypes.data)
        base_dev = ctypes.c_void_p()
        _lib.check(_lib.lib.rlpyt_host_device_pointer(ctypes.c_void_p(words.ctypes.data),
                                                      ctypes.byref(base_dev)),
                   "rlpyt_host_device_pointer")
        arr = (_lib.AheadGroup * len(self.groups))()
        for G, ag in zip(self.groups, arr):
            ag.act_word, ag.obs_word = self.sync.act[G.idx], self.sync.obs[G.idx]
            ag.act_word_dev = base_dev.value + 128 * G.idx
            ag.obs_word_dev = base_dev.value + 128 * G.idx + 64
            ag.n_workers = G.n_workers
            ag.graph_exec = G.graph.raw_cuda_graph_exec()
            ag.tail_graph_exec = None
            ag.stream = (G.stream or torch.cuda.current_stream(self.device)).cuda_stream
        self._ahead = arr
        self._ahead_timing = (ctypes.c_double * 2)()
        logger.log("GpuSampler: time-step loop enqueued ahead of the env workers "
                   "(rlpyt_sampler_serve_ahead: stream waits on the arrival counters).")
        return True

    def _serve_ahead(self, T):
        from .. import _lib
        arr = self._ahead
        for G, ag in zip(self.groups, arr):
            ag.acts = self.sync.acts[G.idx] & 0xffffffff
            ag.rounds = self.sync.rounds[G.idx] & 0xffffffff
        tmg = self._ahead_timing
        tmg[0] = tmg[1] = 0.
        _lib.check(_lib.lib.rlpyt_sampler_serve_ahead(arr, len(self.groups), T, 120000, tmg),
                   "rlpyt_sampler_serve_ahead")
        for G in self.groups:
            self.sync.acts[G.idx] += T
            self.sync.rounds[G.idx] += T
            G.calls += T
        self.timing["device_issue_s"] += tmg[0]
        self.timing["device_wait_s"] += tmg[1]

    def _native_ready_safe(self):
        try:
            return self._native_ready()
        except Exception as e:  # noqa: BLE001
            logger.log(f"GpuSampler: native step loop unavailable ({type(e).__name__}: {e}); "
                       "using the Python loop.")
            self.native_loop = False
            self._native = None
            return False

    def _ahead_ready_safe(self):
        try:
            return self._ahead_ready()
        except Exception as e:  # noqa: BLE001
            logger.log(f"GpuSampler: enqueue-ahead step loop unavailable ({type(e).__name__}: {e}); "
                       "host-driven loop.")
            self._ahead_failed = True
            self._ahead = None
            return False

    def _serve_spin(self):
        """Idle passes the two serve threads may poll before they start sleeping: polling needs
        two spare cores per rank on top of the env workers; under a tight CPU quota (several
        ranks in one quota-limited container) the threads sleep between hand-offs instead."""
        if getattr(self, "_spin", None) is None:
            import os
            from ..utils.misc import usable_cpus
            per_rank = usable_cpus() / max(self.world_size, 1)
            self._spin = 20000 if per_rank >= 6 else 0
            if os.environ.get("RLPYT_SERVE_SPIN"):          # A/B experiments (rollout sweep)
                self._spin = int(os.environ["RLPYT_SERVE_SPIN"])
        return self._spin

    def _serve_native(self, T):
        from .. import _lib
        arr = self._native
        for G, sg in zip(self.groups, arr):
            sg.acts, sg.rounds = self.sync.acts[G.idx] & 0xffffffff, self.sync.rounds[G.idx] & 0xffffffff
        tmg = self._native_timing
        for i in range(8):
            tmg[i] = 0.
        _lib.check(_lib.lib.rlpyt_sampler_serve(arr, len(self.groups), 0, T, self._serve_spin(),
                                                120000, tmg), "rlpyt_sampler_serve")
        for G in self.groups:
            self.sync.acts[G.idx] += T
            self.sync.rounds[G.idx] += T
            G.calls += T
        self.timing["wait_env_s"] += tmg[0]
        self.timing["device_issue_s"] += tmg[1]
        self.timing["device_wait_s"] += tmg[2]
        for k, i in (("chain_issue_s", 3), ("chain_device_s", 4), ("chain_post_s", 5),
                     ("chain_steps", 6)):
            self.timing[k] = self.timing.get(k, 0.) + tmg[i]

    def _capture(self, G):
        """Capture the device work of one group's step into a hipGraph (torch.cuda.CUDAGraph
        is the HIP graph API on ROCm).  Warm-up calls ran eagerly before, so hipBLASLt has
        picked its kernels and no allocation or search happens under capture.  The H2D / D2H
        copies stay outside: as memcpy nodes they measured slower on ROCm 7.2 (124 vs 92+39 us
        per group-step) and stalled a single-stream capture."""
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        if G.gen is not None and G.u_all is None:
            graph.register_generator_state(G.gen)
        # capture on the group's OWN stream: library workspaces (hipBLASLt split-K buffers)
        # are keyed by stream, and two groups' graphs replay concurrently
        # thread_local: helper threads of this process (e.g. the RCCL watchdog polling its events
        # under DistributedDataParallel) must not invalidate the capture
        with torch.cuda.graph(graph, stream=G.stream, capture_error_mode="thread_local"):
            self._step_body(G, capturing=True)
        if G.post_entries is not None:
            G.post_commit.set_entries(G.post_entries)
        torch.cuda.synchronize()
        logger.log(f"GpuSampler: captured the step graph of pipeline group {G.idx}.")
        return graph

    # --------------------------------------------------------------------- obtain_samples
    def obtain_samples(self, itr):
        self._ensure_device()
        T, B = self.batch_spec
        agent = self.agent
        agent.sample_mode(itr)
        completed = []
        par = self.n_workers > 0
        tm = self.timing
        tp0 = time.perf_counter()
        if par:
            self.sync.master_start_batch()
        else:
            for _, rn in self.runners[0]:
                rn.begin_batch()
        cuda = self.device.type == "cuda"
        for G in self.groups:
            if G.stream is not None:
                G.stream.wait_stream(torch.cuda.current_stream())   # see the updated weights
            with self._on_stream(G):
                if G.t_ctr is not None:
                    G.t_ctr.zero_()
                if G.u_all is not None:
                    G.u_all.uniform_(generator=G.gen)
                # leading prev_action row (collectors.py:23-24); prev_reward[0] and the
                # done carry are committed from the staging block by the first step
                _map(lambda d, s: d[0, G.lo:G.hi].copy_(s, non_blocking=True),
                     self._all_action, G.step_pyt.action)
        tp1 = time.perf_counter()
        if par and cuda and self.native_loop and self._ahead_ready_safe():
            self._serve_ahead(T)
        elif par and cuda and self.native_loop and self._native_ready_safe():
            self._serve_native(T)
        else:
            for t in range(T):
                for G in self.groups:
                    if par:
                        t0 = time.perf_counter()
                        self._wait_obs(G)
                        tm["wait_env_s"] += time.perf_counter() - t0
                    self._issue(G, t, first=(t == 0))
                    if not par:
                        self._finish(G)
                        t0 = time.perf_counter()
                        self.runners[0][G.idx][1].step_all(t, completed)
                        tm["wait_env_s"] += time.perf_counter() - t0
                if par:
                    for G in self.groups:
                        self._finish(G)
                        self.sync.master_post_act(G.idx)
        tp2 = time.perf_counter()
        for G in self.groups:
            if par:
                self._wait_obs(G)
            with self._on_stream(G):
                if getattr(G, "tail_fused", True) and self._tail_fused(G, cuda):
                    continue
                G.tail_fused = False      # decided once per group: the conditions do not change
                _copy_leaves(G.obs_stage, G.step_pyt.observation, non_blocking=cuda)
                G.misc_stage.copy_(G.misc_h, non_blocking=cuda)
                self._tail_body(G)
        if cuda:
            for G in self.groups:
                (G.stream or torch.cuda.current_stream()).synchronize()
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
        return self.samples, completed

    def _wait_obs(self, G):
        self.sync.master_wait_obs(G.idx)

    def _collect_traj_infos(self):
        """Completed-trajectory statistics of this batch from the shared table (queue for
        the workers that could not use it)."""
        out = []
        keys, table, count = self.ctrl.ti_keys, self.ctrl.ti_table, self.ctrl.ti_count
        proto = self._ti_proto
        # which columns come back as int: those whose prototype value is an int (when integral)
        as_int = [isinstance(proto[k], int) for k in keys]
        Cls = self.TrajInfoCls
        n_queue = 0
        for w in range(self.n_workers):
            n = int(count[w])
            if n < 0:
                n_queue += -n
                continue
            for row in table[w, :n].tolist():          # one bulk conversion to Python floats
                ti = Cls()
                dict.update(ti, zip(keys, (int(v) if (i and v.is_integer()) else v
                                           for v, i in zip(row, as_int))))
                out.append(ti)
        q = self.ctrl.traj_infos_queue
        for _ in range(n_queue):
            ti = self.TrajInfoCls()
            ti.update(q.get(block=True, timeout=5))
            out.append(ti)
        return out

    def _ensure_eval_device(self):
        E = self.eval
        if E.device_ready:
            return
        dev = self.agent.device
        E.step_pyt = torchify_buffer(E.step_np)
        E.obs_dev = buffer_from_example(self.examples["observation"], (E.Be,), device=dev)
        E.act_dev = buffer_from_example(self.examples["action"], (E.Be,), device=dev)
        E.rew_dev = torch.zeros(E.Be, dtype=torch.float32, device=dev)
        E.done_dev = torch.zeros(E.Be, dtype=torch.bool, device=dev)
        E.device_ready = True

    def evaluate_agent(self, itr):
        """Offline evaluation with the agent's current parameters (the caller has put the agent
        in eval mode): role of ``ParallelSamplerBase.evaluate_agent`` +
        ``ActionServer.serve_actions_evaluation`` (rlpyt/samplers/parallel/base.py:115-145,
        gpu/action_server.py:76-120).  Separate env instances are stepped by the same worker
        processes; the batched forward runs on the device, observations go up and actions come
        down once per step (eager launches -- evaluation is outside the timed training path).
        Stops after ``eval_max_steps`` env steps or, if given, once ``eval_max_trajectories``
        have completed (checked every EVAL_TRAJ_CHECK steps).  Returns the completed TrajInfos."""
        if self.eval is None:
            raise RuntimeError("GpuSampler.evaluate_agent: construct the sampler with "
                               "eval_n_envs > 0 (and eval_max_steps) to evaluate offline.")
        self._ensure_eval_device()
        E, agent = self.eval, self.agent
        par = self.n_workers > 0
        cuda = agent.device.type == "cuda"
        step_np, step_pyt = E.step_np, E.step_pyt
        traj_infos = []

        def take(info):
            ti = self.TrajInfoCls()
            ti.update(info)
            traj_infos.append(ti)

        def drain(block_for_sentinels=0):
            q, n_sent = self.ctrl.eval_traj_infos_queue, 0
            while True:
                try:
                    item = q.get(block=block_for_sentinels > 0, timeout=20)
                except queue_mod.Empty:
                    if block_for_sentinels > 0:
                        raise RuntimeError("GpuSampler.evaluate_agent: an env worker did not "
                                           "finish its evaluation run.")
                    return
                if item is None:
                    n_sent += 1
                    if n_sent >= block_for_sentinels > 0:
                        return
                else:
                    take(item)

        agent.reset()
        agent.select_envs(None, None)
        if agent.recurrent:
            agent.select_slot("eval")
        g_eval = len(self.groups)
        if par:
            self.ctrl.stop_eval.value = False
            self.ctrl.do_eval.value = True
            self.sync.master_start_batch()
        else:
            E.runners[0].begin()
        stop = False
        for t in range(self.eval_max_T):
            if par:
                if t % EVAL_TRAJ_CHECK == 0:
                    drain()
                self.sync.master_wait_obs(g_eval)
            dn = step_np.done
            if np.any(dn):      # null prev action / reward after a reset (action_server.py:95-98)
                _map(lambda x: x.__setitem__(dn, 0), step_np.action)
                step_np.reward[dn] = 0
            _copy_leaves(E.obs_dev, step_pyt.observation, non_blocking=False)
            _copy_leaves(E.act_dev, step_pyt.action)
            E.rew_dev.copy_(step_pyt.reward)
            if agent.recurrent:
                E.done_dev.copy_(step_pyt.done)
                agent.reset_where(E.done_dev)
            action, _agent_info = agent.step(E.obs_dev, E.act_dev, E.rew_dev)
            _copy_leaves(step_pyt.action, action)     # D2H (synchronous)
            if self.eval_max_trajectories is not None and t % EVAL_TRAJ_CHECK == 0:
                stop = len(traj_infos) >= self.eval_max_trajectories
            if par:
                self.ctrl.stop_eval.value = stop
                self.sync.master_post_act(g_eval)
            elif not stop:
                E.runners[0].step_all(take)
            if stop:
                logger.log(f"Evaluation reached max num trajectories "
                           f"({self.eval_max_trajectories}).")
                break
        if not stop and self.eval_max_trajectories is not None:
            logger.log(f"Evaluation reached max num time steps ({self.eval_max_T}).")
        if par:
            self.sync.master_wait_obs(g_eval)     # the workers' closing arrival
            self.sync.master_wait_batch_done()
            drain(block_for_sentinels=self.n_workers)
            self.ctrl.do_eval.value = False
        if cuda:
            torch.cuda.current_stream().synchronize()
        return traj_infos

    def shutdown(self):
        if self.n_workers > 0 and self.workers:
            self.ctrl.quit.value = True
            self.sync.master_start_batch()
            for p in self.workers:
                p.join(timeout=5)
                if p.is_alive():
                    p.terminate()
            self.workers = []
        for G in getattr(self, "groups", []):
            G.graph = None
        if self._pinned_ptrs:
            from .. import _lib
            for p in self._pinned_ptrs:
                _lib.lib.rlpyt_host_unregister(ctypes.c_void_p(p))
            self._pinned_ptrs = []
        logger.log("GpuSampler shut down.")
