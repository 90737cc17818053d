"""Host side of the MI355X rollout sampler: what runs on the CPU cores.

The env workers of the reference's GPU sampler (rlpyt/samplers/parallel/worker.py:37-101 and the
collectors of rlpyt/samplers/parallel/gpu/collectors.py:18-161) as this repo runs them:

* ``EnvRunner`` / ``EvalRunner``: step a slice of the environments against the fork-shared,
  page-locked step buffer (training / offline evaluation);
* ``worker_loop``: the forked worker process' main;
* ``StepSync``: the per-group step hand-off between master and workers on two fork-shared 32-bit
  words (futex; ``rlpyt_seq_*`` of the C ABI) instead of the reference's per-worker semaphores.

Nothing here touches the device (as in the reference, samplers/parallel/gpu/sampler.py:103-106).
"""
import ctypes
import os

import numpy as np
import torch

from ..utils.buffer import buffer_leaves
from ..utils.collections import AttrDict
from ..utils.seed import set_seed


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
    # body below (TrajInfo dict updates, numpy scalar stores, attribute probing).  With the stock
    # trajectory statistics that body -- reset and wait-reset collector alike -- runs in C
    # (rlpyt_amd/_envloop, csrc/envloop.c); ``env.step`` stays the Python call it is.  The running
    # statistics live in numpy arrays typed as the reference's per-step updates leave them
    # (np.float32 rewards: float32 sums, float64 discount); a finished trajectory is turned back
    # into a ``TrajInfoCls`` record in ``_native_on_done``.
    def _native_ok(self, first_obs):
        from .collections import AtariTrajInfo, TrajInfo
        if not (self.use_native and self.TrajInfoCls in (TrajInfo, AtariTrajInfo)
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
            f64_mode=int(f64), float32_type=np.float32,
            on_done=self._native_on_done if self.mid_batch_reset else self._native_on_done_wait,
            wait_reset=int(not self.mid_batch_reset),
            force_full=None if self.mid_batch_reset else self.force_full)
        self._completed = None

    def _native_on_done_wait(self, b, final_obs, traj_done, done):
        """Wait-reset collector (collectors.py:92-103): a finished trajectory is recorded and its env
        marked for the reset between batches; a done env's final observation is held for the first row
        of the next batch.  The env is NOT reset here."""
        if traj_done:
            self._native_record(b, final_obs)
            self.need_reset[b] = True
        if done:
            self.temp_observation[b] = final_obs
            self.last_obs[b] = 0

    def _native_on_done(self, b, final_obs):
        """Env ``b`` finished a trajectory: record it (fields typed as the reference's updates leave
        them), restart the statistics, reset the env; returns the first observation."""
        self._native_record(b, final_obs)
        o = self.envs[b].reset()
        self.last_obs[b] = o
        return o

    def _native_record(self, b, final_obs):
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
                if frames:
                    # its blank stack went up whole with the step that finished it; "previous stack
                    # shifted + blank newest frame" is that same blank stack: nothing to upload
                    reset_buf[b] = False
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


def die_with_parent():
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


def worker_loop(rank, runners, ctrl, batch_T, seed, cpus, eval_runner=None):
    """Forked sampler worker (rlpyt/samplers/parallel/worker.py:37-101).  ``runners`` =
    [(group index, EnvRunner)]: this worker's environments, served in group order (with
    dedicated workers per pipeline group there is exactly one entry)."""
    die_with_parent()
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
    seq = StepSync(ctrl.sync_words, ctrl.group_workers, ctrl.n_workers, spin)
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


class StepSync:
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
