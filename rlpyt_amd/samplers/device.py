"""Device side of the MI355X rollout sampler: the ``[T, B]`` batch in HBM and one time step's work.

What replaces the body of ``ActionServer.serve_actions`` (rlpyt/samplers/parallel/gpu/
action_server.py:44-58) and the collectors' row writes (collectors.py:30-47) for one pipeline
group and one time step:

  upload (ONE DMA of the group's page-locked block: newest frames + reward / slot / done / reset /
  t; full stacks only for reset envs) -> captured hipGraph of the step body {frame-stack rebuild +
  conv stack, trunk, heads + softmax + draw + the step's row writes} -> the head kernel leaves the
  actions in the page-locked step buffer the workers read (zero copy).

``DeviceBatch`` owns the HBM batch (``samples``), the per-group staging buffers / streams / graphs
and the step / tail bodies; the sampler (``samplers/gpu.py``) owns the host side and the order of
calls, ``samplers/serve.py`` drives the T steps of a batch.
"""
import ctypes
import os
import time

import numpy as np
import torch

from ..utils import logger
from ..utils.buffer import _map, buffer_from_example, buffer_leaves, torchify_buffer
from .tail import TailPass, copy_leaves
from .collections import (AgentSamples, AgentSamplesBsv, EnvSamples, FramePush, Samples,
                          StepBinding)


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class DeviceBatch(TailPass):
    """HBM-resident sample batch + per-group step machinery.

    ``groups``: the sampler's pipeline-group records (host fields ``lo / hi / Bg / step_np /
    misc_np / blk_np / fr_bytes / t_np / t_off``); this object adds the device fields.
    ``opts``: the sampler's switches (``use_graph, fused_step, fused_push, zero_copy,
    pin_step_buffer, dedup_capable, mid_batch_reset, bootstrap``)."""

    GRAPH_WARMUP_CALLS = 3   # eager calls per group before capture (MIOpen / hipBLASLt find)

    def __init__(self, agent, examples, batch_spec, groups, env_info_np, opts, timing):
        self.agent, self.examples, self.batch_spec = agent, examples, batch_spec
        self.groups, self.opts, self.timing = groups, opts, timing
        self.device = dev = agent.device
        self.cuda = dev.type == "cuda"
        self.pinned_ptrs = []
        self.use_graph = bool(opts.use_graph)
        T, B = batch_spec
        ex = examples
        all_action = buffer_from_example(ex["action"], (T + 1, B), device=dev)
        all_reward = buffer_from_example(ex["reward"], (T + 1, B), device=dev)
        all_done = buffer_from_example(ex["done"], (T + 1, B), device=dev)
        agent_info = buffer_from_example(ex["agent_info"], (T, B), device=dev)
        observation = buffer_from_example(ex["observation"], (T, B), device=dev)
        agent_buf = AgentSamples(action=all_action[1:], prev_action=all_action[:-1],
                                 agent_info=agent_info)
        if opts.bootstrap:
            bv = buffer_from_example(ex["agent_info"].value, (1, B), device=dev)
            agent_buf = AgentSamplesBsv(*agent_buf, bootstrap_value=bv)
        # done[t] lives in row t+1 of a [T+1,B] array whose row 0 carries the previous
        # batch's last ``done`` (the reset flag the first step of a batch sees).
        env_buf = EnvSamples(observation=observation, reward=all_reward[1:],
                             prev_reward=all_reward[:-1], done=all_done[1:],
                             env_info=env_info_np)
        self.samples = Samples(agent=agent_buf, env=env_buf)
        self.all_action, self.all_reward, self.all_done = all_action, all_reward, all_done
        for G in groups:
            self._allocate_group(G, observation, agent_info)
        for G in groups:
            self._map_action_buffer(G)

    # ------------------------------------------------------------------ allocation
    def _allocate_group(self, G, observation, agent_info):
        T, B = self.batch_spec
        dev, cuda, ex, opts = self.device, self.cuda, self.examples, self.opts
        Bg = G.Bg
        G.step_pyt = torchify_buffer(G.step_np)
        G.misc_h = torch.from_numpy(G.misc_np)
        G.obs_stage = buffer_from_example(ex["observation"], (Bg,), device=dev)
        G.blk_h = torch.from_numpy(G.blk_np)
        G.blk_stage = torch.zeros(G.blk_np.size, dtype=torch.uint8, device=dev)
        G.misc_stage = G.blk_stage[G.fr_bytes:]
        G.reward_stage = G.misc_stage[:4 * Bg].view(torch.float32)
        G.done_stage = G.misc_stage[8 * Bg:9 * Bg].view(torch.bool)
        G.dedup = opts.dedup_capable and cuda
        if G.dedup:
            G.slot_np = G.misc_np[4 * Bg:8 * Bg].view(np.int32)
            G.slot_stage = G.misc_stage[4 * Bg:8 * Bg].view(torch.int32)
            G.frame_stage = G.blk_stage[:G.fr_bytes].view((Bg,) + tuple(observation.shape[3:]))
            G.full_rows = torch.zeros((Bg,) + tuple(observation.shape[2:]),
                                      dtype=torch.uint8, device=dev)
            G.slot_all = np.arange(Bg, dtype=np.int32)
        G.action_out = buffer_from_example(ex["action"], (Bg,), device=dev)
        # the time index travels with the reward/done block: no counter kernel per step
        G.t_dev = G.misc_stage[G.t_off:G.t_off + 8].view(torch.int64)
        G.pre_commit = G.post_commit = G.post_entries = None
        # uniforms for the whole batch are drawn once per batch (one RNG call instead of one per
        # step, and the captured step graph holds no RNG state)
        G.u_all = None
        if cuda and getattr(self.agent, "supports_sample_uniforms", False):
            G.u_all = torch.zeros((T, Bg), dtype=torch.float32, device=dev)
        if cuda:
            from .. import ops
            pre = [(self.all_reward, G.reward_stage, G.lo, 0), (self.all_done, G.done_stage, G.lo, 0)]
            if not G.dedup:
                pre += [(d, x, G.lo, 0) for d, x in zip(buffer_leaves(observation),
                                                        buffer_leaves(G.obs_stage))]
            G.pre_commit = ops.RowCommit(len(pre), dev)
            G.pre_commit.set_entries(pre)
            n_post = 2 * len(buffer_leaves(self.all_action)) + len(buffer_leaves(agent_info))
            G.post_commit = ops.RowCommit(n_post, dev)
        G.event = torch.cuda.Event() if cuda else None
        # one HIP stream per pipeline group: the H2D of one group overlaps the forward of the
        # other (a single group keeps torch's current stream)
        G.stream = torch.cuda.Stream(device=dev) if (cuda and len(self.groups) > 1) else None
        # one RNG stream per group: hipGraphs that replay concurrently must not share the
        # generator's device-side philox offset, or the draws depend on timing
        G.gen = None
        if cuda and len(self.groups) > 1:
            G.gen = torch.Generator(device=dev)
            G.gen.manual_seed(int(torch.initial_seed() % (2 ** 31)) + 7919 * (G.idx + 1))
        # pin the shared step buffer so the per-step copies are true async DMA
        if cuda and opts.pin_step_buffer:
            from .. import _lib
            arrs = buffer_leaves(G.step_np.observation) + buffer_leaves(G.step_np.action)
            for arr in arrs + [G.blk_np]:      # (frames + misc are one block)
                rc = _lib.lib.rlpyt_host_register(ctypes.c_void_p(arr.ctypes.data),
                                                  int(arr.nbytes))
                if rc == 0:
                    self.pinned_ptrs.append(arr.ctypes.data)
                else:
                    logger.log(f"hipHostRegister failed ({_lib.last_error()}); "
                               "falling back to pageable copies.")

    def _map_action_buffer(self, G):
        """Zero-copy hand-off of the actions (default): the head kernel of the step writes them in
        place in the page-locked step buffer the workers read -- no D2H launch per group-step (it
        was a 3.8 us blit kernel on the device's serial chain plus one API call on the host's)."""
        G.zc = False
        if (self.cuda and self.opts.zero_copy and self.opts.pin_step_buffer
                and isinstance(G.step_np.action, np.ndarray)
                and G.step_np.action.ctypes.data in self.pinned_ptrs):
            try:
                from .. import _lib
                G.action_out = _lib.host_mapped_tensor(G.step_np.action, self.device)
                G.zc = True
            except Exception as e:  # noqa: BLE001
                logger.log(f"GpuSampler: zero-copy action hand-off unavailable ({e}); using DMA.")

    def release(self):
        for G in self.groups:
            G.graph = None
            G.tail_graph = None
        if self.pinned_ptrs:
            from .. import _lib
            for p in self.pinned_ptrs:
                _lib.lib.rlpyt_host_unregister(ctypes.c_void_p(p))
            self.pinned_ptrs = []

    # ------------------------------------------------------------------ per-step device work
    def _commit_rows(self, dst, src, G, t_idx):
        """``dst[t, lo:hi] = src`` for every leaf, ``t`` being a device index tensor."""
        lo, hi = G.lo, G.hi
        _map(lambda d, s: d[:, lo:hi].index_copy_(0, t_idx, s.unsqueeze(0)), dst, src)

    def _push_binding(self, G):
        s = self.samples
        return StepBinding(
            action_rows=self.all_action, agent_info_rows=s.agent.agent_info,
            action_out=G.action_out, uniforms=G.u_all, t_dev=G.t_dev, lo=G.lo,
            push=FramePush(obs=s.env.observation, new_frame=G.frame_stage,
                           full_rows=G.full_rows, slot=G.slot_stage,
                           scalar_rows=(self.all_reward, G.reward_stage, self.all_done,
                                        G.done_stage)))

    def step_body(self, G, capturing=False):
        """Device work of one time step of group ``G`` (graph-capturable: fixed addresses,
        the time index ``G.t_dev`` arrives with the reward/done block of the step).

        Staging holds obs_t and the (reward, done) produced by env step t-1 (at t=0: the
        carry from the previous batch).  Commits obs -> row t, reward -> all_reward[t]
        (= reward[t-1] = prev_reward[t]), done -> all_done[t] (= done[t-1]); runs
        ``agent.step``; writes action -> all_action[t+1] (= action[t]) and agent_info[t].
        On the GPU the row writes are two ``rlpyt_commit_rows`` launches (all leaves at
        once); elsewhere torch ``index_copy_`` does the same thing leaf by leaf."""
        s, t, agent, opts = self.samples, G.t_dev, self.agent, self.opts
        lo, hi = G.lo, G.hi
        if os.environ.get("RLPYT_NULL_STEP") == "1":
            # diagnostics only: no device work in the step (action 0 everywhere) -- what is left
            # of a time step is the host side (env stepping, hand-offs, launches, DMA)
            _map(lambda x: x.zero_(), G.action_out)
            G.post_entries = None
            return
        uses_prev = getattr(agent, "uses_prev_inputs", True)
        fusable = (G.u_all is not None and opts.mid_batch_reset and opts.fused_step
                   and isinstance(self.all_action, torch.Tensor))
        if (fusable and G.dedup and G.pre_commit is not None and opts.fused_push
                and not agent.recurrent and not uses_prev):
            # frame push + forward + row writes all inside the agent's kernels
            if agent.step_into(None, None, None, self._push_binding(G)):
                G.post_entries = None
                return
        if G.pre_commit is not None:
            if G.dedup:
                # one launch: rebuild the frame stacks of row t + commit the reward/done rows
                from .. import ops
                ops.frame_push(s.env.observation, t, lo, G.frame_stage, G.full_rows,
                               G.slot_stage, stage=G.obs_stage,
                               scalar_rows=(self.all_reward, G.reward_stage, self.all_done,
                                            G.done_stage))
            else:
                G.pre_commit.launch(t)
        else:
            self._commit_rows(s.env.observation, G.obs_stage, G, t)
            self.all_reward[:, lo:hi].index_copy_(0, t, G.reward_stage.unsqueeze(0))
            self.all_done[:, lo:hi].index_copy_(0, t, G.done_stage.unsqueeze(0))
        if uses_prev:
            prev_action = _map(lambda x: x[:, lo:hi].index_select(0, t).squeeze(0),
                               self.all_action)
            prev_reward = G.reward_stage
        else:
            prev_action = prev_reward = None
        agent.select_envs(lo, hi)
        stepped = None
        if agent.recurrent:
            # one persistent [N, B_g, H] state per pipeline group
            agent.select_slot(G.idx)
            if uses_prev and hasattr(agent, "step_with_reset"):
                # agents that fold the reset handling into their step's kernels take the inputs as
                # stored plus the mask of the environments reset before this step
                agent.sample_generator = G.gen
                agent.sample_uniforms = None if G.u_all is None else (G.u_all, t)
                stepped = agent.step_with_reset(G.obs_stage, prev_action, prev_reward,
                                                G.done_stage if opts.mid_batch_reset else None)
                agent.sample_generator = agent.sample_uniforms = None
        if stepped is None and opts.mid_batch_reset:
            if uses_prev:
                # after a reset the agent sees null prev action/reward
                # (action_server.py:49-53); the stored rows stay untouched.
                dn = G.done_stage
                prev_action = _map(lambda x: torch.where(
                    dn.reshape((-1,) + (1,) * (x.dim() - 1)), torch.zeros_like(x), x),
                    prev_action)
                prev_reward = torch.where(dn, torch.zeros_like(prev_reward), prev_reward)
            if agent.recurrent:
                # ... and starts from a zero state (action_server.py:49-53)
                agent.reset_where(G.done_stage)
        if fusable and stepped is None:
            # the agent runs the forward AND writes the step's rows (fused head kernel)
            binding = StepBinding(action_rows=self.all_action, agent_info_rows=s.agent.agent_info,
                                  action_out=G.action_out, uniforms=G.u_all, t_dev=t, lo=lo,
                                  push=None)
            if agent.step_into(G.obs_stage, prev_action, prev_reward, binding):
                G.post_entries = None
                return
        if stepped is not None:
            action, agent_info = stepped
        else:
            agent.sample_generator = G.gen
            agent.sample_uniforms = None if G.u_all is None else (G.u_all, t)
            action, agent_info = agent.step(G.obs_stage, prev_action, prev_reward)
            agent.sample_generator = agent.sample_uniforms = None
        zw = None
        if not opts.mid_batch_reset and capturing and G.done_stage.element_size() == 1:
            # wait-reset: finished envs record blank action / agent_info (collectors.py:85-91) -- the
            # row-commit launch below writes zeros for them (no masking launches per leaf)
            zw = G.done_stage
        elif not opts.mid_batch_reset:
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
                [(d, x, lo, 1, zw) for d, x in zip(buffer_leaves(self.all_action), a_src)]
                + [(d, x, lo, 0, zw) for d, x in zip(buffer_leaves(s.agent.agent_info), i_src)]
                + [(d, x, None, 0, zw) for d, x in zip(buffer_leaves(G.action_out), a_src)])
            G.post_commit.launch(t)
        else:
            self._commit_rows(self.all_action, action, G, t + 1)
            self._commit_rows(s.agent.agent_info, agent_info, G, t)
            copy_leaves(G.action_out, action)

    # ------------------------------------------------------------------ uploads / downloads
    def upload_special(self, G, first):
        """Host-dependent part of the upload (frame-stacked envs only): full stacks for the
        first step of a batch and for the few envs whose stack was reset."""
        if not G.dedup:
            return
        nb = self.cuda
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

    def upload_steady(self, G):
        """Fixed-address part of the upload: newest frames (or whole observations) + the
        reward/slot/done/reset block."""
        nb = self.cuda
        if G.dedup:
            G.blk_stage.copy_(G.blk_h, non_blocking=nb)      # newest frames + misc: one transfer
            return
        copy_leaves(G.obs_stage, G.step_pyt.observation, non_blocking=nb)
        G.misc_stage.copy_(G.misc_h, non_blocking=nb)

    def download(self, G):
        if not G.zc:      # zero-copy: the step kernel already wrote the host buffer
            copy_leaves(G.step_pyt.action, G.action_out, non_blocking=self.cuda)

    def on_stream(self, G):
        return torch.cuda.stream(G.stream) if G.stream is not None else _NullCtx()

    def begin_batch(self):
        """Per batch, before the first step: see the updated weights, draw the batch's uniforms,
        seed the leading prev_action row (collectors.py:23-24; prev_reward[0] and the done carry
        are committed from the staging block by the first step)."""
        for G in self.groups:
            if G.stream is not None:
                G.stream.wait_stream(torch.cuda.current_stream())
            with self.on_stream(G):
                if G.u_all is not None:
                    G.u_all.uniform_(generator=G.gen)
                _map(lambda d, s: d[0, G.lo:G.hi].copy_(s, non_blocking=True),
                     self.all_action, G.step_pyt.action)

    def synchronize(self):
        if self.cuda:
            for G in self.groups:
                (G.stream or torch.cuda.current_stream()).synchronize()

    # ------------------------------------------------------------------ issue / capture
    def issue(self, G, t, first=False):
        """Enqueue H2D staging -> (graph of) step body -> D2H action on the group's stream."""
        t0 = time.perf_counter()
        G.t_np[0] = t
        with self.on_stream(G):
            self.upload_special(G, first)
            if (self.cuda and self.use_graph and G.graph is None
                    and G.calls >= self.GRAPH_WARMUP_CALLS):
                try:
                    G.graph = self.capture(G)
                except Exception as e:  # noqa: BLE001  (keep sampling: eager step is correct)
                    logger.log(f"GpuSampler: hipGraph capture failed ({type(e).__name__}: {e}); "
                               "continuing with eager per-step launches.")
                    self.use_graph = False
                    G.graph = None
                    torch.cuda.synchronize()
            self.upload_steady(G)
            if G.graph is not None:
                G.graph.replay()
            else:
                self.step_body(G)
            self.download(G)
            G.calls += 1
            if self.cuda:
                G.event.record()
        self.timing["device_issue_s"] += time.perf_counter() - t0

    def finish(self, G):
        """Block until the group's actions are visible to the host."""
        if G.event is not None:
            t0 = time.perf_counter()
            G.event.synchronize()
            self.timing["device_wait_s"] += time.perf_counter() - t0

    def capture(self, G):
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
            self.step_body(G, capturing=True)
        if G.post_entries is not None:
            G.post_commit.set_entries(G.post_entries)
        torch.cuda.synchronize()
        logger.log(f"GpuSampler: captured the step graph of pipeline group {G.idx}.")
        return graph
