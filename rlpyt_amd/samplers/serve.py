"""The T time steps of a batch: the role of the loop in ``ActionServer.serve_actions``
(rlpyt/samplers/parallel/gpu/action_server.py:44-58).

Two drivers over the same per-group step (``DeviceBatch.issue`` / captured graphs):

* ``serve_python``: wait for a group's workers, issue its step, publish the actions -- used while
  the step graphs are being warmed up / captured, on CPU, and for agents whose step holds RNG state;
* ``NativeServe``: once every group has an RNG-free captured graph the whole loop runs in C
  (``rlpyt_sampler_serve``, csrc/serve.cpp): event-driven, one issuing and one retiring thread,
  groups free to sit at different time steps.
"""
import ctypes
import os
import time

import torch

from ..utils import logger
from ..utils.buffer import buffer_leaves
from ..utils.misc import usable_cpus


def serve_python(dev, sync, runners0, T, timing, completed):
    """Python time-step loop.  ``sync``: the master's ``StepSync`` (None: envs are stepped inline
    by ``runners0``, the [(group, EnvRunner)] list of the single in-process "worker")."""
    par = sync is not None
    for t in range(T):
        for G in dev.groups:
            if par:
                t0 = time.perf_counter()
                sync.master_wait_obs(G.idx)
                timing["wait_env_s"] += time.perf_counter() - t0
            dev.issue(G, t, first=(t == 0))
            if not par:
                dev.finish(G)
                t0 = time.perf_counter()
                runners0[G.idx][1].step_all(t, completed)
                timing["wait_env_s"] += time.perf_counter() - t0
        if par:
            for G in dev.groups:
                dev.finish(G)
                sync.master_post_act(G.idx)


class NativeServe:
    """Group table of ``rlpyt_sampler_serve`` + the call.  ``build`` returns None when the native
    loop does not apply (a group without a captured, RNG-free graph; too many copy descriptors)."""

    def __init__(self, table, dev, sync, world_size):
        self.table, self.dev, self.sync = table, dev, sync
        self.tmg = (ctypes.c_double * 8)()
        self.spin = None
        self.world_size = world_size

    @classmethod
    def build(cls, dev, sync, world_size):
        from .. import _lib
        groups = dev.groups
        if not all(G.graph is not None and G.u_all is not None for G in groups):
            return None
        arr = (_lib.StepGroup * len(groups))()
        for G, sg in zip(groups, arr):
            sg.act_word, sg.obs_word = sync.act[G.idx], sync.obs[G.idx]
            sg.n_workers = G.n_workers
            h2d = []
            if G.dedup:
                h2d.append((G.blk_stage, G.blk_h))
            else:
                h2d += list(zip(buffer_leaves(G.obs_stage), buffer_leaves(G.step_pyt.observation)))
                h2d.append((G.misc_stage, G.misc_h))
            d2h = ([] if G.zc else
                   list(zip(buffer_leaves(G.step_pyt.action), buffer_leaves(G.action_out))))
            if len(h2d) > 8 or len(d2h) > 4:
                return None
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
            stream = G.stream or torch.cuda.current_stream(dev.device)
            sg.stream = stream.cuda_stream
            G.event.record(stream)
            sg.event = G.event.cuda_event
        logger.log("GpuSampler: time-step loop handed to rlpyt_sampler_serve (native).")
        return cls(arr, dev, sync, world_size)

    def _spin(self):
        """Idle passes the two serve threads may poll before they start sleeping: polling needs
        two spare cores per rank on top of the env workers; under a tight CPU quota (several
        ranks in one quota-limited container) the threads sleep between hand-offs instead."""
        if self.spin is None:
            per_rank = usable_cpus() / max(self.world_size, 1)
            self.spin = 20000 if per_rank >= 6 else 0
            if os.environ.get("RLPYT_SERVE_SPIN"):          # A/B experiments (rollout sweep)
                self.spin = int(os.environ["RLPYT_SERVE_SPIN"])
        return self.spin

    def serve(self, T, timing):
        """Run the T steps of a batch (and, once every group has a captured tail graph, the
        bootstrap pass behind them).  Returns True when the tail ran here."""
        from .. import _lib
        sync, groups = self.sync, self.dev.groups
        tail = all(G.get("tail_graph") is not None for G in groups)
        for G, sg in zip(groups, self.table):
            sg.acts, sg.rounds = sync.acts[G.idx] & 0xffffffff, sync.rounds[G.idx] & 0xffffffff
            sg.tail_graph_exec = G.tail_graph.raw_cuda_graph_exec() if tail else None
        tmg = self.tmg
        for i in range(8):
            tmg[i] = 0.
        _lib.check(_lib.lib.rlpyt_sampler_serve(self.table, len(groups), 0, T, self._spin(),
                                                120000, tmg), "rlpyt_sampler_serve")
        for G in groups:
            sync.acts[G.idx] += T
            sync.rounds[G.idx] += T + (1 if tail else 0)
            G.calls += T
        timing["wait_env_s"] += tmg[0]
        timing["device_issue_s"] += tmg[1]
        timing["device_wait_s"] += tmg[2]
        for k, i in (("chain_issue_s", 3), ("chain_device_s", 4), ("chain_post_s", 5),
                     ("chain_steps", 6)):
            timing[k] = timing.get(k, 0.) + tmg[i]
        return tail
