"""End of a batch on the device: the bootstrap-value pass on the observation after the last step
(rlpyt/samplers/parallel/gpu/action_server.py:60-68) -- as ONE more step of a group's fused kernels
where that applies (eagerly the first time, a captured hipGraph afterwards, which is also what lets the C
serve loop run it), else through ``agent.value``.  Mixed into ``DeviceBatch`` (``device.py``), whose
buffers, groups and upload methods it uses."""
import torch

from ..utils import logger
from ..utils.buffer import _map, buffer_leaves


def copy_leaves(dst, src, non_blocking=False):
    for d, s in zip(buffer_leaves(dst), buffer_leaves(src)):
        d.copy_(s, non_blocking=non_blocking)


class TailPass:
    def _tail_fused_applies(self, G):
        s, agent, opts = self.samples, self.agent, self.opts
        if not (self.cuda and G.dedup and G.u_all is not None and opts.mid_batch_reset
                and opts.fused_step and opts.fused_push
                and "bootstrap_value" in s.agent and not agent.recurrent
                and not getattr(agent, "uses_prev_inputs", True)
                and hasattr(agent, "value_into") and isinstance(self.all_action, torch.Tensor)):
            return False
        bv = s.agent.bootstrap_value
        return isinstance(bv, torch.Tensor) and bv.dtype == torch.float32 and bv.is_contiguous()

    def _tail_fused_device(self, G):
        """Device part of the fused tail (capturable: fixed addresses, t = T arrives in the misc
        block): rebuild obs_T into the staging buffer, commit reward / done rows T, trunk + VALUE
        head only -> bootstrap_value[0, lo:hi]."""
        self.agent.select_envs(G.lo, G.hi)
        bv = self.samples.agent.bootstrap_value
        return bool(self.agent.value_into(self._push_binding(G), G.obs_stage, bv[0, G.lo:G.hi]))

    def tail_fused(self, G):
        """The tail as ONE more step of the group's fused kernels (non-recurrent agents that ignore
        prev inputs, frame-stacked uploads, mid-batch reset): upload the newest frames + misc
        block with t = T, then ``_tail_fused_device`` -- eagerly the first time, as a captured
        hipGraph from the second batch on (which is also what lets the C serve loop run the tail
        itself, ``NativeServe``).  Returns False when it does not apply (the caller then runs
        ``tail_body``)."""
        if not self._tail_fused_applies(G):
            return False
        G.t_np[0] = self.batch_spec.T
        self.upload_special(G, first=False)
        self.upload_steady(G)
        if G.get("tail_graph") is not None:
            G.tail_graph.replay()
            return True
        if self.use_graph and G.get("tail_calls", 0) >= 1:
            try:
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=G.stream, capture_error_mode="thread_local"):
                    ok = self._tail_fused_device(G)
                if ok:
                    G.tail_graph = graph
                    graph.replay()
                    return True
            except Exception as e:  # noqa: BLE001  (the eager tail is correct)
                logger.log(f"GpuSampler: tail-graph capture failed ({type(e).__name__}: {e}); "
                           "eager bootstrap pass.")
                torch.cuda.synchronize()
            G.tail_calls = -(1 << 30)          # do not try again
        ok = self._tail_fused_device(G)
        G.tail_calls = G.get("tail_calls", 0) + 1
        return ok

    def tail_body(self, G):
        """After the last env step of the batch: commit reward/done of step T-1 and compute
        the bootstrap value on obs_T (action_server.py:60-62)."""
        T = self.batch_spec.T
        s, agent = self.samples, self.agent
        lo, hi = G.lo, G.hi
        copy_leaves(G.obs_stage, G.step_pyt.observation, non_blocking=self.cuda)
        G.misc_stage.copy_(G.misc_h, non_blocking=self.cuda)
        self.all_reward[T, lo:hi] = G.reward_stage
        self.all_done[T, lo:hi] = G.done_stage
        # THIS group's envs / recurrent state before any agent call: value() of a recurrent
        # agent reads the selected slot's LSTM state (with several pipeline groups the slot
        # still selected is the group stepped last)
        agent.select_envs(lo, hi)
        if agent.recurrent:
            agent.select_slot(G.idx)
        if "bootstrap_value" in s.agent:
            # as the reference: the value call sees the last action / reward as they are -- the
            # null-after-reset of prev inputs happens AFTER it (action_server.py:60-68); for an
            # env that just finished the bootstrap value is masked by (1 - done) anyway
            prev_action = _map(lambda x: x[T, lo:hi], self.all_action)
            prev_reward = G.reward_stage
            s.agent.bootstrap_value[0, lo:hi] = agent.value(G.obs_stage, prev_action, prev_reward)
        if agent.recurrent:     # end of batch: finished envs restart from a zero state
            agent.reset_where(G.done_stage)   # (action_server.py:63-68)

    def tail(self, G):
        with self.on_stream(G):
            if getattr(G, "tail_is_fused", True) and self.tail_fused(G):
                return
            G.tail_is_fused = False      # decided once per group: the conditions do not change
            self.tail_body(G)

