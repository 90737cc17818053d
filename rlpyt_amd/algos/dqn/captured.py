"""One DQN-family update as ONE hipGraph.

A batch-128 update of BASELINE config #3 is ~100 kernel launches of a few microseconds each (tree
descent, 14 small gathers, the 32-64-64 conv stack forward / backward through the vendor library,
target pass, fused loss, clip + Adam, priority write-back): issued one by one it is launch-bound --
2.27 ms per update, 440 updates/s (profiles/r4_bench_dqn.json) -- although its kernels add up to a
fraction of that.  Here the update

    draw (device sum tree) -> gather the step batch -> online + target passes -> fused loss ->
    backward -> clip + Adam -> priority write-back -> diagnostics row

is captured once and replayed per update.  What differs between updates comes from device memory:
``ops.update_tick`` copies row ``ctr`` of per-call host tables -- the batch's uniforms (the
reference's ``np.random.rand(n)`` stream, drawn for all updates of the call in the same order;
uniform replay: its ``randint`` index pairs), ``lr / bc1`` and ``1 / sqrt(bc2)`` of the Adam step
(host double arithmetic, as the eager kernel's) -- to fixed addresses, the importance exponent lives
in a device scalar, ``ClipAdam.launch_captured_step`` advances ``ctr``.  Target-network updates and
the once-per-call read-back of the diagnostics stay outside the graph.

Reference semantics kept: rlpyt/algos/dqn/dqn.py:158-190 (update loop, target interval, priority
update), replays/non_sequence/prioritized.py:56-79 (draw, importance weights).  The eager loop of
``DQN.optimize_agent`` is the fallback (and the A/B: ``RLPYT_DQN_GRAPH=0``).
"""
import os

import numpy as np
import torch

from ... import ops
from ...utils import logger

ENABLED = os.environ.get("RLPYT_DQN_GRAPH", "1") != "0"


class CapturedUpdates:
    def __init__(self, algo):
        self.algo = algo
        self.graph = None
        self.failed = False
        self.bufs = None

    # ------------------------------------------------------------------ applicability
    def applies(self):
        """Capture needs: the fused optimizer with initialised state on one common step count (an
        eager update ran), a replay buffer that can draw from device-resident uniforms / indices,
        a device-resident ring, and a fixed number of updates per call."""
        a = self.algo
        opt, rb = a.optimizer, a.replay_buffer
        # single rank only: under DistributedDataParallel the reducer's bookkeeping and the RCCL
        # all-reduces would land inside the capture (and ranks could end up on different paths)
        single = (getattr(a, "world_size", 1) == 1 and not isinstance(
            getattr(a.agent, "model", None), torch.nn.parallel.DistributedDataParallel))
        return (ENABLED and single and not self.failed and a.agent.device.type == "cuda"
                and hasattr(opt, "captured_ready") and opt.captured_ready()
                and hasattr(rb, "sample_batch_device") and rb.can_sample_on_device()
                and a.updates_per_optimize >= 1 and getattr(a, "CAPTURABLE", False))

    # ------------------------------------------------------------------ static buffers
    def _allocate(self):
        a = self.algo
        dev, n, k = a.agent.device, int(a.batch_size), int(a.updates_per_optimize)
        rb = a.replay_buffer
        words = n if rb.PRIORITIZED else 2 * n       # f64 uniforms, or (T, B) int64 index pairs
        self.bufs = dict(
            ctr=torch.zeros(1, dtype=torch.int64, device=dev),
            tick_idx=torch.zeros(1, dtype=torch.int64, device=dev),
            hyper=torch.zeros(4, dtype=torch.float32, device=dev),
            table=torch.zeros((k, 4), dtype=torch.float32, device=dev),
            draw_all=torch.zeros(k * words, dtype=torch.int64, device=dev),
            draw_cur=torch.zeros(words, dtype=torch.int64, device=dev),
            beta=torch.zeros(1, dtype=torch.float64, device=dev),
            ring=torch.zeros((k, 2), dtype=torch.float32, device=dev),
            vec=torch.zeros((k, len(range(0, n, 8))), dtype=torch.float32, device=dev))
        self.k, self.n, self.words = k, n, words

    def _body(self):
        """The captured update (also run eagerly once, right before the capture)."""
        a, b = self.algo, self.bufs
        rb = a.replay_buffer
        ops.update_tick(b["ctr"], b["table"], b["hyper"], b["draw_all"], b["draw_cur"], b["tick_idx"])
        if rb.PRIORITIZED:
            batch = rb.sample_batch_device(self.n, uniforms=b["draw_cur"].view(torch.float64),
                                           beta=b["beta"])
        else:
            batch = rb.sample_batch_device(self.n, idxs=(b["draw_cur"][:self.n],
                                                         b["draw_cur"][self.n:]))
        a.optimizer.zero_grad(set_to_none=True)
        loss, td_abs = a.loss(batch)
        loss.backward(ops.unit_seed(loss.device))
        grad_norm = a.optimizer.launch_captured_step(a.clip_grad_norm, b["hyper"], b["ctr"])
        if a.prioritized_replay:
            rb.update_batch_priorities(td_abs)
        row = torch.stack([loss.detach().float(), grad_norm.to(torch.float32)])
        b["ring"].index_copy_(0, b["tick_idx"], row.unsqueeze(0))
        b["vec"].index_copy_(0, b["tick_idx"], td_abs.detach().float()[::8].unsqueeze(0))

    def _fill_tables(self, itr):
        """Host side of one ``optimize_agent`` call: the k updates' draws (consuming ``np.random``
        exactly as k eager updates would) and Adam scalars."""
        a, b, k, n = self.algo, self.bufs, self.k, self.n
        rb = a.replay_buffer
        if rb.PRIORITIZED:
            draws = np.random.rand(k, n).view(np.int64)
            b["beta"].fill_(float(rb.draws.beta))
        else:
            draws = np.stack([np.concatenate(rb.draws.draw(n)[:2]) for _ in range(k)]).astype(np.int64)
        b["draw_all"].copy_(torch.from_numpy(np.ascontiguousarray(draws).reshape(-1)),
                            non_blocking=True)
        rows = [r + (0., 0.) for r in a.optimizer.hyper_rows(k)]
        b["table"].copy_(torch.tensor(rows, dtype=torch.float32), non_blocking=True)
        b["ctr"].zero_()

    # ------------------------------------------------------------------ one optimize_agent call
    def run(self, itr, log):
        """``updates_per_optimize`` updates; diagnostics go to ``log`` (an ``UpdateLog``).  Returns
        False when the graph path is unavailable (the caller runs its eager loop)."""
        if not self.applies():
            return False
        a = self.algo
        if self.bufs is None:
            self._allocate()
        self._fill_tables(itr)
        done = 0
        if self.graph is None:
            # the first body runs eagerly and OUTSIDE the try: an error of the update itself (a
            # shape assertion, a missing kernel) must surface, not turn into a slower run
            self._body()                           # stream workspaces, autograd buffers
            done = 1
            self._after_update()
            torch.cuda.synchronize()
            a.optimizer.zero_grad(set_to_none=True)
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    self._body()
                self.graph = graph
                logger.log(f"{type(a).__name__}: one update captured as a hipGraph "
                           f"({self.k} replays per iteration).")
            except RuntimeError as e:     # the capture itself refused: eager bodies are correct
                logger.log(f"{type(a).__name__}: update-graph capture failed "
                           f"({type(e).__name__}: {e}); continuing with eager updates.")
                self.failed = True
                self.graph = None
                torch.cuda.synchronize()
        for _ in range(done, self.k):
            if self.graph is not None:
                self.graph.replay()
            else:
                self._body()
            self._after_update()
        a.optimizer.advance_steps(self.k)
        log.add_rows(self.bufs["ring"], tdAbsErr=self.bufs["vec"])
        return True

    def _after_update(self):
        a = self.algo
        a.update_counter += 1
        if a.update_counter % a.target_update_interval == 0:
            a.agent.update_target(a.target_update_tau)
