"""PPO on the MI355X (API and hyper-parameters of rlpyt/algos/pg/ppo.py:16-154).

What changed underneath, relative to the reference's ``optimize_agent``:
* the sample batch is already in HBM (GpuSampler) -- no bulk H2D (ppo.py:72);
* ``process_returns`` is one fused HIP scan (+ an in-place normalise);
* each minibatch is gathered on the device with ``rlpyt_gather_tb`` honouring the
  reference's index map ``idx -> (idx % T, idx // T)`` (ppo.py:94-95);
* ratio / clip / min-surrogate / value MSE / entropy / perplexity and all their gradients
  are ONE fused forward+backward kernel (ppo.py:136-153) feeding autograd;
* per-minibatch diagnostics stay on the device; a single D2H per iteration replaces the
  reference's 4 ``.item()`` syncs per minibatch (ppo.py:106-109).
"""
import os
import numpy as np
import torch

from ... import ops
from ...agents.base import AgentInputs
from ...utils.buffer import buffer_func, buffer_method
from ...utils.collections import namedarraytuple
from ...utils.misc import iterate_mb_idxs
from ...utils.quick_args import save__init__args
from .base import OptInfo, PolicyGradientAlgo

LossInputs = namedarraytuple("LossInputs", ["agent_inputs", "action", "return_", "advantage",
                                            "valid", "old_dist_info"])


class PPO(PolicyGradientAlgo):
    supports_recurrent = True

    def __init__(self, discount=0.99, learning_rate=0.001, value_loss_coeff=1.,
                 entropy_loss_coeff=0.01, OptimCls=torch.optim.Adam, optim_kwargs=None,
                 clip_grad_norm=1., initial_optim_state_dict=None, gae_lambda=1,
                 minibatches=4, epochs=4, ratio_clip=0.1, linear_lr_schedule=True,
                 normalize_advantage=False, fused_head_loss=True):
        if optim_kwargs is None:
            optim_kwargs = dict()
        save__init__args(locals())

    def initialize(self, *args, **kwargs):
        super().initialize(*args, **kwargs)
        self._batch_size = self.batch_spec.size // self.minibatches     # per-update batch
        if self.linear_lr_schedule:
            # both the learning rate and the ratio clip decay linearly to 0 over the run
            # (ppo.py:46-57,112-114): remaining(itr) is the factor for iteration itr
            self._ratio_clip = self.ratio_clip
            self.lr_scheduler = torch.optim.lr_scheduler.LambdaLR(self.optimizer, self.remaining)

    def remaining(self, itr):
        return (self.n_itr - itr) / self.n_itr

    def optimize_agent(self, itr, samples):
        recurrent = self.agent.recurrent
        mv = self.on_device
        dev = self.agent.device
        agent_inputs = AgentInputs(observation=mv(samples.env.observation),
                                   prev_action=mv(samples.agent.prev_action),
                                   prev_reward=mv(samples.env.prev_reward))
        if hasattr(self.agent, "update_obs_rms"):
            self.agent.update_obs_rms(agent_inputs.observation)
        return_, advantage, valid = self.process_returns(samples)
        action = mv(samples.agent.action).contiguous()
        old_prob = mv(samples.agent.agent_info.dist_info.prob).contiguous()
        uses_prev = getattr(self.agent, "uses_prev_inputs", True)
        fused_idx = (self.fused_head_loss and not recurrent and not uses_prev
                     and getattr(self.agent, "supports_fused_head_loss", False))
        if recurrent:
            init_rnn_state = samples.agent.agent_info.prev_rnn_state[0]
        T, B = samples.env.reward.shape[:2]
        batch_size = B if recurrent else T * B
        mb_size = batch_size // self.minibatches
        if fused_idx and valid is None and self._update_graph_ok(agent_inputs.observation):
            return self._optimize_captured(itr, agent_inputs.observation, action, return_,
                                           advantage, old_prob, batch_size, mb_size)
        stats = []
        # [T,B] fields the gather kernel can slice need contiguous storage; prev_action /
        # prev_reward are [:-1] views of [T+1,B] arrays (contiguous as [T,B] blocks).
        for _ in range(self.epochs):
            # one upload of the epoch's shuffled index chunks (utils/misc.py:6-17 order kept)
            chunks = list(iterate_mb_idxs(batch_size, mb_size, shuffle=True))
            epoch_idx = torch.from_numpy(np.ascontiguousarray(np.concatenate(chunks))).to(
                dev, non_blocking=True) if chunks else None
            for k in range(len(chunks)):
                self.optimizer.zero_grad(set_to_none=True)
                idx_dev = epoch_idx[k * mb_size:(k + 1) * mb_size]
                if recurrent:
                    # whole trajectories: only the Batch axis is shuffled; every column restarts
                    # from the LSTM state recorded at row 0 (ppo.py:84-86,93-99)
                    col = lambda x: x.index_select(1, idx_dev)      # noqa: E731
                    mb_inputs = AgentInputs(observation=col(agent_inputs.observation),
                                            prev_action=col(agent_inputs.prev_action),
                                            prev_reward=col(agent_inputs.prev_reward))
                    rnn_state = buffer_func(init_rnn_state, lambda x: mv(x).index_select(0, idx_dev))
                    loss, scalars = self.loss(mb_inputs, col(action), col(return_), col(advantage),
                                              None if valid is None else col(valid),
                                              col(old_prob), init_rnn_state=rnn_state)
                    loss.backward()
                    grad_norm = self.clip_and_step()
                    stats.append(torch.stack([scalars[0], grad_norm.to(scalars.dtype), scalars[3],
                                              scalars[4]]))
                    self.update_counter += 1
                    continue
                mb_obs = self.agent.gather_observation(agent_inputs.observation, idx_dev)
                if fused_idx and valid is None:
                    # index mode: the conv kernels and the head+loss kernel read the [T,B] batch
                    # arrays at (idx % T, idx // T) themselves -- no gather launch at all
                    loss, scalars = self.loss(AgentInputs(mb_obs, None, None), action, return_,
                                              advantage, None, old_prob, flat_idx=idx_dev)
                else:
                    if uses_prev:
                        mb_pa = ops.gather_tb(agent_inputs.prev_action.contiguous(), idx_dev)
                        mb_pr = ops.gather_tb(agent_inputs.prev_reward.contiguous(), idx_dev)
                    else:
                        mb_pa = mb_pr = None
                    mb_inputs = AgentInputs(observation=mb_obs, prev_action=mb_pa, prev_reward=mb_pr)
                    mb_action = ops.gather_tb(action, idx_dev)
                    mb_return = ops.gather_tb(return_, idx_dev)
                    mb_adv = ops.gather_tb(advantage, idx_dev)
                    mb_valid = None if valid is None else ops.gather_tb(valid, idx_dev)
                    mb_old_prob = ops.gather_tb(old_prob, idx_dev)
                    loss, scalars = self.loss(mb_inputs, mb_action, mb_return, mb_adv, mb_valid,
                                              mb_old_prob)
                loss.backward()
                grad_norm = self.clip_and_step()
                # loss, gradNorm, entropy, perplexity -- kept on the device until the end
                stats.append(torch.stack([scalars[0], grad_norm.to(scalars.dtype), scalars[3],
                                          scalars[4]]))
                self.update_counter += 1
        if self.linear_lr_schedule:
            self.lr_scheduler.step()
            self.ratio_clip = self._ratio_clip * (self.n_itr - itr) / self.n_itr
        host = self.diagnostics_to_host(stats)
        opt_info = OptInfo(*([row[k] for row in host] for k in range(4)))
        return opt_info

    # ------------------------------------------------------------------ captured updates
    # The 16 minibatch updates of an iteration launch the same ~20 kernels on the same addresses;
    # issued one by one they leave 6 x 10 us + 2 x 6 us of launch gaps per update and three ATen
    # helper launches (profiles/r3_trace_gaps.txt: period 1369 us for 1310 us of kernels).  After
    # the first eager iteration ONE update is captured into a hipGraph and replayed for every
    # minibatch: what changes between updates -- the minibatch indices, lr / Adam bias corrections,
    # the ratio clip -- comes from device memory (``ops.update_tick`` fills it from a per-iteration
    # host table, same double-precision arithmetic as the eager path), the diagnostics row goes to
    # a device ring.  MEASURED (profiles/r4_update_graph_trace_gaps.txt): the replayed update has no
    # gaps between its kernels, yet its period is the eager loop's (1301.6 vs 1299.2 us per
    # minibatch) -- what the eager trace shows as ~10 us "gaps" is the dispatch / cache write-back
    # latency between dependent kernels, which a graph pays inside the next kernel's duration
    # instead.  Bit-identical results (tests/test_pg_gpu.py), no speed-up: opt-in only
    # (RLPYT_UPDATE_GRAPH=1 or ``use_update_graph = True``).
    use_update_graph = os.environ.get("RLPYT_UPDATE_GRAPH", "0") == "1"

    def _update_graph_ok(self, observation):
        opt = self.optimizer
        return (self.use_update_graph and self.world_size == 1 and observation.is_cuda
                and hasattr(opt, "captured_ready") and opt.captured_ready()
                and all(p.grad is not None for p in self.agent.parameters())
                and not getattr(self, "_update_graph_failed", False))

    def _optimize_captured(self, itr, observation, action, return_, advantage, old_prob,
                           batch_size, mb_size):
        dev = observation.device
        n_up = self.epochs * (batch_size // mb_size)
        chunks = []
        for _ in range(self.epochs):          # the reference's shuffle order (utils/misc.py:6-17)
            chunks += list(iterate_mb_idxs(batch_size, mb_size, shuffle=True))
        idx_all = torch.from_numpy(np.ascontiguousarray(np.concatenate(chunks))).to(
            dev, non_blocking=True)
        rows = [r + (float(self.ratio_clip), 0.) for r in self.optimizer.hyper_rows(n_up)]
        key = (observation.data_ptr(), action.data_ptr(), return_.data_ptr(), advantage.data_ptr(),
               old_prob.data_ptr(), mb_size, n_up, float(self.clip_grad_norm or 0.),
               float(self.value_loss_coeff), float(self.entropy_loss_coeff))
        G = getattr(self, "_ug", None)
        if G is None or G["key"] != key:
            G = self._ug = dict(
                key=key, graph=None,
                ctr=torch.zeros(1, dtype=torch.int64, device=dev),
                tick_idx=torch.zeros(1, dtype=torch.int64, device=dev),
                hyper=torch.zeros(4, dtype=torch.float32, device=dev),
                table=torch.zeros((n_up, 4), dtype=torch.float32, device=dev),
                idx_all=torch.zeros(n_up * mb_size, dtype=torch.int64, device=dev),
                idx=torch.zeros(mb_size, dtype=torch.int64, device=dev),
                ring=torch.zeros((n_up, 4), dtype=torch.float32, device=dev))
        G["table"].copy_(torch.tensor(rows, dtype=torch.float32), non_blocking=True)
        G["idx_all"].copy_(idx_all, non_blocking=True)
        G["ctr"].zero_()

        def one_update():
            ops.update_tick(G["ctr"], G["table"], G["hyper"], G["idx_all"], G["idx"], G["tick_idx"])
            mb_obs = self.agent.gather_observation(observation, G["idx"])
            loss, scalars = self.loss(AgentInputs(mb_obs, None, None), action, return_, advantage,
                                      None, old_prob, flat_idx=G["idx"],
                                      ratio_clip=G["hyper"][2:3])
            loss.backward()
            grad_norm = self.optimizer.launch_captured_step(self.clip_grad_norm, G["hyper"], G["ctr"])
            row = torch.stack([scalars[0], grad_norm.to(scalars.dtype), scalars[3], scalars[4]])
            G["ring"].index_copy_(0, G["tick_idx"], row.unsqueeze(0))

        done = 0
        if G["graph"] is None:
            # one eager update first (workspaces of this stream, autograd buffers), then capture
            self.optimizer.zero_grad(set_to_none=True)
            one_update()
            done = 1
            try:
                torch.cuda.synchronize()
                self.optimizer.zero_grad(set_to_none=True)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    one_update()
                G["graph"] = graph
            except Exception as e:  # noqa: BLE001  (keep training: eager updates are correct)
                from ...utils import logger
                logger.log(f"PPO: update-graph capture failed ({type(e).__name__}: {e}); "
                           "continuing with eager minibatch updates.")
                self._update_graph_failed = True      # next iterations: the plain eager loop
                torch.cuda.synchronize()
        for _ in range(done, n_up):
            if G["graph"] is not None:
                G["graph"].replay()
            else:                                     # same update, launched kernel by kernel
                self.optimizer.zero_grad(set_to_none=True)
                one_update()
        self.optimizer.advance_steps(n_up)
        self.update_counter += n_up
        if self.linear_lr_schedule:
            self.lr_scheduler.step()
            self.ratio_clip = self._ratio_clip * (self.n_itr - itr) / self.n_itr
        host = G["ring"].cpu().tolist()
        return OptInfo(*([row[k] for row in host] for k in range(4)))

    def loss(self, agent_inputs, action, return_, advantage, valid, old_prob,
             init_rnn_state=None, flat_idx=None, ratio_clip=None):
        """Fused PPO loss on a minibatch (already gathered, all in HBM).  Returns
        ``(loss, scalars)`` with scalars = [loss, pi_loss, value_loss, entropy, perplexity].
        ``flat_idx``: the loss inputs are whole ``[T,B,...]`` arrays and sample m is row
        ``(idx % T, idx // T)`` (fused head+loss kernel only)."""
        if init_rnn_state is None and self.fused_head_loss and getattr(
                self.agent, "supports_fused_head_loss", False):
            # heads + softmax + loss + all their gradients in one kernel pass over the trunk
            # (and the trunk's bias + ReLU where the model hands out its pre-activation)
            if hasattr(self.agent, "trunk_pre") and os.environ.get("RLPYT_TRUNK_FUSION", "1") != "0":
                h, tb, pi_m, v_m = self.agent.trunk_pre(*agent_inputs)
            else:
                (h, pi_m, v_m), tb = self.agent.trunk(*agent_inputs), None
            return ops.ppo_head_loss(h, pi_m.weight, pi_m.bias, v_m.weight, v_m.bias, old_prob,
                                     action, advantage, return_, valid,
                                     self.ratio_clip if ratio_clip is None else ratio_clip,
                                     self.value_loss_coeff, self.entropy_loss_coeff,
                                     flat_idx=flat_idx, trunk_bias=tb)
        assert flat_idx is None, "index-mode loss needs the fused head+loss kernel"
        if init_rnn_state is not None:
            init_rnn_state = buffer_method(init_rnn_state, "transpose", 0, 1)
            init_rnn_state = buffer_method(init_rnn_state, "contiguous")
            dist_info, value, _ = self.agent(*agent_inputs, init_rnn_state)
        else:
            dist_info, value = self.agent(*agent_inputs)
        return ops.ppo_loss(dist_info.prob, value, old_prob, action, advantage, return_, valid,
                            self.ratio_clip, self.value_loss_coeff, self.entropy_loss_coeff)
