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
from ...utils.deferred import PendingOptInfo
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
        stats = []
        # index mode: the loss kernel and the optimizer write their scalars straight into one table
        # row per update (loss, pi_loss, value_loss, entropy, perplexity | gradNorm) -- no stack / cat
        # launch per minibatch, and the backward pass is seeded with a standing 1
        table = seed = None
        if fused_idx and valid is None and hasattr(self.optimizer, "clip_and_step"):
            table = torch.empty((self.epochs * self.minibatches, 6), dtype=torch.float32, device=dev)
            seed = self._backward_seed(dev)
        n_rows = 0
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
                if table is not None:
                    # index mode: the conv kernels and the head+loss kernel read the [T,B] batch
                    # arrays at (idx % T, idx // T) themselves -- no gather launch at all
                    row = table[n_rows]
                    loss, _ = self.loss(AgentInputs(mb_obs, None, None), action, return_,
                                        advantage, None, old_prob, flat_idx=idx_dev,
                                        unit_grad=True, scalars_out=row[:5])
                    torch.autograd.backward(loss, grad_tensors=seed)
                    self.optimizer.clip_and_step(self.clip_grad_norm, norm_out=row[5:])
                    n_rows += 1
                    self.update_counter += 1
                    continue
                if fused_idx and valid is None:
                    loss, scalars = self.loss(AgentInputs(mb_obs, None, None), action, return_,
                                              advantage, None, old_prob, flat_idx=idx_dev,
                                              unit_grad=True)      # loss.backward() right below
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
        # ONE read-back of the diagnostics per call, and not a blocking one: the returned object waits
        # for the copy when a field is first read (utils/deferred.py) -- a runner that stores
        # diagnostics every iteration sees no difference, one that reads them when it logs lets the
        # host start on the next sampling phase while the last minibatches run
        if table is not None:
            rows, cols = table[:n_rows], (0, 5, 3, 4)
        elif stats:
            rows, cols = torch.stack(stats), (0, 1, 2, 3)
        else:
            return OptInfo(*([] for _ in range(4)))

        def build(host):
            data = host[0].tolist()
            return OptInfo(*([r[c] for r in data] for c in cols))

        return PendingOptInfo(OptInfo, [rows], build)

    def _backward_seed(self, dev):
        """A standing scalar 1 on the device: ``loss.backward()`` would fill a fresh one per minibatch."""
        one = getattr(self, "_seed_one", None)
        if one is None or one.device != dev:
            one = self._seed_one = torch.ones((), dtype=torch.float32, device=dev)
        return one

    def loss(self, agent_inputs, action, return_, advantage, valid, old_prob,
             init_rnn_state=None, flat_idx=None, unit_grad=False, scalars_out=None):
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
                                     action, advantage, return_, valid, self.ratio_clip,
                                     self.value_loss_coeff, self.entropy_loss_coeff,
                                     flat_idx=flat_idx, trunk_bias=tb, unit_grad=unit_grad,
                                     scalars_out=scalars_out)
        assert flat_idx is None and scalars_out is None, "index-mode loss needs the fused head+loss kernel"
        if init_rnn_state is not None:
            init_rnn_state = buffer_method(init_rnn_state, "transpose", 0, 1)
            init_rnn_state = buffer_method(init_rnn_state, "contiguous")
            dist_info, value, _ = self.agent(*agent_inputs, init_rnn_state)
        else:
            dist_info, value = self.agent(*agent_inputs)
        return ops.ppo_loss(dist_info.prob, value, old_prob, action, advantage, return_, valid,
                            self.ratio_clip, self.value_loss_coeff, self.entropy_loss_coeff)
