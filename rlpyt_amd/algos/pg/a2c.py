"""A2C on the MI355X (API of rlpyt/algos/pg/a2c.py:12-103): one gradient step per batch,
returns from the fused scan, loss + gradients from the fused A2C kernel."""
import torch

from ... import ops
from ...utils.deferred import PendingOptInfo
from ...utils.buffer import buffer_method
from ...agents.base import AgentInputs
from ...utils.quick_args import save__init__args
from .base import OptInfo, PolicyGradientAlgo


def mv_tree(x, mv):
    """``mv`` over every leaf of a (nested) namedarraytuple."""
    from ...utils.buffer import buffer_func
    return buffer_func(x, mv)


class A2C(PolicyGradientAlgo):
    supports_recurrent = True

    def __init__(self, discount=0.99, learning_rate=0.001, value_loss_coeff=0.5,
                 entropy_loss_coeff=0.01, OptimCls=torch.optim.Adam, optim_kwargs=None,
                 clip_grad_norm=1., initial_optim_state_dict=None, gae_lambda=1,
                 normalize_advantage=False):
        if optim_kwargs is None:
            optim_kwargs = dict()
        save__init__args(locals())

    def initialize(self, *args, **kwargs):
        super().initialize(*args, **kwargs)
        self._batch_size = self.batch_spec.size

    def optimize_agent(self, itr, samples):
        if hasattr(self.agent, "update_obs_rms"):
            self.agent.update_obs_rms(samples.env.observation)
        self.optimizer.zero_grad(set_to_none=True)
        loss, scalars = self.loss(samples)
        loss.backward()
        grad_norm = self.clip_and_step()
        row = torch.stack([scalars[0], grad_norm.to(scalars.dtype), scalars[3], scalars[4]])
        self.update_counter += 1
        # (read back when a field is first looked at: utils/deferred.py)
        return PendingOptInfo(OptInfo, [row], lambda host: OptInfo(*host[0].tolist()))

    def loss(self, samples):
        mv = self.on_device
        agent_inputs = AgentInputs(observation=mv(samples.env.observation),
                                   prev_action=mv(samples.agent.prev_action),
                                   prev_reward=mv(samples.env.prev_reward))
        if self.agent.recurrent:
            # whole columns restart from the state recorded at row 0 (a2c.py:80-85)
            init_rnn_state = buffer_method(mv_tree(samples.agent.agent_info.prev_rnn_state[0], mv),
                                           "transpose", 0, 1)
            init_rnn_state = buffer_method(init_rnn_state, "contiguous")
            dist_info, value, _rnn_state = self.agent(*agent_inputs, init_rnn_state)
        else:
            dist_info, value = self.agent(*agent_inputs)
        return_, advantage, valid = self.process_returns(samples)
        return ops.a2c_loss(dist_info.prob, value, mv(samples.agent.action), advantage, return_,
                            valid, self.value_loss_coeff, self.entropy_loss_coeff)
