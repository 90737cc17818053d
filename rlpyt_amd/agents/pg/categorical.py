"""Categorical policy-gradient agent (rlpyt/agents/pg/categorical.py:11-51)."""
import torch

from ...distributions.categorical import Categorical, DistInfo
from ...utils.collections import namedarraytuple
from ..base import AgentStep, BaseAgent, RecurrentAgentMixin

from collections import namedtuple

AgentInfo = namedarraytuple("AgentInfo", ["dist_info", "value"])
AgentInfoRnn = namedarraytuple("AgentInfoRnn", ["dist_info", "value", "prev_rnn_state"])
_HeadOut = namedtuple("_HeadOut", ["prob_rows", "value_rows", "action_rows", "action_out",
                                   "uniforms", "t_dev", "lo"])


class CategoricalPgAgent(BaseAgent):
    def __call__(self, observation, prev_action, prev_reward):
        """Training forward: (DistInfo(prob), value), differentiable, in HBM."""
        if self.uses_prev_inputs:
            prev_action = self.distribution.to_onehot(prev_action)
        else:
            prev_action = prev_reward = None
        obs, pa, pr = self._to_model_device(observation, prev_action, prev_reward)
        pi, value = self.model(obs, pa, pr)
        return self._out((DistInfo(prob=pi), value))

    @property
    def supports_sample_uniforms(self):
        return hasattr(self.sampling_model, "sample_step") and self.device.type == "cuda"

    @property
    def supports_fused_head_loss(self):
        return bool(getattr(self.sampling_model, "fused_head_loss", False))

    def trunk(self, observation, prev_action, prev_reward):
        """Training forward up to the heads: the trunk output ``[M, fc]`` (differentiable) and
        the head modules, for ``ops.ppo_head_loss``.  Goes through ``self.model`` so that a
        DistributedDataParallel wrapper sees the forward."""
        obs, _, _ = self._to_model_device(observation, None, None)
        h = self.model(obs, None, None, features_only=True)
        m = self.sampling_model
        return h, m.pi, m.value

    def trunk_pre(self, observation, prev_action, prev_reward):
        """As ``trunk`` but returns ``(z, trunk_bias, pi, value)``: the trunk's pre-activation
        without its bias and that bias (``ops.ppo_head_loss(..., trunk_bias=)`` applies both), or
        the post-activation output and ``None`` where the model cannot split them."""
        obs, _, _ = self._to_model_device(observation, None, None)
        z, tb = self.model(obs, None, None, features_only="pre")
        m = self.sampling_model
        return z, tb, m.pi, m.value

    def make_distribution(self, n_actions):
        return Categorical(dim=n_actions)

    @torch.no_grad()
    def step(self, observation, prev_action, prev_reward):
        """Sampling forward: one batched model call + on-device categorical draw.  Models
        that provide ``sample_step`` (AtariFfModel) fuse heads, softmax and the draw."""
        m = self.sampling_model
        fused = (hasattr(m, "sample_step") and isinstance(observation, torch.Tensor)
                 and observation.is_cuda and observation.dim() == 4)
        if self.uses_prev_inputs and prev_action is not None:
            prev_action = self.distribution.to_onehot(prev_action)
        else:
            prev_action = prev_reward = None
        obs, pa, pr = self._to_model_device(observation, prev_action, prev_reward)
        if fused:
            action, pi, value = m.sample_step(obs, pa, pr, generator=self.sample_generator,
                                              uniforms=self.sample_uniforms)
            dist_info = DistInfo(prob=pi)
        else:
            pi, value = m(obs, pa, pr)
            dist_info = DistInfo(prob=pi)
            action = self.distribution.sample(dist_info, generator=self.sample_generator)
        agent_info = AgentInfo(dist_info=dist_info, value=value)
        return self._out(AgentStep(action=action, agent_info=agent_info))

    @torch.no_grad()
    def step_into(self, observation, prev_action, prev_reward, binding):
        m = self.sampling_model
        if self.uses_prev_inputs or not hasattr(m, "sample_step_into"):
            return False
        info = binding.agent_info_rows
        if not (isinstance(binding.action_rows, torch.Tensor) and hasattr(info, "dist_info")):
            return False
        out = _HeadOut(prob_rows=info.dist_info.prob, value_rows=info.value,
                       action_rows=binding.action_rows, action_out=binding.action_out,
                       uniforms=binding.uniforms, t_dev=binding.t_dev, lo=binding.lo)
        push = getattr(binding, "push", None)
        if push is not None:      # the model also rebuilds the frame stacks of row t
            return bool(m.sample_step_into(None, out, push=push))
        return bool(m.sample_step_into(observation, out))

    @torch.no_grad()
    def value_into(self, binding, dst_stage, bootstrap_out):
        """Bootstrap value at ``t = T`` through the step's fused kernels (the sampler's tail):
        ``binding`` as for ``step_into`` with ``push`` set; the rebuilt observations go to
        ``dst_stage``, the values to ``bootstrap_out [Bg]``.  False: not supported here."""
        m = self.sampling_model
        push = getattr(binding, "push", None)
        if self.uses_prev_inputs or push is None or not hasattr(m, "sample_value_into"):
            return False
        out = _HeadOut(prob_rows=None, value_rows=None, action_rows=None, action_out=None,
                       uniforms=None, t_dev=binding.t_dev, lo=binding.lo)
        return bool(m.sample_value_into(out, push, dst_stage, bootstrap_out))

    @torch.no_grad()
    def value(self, observation, prev_action, prev_reward):
        if self.uses_prev_inputs and prev_action is not None:
            prev_action = self.distribution.to_onehot(prev_action)
        else:
            prev_action = prev_reward = None
        obs, pa, pr = self._to_model_device(observation, prev_action, prev_reward)
        _pi, value = self.sampling_model(obs, pa, pr)
        return self._out(value)


class RecurrentCategoricalPgAgentBase(BaseAgent):
    """Recurrent categorical policy-gradient agent (rlpyt/agents/pg/categorical.py:54-106):
    the model takes and returns an LSTM state; ``step`` records the state the agent ENTERED the
    step with (``prev_rnn_state``, stored ``[B, N, H]``) so that the algorithm can restart whole
    columns from row 0 of a batch."""

    def __call__(self, observation, prev_action, prev_reward, init_rnn_state):
        """``init_rnn_state`` already ``[N, B, H]``; returns (DistInfo, value, next_rnn_state)."""
        prev_action = self.distribution.to_onehot(prev_action)
        obs, pa, pr = self._to_model_device(observation, prev_action, prev_reward)
        pi, value, next_rnn_state = self.model(obs, pa, pr, init_rnn_state)
        dist_info, value = self._out((DistInfo(prob=pi), value))
        return dist_info, value, next_rnn_state      # the state stays on the device

    def make_distribution(self, n_actions):
        return Categorical(dim=n_actions)

    @torch.no_grad()
    def step(self, observation, prev_action, prev_reward):
        from ...utils.buffer import buffer_func
        prev_action = self.distribution.to_onehot(prev_action)
        obs, pa, pr = self._to_model_device(observation, prev_action, prev_reward)
        pi, value, rnn_state = self.sampling_model(obs, pa, pr, self.prev_rnn_state)
        dist_info = DistInfo(prob=pi)
        action = self.distribution.sample(dist_info, generator=self.sample_generator)
        prev = self.prev_rnn_state
        if prev is None:
            prev = buffer_func(rnn_state, torch.zeros_like)
        # [N,B,H] -> [B,N,H] for storage (categorical.py:84-88); a real copy, because the
        # persistent state buffer is overwritten in place just below
        prev_rnn_state = buffer_func(prev, lambda x: x.transpose(0, 1).clone(
            memory_format=torch.contiguous_format))
        agent_info = AgentInfoRnn(dist_info=dist_info, value=value, prev_rnn_state=prev_rnn_state)
        self.advance_rnn_state(rnn_state)
        return self._out(AgentStep(action=action, agent_info=agent_info))

    @torch.no_grad()
    def value(self, observation, prev_action, prev_reward):
        prev_action = self.distribution.to_onehot(prev_action)
        obs, pa, pr = self._to_model_device(observation, prev_action, prev_reward)
        _pi, value, _rnn_state = self.sampling_model(obs, pa, pr, self.prev_rnn_state)
        return self._out(value)


class RecurrentCategoricalPgAgent(RecurrentAgentMixin, RecurrentCategoricalPgAgentBase):
    pass
