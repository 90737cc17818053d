"""Recurrent DQN agent (rlpyt/agents/dqn/r2d1_agent.py:12-59, atari/atari_r2d1_agent.py): Q-values,
epsilon-greedy draw and the LSTM state all stay in HBM; ``agent_info.prev_rnn_state`` is stored
``[B, N, H]`` as in the reference."""
import os

import torch

from ...models.dqn.atari_r2d1_model import AtariR2d1Model, RnnState
from ...utils.buffer import buffer_func, buffer_method
from ...utils.collections import namedarraytuple
from ..base import AgentStep, RecurrentAgentMixin
from .dqn_agent import DqnAgent

AgentInfo = namedarraytuple("AgentInfo", ["q", "prev_rnn_state"])


class R2d1AgentBase(DqnAgent):
    # set False (or RLPYT_R2D1_FUSED_STEP=0) to keep the eager reset handling + step (A/B tests)
    use_fused_step = os.environ.get("RLPYT_R2D1_FUSED_STEP", "1") != "0"

    def __call__(self, observation, prev_action, prev_reward, init_rnn_state):
        """``init_rnn_state`` already ``[N, B, H]``; returns (q, rnn_state), both on device."""
        prev_action = self.distribution.to_onehot(prev_action)
        obs, pa, pr = self._to_model_device(observation, prev_action, prev_reward)
        q, rnn_state = self.model(obs, pa, pr, init_rnn_state)
        return self._out(q), rnn_state

    @torch.no_grad()
    def step(self, observation, prev_action, prev_reward):
        prev_action = self.distribution.to_onehot(prev_action)
        obs, pa, pr = self._to_model_device(observation, prev_action, prev_reward)
        q, rnn_state = self.sampling_model(obs, pa, pr, self.prev_rnn_state)
        action = self.distribution.sample(q, generator=self.sample_generator,
                                          uniforms=self.sample_uniforms)
        prev = self.prev_rnn_state
        if prev is None:
            prev = buffer_func(rnn_state, torch.zeros_like)
        # [N,B,H] -> [B,N,H] for storage (r2d1_agent.py:37-40); a real copy, because the
        # persistent state buffer is overwritten in place just below
        prev_rnn_state = buffer_func(prev, lambda x: x.transpose(0, 1).clone(
            memory_format=torch.contiguous_format))
        agent_info = AgentInfo(q=q, prev_rnn_state=prev_rnn_state)
        self.advance_rnn_state(rnn_state)
        return self._out(AgentStep(action=action, agent_info=agent_info))

    @torch.no_grad()
    def step_with_reset(self, observation, prev_action, prev_reward, done):
        """``step`` for the HBM sampler with the reset handling of
        rlpyt/samplers/parallel/gpu/action_server.py:49-53 folded in: ``prev_action`` (indices) and
        ``prev_reward`` arrive as stored, ``done [B]`` (None: no resets) names the environments that
        were reset before this step -- the nulling of their previous action / reward, the zeroing
        of their recurrent state and the in-place state update all happen inside the model's fused
        step (``AtariR2d1Model.sample_step``) instead of as a dozen eager launches.  Returns None
        when the fused step does not apply; the caller then does the resets itself and calls
        ``step``."""
        m = self.sampling_model
        ok = getattr(m, "sample_step_ok", None)
        if (not self.use_fused_step or self.host_outputs or self.device.type != "cuda" or ok is None
                or not isinstance(observation, torch.Tensor) or observation.device != self.device
                or not ok(observation, prev_action)):
            return None
        state = self.prev_rnn_state
        if state is None:
            if torch.cuda.is_current_stream_capturing():
                return None          # the state buffers must exist before a step graph is captured
            B, H = observation.shape[0], m.lstm.hidden_size
            state = RnnState(h=torch.zeros((1, B, H), dtype=torch.float32, device=self.device),
                             c=torch.zeros((1, B, H), dtype=torch.float32, device=self.device))
            self._rnn_states[self._slot] = state
        q, prev_h, prev_c = m.sample_step(observation, prev_action, prev_reward, done,
                                          state.h[0], state.c[0])
        action = self.distribution.sample(q, generator=self.sample_generator,
                                          uniforms=self.sample_uniforms)
        # [B, N = 1, H], the reference's storage order (r2d1_agent.py:37-40)
        agent_info = AgentInfo(q=q, prev_rnn_state=RnnState(h=prev_h.unsqueeze(1),
                                                            c=prev_c.unsqueeze(1)))
        return AgentStep(action=action, agent_info=agent_info)

    def target(self, observation, prev_action, prev_reward, init_rnn_state):
        prev_action = self.distribution.to_onehot(prev_action)
        obs, pa, pr = self._to_model_device(observation, prev_action, prev_reward)
        target_q, rnn_state = self.target_model(obs, pa, pr, init_rnn_state)
        return self._out(target_q), rnn_state


class R2d1Agent(RecurrentAgentMixin, R2d1AgentBase):
    pass


class AtariR2d1Agent(R2d1Agent):
    def __init__(self, ModelCls=AtariR2d1Model, **kwargs):
        super().__init__(ModelCls=ModelCls, **kwargs)

    def make_env_to_model_kwargs(self, env_spaces):
        return dict(image_shape=env_spaces.observation.shape,
                    output_size=env_spaces.action.n)
