"""Categorical DQN agent (rlpyt/agents/dqn/catdqn_agent.py:12-46, atari/atari_catdqn_agent.py)."""
import torch

from ...distributions.epsilon_greedy import CategoricalEpsilonGreedy
from ...models.dqn.atari_catdqn_model import AtariCatDqnModel
from ...utils.collections import namedarraytuple
from ..base import AgentStep
from .dqn_agent import AtariDqnAgent, DqnAgent

AgentInfo = namedarraytuple("AgentInfo", ["p"])


class CatDqnAgent(DqnAgent):
    def __init__(self, n_atoms=51, **kwargs):
        super().__init__(**kwargs)
        self.n_atoms = self.model_kwargs["n_atoms"] = n_atoms

    def initialize(self, env_spaces, share_memory=False, global_B=1, env_ranks=None):
        super().initialize(env_spaces, share_memory, global_B, env_ranks)
        # z is a placeholder until the algorithm hands over V_min / V_max
        self.distribution = CategoricalEpsilonGreedy(dim=env_spaces.action.n,
                                                     z=torch.linspace(-1, 1, self.n_atoms))

    def to_device(self, cuda_idx=None):
        super().to_device(cuda_idx)
        self.distribution.set_z(self.distribution.z.to(self.device))

    def give_V_min_max(self, V_min, V_max):
        self.V_min, self.V_max = V_min, V_max
        self.distribution.set_z(torch.linspace(V_min, V_max, self.n_atoms, device=self.device))

    @torch.no_grad()
    def step(self, observation, prev_action, prev_reward):
        prev_action = self.distribution.to_onehot(prev_action)
        obs, pa, pr = self._to_model_device(observation, prev_action, prev_reward)
        p = self.model(obs, pa, pr)
        action = self.distribution.sample(p, generator=self.sample_generator,
                                          uniforms=self.sample_uniforms)
        return self._out(AgentStep(action=action, agent_info=AgentInfo(p=p)))


class AtariCatDqnAgent(CatDqnAgent):
    def __init__(self, ModelCls=AtariCatDqnModel, **kwargs):
        super().__init__(ModelCls=ModelCls, **kwargs)

    make_env_to_model_kwargs = AtariDqnAgent.make_env_to_model_kwargs
