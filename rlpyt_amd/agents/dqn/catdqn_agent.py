"""Categorical DQN agent (rlpyt/agents/dqn/catdqn_agent.py:12-46, atari/atari_catdqn_agent.py)."""
import torch

from ...distributions.epsilon_greedy import CategoricalEpsilonGreedy
from ...models.dqn.atari_catdqn_model import AtariCatDqnModel
from ...utils.collections import namedarraytuple
from ..base import AgentStep
from .dqn_agent import AtariDqnAgent, DqnAgent

AgentInfo = namedarraytuple("AgentInfo", ["p"])


class CatDqnAgent(DqnAgent):
    """Acts greedily on the MEAN of its categorical return distributions: the atom grid ``z``
    belongs to the algorithm (``V_min`` / ``V_max``), which hands it over with ``set_support``;
    until then a unit grid stands in."""

    def __init__(self, n_atoms=51, **kwargs):
        super().__init__(**kwargs)
        self.n_atoms = self.model_kwargs["n_atoms"] = n_atoms

    def make_distribution(self, n_actions):
        return CategoricalEpsilonGreedy(dim=n_actions, z=torch.linspace(-1, 1, self.n_atoms))

    def to_device(self, cuda_idx=None):
        super().to_device(cuda_idx)
        self.distribution.set_z(self.distribution.z.to(self.device))

    def set_support(self, V_min, V_max):
        self.V_min, self.V_max = V_min, V_max
        self.distribution.set_z(torch.linspace(V_min, V_max, self.n_atoms, device=self.device))

    give_V_min_max = set_support       # the reference's name for it (catdqn_agent.py:28)

    @torch.no_grad()
    def step(self, observation, prev_action, prev_reward):
        prev_action = self._onehot(prev_action)
        obs, pa, pr = self._to_model_device(observation, prev_action, prev_reward)
        p = self.model(obs, pa, pr)
        action = self.distribution.sample(p, generator=self.sample_generator,
                                          uniforms=self.sample_uniforms)
        return self._out(AgentStep(action=action, agent_info=AgentInfo(p=p)))


class AtariCatDqnAgent(CatDqnAgent):
    def __init__(self, ModelCls=AtariCatDqnModel, **kwargs):
        super().__init__(ModelCls=ModelCls, **kwargs)

    make_env_to_model_kwargs = AtariDqnAgent.make_env_to_model_kwargs
